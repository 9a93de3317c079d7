"""SVC_MMA_BF16X6 (include/svc_hip.h): fp32-level convolution products and weight gradients on the bf16 matrix instruction — every
fp32 operand taken apart into three bf16 pieces (exactly), the six piece products of weight >= 2^-16 accumulated in fp32.  A precision
mode of FP32 training (`train.mma: "bf16x6"`): the reference's fp32 step (train.py:150-213 without `fp16_run`) is what it must match.

The claim under test is "as exact as the fp32 MFMA kernels": every kernel-level case is computed in FLOAT64 and BOTH engine paths are
measured against it; the x6 error must stay within a small multiple of the fp32 kernel's.  Range: the same cases with operands scaled
by 1e-9 and 1e+6 (an fp16 operand format would flush the first and overflow the second) give the same RELATIVE error.  Model level:
the real reference's fp32 training-step golden, under the bounds of the fp32 test."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _every_shape_through_the_x6_kernels():
    """The dispatcher sends one-tap convolutions and launches of fewer than 224 workgroups to the fp32 kernels (they are faster
    there); the parity cases below want the x6 kernel on every shape."""
    import svc_hip as S
    S.lib().svc_debug_set_sp(1, 0)
    yield
    S.lib().svc_debug_set_sp(2, 224)

CASES = [
    # B, Cin, Cout, T, K, dil           tiling the dispatcher picks (the 16-bit instantiations: tests/test_bf16_gpu.py)
    (16, 192, 192, 768, 1, 1),        # 64 x 192
    (16, 384, 192, 768, 5, 1),        # 64 x 192
    (32, 1024, 1024, 132, 5, 11),     # 128 x 160 (DiscriminatorP period 11)
    (1, 256, 256, 6896, 7, 3),        # 64 x 128
    (2, 128, 128, 20000, 3, 1),       # 128 x 128
    (16, 96, 192, 700, 3, 1),         # Cin = 6 x 16
    (16, 192, 768, 768, 3, 1),        # the prior encoder's FFN
]


def _err(a, exact):
    return (a.double().cpu() - exact).abs().max().item() / max(exact.abs().max().item(), 1e-300)


@pytest.mark.parametrize("B,Cin,Cout,T,K,dil", CASES)
def test_conv1d_x6_is_as_exact_as_the_fp32_kernel(dev, B, Cin, Cout, T, K, dil):
    import svc_hip as S
    g = torch.Generator().manual_seed(B + Cin + Cout + T + K)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, T, generator=g)
    pad = (K * dil - dil) // 2
    xd, wp, bd, resd = x.to(dev), S.pack_conv1d_weight(w.to(dev)), b.to(dev), res.to(dev)
    exact = F.conv1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), dilation=dil, padding=pad) + res.double()
    n0 = S.lib().svc_debug_bf16(-1)
    y = S.conv1d(xd, wp, Cout, K, bias=bd, dil=dil, pad_left=pad, pre_slope=0.1, res=resd, res_mode=1, mma=S.MMA_BF16X6)
    assert S.lib().svc_debug_bf16(-1) == n0 + 1, "the 16-bit-instruction kernel was not the one that ran"
    y32 = S.conv1d(xd, wp, Cout, K, bias=bd, dil=dil, pad_left=pad, pre_slope=0.1, res=resd, res_mode=1)
    e6, e32 = _err(y, exact), _err(y32, exact)
    print(f"conv {Cin}->{Cout} k{K} d{dil} B{B} T{T}: x6 {e6:.2e}, fp32 kernel {e32:.2e} (of max |exact|)")
    assert e6 < 2e-6 and e6 <= 3 * e32 + 1e-7, (e6, e32)
    with S.mma_mode(S.MMA_BF16X6):                                        # the region form of the same switch
        y2 = S.conv1d(xd, wp, Cout, K, bias=bd, dil=dil, pad_left=pad, pre_slope=0.1, res=resd, res_mode=1)
    assert torch.equal(y2, y) and S.current_mma() == S.MMA_F32


@pytest.mark.parametrize("scale", [1e-9, 1e6])
def test_conv1d_x6_keeps_fp32s_exponent_range(dev, scale):
    """Gradients of a GAN step span many decades; bf16 pieces have fp32's exponent, so a tensor at 1e-9 (below fp16's smallest
    subnormal) or 1e+6 (above fp16's largest number) is multiplied as accurately as one at 1."""
    import svc_hip as S
    g = torch.Generator().manual_seed(7)
    B, Cin, Cout, T, K = 16, 192, 192, 768, 5
    x = torch.randn(B, Cin, T, generator=g) * scale
    w = torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5
    exact = F.conv1d(x.double(), w.double(), padding=2)
    y = S.conv1d(x.to(dev), S.pack_conv1d_weight(w.to(dev)), Cout, K, pad_left=2, mma=S.MMA_BF16X6)
    assert _err(y, exact) < 2e-6


@pytest.mark.parametrize("B,Ca,Cb,T,K,dil", [(16, 384, 192, 768, 5, 1), (16, 192, 192, 768, 1, 1), (16, 192, 768, 768, 3, 1),
                                             (8, 128, 128, 1024, 11, 1), (32, 1024, 1024, 132, 5, 11), (3, 100, 70, 333, 7, 2)])
def test_wgrad_x6_is_as_exact_as_the_fp32_kernel(dev, B, Ca, Cb, T, K, dil):
    import svc_hip as S
    g = torch.Generator().manual_seed(Ca + Cb + T + K)
    dy = torch.randn(B, Ca, T, generator=g) * 1e-4          # a gradient's magnitude
    x = torch.randn(B, Cb, T, generator=g)
    pad = (K * dil - dil) // 2
    n0 = S.tlib().svc_debug_wgrad_bf16_launches()
    db = torch.zeros(Ca, device=dev)
    G = S.conv1d_wgrad(dy.to(dev), x.to(dev), K, dil, pad, out=torch.zeros(Ca, Cb, K, device=dev), accumulate=True, dbias=db,
                       mma=S.MMA_BF16X6)
    G32 = S.conv1d_wgrad(dy.to(dev), x.to(dev), K, dil, pad, out=torch.zeros(Ca, Cb, K, device=dev), accumulate=True)
    torch.cuda.synchronize()
    assert S.tlib().svc_debug_wgrad_bf16_launches() > n0
    wz = torch.zeros(Ca, Cb, K, dtype=torch.float64, requires_grad=True)
    (exact,) = torch.autograd.grad(F.conv1d(x.double(), wz, dilation=dil, padding=pad), wz, dy.double())
    e6, e32 = _err(G, exact), _err(G32, exact)
    print(f"wgrad {Ca}x{Cb} k{K} d{dil} B{B} T{T}: x6 {e6:.2e}, fp32 kernel {e32:.2e}")
    assert e6 < 3e-6 and e6 <= 3 * e32 + 1e-7, (e6, e32)
    assert (db.cpu() - dy.sum((0, 2))).abs().max().item() <= 1e-4 * dy.sum((0, 2)).abs().max().item()


def test_autograd_conv_in_x6_mode_matches_fp32_gradients(dev):
    """Forward, input gradient and weight gradient of an op recorded inside `mma_mode(MMA_BF16X6)` all run on the 16-bit
    instruction (launch counters) and agree with the fp32 kernels' to fp32 rounding."""
    import svc_autograd as A
    import svc_hip as S
    g = torch.Generator().manual_seed(11)
    x = torch.randn(16, 192, 768, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(384, 192, 5, generator=g) * 0.03).to(dev).requires_grad_(True)
    b = torch.zeros(384, device=dev, requires_grad=True)
    with S.mma_mode(S.MMA_BF16X6):
        y = A.conv1d(x, w, b, padding=2)
    c0, w0 = S.lib().svc_debug_bf16(-1), S.tlib().svc_debug_wgrad_bf16_launches()
    y.square().sum().backward()
    assert S.lib().svc_debug_bf16(-1) == c0 + 1 and S.tlib().svc_debug_wgrad_bf16_launches() == w0 + 1
    gx, gw, y6 = x.grad.clone(), w.grad.clone(), y.detach().clone()
    x.grad = w.grad = b.grad = None
    y = A.conv1d(x, w, b, padding=2)
    y.square().sum().backward()
    for name, a_, b_ in (("y", y6, y.detach()), ("dx", gx, x.grad), ("dw", gw, w.grad)):
        rel = (a_ - b_).abs().max().item() / b_.abs().max().item()
        assert rel < 1e-5, (name, rel)            # (two fp32-level results of a 960- / 12 288-term reduction, each ~1e-6 from exact; bf16 operands: 1e-2)


def test_training_step_in_x6_mode_meets_the_fp32_bounds_against_the_reference(dev):
    """The REAL reference's fp32 training step (tests/golden via train_common.load_case: losses, y_hat, gradient norms) against the
    engine's step with every convolution and conv gradient in x6 mode — held to the bounds of the fp32 test (test_train_gpu)."""
    import numpy as np
    from train_common import LOSS_KEYS, load_case
    from test_train_gpu import _build, _step
    import svc_hip as S
    cs = load_case()
    z = cs["z"]
    net_g, net_d = _build(cs, dev)
    n0 = S.lib().svc_debug_bf16(-1) + S.tlib().svc_debug_wgrad_bf16_launches()
    with S.mma_mode(S.MMA_BF16X6):
        out = _step(cs, net_g, net_d, dev)
    for k in LOSS_KEYS:
        ref, got = float(z["loss." + k]), float(out[k])
        assert abs(got - ref) <= 1e-4 * max(1.0, abs(ref)), (k, got, ref)
    yh = out["y_hat"].detach().cpu().numpy()
    assert np.abs(yh - z["y_hat"]).max() <= 2e-4 * max(1.0, np.abs(z["y_hat"]).max())
    out["loss_disc"].backward(retain_graph=True)
    gd = {k: p.grad.detach().norm().item() for k, p in net_d.named_parameters()}
    net_d.zero_grad()
    out["loss_gen_all"].backward()
    gg = {k: p.grad.detach().norm().item() for k, p in net_g.named_parameters() if p.grad is not None}
    n1 = S.lib().svc_debug_bf16(-1) + S.tlib().svc_debug_wgrad_bf16_launches()
    assert n1 > n0, "no convolution of the step ran on the 16-bit instruction"
    for got, keys, ref in ((gd, z["gnorm_d_keys"], z["gnorm_d"]), (gg, z["gnorm_g_keys"], z["gnorm_g"])):
        rel = sorted(abs(got[str(k)] - n) / max(n, 1e-5) for k, n in zip(keys, ref) if not str(k).endswith("conv_k.bias"))
        med, p90 = rel[len(rel) // 2], rel[int(0.9 * len(rel))]
        print(f"gradient norms vs the reference's fp32 step: median rel {med:.2e}, p90 {p90:.2e}, max {rel[-1]:.2e}")
        assert rel[-1] <= 2e-3, (med, p90, rel[-1])           # the fp32 test's bound on every gradient norm


def test_train_step_object_honours_train_mma(dev):
    """train.TrainStep: `train.mma: "bf16x6"` (fp16_run false) -> MMA_BF16X6 inside the regions, no loss scaler, graph replay allowed."""
    import svc_hip as S
    import train as TR
    from test_train_gpu import _bench_like_items
    hps, items = _bench_like_items(dev)(False, "bf16")
    hps["train"]["mma"] = "bf16x6"
    net_g, net_d, og, od = TR.build(hps, dev)
    step = TR.TrainStep(hps, net_g, net_d, og, od)
    assert step.mma == S.MMA_BF16X6 and step.scaler is None
    n0 = S.lib().svc_debug_bf16(-1) + S.tlib().svc_debug_wgrad_bf16_launches()
    out = step(items)
    torch.cuda.synchronize()
    assert S.lib().svc_debug_bf16(-1) + S.tlib().svc_debug_wgrad_bf16_launches() > n0
    assert all(torch.isfinite(v) for v in out.values() if torch.is_tensor(v))
    og.release(); od.release()


def test_x6_dispatch_keeps_small_launches_on_the_fp32_kernels(dev):
    """Default rule: the split-structure kernel takes launches of at least 224 workgroups and two taps; anything else runs the
    fp32 kernel of the same shape — bit-equal to fp32 mode."""
    import svc_hip as S
    S.lib().svc_debug_set_sp(2, 224)
    for (B, Cin, Cout, T, K), expect in (((16, 192, 384, 768, 5), True), ((16, 192, 192, 768, 1), False), ((2, 768, 192, 768, 3), False)):
        x = torch.randn(B, Cin, T, device=dev)
        wp = S.pack_conv1d_weight(torch.randn(Cout, Cin, K, device=dev) * 0.05)
        n0 = S.lib().svc_debug_bf16(-1)
        y = S.conv1d(x, wp, Cout, K, pad_left=K // 2, mma=S.MMA_BF16X6)
        assert (S.lib().svc_debug_bf16(-1) > n0) == expect, (B, Cin, Cout, T, K)
        if not expect:
            assert torch.equal(y, S.conv1d(x, wp, Cout, K, pad_left=K // 2))
