"""bf16 operands on the matrix pipe (svc_conv1d_args.mma / svc_wgrad_args.mma = SVC_MMA_BF16: the engine's form of the reference's
`fp16_run` + `half_type: bf16` autocast mode, train.py:114,143,166,187,198) — kernel-level parity.

bf16 x bf16 products are exact in fp32 and the kernels accumulate in fp32, so the bf16 path must equal an fp32 convolution of
the operands ROUNDED TO BF16 (round to nearest even, what torch's `.bfloat16()` does) up to fp32 summation order: tolerance 2e-5
of the output scale — three orders of magnitude below the bf16 rounding itself (2^-8), which the second assertion shows is really
there.  Shapes cover every tiling the bf16 instantiations exist for (64x192, 128x160, 64x128, 128x128, 64x256)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _r(t, fmt="bf16"):
    return t.bfloat16().float() if fmt == "bf16" else t.half().float()


FORMATS = ["bf16", "fp16"]


def _mma(S, fmt):
    return S.MMA_BF16 if fmt == "bf16" else S.MMA_F16


CASES = [
    # B, Cin, Cout, T, K, dil           tiling the dispatcher picks
    (16, 192, 192, 768, 1, 1),        # 64 x 192
    (16, 384, 192, 768, 5, 1),        # 64 x 192
    (16, 192, 384, 768, 5, 2),        # 64 x 192
    (32, 1024, 1024, 132, 5, 11),     # 128 x 160 (DiscriminatorP period 11)
    (1, 256, 256, 6896, 7, 3),        # 64 x 128
    (2, 128, 128, 20000, 3, 1),       # 128 x 128
    (2, 128, 128, 20000, 5, 1),       # 128 x 128, 5 taps: 104 KiB of chunk buffers
    (16, 96, 192, 700, 3, 1),         # Cin = 6 x 16
    (16, 192, 768, 768, 3, 1),        # the prior encoder's FFN
]


@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("B,Cin,Cout,T,K,dil", CASES)
def test_conv1d_bf16_equals_fp32_conv_of_bf16_rounded_operands(dev, B, Cin, Cout, T, K, dil, fmt):
    import svc_hip as S
    _r = lambda t: globals()["_r"](t, fmt)        # noqa: E731  (fp16: 11 significand bits instead of 8 — the same statement holds)
    lo = 1e-4 if fmt == "bf16" else 1e-5
    g = torch.Generator().manual_seed(B + Cin + Cout + T + K)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, T, generator=g)
    pad = (K * dil - dil) // 2
    xd, wp, bd, resd = x.to(dev), S.pack_conv1d_weight(w.to(dev)), b.to(dev), res.to(dev)
    n0 = S.lib().svc_debug_bf16(-1)
    for name, kw, ref_fn in (
            ("plain", dict(), lambda: F.conv1d(_r(x), _r(w), b, dilation=dil, padding=pad)),
            ("lrelu-in + residual", dict(pre_slope=0.1, res=resd, res_mode=1),
             lambda: F.conv1d(_r(F.leaky_relu(x, 0.1)), _r(w), b, dilation=dil, padding=pad) + res)):
        y = S.conv1d(xd, wp, Cout, K, bias=bd, dil=dil, pad_left=pad, mma=_mma(S, fmt), **kw)
        y32 = S.conv1d(xd, wp, Cout, K, bias=bd, dil=dil, pad_left=pad, **kw)
        torch.cuda.synchronize()
        ref = ref_fn()
        scale = ref.abs().max().item()
        assert (y.cpu() - ref).abs().max().item() <= 2e-5 * scale, (name, (y.cpu() - ref).abs().max().item(), scale)
        d = (y - y32).abs().max().item()
        assert lo * scale < d < 3e-2 * scale, (name, d, scale)         # it IS a 16-bit-operand computation, and only that far from fp32
    assert S.lib().svc_debug_bf16(-1) - n0 == 2, "the bf16 kernel was not the one that ran"
    with S.mma_mode(_mma(S, fmt)):                                        # the region form of the same switch
        y2 = S.conv1d(xd, wp, Cout, K, bias=bd, dil=dil, pad_left=pad, pre_slope=0.1, res=resd, res_mode=1)
    assert torch.equal(y2, y) and S.current_mma() == S.MMA_F32


def test_conv1d_bf16_request_falls_back_to_fp32_where_no_kernel_exists(dev):
    """Cin not a multiple of 16 (the posterior encoder's 1025-bin input), unaligned rows: fp32, bit-equal to the default."""
    import svc_hip as S
    # ... and 11 taps on a 128-row tile: 16 channels of it are 25 LDS-DMA pieces per wave, the issue code holds 16
    for B, Cin, Cout, T, K in ((2, 1025, 192, 400, 1), (2, 192, 192, 399, 5), (1, 100, 64, 512, 3), (2, 128, 128, 20000, 11)):
        x = torch.randn(B, Cin, T, device=dev)
        wp = S.pack_conv1d_weight(torch.randn(Cout, Cin, K, device=dev) * 0.05)
        n0 = S.lib().svc_debug_bf16(-1)
        a = S.conv1d(x, wp, Cout, K, pad_left=K // 2, mma=S.MMA_BF16)
        assert S.lib().svc_debug_bf16(-1) == n0
        assert torch.equal(a, S.conv1d(x, wp, Cout, K, pad_left=K // 2))


@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("B,Ca,Cb,T,K,dil", [(16, 384, 192, 768, 5, 1), (16, 192, 192, 768, 1, 1), (16, 192, 768, 768, 3, 1),
                                             (8, 128, 128, 1024, 11, 1), (32, 1024, 1024, 132, 5, 11), (3, 100, 70, 333, 7, 2)])
def test_wgrad_bf16_equals_fp32_wgrad_of_bf16_rounded_operands(dev, B, Ca, Cb, T, K, dil, fmt):
    import svc_hip as S
    _r = lambda t: globals()["_r"](t, fmt)        # noqa: E731
    lo = 1e-4 if fmt == "bf16" else 1e-5
    g = torch.Generator().manual_seed(Ca + Cb + T + K)
    dy = torch.randn(B, Ca, T, generator=g)
    x = torch.randn(B, Cb, T, generator=g)
    pad = (K * dil - dil) // 2
    n0 = S.tlib().svc_debug_wgrad_bf16_launches()
    db = torch.zeros(Ca, device=dev)
    G = S.conv1d_wgrad(dy.to(dev), x.to(dev), K, dil, pad, out=torch.zeros(Ca, Cb, K, device=dev), accumulate=True, dbias=db,
                       mma=_mma(S, fmt))
    G32 = S.conv1d_wgrad(dy.to(dev), x.to(dev), K, dil, pad, out=torch.zeros(Ca, Cb, K, device=dev), accumulate=True)
    torch.cuda.synchronize()
    assert S.tlib().svc_debug_wgrad_bf16_launches() > n0
    xr = _r(x).double().requires_grad_(False)
    w = torch.zeros(Ca, Cb, K, dtype=torch.float64, requires_grad=True)
    y = F.conv1d(xr, w, dilation=dil, padding=pad)
    (ref,) = torch.autograd.grad(y, w, _r(dy).double())
    scale = ref.abs().max().item()
    assert (G.cpu().double() - ref).abs().max().item() <= 3e-5 * scale
    d = (G - G32).abs().max().item()
    assert lo * scale < d < 3e-2 * scale, (d, scale)
    assert (db.cpu() - dy.sum((0, 2))).abs().max().item() <= 1e-4 * dy.sum((0, 2)).abs().max().item()      # bias gradient stays fp32


def test_autograd_conv_uses_its_forward_mode_in_backward(dev):
    """An op recorded inside `mma_mode(MMA_BF16)` back-propagates with bf16 operands after the region was left (autocast's
    rule: backward ops run in the dtype of their forward), one recorded outside stays fp32."""
    import svc_autograd as A
    import svc_hip as S
    x = torch.randn(16, 192, 768, device=dev, requires_grad=True)
    w = (torch.randn(384, 192, 5, device=dev) * 0.03).requires_grad_(True)
    b = torch.zeros(384, device=dev, requires_grad=True)
    with S.mma_mode(S.MMA_BF16):
        y = A.conv1d(x, w, b, padding=2)
    c0, w0 = S.lib().svc_debug_bf16(-1), S.tlib().svc_debug_wgrad_bf16_launches()
    y.square().sum().backward()
    assert S.lib().svc_debug_bf16(-1) == c0 + 1 and S.tlib().svc_debug_wgrad_bf16_launches() == w0 + 1
    gx, gw = x.grad.clone(), w.grad.clone()
    x.grad = w.grad = b.grad = None
    y = A.conv1d(x, w, b, padding=2)
    c0 = S.lib().svc_debug_bf16(-1)
    y.square().sum().backward()
    assert S.lib().svc_debug_bf16(-1) == c0
    for a_, b_ in ((gx, x.grad), (gw, w.grad)):
        rel = (a_ - b_).abs().max().item() / b_.abs().max().item()
        assert 1e-5 < rel < 5e-2, rel


@pytest.mark.parametrize("fmt", FORMATS)
def test_training_step_in_bf16_mode_stays_within_the_references_own_autocast_noise(dev, fmt):
    """The REAL reference's `fp16_run: true, half_type: bf16` step (tests/golden/train_amp_bf16_small.npz, made by
    make_golden_train_amp.py under torch.autocast(dtype=bfloat16) with the regions of train.py:166-211) against the engine's
    step with the same regions as svc_hip.mma_mode(MMA_BF16).  The two cannot agree to better than bf16 rounding: autocast
    also rounds every conv OUTPUT and the activations between the ops to bf16 (y_hat, feature maps and mels are bf16 tensors
    there), the engine only the matrix operands.  The golden records how far the reference's own bf16 step is from its fp32
    step (`amp_vs_fp32.*`: losses up to 4e-3, y_hat 7e-3, gradient norms median 4e-3 / p90 2.5e-2); the engine's distance from the
    golden must stay within 3x that noise (+ 2e-3), i.e. the engine is as close to the autocast step as the autocast step is to
    fp32 — and, having fewer roundings, it must be CLOSER to the fp32 golden than the autocast step is."""
    import json
    import os
    import numpy as np
    from train_common import G, LOSS_KEYS, load_case
    from test_train_gpu import _build, _step
    import svc_hip as S
    cs = load_case()
    z = np.load(os.path.join(G, f"train_amp_{fmt}_small.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    assert meta["half_type"] == fmt
    # fp16: the golden was taken at the loss scale the reference's GradScaler settles on for this step (128: larger scales
    # overflow ITS fp16 tensors and are skipped); the engine scales the same way — its tensors are fp32, only operands are fp16
    ls = float(meta.get("loss_scale", 1.0))
    net_g, net_d = _build(cs, dev)
    with S.mma_mode(_mma(S, fmt)):
        out = _step(cs, net_g, net_d, dev)
    z32 = cs["z"]
    for k in LOSS_KEYS:
        ref, noise = float(z["loss." + k]), float(z["amp_vs_fp32." + k])
        got = float(out[k])
        assert abs(got - ref) <= (3 * noise + 2e-3) * max(1.0, abs(ref)), (k, got, ref, noise)
        assert abs(got - float(z32["loss." + k])) <= (noise + 2e-3) * max(1.0, abs(ref)), ("vs fp32", k, got)
    yh = out["y_hat"].detach().cpu().numpy()
    ny = float(z["amp_vs_fp32_y_hat"])
    assert np.abs(yh - z["y_hat"]).max() <= (3 * ny + 2e-3) * max(1.0, np.abs(z["y_hat"]).max())
    (out["loss_disc"] * ls).backward(retain_graph=True)
    gd = {k: p.grad.detach().norm().item() / ls for k, p in net_d.named_parameters()}
    rel_d = sorted(abs(gd[str(k)] - n) / max(n, 1e-6) for k, n in zip(z["gnorm_d_keys"], z["gnorm_d"]))
    net_d.zero_grad()
    (out["loss_gen_all"] * ls).backward()
    gg = {k: p.grad.detach().norm().item() / ls for k, p in net_g.named_parameters() if p.grad is not None}
    rel_g = sorted(abs(gg[str(k)] - n) / max(n, 1e-5) for k, n in zip(z["gnorm_g_keys"], z["gnorm_g"]) if not str(k).endswith("conv_k.bias"))
    for rel, noise in ((rel_g, z["amp_vs_fp32_gnorm_g"]), (rel_d, z["amp_vs_fp32_gnorm_d"])):
        med, p90 = rel[len(rel) // 2], rel[int(0.9 * len(rel))]
        assert med <= 3 * noise[0] + 2e-3 and p90 <= 3 * noise[1] + 2e-3, (med, p90, list(noise))


def test_train_step_object_honours_fp16_run_half_type(dev):
    """train.TrainStep: `fp16_run: true` -> 16-bit operands inside the autocast regions (launches counted): `half_type: bf16`
    as is, `half_type: fp16` with the GradScaler rule (a LossScaler, eager launches even when graphs are enabled);
    `fp16_run: false` -> fp32, no 16-bit launch."""
    import synthetic_data as W
    import svc_hip as S
    import train as TR
    from test_train_gpu import _bench_like_items
    base = _bench_like_items(dev)
    for fp16_run, half, expect in ((True, "bf16", True), (True, "fp16", True), (False, "bf16", False)):
        hps, items = base(fp16_run, half)
        net_g, net_d, og, od = TR.build(hps, dev)
        step = TR.TrainStep(hps, net_g, net_d, og, od)
        assert (step.scaler is not None) == (fp16_run and half == "fp16")
        n0 = S.lib().svc_debug_bf16(-1) + S.tlib().svc_debug_wgrad_bf16_launches()
        out = step(items)
        torch.cuda.synchronize()
        n1 = S.lib().svc_debug_bf16(-1) + S.tlib().svc_debug_wgrad_bf16_launches()
        assert (n1 > n0) == expect, (fp16_run, half, n1 - n0)
        assert all(torch.isfinite(v) for v in out.values() if torch.is_tensor(v))
        og.release(); od.release()


def test_fp16_mode_skips_overflowing_steps_and_backs_the_scale_off(dev):
    """GradScaler's rule on the engine (train.py:192-213; optim.LossScaler): starting from 65536 the first iterations overflow
    fp16 operands (a scaled gradient above 65504 becomes inf), those optimizer steps are SKIPPED — parameters bit-identical —
    and the scale halves once per such iteration until a step goes through; from then on parameters move and stay finite."""
    import train as TR
    from test_train_gpu import _bench_like_items
    hps, items = _bench_like_items(dev)(True, "fp16")
    net_g, net_d, og, od = TR.build(hps, dev)
    step = TR.TrainStep(hps, net_g, net_d, og, od).enable_graph(True)        # graphs asked for: the fp16 mode must stay eager
    p0 = og.arena.param.clone()
    scales, moved_g, moved_d, skipped = [], [], [], []
    for _ in range(16):
        bg, bd = og.arena.param.clone(), od.arena.param.clone()
        out = step(items)
        scales.append(out["loss_scale"])
        moved_g.append(bool((og.arena.param != bg).any()))
        moved_d.append(bool((od.arena.param != bd).any()))
        skipped.append(step.scaler.skipped)
    assert not step._graphs
    prev = 65536.0
    for i, sc in enumerate(scales):
        n_skip = skipped[i] - (skipped[i - 1] if i else 0)
        assert sc == (prev * 0.5 if n_skip else prev), (i, scales, skipped)          # halves exactly when a step of the iteration was skipped
        assert n_skip == (not moved_g[i]) + (not moved_d[i]), (i, moved_g, moved_d, skipped)   # a skipped step leaves its parameters bit-identical
        prev = sc
    assert scales[0] < 65536.0, "a loss scale of 65536 cannot survive fp16 operands here (a scaled gradient above 65504 is inf)"
    assert all(moved_g[-3:]) and all(moved_d[-3:]) and len(set(scales[-3:])) == 1, (scales, moved_g, moved_d)
    assert torch.isfinite(og.arena.param).all() and torch.isfinite(od.arena.param).all() and (og.arena.param != p0).any()
    og.release(); od.release()
