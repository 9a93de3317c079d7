"""The split pipeline (csrc/conv1d_hl.hip): the generator's convolutions (vdecoder/hifigan/models.py:41-67,340-342,378,390-392) on
the fp16 matrix instruction at fp32-level precision — every value as a hi and a lo fp16 plane, every product as three instructions.

The claim under test is "as close to the exact result as the fp32 MFMA kernels are": every kernel-level case computes the
convolution in FLOAT64 (torch, CPU) and measures BOTH engine paths against it — the split kernel's error must stay within a small
multiple of the fp32 kernel's and below 2e-6 of the output scale (an fp16 pipeline sits at 1e-3).  Model level: the full template
through `SynthesizerTrn.split_f16()` against the real reference's fp32 output (tests/golden/infer_full_T24.npz) under the SAME
bound the fp32 path is held to, and against the engine's own fp32 path at the benchmarked shape."""
import json
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import svc_oracle as O
from oracle import weights as W

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SCALE_BOUND = 2e-6      # max |err| / max |exact|
# "as exact as the fp32 kernel": within 4x the fp32 kernel's own distance from float64 — or within 1.2e-6 where the fp32 launch is
# the short-sequence kernel, which since round 6 sums on four accumulators (pairwise-like, ~2e-7: closer to float64 than the
# chained fp32 sum of the other kernels, ~5e-7, that the criterion was written against)
CHAIN_FLOOR = 1.2e-6


def _err(a, exact):
    return (a.double() - exact).abs().max().item() / max(exact.abs().max().item(), 1e-30)


def test_split_planes_carry_22_bits(dev):
    import svc_hip as S
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 24, 301, generator=g) * torch.logspace(-3, 2, 301)          # five decades of magnitude
    add = torch.randn(2, 24, 301, generator=g)
    xh = S.to_h(x.to(dev), split=True)
    assert xh.shape == (2, 2, 3, 301, 8) and xh.dtype == torch.float16
    xs = 32.0 * x.view(2, 3, 8, 301).permute(0, 1, 3, 2)                             # activation planes hold 32 v (csrc/conv1d_hl.hip, ASC)
    hi = xs.half()
    assert torch.equal(xh[0].cpu(), hi)                                              # plane 0 = the stored value rounded to fp16
    assert torch.equal(xh[1].cpu(), (xs - hi.float()).half())
    back = S.from_h(xh).cpu()
    assert ((back - x).abs() <= x.abs() * 2.0 ** -21 + 2.0 ** -29).all()
    back2 = S.from_h(S.to_h(x.to(dev), add=add.to(dev), split=True)).cpu()
    assert ((back2 - (x + add)).abs() <= (x + add).abs() * 2.0 ** -21 + 2.0 ** -29).all()


CONV_CASES = [
    # B, Cin, Cout, T, KS, dil       (every tile form; channel chunking: 256 channels x 178 columns is four chunks)
    (1, 256, 256, 300, 11, 5), (1, 128, 128, 1000, 7, 3), (2, 64, 64, 515, 3, 1), (1, 32, 32, 2100, 11, 1), (1, 16, 16, 4099, 7, 5),
    (1, 16, 16, 37, 3, 3), (2, 128, 128, 129, 3, 5), (1, 256, 256, 6896, 7, 1), (1, 64, 64, 20000, 11, 3), (1, 32, 16, 700, 1, 1),
    (1, 48, 80, 260, 3, 1), (1, 512, 256, 200, 7, 1),
]


@pytest.mark.parametrize("B,Cin,Cout,T,KS,dil", CONV_CASES)
def test_conv1d_split_is_as_exact_as_the_fp32_kernel(dev, B, Cin, Cout, T, KS, dil):
    import svc_hip as S
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + T + KS)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5
    b = torch.randn(Cout, generator=g)
    pad = (KS * dil - dil) // 2
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    exact = F.conv1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), dilation=dil, padding=pad)
    xh = S.to_h(xd, split=True)
    wp = S.pack_conv1d_h(wd, split=True)
    assert wp.shape[0] == 2 and wp.dim() == 5
    y = S.from_h(S.conv1d_h(xh, wp, Cout, bias=bd, dil=dil, pad_left=pad, pre_slope=0.1)).cpu()
    y32 = S.conv1d(xd, S.pack_conv1d_weight(wd), Cout, KS, bias=bd, dil=dil, pad_left=pad, pre_slope=0.1).cpu()
    e_split, e_f32 = _err(y, exact), _err(y32, exact)
    print(f"conv {Cin}->{Cout} k{KS} d{dil} T{T}: split {e_split:.2e}, fp32 kernel {e_f32:.2e} (of max |exact|)")
    assert y.shape == exact.shape
    assert e_split < SCALE_BOUND and e_split < max(4 * e_f32 + 2e-7, CHAIN_FLOOR), (e_split, e_f32)
    # leaky_relu behind; residual + accumulate / divide epilogue of the MRF mean (:382-389)
    y2 = S.from_h(S.conv1d_h(xh, wp, Cout, bias=bd, dil=dil, pad_left=pad, pre_slope=0.1, post_slope=0.1)).cpu()
    assert _err(y2, F.leaky_relu(exact, 0.1)) < SCALE_BOUND
    if Cin == Cout:
        old = torch.randn(B, Cout, T, generator=g)
        out = S.to_h(old.to(dev), split=True)
        S.conv1d_h(xh, wp, Cout, bias=bd, dil=dil, pad_left=pad, pre_slope=0.1, res=xh, out=out, beta=1.0, out_div=3.0)
        assert _err(S.from_h(out).cpu(), (old.double() + exact + x.double()) / 3) < SCALE_BOUND


@pytest.mark.parametrize("B,C,T,KS,d1", [(2, 64, 515, 7, 3), (1, 64, 3000, 11, 5), (1, 32, 2100, 3, 1), (1, 16, 4099, 11, 3), (1, 16, 37, 7, 5),
                                         (1, 48, 129, 11, 1), (1, 32, 30000, 7, 5), (1, 64, 260, 3, 5), (1, 128, 1000, 11, 5), (2, 128, 300, 3, 1),
                                         (1, 96, 515, 7, 3), (1, 128, 20000, 7, 1)])
def test_resblock_pair_split_vs_float64_and_two_launches(dev, B, C, T, KS, d1):
    """svc_resblock_pair_hl (one launch, the intermediate's two planes in LDS, the input staged in channel chunks) against the pair in
    float64 and against the two svc_conv1d_hl launches it replaces; with the MRF accumulate / divide epilogue; tile borders, sequence
    ends shorter than a halo, channel counts that leave row tiles partly empty."""
    import svc_hip as S
    g = torch.Generator().manual_seed(C + T + KS)
    x = torch.randn(B, C, T, generator=g)
    w1 = torch.randn(C, C, KS, generator=g) / (C * KS) ** 0.5
    w2 = torch.randn(C, C, KS, generator=g) / (C * KS) ** 0.5
    b1, b2 = torch.randn(C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.3
    old = torch.randn(B, C, T, generator=g)
    p1, p2 = (KS - 1) * d1 // 2, (KS - 1) // 2
    mid = F.leaky_relu(F.conv1d(F.leaky_relu(x.double(), 0.1), w1.double(), b1.double(), dilation=d1, padding=p1), 0.1)
    exact = F.conv1d(mid, w2.double(), b2.double(), padding=p2) + x.double()
    xh = S.to_h(x.to(dev), split=True)
    w1p, w2p = S.pack_conv1d_h(w1.to(dev), split=True), S.pack_conv1d_h(w2.to(dev), split=True)
    xt = S.conv1d_h(xh, w1p, C, bias=b1.to(dev), dil=d1, pad_left=p1, pre_slope=0.1, post_slope=0.1)
    two = S.to_h(old.to(dev), split=True)
    S.conv1d_h(xt, w2p, C, bias=b2.to(dev), pad_left=p2, res=xh, out=two, beta=1.0, out_div=3.0)
    one = S.to_h(old.to(dev), split=True)
    S.resblock_pair_h(xh, w1p, b1.to(dev), w2p, b2.to(dev), d1, out=one, beta=1.0, out_div=3.0)
    a, b_ = S.from_h(one).cpu(), S.from_h(two).cpu()
    ref = (old.double() + exact) / 3
    print(f"pair C={C} k{KS} d{d1} T{T}: fused {_err(a, ref):.2e}, two launches {_err(b_, ref):.2e}")
    assert _err(a, ref) < SCALE_BOUND and _err(b_, ref) < SCALE_BOUND
    plain = S.from_h(S.resblock_pair_h(xh, w1p, b1.to(dev), w2p, b2.to(dev), d1)).cpu()
    assert _err(plain, exact) < SCALE_BOUND


@pytest.mark.parametrize("B,Cin,L,K,u", [(1, 256, 300, 16, 8), (2, 128, 515, 4, 2), (1, 64, 1000, 4, 2), (1, 32, 2077, 4, 2),
                                         (1, 256, 6896, 16, 8), (1, 32, 97, 8, 4)])
def test_conv_transpose1d_split(dev, B, Cin, L, K, u):
    """ups[i] (vdecoder/hifigan/models.py:340-342,377-381): leaky_relu + ConvTranspose1d + the noise-conv addend."""
    import svc_hip as S
    Cout = Cin // 2
    g = torch.Generator().manual_seed(Cin + L + K)
    x = torch.randn(B, Cin, L, generator=g)
    w = torch.randn(Cin, Cout, K, generator=g) / (Cin * K / u) ** 0.5
    b = torch.randn(Cout, generator=g)
    pad = (K - u + 1) // 2
    exact = F.conv_transpose1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), stride=u, padding=pad)
    add = torch.randn(exact.shape, generator=g)
    wp = S.pack_conv1d_h(w.to(dev), u=u, split=True)
    xh = S.to_h(x.to(dev), split=True)
    y = S.from_h(S.conv_transpose1d_h(xh, wp, Cout, K, u, pad, bias=b.to(dev), pre_slope=0.1, res=S.to_h(add.to(dev), split=True))).cpu()
    assert y.shape == exact.shape
    assert _err(y, exact + add.double()) < SCALE_BOUND


def test_conv_post_split(dev):
    import svc_hip as S
    g = torch.Generator().manual_seed(5)
    B, Cc, T = 2, 16, 3001
    x = torch.randn(B, Cc, T, generator=g) * 2
    w = torch.randn(1, Cc, 7, generator=g) / (Cc * 7) ** 0.5
    b = torch.randn(1, generator=g) * 0.1
    exact = torch.tanh(F.conv1d(F.leaky_relu(x.double(), 0.01), w.double(), b.double(), padding=3))
    y = S.conv_post_h(S.to_h(x.to(dev), split=True), w.to(dev).reshape(Cc, 7), b.to(dev), 7, 3, pre_slope=0.01).cpu()
    assert y.shape == exact.shape and y.dtype == torch.float32
    assert (y.double() - exact).abs().max().item() < 1e-6


def test_split_and_plain_tensors_do_not_mix(dev):
    import svc_hip as S
    x = torch.randn(1, 16, 64).to(dev)
    w = torch.randn(16, 16, 3).to(dev)
    with pytest.raises(S.SvcError):
        S.conv1d_h(S.to_h(x, split=True), S.pack_conv1d_h(w), 16, pad_left=1)
    with pytest.raises(S.SvcError):
        S.conv1d_h(S.to_h(x), S.pack_conv1d_h(w, split=True), 16, pad_left=1)
    with pytest.raises(S.SvcError):
        S.resblock_pair_h(S.to_h(x, split=True), S.pack_conv1d_h(w), torch.zeros(16, device=dev), S.pack_conv1d_h(w),
                          torch.zeros(16, device=dev), 1)
    x256, w256 = torch.randn(1, 256, 64).to(dev), S.pack_conv1d_h(torch.randn(256, 256, 3).to(dev), split=True)
    with pytest.raises(S.SvcError):                        # the fused split pair is built for up to 128 channels
        S.resblock_pair_h(S.to_h(x256, split=True), w256, torch.zeros(256, device=dev), w256, torch.zeros(256, device=dev), 1)


def _build(cfg, seed, dev):
    import models
    kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
    net = models.SynthesizerTrn(cfg["spec_channels"], cfg["segment_size"], **kw)
    sd = W.make_state_dict(cfg, seed)
    net.load_state_dict(sd, strict=True)
    return net.to(dev).eval(), sd


def test_split_inference_meets_the_fp32_bound_against_the_reference(dev):
    """Full template, T = 24: the real reference's fp32 output (infer_full_T24.npz) — the split mode is held to the bound of the fp32
    path (2e-4 of the waveform's largest sample) and must sit within a few of the fp32 path's own distance; `float()` returns to
    the fp32 kernels bit for bit."""
    z = np.load(os.path.join(G, "infer_full_T24.npz"))
    meta = json.loads(str(np.load(os.path.join(G, "infer_full_T24_half.npz"))["meta"]))
    net, _ = _build(W.full_config(), meta["seed"], dev)
    t = lambda k: torch.from_numpy(z[k]).to(dev)
    noise = dict(enc_p=t("noise_enc_p"), rand_ini=t("noise_rand_ini"), sine=t("noise_sine"))
    run = lambda: net.infer(t("c"), t("f0"), t("uv"), g=t("sid"), noice_scale=meta["noice_scale"], noise=noise)[0]
    o32 = run()
    net.split_f16()
    assert net.dec.half_mode == "split" and next(net.parameters()).dtype == torch.float32
    os_ = run()
    ref = torch.from_numpy(z["o"])
    d32, dsp = (o32.cpu() - ref).abs().max().item(), (os_.cpu() - ref).abs().max().item()
    print(f"full template T=24 vs the reference's fp32 waveform: fp32 kernels max|err| {d32:.3e}, split pipeline {dsp:.3e}, "
          f"split vs fp32 kernels {(os_ - o32).abs().max().item():.3e}; max|ref| {ref.abs().max().item():.3f}")
    assert os_.shape == o32.shape and os_.dtype == torch.float32
    assert dsp <= 2e-4 * ref.abs().max().item()
    assert dsp <= 4 * d32 + 2e-6
    net.float()
    assert torch.equal(run(), o32)


def test_split_inference_at_the_benchmarked_shape(dev):
    """BASELINE configs[1] (B = 1, T = 862) against the fp32 CPU oracle, next to the fp32 kernels on the same inputs; hipGraph replay
    bit-equal to the eager launches."""
    import bench
    cfg = W.full_config()
    net, sd = _build(cfg, 1234, dev)
    B, T = 1, bench.T_FRAMES
    c, f0, uv, sid = W.make_inputs(cfg, B, T, seed=1234)
    noise = W.make_noise(cfg, B, T, seed=99)
    with torch.no_grad():
        ref, _ = O.synth_infer(sd, cfg, c, f0, uv, sid, noise, noice_scale=0.4)
    nd = {k: v.to(dev) for k, v in noise.items()}
    run = lambda: net.infer(c.to(dev), f0.to(dev), uv.to(dev), g=sid.to(dev), noice_scale=0.4, noise=nd)[0]
    o32 = run()
    net.split_f16()
    o = run()
    mse32, mse = (o32.cpu() - ref).pow(2).mean().item(), (o.cpu() - ref).pow(2).mean().item()
    mx32, mx = (o32.cpu() - ref).abs().max().item(), (o.cpu() - ref).abs().max().item()
    print(f"T=862 vs fp32 oracle: fp32 kernels MSE {mse32:.3e} max {mx32:.3e}; split pipeline MSE {mse:.3e} max {mx:.3e}; "
          f"split vs fp32 kernels max {(o - o32).abs().max().item():.3e}")
    assert mse < 1e-4 and mse <= 10 * mse32 + 1e-12, (mse, mse32)
    assert mx <= 4 * mx32 + 2e-6, (mx, mx32)
    net.enable_graph(True)
    o2, o3 = run(), run()
    assert torch.equal(o2, o) and torch.equal(o3, o)


@pytest.mark.parametrize("B,C,T", [(1, 16, 1000), (2, 32, 257), (1, 128, 256), (1, 8, 7), (1, 64, 3001)])
def test_snake_alias_split_vs_fp32_kernel(dev, B, C, T):
    """svc_snake_alias_hl (split planes in / out, fp32 arithmetic) against svc_snake_alias_f32 on the same input: what differs is the
    22-bit carriage of input and output (and libm-level differences of sin / exp)."""
    import svc_hip as S
    g = torch.Generator().manual_seed(C + T)
    x = torch.randn(B, C, T, generator=g) * 2.0
    alpha, beta = 0.4 * torch.randn(C, generator=g), 0.4 * torch.randn(C, generator=g)
    taps = W.snake_filter().tolist()
    ref = S.snake_alias(x.to(dev), alpha.to(dev), beta.to(dev), taps).cpu()
    y = S.from_h(S.snake_alias_h(S.to_h(x.to(dev), split=True), alpha.to(dev), beta.to(dev), taps)).cpu()
    assert y.shape == ref.shape
    assert (y - ref).abs().max().item() <= 3e-6 * max(1.0, ref.abs().max().item())


def test_snake_generator_split_inference(dev):
    """BASELINE configs[3]'s generator (nsf-snake-hifigan, full template widths) on the split pipeline against the fp32 CPU oracle and
    the engine's fp32 kernels; hipGraph replay bit-equal to the eager launches."""
    cfg = W.full_config()
    cfg["vocoder_name"] = "nsf-snake-hifigan"
    net, sd = _build(cfg, 77, dev)
    B, T = 2, 60
    c, f0, uv, sid = W.make_inputs(cfg, B, T, seed=21)
    noise = W.make_noise(cfg, B, T, seed=22)
    with torch.no_grad():
        ref, _ = O.synth_infer(sd, cfg, c, f0, uv, sid, noise, noice_scale=0.4)
    nd = {k: v.to(dev) for k, v in noise.items()}
    run = lambda: net.infer(c.to(dev), f0.to(dev), uv.to(dev), g=sid.to(dev), noice_scale=0.4, noise=nd)[0]
    o32 = run()
    net.split_f16()
    assert net.dec.half_mode == "split"
    o = run()
    mx32, mx = (o32.cpu() - ref).abs().max().item(), (o.cpu() - ref).abs().max().item()
    print(f"snake generator, B={B} T={T} vs fp32 oracle: fp32 kernels max {mx32:.3e}, split pipeline max {mx:.3e}, "
          f"split vs fp32 kernels {(o - o32).abs().max().item():.3e}; max|ref| {ref.abs().max().item():.3f}")
    assert o.shape == ref.shape and mx <= 4 * mx32 + 5e-6, (mx, mx32)
    net.enable_graph(True)
    assert torch.equal(run(), o)


# ---- range of the representation (include/svc_hip.h, RANGE) ---------------------------------------------------------
@pytest.mark.parametrize("wmag", [1e-6, 1e-3, 1.0, 1e2, 1e5])
def test_weights_of_any_magnitude_keep_22_bits(dev, wmag):
    """Weight tensors from 1e-6 to 1e5 (a nearly dead layer ... a weight-norm gain of trained magnitude): the pack's per-tensor power of
    two puts max |w| at 2^14 whatever the magnitude, so the error against float64 — in units of the output scale — is the same at every
    magnitude.  The INPUT is scaled the other way (clamped to the planes' range) so that the output stays O(1): what is measured is
    the weights' carriage, not the output's.  (Round 5 packed unscaled: |w| > 65504 became inf, |w| ~ 1e-6 kept four bits.)"""
    import svc_hip as S
    g = torch.Generator().manual_seed(11)
    B, Cin, Cout, T, KS = 1, 64, 64, 700, 7
    xmag = min(max(1.0 / wmag, 1e-2), 300.0)
    x = torch.randn(B, Cin, T, generator=g) * xmag
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5 * wmag
    b = torch.randn(Cout, generator=g) * wmag * xmag
    exact = F.conv1d(x.double(), w.double(), b.double(), padding=3)
    wp = S.pack_conv1d_h(w.to(dev), split=True)
    assert 2.0 ** 13 <= w.abs().max().item() / wp.acc_scale <= 2.0 ** 14 and torch.isfinite(wp.float()).all()
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    S.hl_range_flag(flag)
    try:
        y = S.from_h(S.conv1d_h(S.to_h(x.to(dev), split=True), wp, Cout, bias=b.to(dev), pad_left=3)).cpu()
    finally:
        S.hl_range_flag(None)
    y32 = S.conv1d(x.to(dev), S.pack_conv1d_weight(w.to(dev)), Cout, KS, bias=b.to(dev), pad_left=3).cpu()
    e, e32 = _err(y, exact), _err(y32, exact)
    print(f"weights x {wmag:g}, input x {xmag:g}: split {e:.2e}, fp32 kernel {e32:.2e} (of max |exact| = {exact.abs().max().item():.3g}), "
          f"range flag {int(flag.item())}")
    if exact.abs().max().item() <= 2000.0:
        assert int(flag.item()) == 0
        assert e < SCALE_BOUND and e < max(4 * e32 + 2e-7, CHAIN_FLOOR), (wmag, e, e32)
    else:                # wmag 1e5 with the input clamped at 1e-2: outputs of ~6e3 are beyond +-2047 — the flag's business, and it says so
        assert int(flag.item()) == 1


def test_rows_of_mixed_gain_inside_one_weight_tensor(dev):
    """weight_norm gains differ per output row: rows at 1e2, 1, 1e-2 and 1e-4 in ONE tensor.  A weight is carried to
    max(2^-22 |w|, 2^-39 max |w|); rows within 2^-17 of the largest keep their 22 bits, the 1e-6-of-the-largest rows keep ~19.  Measured per
    row in units of that row's own output scale — which for the small rows is itself small, so the activation planes' absolute floor
    (2^-30) shows in the last group's bound: stated, not hidden."""
    import svc_hip as S
    g = torch.Generator().manual_seed(12)
    Cin, Cout, T, KS = 32, 64, 900, 3
    gain = torch.tensor([1e2, 1.0, 1e-2, 1e-4]).repeat_interleave(16)
    x = torch.randn(1, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5 * gain.view(-1, 1, 1)
    exact = F.conv1d(x.double(), w.double(), padding=1)
    y = S.from_h(S.conv1d_h(S.to_h(x.to(dev), split=True), S.pack_conv1d_h(w.to(dev), split=True), Cout, pad_left=1)).cpu()
    per_row = (y.double() - exact).abs().amax(dim=(0, 2)) / exact.abs().amax(dim=(0, 2))
    groups = per_row.view(4, 16).amax(dim=1).tolist()
    print("per-row error (of the row's own scale) at gains 1e2 / 1 / 1e-2 / 1e-4:", " ".join(f"{v:.2e}" for v in groups))
    assert groups[0] < SCALE_BOUND and groups[1] < SCALE_BOUND and groups[2] < SCALE_BOUND
    assert groups[3] < 2e-5


def test_activation_overflow_raises_the_range_flag(dev):
    """|v| > 2047 cannot be carried by the activation planes (32 v as two fp16 pieces).  Every value the split kernels PRODUCE is checked
    as it is encoded: the launches report into the int32 word registered with svc_hl_range_flag.  In range: flag stays 0 and the
    result is fp32-level; out of range: flag set (by the conversion, by a convolution's epilogue, by the fused pair's intermediate)."""
    import svc_hip as S
    g = torch.Generator().manual_seed(13)
    C_, T = 32, 600
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    w = (torch.randn(C_, C_, 3, generator=g) / (C_ * 3) ** 0.5).to(dev)
    wp = S.pack_conv1d_h(w, split=True)
    zb = torch.zeros(C_, device=dev)
    S.hl_range_flag(flag)
    try:
        x_ok = (torch.randn(1, C_, T, generator=g) * 300.0).to(dev)                     # large but inside: max ~ 1.3e3
        assert x_ok.abs().max().item() < 2000.0
        xh = S.to_h(x_ok, split=True)
        y = S.from_h(S.conv1d_h(xh, wp, C_, pad_left=1)).cpu()
        assert int(flag.item()) == 0
        assert _err(y, F.conv1d(x_ok.cpu().double(), w.cpu().double(), padding=1)) < SCALE_BOUND
        x_bad = x_ok.clone()
        x_bad[0, 3, 17] = 3.0e3
        S.to_h(x_bad, split=True)                                                       # the conversion itself reports
        assert int(flag.item()) == 1
        flag.zero_()
        big = S.pack_conv1d_h(w * 8.0, split=True)                                      # inputs in range, OUTPUT ~ 1e4 out of it
        S.conv1d_h(xh, big, C_, pad_left=1)
        assert int(flag.item()) == 1
        flag.zero_()
        S.resblock_pair_h(xh, big, zb, wp, zb, 1)                                       # the fused pair's intermediate
        assert int(flag.item()) == 1
        flag.zero_()
        S.to_h(torch.full((1, 8, 5), float("nan"), device=dev), split=True)             # nan counts as out of range
        assert int(flag.item()) == 1
    finally:
        S.hl_range_flag(None)
    flag.zero_()
    S.to_h(x_bad, split=True)                                                           # withdrawn: no reporting, no crash
    assert int(flag.item()) == 0


@pytest.mark.parametrize("xmag", [1e-2, 1e-4, 1e-6])
def test_small_activations_and_the_absolute_floor(dev, xmag):
    """The activation planes hold 32 v: 22 bits down to |v| = 0.004 and an ABSOLUTE 2^-30 (9.3e-10) below (round 5, unscaled: 0.125 and
    3e-8).  On a 1e-2-scale tensor the result is still fp32-level; at 1e-4 / 1e-6 (digital silence through the generator) the error is
    bounded by the floor: 2^-30 per input element through sum |w|, plus 2^-30 on the output — inaudible, and stated as what it is."""
    import svc_hip as S
    g = torch.Generator().manual_seed(14)
    C_, T, KS = 32, 800, 7
    x = torch.randn(1, C_, T, generator=g) * xmag
    w = torch.randn(C_, C_, KS, generator=g) / (C_ * KS) ** 0.5
    exact = F.conv1d(x.double(), w.double(), padding=3)
    y = S.from_h(S.conv1d_h(S.to_h(x.to(dev), split=True), S.pack_conv1d_h(w.to(dev), split=True), C_, pad_left=3)).cpu()
    abs_err = (y.double() - exact).abs().max().item()
    floor = 2.0 ** -30 * (w.abs().sum(dim=(1, 2)).max().item() + 1.0)
    rel = abs_err / exact.abs().max().item()
    print(f"{xmag:g}-scale input: max abs error {abs_err:.2e} = {rel:.2e} of the output scale {exact.abs().max().item():.2e} "
          f"(floor bound {floor:.2e})")
    assert abs_err <= floor + SCALE_BOUND * exact.abs().max().item()
    if xmag >= 1e-2:
        assert rel < SCALE_BOUND
