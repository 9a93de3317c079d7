"""GPU parity of the fused fp32-MFMA conv1d (svc_conv1d_f32) against torch CPU fp32 conv1d — the op every
nn.Conv1d of the reference path lowers to on its CPU path (mkldnn_convolution)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


CASES = [
    # B, Cin, Cout, T, KS, dil
    (1, 64, 64, 300, 3, 1),
    (2, 128, 128, 515, 7, 3),
    (1, 256, 256, 200, 11, 5),
    (1, 32, 32, 1000, 3, 5),
    (1, 16, 16, 2100, 11, 3),
    (1, 16, 16, 700, 7, 1),
    (2, 192, 384, 77, 5, 1),
    (1, 768, 192, 50, 5, 1),
    (1, 192, 576, 33, 1, 1),
    (1, 100, 50, 260, 3, 1),
    (3, 12, 25, 97, 7, 2),
    (1, 128, 128, 20000, 3, 1),
    (1, 64, 64, 20000, 7, 5),
    (1, 192, 512, 20000, 7, 1),
    # batched training shapes: the 64 x 192 tiling (three column tiles per wave) is selected for these
    (16, 192, 192, 768, 1, 1),
    (16, 384, 192, 768, 5, 1),
    (4, 192, 384, 700, 5, 2),
    (16, 96, 192, 767, 3, 1),      # T not a multiple of 4: the scalar-staging instantiation of the same tiling
]


@pytest.mark.parametrize("B,Cin,Cout,T,KS,dil", CASES)
def test_conv1d_plain(dev, B, Cin, Cout, T, KS, dil):
    import svc_hip as S
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + T + KS)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5
    b = torch.randn(Cout, generator=g)
    pad = (KS * dil - dil) // 2
    ref = F.conv1d(x, w, b, dilation=dil, padding=pad)
    wp = S.pack_conv1d_weight(w.to(dev))
    y = S.conv1d(x.to(dev), wp, Cout, KS, bias=b.to(dev), dil=dil, pad_left=pad)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert _rel(y.cpu(), ref) < 2e-6


def test_conv1d_weight_norm_lrelu_residual(dev):
    """ResBlock1 inner step: xt = c2(lrelu(c1(lrelu(x)))) + x with weight-normed convs
    (vdecoder/hifigan/models.py:60-67)."""
    import svc_hip as S
    g = torch.Generator().manual_seed(7)
    B, C, T, KS, dil = 2, 64, 777, 7, 3
    x = torch.randn(B, C, T, generator=g)
    v = torch.randn(C, C, KS, generator=g) * 0.05
    gw = torch.rand(C, 1, 1, generator=g) + 0.5
    b = torch.randn(C, generator=g) * 0.1
    w = v * (gw / v.flatten(1).norm(dim=1).view(-1, 1, 1))
    pad = (KS * dil - dil) // 2
    ref = F.conv1d(F.leaky_relu(x, 0.1), w, b, dilation=dil, padding=pad) + x
    wp = S.pack_conv1d_weight(v.to(dev), gw.to(dev))
    xd = x.to(dev)
    y = S.conv1d(xd, wp, C, KS, bias=b.to(dev), dil=dil, pad_left=pad, pre_slope=0.1, res=xd, res_mode=1)
    assert _rel(y.cpu(), ref) < 2e-6
    # accumulate + divide epilogue: xs = (xs_old + y) / 3
    xs = torch.randn(B, C, T, generator=g)
    out = xs.to(dev).clone()
    S.conv1d(xd, wp, C, KS, bias=b.to(dev), dil=dil, pad_left=pad, pre_slope=0.1, res=xd, res_mode=1, out=out,
             beta=1.0, out_div=3.0)
    assert _rel(out.cpu(), (xs + ref) / 3) < 2e-6


def test_conv1d_gate_and_res_skip(dev):
    """One WN layer (modules/modules.py:118-136): in_layer conv + cond -> tanh*sigmoid gate -> res/skip 1x1."""
    import svc_hip as S
    g = torch.Generator().manual_seed(11)
    B, H, T, KS = 2, 192, 333, 5
    x = torch.randn(B, H, T, generator=g)
    w_in = torch.randn(2 * H, H, KS, generator=g) / (H * KS) ** 0.5
    b_in = torch.randn(2 * H, generator=g) * 0.1
    cond = torch.randn(B, 2 * H, 1, generator=g)
    w_rs = torch.randn(2 * H, H, 1, generator=g) / H ** 0.5
    b_rs = torch.randn(2 * H, generator=g) * 0.1
    mask = (torch.arange(T)[None, :] < torch.tensor([T, T - 40])[:, None]).float().unsqueeze(1)
    out_prev = torch.randn(B, H, T, generator=g)

    x_in = F.conv1d(x, w_in, b_in, padding=2) + cond
    acts = torch.tanh(x_in[:, :H]) * torch.sigmoid(x_in[:, H:])
    rs = F.conv1d(acts, w_rs, b_rs)
    x_new = (x + rs[:, :H]) * mask
    out_new = out_prev + rs[:, H:]

    d = dev
    wp_in = S.pack_conv1d_weight(w_in.to(d), gate_half=H)
    acts_d = S.conv1d(x.to(d), wp_in, 2 * H, KS, bias=b_in.to(d), pad_left=2, cond=cond.to(d), epi=S.EPI_GATE)
    assert _rel(acts_d.cpu(), acts) < 5e-6
    wp_rs = S.pack_conv1d_weight(w_rs.to(d))
    xd = x.to(d).clone()
    od = out_prev.to(d).clone()
    S.conv1d(acts_d, wp_rs, 2 * H, 1, bias=b_rs.to(d), mask=mask.to(d), res=xd, out=xd, out2=od, beta=1.0,
             epi=S.EPI_RES_SKIP, skip_from=H)
    assert _rel(xd.cpu(), x_new) < 5e-6
    assert _rel(od.cpu(), out_new) < 5e-6


def test_conv1d_flipped_views(dev):
    """Channel Flip (modules/modules.py:232-239) folded into strides: conv over flip(x) writing flip(y)."""
    import svc_hip as S
    g = torch.Generator().manual_seed(3)
    B, C, T = 2, 96, 211
    x = torch.randn(B, 2 * C, T, generator=g)
    w = torch.randn(192, C, 1, generator=g) / C ** 0.5
    b = torch.randn(192, generator=g)
    xf = torch.flip(x, [1])
    ref = F.conv1d(xf[:, :C], w, b)
    xd = x.to(dev)
    wp = S.pack_conv1d_weight(w.to(dev))
    y = S.conv1d(S.flip_view(xd).narrow_c(0, C), wp, 192, 1, bias=b.to(dev))
    assert _rel(y.cpu(), ref) < 2e-6


@pytest.mark.parametrize("C,K,d,B,T", [(16, 3, 1, 2, 1000), (16, 7, 3, 1, 517), (16, 11, 5, 2, 2049), (16, 3, 5, 1, 777),
                                       (16, 7, 1, 2, 1500), (16, 11, 3, 2, 131), (16, 11, 1, 1, 7)])
def test_resblock_pair_equals_two_conv_launches(dev, C, K, d, B, T):
    """svc_resblock_pair_f32 (narrow MRF stages: conv1 -> lrelu -> conv2 -> + x in one kernel, intermediate and residual in
    LDS) against the two svc_conv1d_f32 launches it replaces — same reduction order, so equal to fp32 round-off — and against
    a plain torch fp32 restatement of vdecoder/hifigan/models.py:62-66; incl. ragged lengths, sequences shorter than a
    tile / the halo, and the MRF-sum epilogue (beta, out_div)."""
    import svc_hip as S
    g = torch.Generator().manual_seed(C * 100 + K * 10 + d)
    x = torch.randn(B, C, T, generator=g).to(dev)
    w1 = (torch.randn(C, C, K, generator=g) / (C * K) ** 0.5).to(dev)
    w2 = (torch.randn(C, C, K, generator=g) / (C * K) ** 0.5).to(dev)
    b1, b2 = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    prev = torch.randn(B, C, T, generator=g).to(dev)
    w1p, w2p = S.pack_conv1d_weight(w1), S.pack_conv1d_weight(w2)
    F = torch.nn.functional
    ref = F.conv1d(F.leaky_relu(F.conv1d(F.leaky_relu(x, 0.1), w1, b1, padding=d * (K - 1) // 2, dilation=d), 0.1), w2, b2,
                   padding=(K - 1) // 2) + x
    for beta, div in ((0.0, 1.0), (1.0, 3.0)):
        xt = S.conv1d(x, w1p, C, K, bias=b1, dil=d, pad_left=d * (K - 1) // 2, pre_slope=0.1, post_act=S.ACT_LRELU, post_slope=0.1)
        two = prev.clone()
        S.conv1d(xt, w2p, C, K, bias=b2, pad_left=(K - 1) // 2, res=x, res_mode=1, out=two, beta=beta, out_div=div)
        one = prev.clone()
        S.resblock_pair(x, w1p, b1, w2p, b2, K, d, slope=0.1, out=one, beta=beta, out_div=div)
        want = (ref + beta * prev) / div
        scale = max(1.0, want.abs().max().item())
        assert (one - two).abs().max().item() <= 2e-6 * scale, (beta, (one - two).abs().max().item())
        assert (one - want).abs().max().item() <= 2e-5 * scale
    with pytest.raises(S.SvcError):
        S.resblock_pair(x, w1p, b1, w2p, b2, K, d, out=x)              # in-place is refused


# ---- conv1d_strip.hip: the one-workgroup-per-CU strip kernel for long dense convs (MRF ResBlock convs) -------------------
# Forced arrangement (svc_debug_set_conv_strip(2 + i)): 0 = 32x32 MFMA 4x1 strips (128 x 224 tile), 1 = 2x2 (64 x 448),
# 2 = 1x4 (32 x 896), 3 = 16x16 MFMA 4x1 (64 x 112), 4 = split-K (one 32 x 224 strip, the four waves split the input
# channels); + 10 = one wave per strip instead of two (4 + 3 tiles).  Shapes cover ragged tails (T not a multiple of the strip), sequences
# shorter than one strip, several row tiles, a wave whose rows lie past Cout, batches, and both tensor-edge paddings.
STRIP_CASES = [
    # arr, B, Cin, Cout, T, KS, dil
    (0, 2, 128, 128, 1000, 11, 5),
    (0, 1, 64, 96, 460, 3, 1),
    (0, 1, 128, 256, 300, 7, 3),
    (1, 1, 64, 64, 1500, 7, 5),
    (1, 2, 64, 64, 100, 3, 1),
    (1, 1, 32, 128, 904, 11, 1),
    (2, 1, 32, 32, 2000, 11, 3),
    (2, 2, 32, 32, 8, 3, 5),
    (3, 1, 256, 256, 300, 11, 5),
    (3, 2, 32, 64, 252, 3, 1),
    (3, 1, 128, 128, 700, 7, 3),
    (4, 1, 256, 256, 300, 11, 5),
    (4, 2, 64, 96, 460, 3, 1),
    (4, 1, 128, 128, 700, 7, 3),
]


@pytest.fixture
def strip_mode():
    import svc_hip as S
    yield lambda m: S.lib().svc_debug_set_conv_strip(m)
    S.lib().svc_debug_set_conv_strip(1)


@pytest.mark.parametrize("wps", [2, 1])
@pytest.mark.parametrize("arr,B,Cin,Cout,T,KS,dil", STRIP_CASES)
def test_conv1d_strip_kernel(dev, strip_mode, arr, B, Cin, Cout, T, KS, dil, wps):
    """Every epilogue form the MRF stage uses (vdecoder/hifigan/models.py:60-67,382-388), strip kernel vs the kernels it
    replaces (fp32 round-off: these short test sequences take split-K / register-fed tilings there, and the 16x16x4 instruction
    groups its products differently from 32x32x2) and vs torch CPU fp32."""
    import svc_hip as S
    g = torch.Generator().manual_seed(arr * 7919 + Cin + Cout + T + KS + dil)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, T, generator=g)
    prev = torch.randn(B, Cout, T, generator=g)
    pad = (KS * dil - dil) // 2
    xd, wp, bd, resd = x.to(dev), S.pack_conv1d_weight(w.to(dev)), b.to(dev), res.to(dev)
    conv = lambda xx: F.conv1d(xx, w, b, dilation=dil, padding=pad)
    forms = [
        ("first conv of a pair", dict(pre_slope=0.1, post_act=S.ACT_LRELU, post_slope=0.1), 0.0, 1.0,
         lambda: F.leaky_relu(conv(F.leaky_relu(x, 0.1)), 0.1)),
        ("second conv", dict(res=resd, res_mode=1), 0.0, 1.0, lambda: conv(x) + res),
        ("second conv, chain end", dict(res=resd, res_mode=1), 1.0, 3.0, lambda: (conv(x) + res + prev) / 3.0),
        ("ResBlock2 conv", dict(pre_slope=0.1, res=resd, res_mode=1), 1.0, 1.0, lambda: conv(F.leaky_relu(x, 0.1)) + res + prev),
    ]
    n0 = S.lib().svc_debug_set_conv_strip(-1)
    for name, kw, beta, div, ref_fn in forms:
        ref = ref_fn()
        outs = []
        for mode in (2 + arr + (10 if wps == 1 else 0), 0):
            strip_mode(mode)
            out = prev.to(dev).clone()
            S.conv1d(xd, wp, Cout, KS, bias=bd, dil=dil, pad_left=pad, out=out, beta=beta, out_div=div, **kw)
            torch.cuda.synchronize()
            outs.append(out.cpu())
        assert _rel(outs[0], outs[1]) < 3e-6, (name, (outs[0] - outs[1]).abs().max().item())
        assert _rel(outs[0], ref) < 3e-6, name
    assert S.lib().svc_debug_set_conv_strip(-1) - n0 == len(forms), "the strip kernel was not the one that ran"


def test_conv1d_strip_auto_selection_mrf_shapes(dev, strip_mode):
    """Automatic routing: the 128 / 64 / 32-channel MRF stage shapes of a 10 s clip (T = 862 frames) go to the strip kernel
    (the 256-channel stage stays on the tiled kernel, which measures faster there than the 16x16 and split-K strips; 16
    channels run the fused pair); short / batched training shapes do not."""
    import svc_hip as S
    strip_mode(1)
    L = 862
    for u, C in zip([8, 8, 2, 2], [256, 128, 64, 32]):
        L *= u
        x = torch.randn(1, C, L, device=dev)
        wp = S.pack_conv1d_weight(torch.randn(C, C, 3, device=dev) * 0.05)
        n0 = S.lib().svc_debug_set_conv_strip(-1)
        y = S.conv1d(x, wp, C, 3, pad_left=1, res=x, res_mode=1)
        assert S.lib().svc_debug_set_conv_strip(-1) == n0 + (1 if C <= 128 else 0), (C, L)
        strip_mode(0)
        y0 = S.conv1d(x, wp, C, 3, pad_left=1, res=x, res_mode=1)
        strip_mode(1)
        if C <= 128:   # same instruction, same order of the reduction, same epilogue expression as the 32x32-tile kernels: bit-equal
            assert torch.equal(y, y0), (C, L)
        else:          # split-K sums four partial reductions
            assert _rel(y, y0) < 3e-6, (C, L)
    x = torch.randn(16, 128, 1024, device=dev)
    wp = S.pack_conv1d_weight(torch.randn(128, 128, 3, device=dev) * 0.05)
    n0 = S.lib().svc_debug_set_conv_strip(-1)
    S.conv1d(x, wp, 128, 3, pad_left=1, res=x, res_mode=1)
    assert S.lib().svc_debug_set_conv_strip(-1) == n0


@pytest.mark.parametrize("Cin,Cout,T,KS,dil", [(128, 128, 20004, 11, 5), (64, 64, 33000, 3, 1), (256, 256, 6896, 7, 3),
                                               (32, 32, 40000, 7, 1), (128, 64, 17000, 5, 1)])
def test_conv1d_direct_epilogue_equals_lds_epilogue(dev, strip_mode, Cin, Cout, T, KS, dil):
    """The tiled LDS-DMA kernels' register -> global epilogue (conv_epilogue_direct) against their LDS-transposed one
    (svc_debug_set_conv_cfg(1000000000) switches it off): same expression, same order -> bit-equal, for every epilogue form
    of the MRF stage, with a ragged last tile; and against torch CPU."""
    import svc_hip as S
    g = torch.Generator().manual_seed(Cin + Cout + T + KS)
    x = torch.randn(1, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5
    b = torch.randn(Cout, generator=g)
    res = torch.randn(1, Cout, T, generator=g)
    prev = torch.randn(1, Cout, T, generator=g)
    pad = (KS * dil - dil) // 2
    xd, wp, bd, resd = x.to(dev), S.pack_conv1d_weight(w.to(dev)), b.to(dev), res.to(dev)
    conv = lambda xx: F.conv1d(xx, w, b, dilation=dil, padding=pad)
    forms = [("conv1", dict(pre_slope=0.1, post_act=S.ACT_LRELU, post_slope=0.1), 0.0, 1.0, lambda: F.leaky_relu(conv(F.leaky_relu(x, 0.1)), 0.1)),
             ("conv2", dict(res=resd, res_mode=1), 0.0, 1.0, lambda: conv(x) + res),
             ("conv2-end", dict(res=resd, res_mode=1), 1.0, 3.0, lambda: (conv(x) + res + prev) / 3.0),
             ("plain", dict(), 0.0, 1.0, lambda: conv(x))]
    strip_mode(0)
    try:
        for name, kw, beta, div, ref_fn in forms:
            outs = []
            for code in (0, 1000000000):
                S.lib().svc_debug_set_conv_cfg(code)
                out = prev.to(dev).clone()
                S.conv1d(xd, wp, Cout, KS, bias=bd, dil=dil, pad_left=pad, out=out, beta=beta, out_div=div, **kw)
                torch.cuda.synchronize()
                outs.append(out.cpu())
            assert torch.equal(outs[0], outs[1]), (name, (outs[0] - outs[1]).abs().max().item())
            assert _rel(outs[0], ref_fn()) < 3e-6, name
    finally:
        S.lib().svc_debug_set_conv_cfg(0)


@pytest.mark.parametrize("KS", [3, 7, 11])
@pytest.mark.parametrize("B,T", [(1, 4099), (2, 700), (1, 100), (1, 441344 // 8)])
def test_resblock16_one_launch_is_bit_equal_to_its_three_pair_launches(dev, KS, B, T):
    """svc_resblock16_f32: the whole 16-channel ResBlock1 (dilations 1, 3, 5; vdecoder/hifigan/models.py:60-67) in ONE launch against the
    three svc_resblock_pair_f32 launches it replaces — same reduction order and epilogue expressions, so torch.equal; tile borders,
    sequences shorter than the halo (T = 100 against 60 halo columns per side at 11 taps), the MRF accumulate / divide epilogue; and
    against torch on the CPU."""
    import svc_hip as S
    g = torch.Generator().manual_seed(KS * 1000 + T)
    x = torch.randn(B, 16, T, generator=g)
    old = torch.randn(B, 16, T, generator=g)
    dils = (1, 3, 5)
    ws = [(torch.randn(16, 16, KS, generator=g) / (16 * KS) ** 0.5, torch.randn(16, generator=g) * 0.2,
           torch.randn(16, 16, KS, generator=g) / (16 * KS) ** 0.5, torch.randn(16, generator=g) * 0.2) for _ in dils]
    xd = x.to(dev)
    packs = [(S.pack_conv1d_weight(w1.to(dev)), b1.to(dev), S.pack_conv1d_weight(w2.to(dev)), b2.to(dev)) for w1, b1, w2, b2 in ws]
    # three pair launches, the last with the accumulate / divide epilogue
    cur = xd
    for j, (w1p, b1, w2p, b2) in enumerate(packs[:-1]):
        cur = S.resblock_pair(cur, w1p, b1, w2p, b2, KS, dils[j])
    ref = old.to(dev).clone()
    S.resblock_pair(cur, *packs[-1], KS, dils[-1], out=ref, beta=1.0, out_div=3.0)
    one = old.to(dev).clone()
    got = S.resblock16(xd, packs, KS, dils, out=one, beta=1.0, out_div=3.0)
    assert got is one
    torch.cuda.synchronize()
    assert torch.equal(one, ref)
    plain = S.resblock16(xd, packs, KS, dils)
    y = x.double()
    for (w1, b1, w2, b2), d in zip(ws, dils):
        t = F.conv1d(F.leaky_relu(y, 0.1), w1.double(), b1.double(), dilation=d, padding=d * (KS - 1) // 2)
        y = F.conv1d(F.leaky_relu(t, 0.1), w2.double(), b2.double(), padding=(KS - 1) // 2) + y
    err = (plain.cpu().double() - y).abs().max().item() / y.abs().max().item()
    assert err < 2e-6, err
    assert S.resblock16(xd, packs, KS, (1, 2, 4)) is None          # other dilation sets keep the pair launches
