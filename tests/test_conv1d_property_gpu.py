"""Property tier of SURVEY.md §4 for the convolution dispatch: `hypothesis` draws (B, T, channel counts, taps, dilation,
prologue / epilogue form) — including the odd decoder widths of config_tiny_template.json (200/100/50/25/12 and the
shrunken proxy's 6) and sequence lengths from one sample to one MRF row of a short clip — and every draw is checked
against torch's CPU fp32 convolution (what nn.Conv1d of the reference lowers to).  The draws are derandomised (same
examples on every run) so a failure is reproducible from the test id alone; `print_blob` gives the shrunk example.

Covered dispatch branches: register-fed direct kernel (B*T <= ~1k columns), tiled LDS kernels with scalar / float4 /
LDS-DMA staging (aligned and unaligned T), the strip kernel (Cout % 32 == 0, long rows), the 16x16x4 narrow-channel tiles,
split-K short rows, and the fused ResBlock pair (C == 16) against the two-launch form and against torch."""
import pytest
import torch
import torch.nn.functional as F
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

pytestmark = pytest.mark.gpu

ODD = [6, 12, 25, 50, 100, 200]
CH = ODD + [16, 32, 64, 128]
COMMON = dict(deadline=None, derandomize=True, print_blob=True,
              suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow, HealthCheck.data_too_large])


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def _t_strategy():
    # short rows, every residue mod 4 / 32 / 224, and rows long enough for the strip / LDS-DMA tiles
    return st.one_of(st.integers(1, 40), st.integers(41, 3000), st.sampled_from([224, 447, 448, 449, 862, 1724, 2048, 2999, 3000]))


@settings(max_examples=70, **COMMON)
@given(B=st.integers(1, 3), T=_t_strategy(), Cin=st.sampled_from(CH), Cout=st.sampled_from(CH),
       KS=st.sampled_from([1, 3, 5, 7, 11]), dil=st.sampled_from([1, 2, 3, 5]), pre=st.booleans(), res=st.booleans(),
       seed=st.integers(0, 2 ** 16))
def test_conv1d_dispatch_vs_torch(dev, B, T, Cin, Cout, KS, dil, pre, res, seed):
    import svc_hip as S
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5
    b = torch.randn(Cout, generator=g)
    pad = (KS * dil - dil) // 2
    ref = F.conv1d(F.leaky_relu(x, 0.1) if pre else x, w, b, dilation=dil, padding=pad)
    r = None
    if res and Cin == Cout:
        r = x
        ref = ref + x
    wp = S.pack_conv1d_weight(w.to(dev))
    xd = x.to(dev)
    y = S.conv1d(xd, wp, Cout, KS, bias=b.to(dev), dil=dil, pad_left=pad, pre_slope=0.1 if pre else 1.0,
                 res=xd if r is not None else None, res_mode=1 if r is not None else 0)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert _rel(y.cpu(), ref) < 3e-6, (B, T, Cin, Cout, KS, dil, pre, res)


@settings(max_examples=40, **COMMON)
@given(B=st.integers(1, 2), T=_t_strategy(), C=st.sampled_from([6, 12, 16, 25, 32, 50, 64, 100]), KS=st.sampled_from([3, 7, 11]),
       seed=st.integers(0, 2 ** 16))
def test_resblock1_vs_torch(dev, B, T, C, KS, seed):
    """vdecoder/hifigan/models.py:60-67 through the mirror module's own dispatch (fused pair at C = 16, strip / tiled / narrow
    kernels elsewhere), accumulating into the MRF mean like Generator.forward (:382-389)."""
    from vdecoder.hifigan.models import ResBlock1
    torch.manual_seed(seed)
    blk = ResBlock1(None, C, KS, (1, 3, 5))
    g = torch.Generator().manual_seed(seed)
    for p in blk.parameters():
        with torch.no_grad():
            if p.dim() == 3 and p.shape[1] == C:                     # weight_v: fan-in scaled so activations stay O(1)
                p.copy_(torch.randn(p.shape, generator=g) / (C * KS) ** 0.5)
            elif p.dim() == 3:                                       # weight_g
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    x = torch.randn(B, C, T, generator=g)
    acc = torch.randn(B, C, T, generator=g)
    ref = x
    for c1, c2 in zip(blk.convs1, blk.convs2):
        w1 = c1.weight_v * (c1.weight_g / c1.weight_v.flatten(1).norm(dim=1).view(-1, 1, 1))
        w2 = c2.weight_v * (c2.weight_g / c2.weight_v.flatten(1).norm(dim=1).view(-1, 1, 1))
        xt = F.conv1d(F.leaky_relu(ref, 0.1), w1, c1.bias, dilation=c1.dilation, padding=(KS - 1) * c1.dilation // 2)
        xt = F.conv1d(F.leaky_relu(xt, 0.1), w2, c2.bias, padding=(KS - 1) // 2)
        ref = xt + ref
    ref = ((acc + ref) / 3).detach()
    blk = blk.to(dev).eval()
    out = acc.to(dev).clone()
    with torch.no_grad():
        y = blk(x.to(dev), out=out, beta=1.0, out_div=3.0)
    torch.cuda.synchronize()
    assert y.data_ptr() == out.data_ptr()
    assert _rel(out.cpu(), ref) < 1e-5, (B, T, C, KS)


@settings(max_examples=25, **COMMON)
@given(B=st.integers(1, 2), L=st.integers(1, 400), Cin=st.sampled_from([12, 25, 50, 100, 200, 400, 32, 64]),
       su=st.sampled_from([(16, 8), (4, 2), (8, 4)]), seed=st.integers(0, 2 ** 16))
def test_conv_transpose1d_vs_torch(dev, B, L, Cin, su, seed):
    """ups[i] (vdecoder/hifigan/models.py:340-342,378) at odd widths: ConvTranspose1d(C -> C // 2, k, u, padding=(k - u + 1) // 2)."""
    import svc_hip as S
    K, u = su
    Cout = Cin // 2
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, L, generator=g)
    v = torch.randn(Cin, Cout, K, generator=g) / (Cin * K / u) ** 0.5
    gw = torch.rand(Cin, 1, 1, generator=g) + 0.5
    b = torch.randn(Cout, generator=g)
    w = v * (gw / v.flatten(1).norm(dim=1).view(-1, 1, 1))
    pad = (K - u + 1) // 2
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=u, padding=pad)
    wp = S.pack_convt1d_weight(v.to(dev), gw.to(dev), u)
    y = S.conv_transpose1d(x.to(dev), wp, Cout, K, u, pad, bias=b.to(dev), pre_slope=0.1)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert _rel(y.cpu(), ref) < 3e-6, (B, L, Cin, K, u)
