"""The register-fed short-sequence conv kernel with 16-byte operand loads (conv1d_mfma_direct4_kernel: lane-linear second weight
pack, one KSC-float load per channel for the taps, partial sums on several accumulators) against torch CPU fp32 conv1d, and
against the 4-byte-load form of the same kernel (a pack without the `d4_ok` mark takes it) at fp32 rounding distance: the two sum
the same products in a different grouping.
Shapes are the ones the encoder / flow / pre convolutions and the phases-as-rows ConvTranspose1d stages of one utterance launch
(models.py:45-52,155-162, modules/attentions.py:337-345, vdecoder/hifigan/models.py:340-342)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def _both(S, x, w, dev, Cout, KS, **kw):
    """(with the lane-linear pack, without it) on the same operands."""
    wp = S.pack_conv1d_weight(w.to(dev), None, kw.pop("gate_half", 0))
    y4 = S.conv1d(x, wp, Cout, KS, **kw)
    used = getattr(wp, "d4", None) is not None
    wp0 = wp.clone()                      # (a clone carries no mark: no second pack, the 4-byte-load kernel)
    y1 = S.conv1d(x, wp0, Cout, KS, **kw)
    torch.cuda.synchronize()
    return y4, y1, used


CASES = [
    # B, Cin, Cout, T, KS, pad_left        (T = 862: one 10 s utterance; the others exercise the edge tiles)
    (1, 192, 768, 862, 3, 1),
    (1, 768, 192, 862, 3, 1),
    (1, 192, 576, 862, 1, 0),
    (1, 192, 192, 33, 1, 0),
    (1, 768, 192, 862, 5, 2),
    (1, 192, 512, 862, 7, 3),
    (2, 192, 384, 77, 5, 2),
    (1, 192, 192, 3, 3, 1),                # Tin == KS: every lane's window is clamped
    (1, 64, 64, 40, 3, 2),                 # causal padding (attentions.py:345-356): pad_left = KS - 1
    (1, 64, 96, 45, 7, 6),
    (1, 128, 64, 31, 5, 0),                # no left padding: only the right edge moves windows
    (3, 256, 128, 100, 2, 1),
]


@pytest.mark.parametrize("B,Cin,Cout,T,KS,pad", CASES)
def test_d4_matches_torch_and_the_4_byte_kernel(dev, B, Cin, Cout, T, KS, pad):
    import svc_hip as S
    g = torch.Generator().manual_seed(B + Cin * 3 + Cout * 5 + T * 7 + KS * 11 + pad)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv1d(F.pad(x, (pad, KS - 1 - pad)), w, b)
    y4, y1, used = _both(S, x.to(dev), w, dev, Cout, KS, bias=b.to(dev), pad_left=pad, Tout=T)
    assert used, "the launch was meant to take the lane-linear pack"
    assert y4.shape == ref.shape
    assert _rel(y4.cpu(), ref) < 2e-6
    assert _rel(y4, y1) < 2e-6


def test_d4_pre_activation_and_flipped_channels(dev):
    """leaky-ReLU prologue (max(x, slope x)) and a channel-flipped input view (the flow's Flip, modules/modules.py:253-262)."""
    import svc_hip as S
    g = torch.Generator().manual_seed(5)
    B, C, T, KS = 1, 192, 862, 5
    x = torch.randn(B, C, T, generator=g)
    w = torch.randn(C, C, KS, generator=g) / (C * KS) ** 0.5
    ref = F.conv1d(F.leaky_relu(torch.flip(x, [1]), 0.1), w, None, padding=2)
    xd = x.to(dev)
    y4, y1, used = _both(S, S.flip_view(xd), w, dev, C, KS, pad_left=2, pre_slope=0.1)
    assert used
    assert _rel(y4.cpu(), ref) < 2e-6
    assert _rel(y4, y1) < 2e-6


def test_d4_gate_and_res_skip_epilogues(dev):
    """One WN layer of the flow on 862 frames: in_layer with the tanh * sigmoid gate and the conditioning row, then the 1 x 1
    res / skip conv (modules/modules.py:110-138) — both through the 16-byte-load kernel."""
    import svc_hip as S
    g = torch.Generator().manual_seed(9)
    B, H, T, KS = 1, 192, 862, 5
    x = torch.randn(B, H, T, generator=g)
    w_in = torch.randn(2 * H, H, KS, generator=g) / (H * KS) ** 0.5
    b_in = torch.randn(2 * H, generator=g)
    cond = torch.randn(B, 2 * H, 1, generator=g)
    w_rs = torch.randn(2 * H, H, 1, generator=g) / H ** 0.5
    b_rs = torch.randn(2 * H, generator=g)
    mask = torch.ones(B, 1, T)
    xin = F.conv1d(x, w_in, b_in, padding=2) + cond
    acts = torch.tanh(xin[:, :H]) * torch.sigmoid(xin[:, H:])
    rs = F.conv1d(acts, w_rs, b_rs)
    ref_x = (x + rs[:, :H]) * mask
    ref_skip = rs[:, H:]
    xd, cd, md = x.to(dev), cond.to(dev), mask.to(dev)
    a4, a1, used = _both(S, xd, w_in, dev, 2 * H, KS, bias=b_in.to(dev), pad_left=2, cond=cd, epi=S.EPI_GATE, gate_half=H)
    assert used
    assert _rel(a4.cpu(), acts) < 5e-6
    assert _rel(a4, a1) < 5e-6
    outs = []
    for marked in (True, False):
        wp = S.pack_conv1d_weight(w_rs.to(dev))
        if not marked:
            wp = wp.clone()
        xo, sk = torch.empty_like(xd), torch.zeros_like(xd)
        S.conv1d(a4, wp, 2 * H, 1, bias=b_rs.to(dev), res=xd, mask=md, epi=S.EPI_RES_SKIP, out=xo, out2=sk, skip_from=H)
        outs.append((xo, sk, getattr(wp, "d4", None) is not None))
    torch.cuda.synchronize()
    assert outs[0][2] and not outs[1][2]
    assert _rel(outs[0][0].cpu(), ref_x) < 5e-6 and _rel(outs[0][1].cpu(), ref_skip) < 5e-6
    assert _rel(outs[0][0], outs[1][0]) < 5e-6 and _rel(outs[0][1], outs[1][1]) < 5e-6


@pytest.mark.parametrize("Cin,Cout,K,u,T", [(512, 256, 16, 8, 120), (256, 128, 16, 8, 300), (256, 64, 4, 2, 1000)])
def test_d4_conv_transpose_rows_layout(dev, Cin, Cout, K, u, T):
    """The decoder's upsampling stages (phases as rows of one 2-tap convolution) with the lane-linear pack."""
    import svc_hip as S
    g = torch.Generator().manual_seed(Cin + K + T)
    x = torch.randn(1, Cin, T, generator=g)
    w = torch.randn(Cin, Cout, K, generator=g) / (Cin * K / u) ** 0.5
    b = torch.randn(Cout, generator=g)
    pad = (K - u) // 2
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=u, padding=pad)
    wp = S.pack_convt1d_weight(w.to(dev), None, u)
    y4 = S.conv_transpose1d(x.to(dev), wp, Cout, K, u, pad, bias=b.to(dev), pre_slope=0.1)
    assert getattr(wp, "d4", None) is not None
    wp0 = wp.clone()
    y1 = S.conv_transpose1d(x.to(dev), wp0, Cout, K, u, pad, bias=b.to(dev), pre_slope=0.1)
    torch.cuda.synchronize()
    assert _rel(y4.cpu(), ref) < 2e-6
    assert _rel(y4, y1) < 2e-6


def test_d4_is_never_derived_from_an_operand_buffer_that_is_rewritten_in_place(dev):
    """Training plans rewrite their operand buffers every step (ConvWeightPlan.prepare): such a buffer carries no `d4_ok` mark, so a
    short launch from it must not cache a second pack that the next step's weights would leave stale."""
    import svc_hip as S
    pl = S.ConvWeightPlan(S.ConvWeightPlan.DENSE, 192, 192, 3)
    v = torch.randn(192, 192, 3, device=dev)
    wp, _ = pl.prepare(v)
    x = torch.randn(1, 192, 100, device=dev)
    S.conv1d(x, wp, 192, 3, pad_left=1)
    assert getattr(wp, "d4", None) is None


def test_second_pack_is_made_only_for_launches_that_read_it(dev):
    """The unit encoder's 768 -> 3072 projection on 500 frames takes an LDS-staged 64 x 128 tiling: svc_conv1d_wants_d4 says no and the
    weights are not duplicated; the same pack on a 40-frame input takes the register-fed kernel and gets its second pack then."""
    import svc_hip as S
    g = torch.Generator().manual_seed(11)
    w = torch.randn(3072, 768, 1, generator=g) / 768 ** 0.5
    wp = S.pack_conv1d_weight(w.to(dev))
    x = torch.randn(1, 768, 500, generator=g)
    y = S.conv1d(x.to(dev), wp, 3072, 1)
    assert getattr(wp, "d4", None) is None
    assert _rel(y.cpu(), F.conv1d(x, w)) < 2e-6
    xs = torch.randn(1, 768, 40, generator=g)
    ys = S.conv1d(xs.to(dev), wp, 3072, 1)
    assert getattr(wp, "d4", None) is not None
    assert _rel(ys.cpu(), F.conv1d(xs, w)) < 2e-6
    y2 = S.conv1d(x.to(dev), wp, 3072, 1)                 # the pack being there changes nothing for the tiled launch
    torch.cuda.synchronize()
    assert torch.equal(y2, y)


from hypothesis import HealthCheck, given, settings            # noqa: E402
from hypothesis import strategies as st                        # noqa: E402


@settings(max_examples=80, deadline=None, derandomize=True, print_blob=True,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow, HealthCheck.data_too_large])
@given(B=st.integers(1, 3), T=st.one_of(st.integers(1, 70), st.integers(71, 900)), Cin=st.sampled_from([32, 64, 96, 128, 192, 256, 768]),
       Cout=st.sampled_from([12, 32, 50, 64, 100, 192, 384]), KS=st.sampled_from([1, 2, 3, 5, 7]), padf=st.floats(0, 1),
       pre=st.booleans(), flip=st.booleans(), seed=st.integers(0, 2 ** 16))
def test_d4_property_any_padding_any_edge(dev, B, T, Cin, Cout, KS, padf, pre, flip, seed):
    """Draws over the shapes that can take the lane-linear pack (and neighbours that cannot: Cin = 96, rows shorter than the taps):
    any left padding 0 .. KS - 1 (the clamped tap window and its select chain at both edges), batch rows, the leaky-ReLU prologue, a
    channel-flipped view — each against torch's CPU fp32 convolution, whichever kernel the dispatch picks."""
    import svc_hip as S
    g = torch.Generator().manual_seed(seed)
    pad = min(KS - 1, int(padf * KS))
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5
    b = torch.randn(Cout, generator=g)
    xin = torch.flip(x, [1]) if flip else x
    xin = F.leaky_relu(xin, 0.1) if pre else xin
    ref = F.conv1d(F.pad(xin, (pad, KS - 1 - pad)), w, b)
    wp = S.pack_conv1d_weight(w.to(dev))
    xd = x.to(dev)
    y = S.conv1d(S.flip_view(xd) if flip else xd, wp, Cout, KS, bias=b.to(dev), pad_left=pad, Tout=T, pre_slope=0.1 if pre else 1.0)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert _rel(y.cpu(), ref) < 3e-6, (B, T, Cin, Cout, KS, pad, pre, flip, getattr(wp, "d4", None) is not None)
