"""CPU suite: the geometry of DiscriminatorP's padded-row layout (models.DiscriminatorP.forward, svc_autograd.conv1d with
out_blocks) — block counts per layer against torch's own Conv2d((k,1),(s,1)) output sizes on the reference's [B,1,T/p,p] view
(models.py:180-199), 16-byte alignment of every physical row, and the arguments the lowering hands to the dense convolution /
the decimation (recorded with stand-in callbacks; no kernel runs)."""
import pytest
import torch
import torch.nn.functional as F

import svc_autograd as A


@pytest.mark.parametrize("period", [2, 3, 5, 7, 11])
@pytest.mark.parametrize("T", [8192, 16384, 2048, 1000, 96, 4097])
def test_block_counts_match_conv2d_and_rows_are_aligned(period, T):
    n_pad = (period - T % period) % period
    H = (T + n_pad) // period
    x = torch.zeros(1, 1, H, period)
    for (k, s, pad) in [(5, 3, 2)] * 4 + [(5, 1, 2), (3, 1, 1)]:
        x = F.conv2d(x, torch.zeros(1, 1, k, 1), None, (s, 1), (pad, 0))
        H = (H + 2 * pad - k) // s + 1                      # models.DiscriminatorP.forward
        assert H == x.shape[2] and H >= 1
        Hp = A.align_blocks(H, period)
        assert Hp >= H and (Hp * period) % 4 == 0 and Hp - H < 4


@pytest.mark.parametrize("period,Hin_logical", [(3, 911), (11, 249), (5, 61), (2, 51), (7, 131)])
def test_strided_lowering_with_out_blocks(period, Hin_logical, monkeypatch):
    """conv1d(..., stride 3, inner=p, out_blocks) on a padded input: the decimation is asked for an aligned block count that
    covers every input block, and the dense conv for exactly out_blocks*p columns with dilation p."""
    Hp_in = A.align_blocks(Hin_logical, period)
    x = torch.zeros(2, 4, Hp_in * period)
    Hout = (Hin_logical + 4 - 5) // 3 + 1
    Hp_out = A.align_blocks(Hout, period)
    seen = {}

    class FakeDec:
        @staticmethod
        def apply(xx, s, off, Q, lp, inner):
            seen["dec"] = (s, off, Q, lp, inner)
            return torch.zeros(xx.shape[0], s * xx.shape[1], Q * inner)
    monkeypatch.setattr(A, "_Decimate", FakeDec)

    def dense(xx, pad, dil, tout=None, exact=False, strided=None):
        seen["dense"] = (tuple(xx.shape), pad, dil, tout, exact, strided)
        return torch.zeros(xx.shape[0], 8, tout)
    y = A._conv1d_lowered(dense, x, 5, 3, 2, 1, period, None, Hp_out)
    s, off, Q, lp, inner = seen["dec"]
    assert (s, off, lp, inner) == (3, 0, None, period)
    assert Q * 3 >= Hp_in and (Q * period) % 4 == 0                      # covers the input, aligned rows
    shape, pad, dil, tout, exact, strided = seen["dense"]
    KSd, shift, m_min = A.strided_geometry(5, 3, 2)
    assert shape == (2, 12, Q * period) and dil == period and pad == -m_min * period
    assert tout == Hp_out * period and exact and strided == (KSd, shift)
    assert y.shape[2] % 4 == 0
