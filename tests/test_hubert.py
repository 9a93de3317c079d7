"""HuBERT-soft unit encoder (reference vencoder/hubert/hubert_model.py, vencoder/HubertSoft.py).
CPU: the oracle restatement against the vector of the REAL in-tree module.  GPU: the HIP mirror against that vector and
against the oracle on other lengths (incl. a length whose conv stack leaves odd intermediate sizes).
Tolerance: 2e-4 of max|ref| (12 post-norm transformer layers of fp32 MFMA / fmaf-chain reductions)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import hubert_oracle as HO

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _golden():
    z = np.load(os.path.join(G, "hubert_soft_1s.npz"))
    return z, json.loads(str(z["meta"]))


def test_oracle_reproduces_reference_hubert_units():
    z, meta = _golden()
    sd = HO.make_state_dict(meta["seed"])
    assert len(sd) == 166 and sum(v.numel() for v in sd.values()) == 94594176
    with torch.no_grad():
        u = HO.units(sd, torch.from_numpy(z["wav"]))
    assert u.shape == z["units"].shape
    assert np.abs(u.numpy() - z["units"]).max() <= 2e-5 * max(1.0, np.abs(z["units"]).max())


def _mirror(seed, dev):
    from vencoder.hubert import hubert_model as HM
    net = HM.HubertSoft()
    missing, unexpected = net.load_state_dict(HO.make_state_dict(seed), strict=True)
    return net.to(dev).eval()


@pytest.mark.gpu
def test_hubert_units_match_reference_golden(dev):
    z, meta = _golden()
    net = _mirror(meta["seed"], dev)
    u = net.units(torch.from_numpy(z["wav"]).to(dev))
    ref = torch.from_numpy(z["units"])
    assert u.shape == ref.shape
    err = (u.cpu() - ref).abs().max().item()
    assert err <= 2e-4 * max(1.0, ref.abs().max().item()), err
    # the SpeechEncoder wrapper Svc talks to (vencoder/HubertSoft.py): [n] -> [1, 256, T]
    from vencoder.HubertSoft import HubertSoft
    enc = HubertSoft(device=dev, model=net)
    c = enc.encoder(torch.from_numpy(z["wav"][0, 0]).to(dev))
    assert c.shape == (1, 256, ref.shape[1]) and torch.equal(c, u.transpose(1, 2))


@pytest.mark.gpu
@pytest.mark.parametrize("B,n", [(1, 40321), (2, 8000), (1, 401)])
def test_hubert_units_match_oracle(dev, B, n):
    sd = HO.make_state_dict(5)
    net = _mirror(5, dev)
    g = torch.Generator().manual_seed(n)
    wav = 0.3 * torch.randn(B, 1, n, generator=g)
    with torch.no_grad():
        ref = HO.units(sd, wav)
    u = net.units(wav.to(dev))
    assert u.shape == ref.shape
    err = (u.cpu() - ref).abs().max().item()
    assert err <= 2e-4 * max(1.0, ref.abs().max().item()), err


@pytest.mark.gpu
def test_unit_encoder_batches_unequal_lengths_exactly(dev):
    """Svc.slice_inference's chunks (reference inference/infer_tool.py:446-495 + :220-224: one encoder call per chunk) as padded
    batches with per-item lengths: masked GroupNorm statistics (svc_channel_norm_gelu_len_f32), zero-masked positional-conv input,
    padding-mask attention.  Every item must equal its own serial encoding (<= 2e-5) — for the soft-unit wrapper (40-sample
    padding, proj) and for the ContentVec-style path (layer-12 features), including groups split by the padding-waste rule."""
    from vencoder.HubertSoft import HubertSoft
    net = _mirror(7, dev)
    enc = HubertSoft(device=dev, model=net)
    g = torch.Generator().manual_seed(3)
    lens = [16000, 13337, 16000, 9001, 12000, 401, 15999]
    wavs = [(0.3 * torch.randn(n, generator=g)).to(dev) for n in lens]
    serial = [enc.encoder(w) for w in wavs]
    calls = []
    orig = net.encode
    net.encode = lambda x, layer=None, lengths=None: (calls.append((x.shape[0], lengths)), orig(x, layer=layer, lengths=lengths))[1]
    batched = enc.encoder_batch(wavs)
    net.encode = orig
    assert len(calls) < len(lens) and max(b for b, _ in calls) >= 3, calls          # really batched, unequal lengths together
    assert any(l is not None and len(set(l)) > 1 for _, l in calls)
    for n, a, b in zip(lens, serial, batched):
        assert a.shape == b.shape, (n, a.shape, b.shape)
        err = (a - b).abs().max().item()
        assert err <= 2e-5 * max(1.0, a.abs().max().item()), (n, err)
    # the raw stack at an inner layer (what ContentVec768L12 / 256L9 read), unequal items in ONE call
    x = torch.zeros(3, 1, 24000, device=dev)
    ns = [24000, 17003, 20480]
    for b, n in enumerate(ns):
        x[b, 0, :n] = 0.3 * torch.randn(n, generator=g).to(dev)
    y, _ = net.encode(x, layer=9, lengths=ns)
    frames = list(net.last_frames)
    for b, n in enumerate(ns):
        yb, _ = net.encode(x[b:b + 1, :, :n].contiguous(), layer=9)
        assert yb.shape[2] == frames[b]
        err = (y[b:b + 1, :, :frames[b]] - yb).abs().max().item()
        assert err <= 2e-5 * max(1.0, yb.abs().max().item()), (n, err)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(1, 500), (2, 77), (1, 31), (1, 130)])
def test_positional_conv_matches_torch(dev, B, T):
    """csrc/posconv.hip (grouped k = 128 conv on the matrix pipe + GELU + residual in one launch) against torch's
    weight-normed Conv1d(768, 768, 128, padding=64, groups=16) exactly as the reference module computes it
    (vencoder/hubert/hubert_model.py:116-129)."""
    import svc_hip as S
    g = torch.Generator().manual_seed(B * 1000 + T)
    conv = torch.nn.Conv1d(768, 768, 128, padding=64, groups=16)
    conv = torch.nn.utils.weight_norm(conv, name="weight", dim=2)
    with torch.no_grad():
        conv.weight_v.copy_(torch.randn(conv.weight_v.shape, generator=g) * 0.02)
        conv.weight_g.copy_(torch.rand(conv.weight_g.shape, generator=g) + 0.5)
        conv.bias.copy_(torch.randn(768, generator=g) * 0.1)
    x = torch.randn(B, 768, T, generator=g)
    with torch.no_grad():
        ref = x + torch.nn.functional.gelu(conv(x)[:, :, :-1])
    wp = S.posconv_pack(conv.weight_v.detach().to(dev), conv.weight_g.detach().to(dev), groups=16)
    y = S.posconv(x.to(dev), wp, conv.bias.detach().to(dev), pad=64)
    torch.cuda.synchronize()
    err = (y.cpu() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err
