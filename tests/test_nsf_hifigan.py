"""Stand-alone NSF-HiFiGAN vocoder (vdecoder.nsf_hifigan.models.Generator / load_model / load_config;
reference vdecoder/nsf_hifigan/models.py:17-35,93-281).  CPU: oracle vs the REAL module's vector.  GPU: the HIP mirror
vs that vector and vs the oracle on another shape (incl. a high constant f0 whose 9th harmonic is ~8 kHz), plus the
load_model() checkpoint/config round trip.  Tolerance: the generator bar of the synthesizer tests."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import nsf_hifigan_oracle as NO

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _golden():
    z = np.load(os.path.join(G, "nsf_hifigan_small.npz"))
    return z, json.loads(str(z["meta"]))


def test_oracle_reproduces_reference_vocoder():
    z, meta = _golden()
    h = NO.small_h()
    sd = NO.make_state_dict(h, meta["seed"])
    t = lambda k: torch.from_numpy(z[k])
    with torch.no_grad():
        y = NO.generator(sd, h, t("mel"), t("f0"), t("rand_ini"), t("noise"))
    assert np.abs(y.numpy() - z["y"]).max() <= 5e-6 * max(np.abs(z["y"]).max(), 1.0)


def _check(o, ref):
    o, ref = o.float().cpu(), ref.float()
    assert (o - ref).pow(2).mean().item() < 1e-4
    assert (o - ref).abs().max().item() <= 2e-4 * max(ref.abs().max().item(), 1e-3)


@pytest.mark.gpu
def test_vocoder_matches_reference_golden_and_load_model(dev, tmp_path):
    from vdecoder.nsf_hifigan import models as M
    from vdecoder.nsf_hifigan.env import AttrDict
    z, meta = _golden()
    h = NO.small_h()
    sd = NO.make_state_dict(h, meta["seed"])
    # checkpoint + config.json in the reference's layout (:17-35)
    d = str(tmp_path)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(h, f)
    torch.save({"generator": sd}, os.path.join(d, "model"))
    net, hh = M.load_model(os.path.join(d, "model"), device=dev)
    assert isinstance(hh, AttrDict) and hh.num_mels == h["num_mels"]
    t = lambda k: torch.from_numpy(z[k]).to(dev)
    y = net(t("mel"), t("f0"), noise=dict(rand_ini=t("rand_ini"), sine=t("noise")))
    assert y.shape == z["y"].shape
    _check(y, torch.from_numpy(z["y"]))


@pytest.mark.gpu
def test_vocoder_matches_oracle(dev):
    from vdecoder.nsf_hifigan import models as M
    from vdecoder.nsf_hifigan.env import AttrDict
    h = NO.small_h()
    sd = NO.make_state_dict(h, 5)
    net = M.Generator(AttrDict(dict(h)))
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    g = torch.Generator().manual_seed(9)
    B, T = 3, 131
    mel = torch.randn(B, h["num_mels"], T, generator=g)
    f0 = 80 + 400 * torch.rand(B, T, generator=g)
    f0[0, 20:40] = 0
    f0[2, :] = 880.0
    upp = int(np.prod(h["upsample_rates"]))
    rand_ini = torch.rand(B, 9, generator=g)
    noise = torch.randn(B, T * upp, 9, generator=g)
    with torch.no_grad():
        ref = NO.generator(sd, h, mel, f0, rand_ini, noise)
    y = net(mel.to(dev), f0.to(dev), noise=dict(rand_ini=rand_ini.to(dev), sine=noise.to(dev)))
    _check(y, ref)
