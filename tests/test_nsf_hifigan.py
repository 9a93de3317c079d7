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


# ---- nvSTFT log-mel (vdecoder/nsf_hifigan/nvSTFT.py) and the diffusion/vocoder.py wrapper -------------------------------
def test_oracle_reproduces_reference_nvstft_mel():
    z = np.load(os.path.join(G, "nvstft_mel.npz"))
    mel = NO.get_mel(torch.from_numpy(z["y"]))
    assert np.abs(mel.numpy() - z["mel"]).max() <= 1e-5


@pytest.mark.gpu
def test_nvstft_and_vocoder_wrapper_match_reference_golden(dev, tmp_path):
    """STFT.get_mel on the rocFFT path vs the real module's output (log-mel, 2e-4 absolute: fp32 FFT + log of small
    magnitudes), then the Vocoder wrapper: extract() layout and infer() == the generator it wraps."""
    from vdecoder.nsf_hifigan.nvSTFT import STFT
    from diffusion.vocoder import Vocoder
    z = np.load(os.path.join(G, "nvstft_mel.npz"))
    y = torch.from_numpy(z["y"]).to(dev)
    mel = STFT(44100, 128, 2048, 2048, 512, 40, 16000).get_mel(y)
    assert mel.shape == z["mel"].shape
    assert np.abs(mel.cpu().numpy() - z["mel"]).max() <= 2e-4
    h = dict(NO.small_h(), n_fft=2048, win_size=2048, hop_size=512, fmin=40, fmax=16000, num_mels=128)
    sd = NO.make_state_dict(h, 5)
    d = str(tmp_path)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(h, f)
    torch.save({"generator": sd}, os.path.join(d, "model"))
    voc = Vocoder("nsf-hifigan", os.path.join(d, "model"), device=dev)
    assert (voc.vocoder_sample_rate, voc.vocoder_hop_size, voc.dimension) == (44100, 512, 128)
    m2 = voc.extract(y, 44100)                                   # [B, frames, bins]
    assert torch.equal(m2, mel.transpose(1, 2))
    with pytest.raises(NotImplementedError):
        voc.extract(y, 16000)
    f0 = 220.0 + torch.zeros(2, m2.shape[1], 1, device=dev)
    torch.manual_seed(3)
    a = voc.infer(m2, f0)
    torch.manual_seed(3)
    b = voc.vocoder.model(m2.transpose(1, 2).contiguous(), f0[:, :, 0])
    assert torch.equal(a, b) and a.shape[-1] == m2.shape[1] * 32 and torch.isfinite(a).all()
