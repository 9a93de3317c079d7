"""Shallow-diffusion model (reference diffusion/{wavenet,diffusion,unit2mel}.py).
CPU: the oracle against vectors of the REAL modules (DDIM, PNDM, ancestral; full and shallow).  GPU: the HIP mirror
(Unit2Mel -> GaussianDiffusion -> WaveNet on libsvc_hip.so) against the same vectors.  Tolerance: 1e-3 of max|ref| on the
de-normalised mel (10..12 chained denoiser calls; the reference is fp32 as well), WaveNet alone 2e-5."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import diffusion_oracle as DO

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = [("ddim_full", "ddim", 10, False, None), ("pndm_full", "pndm", 10, False, None),
         ("ddim_shallow", "ddim", 5, True, 40), ("naive_shallow", None, 1, True, None)]


def _load():
    z = np.load(os.path.join(G, "diffusion_small.npz"))
    return z, json.loads(str(z["meta"]))


@pytest.mark.parametrize("name,method,speedup,shallow,k_step", CASES)
def test_oracle_reproduces_reference_samplers(name, method, speedup, shallow, k_step):
    z, meta = _load()
    c = DO.small_cfg()
    sd = DO.make_state_dict(c, meta["seed"])
    t = lambda k: torch.from_numpy(z[k])
    nb = 1 if method == "pndm" else meta["B"]
    cond = DO.condition(sd, c, t("units"), t("f0"), t("volume"), t("spk_id"))[:nb]
    k_step = meta["K"] if name == "naive_shallow" else k_step
    with torch.no_grad():
        mel = DO.sample(sd, c, cond, method, speedup, gt_spec=t("gt")[:nb] if shallow else None, k_step=k_step,
                        x_T=t("x_T")[:nb], step_noise=list(t("steps")))
    ref = z["mel_" + name]
    assert np.abs(mel.numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def _mirror(c, seed, dev):
    from diffusion.unit2mel import Unit2Mel
    net = Unit2Mel(c["input_channel"], c["n_spk"], c["use_pitch_aug"], c["out_dims"], c["n_layers"], c["n_chans"],
                   c["n_hidden"], c["timesteps"], c["k_step_max"])
    missing, unexpected = net.load_state_dict(DO.make_state_dict(c, seed), strict=False)
    assert not unexpected and all(k.startswith("decoder.") and "denoise_fn" not in k for k in missing)   # schedule buffers only
    return net.to(dev).eval()


@pytest.mark.gpu
def test_wavenet_matches_oracle(dev):
    c = DO.small_cfg()
    sd = DO.make_state_dict(c, 3)
    net = _mirror(c, 3, dev)
    g = torch.Generator().manual_seed(1)
    B, T = 2, 77
    spec = torch.randn(B, 1, c["out_dims"], T, generator=g)
    cond = torch.randn(B, c["n_hidden"], T, generator=g)
    for step in (0, 7, 99):
        t = torch.full((B,), step, dtype=torch.long)
        with torch.no_grad():
            ref = DO.wavenet(sd, c, spec, t, cond)
        out = net.decoder.denoise_fn(spec.to(dev), t.to(dev), cond=cond.to(dev))
        err = (out.cpu() - ref).abs().max().item()
        assert err <= 2e-5 * max(1.0, ref.abs().max().item()), (step, err)


@pytest.mark.gpu
@pytest.mark.parametrize("name,method,speedup,shallow,k_step", CASES)
def test_unit2mel_matches_reference_golden(dev, name, method, speedup, shallow, k_step):
    z, meta = _load()
    c = DO.small_cfg()
    net = _mirror(c, meta["seed"], dev)
    nb = 1 if method == "pndm" else meta["B"]
    t = lambda k: torch.from_numpy(z[k])[:nb].to(dev)
    k_step = meta["K"] if name == "naive_shallow" else (k_step or 300)
    noise = dict(x_T=t("x_T"), steps=[s[:nb].to(dev) for s in torch.from_numpy(z["steps"])])
    mel = net(t("units"), t("f0"), t("volume"), spk_id=t("spk_id"), gt_spec=t("gt") if shallow else None, infer=True,
              infer_speedup=speedup, method=method, k_step=k_step, use_tqdm=False, noise=noise)
    ref = torch.from_numpy(z["mel_" + name])
    assert mel.shape == ref.shape
    err = (mel.cpu() - ref).abs().max().item()
    assert err <= 1e-3 * max(1.0, ref.abs().max().item()), err
