"""MultiPeriodDiscriminator(use_spectral_norm=True) (models.py:170,205,230-252): the oracle against the REAL reference's vectors
(tests/golden/mpd_spectral_small.npz, tests/golden/make_golden_spectral.py) on CPU, the HIP path against both on the GPU."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import train_oracle as TO
from oracle import weights as W

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _case():
    z = np.load(os.path.join(G, "mpd_spectral_small.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return z, meta, W.make_mpd_sn_state_dict(meta["seed"])


def test_oracle_reproduces_reference_spectral_norm_discriminators():
    z, meta, sd = _case()
    so = {k: (v.clone().requires_grad_(True) if k.endswith(("weight_orig", "bias")) else v.clone()) for k, v in sd.items()}
    rs, gs, frs, fgs = TO.mpd(so, torch.from_numpy(z["y"]), torch.from_numpy(z["y_hat"]))
    loss = sum(((1 - r) ** 2).mean() + (g ** 2).mean() for r, g in zip(rs, gs))
    assert abs(float(loss) - float(z["loss"])) <= 1e-5 * max(1.0, abs(float(z["loss"])))
    for i, (r, g) in enumerate(zip(rs, gs)):
        assert np.abs(r.detach().numpy() - z[f"logit_r.{i}"]).max() <= 1e-5 * max(1.0, np.abs(z[f"logit_r.{i}"]).max())
        assert np.abs(g.detach().numpy() - z[f"logit_g.{i}"]).max() <= 1e-5 * max(1.0, np.abs(z[f"logit_g.{i}"]).max())
    keys = [str(k) for k in z["gnorm_keys"]]
    grads = dict(zip(keys, torch.autograd.grad(loss, [so[k] for k in keys])))
    for k, n in zip(keys, z["gnorm"]):
        assert abs(grads[k].norm().item() - n) <= 1e-4 * max(n, 1e-6), k
    for name in z.files:
        if name.startswith("grad."):
            assert np.abs(grads[name[5:]].numpy() - z[name]).max() <= 1e-4 * max(np.abs(z[name]).max(), 1e-9), name
        if name.startswith("after."):           # the buffers after two power iterations (y, then y_hat)
            assert np.abs(so[name[6:]].numpy() - z[name]).max() <= 1e-5, name


@pytest.mark.gpu
def test_spectral_norm_discriminators_match_reference(dev):
    """The HIP path (svc_spectral_norm_{fwd,bwd}_f32 + the conv kernels): logits, last 1024-channel feature maps, LSGAN
    discriminator loss, its gradients with respect to weight_orig / bias, and the weight_u / weight_v buffers after the call,
    against the REAL reference; then eval mode (no power iteration: buffers unchanged, same function of the stored vectors)."""
    import models
    from modules.losses import discriminator_loss
    z, meta, sd = _case()
    net = models.MultiPeriodDiscriminator(use_spectral_norm=True)
    assert set(net.state_dict()) == set(sd)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).train()
    y, y_hat = torch.from_numpy(z["y"]).to(dev), torch.from_numpy(z["y_hat"]).to(dev)
    rs, gs, frs, fgs = net(y, y_hat)
    for i, (r, g) in enumerate(zip(rs, gs)):
        assert r.shape == z[f"logit_r.{i}"].shape
        assert np.abs(r.detach().cpu().numpy() - z[f"logit_r.{i}"]).max() <= 2e-4 * max(1.0, np.abs(z[f"logit_r.{i}"]).max())
        assert np.abs(g.detach().cpu().numpy() - z[f"logit_g.{i}"]).max() <= 2e-4 * max(1.0, np.abs(z[f"logit_g.{i}"]).max())
        f = fgs[i][-2].detach().cpu().numpy()
        assert f.shape == z[f"fmap_g_last.{i}"].shape
        assert np.abs(f - z[f"fmap_g_last.{i}"]).max() <= 2e-4 * max(1.0, np.abs(z[f"fmap_g_last.{i}"]).max())
    loss, _, _ = discriminator_loss(rs, gs)
    assert abs(float(loss) - float(z["loss"])) <= 1e-4 * max(1.0, abs(float(z["loss"])))
    loss.backward()
    gg = {k: p.grad.detach().cpu() for k, p in net.named_parameters()}
    for k, n in zip([str(k) for k in z["gnorm_keys"]], z["gnorm"]):
        assert abs(gg[k].norm().item() - n) <= 5e-3 * max(n, 1e-6), (k, gg[k].norm().item(), n)
    after = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    for name in z.files:
        if name.startswith("grad."):
            d = np.abs(gg[name[5:]].numpy() - z[name])
            m = max(np.abs(z[name]).max(), 1e-9)
            assert d.max() <= 5e-2 * m and (d > 1e-3 * m).mean() <= 0.10, name       # robust to leaky-ReLU branch flips
        if name.startswith("after."):
            assert np.abs(after[name[6:]] - z[name]).max() <= 1e-5, name
    # eval: no power iteration
    net.eval()
    before = {k: v.clone() for k, v in net.state_dict().items() if k.endswith(("weight_u", "weight_v"))}
    with torch.no_grad():
        r1 = net(y, y_hat)[0]
        r2 = net(y, y_hat)[0]
    for k, v in net.state_dict().items():
        if k in before:
            assert torch.equal(v, before[k]), k
    assert all(torch.equal(a, b) for a, b in zip(r1, r2))
