"""Golden vectors for MultiPeriodDiscriminator(use_spectral_norm=True) (models.py:170,205,230-252) from the REAL reference
(build container only; see make_golden.py): one net_d(y, y_hat) call in train() mode on short signals — logits, the last
feature map of every sub-discriminator, the LSGAN discriminator loss, the gradient of that loss with respect to every
weight_orig / bias (norms of all, three in full) and the weight_u / weight_v buffers AFTER the call (every layer has run two
power iterations: once for y, once for y_hat).  Asserts that oracle/train_oracle.py reproduces all of it.

usage: python tests/golden/make_golden_spectral.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402

FULL = ["discriminators.0.convs.2.weight_orig", "discriminators.3.convs.1.weight_orig", "discriminators.5.conv_post.weight_orig"]
BUFS = ["discriminators.0.convs.1", "discriminators.0.convs.5", "discriminators.2.convs.0", "discriminators.4.convs.3",
        "discriminators.5.conv_post"]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from oracle import train_oracle as TO
    from oracle import weights as W
    models, _ = import_reference()
    from modules.losses import discriminator_loss
    seed, B, T = 31, 2, 2048
    sd = W.make_mpd_sn_state_dict(seed)
    net = models.MultiPeriodDiscriminator(use_spectral_norm=True)
    assert set(net.state_dict()) == set(sd), (sorted(set(net.state_dict()) ^ set(sd))[:6])
    net.load_state_dict(sd)
    net.train()
    gen = torch.Generator().manual_seed(seed)
    y = torch.randn(B, 1, T, generator=gen) * 0.5
    y_hat = torch.randn(B, 1, T, generator=gen) * 0.5
    rs, gs, frs, fgs = net(y, y_hat)
    loss, _, _ = discriminator_loss(rs, gs)
    loss.backward()
    gg = {k: p.grad.clone() for k, p in net.named_parameters()}
    after = {k: v.clone() for k, v in net.state_dict().items()}

    # ---- oracle vs reference ----
    so = {k: (v.clone().requires_grad_(True) if k.endswith(("weight_orig", "bias")) else v.clone()) for k, v in sd.items()}
    ors, ogs, ofr, ofg = TO.mpd(so, y, y_hat)
    oloss = sum(((1 - r) ** 2).mean() + (g ** 2).mean() for r, g in zip(ors, ogs))
    assert abs(float(oloss) - float(loss)) <= 1e-6 * max(1.0, abs(float(loss))), (float(oloss), float(loss))
    for a, b in zip(ors + ogs, rs + gs):
        assert (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item())
    keys = list(gg)
    og = torch.autograd.grad(oloss, [so[k] for k in keys])
    worst = max((g - gg[k]).abs().max().item() / max(gg[k].abs().max().item(), 1e-9) for k, g in zip(keys, og))
    print("oracle: loss", float(oloss), "vs", float(loss), " worst relative gradient error", worst)
    assert worst <= 1e-4
    for p in BUFS:
        for b in ("weight_u", "weight_v"):
            assert (so[f"{p}.{b}"] - after[f"{p}.{b}"]).abs().max().item() <= 1e-6, (p, b)
    np.savez_compressed(
        os.path.join(HERE, "mpd_spectral_small.npz"),
        y=y.numpy(), y_hat=y_hat.numpy(), loss=np.float64(float(loss)),
        **{f"logit_r.{i}": r.detach().numpy() for i, r in enumerate(rs)}, **{f"logit_g.{i}": g.detach().numpy() for i, g in enumerate(gs)},
        **{f"fmap_g_last.{i}": f[-2].detach().numpy() for i, f in enumerate(fgs)},
        gnorm_keys=np.array(keys), gnorm=np.array([gg[k].norm().item() for k in keys], dtype=np.float64),
        **{f"grad.{k}": gg[k].numpy() for k in FULL},
        **{f"after.{p}.{b}": after[f"{p}.{b}"].numpy() for p in BUFS for b in ("weight_u", "weight_v")},
        meta=json.dumps(dict(seed=seed, B=B, T=T)))
    print("wrote mpd_spectral_small.npz")


if __name__ == "__main__":
    main()
