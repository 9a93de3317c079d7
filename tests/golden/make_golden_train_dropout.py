"""Golden vectors for the TRAINING graph WITH dropout (p_dropout = 0.1, the shipped configs' value) from the REAL
reference (build container only; see make_golden.py / make_golden_train.py).

Same case as train_small.npz but with cfg["p_dropout"] = 0.1 and net_g.train(): every active nn.Dropout site — the
attention probabilities (modules/attentions.py:232), the attention / FFN outputs (:51,:55,:100,:104) and the FFN hidden
activations (:344) of f0_decoder and enc_p — draws its keep mask from an injected uniform tensor (torch.nn.functional.dropout
is replaced by  x * (u >= p) / (1 - p), the same distribution torch's own dropout samples).  Stores the generator-side
losses (the discriminator is not involved), the prior statistics, pred_lf0, and the gradient of loss_kl + loss_lf0 with
respect to every enc_p / f0_decoder / pre / embedding parameter (norms of all, a few in full).  Asserts that
oracle/train_oracle.py reproduces all of it.

usage: python tests/golden/make_golden_train_dropout.py
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
from make_golden_train import DATA, Injector  # noqa: E402

P_DROP = 0.1
FULL_GRADS = ["pre.weight", "enc_p.enc_.attn_layers.0.emb_rel_k", "enc_p.enc_.attn_layers.1.conv_v.weight",
              "enc_p.enc_.ffn_layers.0.conv_1.weight", "f0_decoder.decoder.self_attn_layers.0.conv_q.weight",
              "f0_decoder.decoder.ffn_layers.1.conv_2.bias", "f0_decoder.proj.weight", "enc_p.f0_emb.weight"]


class DropoutInjector:
    """F.dropout(x, p, training) -> x * (u >= p) / (1 - p) with u popped from a queue (sites with p == 0 pass through)."""

    def __init__(self, us):
        self.us = list(us)
        self.sites = 0

    def __enter__(self):
        import torch.nn.functional as F
        self.orig = F.dropout
        inj = self

        def dropout(x, p=0.5, training=True, inplace=False):
            if not training or p == 0:
                return x
            u = inj.us.pop(0)
            assert tuple(u.shape) == tuple(x.shape), (u.shape, x.shape)
            inj.sites += 1
            return x * ((u >= p).to(x.dtype) * (1.0 / (1.0 - p)))

        F.dropout = dropout
        return self

    def __exit__(self, *a):
        import torch.nn.functional as F
        F.dropout = self.orig


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from oracle import train_oracle as TO
    from oracle import weights as W
    models, utils = import_reference()
    from modules.losses import kl_loss

    cfg = W.train_config()
    cfg["p_dropout"] = P_DROP
    cfg["spec_channels"] = DATA["n_fft"] // 2 + 1
    cfg.update(upsample_rates=[4, 2, 2, 2], upsample_kernel_sizes=[8, 4, 4, 4])
    B, T, seed = 2, 40, 21
    hop = DATA["hop"]
    sd_g = W.make_train_state_dict(cfg, seed)
    kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
    net_g = models.SynthesizerTrn(cfg["spec_channels"], cfg["segment_size"], **kw)
    net_g.load_state_dict(sd_g)
    net_g.train()
    c, f0, uv, spec, y, sid, lengths = W.make_train_batch(cfg, B, T, seed, hop=hop)
    noise = W.make_train_noise(cfg, B, T, lengths, seed + 2, hop=hop)
    noise["dropout_u"] = W.make_dropout_draws(cfg, B, T, seed + 3)
    inj = Injector([noise["f0_factor"]], [noise["enc_p"], noise["enc_q"], noise["sine"], None],
                   [noise["ids_rand"], noise["rand_ini"]])
    dinj = DropoutInjector(noise["dropout_u"])
    with inj, dinj:
        y_hat, ids_slice, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), pred_lf0, norm_lf0, lf0 = net_g(
            c, f0, uv, spec, g=sid, c_lengths=lengths, spec_lengths=lengths)
    assert dinj.sites == len(noise["dropout_u"]) and not dinj.us, (dinj.sites, len(noise["dropout_u"]))
    loss_kl = kl_loss(z_p, logs_q, m_p, logs_p, z_mask)
    loss_lf0 = torch.nn.functional.mse_loss(pred_lf0, lf0)
    (loss_kl + loss_lf0).backward()
    gg = {k: p.grad.clone() for k, p in net_g.named_parameters() if p.grad is not None}
    print(dict(loss_kl=float(loss_kl), loss_lf0=float(loss_lf0), sites=dinj.sites))

    # the same case WITHOUT dropout must differ (guards against a silently inactive injector)
    net_g.eval()
    with Injector([noise["f0_factor"]], [noise["enc_p"], noise["enc_q"], noise["sine"], None], [noise["ids_rand"], noise["rand_ini"]]):
        out0 = net_g(c, f0, uv, spec, g=sid, c_lengths=lengths, spec_lengths=lengths)
    assert (out0[3][2] - m_p).abs().max().item() > 1e-3

    # ---- oracle vs reference ----
    sg = {k: v.clone().requires_grad_(True) for k, v in sd_g.items()}
    o = TO.synth_forward(sg, cfg, c, f0, uv, spec, sid, lengths, lengths, noise)
    o_mp, o_logs_p, o_pred = o[3][2], o[3][3], o[4]
    assert (o_mp - m_p).abs().max().item() <= 2e-5 * max(1.0, m_p.abs().max().item())
    assert (o_pred - pred_lf0).abs().max().item() <= 2e-5 * max(1.0, pred_lf0.abs().max().item())
    o_kl = TO.kl_loss(o[3][1], o[3][5], o_mp, o_logs_p, o[2])
    o_lf0 = torch.nn.functional.mse_loss(o_pred, o[6])
    assert abs(float(o_kl) - float(loss_kl)) <= 2e-5 * max(1.0, abs(float(loss_kl)))
    assert abs(float(o_lf0) - float(loss_lf0)) <= 2e-5 * max(1.0, abs(float(loss_lf0)))
    keys = list(gg)
    og = torch.autograd.grad(o_kl + o_lf0, [sg[k] for k in keys], allow_unused=True)
    worst = 0.0
    for k, g in zip(keys, og):
        if g is None:
            assert gg[k].abs().max().item() == 0, k
            continue
        if k.endswith("conv_k.bias"):
            continue
        e = (g - gg[k]).abs().max().item() / max(gg[k].abs().max().item(), 1e-6)
        if e > worst:
            worst, wk = e, k
    print("oracle gradients with dropout: worst relative max-err", worst, wk)
    assert worst <= 2e-3

    np.savez_compressed(
        os.path.join(HERE, "train_dropout_small.npz"),
        m_p=m_p.detach().numpy(), logs_p=logs_p.detach().numpy(), pred_lf0=pred_lf0.detach().numpy(),
        loss_kl=np.float64(float(loss_kl)), loss_lf0=np.float64(float(loss_lf0)),
        gnorm_keys=np.array(keys), gnorm=np.array([gg[k].norm().item() for k in keys], dtype=np.float64),
        **{f"grad.{k}": gg[k].numpy() for k in FULL_GRADS},
        meta=json.dumps(dict(B=B, T=T, seed=seed, p_dropout=P_DROP, data=DATA, upsample_rates=cfg["upsample_rates"],
                             upsample_kernel_sizes=cfg["upsample_kernel_sizes"], n_sites=dinj.sites)))
    print("wrote train_dropout_small.npz")


if __name__ == "__main__":
    main()
