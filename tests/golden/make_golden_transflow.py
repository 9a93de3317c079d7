"""Golden vectors for `use_transformer_flow=True` (models.py:438-439: TransformerCouplingBlock :54-92,
modules.TransformerCouplingLayer modules/modules.py:309-356, attentions.FFT(isflow=True) modules/attentions.py:24-61)
from the REAL reference (build container only; see make_golden.py).

  infer_transflow_T40         small config + use_transformer_flow, separate FFT per coupling, B=2, T=40
  infer_transflow_shared_T40  the same with flow_share_parameter (one FFT registered as flow.wn and as every coupling's enc)
  train_transflow_small       one generator forward in train() mode with p_dropout = 0.1 (the flow's FFT layers are dropout
                              sites, drawn after f0_decoder's and enc_p's), loss_kl + loss_lf0 and its gradient with respect
                              to every parameter that receives one (norms of all, a few of the flow's in full)

usage: python tests/golden/make_golden_transflow.py
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
from make_golden import import_reference, run_case  # noqa: E402
from make_golden_train import DATA, Injector  # noqa: E402
from make_golden_train_dropout import DropoutInjector  # noqa: E402

P_DROP = 0.1
FULL_GRADS = ["flow.flows.0.pre.weight", "flow.flows.0.enc.cond_pre.weight", "flow.flows.2.enc.cond_layer.weight_v",
              "flow.flows.2.enc.cond_layer.weight_g", "flow.flows.4.enc.self_attn_layers.1.conv_k.weight",
              "flow.flows.6.enc.ffn_layers.0.conv_1.weight", "flow.flows.6.post.bias", "enc_q.proj.weight", "emb_g.weight"]


def train_case(models):
    from oracle import train_oracle as TO
    from oracle import weights as W
    from modules.losses import kl_loss
    cfg = W.train_config()
    cfg.update(p_dropout=P_DROP, use_transformer_flow=True, n_layers_trans_flow=2)
    cfg["spec_channels"] = DATA["n_fft"] // 2 + 1
    cfg.update(upsample_rates=[4, 2, 2, 2], upsample_kernel_sizes=[8, 4, 4, 4])
    B, T, seed = 2, 40, 23
    hop = DATA["hop"]
    sd_g = W.make_train_state_dict(cfg, seed)
    kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
    net_g = models.SynthesizerTrn(cfg["spec_channels"], cfg["segment_size"], **kw)
    assert set(net_g.state_dict()) == set(sd_g)
    net_g.load_state_dict(sd_g)
    net_g.train()
    c, f0, uv, spec, y, sid, lengths = W.make_train_batch(cfg, B, T, seed, hop=hop)
    noise = W.make_train_noise(cfg, B, T, lengths, seed + 2, hop=hop)
    noise["dropout_u"] = W.make_dropout_draws(cfg, B, T, seed + 3)

    def fresh():
        return Injector([noise["f0_factor"]], [noise["enc_p"], noise["enc_q"], noise["sine"], None],
                        [noise["ids_rand"], noise["rand_ini"]])
    dinj = DropoutInjector(noise["dropout_u"])
    with fresh(), dinj:
        y_hat, ids_slice, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), pred_lf0, norm_lf0, lf0 = net_g(
            c, f0, uv, spec, g=sid, c_lengths=lengths, spec_lengths=lengths)
    assert dinj.sites == len(noise["dropout_u"]) and not dinj.us, (dinj.sites, len(noise["dropout_u"]))
    loss_kl = kl_loss(z_p, logs_q, m_p, logs_p, z_mask)
    loss_lf0 = torch.nn.functional.mse_loss(pred_lf0, lf0)
    (loss_kl + loss_lf0).backward()
    gg = {k: p.grad.clone() for k, p in net_g.named_parameters() if p.grad is not None}
    print(dict(loss_kl=float(loss_kl), loss_lf0=float(loss_lf0), sites=dinj.sites))

    sg = {k: v.clone().requires_grad_(True) for k, v in sd_g.items()}
    o = TO.synth_forward(sg, cfg, c, f0, uv, spec, sid, lengths, lengths, noise)
    o_zp, o_mp = o[3][1], o[3][2]
    assert (o_zp - z_p).abs().max().item() <= 2e-5 * max(1.0, z_p.abs().max().item()), (o_zp - z_p).abs().max().item()
    o_kl = TO.kl_loss(o_zp, o[3][5], o_mp, o[3][3], o[2])
    o_lf0 = torch.nn.functional.mse_loss(o[4], o[6])
    assert abs(float(o_kl) - float(loss_kl)) <= 2e-5 * max(1.0, abs(float(loss_kl)))
    keys = list(gg)
    og = torch.autograd.grad(o_kl + o_lf0, [sg[k] for k in keys], allow_unused=True)
    worst, wk = 0.0, None
    for k, g in zip(keys, og):
        if g is None:
            assert gg[k].abs().max().item() == 0, k
            continue
        if k.endswith("conv_k.bias"):
            continue
        e = (g - gg[k]).abs().max().item() / max(gg[k].abs().max().item(), 1e-6)
        if e > worst:
            worst, wk = e, k
    print("oracle gradients (transformer flow, dropout): worst relative max-err", worst, wk)
    assert worst <= 2e-3
    for k in FULL_GRADS:
        assert k in gg, k
    np.savez_compressed(
        os.path.join(HERE, "train_transflow_small.npz"),
        z_p=z_p.detach().numpy(), m_p=m_p.detach().numpy(), pred_lf0=pred_lf0.detach().numpy(),
        loss_kl=np.float64(float(loss_kl)), loss_lf0=np.float64(float(loss_lf0)),
        gnorm_keys=np.array(keys), gnorm=np.array([gg[k].norm().item() for k in keys], dtype=np.float64),
        **{f"grad.{k}": gg[k].numpy() for k in FULL_GRADS},
        meta=json.dumps(dict(B=B, T=T, seed=seed, p_dropout=P_DROP, n_layers_trans_flow=2, data=DATA,
                             upsample_rates=cfg["upsample_rates"], upsample_kernel_sizes=cfg["upsample_kernel_sizes"],
                             n_sites=dinj.sites)))
    print("wrote train_transflow_small.npz")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    models, utils = import_reference()
    from oracle import weights as W
    cfg = W.small_config()
    cfg["use_transformer_flow"] = True
    run_case(models, "transflow_T40", cfg, B=2, T=40, seed=16)
    shared = dict(cfg, flow_share_parameter=True, n_flow_layer=3, n_layers_trans_flow=2)
    run_case(models, "transflow_shared_T40", shared, B=2, T=40, seed=17)
    train_case(models)


if __name__ == "__main__":
    main()
