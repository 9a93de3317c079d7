"""Golden vectors for the TRAINING graph from the REAL reference (build container only; see make_golden.py).

Builds the reference's SynthesizerTrn (train mode, p_dropout=0) and MultiPeriodDiscriminator with the deterministic
synthetic checkpoints, runs models.py:463-493 + the loss assembly of train.py:167-207 with the random draws replaced by
explicit tensors (order: utils.py:39 uniform_, models.py:160 randn_like, :123 randn_like, modules/commons.py:20 rand,
vdecoder/hifigan/models.py:147 rand, :266 randn_like, :319 randn_like), backpropagates loss_disc and loss_gen_all, and
stores the scalar losses, y_hat, the gradient L2 norm of EVERY parameter tensor and a few full gradients.  Also asserts
that oracle/train_oracle.py reproduces all of it.

usage: python tests/golden/make_golden_train.py
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

DATA = dict(n_fft=128, hop=32, win=128, n_mels=20, sr=44100, fmin=0.0, fmax=22050)
FULL_GRADS = ["pre.weight", "emb_g.weight", "enc_p.enc_.attn_layers.0.emb_rel_k", "enc_p.f0_emb.weight",
              "flow.flows.0.post.bias", "dec.conv_post.weight_v", "dec.m_source.l_linear.weight",
              "dec.ups.0.weight_g", "f0_decoder.proj.weight", "enc_q.enc.cond_layer.weight_g"]
FULL_GRADS_D = ["discriminators.0.convs.1.weight_v", "discriminators.3.convs.0.weight_g", "discriminators.5.conv_post.bias"]


class Injector:
    def __init__(self, uniform, randn_likes, rands):
        self.uniform, self.randn_likes, self.rands = list(uniform), list(randn_likes), list(rands)

    def __enter__(self):
        self.o = (torch.randn_like, torch.rand, torch.Tensor.uniform_)
        inj = self

        def randn_like(t, **kw):
            n = inj.randn_likes.pop(0)
            if n is None:
                return inj.o[0](t, **kw)
            assert tuple(n.shape) == tuple(t.shape), (n.shape, t.shape)
            return n.clone()

        def rand(*size, **kw):
            n = inj.rands.pop(0)
            if len(size) == 1 and isinstance(size[0], (list, tuple)):
                size = tuple(size[0])
            assert tuple(n.shape) == tuple(size), (n.shape, size)
            return n.clone()

        def uniform_(self_t, a=0, b=1, **kw):
            n = inj.uniform.pop(0)
            assert tuple(n.shape) == tuple(self_t.shape)
            return n.clone()

        torch.randn_like, torch.rand, torch.Tensor.uniform_ = randn_like, rand, uniform_
        return self

    def __exit__(self, *a):
        torch.randn_like, torch.rand, torch.Tensor.uniform_ = self.o


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from oracle import mel as OM
    from oracle import train_oracle as TO
    from oracle import weights as W
    models, utils = import_reference()
    sys.modules["librosa.filters"].mel = lambda sr, n_fft, n_mels, fmin, fmax: OM.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    import modules.commons as commons
    import modules.mel_processing as MP
    from modules.losses import discriminator_loss, feature_loss, generator_loss, kl_loss
    MP.librosa_mel_fn = sys.modules["librosa.filters"].mel

    cfg = W.train_config()
    cfg["spec_channels"] = DATA["n_fft"] // 2 + 1
    # hop 32: four upsample stages whose product is 32
    cfg.update(upsample_rates=[4, 2, 2, 2], upsample_kernel_sizes=[8, 4, 4, 4])
    B, T, seed = 2, 40, 21
    hop = DATA["hop"]
    sd_g = W.make_train_state_dict(cfg, seed)
    sd_d = W.make_mpd_state_dict(seed + 1)
    kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
    net_g = models.SynthesizerTrn(cfg["spec_channels"], cfg["segment_size"], **kw)
    net_g.load_state_dict(sd_g)
    net_g.train()
    net_d = models.MultiPeriodDiscriminator()
    assert {k: tuple(v.shape) for k, v in net_d.state_dict().items()} == {k: tuple(v) for k, v in W.mpd_param_shapes().items()}
    net_d.load_state_dict(sd_d)
    net_d.train()

    c, f0, uv, spec, y, sid, lengths = W.make_train_batch(cfg, B, T, seed, hop=hop)
    noise = W.make_train_noise(cfg, B, T, lengths, seed + 2, hop=hop)
    L = cfg["segment_size"] * hop
    inj = Injector([noise["f0_factor"]], [noise["enc_p"], noise["enc_q"], noise["sine"], None],
                   [noise["ids_rand"], noise["rand_ini"]])
    mel = MP.spec_to_mel_torch(spec, DATA["n_fft"], DATA["n_mels"], DATA["sr"], DATA["fmin"], DATA["fmax"])
    with inj:
        y_hat, ids_slice, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), pred_lf0, norm_lf0, lf0 = net_g(
            c, f0, uv, spec, g=sid, c_lengths=lengths, spec_lengths=lengths)
    assert torch.equal(ids_slice, noise["ids_slice"]), (ids_slice, noise["ids_slice"])
    seg = cfg["segment_size"]
    y_mel = commons.slice_segments(mel, ids_slice, seg)
    y_hat_mel = MP.mel_spectrogram_torch(y_hat.squeeze(1), DATA["n_fft"], DATA["n_mels"], DATA["sr"], hop, DATA["win"],
                                         DATA["fmin"], DATA["fmax"])
    y_seg = commons.slice_segments(y, ids_slice * hop, seg * hop)
    rs, gs, _, _ = net_d(y_seg, y_hat.detach())
    loss_disc, _, _ = discriminator_loss(rs, gs)
    loss_disc.backward()
    gd = {k: p.grad.clone() for k, p in net_d.named_parameters()}
    net_d.zero_grad()
    rs, gs, fr, fg = net_d(y_seg, y_hat)
    loss_mel = torch.nn.functional.l1_loss(y_mel, y_hat_mel) * 45.0
    loss_kl = kl_loss(z_p, logs_q, m_p, logs_p, z_mask) * 1.0
    loss_fm = feature_loss(fr, fg)
    loss_gen, _ = generator_loss(gs)
    loss_lf0 = torch.nn.functional.mse_loss(pred_lf0, lf0)
    loss_gen_all = loss_gen + loss_fm + loss_mel + loss_kl + loss_lf0
    loss_gen_all.backward()
    gg = {k: p.grad.clone() for k, p in net_g.named_parameters() if p.grad is not None}
    ref = dict(loss_disc=loss_disc, loss_gen=loss_gen, loss_fm=loss_fm, loss_mel=loss_mel, loss_kl=loss_kl,
               loss_lf0=loss_lf0, loss_gen_all=loss_gen_all)
    print({k: float(v) for k, v in ref.items()})

    # ---- oracle vs reference ----
    sg = {k: v.clone().requires_grad_(True) for k, v in sd_g.items()}
    sdd = {k: v.clone().requires_grad_(True) for k, v in sd_d.items()}
    mb = torch.from_numpy(OM.mel_filterbank(DATA["sr"], DATA["n_fft"], DATA["n_mels"], DATA["fmin"], DATA["fmax"]))
    out = TO.gan_step_losses(sg, sdd, cfg, DATA, (c, f0, uv, spec, y, sid, lengths), noise, mb)
    for k in ref:
        d = abs(float(out[k]) - float(ref[k]))
        print(f"  {k:14s} ref {float(ref[k]):+.6e} oracle {float(out[k]):+.6e} diff {d:.2e}")
        assert d <= 2e-5 * max(1.0, abs(float(ref[k]))), k
    assert (out["y_hat"] - y_hat).abs().max().item() <= 1e-5 * max(1.0, y_hat.abs().max().item())
    og_d = torch.autograd.grad(out["loss_disc"], [sdd[k] for k in gd], retain_graph=True)
    worst_d = 0.0
    for k, g in zip(gd, og_d):
        e = (g - gd[k]).abs().max().item() / max(gd[k].abs().max().item(), 1e-6)
        if e > worst_d:
            worst_d, wk = e, k
    print("oracle D gradients: worst relative max-err", worst_d, wk)
    assert worst_d <= 2e-3
    keys_g = [k for k in gg]
    og_g = torch.autograd.grad(out["loss_gen_all"], [sg[k] for k in keys_g], allow_unused=True)
    worst = 0.0
    errs = []
    for k, g in zip(keys_g, og_g):
        if g is None:
            assert gg[k].abs().max().item() == 0, k
            continue
        if k.endswith("conv_k.bias"):
            continue   # softmax is invariant to a per-query constant: d/d(conv_k.bias) is exactly 0 up to round-off noise
        e = (g - gg[k]).abs().max().item() / max(gg[k].abs().max().item(), 1e-6)
        errs.append((e, k, gg[k].abs().max().item()))
        if e > worst:
            worst, wkg = e, k
    print("oracle G gradients: worst relative max-err", worst, wkg)
    for e in sorted(errs, reverse=True)[:5]:
        print("     ", e)
    assert worst <= 2e-3

    np.savez_compressed(
        os.path.join(HERE, "train_small.npz"),
        y_hat=y_hat.detach().numpy(),
        **{f"loss.{k}": np.float64(float(v)) for k, v in ref.items()},
        gnorm_g_keys=np.array(list(gg.keys())), gnorm_g=np.array([gg[k].norm().item() for k in gg], dtype=np.float64),
        gnorm_d_keys=np.array(list(gd.keys())), gnorm_d=np.array([gd[k].norm().item() for k in gd], dtype=np.float64),
        **{f"grad_g.{k}": gg[k].numpy() for k in FULL_GRADS},
        **{f"grad_d.{k}": gd[k].numpy() for k in FULL_GRADS_D},
        meta=json.dumps(dict(B=B, T=T, seed=seed, data=DATA, upsample_rates=cfg["upsample_rates"],
                             upsample_kernel_sizes=cfg["upsample_kernel_sizes"], c_mel=45.0, c_kl=1.0)))
    print("wrote train_small.npz")


if __name__ == "__main__":
    main()
