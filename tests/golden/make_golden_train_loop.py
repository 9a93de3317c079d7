"""Golden vectors for the TRAINING LOOP (optimizer included) from the REAL reference (build container only).

Runs N_ITER iterations of train.py:150-213 with the reference's own modules (SynthesizerTrn, MultiPeriodDiscriminator,
losses, mel_processing) and torch.optim.AdamW exactly as train.py:79-88 (betas (0.8, 0.99), eps 1e-9, default weight
decay), on the batch / injected noise of train_small.npz, repeating the same batch each iteration.  Stores the seven
scalar losses of every iteration and a few parameter tensors after the last one; asserts that
oracle/train_oracle.gan_train_loop reproduces them.

usage: python tests/golden/make_golden_train_loop.py
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

N_ITER = 3
LR = 2e-4
PARAMS_G = ["pre.bias", "dec.conv_post.weight_g", "flow.flows.0.pre.bias", "enc_p.proj.bias", "dec.ups.1.bias"]
PARAMS_D = ["discriminators.0.conv_post.bias", "discriminators.2.convs.0.weight_g"]


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
    net_d.load_state_dict(sd_d)
    net_d.train()
    optim_g = torch.optim.AdamW(net_g.parameters(), LR, betas=(0.8, 0.99), eps=1e-9)
    optim_d = torch.optim.AdamW(net_d.parameters(), LR, betas=(0.8, 0.99), eps=1e-9)

    c, f0, uv, spec, y, sid, lengths = W.make_train_batch(cfg, B, T, seed, hop=hop)
    noise = W.make_train_noise(cfg, B, T, lengths, seed + 2, hop=hop)
    seg = cfg["segment_size"]
    hist = []
    for it in range(N_ITER):
        inj = Injector([noise["f0_factor"]], [noise["enc_p"], noise["enc_q"], noise["sine"], None],
                       [noise["ids_rand"], noise["rand_ini"]])
        mel = MP.spec_to_mel_torch(spec, DATA["n_fft"], DATA["n_mels"], DATA["sr"], DATA["fmin"], DATA["fmax"])
        with inj:
            y_hat, ids_slice, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), pred_lf0, norm_lf0, lf0 = net_g(
                c, f0, uv, spec, g=sid, c_lengths=lengths, spec_lengths=lengths)
        y_mel = commons.slice_segments(mel, ids_slice, seg)
        y_hat_mel = MP.mel_spectrogram_torch(y_hat.squeeze(1), DATA["n_fft"], DATA["n_mels"], DATA["sr"], hop,
                                             DATA["win"], DATA["fmin"], DATA["fmax"])
        y_seg = commons.slice_segments(y, ids_slice * hop, seg * hop)
        rs, gs, _, _ = net_d(y_seg, y_hat.detach())
        loss_disc, _, _ = discriminator_loss(rs, gs)
        optim_d.zero_grad()
        loss_disc.backward()
        optim_d.step()
        rs, gs, fr, fg = net_d(y_seg, y_hat)
        loss_mel = torch.nn.functional.l1_loss(y_mel, y_hat_mel) * 45.0
        loss_kl = kl_loss(z_p, logs_q, m_p, logs_p, z_mask) * 1.0
        loss_fm = feature_loss(fr, fg)
        loss_gen, _ = generator_loss(gs)
        loss_lf0 = torch.nn.functional.mse_loss(pred_lf0, lf0)
        loss_gen_all = loss_gen + loss_fm + loss_mel + loss_kl + loss_lf0
        optim_g.zero_grad()
        loss_gen_all.backward()
        optim_g.step()
        hist.append({k: float(v) for k, v in dict(loss_disc=loss_disc, loss_gen=loss_gen, loss_fm=loss_fm,
                                                   loss_mel=loss_mel, loss_kl=loss_kl, loss_lf0=loss_lf0,
                                                   loss_gen_all=loss_gen_all).items()})
        print(it, hist[-1])

    mb = torch.from_numpy(OM.mel_filterbank(DATA["sr"], DATA["n_fft"], DATA["n_mels"], DATA["fmin"], DATA["fmax"]))
    ohist, osg, osd = TO.gan_train_loop(sd_g, sd_d, cfg, DATA, (c, f0, uv, spec, y, sid, lengths), noise, mb, N_ITER, lr=LR)
    for it in range(N_ITER):
        for k in hist[it]:
            d = abs(ohist[it][k] - hist[it][k])
            print(f"  it{it} {k:14s} ref {hist[it][k]:+.6e} oracle {ohist[it][k]:+.6e} diff {d:.2e}")
            assert d <= 1e-3 * max(1.0, abs(hist[it][k])), (it, k)
    pg = dict(net_g.named_parameters())
    pd = dict(net_d.named_parameters())
    for k in PARAMS_G:
        e = (osg[k] - pg[k].detach()).abs().max().item()
        print("  param", k, "max diff", e, "moved", (pg[k].detach() - sd_g[k]).abs().max().item())
    np.savez_compressed(
        os.path.join(HERE, "train_loop_small.npz"),
        **{f"it{it}.{k}": np.float64(v) for it in range(N_ITER) for k, v in hist[it].items()},
        **{f"param_g.{k}": pg[k].detach().numpy() for k in PARAMS_G},
        **{f"param_d.{k}": pd[k].detach().numpy() for k in PARAMS_D},
        meta=json.dumps(dict(n_iter=N_ITER, lr=LR, betas=[0.8, 0.99], eps=1e-9)))
    print("wrote train_loop_small.npz")


if __name__ == "__main__":
    main()
