"""Golden vector for the reference's `fp16_run` + `half_type: bf16` training step, from the REAL reference (build container only).

Same model, batch and injected random draws as make_golden_train.py (-> train_small.npz, the fp32 step), run the way
train.py:166-211 runs it with `fp16_run: true, half_type: "bf16"`: generator forward, mel of y_hat and both discriminator forwards
inside `torch.autocast(dtype=torch.bfloat16)`, every loss under `autocast(enabled=False)`; the GradScaler only multiplies the loss
by a power of two and divides the gradients by it again (train.py:192-213), which changes nothing in bf16's exponent range and is
left out.  Device type "cpu": the only one this container has — CPU autocast lowers the same op classes (conv / linear / matmul)
to bf16 and keeps softmax / layer_norm / losses in fp32, like the CUDA list the reference runs under.

What the vector is for: the engine's bf16 mode (svc_conv1d_args.mma) keeps MORE in fp32 than autocast does (activations between
convolutions, the mel matmul, attention products), so it cannot reproduce these numbers to better than the reference's own
bf16 rounding noise; the file records that noise (distance of this step from the fp32 step, `amp_vs_fp32.*`) and the test
bounds the engine's distance from the golden by a small multiple of it.

`half_type: fp16` (second file, train_amp_fp16_small.npz): the same regions with dtype=float16 AND the GradScaler's arithmetic —
the loss is multiplied by the scale before backward and the gradients divided by it afterwards (train.py:192-213,
torch.cuda.amp.GradScaler defaults: init scale 65536, halved while a gradient overflows): without it fp16 gradients underflow.

usage: python tests/golden/make_golden_train_amp.py [bf16|fp16]
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


def main(half_type="bf16"):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from oracle import mel as OM
    from oracle import weights as W
    models, utils = import_reference()
    sys.modules["librosa.filters"].mel = lambda sr, n_fft, n_mels, fmin, fmax: OM.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    import modules.commons as commons
    import modules.mel_processing as MP
    from modules.losses import discriminator_loss, feature_loss, generator_loss, kl_loss
    MP.librosa_mel_fn = sys.modules["librosa.filters"].mel
    from torch import autocast

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
    c, f0, uv, spec, y, sid, lengths = W.make_train_batch(cfg, B, T, seed, hop=hop)
    noise = W.make_train_noise(cfg, B, T, lengths, seed + 2, hop=hop)
    def attempt(scale):
        net_g.zero_grad()
        net_d.zero_grad()
        inj = Injector([noise["f0_factor"]], [noise["enc_p"], noise["enc_q"], noise["sine"], None],
                       [noise["ids_rand"], noise["rand_ini"]])
        half = torch.bfloat16 if half_type == "bf16" else torch.float16
        mel = MP.spec_to_mel_torch(spec, DATA["n_fft"], DATA["n_mels"], DATA["sr"], DATA["fmin"], DATA["fmax"])       # train.py:158-164
        seg = cfg["segment_size"]
        with autocast("cpu", enabled=True, dtype=half):                                                               # :166
            with inj:
                y_hat, ids_slice, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), pred_lf0, norm_lf0, lf0 = net_g(
                    c, f0, uv, spec, g=sid, c_lengths=lengths, spec_lengths=lengths)
            y_mel = commons.slice_segments(mel, ids_slice, seg)
            y_hat_mel = MP.mel_spectrogram_torch(y_hat.squeeze(1), DATA["n_fft"], DATA["n_mels"], DATA["sr"], hop, DATA["win"],
                                                 DATA["fmin"], DATA["fmax"])
            y_seg = commons.slice_segments(y, ids_slice * hop, seg * hop)
            rs, gs, _, _ = net_d(y_seg, y_hat.detach())                                                               # :185
            with autocast("cpu", enabled=False):
                loss_disc, _, _ = discriminator_loss(rs, gs)
        assert torch.equal(ids_slice, noise["ids_slice"])
        (loss_disc * scale).backward()
        gd = {k: p.grad.clone().float() / scale for k, p in net_d.named_parameters()}
        if not all(torch.isfinite(g).all() for g in gd.values()):
            return None
        net_d.zero_grad()
        with autocast("cpu", enabled=True, dtype=half):                                                               # :198
            rs, gs, fr, fg = net_d(y_seg, y_hat)
            with autocast("cpu", enabled=False):
                loss_mel = torch.nn.functional.l1_loss(y_mel, y_hat_mel) * 45.0
                loss_kl = kl_loss(z_p, logs_q, m_p, logs_p, z_mask) * 1.0
                loss_fm = feature_loss(fr, fg)
                loss_gen, _ = generator_loss(gs)
                loss_lf0 = torch.nn.functional.mse_loss(pred_lf0, lf0)
                loss_gen_all = loss_gen + loss_fm + loss_mel + loss_kl + loss_lf0
        (loss_gen_all * scale).backward()
        gg = {k: p.grad.clone().float() / scale for k, p in net_g.named_parameters() if p.grad is not None}
        if not all(torch.isfinite(g).all() for g in gg.values()):
            return None
        return (y_hat, z_p, fr, y_hat_mel, gd, gg, loss_disc, loss_gen, loss_fm, loss_mel, loss_kl, loss_lf0, loss_gen_all)

    # GradScaler (train.py:143,192-213): a step whose gradients overflow is skipped and the scale halved; the golden is the first
    # step that goes through
    scale = 1.0 if half_type == "bf16" else 65536.0
    out = attempt(scale)
    while out is None:
        scale *= 0.5
        print("overflow: scale ->", scale)
        out = attempt(scale)
    y_hat, z_p, fr, y_hat_mel, gd, gg, loss_disc, loss_gen, loss_fm, loss_mel, loss_kl, loss_lf0, loss_gen_all = out
    ref = dict(loss_disc=loss_disc, loss_gen=loss_gen, loss_fm=loss_fm, loss_mel=loss_mel, loss_kl=loss_kl, loss_lf0=loss_lf0,
               loss_gen_all=loss_gen_all)
    print("dtypes: y_hat", y_hat.dtype, "z_p", z_p.dtype, "fmap", fr[0][0].dtype, "y_hat_mel", y_hat_mel.dtype)
    # distance from the fp32 step of the same reference (train_small.npz): the reference's own bf16 noise
    z32 = np.load(os.path.join(HERE, "train_small.npz"), allow_pickle=False)
    noise_l = {k: abs(float(v) - float(z32["loss." + k])) / max(1.0, abs(float(z32["loss." + k]))) for k, v in ref.items()}
    yh = y_hat.detach().float().numpy()
    noise_y = float(np.abs(yh - z32["y_hat"]).max() / max(1.0, np.abs(z32["y_hat"]).max()))
    n32_g = dict(zip([str(k) for k in z32["gnorm_g_keys"]], z32["gnorm_g"]))
    n32_d = dict(zip([str(k) for k in z32["gnorm_d_keys"]], z32["gnorm_d"]))
    rel_g = sorted((abs(gg[k].norm().item() - n32_g[k]) / max(n32_g[k], 1e-5), k) for k in gg if not k.endswith("conv_k.bias"))
    rel_d = sorted((abs(gd[k].norm().item() - n32_d[k]) / max(n32_d[k], 1e-6), k) for k in gd)
    print("losses amp:", {k: round(float(v), 5) for k, v in ref.items()})
    print("relative distance from the fp32 step: losses", {k: f"{v:.2e}" for k, v in noise_l.items()}, "y_hat", f"{noise_y:.2e}")
    print("grad norms G: median %.2e  p90 %.2e  max %.2e (%s)" % (rel_g[len(rel_g) // 2][0], rel_g[int(0.9 * len(rel_g))][0], rel_g[-1][0], rel_g[-1][1]))
    print("grad norms D: median %.2e  p90 %.2e  max %.2e (%s)" % (rel_d[len(rel_d) // 2][0], rel_d[int(0.9 * len(rel_d))][0], rel_d[-1][0], rel_d[-1][1]))
    np.savez_compressed(
        os.path.join(HERE, f"train_amp_{half_type}_small.npz"), y_hat=yh,
        **{f"loss.{k}": np.float64(float(v)) for k, v in ref.items()},
        **{f"amp_vs_fp32.{k}": np.float64(v) for k, v in noise_l.items()}, amp_vs_fp32_y_hat=np.float64(noise_y),
        amp_vs_fp32_gnorm_g=np.array([rel_g[len(rel_g) // 2][0], rel_g[int(0.9 * len(rel_g))][0], rel_g[-1][0]]),
        amp_vs_fp32_gnorm_d=np.array([rel_d[len(rel_d) // 2][0], rel_d[int(0.9 * len(rel_d))][0], rel_d[-1][0]]),
        gnorm_g_keys=np.array(list(gg.keys())), gnorm_g=np.array([gg[k].norm().item() for k in gg], dtype=np.float64),
        gnorm_d_keys=np.array(list(gd.keys())), gnorm_d=np.array([gd[k].norm().item() for k in gd], dtype=np.float64),
        meta=json.dumps(dict(B=B, T=T, seed=seed, data=DATA, upsample_rates=cfg["upsample_rates"],
                             upsample_kernel_sizes=cfg["upsample_kernel_sizes"], c_mel=45.0, c_kl=1.0, half_type=half_type,
                             loss_scale=scale, torch=torch.__version__)))
    print(f"wrote train_amp_{half_type}_small.npz")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "bf16")
