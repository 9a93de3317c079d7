"""CPU oracle for the TRAINING graph.  TEST INFRASTRUCTURE ONLY (see svc_oracle.py).

torch-CPU fp32 restatement of SynthesizerTrn.forward (models.py:463-493), MultiPeriodDiscriminator (:165-252),
the mel pipeline (modules/mel_processing.py:40-83), the losses (modules/losses.py) and the loss assembly of one GAN
step (train.py:167-207).  Gradients come from torch's CPU autograd over these same ops, i.e. exactly what the reference
computes.  Pinned by tests/golden/train_small.npz, generated from the REAL reference by tests/golden/make_golden_train.py.
"""
import math

import torch
import torch.nn.functional as F

from . import svc_oracle as O

LRELU_SLOPE = 0.1


def synth_forward(sd, cfg, c, f0, uv, spec, sid, c_lengths, spec_lengths, noise, vol=None):
    """models.py:463-493 with the random draws explicit: noise = dict(f0_factor [B,1], enc_p, enc_q [B,inter,T],
    ids_slice [B] int64, rand_ini [B,9], sine [B, seg*hop, 9], optional dropout_u = the uniform draws of every nn.Dropout
    site with cfg["p_dropout"] > 0, in call order: f0_decoder's layers first (models.py:476), then enc_p's (:477), then — with use_transformer_flow — the
    flow's FFT layers (:482))."""
    B, _, T = c.shape
    drop = O.DropSeq(cfg.get("p_dropout", 0.0), noise.get("dropout_u"))
    g = sd["emb_g.weight"][sid].transpose(1, 2)
    x_mask = O.sequence_mask(c_lengths, T).unsqueeze(1).to(c.dtype)
    x = O.conv1d(c, sd, "pre", padding=2) * x_mask + sd["emb_uv.weight"][uv.long()].transpose(1, 2)
    if vol is not None and cfg.get("vol_embedding", False):      # models.py:469
        x = x + F.linear(vol[:, :, None], sd["emb_vol.weight"], sd["emb_vol.bias"]).transpose(1, 2)
    lf0 = 2595. * torch.log10(1. + f0.unsqueeze(1) / 700.) / 500
    norm_lf0 = O.normalize_f0(lf0, x_mask, uv, noise["f0_factor"])
    pred_lf0 = O.f0_decoder(x.detach(), norm_lf0, x_mask, g, sd, cfg, drop=drop)
    z_ptemp, m_p, logs_p = O.text_encoder(x, x_mask, O.f0_to_coarse(f0), sd, cfg, noise["enc_p"], 1.0, drop=drop)
    spec_mask = O.sequence_mask(spec_lengths, spec.shape[2]).unsqueeze(1).to(spec.dtype)
    z, m_q, logs_q = O.posterior_encoder(spec, spec_mask, g, sd, cfg, noise["enc_q"])
    z_p = O.flow(z, spec_mask, g, sd, cfg, reverse=False, drop=drop)      # transformer flow: its dropout sites come last
    seg = cfg["segment_size"]
    ids = noise["ids_slice"]
    z_slice = torch.stack([z[i, :, ids[i]:ids[i] + seg] for i in range(B)])
    pitch_slice = torch.stack([f0[i, ids[i]:ids[i] + seg] for i in range(B)])
    o = O.generator(z_slice, pitch_slice, g, sd, cfg, noise["rand_ini"], noise["sine"])
    return o, ids, spec_mask, (z, z_p, m_p, logs_p, m_q, logs_q), pred_lf0, norm_lf0, lf0


# ------------------------------------------------------------------------------------------------------------
# discriminators (models.py:165-252); state_dict keys discriminators.N.convs.M.{bias,weight_g,weight_v}
# ------------------------------------------------------------------------------------------------------------
def _wn(sd, prefix):
    v, g = sd[prefix + ".weight_v"], sd[prefix + ".weight_g"]
    norm = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return v * (g / norm)


def _eff_weight(sd, prefix, training=True):
    """The effective conv weight of a normalised layer: weight_norm (models.py default) or, when the state dict holds
    `weight_orig`, torch.nn.utils.spectral_norm's compute_weight — one power iteration per call in training mode that UPDATES
    sd[prefix.weight_u / weight_v] (the module's buffers are updated in place), sigma = u . (W v), w = W / sigma with the
    gradient flowing through sigma at constant u, v."""
    if prefix + ".weight_orig" not in sd:
        return _wn(sd, prefix)
    W, u, v = sd[prefix + ".weight_orig"], sd[prefix + ".weight_u"], sd[prefix + ".weight_v"]
    Wm = W.reshape(W.shape[0], -1)
    if training:
        with torch.no_grad():
            v = F.normalize(torch.mv(Wm.t(), u), dim=0, eps=1e-12)
            u = F.normalize(torch.mv(Wm, v), dim=0, eps=1e-12)
        sd[prefix + ".weight_u"], sd[prefix + ".weight_v"] = u, v
    sigma = torch.dot(u, torch.mv(Wm, v))
    return W / sigma


def disc_p(x, sd, prefix, period):
    fmap = []
    b, c, t = x.shape
    if t % period != 0:
        n_pad = period - (t % period)
        x = F.pad(x, (0, n_pad), "reflect")
        t = t + n_pad
    x = x.view(b, c, t // period, period)
    for i, s in enumerate([3, 3, 3, 3, 1]):
        x = F.conv2d(x, _eff_weight(sd, f"{prefix}.convs.{i}"), sd[f"{prefix}.convs.{i}.bias"], (s, 1), (2, 0))
        x = F.leaky_relu(x, LRELU_SLOPE)
        fmap.append(x)
    x = F.conv2d(x, _eff_weight(sd, prefix + ".conv_post"), sd[prefix + ".conv_post.bias"], 1, (1, 0))
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def disc_s(x, sd, prefix):
    fmap = []
    cfgs = [(1, 7, 1), (4, 20, 4), (4, 20, 16), (4, 20, 64), (4, 20, 256), (1, 2, 1)]
    for i, (s, p, g) in enumerate(cfgs):
        x = F.conv1d(x, _eff_weight(sd, f"{prefix}.convs.{i}"), sd[f"{prefix}.convs.{i}.bias"], s, p, 1, g)
        x = F.leaky_relu(x, LRELU_SLOPE)
        fmap.append(x)
    x = F.conv1d(x, _eff_weight(sd, prefix + ".conv_post"), sd[prefix + ".conv_post.bias"], 1, 1)
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def mpd(sd, y, y_hat):
    rs, gs, frs, fgs = [], [], [], []
    for i, p in enumerate([None, 2, 3, 5, 7, 11]):
        f = (lambda x: disc_s(x, sd, f"discriminators.{i}")) if p is None else (lambda x: disc_p(x, sd, f"discriminators.{i}", p))
        r, fr = f(y)
        g, fg = f(y_hat)
        rs.append(r); gs.append(g); frs.append(fr); fgs.append(fg)
    return rs, gs, frs, fgs


# ------------------------------------------------------------------------------------------------------------
# mel (modules/mel_processing.py:40-83); mel_basis [n_mels, bins] is passed in (librosa is not importable)
# ------------------------------------------------------------------------------------------------------------
def spectrogram(y, n_fft, hop, win):
    y = F.pad(y.unsqueeze(1), (int((n_fft - hop) / 2), int((n_fft - hop) / 2)), mode="reflect").squeeze(1)
    spec = torch.stft(y, n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win), center=False,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    spec = torch.view_as_real(spec)
    return torch.sqrt(spec.pow(2).sum(-1) + 1e-6)


def spec_to_mel(spec, mel_basis):
    return torch.log(torch.clamp(torch.matmul(mel_basis, spec), min=1e-5))


# ------------------------------------------------------------------------------------------------------------
# losses (modules/losses.py) and one GAN step's loss assembly (train.py:167-207)
# ------------------------------------------------------------------------------------------------------------
def feature_loss(fr, fg):
    loss = 0
    for dr, dg in zip(fr, fg):
        for rl, gl in zip(dr, dg):
            loss = loss + torch.mean(torch.abs(rl.float().detach() - gl.float()))
    return loss * 2


def discriminator_loss(rs, gs):
    loss = 0
    for dr, dg in zip(rs, gs):
        loss = loss + torch.mean((1 - dr.float()) ** 2) + torch.mean(dg.float() ** 2)
    return loss


def generator_loss(gs):
    loss = 0
    for dg in gs:
        loss = loss + torch.mean((1 - dg.float()) ** 2)
    return loss


def kl_loss(z_p, logs_q, m_p, logs_p, z_mask):
    kl = logs_p - logs_q - 0.5
    kl = kl + 0.5 * ((z_p - m_p) ** 2) * torch.exp(-2. * logs_p)
    return torch.sum(kl * z_mask) / torch.sum(z_mask)


def slice_segments(x, ids, size):
    return torch.stack([x[i, :, ids[i]:ids[i] + size] for i in range(x.shape[0])])


def gan_step_losses(sd_g, sd_d, cfg, data, batch, noise, mel_basis, c_mel=45.0, c_kl=1.0):
    """train.py:167-207 for one batch: returns dict of the scalar losses (+ y_hat).  data = dict(n_fft, hop, win)."""
    c, f0, uv, spec, y, sid, lengths = batch
    y_hat, ids, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), pred_lf0, norm_lf0, lf0 = synth_forward(
        sd_g, cfg, c, f0, uv, spec, sid, lengths, lengths, noise)
    seg, hop = cfg["segment_size"], data["hop"]
    mel = spec_to_mel(spec, mel_basis)
    y_mel = slice_segments(mel, ids, seg)
    y_hat_mel = spec_to_mel(spectrogram(y_hat.squeeze(1), data["n_fft"], hop, data["win"]), mel_basis)
    y_seg = slice_segments(y, ids * hop, seg * hop)
    rs, gs, _, _ = mpd(sd_d, y_seg, y_hat.detach())
    loss_disc = discriminator_loss(rs, gs)
    rs, gs, frs, fgs = mpd(sd_d, y_seg, y_hat)
    loss_mel = F.l1_loss(y_mel, y_hat_mel) * c_mel
    loss_kl = kl_loss(z_p, logs_q, m_p, logs_p, z_mask) * c_kl
    loss_fm = feature_loss(frs, fgs)
    loss_gen = generator_loss(gs)
    loss_lf0 = F.mse_loss(pred_lf0, lf0)
    loss_gen_all = loss_gen + loss_fm + loss_mel + loss_kl + loss_lf0
    return dict(loss_disc=loss_disc, loss_gen=loss_gen, loss_fm=loss_fm, loss_mel=loss_mel, loss_kl=loss_kl,
                loss_lf0=loss_lf0, loss_gen_all=loss_gen_all, y_hat=y_hat)


def gan_train_loop(sd_g, sd_d, cfg, data, batch, noise, mel_basis, n_iter, lr=1e-4, betas=(0.8, 0.99), eps=1e-9,
                   c_mel=45.0, c_kl=1.0):
    """n_iter iterations of train.py:150-213 in the reference's ORDER (D loss -> backward -> optim_d.step -> D forward
    again with the UPDATED discriminator -> G losses -> backward -> optim_g.step), torch.optim.AdamW as train.py:79-88,
    the same batch and injected noise every iteration.  Returns (list of per-iteration loss dicts, sd_g, sd_d)."""
    sg = {k: v.clone().requires_grad_(True) for k, v in sd_g.items()}
    sdd = {k: v.clone().requires_grad_(True) for k, v in sd_d.items()}
    og = torch.optim.AdamW(list(sg.values()), lr, betas=betas, eps=eps)
    od = torch.optim.AdamW(list(sdd.values()), lr, betas=betas, eps=eps)
    c, f0, uv, spec, y, sid, lengths = batch
    seg, hop = cfg["segment_size"], data["hop"]
    hist = []
    for _ in range(n_iter):
        y_hat, ids, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), pred_lf0, norm_lf0, lf0 = synth_forward(
            sg, cfg, c, f0, uv, spec, sid, lengths, lengths, noise)
        mel = spec_to_mel(spec, mel_basis)
        y_mel = slice_segments(mel, ids, seg)
        y_hat_mel = spec_to_mel(spectrogram(y_hat.squeeze(1), data["n_fft"], hop, data["win"]), mel_basis)
        y_seg = slice_segments(y, ids * hop, seg * hop)
        rs, gs, _, _ = mpd(sdd, y_seg, y_hat.detach())
        loss_disc = discriminator_loss(rs, gs)
        od.zero_grad()
        loss_disc.backward()
        od.step()
        rs, gs, frs, fgs = mpd(sdd, y_seg, y_hat)
        loss_mel = F.l1_loss(y_mel, y_hat_mel) * c_mel
        loss_kl = kl_loss(z_p, logs_q, m_p, logs_p, z_mask) * c_kl
        loss_fm = feature_loss(frs, fgs)
        loss_gen = generator_loss(gs)
        loss_lf0 = F.mse_loss(pred_lf0, lf0)
        loss_gen_all = loss_gen + loss_fm + loss_mel + loss_kl + loss_lf0
        og.zero_grad()
        od.zero_grad()
        loss_gen_all.backward()
        og.step()
        hist.append({k: float(v) for k, v in dict(loss_disc=loss_disc, loss_gen=loss_gen, loss_fm=loss_fm,
                                                   loss_mel=loss_mel, loss_kl=loss_kl, loss_lf0=loss_lf0,
                                                   loss_gen_all=loss_gen_all).items()})
    return hist, {k: v.detach() for k, v in sg.items()}, {k: v.detach() for k, v in sdd.items()}
