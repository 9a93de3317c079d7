"""MI355X-native mirror of modules/losses.py: the four GAN/VITS loss terms train.py imports by name.  Sums run as HIP
reductions (svc_reduce_scalar_f64 / svc_kl_fwd_f64, backward through svc_autograd); the 0-dim results are combined with
plain scalar arithmetic, as train.py itself does with the returned losses."""
import torch

import svc_autograd as A


def feature_loss(fmap_r, fmap_g):
    """Reference modules/losses.py:4-12: 2 * sum over discriminators/layers of mean|r - g| (r detached)."""
    loss = 0
    for dr, dg in zip(fmap_r, fmap_g):
        for rl, gl in zip(dr, dg):
            n = gl.numel()
            rp, gp = getattr(rl, "_svc_padded", None), getattr(gl, "_svc_padded", None)
            if rp is not None and gp is not None and rp.shape == gp.shape:
                rl, gl = rp, gp         # DiscriminatorP's padded buffers: zero tails on both sides, same sum, no gather copy
            loss = loss + A.sum_abs_diff(rl.float().detach(), gl.float()) / n
    return loss * 2


def discriminator_loss(disc_real_outputs, disc_generated_outputs):
    """Reference :15-28.  r_losses / g_losses are returned as device scalars (the reference calls .item() on each of
    the 12 terms — a host sync per term)."""
    loss = 0
    r_losses, g_losses = [], []
    for dr, dg in zip(disc_real_outputs, disc_generated_outputs):
        r_loss = A.sum_sq_one_minus(dr.float()) / dr.numel()
        g_loss = A.sum_sq(dg.float()) / dg.numel()
        loss = loss + (r_loss + g_loss)
        r_losses.append(r_loss.detach())
        g_losses.append(g_loss.detach())
    return loss, r_losses, g_losses


def generator_loss(disc_outputs):
    """Reference :31-40."""
    loss = 0
    gen_losses = []
    for dg in disc_outputs:
        l = A.sum_sq_one_minus(dg.float()) / dg.numel()
        gen_losses.append(l)
        loss = loss + l
    return loss, gen_losses


def kl_loss(z_p, logs_q, m_p, logs_p, z_mask):
    """Reference :43-58: sum(kl * mask) / sum(mask), kl = logs_p - logs_q - 0.5 + 0.5 (z_p - m_p)^2 exp(-2 logs_p)."""
    s = A.kl_sums(z_p.float(), logs_q.float(), m_p.float(), logs_p.float(), z_mask.float())
    return s[0] / s[1]
