"""MI355X-native mirror of modules/losses.py: the four GAN/VITS loss terms train.py imports by name.  Sums run as HIP
reductions (svc_reduce_scalar_f64 / svc_kl_fwd_f64, backward through svc_autograd); the 0-dim results are combined with
plain scalar arithmetic, as train.py itself does with the returned losses."""
import torch

import svc_autograd as A
import svc_hip as S


def feature_loss(fmap_r, fmap_g):
    """Reference modules/losses.py:4-12: 2 * sum over discriminators/layers of mean|r - g| (r detached) — all 41 terms
    accumulate into one device scalar (svc_autograd.weighted_sums)."""
    terms = []
    for dr, dg in zip(fmap_r, fmap_g):
        for rl, gl in zip(dr, dg):
            n = gl.numel()
            rp, gp = getattr(rl, "_svc_padded", None), getattr(gl, "_svc_padded", None)
            if rp is not None and gp is not None and rp.shape == gp.shape:
                rl, gl = rp, gp         # DiscriminatorP's padded buffers: zero tails on both sides, same sum, no gather copy
            terms.append((S.RED_ABS_DIFF, 2.0 / n, rl.float().detach(), gl.float()))
    return A.weighted_sums(terms)


def discriminator_loss(disc_real_outputs, disc_generated_outputs):
    """Reference :15-28.  r_losses / g_losses are returned as device scalars (the reference calls .item() on each of
    the 12 terms — a host sync per term)."""
    terms = []
    for dr, dg in zip(disc_real_outputs, disc_generated_outputs):
        terms.append((S.RED_SQ_ONE_MINUS, 1.0 / dr.numel(), dr.float()))
        terms.append((S.RED_SQ, 1.0 / dg.numel(), dg.float()))
    loss, vals = A.weighted_sums(terms, per_term=True)
    return loss, list(vals[0::2].unbind(0)), list(vals[1::2].unbind(0))


def generator_loss(disc_outputs):
    """Reference :31-40."""
    loss, vals = A.weighted_sums([(S.RED_SQ_ONE_MINUS, 1.0 / dg.numel(), dg.float()) for dg in disc_outputs], per_term=True)
    return loss, list(vals.unbind(0))


def kl_loss(z_p, logs_q, m_p, logs_p, z_mask):
    """Reference :43-58: sum(kl * mask) / sum(mask), kl = logs_p - logs_q - 0.5 + 0.5 (z_p - m_p)^2 exp(-2 logs_p)."""
    s = A.kl_sums(z_p.float(), logs_q.float(), m_p.float(), logs_p.float(), z_mask.float())
    return s[0] / s[1]
