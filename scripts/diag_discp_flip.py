"""Diagnostic (GPU): the [11-8192] case of tests/test_train_ops_gpu.py::test_discriminator_p_padded_rows — where do the padded and the
unpadded layout disagree?  Records every layer's pre-activation in both layouts (same data as the test), counts sign differences
(leaky_relu branch flips), and locates the input-gradient differences relative to the row ends."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
sys.path.insert(0, ROOT)
import models  # noqa: E402
import svc_autograd as A  # noqa: E402
import synthetic_data as W  # noqa: E402
from modules.losses import feature_loss  # noqa: E402

dev = torch.device("cuda:0")
period, T, B = 11, 8192, 2
torch.manual_seed(period)
sd_all = W.make_mpd_state_dict(77)
prefix = "discriminators.5"
sd = {k[len(prefix) + 1:]: v for k, v in sd_all.items() if k.startswith(prefix + ".")}
net = models.DiscriminatorP(period)
net.load_state_dict(sd, strict=True)
net = net.to(dev).train()
y, y_hat = torch.randn(B, 1, T) * 0.5, torch.randn(B, 1, T) * 0.5
rec = {}
orig_tail, orig_lrelu = A.leaky_relu_tail, A.leaky_relu


def tail(x, slope, valid):
    rec[cur].append((x.detach().clone(), valid))
    return orig_tail(x, slope, valid)


def lrelu(x, slope):
    rec[cur].append((x.detach().clone(), x.shape[2]))
    return orig_lrelu(x, slope)


A.leaky_relu_tail, A.leaky_relu = tail, lrelu
grads = {}
for padded in (True, False):
    cur = padded
    rec[cur] = []
    models._DISCP_PAD_ROWS = padded
    net.zero_grad(set_to_none=True)
    yh = y_hat.clone().to(dev).requires_grad_(True)
    out, fmap = net(torch.cat([y.to(dev), yh], 0))
    halves = [models._split_map(f, B) for f in fmap]
    loss = feature_loss([[a for a, _ in halves]], [[b for _, b in halves]]) + A.sum_sq_one_minus(out[B:]) / out[B:].numel() \
        + A.sum_sq(out[:B]) / out[:B].numel()
    loss.backward()
    grads[padded] = yh.grad.cpu()
for li, ((xp, vp), (xu, vu)) in enumerate(zip(rec[True], rec[False])):
    a, b = xp[:, :, :vu].cpu(), xu[:, :, :vu].cpu()
    d = (a - b).abs()
    flips = ((a > 0) != (b > 0))
    print(f"layer {li}: valid {vu} of pitch {xp.shape[2]}  max|pre_padded - pre_unpadded| {d.max().item():.3e} (max|pre| {b.abs().max().item():.3e})"
          f"  bit-equal {bool(torch.equal(a, b))}  sign flips {int(flips.sum())}  |pre| at flips {[f'{v:.2e}' for v in b[flips].abs().tolist()[:6]]}")
g, r = grads[True], grads[False]
d = (g - r).abs()
m = r.abs().max().item()
idx = (d > 1e-3 * m).nonzero()
print(f"input grad: max diff {d.max().item():.3e} = {d.max().item() / m:.3e} of max; entries > 1e-3 max: {idx.shape[0]}; their time positions "
      f"(T = {T}): min {int(idx[:, 2].min()) if idx.numel() else -1} max {int(idx[:, 2].max()) if idx.numel() else -1}; batch rows {sorted(set(idx[:, 0].tolist()))}")
