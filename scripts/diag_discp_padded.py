"""Diagnostic (GPU): DiscriminatorP input / weight gradients — padded-row layout vs the unpadded one vs the torch-CPU restatement.
Separates a tail-handling bug (padded differs from unpadded) from fp32 noise through the non-smooth points of the graph
(|r - g| of the feature loss, leaky_relu at 0: sign flips between implementations, equally present in both layouts)."""
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
from oracle import train_oracle as TO  # noqa: E402

dev = torch.device("cuda:0")
T, B = 8192, 2
probe = ["convs.0.weight_v", "convs.1.weight_v", "convs.3.weight_g", "convs.4.weight_v", "convs.4.bias", "conv_post.weight_v"]
for period in (3, 11, 5):
    torch.manual_seed(period)
    sd_all = W.make_mpd_state_dict(77)
    idx = [None, 2, 3, 5, 7, 11].index(period)
    prefix = f"discriminators.{idx}"
    sd = {k[len(prefix) + 1:]: v for k, v in sd_all.items() if k.startswith(prefix + ".")}
    y, y_hat = torch.randn(B, 1, T) * 0.5, torch.randn(B, 1, T) * 0.5
    res = {}
    for dtype in (torch.float32, torch.float64):
        sr = {prefix + "." + k: v.clone().to(dtype).requires_grad_(k in probe) for k, v in sd.items()}
        yh = y_hat.clone().to(dtype).requires_grad_(True)
        lr_, fr = TO.disc_p(y.to(dtype), sr, prefix, period)
        lg_, fg = TO.disc_p(yh, sr, prefix, period)
        loss = sum((a.detach() - b).abs().mean() for a, b in zip(fr, fg)) * 2 + ((1 - lg_) ** 2).mean() + (lr_ ** 2).mean()
        loss.backward()
        res["ref32" if dtype == torch.float32 else "ref64"] = dict(x=yh.grad.float(), **{k: sr[prefix + "." + k].grad.float() for k in probe})
    for pad in (True, False):
        models._DISCP_PAD_ROWS = pad
        net = models.DiscriminatorP(period)
        net.load_state_dict(sd, strict=True)
        net = net.to(dev).train()
        yh = y_hat.clone().to(dev).requires_grad_(True)
        out, fmap = net(torch.cat([y.to(dev), yh], 0))
        halves = [models._split_map(f, B) for f in fmap]
        loss = feature_loss([[a for a, _ in halves]], [[b for _, b in halves]]) + A.sum_sq_one_minus(out[B:]) / out[B:].numel() \
            + A.sum_sq(out[:B]) / out[:B].numel()
        loss.backward()
        named = dict(net.named_parameters())
        res["pad" if pad else "nopad"] = dict(x=yh.grad.cpu(), **{k: named[k].grad.cpu() for k in probe})

    def cmp(a, b, k):
        d = (res[a][k] - res[b][k]).abs()
        m = res[b][k].abs().max().item()
        flat = d.flatten()
        i = int(flat.argmax())
        return f"{d.max().item() / max(m, 1e-12):.2e}@{i}/{flat.numel()} n>1e-3:{int((flat > 1e-3 * m).sum())}"
    print(f"== period {period}")
    for k in ["x"] + probe:
        print(f"  {k:20s} pad-ref64 {cmp('pad', 'ref64', k)} | nopad-ref64 {cmp('nopad', 'ref64', k)} | pad-nopad {cmp('pad', 'nopad', k)}"
              f" | ref32-ref64 {cmp('ref32', 'ref64', k)}")
