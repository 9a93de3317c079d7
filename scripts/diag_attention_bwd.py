"""Diagnostic: svc_autograd.attention forward/backward vs a torch restatement over a grid of shapes / options."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "so-vits-svc_amd")]
import svc_autograd as A  # noqa: E402

dev = torch.device("cuda:0")


def run(B, H, dk, T, w, p, mode, seed=3):
    g = torch.Generator().manual_seed(seed)
    q, k, v = [torch.randn(B, H * dk, T, generator=g).to(dev).requires_grad_(True) for _ in range(3)]
    ek = ev = None
    if w:
        ek, ev = [(torch.randn(1, 2 * w + 1, dk, generator=g) * 0.2).to(dev).requires_grad_(True) for _ in range(2)]
    u = torch.rand(B, H, T, T, generator=g).to(dev) if p > 0 else None
    lens = torch.tensor([T - 9 * b for b in range(B)]).clamp(min=1)
    mask = (torch.arange(T)[None] < lens[:, None]).float().to(dev) if mode == 1 else None
    out = A.attention(q, k, v, H, ek, ev, w, mask, mode, drop_u=u, p_drop=p)
    go = torch.randn(B, H * dk, T, generator=g).to(dev)
    out.backward(go)
    leaves = [q, k, v] + ([ek, ev] if w else [])
    got = [t.grad.clone() for t in leaves]

    def ref(q, k, v, ek, ev):
        qh = q.view(B, H, dk, T).transpose(2, 3) / math.sqrt(dk)
        kh = k.view(B, H, dk, T).transpose(2, 3)
        vh = v.view(B, H, dk, T).transpose(2, 3)
        sc = qh @ kh.transpose(-2, -1)
        idx = torch.arange(T, device=dev)
        band = idx[None, :] - idx[:, None]
        inb = band.abs() <= w
        ii, jj, rr = idx[:, None].expand(T, T)[inb], idx[None, :].expand(T, T)[inb], (band + w)[inb]
        if w:
            rel = qh @ ek[0].t()
            relfull = torch.zeros_like(sc)
            relfull[:, :, ii, jj] = rel[:, :, ii, rr]
            sc = sc + relfull
        if mode == 1:
            am = mask[:, None, :, None] * mask[:, None, None, :]
            sc = sc.masked_fill(am == 0, -1e4)
        elif mode == 2:
            sc = sc.masked_fill(torch.tril(torch.ones(T, T, device=dev)) == 0, -1e4)
        pa = torch.softmax(sc, -1)
        if p > 0:
            pa = pa * ((u >= p).float() * (1.0 / (1.0 - p)))
        o = pa @ vh
        if w:
            pb = torch.zeros(B, H, T, 2 * w + 1, device=dev)
            pb[:, :, ii, rr] = pa[:, :, ii, jj]
            o = o + pb @ ev[0]
        return o.transpose(2, 3).contiguous().view(B, H * dk, T)

    rl = [t.detach().clone().requires_grad_(True) for t in leaves]
    ro = ref(*(rl + [None, None])[:5])
    ro.backward(go)
    errs = [((a - b.grad).abs().max().item() / max(1e-6, b.grad.abs().max().item())) for a, b in zip(got, rl)]
    fe = (out - ro).abs().max().item() / max(1e-6, ro.abs().max().item())
    print(f"B{B} H{H} dk{dk} T{T} w{w} p{p} mode{mode}: fwd {fe:.1e}  grads " + " ".join(f"{e:.1e}" for e in errs), flush=True)


for cfg in [(2, 2, 24, 37, 4, 0.25, 1), (2, 2, 24, 37, 4, 0.0, 1), (2, 2, 24, 37, 0, 0.0, 1), (2, 2, 24, 37, 4, 0.0, 0),
            (2, 2, 24, 37, 0, 0.0, 0), (2, 2, 32, 40, 4, 0.0, 1), (2, 2, 32, 40, 4, 0.1, 1), (2, 2, 32, 37, 4, 0.0, 0),
            (2, 2, 24, 40, 4, 0.0, 0), (1, 2, 96, 128, 4, 0.1, 1), (2, 2, 96, 100, 0, 0.1, 2), (1, 1, 24, 37, 4, 0.0, 0)]:
    run(*cfg)
