"""Diagnostic: bf16-operand conv against fp32 conv of bf16-rounded operands, per shape (launch counter + error split)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import torch.nn.functional as F
import svc_hip as S
dev = torch.device("cuda:0")
r = lambda t: t.bfloat16().float()
CASES = [(16, 192, 192, 768, 1, 1), (16, 384, 192, 768, 5, 1), (16, 192, 384, 768, 5, 2), (32, 1024, 1024, 132, 5, 11),
         (1, 256, 256, 6896, 7, 3), (2, 128, 128, 20000, 3, 1), (1, 64, 64, 20000, 7, 5), (4, 96, 192, 700, 3, 1),
         (16, 192, 192, 768, 3, 1), (16, 192, 192, 768, 2, 1), (2, 128, 128, 20000, 5, 1), (2, 128, 128, 20000, 7, 1)]
for B, Cin, Cout, T, K, dil in CASES:
    g = torch.Generator().manual_seed(B + Cin + Cout + T + K)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5
    pad = (K * dil - dil) // 2
    xd, wp = x.to(dev), S.pack_conv1d_weight(w.to(dev))
    n0 = S.lib().svc_debug_bf16(-1)
    y = S.conv1d(xd, wp, Cout, K, dil=dil, pad_left=pad, mma=S.MMA_BF16).cpu()
    n1 = S.lib().svc_debug_bf16(-1)
    y32 = S.conv1d(xd, wp, Cout, K, dil=dil, pad_left=pad).cpu()
    ref_r = F.conv1d(r(x), r(w), dilation=dil, padding=pad)
    ref_xr = F.conv1d(r(x), w, dilation=dil, padding=pad)
    ref_wr = F.conv1d(x, r(w), dilation=dil, padding=pad)
    ref = F.conv1d(x, w, dilation=dil, padding=pad)
    sc = ref.abs().max().item()
    e = lambda a, b: (a - b).abs().max().item() / sc
    print(f"B{B} {Cin}->{Cout} T{T} K{K} d{dil}: bf16 launches {n1 - n0}  |y-ref(rx,rw)| {e(y, ref_r):.2e}  |y-ref(rx,w)| {e(y, ref_xr):.2e}  "
          f"|y-ref(x,rw)| {e(y, ref_wr):.2e}  |y-ref| {e(y, ref):.2e}  |y32-ref| {e(y32, ref):.2e}")
    # per-tap check: zero all taps but one
    if K > 1 and n1 > n0:
        for k in range(K):
            wk = torch.zeros_like(w); wk[:, :, k] = w[:, :, k]
            yk = S.conv1d(xd, S.pack_conv1d_weight(wk.to(dev)), Cout, K, dil=dil, pad_left=pad, mma=S.MMA_BF16).cpu()
            rk = F.conv1d(r(x), r(wk), dilation=dil, padding=pad)
            print(f"      tap {k}: |y-ref| {(yk - rk).abs().max().item() / max(rk.abs().max().item(), 1e-9):.2e}")
