"""Run one svc_conv1d_h / svc_conv1d_hl shape (or, with pairhalf / pairsplit, one fused ResBlock pair svc_resblock_pair_h / _hl) a few
times (for rocprofv3 --pmc).  usage: conv_hl_one.py split|half|pairsplit|pairhalf C L k d [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import svc_hip as S
mode = sys.argv[1]
C, L, k, d = [int(a) for a in sys.argv[2:6]]
n = int(sys.argv[6]) if len(sys.argv) > 6 else 6
dev = torch.device("cuda:0")
sp = mode in ("split", "pairsplit")
x = S.to_h(torch.randn(1, C, L, device=dev), split=sp)
w = S.pack_conv1d_h(torch.randn(C, C, k, device=dev) / (C * k) ** 0.5, split=sp)
b = torch.randn(C, device=dev)
y = torch.empty_like(x)
w2 = S.pack_conv1d_h(torch.randn(C, C, k, device=dev) / (C * k) ** 0.5, split=sp)
for _ in range(n):
    if mode.startswith("pair"):
        S.resblock_pair_h(x, w, b, w2, b, d, out=y)
    else:
        S.conv1d_h(x, w, C, bias=b, dil=d, pad_left=(k * d - d) // 2, pre_slope=0.1, post_slope=0.1, out=y)
torch.cuda.synchronize()
