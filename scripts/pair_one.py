"""Run one svc_resblock_pair_f32 shape a few times (for rocprofv3 --pmc / --kernel-trace).  usage: pair_one.py C T K dil [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import svc_hip as S
C, T, K, d = [int(a) for a in sys.argv[1:5]]
n = int(sys.argv[5]) if len(sys.argv) > 5 else 6
dev = torch.device("cuda:0")
x = torch.randn(1, C, T, device=dev)
w1 = S.pack_conv1d_weight(torch.randn(C, C, K, device=dev) / (C * K) ** 0.5)
w2 = S.pack_conv1d_weight(torch.randn(C, C, K, device=dev) / (C * K) ** 0.5)
b1, b2 = torch.randn(C, device=dev), torch.randn(C, device=dev)
o = torch.empty_like(x)
for _ in range(n):
    S.resblock_pair(x, w1, b1, w2, b2, K, d, out=o)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    S.resblock_pair(x, w1, b1, w2, b2, K, d, out=o)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
print(f"pair C{C} T{T} K{K} d{d}: {us:.1f} us  {4.0 * C * C * K * T / us * 1e-6:.1f} TFLOP/s")
