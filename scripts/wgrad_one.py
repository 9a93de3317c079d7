"""Run one conv1d_wgrad shape a few times (for rocprofv3 --pmc / --kernel-trace).  usage: wgrad_one.py B Ca Cb T K dil [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import svc_hip as S
B, Ca, Cb, T, K, d = [int(a) for a in sys.argv[1:7]]
n = int(sys.argv[7]) if len(sys.argv) > 7 else 6
dev = torch.device("cuda:0")
if os.environ.get("WGRAD_TARGET"):
    S.tlib().svc_debug_set_wgrad_target(int(os.environ["WGRAD_TARGET"]))
MMA = S.MMA_BF16 if os.environ.get("WGRAD_BF16") == "1" else S.MMA_F32
dy = torch.randn(B, Ca, T, device=dev)
x = torch.randn(B, Cb, T, device=dev)
out = torch.zeros(Ca, Cb, K, device=dev)
pad = d * (K - 1) // 2
for _ in range(n):
    S.conv1d_wgrad(dy, x, K, d, pad, out=out, accumulate=True, mma=MMA)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    S.conv1d_wgrad(dy, x, K, d, pad, out=out, accumulate=True, mma=MMA)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
print(f"wgrad B{B} Ca{Ca} Cb{Cb} T{T} K{K} d{d}: {us:.1f} us  {2.0 * B * Ca * Cb * K * T / us * 1e-6:.1f} TFLOP/s")
