"""Micro-benchmark of svc_conv1d_f32 on the decoder's MRF stage shapes (T=862 -> 10 s clip)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import svc_hip as S

dev = torch.device("cuda:0")
T0 = 862
shapes = []
L = T0
for i, (u, C) in enumerate(zip([8, 8, 2, 2, 2], [256, 128, 64, 32, 16])):
    L *= u
    for k in (3, 7, 11):
        for d in (1, 5):
            shapes.append((C, L, k, d))
extra = [(192, 862, 5, 1, 384), (192, 862, 3, 1, 768), (768, 862, 3, 1, 192), (192, 862, 1, 1, 576)]

def run(Cin, L, k, d, Cout=None, iters=10):
    Cout = Cout or Cin
    x = torch.randn(1, Cin, L, device=dev)
    w = torch.randn(Cout, Cin, k, device=dev) / (Cin * k) ** 0.5
    b = torch.randn(Cout, device=dev)
    wp = S.pack_conv1d_weight(w)
    out = torch.empty(1, Cout, L, device=dev)
    pad = (k * d - d) // 2
    for _ in range(2):
        S.conv1d(x, wp, Cout, k, bias=b, dil=d, pad_left=pad, pre_slope=0.1, res=x if Cout == Cin else None,
                 res_mode=1 if Cout == Cin else 0, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        S.conv1d(x, wp, Cout, k, bias=b, dil=d, pad_left=pad, pre_slope=0.1, res=x if Cout == Cin else None,
                 res_mode=1 if Cout == Cin else 0, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * Cout * Cin * k * L
    print(f"Cin={Cin:4d} Cout={Cout:4d} L={L:7d} k={k:2d} d={d}  {ms*1e3:9.1f} us  {fl/ms/1e9:7.1f} TFLOP/s  "
          f"{(Cin+2*Cout)*L*4/ms/1e6:7.1f} GB/s")
    return ms, fl

tot_ms = tot_fl = 0
for (C, L, k, d) in shapes:
    ms, fl = run(C, L, k, d)
    tot_ms += ms; tot_fl += fl
print(f"MRF-shape mean: {tot_fl/tot_ms/1e9:.1f} TFLOP/s")
for (Cin, L, k, d, Cout) in extra:
    run(Cin, L, k, d, Cout)
