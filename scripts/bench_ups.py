"""Micro-benchmark of svc_conv_transpose1d_f32 on the decoder's five upsampling stages of a 10 s clip (T = 862 frames,
rates 8, 8, 2, 2, 2, kernels 16, 16, 4, 4, 4; lrelu pre-activation + noise-conv residual), N launches per hipGraph replay."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import svc_hip as S

dev = torch.device("cuda:0")
if len(sys.argv) > 1:
    S.tlib().svc_debug_set_conv_cfg(int(sys.argv[1]))
N = 10
T = 862
tot = 0.0
for (Cin, u, K) in [(512, 8, 16), (256, 8, 16), (128, 2, 4), (64, 2, 4), (32, 2, 4)]:
    Cout = Cin // 2
    pad = (K - u) // 2
    x = torch.randn(1, Cin, T, device=dev)
    v = torch.randn(Cin, Cout, K, device=dev) * 0.05
    g = torch.rand(Cin, 1, 1, device=dev) + 0.5
    b = torch.randn(Cout, device=dev)
    wp = S.pack_convt1d_weight(v, g, u)
    Tout = (T - 1) * u - 2 * pad + K
    res = torch.randn(1, Cout, Tout, device=dev)
    out = torch.empty(1, Cout, Tout, device=dev)
    run = lambda: S.conv_transpose1d(x, wp, Cout, K, u, pad, bias=b, pre_slope=0.1, res=res, out=out)
    run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(N):
            run()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay(); gr.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (2 * N)
    fl = 2.0 * Cin * Cout * K * T
    by = 4.0 * (Cin * T + 2 * Cout * Tout)
    print(f"ups {Cin:3d}->{Cout:3d} x{u} k={K:2d} T={T:6d}->{Tout:6d}  {ms*1e3:7.1f} us  {fl/ms/1e9:6.1f} TFLOP/s  {by/ms/1e6:7.1f} GB/s", flush=True)
    tot += ms
    T = Tout
print(f"sum {tot*1e3:.0f} us")
