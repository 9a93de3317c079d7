"""Micro-benchmark of svc_conv1d_f32 on the TRAINING graph's batched shapes (B = 16, T = 768: WN layers of the flows and
the posterior encoder, the prior encoder's FFN, DiscriminatorP's folded period convs), plain epilogue, N launches per
hipGraph replay.  usage: bench_train_conv.py [cfgcode ...]   (svc_debug_set_conv_cfg codes; 0 = the dispatcher's choice,
4 = 128x128, 5 = 64x128, 10 = 64x192)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import svc_hip as S

dev = torch.device("cuda:0")
SHAPES = [  # B, Cin, Cout, T, K, d
    (16, 768, 192, 768, 3, 1), (16, 192, 768, 768, 3, 1), (16, 384, 192, 768, 5, 1), (16, 192, 384, 768, 5, 1),
    (16, 192, 192, 768, 1, 1), (16, 384, 192, 768, 1, 1), (16, 192, 384, 768, 1, 1), (16, 192, 192, 768, 5, 1),
    (16, 256, 256, 128, 11, 1), (16, 128, 128, 1024, 11, 1), (32, 1024, 1024, 132, 5, 11), (32, 1024, 1024, 112, 5, 7),
    (32, 1536, 1024, 132, 2, 11), (16, 1024, 1024, 132, 5, 11), (1, 256, 256, 6896, 7, 3), (1, 256, 256, 6896, 11, 5),
]
N = 10


def run(B, Cin, Cout, T, k, d):
    x = torch.randn(B, Cin, T, device=dev)
    w = torch.randn(Cout, Cin, k, device=dev) / (Cin * k) ** 0.5
    b = torch.randn(Cout, device=dev)
    wp = S.pack_conv1d_weight(w)
    out = torch.empty(B, Cout, T, device=dev)
    kw = dict(bias=b, dil=d, pad_left=(k * d - d) // 2, out=out)
    S.conv1d(x, wp, Cout, k, **kw)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(N):
            S.conv1d(x, wp, Cout, k, **kw)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay(); g.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (2 * N)
    fl = 2.0 * B * Cout * Cin * k * T
    print(f"B={B:2d} Cin={Cin:4d} Cout={Cout:4d} T={T:5d} k={k:2d} d={d}  {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TFLOP/s", flush=True)
    return ms


for arg in (sys.argv[1:] or ["0"]):
    S.lib().svc_debug_set_conv_cfg(int(arg))
    print(f"--- debug cfg {arg}")
    tot = sum(run(*sh) for sh in SHAPES)
    print(f"sum {tot*1e3:.0f} us")
