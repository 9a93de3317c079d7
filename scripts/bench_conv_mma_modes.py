"""Micro-benchmark: svc_conv1d_f32 on training-step shapes under the matrix-pipe operand modes (fp32 / bf16 / bf16x6): us and
delivered TFLOP/s per launch, 10 launches per hipGraph replay."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import svc_hip as S
dev = torch.device("cuda:0")
N = 10


def timeit(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(N):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * N) * 1e3


SHAPES = [  # B, Cin, Cout, T, K, dil
    (16, 192, 384, 768, 5, 1), (16, 192, 192, 768, 1, 1), (16, 192, 768, 768, 3, 1), (16, 768, 192, 768, 3, 1),
    (16, 256, 256, 1024, 7, 1), (16, 128, 128, 2048, 11, 5), (16, 64, 64, 4096, 3, 1), (32, 1024, 1024, 132, 5, 1),
    (32, 512, 1024, 400, 5, 1), (32, 128, 512, 1200, 5, 1), (1, 256, 256, 6896, 11, 5), (1, 128, 128, 55168, 7, 3),
]
for (B, Cin, Cout, T, K, d) in SHAPES:
    x = torch.randn(B, Cin, T, device=dev)
    wp = S.pack_conv1d_weight(torch.randn(Cout, Cin, K, device=dev) / (Cin * K) ** 0.5)
    b = torch.randn(Cout, device=dev)
    y = torch.empty(B, Cout, T, device=dev)
    pad = (K * d - d) // 2
    fl = 2.0 * B * Cout * Cin * K * T
    line = f"B={B:2d} {Cin:4d}->{Cout:4d} k={K:2d} d={d} T={T:5d}:"
    for name, mode in (("f32", S.MMA_F32), ("bf16", S.MMA_BF16), ("x6", S.MMA_BF16X6)):
        n0 = S.lib().svc_debug_bf16(-1)
        us = timeit(lambda: S.conv1d(x, wp, Cout, K, bias=b, dil=d, pad_left=pad, out=y, mma=mode))
        took = S.lib().svc_debug_bf16(-1) > n0
        line += f"  {name} {us:7.1f} us {fl / us / 1e6:6.1f} TF{'*' if took else ' '}"
    print(line)
print("(* = the 16-bit-instruction kernel ran)")
