"""Micro-benchmark: one ResBlock1 pair of the narrow decoder stages, fused (svc_resblock_pair_f32) vs two svc_conv1d_f32."""
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


tot = [0.0, 0.0]
for C, T in ((16, 441344),):
    for K in (3, 7, 11):
        for d in (1, 3, 5):
            x = torch.randn(1, C, T, device=dev)
            w1 = torch.randn(C, C, K, device=dev) / (C * K) ** 0.5
            w2 = torch.randn(C, C, K, device=dev) / (C * K) ** 0.5
            b1, b2 = torch.randn(C, device=dev), torch.randn(C, device=dev)
            w1p, w2p = S.pack_conv1d_weight(w1), S.pack_conv1d_weight(w2)
            xt, o = torch.empty_like(x), torch.empty_like(x)

            def two():
                S.conv1d(x, w1p, C, K, bias=b1, dil=d, pad_left=d * (K - 1) // 2, pre_slope=0.1, post_act=S.ACT_LRELU, post_slope=0.1, out=xt)
                S.conv1d(xt, w2p, C, K, bias=b2, pad_left=(K - 1) // 2, res=x, res_mode=1, out=o)

            def one():
                S.resblock_pair(x, w1p, b1, w2p, b2, K, d, out=o)
            t2, t1 = timeit(two), timeit(one)
            fl = 4.0 * C * C * K * T
            tot[0] += t2; tot[1] += t1
            print(f"C={C:3d} T={T} K={K:2d} d={d}: two launches {t2:7.1f} us ({fl/t2/1e6:5.1f} TF)   fused {t1:7.1f} us ({fl/t1/1e6:5.1f} TF, "
                  f"{8.0*C*T/t1/1e6:5.2f} TB/s algorithmic)")
print(f"sum: two launches {tot[0]:.0f} us   fused {tot[1]:.0f} us")
