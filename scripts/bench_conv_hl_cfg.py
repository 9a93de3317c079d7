"""Tuning aid: svc_conv1d_hl on the generator's MRF shapes under the switches of svc_debug_set_conv_hl (argv: cfg codes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import svc_hip as S
dev = torch.device("cuda:0")
N = 10
T0 = 862
cfgs = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4]


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


tot = {c: 0.0 for c in cfgs}
for (C, L) in ((256, T0 * 8), (128, T0 * 64), (64, T0 * 128), (32, T0 * 256), (16, T0 * 512)):
    xs = S.to_h(torch.randn(1, C, L, device=dev), split=True)
    ys = torch.empty_like(xs)
    for k in (3, 7, 11):
        for d in (1, 5):
            ws = S.pack_conv1d_h(torch.randn(C, C, k, device=dev) / (C * k) ** 0.5, split=True)
            b = torch.randn(C, device=dev)
            pad = (k * d - d) // 2
            fl = 2.0 * C * C * k * L
            for mode in ("c1", "c2"):
                if mode == "c2" and d != 1:
                    continue
                kw = dict(pre_slope=0.1, post_slope=0.1) if mode == "c1" else dict(res=xs)
                line = f"C={C:3d} L={L:6d} k={k:2d} d={d} {mode}:"
                for c in cfgs:
                    S.lib().svc_debug_set_conv_hl(c)
                    us = timeit(lambda: S.conv1d_h(xs, ws, C, bias=b, dil=d, pad_left=pad, out=ys, **kw))
                    tot[c] += us
                    line += f"  cfg{c} {us:6.1f} us {fl / us / 1e6:6.1f} TF"
                print(line)
print("sums (us):", {k: round(v, 1) for k, v in tot.items()})
