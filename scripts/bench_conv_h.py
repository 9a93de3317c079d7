"""Micro-benchmark: svc_conv1d_h on the generator's MRF shapes of one 10 s clip (both convs of a ResBlock1 pair: leaky_relu in
front + behind / residual epilogue) and the transposed stages, N launches per hipGraph replay.  Prints us, TFLOP/s, and the GB/s
of the algorithmic HBM bytes (x in + y out (+ residual)).  argv: cfg codes for svc_debug_set_conv_h to compare (default 0 1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import svc_hip as S
dev = torch.device("cuda:0")
N = 10
T0 = 862
cfgs = [int(a) for a in sys.argv[1:]] or [0, 1]


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
    x = S.to_h(torch.randn(1, C, L, device=dev))
    y = torch.empty_like(x)
    for k in (3, 7, 11):
        for d in (1, 3, 5):
            w = S.pack_conv1d_h(torch.randn(C, C, k, device=dev) / (C * k) ** 0.5)
            b = torch.randn(C, device=dev)
            pad = (k * d - d) // 2
            fl = 2.0 * C * C * k * L
            for mode in ("c1", "c2"):
                if mode == "c2" and d != 1:
                    continue
                kw = dict(pre_slope=0.1, post_slope=0.1) if mode == "c1" else dict(res=x)
                by = 2.0 * C * L * (2 if mode == "c1" else 3)
                line = f"C={C:3d} L={L:6d} k={k:2d} d={d} {mode}:"
                for c in cfgs:
                    S.lib().svc_debug_set_conv_h(c)
                    us = timeit(lambda: S.conv1d_h(x, w, C, bias=b, dil=d, pad_left=pad, out=y, **kw))
                    # a ResBlock1 has 3 first convs (d = 1, 3, 5) and 3 second convs (d = 1)
                    tot[c] += us * (3 if mode == "c2" else 1)
                    line += f"  cfg{c} {us:7.1f} us {fl / us / 1e6:6.1f} TF {by / us / 1e3:6.0f} GB/s"
                print(line)
# fused pair (svc_resblock_pair_h) against its two launches
pair = {"two": 0.0, "one": 0.0}
for (C, L) in ((128, T0 * 64), (64, T0 * 128), (32, T0 * 256), (16, T0 * 512)):
    x = S.to_h(torch.randn(1, C, L, device=dev))
    xt, y = torch.empty_like(x), torch.empty_like(x)
    for k in (3, 7, 11):
        for d in (1, 3, 5):
            w1 = S.pack_conv1d_h(torch.randn(C, C, k, device=dev) / (C * k) ** 0.5)
            w2 = S.pack_conv1d_h(torch.randn(C, C, k, device=dev) / (C * k) ** 0.5)
            b = torch.randn(C, device=dev)
            p1, p2 = (k * d - d) // 2, (k - 1) // 2

            def two():
                S.conv1d_h(x, w1, C, bias=b, dil=d, pad_left=p1, pre_slope=0.1, post_slope=0.1, out=xt)
                S.conv1d_h(xt, w2, C, bias=b, pad_left=p2, res=x, out=y)
            t2 = timeit(two)
            t1 = timeit(lambda: S.resblock_pair_h(x, w1, b, w2, b, d, out=y))
            pair["two"] += t2
            pair["one"] += t1
            fl = 4.0 * C * C * k * L
            print(f"pair C={C:3d} L={L:6d} k={k:2d} d={d}: two launches {t2:7.1f} us   fused {t1:7.1f} us  {fl / t1 / 1e6:6.1f} TF  "
                  f"{2.0 * C * L * 3 / t1 / 1e3:6.0f} GB/s")
print("sum over one clip's pairs of the <= 128-channel stages (us):", {k: round(v, 1) for k, v in pair.items()})
for (Cin, L, K, u) in ((256, T0 * 8, 16, 8), (128, T0 * 64, 4, 2), (64, T0 * 128, 4, 2), (32, T0 * 256, 4, 2)):
    x = S.to_h(torch.randn(1, Cin, L, device=dev))
    w = S.pack_conv1d_h(torch.randn(Cin, Cin // 2, K, device=dev) * 0.05, u=u)
    b = torch.randn(Cin // 2, device=dev)
    pad = (K - u + 1) // 2
    res = S.to_h(torch.randn(1, Cin // 2, L * u, device=dev))
    line = f"convT {Cin}->{Cin // 2} L={L} K={K} u={u}:"
    for c in cfgs:
        S.lib().svc_debug_set_conv_h(c)
        us = timeit(lambda: S.conv_transpose1d_h(x, w, Cin // 2, K, u, pad, bias=b, pre_slope=0.1, res=res))
        tot[c] += us
        line += f"  cfg{c} {us:7.1f} us {2.0 * Cin * (Cin // 2) * K * L / us / 1e6:6.1f} TF"
    print(line)
S.lib().svc_debug_set_conv_h(0)
print("sum over one clip's 16-bit conv launches (us):", {c: round(v, 1) for c, v in tot.items()})
