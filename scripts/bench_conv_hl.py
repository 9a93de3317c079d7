"""Micro-benchmark: the split pipeline's convolution (svc_conv1d_hl: hi / lo fp16 planes, three fp16 MFMA per product) against the fp32
MFMA kernel (svc_conv1d_f32) and the 16-bit one (svc_conv1d_h) on the generator's MRF shapes of one 10 s clip (first conv of a
ResBlock1 pair: leaky_relu in front + behind; second: residual epilogue).  Prints us and delivered TFLOP/s (one multiply-add per
product whatever the instruction count)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import svc_hip as S
dev = torch.device("cuda:0")
N = 10
T0 = 862


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


tot = dict(f32=0.0, split=0.0, h=0.0)
for (C, L) in ((256, T0 * 8), (128, T0 * 64), (64, T0 * 128), (32, T0 * 256), (16, T0 * 512)):
    xf = torch.randn(1, C, L, device=dev)
    yf = torch.empty_like(xf)
    xs, xh = S.to_h(xf, split=True), S.to_h(xf)
    ys, yh = torch.empty_like(xs), torch.empty_like(xh)
    for k in (3, 7, 11):
        for d in (1, 3, 5):
            wd = torch.randn(C, C, k, device=dev) / (C * k) ** 0.5
            wf, ws, wh = S.pack_conv1d_weight(wd), S.pack_conv1d_h(wd, split=True), S.pack_conv1d_h(wd)
            b = torch.randn(C, device=dev)
            pad = (k * d - d) // 2
            fl = 2.0 * C * C * k * L
            for mode in ("c1", "c2"):
                if mode == "c2" and d != 1:
                    continue
                if mode == "c1":
                    f32 = lambda: S.conv1d(xf, wf, C, k, bias=b, dil=d, pad_left=pad, pre_slope=0.1, post_act=S.ACT_LRELU, post_slope=0.1, out=yf)
                    kw = dict(pre_slope=0.1, post_slope=0.1)
                else:
                    f32 = lambda: S.conv1d(xf, wf, C, k, bias=b, dil=d, pad_left=pad, res=xf, res_mode=1, out=yf)
                    kw = dict()
                t32 = timeit(f32)
                tsp = timeit(lambda: S.conv1d_h(xs, ws, C, bias=b, dil=d, pad_left=pad, out=ys, res=xs if mode == "c2" else None, **kw))
                th = timeit(lambda: S.conv1d_h(xh, wh, C, bias=b, dil=d, pad_left=pad, out=yh, res=xh if mode == "c2" else None, **kw))
                m = 3 if mode == "c2" else 1      # a ResBlock1 has 3 first convs (d = 1, 3, 5) and 3 second convs (d = 1)
                tot["f32"] += t32 * m; tot["split"] += tsp * m; tot["h"] += th * m
                print(f"C={C:3d} L={L:6d} k={k:2d} d={d} {mode}:  f32 {t32:7.1f} us {fl / t32 / 1e6:6.1f} TF   split {tsp:7.1f} us {fl / tsp / 1e6:6.1f} TF"
                      f"   16-bit {th:7.1f} us {fl / th / 1e6:6.1f} TF")
print("sum over one clip's MRF convolutions, single launches (us):", {k: round(v, 1) for k, v in tot.items()})

# fused pair (svc_resblock_pair_hl) against its two launches
pair = {"two": 0.0, "one": 0.0}
for (C, L) in ((128, T0 * 64), (64, T0 * 128), (32, T0 * 256), (16, T0 * 512)):
    x = S.to_h(torch.randn(1, C, L, device=dev), split=True)
    xt, y = torch.empty_like(x), torch.empty_like(x)
    for k in (3, 7, 11):
        for d in (1, 3, 5):
            w1 = S.pack_conv1d_h(torch.randn(C, C, k, device=dev) / (C * k) ** 0.5, split=True)
            w2 = S.pack_conv1d_h(torch.randn(C, C, k, device=dev) / (C * k) ** 0.5, split=True)
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
                  f"{4.0 * C * L * 3 / t1 / 1e3:6.0f} GB/s")
print("sum over one clip's split pairs of the <= 128-channel stages (us):", {k: round(v, 1) for k, v in pair.items()})
