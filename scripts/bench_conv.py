"""Micro-benchmark of svc_conv1d_f32 on the decoder's MRF stage shapes (T=862 -> 10 s clip).  Launches are captured
into a hipGraph (N per replay) so host/ctypes overhead does not pollute the timing.
usage: bench_conv.py [dbgcfg | strip=M ...]   (svc_debug_set_conv_cfg codes: dbg*1000 + noksc*100 + (tilecfg+1);
strip=M sets svc_debug_set_conv_strip(M) for the runs that follow: 0 tiled kernels only, 1 automatic)"""
import os, sys
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
        shapes.append((C, L, k, 1 if k == 3 else 5, C))
extra = [(192, 862, 5, 1, 384), (192, 862, 3, 1, 768), (768, 862, 3, 1, 192), (192, 862, 1, 1, 576)]
N = 10
MODE = os.environ.get("BENCH_CONV_MODE", "both")


def run(Cin, L, k, d, Cout, quiet=False):
    x = torch.randn(1, Cin, L, device=dev)
    w = torch.randn(Cout, Cin, k, device=dev) / (Cin * k) ** 0.5
    b = torch.randn(Cout, device=dev)
    wp = S.pack_conv1d_weight(w)
    out = torch.empty(1, Cout, L, device=dev)
    pad = (k * d - d) // 2
    if MODE == "conv1":      # first conv of a ResBlock1 pair: lrelu in, lrelu out, no residual
        kw = dict(bias=b, dil=d, pad_left=pad, pre_slope=0.1, post_act=S.ACT_LRELU, post_slope=0.1, out=out)
    elif MODE == "pre":      # pre-activation only
        kw = dict(bias=b, dil=d, pad_left=pad, pre_slope=0.1, out=out)
    elif MODE == "post":     # post-activation only
        kw = dict(bias=b, dil=d, pad_left=pad, post_act=S.ACT_LRELU, post_slope=0.1, out=out)
    elif MODE == "plain":
        kw = dict(bias=b, dil=d, pad_left=pad, out=out)
    elif MODE == "conv2":    # second conv: plain in, residual add
        kw = dict(bias=b, dil=d, pad_left=pad, res=x if Cout == Cin else None, res_mode=1 if Cout == Cin else 0, out=out)
    else:                    # round-1/2 form: lrelu in + residual
        kw = dict(bias=b, dil=d, pad_left=pad, pre_slope=0.1, res=x if Cout == Cin else None,
                  res_mode=1 if Cout == Cin else 0, out=out)
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
    fl = 2.0 * Cout * Cin * k * L
    if not quiet:
        print(f"Cin={Cin:4d} Cout={Cout:4d} L={L:7d} k={k:2d} d={d}  {ms*1e3:9.1f} us  {fl/ms/1e9:7.1f} TFLOP/s  "
              f"{(Cin+2*Cout)*L*4/ms/1e6:7.1f} GB/s")
    return ms, fl


for arg in (sys.argv[1:] or ["0"]):
    if arg.startswith("strip="):
        S.lib().svc_debug_set_conv_strip(int(arg[6:]))
        print(f"=== conv strip mode {arg[6:]}")
        continue
    code = int(arg)
    S.lib().svc_debug_set_conv_cfg(code)
    print(f"--- debug cfg {code}")
    tot_ms = tot_fl = 0
    for sh in shapes:
        ms, fl = run(*sh)
        tot_ms += ms; tot_fl += fl
    print(f"MRF-shape mean: {tot_fl/tot_ms/1e9:.1f} TFLOP/s   sum {tot_ms*1e3:.0f} us")
    for sh in extra:
        run(*sh)
