"""The encoder / flow convolutions of ONE utterance (B = 1, T = 862) launch by launch: microseconds per launch with the
operands hot (the same x, w every launch: L2 hits) and cold (every launch of the captured graph has its own x, w out of a pool
larger than L2 + Infinity Cache, as in a clip's replay where each weight is used once).
    python scripts/bench_front_conv.py            (GPU box; SVC_CONV_DIRECT_WK=0 for the four-wave reduction split)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import svc_hip as S  # noqa: E402

dev = torch.device("cuda:0")
T = int(os.environ.get("T", "862"))
SHAPES = [  # name, Cin, Cout, k, epi
    ("ffn1 192->768 k3", 192, 768, 3, 0),
    ("ffn2 768->192 k3", 768, 192, 3, 0),
    ("qkv 192->576 k1", 192, 576, 1, 0),
    ("o 192->192 k1", 192, 192, 1, 0),
    ("wn in 192->384 k5 gate", 192, 384, 5, 1),
    ("pre 768->192 k5", 768, 192, 5, 0),
    ("conv_pre 192->512 k7", 192, 512, 7, 0),
]
POOL = int(os.environ.get("POOL", "64"))


def graph_us(fn_list, reps=3):
    for f in fn_list[:2]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fn_list:
            f()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(fn_list))


def main():
    if os.environ.get("FORCE_CFG"):      # svc_debug_set_conv_cfg code (tile config + 1): 6 = 64 x 32 tiles, 7 = 32 x 32 tiles
        S.tlib().svc_debug_set_conv_cfg(int(os.environ["FORCE_CFG"]))
    print(f"T = {T}, pool {POOL}; us per launch (graph replay, launches back to back)")
    print(f"{'shape':26s} {'hot':>8s} {'cold':>8s} {'MB/launch':>10s} {'GFLOP':>7s}")
    for name, Cin, Cout, k, epi in SHAPES:
        xs, wps, bs = [], [], []
        for i in range(POOL):
            xs.append(torch.randn(1, Cin, T, device=dev))
            w = torch.randn(Cout, Cin, k, device=dev) / (Cin * k) ** 0.5
            wps.append(S.pack_conv1d_weight(w, None, Cout // 2 if epi == 1 else 0))
            bs.append(torch.randn(Cout, device=dev))
        kw = dict(pad_left=(k - 1) // 2, Tout=T)
        if epi == 1:
            kw.update(epi=S.EPI_GATE)
        ys = [None] * POOL

        def mk(i):
            return lambda: S.conv1d(xs[i], wps[i], Cout, k, bias=bs[i], **kw)
        hot = graph_us([mk(0)] * POOL)
        cold = graph_us([mk(i) for i in range(POOL)])
        mb = 4.0 * (Cin * T + Cout * Cin * k + (Cout // (2 if epi else 1)) * T) / 1e6
        print(f"{name:26s} {hot:8.1f} {cold:8.1f} {mb:10.2f} {2.0 * Cout * Cin * k * T / 1e9:7.2f}")


if __name__ == "__main__":
    main()
