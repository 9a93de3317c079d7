"""conv1d_wgrad: register-staged vs LDS-DMA tiles (128- and 64-row blocks), fp32 / bf16 operands, time-split targets, on the
training step's shapes.
usage: wgrad_sweep.py [targets...]   (default 256 512)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import svc_hip as S

dev = torch.device("cuda:0")
SHAPES = [(16, 384, 192, 768, 5, 1), (16, 192, 192, 768, 1, 1), (16, 384, 192, 768, 1, 1), (16, 192, 768, 768, 3, 1),
          (16, 768, 192, 768, 3, 1), (16, 128, 128, 1024, 11, 1), (16, 256, 256, 128, 11, 1), (16, 64, 64, 2048, 7, 1),
          (32, 1024, 1024, 132, 5, 11), (32, 1024, 1536, 132, 2, 11), (16, 192, 192, 768, 5, 1), (16, 192, 192, 768, 3, 1)]
targets = [int(a) for a in sys.argv[1:]] or [256, 512]
# (staging form, block rows): register-staged 128 x 64, LDS-DMA 128 x 64, LDS-DMA 64 x 64, the dispatcher's rule
FORMS = {"reg": (0, 2), "dma128": (1, 2), "dma64": (1, 1), "rule": (2, 0)}
N = 20


def run(dy, x, K, d, mma):
    out = torch.zeros(dy.shape[1], x.shape[1], K, device=dev)
    pad = d * (K - 1) // 2
    S.conv1d_wgrad(dy, x, K, d, pad, out=out, accumulate=True, mma=mma)
    ref = out.clone()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        S.conv1d_wgrad(dy, x, K, d, pad, out=out, accumulate=True, mma=mma)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3, ref


for (B, Ca, Cb, T, K, d) in SHAPES:
    torch.manual_seed(Ca + Cb + K)
    dy = torch.randn(B, Ca, T, device=dev)
    x = torch.randn(B, Cb, T, device=dev)
    gf = 2.0 * B * Ca * Cb * K * T * 1e-9
    print(f"B{B} Ca{Ca} Cb{Cb} T{T} K{K} d{d}  ({gf:.2f} GF)")
    for mma, name in ((S.MMA_F32, "f32 "), (S.MMA_BF16, "bf16")):
        row = {}
        for form in FORMS:
            S.tlib().svc_debug_set_wgrad_target(200000 + FORMS[form][0])
            S.tlib().svc_debug_set_wgrad_target(300000 + FORMS[form][1])
            for tg in targets:
                S.tlib().svc_debug_set_wgrad_target(tg)
                S.tlib().svc_debug_set_wgrad_target(100000 + tg)
                us, ref = run(dy, x, K, d, mma)
                row[(form, tg)] = (us, ref)
        base = row[("reg", targets[0])][1]
        err = max((row[(f, tg)][1] - base).abs().max().item() for tg in targets for f in FORMS) / base.abs().max().item()
        print(f"  {name} " + "  ".join(f"{f}@{tg}: {row[(f, tg)][0]:6.1f} us ({gf / row[(f, tg)][0] * 1e3:5.1f} TF)"
                                        for f in FORMS for tg in targets) + f"   max rel diff vs reg {err:.1e}")
S.tlib().svc_debug_set_wgrad_target(200002)
S.tlib().svc_debug_set_wgrad_target(300000)
S.tlib().svc_debug_set_wgrad_target(256)
S.tlib().svc_debug_set_wgrad_target(100256)
