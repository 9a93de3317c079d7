"""Diagnostic: svc_conv1d_f32 direct (register-fed) kernel vs the LDS-staged kernels on random training-like shapes, repeated
to expose nondeterminism."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import svc_hip as S
dev = torch.device("cuda:0")
L = S.tlib()
random.seed(0)
bad = 0
shapes = []
for _ in range(150):
    B = random.choice([1, 2, 16, 32])
    Cin = random.choice([2, 6, 16, 32, 48, 96, 128, 192, 200, 256, 512, 768, 1024])
    Cout = random.choice([1, 2, 16, 24, 32, 64, 96, 128, 192, 384, 512, 576, 1024])
    K = random.choice([1, 3, 5, 7])
    d = random.choice([1, 1, 2, 3, 5, 7, 11])
    T = random.choice([7, 16, 33, 102, 130, 306, 511, 700, 862])
    if B * T > 16000:
        B = max(1, 16000 // T)
    pad = random.choice([0, d * (K - 1) // 2, d * (K - 1)])
    Tout = T + 2 * pad - d * (K - 1)
    if Tout < 1 or (K - 1) * d > 50:
        continue
    shapes.append((B, Cin, Cout, K, d, T, pad, Tout))
for (B, Cin, Cout, K, d, T, pad, Tout) in shapes:
    g = torch.Generator().manual_seed(B * 7 + Cin + Cout * 3 + K + T)
    x = torch.randn(B, Cin, T, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    wp = S.pack_conv1d_weight(w)
    outs = []
    for code in (1000000, 0, 0, 0):
        L.svc_debug_set_conv_cfg(code)
        y = torch.full((B, Cout, Tout), float("nan"), device=dev)
        S.conv1d(x, wp, Cout, K, bias=b, dil=d, pad_left=pad, Tout=Tout, out=y)
        outs.append(y)
    L.svc_debug_set_conv_cfg(0)
    ref = torch.nn.functional.conv1d(x, w, b, padding=pad, dilation=d)[:, :, :Tout]
    sc = max(1.0, ref.abs().max().item())
    e_old = (outs[0] - ref).abs().max().item() / sc
    e_new = max((o - ref).abs().max().item() for o in outs[1:]) / sc
    nd = max((outs[1] - o).abs().max().item() for o in outs[2:])
    flag = "" if (e_new < 1e-4 and nd == 0 and not any(torch.isnan(o).any() for o in outs[1:])) else "   <<<<<< BAD"
    if flag:
        bad += 1
    print(f"B{B} Ci{Cin} Co{Cout} K{K} d{d} T{T} pad{pad} Tout{Tout}: old {e_old:.1e} new {e_new:.1e} nondet {nd:.1e}{flag}")
print("bad:", bad, "of", len(shapes))
