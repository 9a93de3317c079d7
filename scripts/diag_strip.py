"""Diagnostic: conv1d_strip.hip per (case, waves-per-strip, epilogue form) against torch CPU, one subprocess per case so that a
GPU fault in one case does not hide the others.  usage: diag_strip.py            (driver)   |   diag_strip.py CASE_INDEX WPS"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [(0, 2, 128, 128, 1000, 11, 5), (0, 1, 64, 96, 460, 3, 1), (0, 1, 128, 256, 300, 7, 3), (1, 1, 64, 64, 1500, 7, 5),
         (1, 2, 64, 64, 100, 3, 1), (1, 1, 32, 128, 904, 11, 1), (2, 1, 32, 32, 2000, 11, 3), (2, 2, 32, 32, 8, 3, 5),
         (3, 1, 256, 256, 300, 11, 5), (3, 2, 32, 64, 252, 3, 1), (3, 1, 128, 128, 700, 7, 3),
         (4, 1, 256, 256, 300, 11, 5), (4, 2, 64, 96, 460, 3, 1), (4, 1, 128, 128, 700, 7, 3)]

if len(sys.argv) == 1:
    for i in range(len(CASES)):
        for wps in (2, 1):
            r = subprocess.run([sys.executable, __file__, str(i), str(wps)], capture_output=True, text=True, timeout=300)
            out = [l for l in r.stdout.splitlines() if l.startswith("case")]
            print("\n".join(out) if out else f"case {i} wps {wps}: NO OUTPUT")
            if r.returncode:
                print(f"case {i} wps {wps}: rc={r.returncode}  {r.stderr.strip().splitlines()[-1] if r.stderr.strip() else ''}")
    sys.exit(0)

import torch
import torch.nn.functional as F
import svc_hip as S
ci, wps = int(sys.argv[1]), int(sys.argv[2])
arr, B, Cin, Cout, T, KS, dil = CASES[ci]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(arr * 7919 + Cin + Cout + T + KS + dil)
x = torch.randn(B, Cin, T, generator=g)
w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5
b = torch.randn(Cout, generator=g)
res = torch.randn(B, Cout, T, generator=g)
prev = torch.randn(B, Cout, T, generator=g)
pad = (KS * dil - dil) // 2
xd, wp, bd, resd = x.to(dev), S.pack_conv1d_weight(w.to(dev)), b.to(dev), res.to(dev)
conv = lambda xx: F.conv1d(xx, w, b, dilation=dil, padding=pad)
forms = [("conv1", dict(pre_slope=0.1, post_act=S.ACT_LRELU, post_slope=0.1), 0.0, 1.0, lambda: F.leaky_relu(conv(F.leaky_relu(x, 0.1)), 0.1)),
         ("conv2", dict(res=resd, res_mode=1), 0.0, 1.0, lambda: conv(x) + res),
         ("conv2-end", dict(res=resd, res_mode=1), 1.0, 3.0, lambda: (conv(x) + res + prev) / 3.0),
         ("resblock2", dict(pre_slope=0.1, res=resd, res_mode=1), 1.0, 1.0, lambda: conv(F.leaky_relu(x, 0.1)) + res + prev)]
for name, kw, beta, div, ref_fn in forms:
    ref = ref_fn()
    S.lib().svc_debug_set_conv_strip(2 + arr + (10 if wps == 1 else 0))
    out = prev.to(dev).clone()
    guard = torch.full((4096,), 7.0, device=dev)       # canary behind the output allocation
    S.conv1d(xd, wp, Cout, KS, bias=bd, dil=dil, pad_left=pad, out=out, beta=beta, out_div=div, **kw)
    torch.cuda.synchronize()
    o = out.cpu()
    err = (o - ref).abs()
    bad = (err > 1e-4 * max(1.0, ref.abs().max().item())).nonzero()
    where = ""
    if len(bad):
        bb, cc, tt = bad[:, 0], bad[:, 1], bad[:, 2]
        where = f" bad={len(bad)} b[{bb.min()}..{bb.max()}] c[{cc.min()}..{cc.max()}] t[{tt.min()}..{tt.max()}] first={bad[0].tolist()}"
    print(f"case {ci} {CASES[ci]} wps {wps} {name:10s} max|err| {err.max().item():.3e} canary {'ok' if bool((guard == 7.0).all()) else 'CLOBBERED'}{where}", flush=True)
