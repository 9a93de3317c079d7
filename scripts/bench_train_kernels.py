"""Micro-benchmark of the training step's kernels at the BASELINE configs[2] shapes (B = 16 per GPU, T = 768 frames, segment 8192):
per-op time (torch.cuda.Event on the current stream = the stream svc_hip launches on), TFLOP/s or GB/s.  A quick same-box A/B tool for
kernel changes (set the SVC_* switches / svc_debug_* knobs around it).  usage: python scripts/bench_train_kernels.py [filter]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
sys.path.insert(0, ROOT)
import svc_hip as S  # noqa: E402

dev = torch.device("cuda:0")
flt = sys.argv[1] if len(sys.argv) > 1 else ""


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # us


def report(name, us, flop=None, bytes_=None):
    extra = ""
    if flop:
        extra += f"  {flop / us * 1e-6:7.1f} TFLOP/s"
    if bytes_:
        extra += f"  {bytes_ / us * 1e-3:7.1f} GB/s"
    print(f"{name:58s} {us:9.1f} us{extra}", flush=True)


def conv_case(name, B, Cin, Cout, T, K, dil=1):
    if flt and flt not in name:
        return
    x = torch.randn(B, Cin, T, device=dev)
    w = torch.randn(Cout, Cin, K, device=dev) * (Cin * K) ** -0.5
    dy = torch.randn(B, Cout, T, device=dev)
    pad = dil * (K - 1) // 2
    wp, wt = S.pack_conv1d_weight(w), S.pack_conv1d_weight_T(w)
    flop = 2.0 * B * Cin * Cout * K * T
    report(f"conv fwd   {name} [B{B},{Cin}->{Cout},k{K},d{dil},T{T}]", timeit(lambda: S.conv1d(x, wp, Cout, K, dil=dil, pad_left=pad, Tout=T)), flop)
    report(f"conv dgrad {name}", timeit(lambda: S.conv1d(dy, wt, Cin, K, dil=dil, pad_left=dil * (K - 1) - pad, Tout=T)), flop)
    out = torch.zeros(Cout, Cin, K, device=dev)
    report(f"conv wgrad {name}", timeit(lambda: S.conv1d_wgrad(dy, x, K, dil, pad, out=out, accumulate=True)), flop)


# DiscriminatorP period 3 (padded rows), last strided layer's output and the stride-1 layer; B = 32 in the D step
conv_case("discP.L5 p3", 32, 1024, 1024, 108, 5, 3)
conv_case("discP.L4 p3 (KSd=2)", 32, 1536, 1024, 108, 2, 3)
conv_case("discP.L5 p11", 32, 1024, 1024, 132, 5, 11)
# WN layers of enc_q / flow, FFN and 1x1 convs of the encoders
conv_case("wn.in k5", 16, 192, 384, 768, 5)
conv_case("wn.res_skip k1", 16, 192, 384, 768, 1)
conv_case("ffn.conv_1 k3", 16, 192, 768, 768, 3)
conv_case("attn.qkv k1", 16, 192, 192, 768, 1)
# MRF ResBlock convs on 8192-sample segments
for C, T in ((256, 128), (128, 1024), (64, 2048), (32, 4096), (16, 8192)):
    for K, d in ((3, 1), (7, 3), (11, 5)):
        conv_case(f"mrf C{C}", 16, C, C, T, K, d)

if not flt or "ln" in flt:
    x = torch.randn(16, 192, 768, device=dev)
    g, b = torch.ones(192, device=dev), torch.zeros(192, device=dev)
    y, mean, rstd = S.layernorm_fwd(x, g, b, 1e-5)
    dy = torch.randn_like(x)
    nb = x.numel() * 4
    report("ln fwd [16,192,768]", timeit(lambda: S.layernorm_fwd(x, g, b, 1e-5)), bytes_=2 * nb)
    report("ln bwd [16,192,768]", timeit(lambda: S.layernorm_bwd(x, g, dy, mean, rstd)), bytes_=3 * nb)

if not flt or "gconv" in flt:
    for (B, Cin, Cout, T, groups) in ((32, 16, 64, 8192, 4), (32, 64, 256, 2048, 16), (32, 256, 1024, 512, 64), (32, 1024, 1024, 128, 256)):
        x = torch.randn(B, Cin, T, device=dev)
        w = torch.randn(Cout, Cin // groups, 41, device=dev) * 0.05
        bias = torch.zeros(Cout, device=dev)
        y = S.gconv1d_fwd(x, w, bias, 4, 20, groups)
        dy = torch.randn_like(y)
        flop = 2.0 * y.numel() * (Cin // groups) * 41
        tag = f"[B{B},{Cin}->{Cout},g{groups},T{T}]"
        report(f"gconv fwd   {tag}", timeit(lambda: S.gconv1d_fwd(x, w, bias, 4, 20, groups)), flop)
        report(f"gconv dgrad {tag}", timeit(lambda: S.gconv1d_dgrad(dy, w, Cin, T, 4, 20, groups)), flop)
        report(f"gconv wgrad {tag}", timeit(lambda: S.gconv1d_wgrad(dy, x, 41, 4, 20, groups)), flop)

if not flt or "gemm" in flt:
    Bh, T, dk = 32, 768, 96
    q, k = torch.randn(Bh, T, dk, device=dev), torch.randn(Bh, T, dk, device=dev)
    report("gemm QK^T [32 x 768x768x96]", timeit(lambda: S.gemm(q, k, (T * dk, dk, 1), (T * dk, 1, dk), Bh, T, T, dk)), 2.0 * Bh * T * T * dk)
    # the training graph's own layouts (svc_autograd.py:658-706): q / k / v / dO [BH, dk, T], P / dS [BH, T, T]
    qc, kc, vc, dO = [torch.randn(Bh, dk, T, device=dev) for _ in range(4)]
    P = torch.randn(Bh, T, T, device=dev)
    out = torch.empty(Bh, dk, T, device=dev)
    fl = 2.0 * Bh * T * T * dk
    report("gemm P = q^T k   (T x T x 96)", timeit(lambda: S.gemm(qc, kc, (dk * T, 1, T), (dk * T, T, 1), Bh, T, T, dk)), fl)
    report("gemm out = v P^T (96 x T x T)", timeit(lambda: S.gemm(vc, P, (dk * T, T, 1), (T * T, 1, T), Bh, dk, T, T, out=out, c_strides=(dk * T, T, 1))), fl)
    report("gemm dV = dO P   (96 x T x T)", timeit(lambda: S.gemm(dO, P, (dk * T, T, 1), (T * T, T, 1), Bh, dk, T, T, out=out, c_strides=(dk * T, T, 1))), fl)
    report("gemm dP = dO^T v (T x T x 96)", timeit(lambda: S.gemm(dO, vc, (dk * T, 1, T), (dk * T, T, 1), Bh, T, T, dk)), fl)
