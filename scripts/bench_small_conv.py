"""Micro-benchmark of svc_conv1d_f32 on the encoder / flow shapes of one 10 s utterance (B=1, T=862), old LDS-staged split-K
kernels (debug cfg 1000000) vs the register-fed direct kernel (default).  N launches per hipGraph replay."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import svc_hip as S

dev = torch.device("cuda:0")
T = int(os.environ.get("T", 862))
B = int(os.environ.get("B", 1))
shapes = [("pre", 768, 192, 5, 0), ("qkv", 192, 576, 1, 0), ("attn.o", 192, 192, 1, 0), ("ffn1", 192, 768, 3, 0),
          ("ffn2", 768, 192, 3, 0), ("proj", 192, 384, 1, 0), ("flow.pre", 96, 192, 1, 0), ("wn.in+gate", 192, 384, 5, 1),
          ("wn.res_skip", 192, 384, 1, 2), ("flow.post", 192, 96, 1, 0), ("conv_pre", 192, 512, 7, 0)]
N = 20


def run(name, Cin, Cout, k, epi):
    x = torch.randn(B, Cin, T, device=dev)
    w = torch.randn(Cout, Cin, k, device=dev) / (Cin * k) ** 0.5
    b = torch.randn(Cout, device=dev)
    wp = S.pack_conv1d_weight(w, None, Cout // 2 if epi == 1 else 0)
    kw = dict(bias=b, pad_left=(k - 1) // 2)
    if epi == 1:
        kw.update(epi=S.EPI_GATE)
    elif epi == 2:
        res = torch.randn(B, Cout // 2, T, device=dev)
        skip = torch.zeros(B, Cout // 2, T, device=dev)
        kw.update(epi=S.EPI_RES_SKIP, res=res, out=torch.empty_like(res), out2=skip, skip_from=Cout // 2, beta=1.0)
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
    g.replay(); g.replay(); g.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (3 * N)
    fl = 2.0 * B * Cout * Cin * k * T
    return ms * 1e3, fl / ms / 1e9


print(f"B={B} T={T}")
res = {}
for code in ([int(a) for a in sys.argv[1:]] or [1000000, 0]):
    S.tlib().svc_debug_set_conv_cfg(code)
    res[code] = [run(*sh) for sh in shapes]
codes = list(res)
print(f"{'shape':14s} " + "  ".join(f"cfg{c:>8d}: us / TF" for c in codes))
for i, sh in enumerate(shapes):
    print(f"{sh[0]:14s} " + "  ".join(f"{res[c][i][0]:9.1f} {res[c][i][1]:6.1f}      " for c in codes))
for c in codes:
    print(f"cfg {c}: sum {sum(r[0] for r in res[c]):.1f} us")
S.tlib().svc_debug_set_conv_cfg(0)
