"""(SHAPES=train: the training step's small launches instead.)  Sweep of the conv tilings on SHORT-sequence GEMM-shaped convolutions: the unit encoder's transformer layers (T = 500 frames,
768 <-> 2304 / 3072), the shallow-diffusion denoiser (T = 862, 512 -> 1024 k3 gate, 1x1 projections) — forced tile configs
(svc_debug_set_conv_cfg(cfg + 1): 3 = 128x128, 4 = 64x128, 5 = 64x32 split-K, 6 = 32x32 split-K, 9 = 64x192, 10 = 128x160; 0 =
the dispatcher's own choice; + 1000000 = register-fed direct kernel off).  N launches per hipGraph replay."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import svc_hip as S

dev = torch.device("cuda:0")
TRAIN = [("attn 1x1", 16, 192, 192, 768, 1, 0), ("wn rs dgrad", 16, 384, 192, 768, 1, 0), ("wn res_skip", 16, 192, 384, 768, 1, 0),
         ("discP last T32", 16, 1024, 1024, 32, 5, 0), ("discP last T32 B32", 32, 1024, 1024, 32, 5, 0), ("mrf 256 k11", 16, 256, 256, 128, 11, 0),
         ("discS last", 16, 2048, 512, 16, 2, 0), ("wn in k5", 16, 192, 384, 768, 5, 0), ("ffn k3", 16, 768, 192, 768, 3, 0)]
shapes = [("hubert.qkv", 1, 768, 2304, 500, 1, 0), ("hubert.o", 1, 768, 768, 500, 1, 0), ("hubert.fc1", 1, 768, 3072, 500, 1, 0),
          ("hubert.fc2", 1, 3072, 768, 500, 1, 0), ("hubert.conv2", 1, 512, 512, 16000, 3, 0), ("hubert.conv6", 1, 512, 512, 500, 2, 0),
          ("wavenet.dil+gate", 1, 512, 1024, 862, 3, 1), ("wavenet.out", 1, 512, 1024, 862, 1, 0), ("wavenet.in", 1, 128, 512, 862, 1, 0),
          ("hubert.fc1 B4", 4, 768, 3072, 500, 1, 0), ("hubert.fc2 B4", 4, 3072, 768, 500, 1, 0)]
N = 10


def run(name, B, Cin, Cout, T, k, epi):
    x = torch.randn(B, Cin, T, device=dev)
    w = torch.randn(Cout, Cin, k, device=dev) / (Cin * k) ** 0.5
    b = torch.randn(Cout, device=dev)
    wp = S.pack_conv1d_weight(w, None, Cout // 2 if epi == 1 else 0)
    kw = dict(bias=b, pad_left=(k - 1) // 2, Tout=T)
    if epi == 1:
        kw.update(epi=S.EPI_GATE)
    try:
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
    except Exception as e:      # noqa: BLE001 — a forced tiling may not exist for a shape
        return float("nan"), 0.0
    ms = e0.elapsed_time(e1) / (3 * N)
    return ms * 1e3, 2.0 * B * Cout * Cin * k * T / ms / 1e9


if os.environ.get("SHAPES") == "train":
    shapes = TRAIN
codes = [int(a) for a in sys.argv[1:]] or [0, 4, 5, 6, 7, 10, 11, 1000005, 1000006, 1000007]
res = {}
for code in codes:
    S.tlib().svc_debug_set_conv_cfg(code)
    res[code] = [run(*sh) for sh in shapes]
S.tlib().svc_debug_set_conv_cfg(0)
print(f"{'shape':18s} " + " ".join(f"{c:>9d}" for c in codes) + "   (us per launch; row below: TFLOP/s)")
for i, sh in enumerate(shapes):
    print(f"{sh[0]:18s} " + " ".join(f"{res[c][i][0]:9.1f}" for c in codes))
    print(f"{'':18s} " + " ".join(f"{res[c][i][1]:9.1f}" for c in codes))
