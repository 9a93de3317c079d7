"""Micro-benchmark: the flow (4 couplings, reverse) of one clip (T = 862) on the fp32 launches, the fused fp16 kernel and the fused split
kernel (csrc/flow_fused.hip), with and without the L2 prefetch workgroups; caches flushed between calls (a 1 GB copy: in a clip the
decoder has streamed hundreds of MB since the flow's weights were last touched) or warm.  us per flow call (hipEvents)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "so-vits-svc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import models  # noqa: E402
import svc_hip as S  # noqa: E402
import synthetic_data as W  # noqa: E402

dev = torch.device("cuda:0")
cfg = W.full_config()
sd = {k[len("flow."):]: v for k, v in W.make_state_dict(cfg, 5).items() if k.startswith("flow.")}
flow = models.ResidualCouplingBlock(192, 192, 5, 1, 4, gin_channels=768)
flow.load_state_dict(sd)
flow = flow.to(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 862
x = torch.randn(B, 192, T, device=dev)
mask = torch.ones(B, 1, T, device=dev)
g = torch.randn(B, 768, 1, device=dev)
junk = torch.empty(256 << 20, device=dev, dtype=torch.float32)
junk2 = torch.empty_like(junk)


def timed(flush, n=12):
    ts = []
    with torch.no_grad():
        for _ in range(3):
            flow(x, mask, g=g, reverse=True)
        for _ in range(n):
            if flush:
                junk2.copy_(junk)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            flow(x, mask, g=g, reverse=True)
            e1.record()
            torch.cuda.synchronize()
            ts.append(1e3 * e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


only = os.environ.get("FLOW_BENCH_ONLY")      # "fp16" / "split" (+ FLOW_BENCH_PF=0/1): one variant, for counter passes
if only:
    flow.set_half(True, split=only == "split")
    S.lib().svc_debug_set_coupling_fused(int(os.environ.get("FLOW_BENCH_PF", "0")) | int(os.environ.get("FLOW_BENCH_NT", "2")) << 4)
    print(f"B={B} T={T} fused {only}: flushed {timed(True, 6):8.1f} us")
    sys.exit(0)
for name, half, split, pf in (("fp32 launches", False, False, 1), ("fused fp16 NC64, prefetch", True, False, 1 | 2 << 4),
                              ("fused fp16 NC64, no prefetch", True, False, 0 | 2 << 4), ("fused fp16 NC32, prefetch", True, False, 1 | 1 << 4),
                              ("fused fp16 NC32, no prefetch", True, False, 0 | 1 << 4),
                              ("fused split NC64, prefetch", True, True, 1 | 2 << 4), ("fused split NC64, no prefetch", True, True, 0 | 2 << 4),
                              ("fused split NC32, prefetch", True, True, 1 | 1 << 4), ("fused split NC32, no prefetch", True, True, 0 | 1 << 4)):
    flow.set_half(half, split=split)
    S.lib().svc_debug_set_coupling_fused(pf)
    print(f"B={B} T={T} {name:28s} flushed {timed(True):8.1f} us   warm {timed(False):8.1f} us   (eager launches, 4 couplings)")
