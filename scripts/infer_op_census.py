"""Which Python lines launch aten kernels during ONE SynthesizerTrn.infer (10 s clip, eager): TorchDispatchMode with stack
attribution, one row per (aten op, innermost engine frame), with the element count of the first argument."""
import os, sys, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench

PKG = os.path.join(ROOT, "so-vits-svc_amd")


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        site = "?"
        for fr in reversed(traceback.extract_stack(limit=40)):
            if fr.filename.startswith(PKG) or fr.filename.endswith("bench.py"):
                site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                break
        n = 0
        for a in args:
            if torch.is_tensor(a):
                n = a.numel()
                break
        on_dev = any(torch.is_tensor(a) and a.is_cuda for a in args) or "device" in (kwargs or {})
        self.rows[(name, site, n, on_dev)] += 1
        return func(*args, **(kwargs or {}))


dev = torch.device("cuda:0")
net, cfg, W = bench.build_model(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
if mode == "split":
    net.split_f16()
elif mode == "half":
    net.half()
c, f0, uv, sid = [t.to(dev) for t in W.make_inputs(cfg, 1, bench.T_FRAMES, seed=1234)]
net.enable_graph(False)
net.infer(c, f0, uv, g=sid, noice_scale=0.4)
torch.cuda.synchronize()
with Census() as cs:
    net.infer(c, f0, uv, g=sid, noice_scale=0.4)
torch.cuda.synchronize()
skip = ("detach", "alias", "view", "_unsafe_view", "reshape", "t.default", "transpose", "permute", "expand", "squeeze", "unsqueeze",
        "select", "slice", "as_strided", "unbind", "split", "empty", "is_", "_local_scalar", "sym_", "lift_fresh", "narrow")
tot = 0
print(f"# aten ops of one eager SynthesizerTrn.infer ({mode}); launches-only (views / allocations filtered)")
for (name, site, n, on_dev), k in sorted(cs.rows.items(), key=lambda kv: (-kv[1], kv[0])):
    if any(name.startswith(s) for s in skip):
        continue
    tot += k
    print(f"{k:4d}  {name:34s} n={n:9d}  {site}")
print("total", tot)
