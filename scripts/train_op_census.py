"""Census of the torch (aten) ops that still launch kernels inside one eager training iteration: which Python lines issue
the small element-wise / copy / fill kernels seen in the rocprof trace.  usage: train_op_census.py > gpurun_out/census.txt"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import bench
import train as TR
import synthetic_data as W
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
cfg = W.full_config()
hps = bench.train_hps(cfg)
torch.manual_seed(1234)
net_g, net_d, og, od = TR.build(hps, dev)
net_g.module.load_state_dict(W.make_train_state_dict(cfg, 1234))
net_d.module.load_state_dict(W.make_mpd_state_dict(1235))
net_g.train(); net_d.train()
step = TR.TrainStep(hps, net_g, net_d, og, od)
items_cpu, T = bench.make_train_items(cfg, bench.TRAIN_B, 4321)
items = tuple(t.to(dev) if t is not None else None for t in items_cpu)
for _ in range(2):
    step(items)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=False) as prof:
    step(items)
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type.name == "CPU" and e.name.startswith("aten::")]
# only leaf aten ops that launched at least one kernel
by = collections.Counter()
cuda_t = collections.Counter()
stacks = collections.defaultdict(collections.Counter)
for e in ev:
    if not e.kernels:
        continue
    if any(c.name.startswith("aten::") and c.kernels for c in e.cpu_children):
        continue
    by[e.name] += 1
    cuda_t[e.name] += sum(k.duration for k in e.kernels)
    fr = [s for s in (e.stack or []) if "/so-vits-svc_amd/" in s or "/repo/" in s]
    stacks[e.name][fr[0] if fr else "(autograd engine / no python frame)"] += 1
print("aten op                         launches   kernel_us")
for k, n in by.most_common(25):
    print(f"{k:30s} {n:8d} {cuda_t[k]:10.0f}")
    for s, c in stacks[k].most_common(8):
        print(f"      {c:6d}  {s}")
print("total launches from aten ops:", sum(by.values()), " kernel time us:", sum(cuda_t.values()))
