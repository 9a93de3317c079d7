"""Which call sites launch the element-wise / copy kernels of one training iteration (eager): wraps svc_hip.ew / ew_bct / copy_bct /
reduce_bct and torch's own element-wise launches (via the autograd profiler's op names) and prints counts and bytes per caller."""
import os, sys, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import bench, svc_hip as S, train as TR
import synthetic_data as W

dev = torch.device("cuda:0")
cfg = W.full_config(); hps = bench.train_hps(cfg)
net_g, net_d, og, od = TR.build(hps, dev)
net_g.module.load_state_dict(W.make_train_state_dict(cfg, 1234)); net_d.module.load_state_dict(W.make_mpd_state_dict(1235))
net_g.train(); net_d.train()
os.environ["SVC_TRAIN_GRAPH"] = "0"
step = TR.TrainStep(hps, net_g, net_d, og, od)
items_cpu, T = bench.make_train_items(cfg, 16, 4321)
items = tuple(t.to(dev) if t is not None else None for t in items_cpu)
step(items); step(items); torch.cuda.synchronize()

sites = collections.Counter(); byts = collections.Counter()


def site():
    out = []
    for fr in traceback.extract_stack()[:-2][::-1]:
        fn = os.path.basename(fr.filename)
        if fn in ("svc_hip.py", "count_ew_sites.py"):
            continue
        out.append(f"{fn}:{fr.lineno}:{fr.name}")
        if len(out) == 3:
            break
    return " < ".join(out)


def wrap(name):
    orig = getattr(S, name)

    def f(*a, **k):
        x = a[1] if name in ("ew", "ew_bct") else a[0]
        key = (name, a[0] if name in ("ew", "ew_bct") else "", site())
        sites[key] += 1; byts[key] += x.numel() * 4
        return orig(*a, **k)
    setattr(S, name, f)


for n in ("ew", "ew_bct", "copy_bct", "reduce_bct"):
    wrap(n)
import svc_autograd as A
step(items); torch.cuda.synchronize()
print(f"{sum(sites.values())} wrapped launches in one iteration")
for key, n in sorted(sites.items(), key=lambda kv: -byts[kv[0]])[:60]:
    print(f"{n:4d} x  {byts[key] / n / 1e6:7.2f} MB  {key[0]}({key[1]})  {key[2]}")
# torch-side element-wise launches
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
    step(items); torch.cuda.synchronize()
ops = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name not in ("aten::empty", "aten::empty_like", "aten::view", "aten::as_strided", "aten::empty_strided",
                                                      "aten::reshape", "aten::select", "aten::slice", "aten::detach", "aten::unsqueeze", "aten::squeeze",
                                                      "aten::transpose", "aten::expand", "aten::_unsafe_view", "aten::alias", "aten::permute", "aten::t",
                                                      "aten::narrow", "aten::flatten", "aten::unflatten", "aten::result_type", "aten::resize_", "aten::stride",
                                                      "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense", "aten::lift_fresh", "aten::to", "aten::contiguous"):
        ops[e.name] += 1
print("torch ops in one iteration:", ", ".join(f"{k} x{v}" for k, v in ops.most_common(30)))
