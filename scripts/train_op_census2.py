"""Which Python lines of the engine still issue torch (aten) device ops inside one EAGER training iteration.  A TorchDispatchMode
sees every aten call (the backward pass runs on this thread: torch.autograd.set_multithreading_enabled(False)) and charges it to the
innermost frame inside the repo.  Ops on CPU tensors and pure view / metadata ops are skipped.
usage: train_op_census2.py > gpurun_out/census2.txt"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench
import synthetic_data as W
import train as TR

VIEWS = {"view", "_unsafe_view", "reshape", "expand", "permute", "transpose", "t", "squeeze", "unsqueeze", "slice", "select", "detach",
         "alias", "as_strided", "narrow", "unbind", "split", "split_with_sizes", "chunk", "empty", "empty_like", "empty_strided",
         "new_empty", "new_empty_strided", "is_same_size", "stride", "size", "sym_size", "_local_scalar_dense", "item", "unfold",
         "lift_fresh", "set_", "view_as", "expand_as", "result_type", "is_pinned", "_to_copy_noop", "numel", "record_stream"}


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.overloadpacket.__name__
        if name in VIEWS:
            return out
        ts = [a for a in list(args) + list((kwargs or {}).values()) if torch.is_tensor(a)]
        for a in args:
            if isinstance(a, (list, tuple)):
                ts += [x for x in a if torch.is_tensor(x)]
        if torch.is_tensor(out):
            ts.append(out)
        if not any(t.is_cuda for t in ts):
            return out
        fr = [f for f in traceback.extract_stack() if "/so-vits-svc_amd/" in f.filename or f.filename.startswith(ROOT)]
        fr = [f for f in fr if not f.filename.endswith("train_op_census2.py")]
        where = f"{os.path.relpath(fr[-1].filename, ROOT)}:{fr[-1].lineno} {fr[-1].name}" if fr else "(torch internals)"
        self.by[(name, where)] += 1
        return out


def main():
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
    import modules.commons as commons
    commons.DEVICE_RNG = True                      # as in the replayed iteration
    for _ in range(2):
        step(items)
    torch.cuda.synchronize()
    with torch.autograd.set_multithreading_enabled(False), Census() as cs:
        step(items)
        torch.cuda.synchronize()
    tot = collections.Counter()
    for (name, where), n in cs.by.items():
        tot[name] += n
    print(f"aten device ops in one training iteration: {sum(tot.values())}")
    for name, n in tot.most_common():
        print(f"{name:28s} {n:6d}")
        for (nm, where), c in sorted(cs.by.items(), key=lambda kv: -kv[1]):
            if nm == name:
                print(f"        {c:5d}  {where}")


if __name__ == "__main__":
    main()
