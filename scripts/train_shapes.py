"""Per-shape profile of one training iteration (SVC_PROF_SHAPES=1): which conv / wgrad shapes take the time.
usage: train_shapes.py [bf16]   (bf16: the fp16_run + half_type bf16 configuration)"""
import os, sys
os.environ["SVC_PROF_SHAPES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import bench, svc_hip as S, train as TR
import synthetic_data as W
dev = torch.device("cuda:0")
cfg = W.full_config(); hps = bench.train_hps(cfg, bf16=len(sys.argv) > 1 and sys.argv[1] == 'bf16')
net_g, net_d, og, od = TR.build(hps, dev)
net_g.module.load_state_dict(W.make_train_state_dict(cfg, 1234)); net_d.module.load_state_dict(W.make_mpd_state_dict(1235))
net_g.train(); net_d.train()
step = TR.TrainStep(hps, net_g, net_d, og, od)
items_cpu, T = bench.make_train_items(cfg, 16, 4321)
items = tuple(t.to(dev) if t is not None else None for t in items_cpu)
step(items); step(items); torch.cuda.synchronize()
S.prof_enable(True); S.prof_reset(); step(items); torch.cuda.synchronize()
buf = S.C.create_string_buffer(1 << 20); n = S.lib().svc_prof_report(buf, len(buf))
rows = []
for line in buf.raw[:max(n, 0)].decode().splitlines():
    name, calls, ms, flop, byt = line.split()
    rows.append((float(ms), name, int(calls), float(flop)))
tot = sum(r[0] for r in rows)
print(f"T={T} total profiled {tot:.1f} ms")
for ms, name, calls, flop in sorted(rows, reverse=True)[:70]:
    print(f"{ms:8.3f} ms {calls:4d} calls {flop/ms/1e9 if ms else 0:7.1f} TF  {name}")
