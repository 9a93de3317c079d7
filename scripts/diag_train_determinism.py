"""Diagnostic: per-iteration losses of the benchmark's training step (same construction as bench.run_train), graph or eager."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "so-vits-svc_amd")]
import torch
import bench
import synthetic_data as W
import train as TR
dev = torch.device("cuda:0")
if os.environ.get("SVC_CONV_CFG"):
    import svc_hip
    svc_hip.tlib().svc_debug_set_conv_cfg(int(os.environ["SVC_CONV_CFG"]))
cfg = W.full_config()
if os.environ.get("SVC_BENCH_PDROP") is not None:
    cfg["p_dropout"] = float(os.environ["SVC_BENCH_PDROP"])
hps = bench.train_hps(cfg)
torch.manual_seed(1234)
net_g, net_d, og, od = TR.build(hps, dev)
net_g.module.load_state_dict(W.make_train_state_dict(cfg, 1234))
net_d.module.load_state_dict(W.make_mpd_state_dict(1235))
net_g.train(); net_d.train()
step = TR.TrainStep(hps, net_g, net_d, og, od)
step.enable_graph(os.environ.get("GRAPH", "1") == "1")
items_cpu, T = bench.make_train_items(cfg, bench.TRAIN_B, 4321)
items = tuple(t.to(dev) if t is not None else None for t in items_cpu)
torch.manual_seed(99)
out = []
SYNC = os.environ.get("SYNC", "1")
ls = []
for it in range(int(os.environ.get("N", "6"))):
    l = step(items)
    if SYNC == "1" or (SYNC.startswith("at") and it + 1 == int(SYNC[2:])):
        torch.cuda.synchronize()
    ls.append(l)
torch.cuda.synchronize()
for l in ls:
    out.append(f"{float(l['loss_disc']):.4f}/{float(l['loss_kl']):.3f}/{float(l['loss_mel']):.3f}")
ps = torch.cat([p.detach().flatten()[:1000] for p in list(net_g.parameters())[:50]]).double().sum().item()
print("SYNC=" + SYNC, "GRAPH=" + os.environ.get("GRAPH", "1"), " ".join(out), f"psum {ps:.6f}")
