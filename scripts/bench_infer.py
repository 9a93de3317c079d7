"""Quick end-to-end timing of SynthesizerTrn.infer (full template, T=862, B=1) + per-kernel-family profile."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import svc_hip as S
import models
import synthetic_data as W

dev = torch.device("cuda:0")
cfg = W.full_config()
kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
net = models.SynthesizerTrn(cfg["spec_channels"], cfg["segment_size"], **kw)
net.load_state_dict(W.make_state_dict(cfg, 1234))
net = net.to(dev).eval()
B, T = int(os.environ.get("B", 1)), int(os.environ.get("T", 862))
c, f0, uv, sid = [t.to(dev) for t in W.make_inputs(cfg, B, T)]

def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        o, _ = net.infer(c, f0, uv, g=sid, noice_scale=0.4)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n, o

dt, o = run(2)
dt, o = run(5)
print(f"eager: {dt*1e3:.2f} ms/clip  {B*T*512/dt/1e6:.2f} Msamples/s  RTF {dt/(T*512/44100):.5f}")
S.prof_enable(True); S.prof_reset()
run(3)
rep = S.prof_report(); S.prof_enable(False)
tot = sum(v["ms"] for v in rep.values())
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {k:16s} calls {v['calls']/3:6.0f}  {v['ms']/3:8.3f} ms  {v['flop']/v['ms']/1e9 if v['ms'] else 0:7.1f} TFLOP/s  {v['bytes']/v['ms']/1e6 if v['ms'] else 0:8.1f} GB/s")
print(f"  sum of kernels {tot/3:.3f} ms")
net.enable_graph(True)
dt, o2 = run(2)
dt, o2 = run(10)
print(f"graph: {dt*1e3:.2f} ms/clip  {B*T*512/dt/1e6:.2f} Msamples/s  RTF {dt/(T*512/44100):.5f}  same={torch.equal(o, o2)}")
