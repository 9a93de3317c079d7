"""Time HubertSoft.units on a 10 s / 16 kHz clip + per-kernel-family profile."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import svc_hip as S
from oracle import hubert_oracle as HO
from vencoder.hubert import hubert_model as HM
dev = torch.device("cuda:0")
net = HM.HubertSoft(); net.load_state_dict(HO.make_state_dict(5)); net = net.to(dev).eval()
wav = 0.3 * torch.randn(1, 1, 160000, device=dev)
for _ in range(3): u = net.units(wav)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): u = net.units(wav)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"eager: {dt*1e3:.2f} ms per 10 s clip -> units {tuple(u.shape)}")
S.prof_enable(True); S.prof_reset()
for _ in range(3): net.units(wav)
torch.cuda.synchronize()
rep = S.prof_report(); S.prof_enable(False)
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {k:20s} calls {v['calls']/3:5.0f} {v['ms']/3:8.3f} ms {v['flop']/v['ms']/1e9 if v['ms'] else 0:7.1f} TF")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    u = net.units(wav)
g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize(); print(f"graph: {(time.perf_counter()-t0)/10*1e3:.2f} ms")
