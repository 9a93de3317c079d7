"""Where does bench_extra.bench_e2e's time go?  Times unit-encoder replay, repeat_expand and infer separately and in the
combinations of the e2e leg."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import bench_extra as X
import models, utils
import synthetic_data as W
from vencoder.ContentVec768L12 import ContentVec768L12
from vencoder.hubert import hubert_model as HM

dev = torch.device("cuda:0")
cfg = W.full_config()
kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
net = models.SynthesizerTrn(cfg["spec_channels"], cfg["segment_size"], **kw)
net.load_state_dict(W.make_state_dict(cfg, 1234))
net = net.to(dev).eval()
T = 862
c, f0, uv, sid = [t.to(dev) for t in W.make_inputs(cfg, 1, T, seed=21)]
net.enable_graph(True)
enc = ContentVec768L12(device=dev, model=HM.Hubert())
wav = 0.3 * torch.randn(int(round(T * 512 / 44100 * 16000)), device=dev)
units = lambda: enc.encoder(wav)
units()
replay, u_static = X._graphed(units)
expand = lambda: utils.repeat_expand_2d(u_static.squeeze(0), T, "left").unsqueeze(0)
infer_c = lambda: net.infer(c, f0, uv, g=sid, noice_scale=0.4)
infer_e = lambda: net.infer(expand(), f0, uv, g=sid, noice_scale=0.4)


def t(name, fn, n=10):
    dt = X._timeit(fn, n, warm=3)
    print(f"{name:40s} {1e3 * dt:8.3f} ms", flush=True)


t("unit encoder replay", replay)
t("repeat_expand", expand)
t("infer(c)", infer_c)
t("infer(expand())", infer_e)
t("replay + infer(c)", lambda: (replay(), infer_c()))
t("replay + expand", lambda: (replay(), expand()))
t("replay + infer(expand())", lambda: (replay(), infer_e()))
x = expand()
t("infer(x fixed from expand)", lambda: net.infer(x, f0, uv, g=sid, noice_scale=0.4))
