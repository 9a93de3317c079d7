"""Run one conv1d_mfma shape a few times (for rocprofv3 --pmc).  usage: conv_one.py Cin Cout L k d [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import svc_hip as S
Cin, Cout, L, k, d = [int(a) for a in sys.argv[1:6]]
n = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda:0")
if os.environ.get("SVC_CONV_CFG"):
    S.tlib().svc_debug_set_conv_cfg(int(os.environ["SVC_CONV_CFG"]))
x = torch.randn(1, Cin, L, device=dev)
w = torch.randn(Cout, Cin, k, device=dev) / (Cin * k) ** 0.5
b = torch.randn(Cout, device=dev)
wp = S.pack_conv1d_weight(w)
out = torch.empty(1, Cout, L, device=dev)
mode = os.environ.get("BENCH_CONV_MODE", "both")
if mode == "conv1":
    kw = dict(pre_slope=0.1, post_act=S.ACT_LRELU, post_slope=0.1)
elif mode == "conv2":
    kw = dict(res=x if Cin == Cout else None, res_mode=1 if Cin == Cout else 0)
else:
    kw = dict(pre_slope=0.1, res=x if Cin == Cout else None, res_mode=1 if Cin == Cout else 0)
for _ in range(n):
    S.conv1d(x, wp, Cout, k, bias=b, dil=d, pad_left=(k * d - d) // 2, out=out, **kw)
torch.cuda.synchronize()
