mkdir -p gpurun_out
python - <<'PY'
import os, sys
sys.path.insert(0, "so-vits-svc_amd"); sys.path.insert(0, "scripts")
import torch, svc_hip as S
import importlib.util
spec = importlib.util.spec_from_file_location("bc", "scripts/bench_conv.py")
# inline re-implementation to choose shapes
dev = torch.device("cuda:0")
def run(Cin, L, k, d, Cout, code, B=16):
    S.lib().svc_debug_set_conv_cfg(code)
    x = torch.randn(B, Cin, L, device=dev); w = torch.randn(Cout, Cin, k, device=dev) / (Cin*k)**0.5; b = torch.randn(Cout, device=dev)
    wp = S.pack_conv1d_weight(w); out = torch.empty(B, Cout, L, device=dev); pad = (k*d-d)//2
    kw = dict(bias=b, dil=d, pad_left=pad, pre_slope=0.1, res=x if Cout==Cin else None, res_mode=1 if Cout==Cin else 0, out=out)
    S.conv1d(x, wp, Cout, k, **kw); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): S.conv1d(x, wp, Cout, k, **kw)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/20
    print(f"code {code:3d} Cin={Cin:4d} Cout={Cout:4d} L={L:7d} k={k:2d} d={d} {ms*1e3:8.1f} us {2.0*B*Cout*Cin*k*L/ms/1e9:7.1f} TF", flush=True)
for sh in [(768,500,1,1,2304,1),(768,500,1,1,768,1),(768,500,1,1,3072,1),(3072,500,1,1,768,1),(768,862,1,1,2304,1),(192,862,1,1,576,1)]:
    for code in (0,3,4,5,6,7):
        try: run(sh[0],sh[1],sh[2],sh[3],sh[4],code,B=sh[5])
        except Exception as e: print("fail", sh, code, str(e)[:80])
PY
