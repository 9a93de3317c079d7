import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch, svc_hip as S
dev = torch.device("cuda:0")
for (B, Ca, Cb, K, d, T) in [(16,16,16,11,1,8192),(16,16,16,7,1,8192),(16,32,32,11,1,4096),(16,32,32,7,1,4096),(32,16,1,15,1,8192),(16,64,64,11,1,2048),(16,128,128,11,1,1024),(16,16,16,3,1,8192)]:
    A = torch.randn(B, Ca, T, device=dev); X = torch.randn(B, Cb, T, device=dev)
    pad = d*(K-1)//2
    db = torch.empty(Ca, device=dev)
    G = S.conv1d_wgrad(A, X, K, d, pad, dbias=db); torch.cuda.synchronize()
    ref = torch.stack([ (A[:, :, max(0,pad-k*d):T-max(0,k*d-pad)].unsqueeze(2) * X[:, :, max(0,k*d-pad):T-max(0,pad-k*d)].unsqueeze(1)).sum((0,3)) for k in range(K)], -1)
    err = (G-ref).abs().max().item()/ref.abs().max().item(); eb = (db - A.sum((0,2))).abs().max().item()/A.sum((0,2)).abs().max().item()
    t0=time.perf_counter()
    for _ in range(20): S.conv1d_wgrad(A, X, K, d, pad, dbias=db)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
    print(f"B{B} Ca{Ca} Cb{Cb} K{K} T{T}: {dt*1e6:8.1f} us  {2*B*Ca*Cb*K*T/dt/1e12:6.2f} TF  relerr {err:.1e} bias {eb:.1e}")
