"""Micro-benchmark: svc_attention_f32 on the encoder shape of one utterance (B=1, H=2, dk=96, T=862, window 4), 8 vs 16 waves."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
import svc_hip as S
dev = torch.device("cuda:0")
N = 20
for (B, H, dk, T, w) in ((1, 2, 96, 862, 4), (1, 2, 96, 2584, 4), (8, 2, 96, 862, 4), (1, 12, 64, 500, 0)):
    qkv = torch.randn(B, 3 * H * dk, T, device=dev)
    C = H * dk
    ek = torch.randn(2 * w + 1, dk, device=dev) * 0.1 if w else None
    ev = torch.randn(2 * w + 1, dk, device=dev) * 0.1 if w else None
    outs = {}
    for nw in (8, 16, 0):          # 0: automatic + the key-split form (two launches) where the shape takes it
        S.lib().svc_debug_set_attention_waves(100 + (1 if nw == 0 else 0))
        S.lib().svc_debug_set_attention_waves(nw)
        fn = lambda: S.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], H, emb_rel_k=ek, emb_rel_v=ev, window=w)
        outs[nw] = fn().clone()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(N):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (2 * N) * 1e3
        print(f"B={B} H={H} dk={dk} T={T} w={w}: {'split/auto' if nw == 0 else str(nw) + ' waves'} {us:7.1f} us  {4.0*B*H*T*T*dk/us/1e6:5.1f} TF")
    print("   max |diff| 8 vs 16 waves:", (outs[8] - outs[16]).abs().max().item(), " split vs 16:", (outs[0] - outs[16]).abs().max().item())
S.lib().svc_debug_set_attention_waves(0)
S.lib().svc_debug_set_attention_waves(101)
