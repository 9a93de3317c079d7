"""Shallow-diffusion training step at the reference's diffusion.yaml size (configs_template/diffusion_template.yaml:
n_layers 20, n_chans 512, n_hidden 256, 768-d units, batch 48, 2 s crops = 172 frames of 128 mel bins)."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "so-vits-svc_amd"))
import torch
from diffusion import solver
from diffusion.unit2mel import Unit2Mel

dev = torch.device("cuda:0")
B, T = int(os.environ.get("B", 48)), int(os.environ.get("T", 172))
torch.manual_seed(0)
net = Unit2Mel(768, 1, False, 128, 20, 512, 256, 1000, 1000).to(dev).train()
torch.nn.init.normal_(net.decoder.denoise_fn.output_projection.weight, std=0.02)
net.decoder.denoise_fn.pack_batches = os.environ.get("PACK", "1") == "1"
step = solver.TrainStep(net, solver.build_optimizer(net, lr=1e-4))
data = dict(units=torch.randn(B, T, 768, device=dev), f0=200 + 100 * torch.rand(B, T, 1, device=dev),
            volume=torch.rand(B, T, 1, device=dev), spk_id=torch.zeros(B, 1, dtype=torch.long, device=dev),
            mel=-6 + 2 * torch.randn(B, T, 128, device=dev))
for graph in ((False,) if os.environ.get("EAGER_ONLY") else (False, True)):
    step.enable_graph(graph)
    for _ in range(3):
        l = step(data)
    torch.cuda.synchronize()
    n = int(os.environ.get("N", 10))
    t0 = time.perf_counter()
    for _ in range(n):
        l = step(data)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    npar = sum(p.numel() for p in net.parameters())
    L, C, H, M = 20, 512, 256, 128
    flop_fwd = 2.0 * B * T * (M * C + L * (C * 2 * C * 3 + H * 2 * C + C * 2 * C) + C * C + C * M + 768 * H)
    print(json.dumps(dict(graph=graph, ms_per_step=round(ms, 2), batch_per_s=round(1e3 / ms, 2), loss=float(l), params=npar,
                          tflops=round(3 * flop_fwd / ms / 1e9, 1))))
