"""Time the full-size shallow-diffusion model (configs_template/diffusion_template.yaml: 20 layers x 512 channels, 256 hidden,
128 mels) on a 10 s clip (T = 862 frames): one denoiser call and a 100-step DDIM run (timesteps 1000, speedup 10)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch
from diffusion.unit2mel import Unit2Mel
from oracle import diffusion_oracle as DO
dev = torch.device("cuda:0")
c = dict(input_channel=768, n_spk=4, use_pitch_aug=False, out_dims=128, n_layers=20, n_chans=512, n_hidden=256, timesteps=1000, k_step_max=1000)
net = Unit2Mel(c["input_channel"], c["n_spk"], False, 128, 20, 512, 256, 1000, 1000)
net.load_state_dict(DO.make_state_dict(c, 1), strict=False)
net = net.to(dev).eval()
B, T = 1, 862
units = torch.randn(B, T, 768, device=dev); f0 = 100 + 300 * torch.rand(B, T, 1, device=dev); vol = torch.rand(B, T, 1, device=dev)
spk = torch.tensor([[1]], device=dev)
wn = net.decoder.denoise_fn
x = torch.randn(B, 1, 128, T, device=dev); cond = torch.randn(B, 256, T, device=dev); t = torch.full((B,), 500, device=dev)
for _ in range(3): wn(x, t, cond=cond)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): wn(x, t, cond=cond)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
fl = 2.0 * T * (128 * 512 + 20 * (512 * 1024 * 3 + 512 * 1024) + 512 * 512 + 512 * 128)
print(f"WaveNet denoiser call (eager): {dt*1e3:.3f} ms  ({fl/dt/1e12:.1f} TFLOP/s on {fl/1e9:.1f} GF)")
for _ in range(1): net(units, f0, vol, spk_id=spk, infer=True, infer_speedup=10, method="ddim", use_tqdm=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
mel = net(units, f0, vol, spk_id=spk, infer=True, infer_speedup=10, method="ddim", use_tqdm=False)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"Unit2Mel DDIM, 100 denoiser steps, 10 s clip: {dt*1e3:.1f} ms -> mel {tuple(mel.shape)}")
