"""MI355X-native mirror of vdecoder/nsf_hifigan/models.py: the stand-alone NSF-HiFiGAN vocoder (mel + f0 -> waveform)
used by the diffusion vocoder wrapper (diffusion/vocoder.py:79-86) and the enhancer (modules/enhancer.py:87) —
SURVEY.md §8b boundary row `vdecoder.nsf_hifigan.models.Generator / load_model / load_config`.

Same generator as vdecoder/hifigan (see that mirror for the kernel mapping) with three differences taken from the
reference: the input is a mel spectrogram (`conv_pre`: num_mels -> upsample_initial_channel, no speaker conditioning,
:230), the ConvTranspose1d / noise-conv paddings are `(k-u)//2` and `stride//2` (:239,244), and the source module
integrates its phase in double precision (SineGen.forward :136-181 -> svc_nsf_source_exact_f32).  `h` is an AttrDict
(config.json next to the checkpoint, :27-35).  Inference only (the reference trains this vocoder elsewhere).
"""
import json
import os

import numpy as np
import torch
from torch import nn

import svc_hip as S
from svc_nn import Conv1d, ConvTranspose1d
from vdecoder.hifigan.models import ResBlock1, ResBlock2, mrf_stage

from .env import AttrDict
from .utils import init_weights

LRELU_SLOPE = 0.1


def load_config(model_path):
    config_file = os.path.join(os.path.split(model_path)[0], "config.json")
    with open(config_file) as f:
        return AttrDict(json.loads(f.read()))


def load_model(model_path, device="cuda"):
    """Reference :17-26: config.json beside the checkpoint, state under 'generator'; weight-norm is folded at pack time
    here, so `remove_weight_norm()` is only called for API parity."""
    h = load_config(model_path)
    generator = Generator(h).to(device)
    cp_dict = torch.load(model_path, map_location=device)
    generator.load_state_dict(cp_dict["generator"])
    generator.eval()
    generator.remove_weight_norm()
    del cp_dict
    return generator, h


class SineGen(nn.Module):
    """Parameter-free (reference :93-181); the arithmetic lives in svc_nsf_source_exact_f32."""

    def __init__(self, samp_rate, harmonic_num=0, sine_amp=0.1, noise_std=0.003, voiced_threshold=0):
        super().__init__()
        if voiced_threshold != 0:
            raise NotImplementedError("non-zero voiced threshold is unused by so-vits-svc")
        self.sine_amp, self.noise_std, self.harmonic_num = sine_amp, noise_std, harmonic_num
        self.dim = harmonic_num + 1
        self.sampling_rate = samp_rate
        self.voiced_threshold = voiced_threshold


class SourceModuleHnNSF(nn.Module):
    def __init__(self, sampling_rate, harmonic_num=0, sine_amp=0.1, add_noise_std=0.003, voiced_threshod=0):
        super().__init__()
        self.sine_amp, self.noise_std = sine_amp, add_noise_std
        self.l_sin_gen = SineGen(sampling_rate, harmonic_num, sine_amp, add_noise_std, voiced_threshod)
        self.l_linear = nn.Linear(harmonic_num + 1, 1)
        self.l_tanh = nn.Tanh()

    def forward(self, f0, upp, noise=None):
        """f0 [B,T] frame rate -> har_source [B,1,T*upp].  noise: optional dict(rand_ini [B,H], sine [B,T*upp,H]);
        otherwise drawn in the reference's order (torch.rand :146, torch.randn_like :178)."""
        B, T = f0.shape
        H = self.l_sin_gen.dim
        if noise is None:
            rand_ini = torch.rand(B, H, device=f0.device)
            nz = torch.randn(B, T * upp, H, device=f0.device)
        else:
            rand_ini, nz = noise["rand_ini"], noise["sine"]
        return S.nsf_source_exact(f0.float(), rand_ini, nz, self.l_linear.weight, self.l_linear.bias, upp,
                                  self.l_sin_gen.sampling_rate, self.sine_amp, self.noise_std)


class Generator(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.h = h
        self.num_kernels = len(h.resblock_kernel_sizes)
        self.num_upsamples = len(h.upsample_rates)
        self.m_source = SourceModuleHnNSF(sampling_rate=h.sampling_rate, harmonic_num=8)
        self.noise_convs = nn.ModuleList()
        c0 = h.upsample_initial_channel
        self.conv_pre = Conv1d(h.num_mels, c0, 7, 1, padding=3, weight_norm=True)
        resblock = ResBlock1 if h.resblock == '1' else ResBlock2
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
            c_cur = c0 // (2 ** (i + 1))
            self.ups.append(ConvTranspose1d(c0 // (2 ** i), c_cur, k, u, padding=(k - u) // 2, weight_norm=True))
            if i + 1 < len(h.upsample_rates):
                stride_f0 = int(np.prod(h.upsample_rates[i + 1:]))
                self.noise_convs.append(Conv1d(1, c_cur, kernel_size=stride_f0 * 2, stride=stride_f0,
                                               padding=stride_f0 // 2))
            else:
                self.noise_convs.append(Conv1d(1, c_cur, kernel_size=1))
        self.resblocks = nn.ModuleList()
        ch = c0
        for i in range(len(self.ups)):
            ch //= 2
            for k, d in zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes):
                self.resblocks.append(resblock(h, ch, k, d))
        self.conv_post = Conv1d(ch, 1, 7, 1, padding=3, weight_norm=True)
        self.ups.apply(init_weights)
        self.conv_post.apply(init_weights)
        self.upp = int(np.prod(h.upsample_rates))

    @torch.no_grad()
    def forward(self, x, f0, noise=None):
        """x [B,num_mels,T], f0 [B,T] -> [B,1,T*upp]  (reference :263-281)."""
        if not x.is_cuda:
            raise S.SvcError("nsf_hifigan.Generator needs CUDA/ROCm tensors: the MI355X engine has no CPU fallback")
        har = self.m_source(f0, self.upp, noise=noise)
        x = self.conv_pre.run(x.float().contiguous())
        for i in range(self.num_upsamples):
            xs = self.noise_convs[i](har)
            x = self.ups[i].run(x, pre_slope=LRELU_SLOPE, res=xs)
            x = mrf_stage(self, [self.resblocks[i * self.num_kernels + j] for j in range(self.num_kernels)], x, xs)
        return self.conv_post.run(x, pre_slope=0.01, post_act=S.ACT_TANH)

    def remove_weight_norm(self):
        print('Removing weight norm...')
        for l in self.ups:
            l.remove_weight_norm()
        for l in self.resblocks:
            l.remove_weight_norm()
        self.conv_pre.remove_weight_norm()
        self.conv_post.remove_weight_norm()
