"""CPU oracle of the stand-alone NSF-HiFiGAN vocoder (vdecoder/nsf_hifigan/models.py).  TEST INFRASTRUCTURE ONLY.

torch-CPU restatement of Generator.forward (:263-281) with SineGen.forward's double-precision phase integration
(:136-181) and SourceModuleHnNSF (:216-218), the random draws (torch.rand :146, torch.randn_like :178) explicit.
Pinned by tests/golden/nsf_hifigan_small.npz from the REAL module (tests/golden/make_golden_nsf_hifigan.py)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import svc_oracle as O
from . import weights as W

LRELU_SLOPE = 0.1


def small_h():
    return dict(num_mels=32, upsample_initial_channel=64, upsample_rates=[4, 4, 2], upsample_kernel_sizes=[8, 8, 4],
                resblock="1", resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, sampling_rate=44100)


def param_shapes(h):
    P = {"m_source.l_linear.weight": (1, 9), "m_source.l_linear.bias": (1,)}

    def conv(name, cout, cin, ks, wn=False, transposed=False):
        shape = (cin, cout, ks) if transposed else (cout, cin, ks)
        P[name + ".bias"] = (cout,)
        if wn:
            P[name + ".weight_g"] = (shape[0], 1, 1)
            P[name + ".weight_v"] = shape
        else:
            P[name + ".weight"] = shape
    c0 = h["upsample_initial_channel"]
    conv("conv_pre", c0, h["num_mels"], 7, wn=True)
    ups = h["upsample_rates"]
    nk = len(h["resblock_kernel_sizes"])
    ch = c0
    for i, (u, k) in enumerate(zip(ups, h["upsample_kernel_sizes"])):
        cin, ch = c0 // 2 ** i, c0 // 2 ** (i + 1)
        conv(f"ups.{i}", ch, cin, k, wn=True, transposed=True)
        if i + 1 < len(ups):
            conv(f"noise_convs.{i}", ch, 1, int(math.prod(ups[i + 1:])) * 2)
        else:
            conv(f"noise_convs.{i}", ch, 1, 1)
        for j, (kk, dd) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            for m in range(len(dd)):
                conv(f"resblocks.{i * nk + j}.convs1.{m}", ch, ch, kk, wn=True)
                conv(f"resblocks.{i * nk + j}.convs2.{m}", ch, ch, kk, wn=True)
    conv("conv_post", 1, ch, 7, wn=True)
    return P


def make_state_dict(h, seed):
    shapes = param_shapes(h)
    return {n: W.make_tensor(n, s, seed, shapes) for n, s in shapes.items()}


def sine_source(f0, sd, rand_ini, noise, upp, sr, sine_amp=0.1, noise_std=0.003):
    """SineGen.forward (:136-181) + SourceModuleHnNSF.forward (:216-218).  f0 [B,T] -> [B, T*upp, 1]."""
    f0 = f0.unsqueeze(-1)
    fn = f0 * torch.arange(1, 10).reshape(1, 1, -1)
    rad = (fn / sr) % 1
    ri = rand_ini.clone()
    ri[:, 0] = 0
    rad[:, 0, :] = rad[:, 0, :] + ri
    over = torch.cumsum(rad.double(), 1).float() * upp
    over = F.interpolate(over.transpose(2, 1), scale_factor=upp, mode="linear", align_corners=True).transpose(2, 1)
    rad_up = F.interpolate(rad.transpose(2, 1), scale_factor=upp, mode="nearest").transpose(2, 1)
    over = over % 1
    idx = (over[:, 1:, :] - over[:, :-1, :]) < 0
    shift = torch.zeros_like(rad_up)
    shift[:, 1:, :] = idx * -1.0
    sines = torch.sin(torch.cumsum(rad_up.double() + shift.double(), dim=1) * 2 * np.pi).float() * sine_amp
    uv = (f0 > 0).float()
    uv = F.interpolate(uv.transpose(2, 1), scale_factor=upp, mode="nearest").transpose(2, 1)
    noise_amp = uv * noise_std + (1 - uv) * sine_amp / 3
    waves = sines * uv + noise_amp * noise
    return torch.tanh(F.linear(waves, sd["m_source.l_linear.weight"], sd["m_source.l_linear.bias"]))


def generator(sd, h, mel, f0, rand_ini, noise):
    ups = h["upsample_rates"]
    upp = int(math.prod(ups))
    har = sine_source(f0, sd, rand_ini, noise, upp, h["sampling_rate"]).transpose(1, 2)
    x = O.conv1d(mel, sd, "conv_pre", padding=3)
    nk = len(h["resblock_kernel_sizes"])
    rb = O.resblock1 if h["resblock"] == "1" else O.resblock2
    for i, (u, k) in enumerate(zip(ups, h["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, O.weight_of(sd, f"ups.{i}"), sd[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        if i + 1 < len(ups):
            s_ = int(math.prod(ups[i + 1:]))
            xs = O.conv1d(har, sd, f"noise_convs.{i}", stride=s_, padding=s_ // 2)
        else:
            xs = O.conv1d(har, sd, f"noise_convs.{i}")
        x = x + xs
        acc = None
        for j, (kk, dd) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            r = rb(x, sd, f"resblocks.{i * nk + j}", kk, dd)
            acc = r if acc is None else acc + r
        x = acc / nk
    x = F.leaky_relu(x)
    return torch.tanh(O.conv1d(x, sd, "conv_post", padding=3))


def get_mel(y, sr=44100, n_mels=128, n_fft=2048, win_size=2048, hop=512, fmin=40, fmax=16000, clip_val=1e-5):
    """vdecoder/nsf_hifigan/nvSTFT.py:63-122 (keyshift 0, speed 1, center False): y [B,L] -> log-mel [B,n_mels,frames].
    The mel basis is oracle.mel.mel_filterbank (librosa itself is absent: basis parity UNPINNED, see oracle/mel.py)."""
    from oracle.mel import mel_filterbank
    basis = torch.from_numpy(mel_filterbank(sr, n_fft, n_mels, fmin, fmax)).float()
    pad_left = (win_size - hop) // 2
    pad_right = max((win_size - hop + 1) // 2, win_size - y.size(-1) - pad_left)
    mode = "reflect" if pad_right < y.size(-1) else "constant"
    yp = F.pad(y.unsqueeze(1), (pad_left, pad_right), mode=mode).squeeze(1)
    spec = torch.stft(yp, n_fft, hop_length=hop, win_length=win_size, window=torch.hann_window(win_size), center=False,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    spec = torch.sqrt(spec.real.pow(2) + spec.imag.pow(2) + 1e-9)
    return torch.log(torch.clamp(torch.matmul(basis, spec), min=clip_val))
