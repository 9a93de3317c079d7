"""CPU oracle of the audio plumbing around `Svc`.  TEST INFRASTRUCTURE ONLY (see svc_oracle.py).

`resample`: torch-CPU restatement of torchaudio.functional.resample as torchaudio 0.13-2.x publish it
(`_get_sinc_resample_kernel` + `_apply_sinc_resample_kernel`: sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99, the
kernel bank built in float64, the signal zero-padded by (width, width + orig) and convolved with stride `orig`) — the call
the reference makes at inference/infer_tool.py:219-222,271-274 (`torchaudio.transforms.Resample`).  torchaudio is not
installed in this image: PARITY UNPINNED against the package itself; the formula is the published one.
`frame_rms`: librosa.feature.rms of librosa 0.9.1 (centred frames, reflect padding) by explicit framing.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def resample(wav, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """wav [..., L] float32 -> [..., ceil(L * new / orig)]."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    if orig == new:
        return wav
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base_freq).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels = (kernels * window * scale).to(torch.float32)             # [new, 1, 2*width + orig]
    shape = wav.shape
    x = wav.reshape(-1, shape[-1]).float()
    length = x.shape[1]
    x = F.pad(x, (width, width + orig))
    y = F.conv1d(x[:, None], kernels, stride=orig)                     # [N, new, frames]
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    target = int(math.ceil(new * length / orig))
    return y[..., :target].reshape(*shape[:-1], target)


def frame_rms(y, frame_length, hop_length):
    y = np.asarray(y, dtype=np.float32)
    yp = np.pad(y, (frame_length // 2, frame_length // 2), mode="reflect")
    n = 1 + (len(yp) - frame_length) // hop_length
    out = np.empty(n, dtype=np.float32)
    for i in range(n):
        fr = yp[i * hop_length:i * hop_length + frame_length].astype(np.float64)
        out[i] = np.sqrt(np.mean(fr * fr))
    return out
