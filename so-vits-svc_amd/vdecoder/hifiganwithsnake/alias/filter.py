"""Mirror of vdecoder/hifiganwithsnake/alias/filter.py: the Kaiser-windowed sinc low-pass design (host-side, 12 numbers)
and the `LowPassFilter1d` buffer holder.  The filtering itself happens inside svc_snake_alias_f32."""
import math

import torch
from torch import nn

__all__ = ["kaiser_sinc_filter1d", "LowPassFilter1d"]


def kaiser_sinc_filter1d(cutoff, half_width, kernel_size):
    """[1,1,kernel_size] unit-DC-gain low-pass: sinc at `cutoff` (cycles/sample) under a Kaiser window whose beta
    follows the standard attenuation formula for a transition of 4*half_width (reference alias/filter.py:29-58)."""
    half = kernel_size // 2
    atten = 2.285 * (half - 1) * math.pi * (4 * half_width) + 7.95
    if atten > 50.0:
        beta = 0.1102 * (atten - 8.7)
    elif atten >= 21.0:
        beta = 0.5842 * (atten - 21) ** 0.4 + 0.07886 * (atten - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    if kernel_size % 2 == 0:
        t = torch.arange(-half, half) + 0.5
    else:
        t = torch.arange(kernel_size) - half
    if cutoff == 0:
        return torch.zeros(1, 1, kernel_size)
    h = 2 * cutoff * window * torch.sinc(2 * cutoff * t)
    h = h / h.sum()
    return h.view(1, 1, kernel_size)


class LowPassFilter1d(nn.Module):
    """Holds the `filter` buffer (state_dict key `...lowpass.filter`, shape [1,1,12]) like the reference (:61-91)."""

    def __init__(self, cutoff=0.5, half_width=0.6, stride=1, padding=True, padding_mode="replicate", kernel_size=12,
                 C=None):
        super().__init__()
        if cutoff < -0.0:
            raise ValueError("Minimum cutoff must be larger than zero.")
        if cutoff > 0.5:
            raise ValueError("A cutoff above 0.5 does not make sense.")
        if padding_mode != "replicate" or not padding:
            raise NotImplementedError("only replicate padding is used on the so-vits-svc path")
        self.kernel_size = kernel_size
        self.even = kernel_size % 2 == 0
        self.pad_left = kernel_size // 2 - int(self.even)
        self.pad_right = kernel_size // 2
        self.stride = stride
        self.register_buffer("filter", kaiser_sinc_filter1d(cutoff, half_width, kernel_size))

    def forward(self, x):
        raise NotImplementedError("LowPassFilter1d runs fused inside SnakeAlias (svc_snake_alias_f32)")
