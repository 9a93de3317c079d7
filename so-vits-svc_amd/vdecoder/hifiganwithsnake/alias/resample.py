"""Mirror of vdecoder/hifiganwithsnake/alias/resample.py: parameter/buffer holders with the reference's state_dict keys
(`upsample.filter`, `downsample.lowpass.filter`); the resampling runs fused inside svc_snake_alias_f32."""
from torch import nn

from .filter import LowPassFilter1d, kaiser_sinc_filter1d

__all__ = ["UpSample1d", "DownSample1d"]


class UpSample1d(nn.Module):
    def __init__(self, ratio=2, kernel_size=None, C=None):
        super().__init__()
        self.ratio = ratio
        self.kernel_size = int(6 * ratio // 2) * 2 if kernel_size is None else kernel_size
        self.stride = ratio
        self.pad = self.kernel_size // ratio - 1
        self.pad_left = self.pad * self.stride + (self.kernel_size - self.stride) // 2
        self.pad_right = self.pad * self.stride + (self.kernel_size - self.stride + 1) // 2
        self.register_buffer("filter", kaiser_sinc_filter1d(cutoff=0.5 / ratio, half_width=0.6 / ratio,
                                                            kernel_size=self.kernel_size))

    def forward(self, x, C=None):
        raise NotImplementedError("UpSample1d runs fused inside SnakeAlias (svc_snake_alias_f32)")


class DownSample1d(nn.Module):
    def __init__(self, ratio=2, kernel_size=None, C=None):
        super().__init__()
        self.ratio = ratio
        self.kernel_size = int(6 * ratio // 2) * 2 if kernel_size is None else kernel_size
        self.lowpass = LowPassFilter1d(cutoff=0.5 / ratio, half_width=0.6 / ratio, stride=ratio,
                                       kernel_size=self.kernel_size, C=C)

    def forward(self, x):
        raise NotImplementedError("DownSample1d runs fused inside SnakeAlias (svc_snake_alias_f32)")
