"""Mirror of vdecoder/hifiganwithsnake/alias/act.py.  `SnakeAlias` (the only activation the nsf-snake-hifigan decoder
uses, :109-130) is ONE HIP kernel here: up-sample x2 -> SnakeBeta (log-scale alpha/beta) -> low-pass down-sample x2,
with the 2x intermediate living in LDS (svc_snake_alias_f32)."""
import torch
from torch import nn
from torch.nn import Parameter

import svc_autograd as A
import svc_hip as S

from .resample import DownSample1d, UpSample1d

__all__ = ["SnakeBeta", "SnakeAlias"]


class SnakeBeta(nn.Module):
    """x + sin^2(a x) / (b + 1e-9), a = alpha (or e^alpha), b likewise (reference :35-92).  Parameter holder: the
    arithmetic is fused into SnakeAlias."""

    def __init__(self, in_features, alpha=1.0, alpha_trainable=True, alpha_logscale=False):
        super().__init__()
        self.in_features = in_features
        self.alpha_logscale = alpha_logscale
        init = torch.zeros(in_features) if alpha_logscale else torch.ones(in_features)
        self.alpha = Parameter(init * alpha)
        self.beta = Parameter(init.clone() * alpha)
        self.alpha.requires_grad = alpha_trainable
        self.beta.requires_grad = alpha_trainable
        self.no_div_by_zero = 0.000000001

    def forward(self, x):
        raise NotImplementedError("SnakeBeta runs fused inside SnakeAlias (svc_snake_alias_f32)")


class SnakeAlias(nn.Module):
    def __init__(self, channels, up_ratio=2, down_ratio=2, up_kernel_size=12, down_kernel_size=12, C=None):
        super().__init__()
        if (up_ratio, down_ratio, up_kernel_size, down_kernel_size) != (2, 2, 12, 12):
            raise NotImplementedError("svc_snake_alias_f32 implements the ratio-2 / 12-tap configuration the decoder uses")
        self.up_ratio = up_ratio
        self.down_ratio = down_ratio
        self.act = SnakeBeta(channels, alpha_logscale=True)
        self.upsample = UpSample1d(up_ratio, up_kernel_size, C)
        self.downsample = DownSample1d(down_ratio, down_kernel_size, C)
        # like the reference (whose depthwise conv blocks are built from the constructor-time filter and are not part
        # of the state_dict), the taps are fixed at construction
        self._taps = [float(v) for v in self.upsample.filter.flatten().tolist()]

    def run_h(self, xh, out=None):
        """The activation on a blocked fp16 tensor of the 16-bit pipeline (svc_snake_alias_h)."""
        return S.snake_alias_h(xh, self.act.alpha, self.act.beta, self._taps, out=out)

    def forward(self, x, C=None, out=None):
        if torch.is_grad_enabled() and (self.act.alpha.requires_grad or getattr(x, "requires_grad", False)):
            return A.snake_alias(x, self.act.alpha, self.act.beta, self._taps)      # training: HIP forward + backward
        return S.snake_alias(x, self.act.alpha, self.act.beta, self._taps, out=out)
