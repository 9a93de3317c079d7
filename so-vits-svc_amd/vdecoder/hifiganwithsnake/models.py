"""MI355X-native mirror of vdecoder/hifiganwithsnake/models.py (vocoder_name="nsf-snake-hifigan", BASELINE configs[3]).

Same NSF-HiFiGAN generator as vdecoder/hifigan/models.py, with every leaky_relu replaced by an anti-aliased Snake
activation (`SnakeAlias`, alias/act.py): `snakes[i]` before each `ups[i]` (:394-396), two per ResBlock1 dilation
(:62-75), `snake_post` before `conv_post` (:409).  Each activation site is one svc_snake_alias_f32 launch (1 read +
1 write of the activation; the reference's three aten ops move the 2x-length intermediate through memory twice); the
convolutions are the same fused MFMA kernels as the plain generator, called without a pre-activation.
"""
import torch
from torch import nn

import svc_autograd as A
import svc_hip as S
from svc_nn import Conv1d, _no_grad_guard
from vdecoder.hifigan import models as base
from vdecoder.hifigan.models import SineGen, SourceModuleHnNSF  # noqa: F401

from .alias.act import SnakeAlias
from .env import AttrDict  # noqa: F401
from .utils import get_padding, init_weights

LRELU_SLOPE = 0.1


class ResBlock1(nn.Module):
    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3, 5), C=None):
        super().__init__()
        self.h = h
        self.convs1 = nn.ModuleList([Conv1d(channels, channels, kernel_size, 1, dilation=d,
                                            padding=get_padding(kernel_size, d), weight_norm=True) for d in dilation])
        self.convs1.apply(init_weights)
        self.convs2 = nn.ModuleList([Conv1d(channels, channels, kernel_size, 1, dilation=1,
                                            padding=get_padding(kernel_size, 1), weight_norm=True) for _ in dilation])
        self.convs2.apply(init_weights)
        self.num_layers = len(self.convs1) + len(self.convs2)
        self.activations = nn.ModuleList([SnakeAlias(channels, C=C) for _ in range(self.num_layers)])

    def forward_train(self, x):
        """Reference :71-77, one autograd op per reference op."""
        acts1, acts2 = self.activations[::2], self.activations[1::2]
        for c1, c2, a1, a2 in zip(self.convs1, self.convs2, acts1, acts2):
            xt = c1.forward_train(a1(x))
            xt = c2.forward_train(a2(xt))
            x = A.add(xt, x)
        return x

    def forward(self, x, out=None, beta=0.0, out_div=1.0, tmp=None, DIM=None, before_last=None):
        """Reference :71-77.  out (+)= resblock(x); epilogue arguments as hifigan.ResBlock1.forward."""
        n = len(self.convs1)
        acts1, acts2 = self.activations[::2], self.activations[1::2]
        bufs = tmp if tmp is not None else [torch.empty_like(x) for _ in range(4)]
        xt, ping, pong, act = bufs
        cur = x
        for j, (c1, c2, a1, a2) in enumerate(zip(self.convs1, self.convs2, acts1, acts2)):
            a1(cur, out=act)
            c1.run(act, out=xt)
            a2(xt, out=act)
            if j == n - 1:
                dst = out if out is not None else (ping if cur is not ping else pong)
                if before_last is not None:
                    before_last()
                c2.run(act, res=cur, res_mode=1, out=dst, beta=beta, out_div=out_div)
                return dst
            dst = ping if cur is not ping else pong
            c2.run(act, res=cur, res_mode=1, out=dst)
            cur = dst

    def forward_h(self, xh, out=None, beta=0.0, out_div=1.0, tmp=None, before_last=None):
        """The same block on the 16-bit pipeline (blocked fp16 tensors): snake -> conv -> snake -> conv + x per dilation."""
        n = len(self.convs1)
        acts1, acts2 = self.activations[::2], self.activations[1::2]
        bufs = tmp if tmp is not None else [torch.empty_like(xh) for _ in range(4)]
        xt, ping, pong, act = bufs
        cur = xh
        for j, (c1, c2, a1, a2) in enumerate(zip(self.convs1, self.convs2, acts1, acts2)):
            a1.run_h(cur, out=act)
            c1.run_h(act, out=xt)
            a2.run_h(xt, out=act)
            if j == n - 1:
                dst = out if out is not None else (ping if cur is not ping else pong)
                if before_last is not None:
                    before_last()
                c2.run_h(act, res=cur, out=dst, beta=beta, out_div=out_div)
                return dst
            dst = ping if cur is not ping else pong
            c2.run_h(act, res=cur, out=dst)
            cur = dst

    def remove_weight_norm(self):
        for l in list(self.convs1) + list(self.convs2):
            l.remove_weight_norm()


class ResBlock2(nn.Module):
    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3), C=None):
        super().__init__()
        self.h = h
        self.convs = nn.ModuleList([Conv1d(channels, channels, kernel_size, 1, dilation=d,
                                           padding=get_padding(kernel_size, d), weight_norm=True) for d in dilation])
        self.convs.apply(init_weights)
        self.num_layers = len(self.convs)
        self.activations = nn.ModuleList([SnakeAlias(channels, C=C) for _ in range(self.num_layers)])

    def forward_train(self, x):
        """Reference :101-106."""
        for c, a in zip(self.convs, self.activations):
            x = A.add(c.forward_train(a(x)), x)
        return x

    def forward(self, x, out=None, beta=0.0, out_div=1.0, tmp=None, DIM=None, before_last=None):
        """Reference :101-106."""
        n = len(self.convs)
        bufs = tmp if tmp is not None else [torch.empty_like(x) for _ in range(4)]
        _, ping, pong, act = bufs
        cur = x
        for j, (c, a) in enumerate(zip(self.convs, self.activations)):
            a(cur, out=act)
            if j == n - 1:
                dst = out if out is not None else (ping if cur is not ping else pong)
                if before_last is not None:
                    before_last()
                c.run(act, res=cur, res_mode=1, out=dst, beta=beta, out_div=out_div)
                return dst
            dst = ping if cur is not ping else pong
            c.run(act, res=cur, res_mode=1, out=dst)
            cur = dst

    def remove_weight_norm(self):
        for l in self.convs:
            l.remove_weight_norm()


class Generator(base.Generator):
    """Reference :337-418."""

    def __init__(self, h):
        super().__init__(h)
        c0 = h["upsample_initial_channel"]
        resblock = ResBlock1 if h["resblock"] == '1' else ResBlock2
        self.snakes = nn.ModuleList()
        self.resblocks = nn.ModuleList()
        ch = c0
        for i in range(len(self.ups)):
            self.snakes.append(SnakeAlias(c0 // (2 ** i), C=c0 >> i))
            ch = c0 // (2 ** (i + 1))
            for k, d in zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"]):
                self.resblocks.append(resblock(h, ch, k, d, C=c0 >> (i + 1)))
        self.snake_post = SnakeAlias(ch, C=c0 >> len(self.ups))

    def forward_train(self, x, f0, g=None, noise=None):
        """Reference :380-413, one autograd op per reference op (SnakeAlias: HIP forward + backward kernels)."""
        har, _, _ = self.m_source(f0, self.upp, noise=noise)
        x = self.conv_pre.forward_train(x)
        if g is not None:
            x = A.add_bcast(x, self.cond.forward_train(g))
        for i in range(self.num_upsamples):
            x = self.snakes[i](x)
            x = self.ups[i].forward_train(x)
            x = A.add(x, self.noise_convs[i].forward_train(har))
            xs = None
            for j in range(self.num_kernels):
                r = self.resblocks[i * self.num_kernels + j].forward_train(x)
                xs = r if xs is None else A.add(xs, r)
            x = A.scale(xs, 1.0 / self.num_kernels)
        x = self.snake_post(x)
        x = self.conv_post.forward_train(x)
        return A.tanh(x)

    def forward_h(self, x, f0, g=None, noise=None, source=None):
        """forward() in the reference's half-precision mode (base.Generator.set_half): the MRF stages, snakes[1:], ups[1:],
        snake_post and conv_post on blocked fp16 tensors (svc_conv1d_h, svc_snake_alias_h); conv_pre, snakes[0], ups[0] and the
        harmonic source in fp32 as in the plain generator."""
        _no_grad_guard()
        if source is None:
            har, _, _ = self.m_source(f0, self.upp, noise=noise)
        gc = self.cond(g) if g is not None else None
        x = self.conv_pre.run(x, cond=gc)
        if source is not None:
            torch.cuda.current_stream().wait_event(source[1])
        xh = None
        nk = self.num_kernels
        sp = self.half_mode == "split"      # the split pipeline (hi + lo fp16 planes: svc_conv1d_hl, svc_snake_alias_hl)
        with self._range_guard(x.device):
            for i in range(self.num_upsamples):
                xs = source[0][i] if source is not None else self.noise_convs[i](har)
                if i == 0:
                    xh = S.to_h(self.ups[0].run(self.snakes[0](x), res=xs), split=sp, pad16=True)
                else:
                    xh = self.ups[i].run_h(self.snakes[i].run_h(xh), res=S.to_h(xs, split=sp, pad16=True))
                xh = base.mrf_stage(self, [self.resblocks[i * nk + j] for j in range(nk)], xh, torch.empty_like(xh), n_tmp=4, half=True)
            cp = self.conv_post
            return S.conv_post_h(self.snake_post.run_h(xh), cp.dense_weight().reshape(cp.in_channels, cp.kernel_size), cp.bias,
                                 cp.kernel_size, cp.padding, pre_slope=1.0, act=S.ACT_TANH)

    def forward(self, x, f0, g=None, noise=None, source=None):
        """x [B,inter,T], f0 [B,T], g [B,gin,1|T] -> [B,1,T*upp]  (reference :380-413); `source`: see base.Generator.forward."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.conv_post.parameters()):
            return self.forward_train(x, f0, g=g, noise=noise)
        if getattr(self, "half_mode", False):
            return self.forward_h(x, f0, g=g, noise=noise, source=source)
        _no_grad_guard()
        if source is None:
            har, _, _ = self.m_source(f0, self.upp, noise=noise)
        gc = self.cond(g) if g is not None else None
        x = self.conv_pre.run(x, cond=gc)
        if source is not None:
            torch.cuda.current_stream().wait_event(source[1])
        for i in range(self.num_upsamples):
            xs = source[0][i] if source is not None else self.noise_convs[i](har)
            x = self.ups[i].run(self.snakes[i](x), res=xs)              # snake + ConvT + noise-conv add (:395-402)
            x = base.mrf_stage(self, [self.resblocks[i * self.num_kernels + j] for j in range(self.num_kernels)], x, xs, n_tmp=4)
        return self.conv_post.run(self.snake_post(x), post_act=S.ACT_TANH)     # (:409-411)
