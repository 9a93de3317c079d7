"""MI355X-native mirror of vdecoder/hifigan/models.py: the NSF-HiFiGAN `dec` of SynthesizerTrn (94 % of the
inference FLOPs).

Reference forward (vdecoder/hifigan/models.py:366-394) = ~165 aten launches per call (15 ResBlocks x (6 convs +
6 leaky_relu + 3 adds), weight-norm recompute, upsample, cumsum ...).  Here:
  * SineGen + SourceModuleHnNSF (:138-166,250-271,307-320)      -> 1 closed-form kernel   (svc_nsf_source_f32)
  * conv_pre + cond(g) (:373-374)                               -> 1 MFMA conv, speaker bias in the epilogue
  * per stage: leaky_relu + ups[i] + noise_convs[i] add (:376-381) -> direct noise conv + polyphase MFMA ConvT
  * ResBlock1 (:60-67): leaky_relu -> conv(k,d) -> leaky_relu -> conv(k,1) -> +x
                                                                -> 2 MFMA convs per dilation, activations in the
                                                                   LDS staging, residual / sum-over-kernels / ÷3
                                                                   in the epilogue (no standalone elementwise op)
  * leaky_relu(0.01) + conv_post + tanh (:390-392)              -> 1 direct conv
"""
import math

import numpy as np
import contextlib

import torch
from torch import nn

import svc_autograd as A
import svc_hip as S
from svc_nn import Conv1d, ConvTranspose1d, _no_grad_guard, training_call

from .env import AttrDict  # noqa: F401
from .utils import get_padding, init_weights

LRELU_SLOPE = 0.1
# training-time epilogue fusions (the ResBlock sums in the conv epilogues); SVC_FUSED_TRAIN=0: one autograd op per reference op
FUSED_TRAIN = __import__("os").environ.get("SVC_FUSED_TRAIN", "1") != "0"
_POSTACT = __import__("os").environ.get("SVC_MRF_POSTACT", "1") != "0"      # A/B switch of the fused second leaky_relu
_MRF_STREAMS = __import__("os").environ.get("SVC_MRF_STREAMS", "1") != "0"   # A/B switch: one HIP stream per MRF ResBlock chain
_FUSE_PAIR = __import__("os").environ.get("SVC_MRF_FUSE_PAIR", "1") != "0"   # A/B switch of svc_resblock_pair_f32
_FUSE_PAIR_H = __import__("os").environ.get("SVC_HALF_FUSE_PAIR", "1") != "0"  # A/B switch of svc_resblock_pair_h (16-bit pipeline)

class ResBlock1(nn.Module):
    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.h = h
        self.convs1 = nn.ModuleList([Conv1d(channels, channels, kernel_size, 1, dilation=d,
                                            padding=get_padding(kernel_size, d), weight_norm=True) for d in dilation])
        self.convs1.apply(init_weights)
        self.convs2 = nn.ModuleList([Conv1d(channels, channels, kernel_size, 1, dilation=1,
                                            padding=get_padding(kernel_size, 1), weight_norm=True) for _ in dilation])
        self.convs2.apply(init_weights)

    def forward_train(self, x):
        """Reference vdecoder/hifigan/models.py:60-67."""
        for c1, c2 in zip(self.convs1, self.convs2):
            if FUSED_TRAIN and c1.fused_train_ok() and c2.fused_train_ok():
                # the leaky ReLU in front of c2 in c1's epilogue (c1's raw output has no other reader), `xt + x` in c2's
                # x feeds lrelu -> c1 and the residual: one autograd node with two outputs, so the gradients meet in one launch
                xa, xr = A.leaky_relu_res(x, LRELU_SLOPE)
                xt = c1.forward_train(xa, post_act=S.ACT_LRELU, post_slope=LRELU_SLOPE)
                x = c2.forward_train(xt, res=xr)
            else:
                xt = c1.forward_train(A.leaky_relu(x, LRELU_SLOPE))
                xt = c2.forward_train(A.leaky_relu(xt, LRELU_SLOPE))
                x = A.add(xt, x)
        return x

    def forward(self, x, out=None, beta=0.0, out_div=1.0, tmp=None, before_last=None):
        """out (+)= resblock(x); the optional epilogue arguments let Generator accumulate the MRF mean in place.
        `before_last()` (optional) is called right before the launch that touches `out` (stream ordering hook)."""
        n = len(self.convs1)
        cur = x
        bufs = tmp if tmp is not None else [torch.empty_like(x) for _ in range(3)]
        xt, ping, pong = bufs
        C = x.shape[1]
        fused = (_FUSE_PAIR and C in S.RESBLOCK_PAIR_CHANNELS and self.convs1[0].kernel_size in S.RESBLOCK_PAIR_KERNELS
                 and not isinstance(x, S.FlipView)
                 and all(240 + (c1.dilation + 1) * (c1.kernel_size - 1) <= 512 for c1 in self.convs1))      # the pair kernel's tile
        if fused and S.RESBLOCK16 and C == 16 and n == 3:
            # the whole block — all three dilation pairs — in ONE launch (svc_resblock16_f32: x read once, result written once, bit-equal
            # to the three pair launches below)
            dst = out if out is not None else ping
            if before_last is not None:
                before_last()
            done = S.resblock16(cur, [(c1.packed(), c1.bias, c2.packed(), c2.bias) for c1, c2 in zip(self.convs1, self.convs2)],
                                self.convs1[0].kernel_size, [c1.dilation for c1 in self.convs1], slope=LRELU_SLOPE, out=dst, beta=beta,
                                out_div=out_div)
            if done is not None:
                return done
        if fused:
            # narrow stages (HBM-bound as separate launches): one kernel per pair, intermediate + residual stay in LDS
            for j, (c1, c2) in enumerate(zip(self.convs1, self.convs2)):
                lastp = j == n - 1
                dst = (out if out is not None else (ping if cur is not ping else pong)) if lastp else (ping if cur is not ping else pong)
                if lastp and before_last is not None:
                    before_last()
                S.resblock_pair(cur, c1.packed(), c1.bias, c2.packed(), c2.bias, c1.kernel_size, c1.dilation, slope=LRELU_SLOPE,
                                out=dst, beta=beta if lastp else 0.0, out_div=out_div if lastp else 1.0)
                cur = dst
            return cur
        # the second leaky_relu of a pair (:64) is applied ONCE, in the epilogue of the conv that produces xt, instead of to
        # every operand the next conv stages (xt has no other consumer): max(v, 0.1 v) == (v > 0 ? v : 0.1 v) bit for bit
        for j, (c1, c2) in enumerate(zip(self.convs1, self.convs2)):
            if _POSTACT:
                c1.run(cur, pre_slope=LRELU_SLOPE, post_act=S.ACT_LRELU, post_slope=LRELU_SLOPE, out=xt)
                ps2 = 1.0
            else:
                c1.run(cur, pre_slope=LRELU_SLOPE, out=xt)
                ps2 = LRELU_SLOPE
            if j == n - 1:
                dst = out if out is not None else (ping if cur is not ping else pong)
                if before_last is not None:
                    before_last()
                c2.run(xt, pre_slope=ps2, res=cur, res_mode=1, out=dst, beta=beta, out_div=out_div)
                return dst
            dst = ping if cur is not ping else pong
            c2.run(xt, pre_slope=ps2, res=cur, res_mode=1, out=dst)
            cur = dst

    def forward_h(self, xh, out=None, beta=0.0, out_div=1.0, tmp=None, before_last=None):
        """The same block on the 16-bit pipeline (blocked fp16 tensors, svc_conv1d_h): out = (beta * out + resblock(x)) / out_div."""
        n = len(self.convs1)
        cur = xh
        xt, ping, pong = tmp if tmp is not None else [torch.empty_like(xh) for _ in range(3)]
        sp = S.is_split(xh)
        if _FUSE_PAIR_H and xh.shape[-3] * 8 <= (S.RESBLOCK_PAIR_HL_MAX_C if sp else S.RESBLOCK_PAIR_H_MAX_C) and \
                self.convs1[0].kernel_size in (3, 7, 11):
            # up to 128 channels: one launch per pair, the intermediate (and its halo) never leaves LDS (svc_resblock_pair_h)
            for j, (c1, c2) in enumerate(zip(self.convs1, self.convs2)):
                lastp = j == n - 1
                dst = (out if out is not None else (ping if cur is not ping else pong)) if lastp else (ping if cur is not ping else pong)
                if lastp and before_last is not None:
                    before_last()
                S.resblock_pair_h(cur, c1.packed_h(sp), c1.bias_h(), c2.packed_h(sp), c2.bias_h(), c1.dilation, slope=LRELU_SLOPE, out=dst,
                                  beta=beta if lastp else 0.0, out_div=out_div if lastp else 1.0)
                cur = dst
            return cur
        for j, (c1, c2) in enumerate(zip(self.convs1, self.convs2)):
            # the second leaky_relu (:64) once, in the epilogue of the conv that produces xt (no other reader)
            c1.run_h(cur, pre_slope=LRELU_SLOPE, post_slope=LRELU_SLOPE, out=xt)
            if j == n - 1:
                dst = out if out is not None else (ping if cur is not ping else pong)
                if before_last is not None:
                    before_last()
                c2.run_h(xt, res=cur, out=dst, beta=beta, out_div=out_div)
                return dst
            dst = ping if cur is not ping else pong
            c2.run_h(xt, res=cur, out=dst)
            cur = dst

    def remove_weight_norm(self):
        for l in list(self.convs1) + list(self.convs2):
            l.remove_weight_norm()


class ResBlock2(nn.Module):
    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3)):
        super().__init__()
        self.h = h
        self.convs = nn.ModuleList([Conv1d(channels, channels, kernel_size, 1, dilation=d,
                                           padding=get_padding(kernel_size, d), weight_norm=True) for d in dilation])
        self.convs.apply(init_weights)

    def forward_train(self, x):
        """Reference vdecoder/hifigan/models.py:88-93."""
        for c in self.convs:
            if FUSED_TRAIN and c.fused_train_ok():
                xa, xr = A.leaky_relu_res(x, LRELU_SLOPE)
                x = c.forward_train(xa, res=xr)
            else:
                x = A.add(c.forward_train(A.leaky_relu(x, LRELU_SLOPE)), x)
        return x

    def forward(self, x, out=None, beta=0.0, out_div=1.0, tmp=None, before_last=None):
        n = len(self.convs)
        cur = x
        bufs = tmp if tmp is not None else [torch.empty_like(x) for _ in range(3)]
        _, ping, pong = bufs
        for j, c in enumerate(self.convs):
            if j == n - 1:
                dst = out if out is not None else (ping if cur is not ping else pong)
                if before_last is not None:
                    before_last()
                c.run(cur, pre_slope=LRELU_SLOPE, res=cur, res_mode=1, out=dst, beta=beta, out_div=out_div)
                return dst
            dst = ping if cur is not ping else pong
            c.run(cur, pre_slope=LRELU_SLOPE, res=cur, res_mode=1, out=dst)
            cur = dst

    def forward_h(self, xh, out=None, beta=0.0, out_div=1.0, tmp=None, before_last=None):
        """The same block on the 16-bit pipeline (reference :88-93: `x = c(leaky_relu(x)) + x` per conv)."""
        n = len(self.convs)
        cur = xh
        _, ping, pong = tmp if tmp is not None else [torch.empty_like(xh) for _ in range(3)]
        for j, c in enumerate(self.convs):
            if j == n - 1:
                dst = out if out is not None else (ping if cur is not ping else pong)
                if before_last is not None:
                    before_last()
                c.run_h(cur, pre_slope=LRELU_SLOPE, res=cur, out=dst, beta=beta, out_div=out_div)
                return dst
            dst = ping if cur is not ping else pong
            c.run_h(cur, pre_slope=LRELU_SLOPE, res=cur, out=dst)
            cur = dst

    def remove_weight_norm(self):
        for l in self.convs:
            l.remove_weight_norm()


class SineGen(nn.Module):
    """Parameter-free; kept for API parity (vdecoder/hifigan/models.py:103-271).  The arithmetic lives in
    svc_nsf_source_f32 together with SourceModuleHnNSF's Linear+tanh."""

    def __init__(self, samp_rate, harmonic_num=0, sine_amp=0.1, noise_std=0.003, voiced_threshold=0,
                 flag_for_pulse=False):
        super().__init__()
        self.sine_amp = sine_amp
        self.noise_std = noise_std
        self.harmonic_num = harmonic_num
        self.dim = harmonic_num + 1
        self.sampling_rate = samp_rate
        self.voiced_threshold = voiced_threshold
        if flag_for_pulse or voiced_threshold != 0:
            raise NotImplementedError("pulse-train SineGen / non-zero voiced threshold are unused by so-vits-svc")
        self.onnx = False


class SourceModuleHnNSF(nn.Module):
    def __init__(self, sampling_rate, harmonic_num=0, sine_amp=0.1, add_noise_std=0.003, voiced_threshod=0):
        super().__init__()
        self.sine_amp = sine_amp
        self.noise_std = add_noise_std
        self.l_sin_gen = SineGen(sampling_rate, harmonic_num, sine_amp, add_noise_std, voiced_threshod)
        self.l_linear = nn.Linear(harmonic_num + 1, 1)   # parameters only; applied inside the source kernel
        self.l_tanh = nn.Tanh()

    def forward(self, f0, upp, noise=None):
        """f0: FRAME-rate [B,T] (the x`upp` nearest upsample of :369 happens inside the kernel).
        noise: optional dict(rand_ini [B,H], sine [B,T*upp,H]); drawn in the reference's order when absent.
        Returns (har_source [B,1,T*upp], None, None)."""
        B, T = f0.shape
        H = self.l_sin_gen.dim
        L = T * upp
        if noise is None:
            rand_ini = torch.rand(B, H, device=f0.device)                   # :147
            nz = torch.randn(B, L, H, device=f0.device)                     # :266
            torch.randn(B, L, 1, device=f0.device)                          # :319 (drawn, unused, keeps RNG stream aligned)
        else:
            rand_ini, nz = noise["rand_ini"], noise["sine"]
        if training_call(self.l_linear.weight, self.l_linear.bias):
            har = A.nsf_source(f0, rand_ini, nz, self.l_linear.weight, self.l_linear.bias, upp,
                               self.l_sin_gen.sampling_rate, self.sine_amp, self.noise_std)
            return har, None, None
        har = S.nsf_source(f0, rand_ini, nz, self.l_linear.weight, self.l_linear.bias, upp,
                           self.l_sin_gen.sampling_rate, self.sine_amp, self.noise_std)
        return har, None, None


def mrf_stage(owner, blocks, x, acc, n_tmp=3, half=False):
    """acc = mean_j blocks[j](x) for the ResBlocks of one decoder stage (`xs += resblocks[j](x)`; `x = xs / num_kernels`,
    vdecoder/hifigan/models.py:382-388), the blocks accumulating into `acc` in order.

    The blocks are independent chains over the same input; only their LAST launch touches the shared accumulator.  With
    _MRF_STREAMS each chain runs on its own HIP stream (fork after the upsample, ordered accumulation through events, join
    before the next stage), so the memory phase of one chain's tiles (epilogue: residual read + store, 14 % of a 128-channel
    k=11 launch with every CU in the same phase) and its last partial round of tiles overlap another chain's matrix work:
    10 s clip 9.10 -> 8.62 ms, same launches, same accumulation order (k = 3, then 7, then 11), bit-identical output.
    Capturable (torch.cuda.graph follows the fork / join).  (Rounds 4-5 also carried a merged form — the same step of the three chains
    as ONE launch, svc_conv1d_multi_f32: faster back to back on one stream, 84.3 against 78.6 TFLOP/s, and slower than this stream
    schedule in wall time, 7.71 against 7.51 ms per clip, profiles/r05d_infer_merge_modes_realtime.txt; removed in round 6.)"""
    n = len(blocks)
    kw = lambda j: dict(out=acc, beta=0.0 if j == 0 else 1.0, out_div=float(n) if j == n - 1 else 1.0)
    run = (lambda blk, *a, **k: blk.forward_h(*a, **k)) if half else (lambda blk, *a, **k: blk(*a, **k))   # half: blocked fp16 tensors
    if not (_MRF_STREAMS and n > 1 and x.is_cuda):
        tmp = [torch.empty_like(x) for _ in range(n_tmp)]
        for j, blk in enumerate(blocks):
            run(blk, x, tmp=tmp, **kw(j))
        return acc
    main = torch.cuda.current_stream()
    side = owner.__dict__.setdefault("_mrf_streams", {})
    key = (x.device.index, n)
    if key not in side:
        side[key] = [torch.cuda.Stream(device=x.device) for _ in range(n - 1)]
    streams = [main] + side[key]
    # scratch of every chain comes from the MAIN stream's allocator: main joins all chains below before anything is freed
    tmps = [[torch.empty_like(x) for _ in range(n_tmp)] for _ in range(n)]
    fork = torch.cuda.Event()
    fork.record(main)
    done = [torch.cuda.Event() for _ in range(n)]
    for j, blk in enumerate(blocks):
        st = streams[j]
        with torch.cuda.stream(st):
            if j:
                st.wait_event(fork)
            hook = (lambda jj=j, ss=st: ss.wait_event(done[jj - 1])) if j else None
            run(blk, x, tmp=tmps[j], before_last=hook, **kw(j))
            done[j].record(st)
    main.wait_event(done[-1])      # chain j's last launch waited for chain j-1's: the last event covers all of them
    return acc


class Generator(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.h = h
        self.num_kernels = len(h["resblock_kernel_sizes"])
        self.num_upsamples = len(h["upsample_rates"])
        self.m_source = SourceModuleHnNSF(sampling_rate=h["sampling_rate"], harmonic_num=8)
        self.noise_convs = nn.ModuleList()
        c0 = h["upsample_initial_channel"]
        self.conv_pre = Conv1d(h["inter_channels"], c0, 7, 1, padding=3, weight_norm=True)
        resblock = ResBlock1 if h["resblock"] == '1' else ResBlock2
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
            c_cur = c0 // (2 ** (i + 1))
            self.ups.append(ConvTranspose1d(c0 // (2 ** i), c_cur, k, u, padding=(k - u + 1) // 2, weight_norm=True))
            if i + 1 < len(h["upsample_rates"]):
                stride_f0 = int(np.prod(h["upsample_rates"][i + 1:]))
                self.noise_convs.append(Conv1d(1, c_cur, kernel_size=stride_f0 * 2, stride=stride_f0,
                                               padding=(stride_f0 + 1) // 2))
            else:
                self.noise_convs.append(Conv1d(1, c_cur, kernel_size=1))
        self.resblocks = nn.ModuleList()
        for i in range(len(self.ups)):
            ch = c0 // (2 ** (i + 1))
            for k, d in zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"]):
                self.resblocks.append(resblock(h, ch, k, d))
        self.conv_post = Conv1d(ch, 1, 7, 1, padding=3, weight_norm=True)
        self.ups.apply(init_weights)
        self.conv_post.apply(init_weights)
        self.cond = Conv1d(h['gin_channels'], c0, 1)
        self.upp = int(np.prod(h["upsample_rates"]))
        self.onnx = False

    def OnnxExport(self):
        raise NotImplementedError("ONNX export is out of scope of the MI355X engine (SURVEY.md §2 row 23)")

    def forward_train(self, x, f0, g=None, noise=None):
        """Reference vdecoder/hifigan/models.py:366-394, one autograd op per reference op."""
        har, _, _ = self.m_source(f0, self.upp, noise=noise)
        x = self.conv_pre.forward_train(x)
        if g is not None:
            x = A.add_bcast(x, self.cond.forward_train(g))
        for i in range(self.num_upsamples):
            x = A.leaky_relu(x, LRELU_SLOPE)
            x = self.ups[i].forward_train(x)
            x = A.add(x, self.noise_convs[i].forward_train(har))
            xs = None
            for j in range(self.num_kernels):
                r = self.resblocks[i * self.num_kernels + j].forward_train(x)
                xs = r if xs is None else A.add(xs, r)
            x = A.scale(xs, 1.0 / self.num_kernels)
        x = A.leaky_relu(x, 0.01)
        x = self.conv_post.forward_train(x)
        return A.tanh(x)

    def start_source(self, f0, noise):
        """The harmonic source and the five noise convs (:368-370,379) depend on f0 only.  Launched on a side stream as soon as
        f0 is known, they run underneath the encoder / flow instead of in front of the decoder (0.35 ms of a 10 s clip:
        the frame scan of the source is a short serial kernel, the noise convs are HBM-bound).  Returns (per-stage noise-conv
        outputs, event to wait for); every output buffer comes from the CALLER's stream allocator."""
        if not (_MRF_STREAMS and f0.is_cuda) or noise is None:
            return None
        main = torch.cuda.current_stream()
        side = self.__dict__.setdefault("_src_stream", {})
        if f0.device.index not in side:
            side[f0.device.index] = torch.cuda.Stream(device=f0.device)
        st = side[f0.device.index]
        B, T = f0.shape
        L = T * self.upp
        har = torch.empty((B, 1, L), device=f0.device, dtype=torch.float32)
        outs, Li = [], T
        for i, nc in enumerate(self.noise_convs):
            Li *= self.h["upsample_rates"][i]
            outs.append(torch.empty((B, nc.out_channels, Li), device=f0.device, dtype=torch.float32))
        fork, done = torch.cuda.Event(), torch.cuda.Event()
        fork.record(main)
        with torch.cuda.stream(st):
            st.wait_event(fork)
            S.nsf_source(f0, noise["rand_ini"], noise["sine"], self.m_source.l_linear.weight, self.m_source.l_linear.bias,
                         self.upp, self.m_source.l_sin_gen.sampling_rate, self.m_source.sine_amp, self.m_source.noise_std, out=har)
            for nc, o in zip(self.noise_convs, outs):
                nc.run(har, out=o)
            done.record(st)
        return outs, done, har

    def forward(self, x, f0, g=None, noise=None, source=None):
        """x [B,inter,T] (tensor or channel-strided view), f0 [B,T], g [B,gin,1|T] -> [B,1,T*upp].  `source`: the handle of an
        earlier start_source(f0, noise) call (same f0 / noise), else the source is computed here."""
        if training_call(self.conv_post.bias) or (torch.is_grad_enabled() and getattr(x, "requires_grad", False)):
            return self.forward_train(x, f0, g=g, noise=noise)
        if getattr(self, "half_mode", False):
            return self.forward_h(x, f0, g=g, noise=noise, source=source)
        _no_grad_guard(self.conv_pre.weight_v if self.conv_pre.is_weight_norm else self.conv_pre.weight)
        if source is None:
            har, _, _ = self.m_source(f0, self.upp, noise=noise)
        gc = self.cond(g) if g is not None else None                  # [B, C0, 1|T]  (:374)
        x = self.conv_pre.run(x, cond=gc)                              # (:373-374)
        if source is not None:
            torch.cuda.current_stream().wait_event(source[1])
        for i in range(self.num_upsamples):
            xs = source[0][i] if source is not None else self.noise_convs[i](har)     # (:379)
            x = self.ups[i].run(x, pre_slope=LRELU_SLOPE, res=xs)      # lrelu + ConvT + add (:377-381)
            # reuse the noise-conv buffer as MRF accumulator
            x = mrf_stage(self, [self.resblocks[i * self.num_kernels + j] for j in range(self.num_kernels)], x, xs)
        # F.leaky_relu default slope 0.01 (:390), conv_post, tanh
        return self.conv_post.run(x, pre_slope=0.01, post_act=S.ACT_TANH)

    # -- half-precision inference: the reference's `net_g_ms.half()` (inference/infer_tool.py:196-198) -----------------------
    half_mode = False

    def set_half(self, on=True, split=False):
        """Run the generator's convolution stack as the 16-bit pipeline of csrc/conv1d_h.hip: fp16 activations (HBM and LDS) and
        fp16 weights from the first MRF stage on, fp32 accumulation.  What stays fp32, and why: the harmonic source (its phase
        integration needs > 16 bits — the reference's own half mode loses it there), conv_pre and ups[0] on the 862-frame input
        (latency-bound launches, 1.5 % of the FLOPs) and the waveform that conv_post + tanh emit.

        split=True: the SPLIT pipeline of csrc/conv1d_hl.hip instead — every tensor as a hi and a lo fp16 plane (22 mantissa bits),
        every product as three fp16 matrix instructions: fp32-level results from the fp16 matrix pipe (a precision mode of fp32
        inference, not of the reference's half mode)."""
        # Stage widths that are not multiples of 16 (the tiny template's 200 / 100 / 50 / 25 / 12) run zero-padded to the next multiple
        # (svc_nn.Conv1d.packed_h); ResBlock2 and 5-tap kernels are built too.  What is left out: tap counts without a 16-bit instantiation.
        if on and not (all(k in (3, 5, 7, 11) for k in self.h["resblock_kernel_sizes"]) and
                       all(-(-k // u) in (1, 2, 3) for u, k in zip(self.h["upsample_rates"], self.h["upsample_kernel_sizes"]))):
            raise NotImplementedError("half-precision generator: needs ResBlock kernel sizes in {3, 5, 7, 11} and upsample kernels of at "
                                      "most 3 taps per phase")
        self.half_mode = ("split" if split else True) if on else False
        return self

    # -- split pipeline: range guard (include/svc_hip.h, RANGE) --------------------------------------------------------------------
    def range_flag(self, device):
        """The sticky int32 word this generator's split launches report an out-of-range value into (created on first use)."""
        f = self.__dict__.get("_range_flag")
        if f is None or f.device != device:
            f = self.__dict__["_range_flag"] = torch.zeros(1, dtype=torch.int32, device=device)
        return f

    def split_range_exceeded(self, clear=True):
        """True if a split launch since the last clear produced a value outside the fp16 range (|v| > 65504) or a nan — its
        results are then not fp32-level and the caller re-runs the fp32 path.  One device read (synchronises)."""
        f = self.__dict__.get("_range_flag")
        if f is None:
            return False
        bad = bool(int(f.item()))
        if bad and clear:
            f.zero_()
        return bad

    @contextlib.contextmanager
    def _range_guard(self, device):
        if getattr(self, "half_mode", False) != "split":
            yield
            return
        S.hl_range_flag(self.range_flag(device))
        try:
            yield
        finally:
            S.hl_range_flag(None)

    def _stage_channels(self):
        c0 = self.h["upsample_initial_channel"]
        return [c0] + [c0 // (2 ** (i + 1)) for i in range(self.num_upsamples)]

    def forward_h(self, x, f0, g=None, noise=None, source=None):
        """forward() with the MRF stages, ups[1:] and conv_post on blocked fp16 tensors; a stage's three ResBlock chains run on
        concurrent streams and accumulate into the stage mean in order, as in the fp32 form (mrf_stage)."""
        _no_grad_guard(self.conv_pre.weight_v if self.conv_pre.is_weight_norm else self.conv_pre.weight)
        if source is None:
            har, _, _ = self.m_source(f0, self.upp, noise=noise)
        gc = self.cond(g) if g is not None else None
        x = self.conv_pre.run(x, cond=gc)                                           # fp32 (:373-374)
        if source is not None:
            torch.cuda.current_stream().wait_event(source[1])
        xh = None
        nk = self.num_kernels
        sp = self.half_mode == "split"
        with self._range_guard(x.device):
            for i in range(self.num_upsamples):
                xs = source[0][i] if source is not None else self.noise_convs[i](har)     # fp32 [B, C_i, L_i] (:379)
                if i == 0:
                    x = self.ups[0].run(x, pre_slope=LRELU_SLOPE, res=xs)                 # fp32: lrelu + ConvT + add (:377-381)
                    xh = S.to_h(x, split=sp, pad16=True)
                else:
                    xh = self.ups[i].run_h(xh, pre_slope=LRELU_SLOPE, res=S.to_h(xs, split=sp, pad16=True))
                xh = mrf_stage(self, [self.resblocks[i * nk + j] for j in range(nk)], xh, torch.empty_like(xh), half=True)
            cp = self.conv_post
            return S.conv_post_h(xh, cp.dense_weight().reshape(cp.in_channels, cp.kernel_size), cp.bias, cp.kernel_size, cp.padding,
                                 pre_slope=0.01, act=S.ACT_TANH)

    def remove_weight_norm(self):
        for l in self.ups:
            l.remove_weight_norm()
        for l in self.resblocks:
            l.remove_weight_norm()
        self.conv_pre.remove_weight_norm()
        self.conv_post.remove_weight_norm()
