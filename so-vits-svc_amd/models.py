"""MI355X-native mirror of the reference's models.py: SynthesizerTrn and its sub-modules with the reference's
constructor signatures, attribute names and state_dict layout (751 keys for the full template), so
`inference/infer_tool.Svc` / `inference_main.py` can use it unchanged.  All arithmetic runs in libsvc_hip.so.

`SynthesizerTrn.infer` (reference models.py:496-532, incl. automatic f0 prediction, optional hipGraph replay of the whole
path) runs the fused inference kernels; `SynthesizerTrn.forward` (:463-493), `Encoder` (enc_q) and the
MultiPeriodDiscriminator run the training graph on svc_autograd Functions (HIP forward + HIP backward), dropout included.
`use_transformer_flow` (TransformerCouplingBlock) and `use_spectral_norm` discriminators included.
"""
import contextlib
import math
import os

import torch
from torch import nn

import modules.attentions as attentions
import modules.commons as commons
import modules.modules as modules
import svc_autograd as A
import svc_hip as S
import svc_nn
import utils
from svc_nn import Conv1d, mask2d, training_call


class ResidualCouplingBlock(nn.Module):
    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, n_flows=4, gin_channels=0,
                 share_parameter=False):
        super().__init__()
        self.channels = channels
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.dilation_rate = dilation_rate
        self.n_layers = n_layers
        self.n_flows = n_flows
        self.gin_channels = gin_channels
        self.flows = nn.ModuleList()
        self.wn = modules.WN(hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=0,
                             gin_channels=gin_channels) if share_parameter else None
        for _ in range(n_flows):
            self.flows.append(modules.ResidualCouplingLayer(channels, hidden_channels, kernel_size, dilation_rate,
                                                            n_layers, gin_channels=gin_channels, mean_only=True,
                                                            wn_sharing_parameter=self.wn))
            self.flows.append(modules.Flip())

    def forward(self, x, x_mask, g=None, reverse=False, cond=None):
        """Reference models.py:45-52.  The channel Flip between couplings is never materialised: the working
        buffer is addressed through a negative channel stride whenever an odd number of flips is pending.  `cond`: the handle of
        an earlier start_cond(g) call (inference)."""
        if training_call(self.flows[0].pre.weight) or (torch.is_grad_enabled() and x.requires_grad):
            if not reverse:
                for flow in self.flows:
                    x, _ = flow(x, x_mask, g=g, reverse=False)
            else:
                for flow in reversed(self.flows):
                    x = flow(x, x_mask, g=g, reverse=True)
            return x
        return self._run_inplace(x, x_mask, g, reverse, cond=cond)

    # -- 16-bit / split inference: one launch per coupling (csrc/flow_fused.hip) ---------------------------------------------------
    fused_mode = False       # False: fp32 kernels (10 launches per coupling); True: fp16 planes; "split": hi / lo planes

    def set_half(self, on=True, split=False):
        """The flow's part of SynthesizerTrn.half() / split_f16(): each coupling layer as ONE kernel on the fp16 matrix instruction
        (svc_coupling_fused_h) where that kernel is built — WaveNet couplings with plain (not depthwise-separable) layers, 192
        channels, kernel size 5, dilation rate 1 — and the fp32 launches everywhere else (transformer flow, other widths)."""
        self.fused_mode = (("split" if split else True) if on else False) if self._fusable() else False
        return self

    def _fusable(self):
        if os.environ.get("SVC_FLOW_FUSED", "1") == "0" or type(self) is not ResidualCouplingBlock:
            return False
        cs = [f for f in self.flows if isinstance(f, modules.ResidualCouplingLayer)]
        ok = self.channels == 192 and self.hidden_channels == 192 and self.kernel_size == 5 and self.dilation_rate == 1 and \
            1 <= self.n_layers <= 6
        for c in cs:
            wn = c.enc
            ok = ok and isinstance(wn, modules.WN) and all(type(l) is Conv1d for l in list(wn.in_layers) + list(wn.res_skip_layers))
        return bool(ok)

    def _coupling_fused(self, c, view, x_mask, g, reverse, gc=None):
        sp = self.fused_mode == "split"
        wn = c.enc
        if gc is None:
            gc = wn.cond_layer(g) if g is not None else None
        S.coupling_fused_h(view, mask2d(x_mask), gc, (c.pre.packed_h(sp), c.pre.bias),
                           [(l.packed_h(sp), l.bias) for l in wn.in_layers], [(l.packed_h(sp), l.bias) for l in wn.res_skip_layers],
                           (c.post.packed_h(sp), c.post.bias), reverse=reverse, split=sp)

    def start_cond(self, g):
        """The conditioning rows of every coupling's WN — `cond_layer(g)` (modules/modules.py:113-114), a function of the speaker embedding
        alone — on a side stream, so that they run under the encoder instead of on the flow's critical path (4 x 11 us per clip; with
        `flow_share_parameter` it is ONE layer, computed once instead of per coupling).  Returns (dict id(WN) -> tensor, done event) for
        forward(..., cond=...), or None when there is nothing to precompute."""
        import vdecoder.hifigan.models as _gen
        if g is None or not g.is_cuda or type(self) is not ResidualCouplingBlock or not _gen._MRF_STREAMS:
            return None      # (_MRF_STREAMS off: one stream only — infer_many's branches must not fork again inside a forked capture)
        wns = []
        for f in self.flows:
            if isinstance(f, modules.ResidualCouplingLayer) and isinstance(f.enc, modules.WN) and f.enc.gin_channels != 0 \
                    and all(f.enc is not w for w in wns):
                wns.append(f.enc)
        if not wns or training_call(*wns[0].cond_layer.parameters()):
            return None
        main = torch.cuda.current_stream()
        side = self.__dict__.setdefault("_cond_stream", {})
        if g.device.index not in side:
            side[g.device.index] = torch.cuda.Stream(device=g.device)
        st = side[g.device.index]
        fork, done = torch.cuda.Event(), torch.cuda.Event()
        fork.record(main)
        with torch.cuda.stream(st):
            st.wait_event(fork)
            out = {id(w): w.cond_layer(g) for w in wns}
            done.record(st)
        return out, done

    def _run_inplace(self, x, x_mask, g, reverse, cond=None):
        buf = S.copy_bct(x)
        flipped = False
        couplings = [f for f in self.flows if isinstance(f, modules.ResidualCouplingLayer)]
        gcs = {}
        if cond is not None:
            gcs = cond[0]
            torch.cuda.current_stream().wait_event(cond[1])
        if self.fused_mode:
            step = lambda c, view: self._coupling_fused(c, view, x_mask, g, reverse, gc=gcs.get(id(c.enc)))
        else:
            step = lambda c, view: c.apply_inplace(view, x_mask, g=g, reverse=reverse, gc=gcs.get(id(c.enc)))
        if not reverse:
            for c in couplings:
                step(c, S.flip_view(buf) if flipped else buf)
                flipped = not flipped
        else:
            for c in reversed(couplings):
                flipped = not flipped
                step(c, S.flip_view(buf) if flipped else buf)
        if flipped:
            buf = S.copy_bct(S.flip_view(buf))
        return buf


class TransformerCouplingBlock(ResidualCouplingBlock):
    """Reference models.py:54-92 (`use_transformer_flow`): n_flows mean-only couplings whose network is a conditioned
    causal FFT stack (modules.TransformerCouplingLayer), optionally ONE stack shared by all couplings and registered as
    `flow.wn` too.  Inference runs the same in-place / flip-view schedule as the WaveNet flow; in training every FFT layer
    is a dropout site (`dropout_u`: injected draws, consumed in call order)."""

    def __init__(self, channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout, n_flows=4,
                 gin_channels=0, share_parameter=False):
        nn.Module.__init__(self)
        self.channels = channels
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.n_layers = n_layers
        self.n_flows = n_flows
        self.gin_channels = gin_channels
        self.flows = nn.ModuleList()
        self.wn = attentions.FFT(hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout, isflow=True,
                                 gin_channels=gin_channels) if share_parameter else None
        for _ in range(n_flows):
            self.flows.append(modules.TransformerCouplingLayer(channels, hidden_channels, kernel_size, n_layers, n_heads,
                                                               p_dropout, filter_channels, mean_only=True,
                                                               wn_sharing_parameter=self.wn, gin_channels=gin_channels))
            self.flows.append(modules.Flip())

    def forward(self, x, x_mask, g=None, reverse=False, dropout_u=None):
        if training_call(self.flows[0].pre.weight) or (torch.is_grad_enabled() and x.requires_grad):
            if not reverse:
                for flow in self.flows:
                    if isinstance(flow, modules.Flip):
                        x, _ = flow(x, x_mask, g=g, reverse=False)
                    else:
                        x, _ = flow(x, x_mask, g=g, reverse=False, dropout_u=dropout_u)
            else:
                for flow in reversed(self.flows):
                    if isinstance(flow, modules.Flip):
                        x = flow(x, x_mask, g=g, reverse=True)
                    else:
                        x = flow(x, x_mask, g=g, reverse=True, dropout_u=dropout_u)
            return x
        return self._run_inplace(x, x_mask, g, reverse)


class Encoder(nn.Module):
    """Posterior encoder enc_q (reference models.py:95-125); parameters are kept for checkpoint compatibility."""

    def __init__(self, in_channels, out_channels, hidden_channels, kernel_size, dilation_rate, n_layers,
                 gin_channels=0):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.dilation_rate = dilation_rate
        self.n_layers = n_layers
        self.gin_channels = gin_channels
        self.pre = Conv1d(in_channels, hidden_channels, 1)
        self.enc = modules.WN(hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=gin_channels)
        self.proj = Conv1d(hidden_channels, out_channels * 2, 1)

    def forward(self, x, x_lengths, g=None, noise=None):
        x_mask = torch.unsqueeze(commons.sequence_mask(x_lengths, x.size(2)), 1).to(x.dtype)
        m = mask2d(x_mask)
        if training_call(self.pre.weight):
            h = A.mul_bcast(self.pre.forward_train(x), x_mask)
            h = self.enc.forward_train(h, x_mask, g=g)
            stats = A.mul_bcast(self.proj.forward_train(h), x_mask)
            if noise is None:
                noise = torch.randn(x.shape[0], self.out_channels, x.shape[2], device=x.device)      # models.py:123
            z = A.reparam(stats, noise, m, 1.0)
            m_, logs_ = A.chunk_channels(stats, 2, views=True)      # (one gradient buffer in the backward, no zero-filled slices)
            return z, m_, logs_, x_mask
        h = self.pre.run(x, mask=m)
        h = self.enc(h, x_mask, g=g)
        stats = self.proj.run(h, mask=m)
        if noise is None:
            noise = torch.randn(x.shape[0], self.out_channels, x.shape[2], device=x.device)
        z = S.reparam(stats, noise, mask=m, scale=1.0)
        return z, stats[:, :self.out_channels], stats[:, self.out_channels:], x_mask


class TextEncoder(nn.Module):
    def __init__(self, out_channels, hidden_channels, kernel_size, n_layers, gin_channels=0, filter_channels=None,
                 n_heads=None, p_dropout=None):
        super().__init__()
        self.out_channels = out_channels
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.n_layers = n_layers
        self.gin_channels = gin_channels
        self.proj = Conv1d(hidden_channels, out_channels * 2, 1)
        self.f0_emb = nn.Embedding(256, hidden_channels)
        self.enc_ = attentions.Encoder(hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout)

    def forward(self, x, x_mask, f0=None, noice_scale=1, noise=None, x_is_embedded=False, full_mask=False, dropout_u=None):
        """Reference models.py:155-162.  `f0` is the COARSE f0 index tensor (f0_to_coarse output) unless
        x_is_embedded=True, in which case `x` already is (x + f0_emb) * mask (fused by svc_prenet_embed_f32)."""
        m = mask2d(x_mask)
        if training_call(self.proj.weight):
            x = A.add(x, A.embedding_bct(f0, self.f0_emb.weight))                    # models.py:156
            h = self.enc_.forward_train(A.mul_bcast(x, x_mask), x_mask, dropout_u=dropout_u)
            stats = A.mul_bcast(self.proj.forward_train(h), x_mask)
            if noise is None:
                noise = torch.randn(stats.shape[0], self.out_channels, stats.shape[2], device=stats.device)   # :160
            z = A.reparam(stats, noise, m, float(noice_scale))
            m_, logs_ = A.chunk_channels(stats, 2, views=True)      # (one gradient buffer in the backward, no zero-filled slices)
            return z, m_, logs_, x_mask
        if not x_is_embedded:
            emb = self.f0_emb.weight[f0].transpose(1, 2).contiguous()      # index gather (no arithmetic)
            x = _add_bc(x, emb)
        h = self.enc_(x, x_mask, full_mask=full_mask)
        stats = self.proj.run(h, mask=m)
        if noise is None:
            noise = torch.randn(stats.shape[0], self.out_channels, stats.shape[2], device=stats.device)   # :160
        z = S.reparam(stats, noise, mask=m, scale=float(noice_scale))
        return z, stats[:, :self.out_channels], stats[:, self.out_channels:], x_mask


class F0Decoder(nn.Module):
    def __init__(self, out_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout,
                 spk_channels=0):
        super().__init__()
        self.out_channels = out_channels
        self.hidden_channels = hidden_channels
        self.filter_channels = filter_channels
        self.n_heads = n_heads
        self.n_layers = n_layers
        self.kernel_size = kernel_size
        self.p_dropout = p_dropout
        self.spk_channels = spk_channels
        self.prenet = Conv1d(hidden_channels, hidden_channels, 3, padding=1)
        self.decoder = attentions.FFT(hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout)
        self.proj = Conv1d(hidden_channels, out_channels, 1)
        self.f0_prenet = Conv1d(1, hidden_channels, 3, padding=1)
        self.cond = Conv1d(spk_channels, hidden_channels, 1)

    def forward(self, x, norm_f0, x_mask, spk_emb=None, dropout_u=None):
        """Reference models.py:328-336:  x += cond(spk); x += f0_prenet(norm_f0); prenet; FFT; proj."""
        m = mask2d(x_mask)
        if training_call(self.prenet.weight):
            x = x.detach()                                                  # models.py:329
            if spk_emb is not None:
                x = A.add_bcast(x, self.cond.forward_train(spk_emb))
            x = A.add(x, self.f0_prenet.forward_train(norm_f0))
            x = A.mul_bcast(self.prenet.forward_train(x), x_mask)
            x = self.decoder.forward_train(A.mul_bcast(x, x_mask), x_mask, dropout_u=dropout_u)
            return A.mul_bcast(self.proj.forward_train(x), x_mask)
        gc = self.cond(spk_emb) if spk_emb is not None else None          # [B,H,1|T]
        # x + cond(g) + f0_prenet(norm_f0): direct conv (Cin=1) with x as residual, then the speaker bias rides as
        # `cond` on the prenet's INPUT side — it is not a per-output bias, so materialise it once:
        h = self.f0_prenet.run(norm_f0, res=x)
        if gc is not None:
            h = _add_bc(h, gc)
        h = self.prenet.run(h, mask=m)
        h = self.decoder(h, x_mask)
        return self.proj.run(h, mask=m)


def _add_bc(x, bc):
    """x[b,c,t] += bc[b,c,0|t]  — expressed as an identity-free epilogue: a 1x1 conv would waste FLOPs, so use the
    copy kernel's sibling: LayerNorm-free residual add via conv1d_direct is overkill; torch's add is plumbing-level
    but we keep arithmetic in HIP: reuse svc_conv1d_f32 with a cached identity weight (C<=256)."""
    C = x.shape[1]
    key = (C, str(x.device))
    w = _add_bc.cache.get(key)
    if w is None:
        w = S.pack_conv1d_weight(torch.eye(C, device=x.device).unsqueeze(-1).contiguous())
        _add_bc.cache[key] = w
    return S.conv1d(x, w, C, 1, cond=bc)


_add_bc.cache = {}


class SynthesizerTrn(nn.Module):
    """Synthesizer (reference models.py:339-532)."""

    def __init__(self, spec_channels, segment_size, inter_channels, hidden_channels, filter_channels, n_heads,
                 n_layers, kernel_size, p_dropout, resblock, resblock_kernel_sizes, resblock_dilation_sizes,
                 upsample_rates, upsample_initial_channel, upsample_kernel_sizes, gin_channels, ssl_dim, n_speakers,
                 sampling_rate=44100, vol_embedding=False, vocoder_name="nsf-hifigan", use_depthwise_conv=False,
                 use_automatic_f0_prediction=True, flow_share_parameter=False, n_flow_layer=4,
                 n_layers_trans_flow=3, use_transformer_flow=False, **kwargs):
        super().__init__()
        self.spec_channels = spec_channels
        self.inter_channels = inter_channels
        self.hidden_channels = hidden_channels
        self.filter_channels = filter_channels
        self.n_heads = n_heads
        self.n_layers = n_layers
        self.kernel_size = kernel_size
        self.p_dropout = p_dropout
        self.resblock = resblock
        self.resblock_kernel_sizes = resblock_kernel_sizes
        self.resblock_dilation_sizes = resblock_dilation_sizes
        self.upsample_rates = upsample_rates
        self.upsample_initial_channel = upsample_initial_channel
        self.upsample_kernel_sizes = upsample_kernel_sizes
        self.segment_size = segment_size
        self.gin_channels = gin_channels
        self.ssl_dim = ssl_dim
        self.vol_embedding = vol_embedding
        self.emb_g = nn.Embedding(n_speakers, gin_channels)
        self.use_depthwise_conv = use_depthwise_conv
        self.use_automatic_f0_prediction = use_automatic_f0_prediction
        self.n_layers_trans_flow = n_layers_trans_flow
        self.use_transformer_flow = use_transformer_flow
        if vol_embedding:
            self.emb_vol = nn.Linear(1, hidden_channels)
        self.pre = Conv1d(ssl_dim, hidden_channels, kernel_size=5, padding=2)
        self.enc_p = TextEncoder(inter_channels, hidden_channels, filter_channels=filter_channels, n_heads=n_heads,
                                 n_layers=n_layers, kernel_size=kernel_size, p_dropout=p_dropout)
        hps = {"sampling_rate": sampling_rate, "inter_channels": inter_channels, "resblock": resblock,
               "resblock_kernel_sizes": resblock_kernel_sizes, "resblock_dilation_sizes": resblock_dilation_sizes,
               "upsample_rates": upsample_rates, "upsample_initial_channel": upsample_initial_channel,
               "upsample_kernel_sizes": upsample_kernel_sizes, "gin_channels": gin_channels,
               "use_depthwise_conv": use_depthwise_conv}
        modules.set_Conv1dModel(self.use_depthwise_conv)
        if vocoder_name == "nsf-snake-hifigan":
            from vdecoder.hifiganwithsnake.models import Generator
        else:
            if vocoder_name != "nsf-hifigan":
                print("[?] Unkown vocoder: use default(nsf-hifigan)")
            from vdecoder.hifigan.models import Generator
        self.dec = Generator(h=hps)
        self.enc_q = Encoder(spec_channels, inter_channels, hidden_channels, 5, 1, 16, gin_channels=gin_channels)
        if use_transformer_flow:            # models.py:438-439
            self.flow = TransformerCouplingBlock(inter_channels, hidden_channels, filter_channels, n_heads,
                                                 n_layers_trans_flow, 5, p_dropout, n_flow_layer, gin_channels=gin_channels,
                                                 share_parameter=flow_share_parameter)
        else:
            self.flow = ResidualCouplingBlock(inter_channels, hidden_channels, 5, 1, n_flow_layer,
                                              gin_channels=gin_channels, share_parameter=flow_share_parameter)
        if self.use_automatic_f0_prediction:
            self.f0_decoder = F0Decoder(1, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout,
                                        spk_channels=gin_channels)
        self.emb_uv = nn.Embedding(2, hidden_channels)
        self.character_mix = False
        self.use_graph = False          # hipGraph replay of the infer path (enable_graph())
        self._graphs = {}

    # -- half-precision inference (reference inference/infer_tool.py:196-198: `net_g_ms.half()`) ----------------------------------
    def half(self):
        """The reference's `.half()` makes every tensor of the model fp16.  Here it switches the NSF-HiFiGAN generator (94 % of the
        FLOPs) to the 16-bit pipeline — fp16 activations and weights, fp32 accumulation (vdecoder.hifigan.models.Generator.set_half)
        — and leaves the fp32 master parameters, the encoder / flow kernels and the harmonic source as they are: nothing is
        LESS precise than the reference's half mode, and `list(net.parameters())[0].dtype` stays float32 (Svc casts its inputs to
        that).  Both generators have a 16-bit form (plain: conv1d_h + fused ResBlock pairs; snake: conv1d_h + snake_alias_h); stage widths
        that are not multiples of 16 (the tiny template's 200/100/50/25/12) raise NotImplementedError."""
        if not hasattr(self.dec, "set_half"):
            raise NotImplementedError(f"half-precision inference is not built for the {type(self.dec).__module__} generator")
        self.dec.set_half(True)
        if hasattr(self.flow, "set_half"):
            self.flow.set_half(True)                 # the flow's couplings as fused fp16 kernels (csrc/flow_fused.hip) where built
        self._graphs.clear()
        return self

    def float(self):
        if hasattr(self.dec, "set_half"):
            self.dec.set_half(False)
        if hasattr(self.flow, "set_half"):
            self.flow.set_half(False)
        self._graphs.clear()
        return super().float()

    def split_f16(self, on=True):
        """fp32 inference with the generator's convolutions on the fp16 matrix pipe at fp32-level precision: every activation and
        weight of the generator as a hi and a lo fp16 plane (22 mantissa bits), every product as three fp16 matrix instructions with
        fp32 accumulation (csrc/conv1d_hl.hip; Generator.set_half(split=True); both generators) — gfx950 multiplies fp16 sixteen times faster than
        fp32.  Same inputs, outputs and parameters as the fp32 mode; `split_f16(False)` / `float()` switch back."""
        if not hasattr(self.dec, "set_half"):
            raise NotImplementedError(f"the split pipeline is not built for the {type(self.dec).__module__} generator")
        self.dec.set_half(bool(on), split=True)
        if hasattr(self.flow, "set_half"):
            self.flow.set_half(bool(on), split=True)
        self._graphs.clear()
        return self

    def split_range_exceeded(self, clear=True):
        """After infer() in split mode: True if a value of the generator left the range two fp16 pieces can carry (|v| > 65504, or a
        nan) — the waveform is then NOT fp32-level; call split_f16(False) and infer again (inference.infer_tool.Svc.infer does).  Weights
        of any magnitude are covered by the packs' power-of-two scales; this is about activations.  Reads one device word."""
        fn = getattr(self.dec, "split_range_exceeded", None)
        return bool(fn(clear)) if fn is not None else False

    def EnableCharacterMix(self, n_speakers_map, device):
        self.speaker_map = torch.zeros((n_speakers_map, 1, 1, self.gin_channels)).to(device)
        for i in range(n_speakers_map):
            self.speaker_map[i] = self.emb_g(torch.LongTensor([[i]]).to(device))
        self.speaker_map = self.speaker_map.unsqueeze(0).to(device)
        self.character_mix = True

    def forward(self, c, f0, uv, spec, g=None, c_lengths=None, spec_lengths=None, vol=None, noise=None):
        """Training graph, reference models.py:463-493.  Every op is a svc_autograd Function (HIP forward + backward).
        `noise` (optional dict: enc_p, enc_q [B,inter,T], f0_factor [B,1], ids_slice [B], rand_ini [B,9],
        sine [B, seg*hop, 9], dropout_u: list of uniform draws, one per active nn.Dropout site in call order — f0_decoder's
        6 layers x (attention probabilities [B,H,T,T], attention output, FFN hidden, FFN output), then enc_p's, then the
        transformer flow's) injects the
        random draws explicitly (parity tests); otherwise they come from torch's generator in the reference's order."""
        noise = noise or {}
        du = list(noise["dropout_u"]) if noise.get("dropout_u") is not None else None
        c, f0, uv, spec = c.float(), f0.float(), uv.float(), spec.float()
        B, _, T = c.shape
        gi = g if g.dim() == 2 else g.view(B, 1)
        gemb = A.embedding_bct(gi.long(), self.emb_g.weight)                            # [B, gin, 1]
        x_mask = torch.unsqueeze(commons.sequence_mask(c_lengths, T), 1).to(c.dtype)
        x = A.mul_bcast(self.pre.forward_train(c), x_mask)
        x = A.add(x, A.embedding_bct(uv.long(), self.emb_uv.weight))                    # :471
        if vol is not None and self.vol_embedding:                                      # :469 emb_vol(vol[:,:,None])
            H = self.hidden_channels
            vt = A.mul_bcast(vol.float().unsqueeze(1).expand(B, H, T), self.emb_vol.weight.view(1, H, 1))
            x = A.add(x, A.add_bcast(vt, self.emb_vol.bias.view(1, H, 1)))
        if self.use_automatic_f0_prediction:
            factor = noise.get("f0_factor")
            if factor is None:
                if commons.DEVICE_RNG:
                    factor = torch.empty(B, 1, device=c.device).uniform_(0.8, 1.2)
                else:
                    factor = torch.empty(B, 1).uniform_(0.8, 1.2).to(c.device)          # utils.py:39 (CPU draw)
            lf0, norm_lf0 = S.f0_norm_lf0(f0, uv, mask=x_mask, factor=factor.reshape(-1))   # :474-475 (no grad: inputs)
            pred_lf0 = self.f0_decoder(x, norm_lf0, x_mask, spk_emb=gemb, dropout_u=du)
        else:
            lf0 = norm_lf0 = pred_lf0 = 0
        z_ptemp, m_p, logs_p, _ = self.enc_p(x, x_mask, f0=utils.f0_to_coarse(f0), noise=noise.get("enc_p"), dropout_u=du)
        z, m_q, logs_q, spec_mask = self.enc_q(spec, spec_lengths, g=gemb, noise=noise.get("enc_q"))
        z_p = self.flow(z, spec_mask, g=gemb, dropout_u=du) if self.use_transformer_flow else self.flow(z, spec_mask, g=gemb)
        ids = noise.get("ids_slice")
        if ids is None:
            z_slice, pitch_slice, ids_slice = commons.rand_slice_segments_with_pitch(z, f0, spec_lengths, self.segment_size)
        else:
            ids_slice = ids.to(c.device)
            z_slice = commons.slice_segments(z, ids_slice, self.segment_size)
            pitch_slice = commons.slice_pitch_segments(f0, ids_slice, self.segment_size)
        dn = dict(rand_ini=noise["rand_ini"], sine=noise["sine"]) if "sine" in noise else None
        o = self.dec(z_slice, pitch_slice.contiguous(), g=gemb, noise=dn)
        return o, ids_slice, spec_mask, (z, z_p, m_p, logs_p, m_q, logs_q), pred_lf0, norm_lf0, lf0

    # ------------------------------------------------------------------------------------------------------
    def enable_graph(self, on=True):
        """Replay the whole infer path from a hipGraph (captured per input shape) instead of ~230 eager launches."""
        self.use_graph = bool(on)
        if not on:
            self._graphs.clear()
        return self

    def _speaker(self, g, c):
        if self.character_mix and len(g) > 1:    # [N, S] * [S, B, 1, H]  (reference models.py:505-509)
            g = g.reshape((g.shape[0], g.shape[1], 1, 1, 1))
            g = g * self.speaker_map
            g = torch.sum(g, dim=1)
            g = g.transpose(0, -1).transpose(0, -2).squeeze(0)        # [B, H, N]
            return g.contiguous()
        if g.dim() == 1:
            g = g.unsqueeze(0)
        return self.emb_g(g).transpose(1, 2).contiguous()            # [B, H, 1]

    def _infer_body(self, c, f0, uv, g, noise, noice_scale, predict_f0, vol, lengths=None):
        """The device work of infer(): every line is one or a few HIP kernels (no torch arithmetic)."""
        B, _, T = c.shape
        # (tried: for T % 4 != 0 — 862 frames = 10 s — run the encoder / flow section on T rounded up to 4 with the extra
        # frames masked off, so that every row is 16-byte aligned and the convs take their float4 / LDS-DMA paths: exact,
        # but same-box A/B 10.46/10.57 ms without vs 10.72/10.62 ms with — those T=862 launches are latency-bound, not
        # staging-bound, and the padding mask costs the attention kernel 6 %.)
        if lengths is None:
            x_mask = torch.ones((B, 1, T), device=c.device, dtype=torch.float32)      # c_lengths == T (models.py:503)
        else:   # extension (Svc.slice_inference(batch_chunks=True)): items of different frame counts padded to T; the mask is
            #     commons.sequence_mask(c_lengths, T) (models.py:504) of the TRUE lengths, built on the device (graph-capturable)
            x_mask = (torch.arange(T, device=c.device).view(1, 1, T) < lengths.view(B, 1, 1)).to(torch.float32)
        m = mask2d(x_mask)
        src_noise = noise if "sine" in noise else None
        early = not (self.use_automatic_f0_prediction and predict_f0) and hasattr(self.dec, "start_source")
        xin = self.pre.run(c, mask=m)                                               # pre(c) * mask
        # The decoder's harmonic source + noise convs only need f0, the flow's conditioning rows only g: both start here, on side
        # streams underneath the encoder and the flow.  AFTER the first launch of the main chain, not before it: a replayed graph
        # releases the successor it put on another queue only behind the first launch of the one it kept on the same queue — with the
        # 55 us single-workgroup frame scan in that place the whole front section started 65 us late (trace: profiles/r10j_*).
        source = self.dec.start_source(f0, src_noise) if early else None
        cond = self.flow.start_cond(g) if hasattr(self.flow, "start_cond") else None
        volv = vol if (vol is not None and self.vol_embedding) else None
        x, x_enc = S.prenet_embed(xin, uv, f0, self.emb_uv.weight, self.enc_p.f0_emb.weight, mask=m, vol=volv,
                                  vol_w=self.emb_vol.weight.view(-1) if volv is not None else None,
                                  vol_b=self.emb_vol.bias if volv is not None else None)
        if self.use_automatic_f0_prediction and predict_f0:
            lf0, norm_lf0 = S.f0_norm_lf0(f0, uv, mask=m)                 # models.py:524-525, utils.py:31-45
            pred_lf0 = self.f0_decoder(x, norm_lf0, x_mask, spk_emb=g)
            f0 = S.lf0_to_f0(pred_lf0).squeeze(1)                         # models.py:527
            _, x_enc = S.prenet_embed(xin, uv, f0, self.emb_uv.weight, self.enc_p.f0_emb.weight, mask=m, vol=volv,
                                      vol_w=self.emb_vol.weight.view(-1) if volv is not None else None,
                                      vol_b=self.emb_vol.bias if volv is not None else None)
        z_p, m_p, logs_p, _ = self.enc_p(x_enc, x_mask, noice_scale=noice_scale, noise=noise.get("enc_p"),
                                         x_is_embedded=True, full_mask=lengths is None)
        # split mode: the flow's fused couplings report into the generator's range flag too (include/svc_hip.h, RANGE)
        guard = self.dec._range_guard(c.device) if getattr(self.flow, "fused_mode", False) == "split" and hasattr(self.dec, "_range_guard") \
            else contextlib.nullcontext()
        with guard:
            z = self.flow(z_p, x_mask, g=g, reverse=True, cond=cond) if cond is not None else self.flow(z_p, x_mask, g=g, reverse=True)
        # `z * c_mask` (models.py:531) is the identity here: the flow's last update already multiplies by the mask
        if source is not None:
            o = self.dec(z, f0, g=g, noise=src_noise, source=source)
        else:
            o = self.dec(z, f0, g=g, noise=src_noise)
        return o, f0

    @torch.no_grad()
    def infer(self, c, f0, uv, g=None, noice_scale=0.35, seed=52468, predict_f0=False, vol=None, noise=None, lengths=None):
        """Reference models.py:495-532.  `noise` (optional dict enc_p/rand_ini/sine) injects the RNG draws
        explicitly (parity tests); otherwise they are drawn from torch's generator for c.device in the reference's
        order after seeding with `seed`.  `lengths` (extension, [B] frame counts on the device): the batch holds items of
        different lengths zero-padded to T; masks come from the true lengths (the reference always passes c.size(-1))."""
        if not c.is_cuda:
            raise S.SvcError("SynthesizerTrn.infer needs CUDA/ROCm tensors: the MI355X engine has no CPU fallback")
        c = c.float().contiguous()
        f0 = f0.float().contiguous()
        uv = uv.float().contiguous()
        torch.manual_seed(seed)      # seeds every device generator (reference :498-501)
        g = self._speaker(g, c)
        B, _, T = c.shape
        if lengths is not None:
            lengths = lengths.to(device=c.device, dtype=torch.int64).contiguous()
        if noise is None and self.use_graph:
            # the draws go straight into the captured graph's input buffers (same generator, same order, same kernels as the
            # torch.randn / torch.rand below: bit-identical values; saves a 16 MB copy of the sine noise in front of every replay)
            return self._infer_graph(c, f0, uv, g, None, noice_scale, predict_f0, vol, lengths)
        if noise is None:
            noise = self._draw_noise(B, T, c.device)
        if self.use_graph:
            return self._infer_graph(c, f0, uv, g, noise, noice_scale, predict_f0, vol, lengths)
        return self._infer_body(c, f0, uv, g, noise, noice_scale, predict_f0, vol, lengths)

    # ------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def infer_many(self, items, noice_scale=0.35, seed=52468, predict_f0=False):
        """Engine extension: several INDEPENDENT infer() calls — `items` = [(c, f0, uv, g[, vol]), ...], each what one infer() call
        takes (the chunks of Svc.slice_inference, requests of a serving queue) — replayed as PARALLEL BRANCHES of one hipGraph.
        One clip alone leaves most CUs idle through its encoder + flow section (~100 latency-bound launches on T columns, a
        quarter of a 10 s clip's time for 6 % of its FLOPs) and replays of different graphs do not overlap on this runtime; as
        branches of one graph the clips' kernels interleave (two 10 s clips: 13.0 ms against 2 x 7.4, profiles/r08h_*).  Every
        item is computed exactly as its own infer() call computes it — same seed, same draws (the reference re-seeds per call,
        models.py:498-501), same kernels — and the outputs are bit-identical to it.  Needs enable_graph(True); returns
        [(audio, f0), ...] in the order of `items`."""
        if not self.use_graph:
            return [self.infer(*it[:3], g=it[3], noice_scale=noice_scale, seed=seed, predict_f0=predict_f0,
                               vol=it[4] if len(it) > 4 else None) for it in items]
        prepared = []
        for it in items:
            c, f0, uv, g = it[:4]
            vol = it[4] if len(it) > 4 else None
            if not c.is_cuda:
                raise S.SvcError("SynthesizerTrn.infer needs CUDA/ROCm tensors: the MI355X engine has no CPU fallback")
            c, f0, uv = c.float().contiguous(), f0.float().contiguous(), uv.float().contiguous()
            torch.manual_seed(seed)
            g = self._speaker(g, c)
            B, _, T = c.shape
            noise = dict(enc_p=torch.randn(B, self.inter_channels, T, device=c.device), rand_ini=torch.rand(B, 9, device=c.device),
                         sine=torch.randn(B, T * self.dec.upp, 9, device=c.device))
            ins = dict(c=c, f0=f0, uv=uv, g=g, enc_p=noise["enc_p"], rand_ini=noise["rand_ini"], sine=noise["sine"])
            if vol is not None:
                ins["vol"] = vol.float().contiguous()
            prepared.append(ins)
        key = ("many", tuple((tuple(p["c"].shape), tuple(p["g"].shape), "vol" in p) for p in prepared), float(noice_scale),
               bool(predict_f0), str(prepared[0]["c"].device))
        ent = self._graphs.get(key)
        if ent is None:
            import vdecoder.hifigan.models as _gen
            static = [{k: v.clone() for k, v in p.items()} for p in prepared]

            def run(i):
                st = static[i]
                return self._infer_body(st["c"], st["f0"], st["uv"], st["g"], dict(enc_p=st["enc_p"], rand_ini=st["rand_ini"], sine=st["sine"]),
                                        noice_scale, predict_f0, st.get("vol"))
            warm = torch.cuda.Stream()
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm):
                for i in range(len(static)):
                    run(i)
            torch.cuda.current_stream().wait_stream(warm)
            sides = [torch.cuda.Stream() for _ in static[1:]]
            graph = torch.cuda.CUDAGraph()
            outs = [None] * len(static)
            with S.graph_capture(graph):
                main = torch.cuda.current_stream()
                fork = torch.cuda.Event()
                fork.record(main)
                dones = []
                # the branches on side streams run their launches on ONE stream each: a fork inside a forked branch (the three MRF
                # chains, the early harmonic source) crashes hipStreamEndCapture on ROCm 7.2 (segmentation fault in capture_end,
                # reproduced by scripts/bench_pipeline_onegraph.py); the clip on the capturing stream keeps its inner streams
                was = _gen._MRF_STREAMS
                try:
                    _gen._MRF_STREAMS = False
                    for i, st in enumerate(sides, start=1):
                        with torch.cuda.stream(st):
                            st.wait_event(fork)
                            outs[i] = run(i)
                            ev = torch.cuda.Event()
                            ev.record(st)
                            dones.append(ev)
                    if os.environ.get("SVC_MANY_MAIN_STREAMS", "0") != "1":      # A/B: inner streams for the capturing stream's clip
                        outs[0] = run(0)
                finally:
                    _gen._MRF_STREAMS = was
                if outs[0] is None:
                    outs[0] = run(0)
                for ev in dones:
                    main.wait_event(ev)
            ent = (graph, static, outs)
            self._graphs[key] = ent
        graph, static, outs = ent
        dst, src = [], []
        for st, p in zip(static, prepared):
            for k, v in p.items():
                if v.dtype == torch.float32 and v.shape == st[k].shape:
                    dst.append(st[k])
                    src.append(v)
                else:
                    st[k].copy_(v, non_blocking=True)
        torch._foreach_copy_(dst, src)
        graph.replay()
        return [(o[0].clone(), o[1].clone()) for o in outs]

    def _draw_noise(self, B, T, device):
        """The reference's draws in the reference's order (models.py:160 randn_like; vdecoder/hifigan/models.py:147 rand, :266 randn)."""
        return dict(enc_p=torch.randn(B, self.inter_channels, T, device=device), rand_ini=torch.rand(B, 9, device=device),
                    sine=torch.randn(B, T * self.dec.upp, 9, device=device))

    def _infer_graph(self, c, f0, uv, g, noise, noice_scale, predict_f0, vol, lengths=None):
        key = (tuple(c.shape), tuple(g.shape), float(noice_scale), bool(predict_f0), vol is not None, lengths is not None,
               str(c.device))
        ent = self._graphs.get(key)
        in_place = noise is None and ent is not None
        if in_place:                     # torch.randn(shape) IS empty(shape).normal_(): the same values, written where the graph reads them
            st = ent[1]
            st["enc_p"].normal_()
            st["rand_ini"].uniform_()
            st["sine"].normal_()
            noise = {}
        elif noise is None:
            noise = self._draw_noise(c.shape[0], c.shape[2], c.device)
        ins = dict(c=c, f0=f0, uv=uv, g=g, **{k: noise[k] for k in ("enc_p", "rand_ini", "sine") if k in noise})
        if vol is not None:
            ins["vol"] = vol.float().contiguous()
        if lengths is not None:
            ins["lengths"] = lengths
        if ent is None:
            static = {k: v.clone() for k, v in ins.items()}
            run = lambda: self._infer_body(static["c"], static["f0"], static["uv"], static["g"],
                                           dict(enc_p=static["enc_p"], rand_ini=static["rand_ini"],
                                                sine=static["sine"]), noice_scale, predict_f0, static.get("vol"),
                                           static.get("lengths"))
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                run()                                   # warm-up: packs weights, sets kernel attributes
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with S.graph_capture(graph):
                out = run()
            ent = (graph, static, out)
            self._graphs[key] = ent
        graph, static, out = ent
        # inputs -> the graph's static buffers: ONE multi-tensor launch for the same-dtype device tensors (8 inputs: 8 x ~5 us
        # of back-to-back copy kernels in front of every replay otherwise); anything else (host tensors, int64 lengths) one by one
        same = [k for k, v in ins.items() if v.is_cuda and v.dtype == torch.float32 and v.shape == static[k].shape]
        if len(same) > 1:
            torch._foreach_copy_([static[k] for k in same], [ins[k] for k in same])
        for k, v in ins.items():
            if len(same) <= 1 or k not in same:
                static[k].copy_(v, non_blocking=True)
        graph.replay()
        return out[0].clone(), out[1].clone()


_DISCP_PAD_ROWS = os.environ.get("SVC_DISCP_PAD", "1") != "0"


class _SpecConv(nn.Module):
    """spectral_norm(Conv1d / Conv2d((k,1))) parameter holder (`use_spectral_norm=True`, models.py:170,205): `weight_orig`,
    `bias` and the `weight_u` / `weight_v` buffers under torch.nn.utils.spectral_norm's names and shapes."""

    def __init__(self, cin, cout, k, stride, padding, groups=1, conv2d=False):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.padding, self.groups, self.conv2d = cin, cout, k, stride, padding, groups, conv2d
        shape = (cout, cin // groups, k, 1) if conv2d else (cout, cin // groups, k)
        w = torch.empty(*shape)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1 / math.sqrt((cin // groups) * k)
        self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        self.weight_orig = nn.Parameter(w)
        self.register_buffer("weight_u", torch.nn.functional.normalize(torch.randn(cout), dim=0, eps=1e-12))
        self.register_buffer("weight_v", torch.nn.functional.normalize(torch.randn((cin // groups) * k), dim=0, eps=1e-12))

    def forward(self, x, inner=1, lp=None, out_blocks=None):
        # one power iteration per forward call in training mode, in place on the buffers (torch's forward pre-hook)
        w = A.spectral_norm(self.weight_orig, self.weight_u, self.weight_v, self.training)
        w = w.view(self.cout, self.cin // self.groups, self.k)
        return A.conv1d(x, w, self.bias, self.stride, self.padding, 1, self.groups, inner=inner, lp=lp, out_blocks=out_blocks)


class _NormConv(nn.Module):
    """weight_norm(Conv1d / Conv2d((k,1))) parameter holder of the discriminators; `weight_v` keeps the REFERENCE shape
    ([Cout,Cin,K] for Conv1d, [Cout,Cin,K,1] for Conv2d) so checkpoints (D_*.pth) load key-for-key."""

    def __init__(self, cin, cout, k, stride, padding, groups=1, conv2d=False):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.padding, self.groups, self.conv2d = cin, cout, k, stride, padding, groups, conv2d
        shape = (cout, cin // groups, k, 1) if conv2d else (cout, cin // groups, k)
        w = torch.empty(*shape)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1 / math.sqrt((cin // groups) * k)
        self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        self.weight_g = nn.Parameter(w.flatten(1).norm(dim=1).view(-1, *([1] * (w.dim() - 1))).clone())
        self.weight_v = nn.Parameter(w)

    def forward(self, x, inner=1, lp=None, out_blocks=None):
        if svc_nn.WEIGHT_PLANS and self.groups == 1:
            plan = self.__dict__.get("_svc_plan")
            if plan is None:
                plan = A.conv_plan((self.cout, self.cin, self.k), self.stride, self.padding)
                self.__dict__["_svc_plan"] = plan
            return A.conv1d_planned(x, plan, self.weight_v, self.weight_g, self.bias, self.stride, self.padding, 1,
                                    inner=inner, lp=lp, out_blocks=out_blocks)
        w = A.weight_norm(self.weight_v, self.weight_g).view(self.cout, self.cin // self.groups, self.k)
        return A.conv1d(x, w, self.bias, self.stride, self.padding, 1, self.groups, inner=inner, lp=lp, out_blocks=out_blocks)


def _padded_map_view(h, b, Hp, H, p):
    """[b, C, Hp*p] padded feature map -> the reference's [b, C, H, p] map as a view; `_svc_padded` keeps the padded buffer
    (zero tail) for consumers that only reduce over the map (modules.losses.feature_loss): same sums, no gather copy."""
    v = h.view(b, h.shape[1], Hp, p)
    if Hp != H:
        v = v[:, :, :H]
        v._svc_padded = h
    return v


def _split_map(f, n):
    """Real / generated halves of a feature map computed on cat([y, y_hat]) (keeps the padded-buffer link)."""
    r, g = f[:n], f[n:]
    full = getattr(f, "_svc_padded", None)
    if full is not None:
        r._svc_padded, g._svc_padded = full[:n], full[n:]
    return r, g


class DiscriminatorP(nn.Module):
    """Reference models.py:165-199.  The Conv2d((k,1),(s,1)) stack acts on each of the `period` columns of the
    [B,1,T/p,p] view independently.  The view's memory order IS the time order, so the signal stays [B,C,H*p] with
    rows = blocks of p consecutive samples: a stride-1 (k,1) conv is a dense Conv1d with dilation p, a stride-3 one a
    block decimation (svc_decimate_f32, w = p, which also folds in the reflect padding of :185-189) followed by a
    dense dilation-p Conv1d.  Every layer therefore sees T = H*p >= ~100 columns per batch row on the MFMA N axis
    (turning the columns into batch rows instead leaves 9-30 columns per row in the 1024-channel layers), and the
    feature maps come out in the reference's [B,C,H,p] layout as plain views.

    Row alignment: H*p is rarely a multiple of 4 floats (segment 8192, period 3: 2731, 911, 304, 102 blocks), and rows that
    do not start on 16-byte boundaries push the 1024-channel convolutions — forward, dgrad and wgrad — onto the
    scalar-staging kernel instantiations (2.5-3.6x the time of the float4 / LDS-DMA ones on the same shape, 23 ms of a 144 ms
    iteration, profiles/r02_final_train_B16_kernel_stats.txt).  Every map is therefore stored with H rounded up to H' so that
    H'*p % 4 == 0, the H' - H tail blocks zero: the convolutions run over the physical rows (zeros past the logical end ARE
    the convolution's implicit padding), `leaky_relu_tail` restores the zero tail after each of them in the forward and in
    the backward direction, and the returned feature maps are the [:, :, :H] views of the padded buffers."""

    def __init__(self, period, kernel_size=5, stride=3, use_spectral_norm=False):
        super().__init__()
        self.period = period
        self.use_spectral_norm = use_spectral_norm
        conv = _SpecConv if use_spectral_norm else _NormConv
        pad = commons.get_padding(kernel_size, 1)
        chans = [(1, 32), (32, 128), (128, 512), (512, 1024)]
        self.convs = nn.ModuleList([conv(a, b, kernel_size, stride, pad, conv2d=True) for a, b in chans] +
                                   [conv(1024, 1024, kernel_size, 1, pad, conv2d=True)])
        self.conv_post = conv(1024, 1, 3, 1, 1, conv2d=True)

    def forward(self, x):
        b, c, t = x.shape
        p = self.period
        n_pad = (p - t % p) % p
        fmap = []
        h = x
        lp = t + n_pad
        H = lp // p                                             # logical number of blocks (rows of the [T/p, p] view)
        for l in list(self.convs) + [self.conv_post]:
            H = (H + 2 * l.padding - l.k) // l.stride + 1
            Hp = A.align_blocks(H, p) if _DISCP_PAD_ROWS else H
            if not _DISCP_PAD_ROWS:         # SVC_DISCP_PAD=0: the unpadded layout (A/B switch)
                y = l(h, inner=p, lp=lp)
                h = A.leaky_relu(y, modules.LRELU_SLOPE) if l is not self.conv_post else y
                lp = None
                fmap.append(h.view(b, h.shape[1], -1, p))
                continue
            y = l(h, inner=p, lp=lp, out_blocks=Hp)             # [b, C, Hp*p]; blocks >= H are don't-care
            # models.py:190-193 lrelu(0.1) after every conv but conv_post (slope 1: tail mask only)
            h = A.leaky_relu_tail(y, modules.LRELU_SLOPE if l is not self.conv_post else 1.0, H * p)
            lp = None
            fmap.append(_padded_map_view(h, b, Hp, H, p))
        return torch.flatten(fmap[-1], 1, -1), fmap


class DiscriminatorS(nn.Module):
    """Reference models.py:202-227."""

    def __init__(self, use_spectral_norm=False):
        super().__init__()
        conv = _SpecConv if use_spectral_norm else _NormConv
        self.convs = nn.ModuleList([
            conv(1, 16, 15, 1, 7), conv(16, 64, 41, 4, 20, groups=4), conv(64, 256, 41, 4, 20, groups=16),
            conv(256, 1024, 41, 4, 20, groups=64), conv(1024, 1024, 41, 4, 20, groups=256),
            conv(1024, 1024, 5, 1, 2)])
        self.conv_post = conv(1024, 1, 3, 1, 1)

    def forward(self, x):
        fmap = []
        for l in self.convs:
            x = A.leaky_relu(l(x), modules.LRELU_SLOPE)
            fmap.append(x)
        x = self.conv_post(x)
        fmap.append(x)
        return torch.flatten(x, 1, -1), fmap


class MultiPeriodDiscriminator(nn.Module):
    """Reference models.py:230-252."""

    def __init__(self, use_spectral_norm=False):
        super().__init__()
        periods = [2, 3, 5, 7, 11]
        self.use_spectral_norm = use_spectral_norm
        self.discriminators = nn.ModuleList([DiscriminatorS(use_spectral_norm=use_spectral_norm)] +
                                            [DiscriminatorP(i, use_spectral_norm=use_spectral_norm) for i in periods])

    def forward(self, y, y_hat):
        y_d_rs, y_d_gs, fmap_rs, fmap_gs = [], [], [], []
        if self.use_spectral_norm:
            # every forward call of a spectrally normalised layer runs a power iteration on its u / v buffers: the reference
            # calls each discriminator on y and then on y_hat (models.py:246-247), i.e. TWO iterations per layer, the second
            # pass seeing the vectors the first one left — one batched pass would not reproduce that
            for d in self.discriminators:
                out_r, fmap_r = d(y)
                out_g, fmap_g = d(y_hat)
                y_d_rs.append(out_r)
                y_d_gs.append(out_g)
                fmap_rs.append(fmap_r)
                fmap_gs.append(fmap_g)
            return y_d_rs, y_d_gs, fmap_rs, fmap_gs
        n = y.shape[0]
        yy = torch.cat([y, y_hat], 0)      # one pass over both signals (same weights): halves the launches
        # (tried: one HIP stream per sub-discriminator — forward and, through autograd's stream bookkeeping, backward launches
        # of the six independent networks overlapping: 150 vs 140 ms per iteration on the same box, the extra event traffic costs
        # more than the overlap of these launch-bound kernels gains inside an already captured graph)
        for out, fmap in (d(yy) for d in self.discriminators):
            out_r, out_g = A.split_batch(out, n)        # (one cat in the backward instead of two zero-filled slice gradients)
            y_d_rs.append(out_r)
            y_d_gs.append(out_g)
            halves = [_split_map(f, n) for f in fmap]
            fmap_rs.append([r for r, _ in halves])
            fmap_gs.append([g for _, g in halves])
        return y_d_rs, y_d_gs, fmap_rs, fmap_gs

    def forward_gen_step(self, y, y_hat):
        """The discriminator pass of the GENERATOR step (train.py:200): same return value as forward(), but the real
        branch — whose logits are unused and whose feature maps `feature_loss` detaches (modules/losses.py:8) — runs
        without an autograd tape, so the generator step back-propagates through half the batch only."""
        y_d_rs, y_d_gs, fmap_rs, fmap_gs = [], [], [], []

        for d in self.discriminators:
            with torch.no_grad():
                out_r, fmap_r = d(y)
            out_g, fmap_g = d(y_hat)
            y_d_rs.append(out_r)
            y_d_gs.append(out_g)
            fmap_rs.append(fmap_r)
            fmap_gs.append(fmap_g)
        return y_d_rs, y_d_gs, fmap_rs, fmap_gs
