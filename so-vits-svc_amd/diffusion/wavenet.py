"""MI355X-native mirror of diffusion/wavenet.py: the WaveNet denoiser of the shallow-diffusion model.

Same classes / `state_dict` keys as the reference (torch modules below are parameter containers only); inference forward
on libsvc_hip.so.  What the engine does differently from the reference's per-call, per-layer op list (:31-108):

  * conditioner_projection of ALL layers is one stacked 1x1 MFMA conv, computed once per `cond` tensor (the sampler calls
    the denoiser 10..1000 times with the same cond) — :47,51
  * the step embedding path (SinusoidalPosEmb -> Linear -> Mish -> Linear, :91-93) and every layer's
    diffusion_projection (:46,50) are three tiny 1x1 convs per call (the L projections stacked into one)
  * per residual layer (:49-65): broadcast add of the step projection, then dilated conv + conditioner + gate in ONE fused
    MFMA conv (WN gate epilogue; the reference's sigmoid(first half) * tanh(second half) is obtained by swapping the weight
    halves at pack time), then output_projection with the residual / skip epilogue ((x + r)/sqrt2 as a constant "mask",
    skip accumulated in place) — 3 launches instead of ~12
  * sum(skip)/sqrt(L) (:100) is folded into the packed skip_projection weight.
"""
import math
from math import sqrt

import torch
import torch.nn as nn

import svc_autograd as A
import svc_hip as S


class Conv1d(torch.nn.Conv1d):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        nn.init.kaiming_normal_(self.weight)


class Mish(nn.Module):
    def forward(self, x):
        raise NotImplementedError("Mish runs as svc_ew_f32(SVC_EW_MISH) inside WaveNet.forward")


class SinusoidalPosEmb(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        return S.sinusoidal_emb(x.reshape(-1), self.dim)


class ResidualBlock(nn.Module):
    def __init__(self, encoder_hidden, residual_channels, dilation):
        super().__init__()
        self.residual_channels = residual_channels
        self.dilation = dilation
        self.dilated_conv = nn.Conv1d(residual_channels, 2 * residual_channels, kernel_size=3, padding=dilation,
                                      dilation=dilation)
        self.diffusion_projection = nn.Linear(residual_channels, residual_channels)
        self.conditioner_projection = nn.Conv1d(encoder_hidden, 2 * residual_channels, 1)
        self.output_projection = nn.Conv1d(residual_channels, 2 * residual_channels, 1)


def _swap_halves(t):
    """Rows [gate ; filter] -> [filter ; gate]: the fused gate epilogue computes tanh(first) * sigmoid(second)."""
    c = t.shape[0] // 2
    return torch.cat([t[c:], t[:c]], 0)


class WaveNet(nn.Module):
    def __init__(self, in_dims=128, n_layers=20, n_chans=384, n_hidden=256):
        super().__init__()
        self.in_dims, self.n_layers, self.n_chans, self.n_hidden = in_dims, n_layers, n_chans, n_hidden
        self.input_projection = Conv1d(in_dims, n_chans, 1)
        self.diffusion_embedding = SinusoidalPosEmb(n_chans)
        self.mlp = nn.Sequential(nn.Linear(n_chans, n_chans * 4), Mish(), nn.Linear(n_chans * 4, n_chans))
        self.residual_layers = nn.ModuleList([ResidualBlock(encoder_hidden=n_hidden, residual_channels=n_chans, dilation=1)
                                              for _ in range(n_layers)])
        self.skip_projection = Conv1d(n_chans, n_chans, 1)
        self.output_projection = Conv1d(n_chans, in_dims, 1)
        nn.init.zeros_(self.output_projection.weight)
        self._cache = {}
        self._cond_cache = None
        self.pack_batches = True        # training: run batches of short crops as one packed row (forward_train)

    # -- packed weights (rebuilt when any parameter changes) --------------------------------------------------------
    def _packs(self):
        key = (sum(p._version for p in self.parameters()), str(self.input_projection.weight.device))
        if self._cache.get("key") != key:
            C, L = self.n_chans, self.n_layers
            pk = lambda w: S.pack_conv1d_weight(w.detach().contiguous())
            d = dict(key=key)
            d["w_in"] = pk(self.input_projection.weight)
            d["w_m0"] = pk(self.mlp[0].weight.unsqueeze(-1))
            d["w_m2"] = pk(self.mlp[2].weight.unsqueeze(-1))
            d["w_dp"] = pk(torch.cat([l.diffusion_projection.weight for l in self.residual_layers], 0).unsqueeze(-1))
            d["b_dp"] = torch.cat([l.diffusion_projection.bias for l in self.residual_layers], 0).detach().contiguous()
            d["w_cp"] = pk(torch.cat([_swap_halves(l.conditioner_projection.weight) for l in self.residual_layers], 0))
            d["b_cp"] = torch.cat([_swap_halves(l.conditioner_projection.bias) for l in self.residual_layers], 0).detach().contiguous()
            d["w_dc"] = [S.pack_conv1d_weight(_swap_halves(l.dilated_conv.weight).detach().contiguous(), None, C)
                         for l in self.residual_layers]
            d["b_dc"] = [_swap_halves(l.dilated_conv.bias).detach().contiguous() for l in self.residual_layers]
            d["w_op"] = [pk(l.output_projection.weight) for l in self.residual_layers]
            d["w_sk"] = pk(self.skip_projection.weight / sqrt(L))          # sum(skip) / sqrt(L) folded in
            d["w_out"] = pk(self.output_projection.weight)
            self._cache = d
            self._cond_cache = None
        return self._cache

    def _cond_proj(self, cond, pk):
        key = (cond.data_ptr(), cond._version, tuple(cond.shape))
        if self._cond_cache is None or self._cond_cache[0] != key:
            cp = S.conv1d(cond.float().contiguous(), pk["w_cp"], 2 * self.n_chans * self.n_layers, 1, bias=pk["b_cp"])
            self._cond_cache = (key, cp, cond)          # keep `cond` alive so that the pointer key stays unique
        return self._cond_cache[1]

    def forward_train(self, spec, diffusion_step, cond):
        """The same map on the autograd ops of svc_autograd.py (HIP forward + backward), for p_losses / train_diff.py.
        All layers' conditioner projections are one stacked conv (one dgrad into `cond`, one wgrad), split into
        contiguous per-layer chunks; the reference's sigmoid(first half) * tanh(second half) is tanh(first) * sigmoid(second)
        on weights whose halves are swapped (pure index reshapes of the parameters)."""
        C, L = self.n_chans, self.n_layers
        B, _, M, T = spec.shape
        layers = self.residual_layers
        gap = max(l.dilation for l in layers)                   # halo of the (only) K=3 conv
        packed = self.pack_batches and B > 1 and T % 128 != 0                         # batches of short crops: run on the packed row (see A.pack_items)
        xin = spec.reshape(B, M, T).float().contiguous()
        if packed:
            xin, cond = A.pack_items(xin, gap), A.pack_items(cond, gap)
        x = A.relu(A.conv1d(xin, self.input_projection.weight, self.input_projection.bias))
        emb = self.diffusion_embedding(diffusion_step.float()).view(B, C, 1)
        h = A.mish(A.conv1d(emb, self.mlp[0].weight.unsqueeze(-1), self.mlp[0].bias))
        step = A.conv1d(h, self.mlp[2].weight.unsqueeze(-1), self.mlp[2].bias)                       # [B, C, 1]
        w_cp = torch.cat([_swap_halves(l.conditioner_projection.weight) for l in layers], 0)
        b_cp = torch.cat([_swap_halves(l.conditioner_projection.bias) for l in layers], 0)
        cps = A.chunk_channels(A.conv1d(cond, w_cp, b_cp), L)                                        # L x [B, 2C, T]
        w_dp = torch.cat([l.diffusion_projection.weight for l in layers], 0).unsqueeze(-1)
        b_dp = torch.cat([l.diffusion_projection.bias for l in layers], 0)
        dps = A.chunk_channels(A.conv1d(step, w_dp, b_dp), L)                                        # L x [B, C, 1]
        skip = None
        r2 = 1.0 / math.sqrt(2.0)
        for l, layer in enumerate(layers):
            y = A.packed_add_item(x, dps[l], B, T, gap) if packed else A.add_bcast(x, dps[l])
            y = A.conv1d(y, _swap_halves(layer.dilated_conv.weight), _swap_halves(layer.dilated_conv.bias),
                         padding=layer.dilation, dilation=layer.dilation)
            acts = A.gate(A.add(y, cps[l]))
            wo, bo = layer.output_projection.weight, layer.output_projection.bias
            if l < L - 1:               # the last layer's residual half feeds nothing (reference :99-100 uses the skips only)
                x = A.add(x, A.conv1d(acts, wo[:C], bo[:C]), alpha=r2, beta=r2)
            sk = A.conv1d(acts, wo[C:], bo[C:])
            skip = sk if skip is None else A.add(skip, sk)
        h = A.relu(A.conv1d(A.scale(skip, 1.0 / sqrt(L)), self.skip_projection.weight, self.skip_projection.bias))
        out = A.conv1d(h, self.output_projection.weight, self.output_projection.bias)
        if packed:
            out = A.unpack_items(out, B, T, gap)
        return out[:, None, :, :]

    def forward(self, spec, diffusion_step, cond):
        """spec [B,1,M,T], diffusion_step [B] (or [B,1]), cond [B,n_hidden,T] -> [B,1,M,T]  (reference :81-108)."""
        if not spec.is_cuda:
            raise S.SvcError("WaveNet.forward needs CUDA/ROCm tensors: the MI355X engine has no CPU fallback")
        if torch.is_grad_enabled() and (self.training or cond.requires_grad or spec.requires_grad):
            return self.forward_train(spec, diffusion_step, cond)
        with torch.no_grad():
            return self._forward_infer(spec, diffusion_step, cond)

    def _forward_infer(self, spec, diffusion_step, cond):
        pk = self._packs()
        C, L = self.n_chans, self.n_layers
        B, _, M, T = spec.shape
        x = S.conv1d(spec.reshape(B, M, T).float().contiguous(), pk["w_in"], C, 1, bias=self.input_projection.bias,
                     post_act=S.ACT_RELU)
        emb = self.diffusion_embedding(diffusion_step.float()).view(B, C, 1)
        h = S.ew(S.EW_MISH, S.conv1d(emb, pk["w_m0"], 4 * C, 1, bias=self.mlp[0].bias))
        step = S.conv1d(h, pk["w_m2"], C, 1, bias=self.mlp[2].bias)                     # [B, C, 1]
        dproj = S.conv1d(step, pk["w_dp"], L * C, 1, bias=pk["b_dp"])                   # [B, L*C, 1]
        cproj = self._cond_proj(cond, pk)                                                # [B, L*2C, T]
        inv_sqrt2 = torch.full((B, 1, T), 1.0 / math.sqrt(2.0), device=spec.device, dtype=torch.float32)
        skip = torch.empty((B, C, T), device=spec.device, dtype=torch.float32)
        y = torch.empty_like(x)
        xn = torch.empty_like(x)
        for l, layer in enumerate(self.residual_layers):
            S.ew_bct(S.EW_ADD, x, dproj[:, l * C:(l + 1) * C], alpha=1.0, beta=1.0, out=y)      # x + diffusion_step
            acts = S.conv1d(y, pk["w_dc"][l], 2 * C, 3, bias=pk["b_dc"][l], dil=layer.dilation, pad_left=layer.dilation,
                            cond=cproj[:, l * 2 * C:(l + 1) * 2 * C], epi=S.EPI_GATE)
            S.conv1d(acts, pk["w_op"][l], 2 * C, 1, bias=layer.output_projection.bias, epi=S.EPI_RES_SKIP, res=x, out=xn,
                     out2=skip, skip_from=C, mask=inv_sqrt2, beta=1.0 if l > 0 else 0.0)
            x, xn = xn, x
        h = S.conv1d(skip, pk["w_sk"], C, 1, bias=self.skip_projection.bias, post_act=S.ACT_RELU)
        out = S.conv1d(h, pk["w_out"], M, 1, bias=self.output_projection.bias)
        return out[:, None, :, :]
