"""MI355X-native mirror of diffusion/unit2mel.py: Unit2Mel = conditioning embeddings + GaussianDiffusion(WaveNet)
(SURVEY.md §8f row 2).  Same constructor / `state_dict` keys / forward signature; inference (samplers) and training (infer=False -> loss).

The conditioning sum (:139-155) is assembled directly in the engine's [B, H, T] layout: unit_embed = 1x1 MFMA conv,
f0_embed / volume_embed = Cin=1 direct convs (log(1 + f0/700) from the lf0 kernel, rescaled), spk_embed = lookup."""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import yaml

import svc_autograd as A
import svc_hip as S

from .diffusion import GaussianDiffusion
from .wavenet import WaveNet


class DotDict(dict):
    def __getattr__(*args):
        val = dict.get(*args)
        return DotDict(val) if type(val) is dict else val

    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__


class Unit2Mel(nn.Module):
    def __init__(self, input_channel, n_spk, use_pitch_aug=False, out_dims=128, n_layers=20, n_chans=384, n_hidden=256,
                 timesteps=1000, k_step_max=1000):
        super().__init__()
        self.unit_embed = nn.Linear(input_channel, n_hidden)
        self.f0_embed = nn.Linear(1, n_hidden)
        self.volume_embed = nn.Linear(1, n_hidden)
        self.aug_shift_embed = nn.Linear(1, n_hidden, bias=False) if use_pitch_aug else None
        self.n_spk = n_spk
        if n_spk is not None and n_spk > 1:
            self.spk_embed = nn.Embedding(n_spk, n_hidden)
        self.timesteps = timesteps if timesteps is not None else 1000
        self.k_step_max = k_step_max if k_step_max is not None and 0 < k_step_max < self.timesteps else self.timesteps
        self.n_hidden = n_hidden
        self.decoder = GaussianDiffusion(WaveNet(out_dims, n_layers, n_chans, n_hidden), timesteps=self.timesteps,
                                         k_step=self.k_step_max, out_dims=out_dims)
        self.input_channel = input_channel

    def _condition(self, units, f0, volume, spk_id, spk_mix_dict, aug_shift):
        """[B,T,n_unit], [B,T,1], [B,T,1] -> cond [B, n_hidden, T]."""
        B, T, _ = units.shape
        H = self.n_hidden
        pk = lambda w: S.pack_conv1d_weight(w.detach().contiguous())
        x = S.conv1d(units.float().transpose(1, 2).contiguous(), pk(self.unit_embed.weight.unsqueeze(-1)), H, 1,
                     bias=self.unit_embed.bias)
        # log(1 + f0/700) = lf0 * 500 ln(10) / 2595  (svc_f0_norm_lf0_f32 computes lf0 = 2595 log10(1 + f0/700) / 500)
        f0r = f0.float().reshape(B, T).contiguous()
        lf0, _ = S.f0_norm_lf0(f0r, (f0r > 0).float(), mask=None)
        lg = S.ew(S.EW_SCALE, lf0.reshape(B, 1, T).contiguous(), alpha=500.0 * math.log(10.0) / 2595.0)
        x = S.ew(S.EW_ADD, x, S.conv1d_direct(lg, pk(self.f0_embed.weight.unsqueeze(-1)), H, 1, bias=self.f0_embed.bias),
                 alpha=1.0, beta=1.0)
        vol = volume.float().reshape(B, 1, T).contiguous()
        x = S.ew(S.EW_ADD, x, S.conv1d_direct(vol, pk(self.volume_embed.weight.unsqueeze(-1)), H, 1,
                                               bias=self.volume_embed.bias), alpha=1.0, beta=1.0)
        if self.n_spk is not None and self.n_spk > 1:
            if spk_mix_dict is not None:
                for k, v in spk_mix_dict.items():
                    e = self.spk_embed.weight[int(k)].detach().view(1, H, 1).expand(B, H, 1).contiguous()
                    x = S.ew_bct(S.EW_ADD, x, e, alpha=1.0, beta=float(v))
            else:
                if spk_id.shape[1] > 1:
                    # per-frame speaker tracks (:150-156): spk_id is the [n_frames, n_spk] mix matrix Svc builds
                    # (infer_tool.py:275-279), x[:, :, t] += sum_s mix[t, s] * spk_embed[s] — one [H x S] x [S x T] product
                    if getattr(self, "speaker_map", None) is None:
                        raise S.SvcError("per-frame speaker mix needs init_spkmix(n_spk) first (Svc(spk_mix_enable=True))")
                    mix = spk_id.float().contiguous()                                             # [T, S]
                    if mix.shape[0] != T or mix.shape[1] != self.speaker_map.shape[0]:
                        raise S.SvcError(f"speaker mix {tuple(mix.shape)} does not match {T} frames x {self.speaker_map.shape[0]} speakers")
                    Sn = mix.shape[1]
                    g = S.gemm(self.speaker_map, mix, (0, 1, H), (0, 1, Sn), 1, H, T, Sn)        # [1, H, T]
                    x = S.ew(S.EW_ADD, x, g if B == 1 else g.expand(B, H, T).contiguous(), alpha=1.0, beta=1.0)
                else:
                    e = self.spk_embed(spk_id.long()).detach().transpose(1, 2).contiguous()       # [B, H, 1] lookup
                    x = S.ew_bct(S.EW_ADD, x, e, alpha=1.0, beta=1.0)
        if self.aug_shift_embed is not None and aug_shift is not None:
            sh = (aug_shift.float() / 5).reshape(B, 1, 1).expand(B, 1, T).contiguous()
            x = S.ew(S.EW_ADD, x, S.conv1d_direct(sh, pk(self.aug_shift_embed.weight.unsqueeze(-1)), H, 1), alpha=1.0, beta=1.0)
        return x

    def init_spkmix(self, n_spk):
        """Reference :119-130 (called by Svc(spk_mix_enable=True) with a diffusion model, infer_tool.py:157-158): build the
        `speaker_map` that forward's per-frame branch (:150-156) multiplies the [n_frames, n_spk] mix matrix with.  As written the
        reference's method cannot run — init_spkembed reads `self.hidden_size`, an attribute Unit2Mel never sets (AttributeError;
        checked against the reference tree), and forward then adds a [B, H, N] term to the [B, N, H] embedding sum.  What both
        evidently mean — and what SynthesizerTrn.EnableCharacterMix (models.py:456-461) does for the main model — is
        speaker_map[s] = spk_embed(s) and x[:, t] += sum_s mix[t, s] * speaker_map[s]: that is what is built here, as the
        [n_spk, n_hidden] matrix the mix product reads (row s = spk_embed.weight[s])."""
        if self.n_spk is None or self.n_spk <= 1:
            raise S.SvcError("init_spkmix: the model has no speaker embedding (n_spk <= 1)")
        if n_spk > self.spk_embed.weight.shape[0]:
            raise S.SvcError(f"init_spkmix: {n_spk} speakers but the embedding has {self.spk_embed.weight.shape[0]} rows")
        self.speaker_map = self.spk_embed.weight.detach()[:n_spk].float().contiguous().clone()

    def _condition_train(self, units, f0, volume, spk_id, aug_shift):
        """The conditioning sum on the autograd ops (HIP forward + backward): gradients reach every embedding."""
        B, T, _ = units.shape
        H = self.n_hidden
        x = A.conv1d(units.float().transpose(1, 2).contiguous(), self.unit_embed.weight.unsqueeze(-1), self.unit_embed.bias)
        f0r = f0.float().reshape(B, T).contiguous()
        lf0, _ = S.f0_norm_lf0(f0r, (f0r > 0).float(), mask=None)
        lg = S.ew(S.EW_SCALE, lf0.reshape(B, 1, T).contiguous(), alpha=500.0 * math.log(10.0) / 2595.0)

        def lin1(inp, lin):            # nn.Linear(1, H) on a [B,1,T] track: w[h] * inp[b,t] (+ b[h])
            y = A.mul_bcast(inp.expand(B, H, T), lin.weight.view(1, H, 1))
            return y if lin.bias is None else A.add_bcast(y, lin.bias.view(1, H, 1))
        x = A.add(x, lin1(lg, self.f0_embed))
        x = A.add(x, lin1(volume.float().reshape(B, 1, T).contiguous(), self.volume_embed))
        if self.n_spk is not None and self.n_spk > 1:
            if spk_id.shape[1] > 1:
                raise NotImplementedError("per-frame speaker mix tracks (speaker_map) are not mirrored for Unit2Mel")
            x = A.add_bcast(x, A.embedding_bct(spk_id.long().reshape(B, 1), self.spk_embed.weight))
        if self.aug_shift_embed is not None and aug_shift is not None:
            sh = (aug_shift.float() / 5).reshape(B, 1, 1).expand(B, 1, T).contiguous()
            x = A.add(x, lin1(sh, self.aug_shift_embed))
        return x

    def forward(self, units, f0, volume, spk_id=None, spk_mix_dict=None, aug_shift=None, gt_spec=None, infer=True,
                infer_speedup=10, method="dpm-solver", k_step=300, use_tqdm=True, noise=None):
        """Reference :124-167: units [B,T,n_unit], f0 / volume [B,T,1] -> mel [B,T,out_dims]; with infer=False (the
        train_diff.py call, diffusion/solver.py:128) -> the training loss."""
        if not units.is_cuda:
            raise S.SvcError("Unit2Mel needs CUDA/ROCm tensors: the MI355X engine has no CPU fallback")
        if not infer:
            if spk_mix_dict is not None:
                raise NotImplementedError("spk_mix_dict is an inference option")
            cond = self._condition_train(units, f0, volume, spk_id, aug_shift)
            return self.decoder(cond.transpose(1, 2), gt_spec=gt_spec, infer=False, k_step=k_step, noise=noise)
        with torch.no_grad():
            return self._infer(units, f0, volume, spk_id, spk_mix_dict, aug_shift, gt_spec, infer_speedup, method, k_step,
                               use_tqdm, noise)

    def _infer(self, units, f0, volume, spk_id, spk_mix_dict, aug_shift, gt_spec, infer_speedup, method, k_step, use_tqdm, noise):
        if gt_spec is not None and k_step > self.k_step_max:
            raise Exception("The shallow diffusion k_step is greater than the maximum diffusion k_step(k_step_max)!")
        if gt_spec is None and self.k_step_max != self.timesteps:
            raise Exception("This model can only be used for shallow diffusion and can not infer alone!")
        cond = self._condition(units, f0, volume, spk_id, spk_mix_dict, aug_shift)
        return self.decoder(cond.transpose(1, 2), gt_spec=gt_spec, infer=True, infer_speedup=infer_speedup, method=method,
                            k_step=k_step, use_tqdm=use_tqdm, noise=noise)


def load_model_vocoder(model_path, device="cuda", config_path=None):
    """Reference :22-58: (model, vocoder, args) — the Unit2Mel mirror, the NSF-HiFiGAN `Vocoder` mirror
    (diffusion/vocoder.py) named by args.vocoder, and the parsed diffusion.yaml."""
    from .vocoder import Vocoder
    config_file = os.path.join(os.path.split(model_path)[0], "config.yaml") if config_path is None else config_path
    with open(config_file, "r") as config:
        args = DotDict(yaml.safe_load(config))
    vocoder = Vocoder(args.vocoder.type, args.vocoder.ckpt, device=device)
    model = Unit2Mel(args.data.encoder_out_channels, args.model.n_spk, args.model.use_pitch_aug, vocoder.dimension,
                     args.model.n_layers, args.model.n_chans, args.model.n_hidden, args.model.timesteps,
                     args.model.k_step_max)
    print(" [Loading] " + model_path)
    ckpt = torch.load(model_path, map_location="cpu")
    model.load_state_dict(ckpt["model"])
    model.to(device).eval()
    print(f"Loaded diffusion model, sampler is {args.infer.method}, speedup: {args.infer.speedup} ")
    return model, vocoder, args
