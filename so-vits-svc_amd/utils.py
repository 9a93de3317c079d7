"""The slice of the reference's utils.py that sits ON the synthesizer path, plus the config / checkpoint helpers
`inference.infer_tool.Svc` and `train.py` import by name.

  f0_to_coarse      utils.py:69-80      -> svc_f0_to_coarse (HIP)
  normalize_f0      utils.py:31-45      -> svc_f0_norm_lf0_f32 (HIP)
  HParams / InferHParams / get_hparams_from_file   utils.py:353-358,514-557   (host, JSON -> attribute dict)
  load_checkpoint / save_checkpoint / latest_checkpoint_path   utils.py:155-200,238-243   (host, torch.save format)

Everything outside that slice (faiss index training, speech-encoder / f0-predictor factories, matplotlib logging)
is out of scope and deliberately absent.
"""
import glob
import json
import logging
import os
import re

import numpy as np
import torch

import svc_hip as S

logger = logging.getLogger(__name__)

f0_bin = 256
f0_max = 1100.0
f0_min = 50.0


def f0_to_coarse(f0):
    """Mel-scale quantisation of f0 (Hz) to an int64 bin, bit-compatible with the reference incl. its >=256 -> 0
    wrap; runs on the GPU (no CPU fallback)."""
    return S.f0_to_coarse(f0.float())


def normalize_f0(f0, x_mask, uv, random_scale=True):
    """f0: log-f0 [B,1,T] as produced by models.py:524; subtracts the voiced mean, scales by U(0.8,1.2) when
    random_scale (training), masks."""
    B = f0.shape[0]
    if random_scale:
        factor = torch.empty(B, 1).uniform_(0.8, 1.2).to(f0.device)       # utils.py:39
    else:
        factor = torch.ones(B, 1, device=f0.device)
    _, norm = S.f0_norm_lf0(f0.float()[:, 0], uv.float(), mask=x_mask.float(), factor=factor.view(-1),
                            input_is_lf0=True)
    return norm


# ------------------------------------------------------------------------------------------------------------
# config
# ------------------------------------------------------------------------------------------------------------
class HParams:
    """Recursive attribute view of a JSON config (same surface as the reference's HParams)."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            self[k] = HParams(**v) if isinstance(v, dict) else v

    def keys(self):
        return self.__dict__.keys()

    def items(self):
        return self.__dict__.items()

    def values(self):
        return self.__dict__.values()

    def get(self, key, default=None):
        return self.__dict__.get(key, default)

    def __len__(self):
        return len(self.__dict__)

    def __getitem__(self, key):
        return getattr(self, key)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return key in self.__dict__

    def __repr__(self):
        return self.__dict__.__repr__()


class InferHParams(HParams):
    """As HParams, but a missing key reads as None (the reference's inference-time behaviour, utils.py:549-557)."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            self[k] = InferHParams(**v) if isinstance(v, dict) else v

    def __getattr__(self, name):
        return None


def get_hparams_from_file(config_path, infer_mode=False):
    with open(config_path, "r") as f:
        config = json.load(f)
    return InferHParams(**config) if infer_mode else HParams(**config)


# ------------------------------------------------------------------------------------------------------------
# checkpoints (torch.save dict: model / iteration / optimizer / learning_rate)
# ------------------------------------------------------------------------------------------------------------
def load_checkpoint(checkpoint_path, model, optimizer=None, skip_optimizer=False):
    assert os.path.isfile(checkpoint_path), checkpoint_path
    ckpt = torch.load(checkpoint_path, map_location="cpu")
    iteration = ckpt.get("iteration", 0)
    learning_rate = ckpt.get("learning_rate", 0.0)
    if optimizer is not None and not skip_optimizer and ckpt.get("optimizer") is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    saved = ckpt["model"]
    target = model.module if hasattr(model, "module") else model
    own = target.state_dict()
    new_state = {}
    for k, v in own.items():
        if k in saved and tuple(saved[k].shape) == tuple(v.shape):
            new_state[k] = saved[k].to(v.dtype)
        else:
            if k not in saved:
                logger.info("%s is not in the checkpoint", k)
            else:
                logger.warning("shape mismatch for %s: checkpoint %s vs model %s", k, tuple(saved[k].shape),
                               tuple(v.shape))
            new_state[k] = v
    target.load_state_dict(new_state)
    logger.info("Loaded checkpoint '%s' (iteration %s)", checkpoint_path, iteration)
    return model, optimizer, learning_rate, iteration


def save_checkpoint(model, optimizer, learning_rate, iteration, checkpoint_path):
    target = model.module if hasattr(model, "module") else model
    torch.save({"model": target.state_dict(), "iteration": iteration,
                "optimizer": optimizer.state_dict() if optimizer is not None else None,
                "learning_rate": learning_rate}, checkpoint_path)


def latest_checkpoint_path(dir_path, regex="G_*.pth"):
    files = glob.glob(os.path.join(dir_path, regex))
    files.sort(key=lambda f: int("".join(filter(str.isdigit, f)) or -1))
    return files[-1]


def load_wav_to_torch(full_path):
    """utils.py:301-303 (scipy.io.wavfile)."""
    from scipy.io.wavfile import read
    sampling_rate, data = read(full_path)
    return torch.FloatTensor(data.astype(np.float32)), sampling_rate


def load_filepaths_and_text(filename, split="|"):
    """utils.py:306-309."""
    with open(filename, encoding="utf-8") as f:
        return [line.strip().split(split) for line in f]


def repeat_expand_2d(content, target_len, mode="left"):
    """utils.py:396-424: stretch [H, Tsrc] units to target_len frames.  'left' = the reference's sequential fill: frame i
    takes source column p(i), where p advances past column c once i >= edge[c+1] = (c+1)*target_len/src_len — but by AT
    MOST ONE column per frame (when target_len < src_len the fill lags behind the edges; kept, it is what the reference
    feeds the model).  p(i) = min(a(i), p(i-1)+1) with a(i) = #{edges[1:] <= i}  ==>  p(i) = i + min_{j<=i}(a(j) - j):
    one searchsorted + one cumulative minimum instead of a Python loop over frames.  Other modes = F.interpolate."""
    if mode != "left":
        return torch.nn.functional.interpolate(content[None], size=target_len, mode=mode)[0]
    src_len = content.shape[-1]
    edges = torch.arange(src_len + 1) * target_len / src_len                     # float32, as the reference computes it
    i = torch.arange(target_len)
    a = torch.searchsorted(edges[1:].contiguous(), i.to(edges.dtype), right=True)
    idx = i + torch.cummin(a - i, dim=0).values
    return content[:, idx.clamp_(max=src_len - 1).to(content.device)].float()


class Volume_Extractor:
    """utils.py:560-572: per-frame RMS of the waveform (reflect pad hop/2, mean of squares over hop, sqrt).  Feeds
    `vol` when vol_embedding=True; host-side plumbing on torch tensors (not on the synthesizer's kernel path)."""

    def __init__(self, hop_size=512):
        self.hop_size = hop_size

    def extract(self, audio):
        if not isinstance(audio, torch.Tensor):
            audio = torch.as_tensor(audio, dtype=torch.float32)
        n_frames = int(audio.size(-1) // self.hop_size)
        a2 = torch.nn.functional.pad((audio ** 2)[:, None, :], (self.hop_size // 2, (self.hop_size + 1) // 2),
                                     mode="reflect")[:, 0]
        return a2[:, :n_frames * self.hop_size].reshape(a2.shape[0], n_frames, self.hop_size).mean(-1)[0].sqrt()
