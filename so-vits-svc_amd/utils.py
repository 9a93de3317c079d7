"""The slice of the reference's utils.py that sits ON the synthesizer path, plus the config / checkpoint helpers
`inference.infer_tool.Svc` and `train.py` import by name.

  f0_to_coarse      utils.py:69-80      -> svc_f0_to_coarse (HIP)
  normalize_f0      utils.py:31-45      -> svc_f0_norm_lf0_f32 (HIP)
  HParams / InferHParams / get_hparams_from_file   utils.py:353-358,514-557   (host, JSON -> attribute dict)
  load_checkpoint / save_checkpoint / latest_checkpoint_path   utils.py:155-200,238-243   (host, torch.save format)

  get_speech_encoder / get_f0_predictor           utils.py:88-153   (factories `Svc` calls; engine encoders or the reference's)
  get_hparams / get_logger / check_git_hash / summarize / clean_checkpoints / plot_*   utils.py:46-66,202-298,312-394
                                                   (train.py's run-directory, logging and TensorBoard helpers: host only)
  change_rms, mix_model                            utils.py:427-459

Every other name of the reference's utils.py (train_index / faiss, ...) falls through to the reference checkout's own
utils.py when one follows this package on sys.path (module __getattr__ at the bottom, svc_overlay).
"""
import glob
import json
import logging
import functools
import os
import re

import numpy as np
import torch

import svc_hip as S
import svc_overlay

svc_overlay.install()

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
    return content[:, _expand_index(content.shape[-1], int(target_len), content.device)].float()


@functools.lru_cache(maxsize=256)
def _expand_index(src_len, target_len, device):
    """The 'left' fill's source column per target frame, on `device`.  It depends on the two lengths only, so it is built
    once per pair: built per call, its pageable host -> device copy behind a running hipGraph cost a 10 s clip 20 ms of
    host-side waiting (bench e2e 32.7 -> 11.5 ms, profiles/r04o_diag_e2e.txt)."""
    edges = torch.arange(src_len + 1) * target_len / src_len                     # float32, as the reference computes it
    i = torch.arange(target_len)
    a = torch.searchsorted(edges[1:].contiguous(), i.to(edges.dtype), right=True)
    idx = i + torch.cummin(a - i, dim=0).values
    return idx.clamp_(max=src_len - 1).to(device)


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


# ------------------------------------------------------------------------------------------------------------
# factories for the front-ends `Svc` builds (utils.py:88-153).  The encoders this engine mirrors come from this package's
# vencoder/; every other name is imported the reference's way and resolves — through svc_overlay — to the reference
# checkout next on sys.path, with whatever third-party packages that module needs.
# ------------------------------------------------------------------------------------------------------------
_F0_PREDICTORS = {"pm": "PMF0Predictor", "crepe": "CrepeF0Predictor", "harvest": "HarvestF0Predictor",
                  "dio": "DioF0Predictor", "rmvpe": "RMVPEF0Predictor", "fcpe": "FCPEF0Predictor"}
_SPEECH_ENCODERS = {"vec768l12": "ContentVec768L12", "vec256l9": "ContentVec256L9", "vec256l9-onnx": "ContentVec256L9_Onnx",
                    "vec256l12-onnx": "ContentVec256L12_Onnx", "vec768l9-onnx": "ContentVec768L9_Onnx",
                    "vec768l12-onnx": "ContentVec768L12_Onnx", "hubertsoft-onnx": "HubertSoft_Onnx", "hubertsoft": "HubertSoft",
                    "whisper-ppg": "WhisperPPG", "cnhubertlarge": "CNHubertLarge", "dphubert": "DPHubert",
                    "whisper-ppg-large": "WhisperPPGLarge", "wavlmbase+": "WavLMBasePlus"}


def get_f0_predictor(f0_predictor, hop_length, sampling_rate, **kargs):
    """utils.py:88-109: name -> modules.F0Predictor.<Class>(hop_length, sampling_rate[, dtype, device, threshold])."""
    import importlib
    cls_name = _F0_PREDICTORS.get(f0_predictor)
    if cls_name is None:
        raise Exception("Unknown f0 predictor")
    cls = getattr(importlib.import_module("modules.F0Predictor." + cls_name), cls_name)
    kw = dict(hop_length=hop_length, sampling_rate=sampling_rate)
    if f0_predictor in ("crepe", "rmvpe", "fcpe"):
        kw.update(device=kargs["device"], threshold=kargs["threshold"])
    if f0_predictor in ("rmvpe", "fcpe"):
        kw.update(dtype=torch.float32)
    return cls(**kw)


def get_speech_encoder(speech_encoder, device=None, **kargs):
    """utils.py:111-153: name -> vencoder.<Class>(device=device)."""
    import importlib
    cls_name = _SPEECH_ENCODERS.get(speech_encoder)
    if cls_name is None:
        raise Exception("Unknown speech encoder")
    return getattr(importlib.import_module("vencoder." + cls_name), cls_name)(device=device)


def get_content(cmodel, y):
    """utils.py:82-86."""
    with torch.no_grad():
        c = cmodel.extract_features(y.squeeze(1))[0]
    return c.transpose(1, 2)


# ------------------------------------------------------------------------------------------------------------
# run directory / logging helpers of train.py (utils.py:202-236,312-394): host-side bookkeeping, no device work
# ------------------------------------------------------------------------------------------------------------
def get_hparams(init=True):
    """`-c config.json -m model_name` -> HParams with .model_dir = ./logs/<model>; the config is copied there on a fresh
    start and read back from there otherwise (utils.py:312-339)."""
    import argparse
    parser = argparse.ArgumentParser()
    parser.add_argument("-c", "--config", type=str, default="./configs/config.json", help="JSON file for configuration")
    parser.add_argument("-m", "--model", type=str, required=True, help="Model name")
    args = parser.parse_args()
    model_dir = os.path.join("./logs", args.model)
    os.makedirs(model_dir, exist_ok=True)
    saved = os.path.join(model_dir, "config.json")
    if init:
        with open(args.config, "r") as f:
            text = f.read()
        with open(saved, "w") as f:
            f.write(text)
    else:
        with open(saved, "r") as f:
            text = f.read()
    hparams = HParams(**json.loads(text))
    hparams.model_dir = model_dir
    return hparams


def get_hparams_from_dir(model_dir):
    hparams = get_hparams_from_file(os.path.join(model_dir, "config.json"))
    hparams.model_dir = model_dir
    return hparams


def check_git_hash(model_dir):
    """Warn when the run directory was started from another commit of the SOURCE tree (utils.py:361-378)."""
    import subprocess
    source_dir = os.path.dirname(os.path.realpath(__file__))
    if not os.path.exists(os.path.join(source_dir, ".git")):
        logger.warning("%s is not a git repository, therefore hash value comparison will be ignored.", source_dir)
        return
    cur_hash = subprocess.getoutput("git rev-parse HEAD")
    path = os.path.join(model_dir, "githash")
    if os.path.exists(path):
        with open(path) as f:
            saved_hash = f.read()
        if saved_hash != cur_hash:
            logger.warning("git hash values are different. %s(saved) != %s(current)", saved_hash[:8], cur_hash[:8])
    else:
        with open(path, "w") as f:
            f.write(cur_hash)


def get_logger(model_dir, filename="train.log"):
    global logger
    logger = logging.getLogger(os.path.basename(model_dir))
    logger.setLevel(logging.DEBUG)
    os.makedirs(model_dir, exist_ok=True)
    h = logging.FileHandler(os.path.join(model_dir, filename))
    h.setLevel(logging.DEBUG)
    h.setFormatter(logging.Formatter("%(asctime)s\t%(name)s\t%(levelname)s\t%(message)s"))
    logger.addHandler(h)
    return logger


def clean_checkpoints(path_to_models="logs/44k/", n_ckpts_to_keep=2, sort_by_time=True):
    """Delete all but the newest `n_ckpts_to_keep` G_*/D_* checkpoints (G_0 / D_0 always stay), utils.py:202-225."""
    files = [f for f in os.listdir(path_to_models) if os.path.isfile(os.path.join(path_to_models, f))]
    if sort_by_time:
        key = lambda f: os.path.getmtime(os.path.join(path_to_models, f))          # noqa: E731
    else:
        key = lambda f: int(re.compile("._(\\d+)\\.pth").match(f).group(1))       # noqa: E731
    for prefix in ("G", "D"):
        ordered = sorted([f for f in files if f.startswith(prefix) and not f.endswith("_0.pth")], key=key)
        for f in ordered[:-n_ckpts_to_keep]:
            os.remove(os.path.join(path_to_models, f))
            logger.info(".. Free up space by deleting ckpt %s", os.path.join(path_to_models, f))


def summarize(writer, global_step, scalars={}, histograms={}, images={}, audios={}, audio_sampling_rate=22050):
    """TensorBoard fan-out (utils.py:227-235); `writer` is whatever SummaryWriter-like object the caller built."""
    for k, v in scalars.items():
        writer.add_scalar(k, v, global_step)
    for k, v in histograms.items():
        writer.add_histogram(k, v, global_step)
    for k, v in images.items():
        writer.add_image(k, v, global_step, dataformats="HWC")
    for k, v in audios.items():
        writer.add_audio(k, v, global_step, audio_sampling_rate)


def _figure_to_numpy(fig):
    fig.canvas.draw()
    data = np.frombuffer(fig.canvas.buffer_rgba(), dtype=np.uint8).reshape(fig.canvas.get_width_height()[::-1] + (4,))
    return np.ascontiguousarray(data[:, :, :3])


def _pyplot():
    import matplotlib
    matplotlib.use("Agg")
    logging.getLogger("matplotlib").setLevel(logging.WARNING)
    import matplotlib.pylab as plt
    return plt


def plot_spectrogram_to_numpy(spectrogram):
    """utils.py:246-269: [bins, frames] -> RGB image array for TensorBoard."""
    plt = _pyplot()
    fig, ax = plt.subplots(figsize=(10, 2))
    im = ax.imshow(spectrogram, aspect="auto", origin="lower", interpolation="none")
    plt.colorbar(im, ax=ax)
    plt.xlabel("Frames")
    plt.ylabel("Channels")
    plt.tight_layout()
    data = _figure_to_numpy(fig)
    plt.close()
    return data


def plot_data_to_numpy(x, y):
    """utils.py:46-66: two curves on one axis -> RGB image array."""
    plt = _pyplot()
    fig, ax = plt.subplots(figsize=(10, 2))
    plt.plot(x)
    plt.plot(y)
    plt.tight_layout()
    data = _figure_to_numpy(fig)
    plt.close()
    return data


def plot_alignment_to_numpy(alignment, info=None):
    """utils.py:272-298."""
    plt = _pyplot()
    fig, ax = plt.subplots(figsize=(6, 4))
    im = ax.imshow(alignment.transpose(), aspect="auto", origin="lower", interpolation="none")
    fig.colorbar(im, ax=ax)
    plt.xlabel("Decoder timestep" + ("\n\n" + info if info is not None else ""))
    plt.ylabel("Encoder timestep")
    plt.tight_layout()
    data = _figure_to_numpy(fig)
    plt.close()
    return data


def change_rms(data1, sr1, data2, sr2, rate):
    """utils.py:440-459 (loudness-envelope transfer, `-lea`): data2 *= rms1^(1-rate) * rms2^(rate-1) with the half-second
    RMS envelopes of the input (data1, numpy) and the output (data2, device tensor) linearly interpolated to data2's length."""
    import svc_audio
    F = torch.nn.functional
    rms1 = svc_audio.frame_rms(np.asarray(data1), sr1 // 2 * 2, sr1 // 2)[None, :]
    rms2 = svc_audio.frame_rms(data2.detach().cpu().numpy(), sr2 // 2 * 2, sr2 // 2)[None, :]
    rms1 = F.interpolate(torch.from_numpy(rms1).to(data2.device).unsqueeze(0), size=data2.shape[0], mode="linear").squeeze()
    rms2 = F.interpolate(torch.from_numpy(rms2).to(data2.device).unsqueeze(0), size=data2.shape[0], mode="linear").squeeze()
    rms2 = torch.max(rms2, torch.zeros_like(rms2) + 1e-6)
    data2 *= torch.pow(rms1, torch.tensor(1 - rate)) * torch.pow(rms2, torch.tensor(rate - 1))
    return data2


def mix_model(model_paths, mix_rate, mode):
    """utils.py:427-438: weighted average of checkpoints (mode 0: softmax of the rates)."""
    rate = torch.FloatTensor(mix_rate) / 100
    base = torch.load(model_paths[0], map_location="cpu")
    models = [torch.load(path, map_location="cpu")["model"] for path in model_paths]
    if mode == 0:
        rate = torch.softmax(rate, dim=0)
    for k in base["model"].keys():
        base["model"][k] = sum(m[k] * rate[i] for i, m in enumerate(models))
    out = os.path.join(os.path.curdir, "output.pth")
    torch.save(base, out)
    return out


def __getattr__(name):
    """Names of the reference's utils.py that this engine does not restate (train_index / faiss, ...) resolve to the
    reference checkout's own utils.py when one is on sys.path behind this package (svc_overlay)."""
    if name.startswith("__"):
        raise AttributeError(name)
    import svc_overlay
    ref = svc_overlay.load_reference_module("utils", "_svc_reference_utils")
    if ref is not None and hasattr(ref, name):
        return getattr(ref, name)
    raise AttributeError(f"module 'utils' (MI355X engine mirror) has no attribute {name!r}"
                         + ("" if ref is not None else " and no reference checkout is on sys.path to take it from"))
