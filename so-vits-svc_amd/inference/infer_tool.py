"""Boundary mirror of the reference's inference/infer_tool.py: the module-level helpers
`inference_main.py` / `flask_api*.py` / `webUI.py` call (`read_temp`, `write_temp`, `timeit`, `format_wav`, `get_end_file`,
`get_md5`, `fill_a_to_b`, `mkdir`, `pad_array`, `split_list_by_n`), `Svc` with the reference's constructor signature,
attributes (`target_sample`, `hop_size`, `spk2id`, `dev`, `net_g_ms`, `hubert_model`, ...) and methods (`infer`,
`slice_inference`, `clear_empty`, `unload_model`, `get_unit_f0`, `load_model`), `F0FilterException` and `RealTimeVC`.

What runs where:
  * synthesizer, shallow diffusion (infer_tool.py:163-181,278-304), stand-alone NSF-HiFiGAN vocoder, the unit encoders
    `vec768l12` / `vec256l9` / `hubertsoft` (vencoder/*), the 44.1 kHz -> 16 kHz resampling (:219-222) -> libsvc_hip.so;
  * front-ends are built the reference's way — `utils.get_speech_encoder(self.speech_encoder, device=self.dev)` (:166) and
    `utils.get_f0_predictor(...)` (:207) — so an UNCHANGED caller works; f0 predictors (parselmouth / pyworld / crepe), k-means /
    faiss retrieval (`cluster`), the enhancer and the encoders this engine does not mirror are resolved from the reference
    checkout next on `sys.path` (svc_overlay) with their own third-party dependencies; a missing one raises ImportError.
    `Svc(..., front_end=obj)` (an extension) injects `hubert_model` / `f0_predictor_object` directly instead;
  * `slice_inference(..., batch_chunks=True)` (extension): voiced chunks with the SAME number of frames —
    what forced clipping (`clip_seconds`) produces — go through ONE `SynthesizerTrn.infer` call with B = number of such chunks.
    Same inputs and noise as the serial order: every chunk re-seeds the generator (models.py:498-501), so items of equal length
    see the same draws; the batched call draws them once at B = 1 and broadcasts (outputs agree to fp32 round-off — the
    conv tiling, hence the summation order, depends on B).
"""
import gc
import hashlib
import io
import json
import logging
import os
import pickle
import time
from pathlib import Path

import numpy as np
import torch

import svc_audio
import utils
from inference import slicer
from models import SynthesizerTrn


# -------------------------------------------------------------------------------------------------------------
# module-level helpers (inference/infer_tool.py:27-108)
# -------------------------------------------------------------------------------------------------------------
def read_temp(file_name):
    """JSON cache of slicer results keyed by wav hash (:27-46): created when missing, entries older than 14 days dropped
    once the file outgrows 50 MB, rebuilt when unreadable."""
    if not os.path.exists(file_name):
        write_temp(file_name, {"info": "temp_dict"})
        return {}
    try:
        with open(file_name, "r") as f:
            data = json.loads(f.read())
        if os.path.getsize(file_name) > 50 * 1024 * 1024:
            print(f"clean {os.path.basename(file_name.replace(chr(92), '/'))}")
            now = int(time.time())
            for key in [k for k in data.keys()]:
                if now - int(data[key]["time"]) > 14 * 24 * 3600:
                    del data[key]
        return data
    except Exception as e:  # noqa: BLE001 — same catch-all as the reference: any damage means "start over"
        print(e)
        print(f"{file_name} error,auto rebuild file")
        return {"info": "temp_dict"}


def write_temp(file_name, data):
    with open(file_name, "w") as f:
        f.write(json.dumps(data))


def timeit(func):
    def run(*args, **kwargs):
        t0 = time.time()
        res = func(*args, **kwargs)
        print("executing '%s' costed %.3fs" % (func.__name__, time.time() - t0))
        return res
    return run


def format_wav(audio_path):
    """Non-wav inputs are decoded and re-written next to the source as .wav (:62-66; needs librosa + soundfile)."""
    if Path(audio_path).suffix == ".wav":
        return
    import librosa
    import soundfile
    raw_audio, raw_sample_rate = librosa.load(audio_path, mono=True, sr=None)
    soundfile.write(Path(audio_path).with_suffix(".wav"), raw_audio, raw_sample_rate)


def get_end_file(dir_path, end):
    found = []
    for root, dirs, files in os.walk(dir_path):
        dirs[:] = [d for d in dirs if d[0] != "."]
        found += [os.path.join(root, f).replace("\\", "/") for f in files if f[0] != "." and f.endswith(end)]
    return found


def get_md5(content):
    return hashlib.new("md5", content).hexdigest()


def fill_a_to_b(a, b):
    """Pad list a with its first element up to len(b), in place (:85-88)."""
    a.extend([a[0]] * max(0, len(b) - len(a)))


def mkdir(paths: list):
    for path in paths:
        if not os.path.exists(path):
            os.mkdir(path)


def pad_array(arr, target_length):
    """Zero-pad symmetrically up to target_length; longer arrays are returned unchanged (:95-104)."""
    n = arr.shape[0]
    if n >= target_length:
        return arr
    lo = (target_length - n) // 2
    return np.pad(arr, (lo, target_length - n - lo), "constant", constant_values=(0, 0))


def split_list_by_n(list_collection, n, pre=0):
    for i in range(0, len(list_collection), n):
        yield list_collection[i - pre if i - pre >= 0 else i: i + n]


repeat_expand_2d = utils.repeat_expand_2d


class F0FilterException(Exception):
    pass


class SvcFrontEndMissing(RuntimeError):
    pass


class Svc(object):
    def __init__(self, net_g_path, config_path, device=None, cluster_model_path="logs/44k/kmeans_10000.pt",
                 nsf_hifigan_enhance=False, diffusion_model_path="logs/44k/diffusion/model_0.pt",
                 diffusion_config_path="configs/diffusion.yaml", shallow_diffusion=False, only_diffusion=False,
                 spk_mix_enable=False, feature_retrieval=False, front_end=None):
        self.net_g_path = net_g_path
        self.only_diffusion = only_diffusion
        self.shallow_diffusion = shallow_diffusion
        self.feature_retrieval = feature_retrieval
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("Svc: no GPU visible and the MI355X engine has no CPU fallback")
            self.dev = torch.device("cuda")
        else:
            self.dev = torch.device(device)
        self.vol_embedding = False
        self.net_g_ms = None
        if not self.only_diffusion:                                          # infer_tool.py:141-148
            self.hps_ms = utils.get_hparams_from_file(config_path, True)
            self.target_sample = self.hps_ms.data.sampling_rate
            self.hop_size = self.hps_ms.data.hop_length
            self.spk2id = self.hps_ms.spk
            self.unit_interpolate_mode = self.hps_ms.data.unit_interpolate_mode or "left"
            self.vol_embedding = bool(self.hps_ms.model.vol_embedding)
            self.speech_encoder = self.hps_ms.model.speech_encoder or "vec768l12"
        self.nsf_hifigan_enhance = nsf_hifigan_enhance
        if self.shallow_diffusion or self.only_diffusion:                    # :151-166
            if os.path.exists(diffusion_model_path) and os.path.exists(diffusion_config_path):
                from diffusion.unit2mel import load_model_vocoder
                self.diffusion_model, self.vocoder, self.diffusion_args = load_model_vocoder(
                    diffusion_model_path, self.dev, config_path=diffusion_config_path)
                if self.only_diffusion:
                    self.target_sample = self.diffusion_args.data.sampling_rate
                    self.hop_size = self.diffusion_args.data.block_size
                    self.spk2id = self.diffusion_args.spk
                    self.dtype = torch.float32
                    self.speech_encoder = self.diffusion_args.data.encoder
                    self.unit_interpolate_mode = self.diffusion_args.data.unit_interpolate_mode or "left"
                if spk_mix_enable:
                    self.diffusion_model.init_spkmix(len(self.spk2id))                     # :157-158
            else:
                print("No diffusion model or config found. Shallow diffusion mode will False")
                self.shallow_diffusion = self.only_diffusion = False
        # load hubert and model (:168-175)
        if not self.only_diffusion:
            self.load_model(spk_mix_enable)
            self.volume_extractor = utils.Volume_Extractor(self.hop_size)
        else:
            self.volume_extractor = utils.Volume_Extractor(self.diffusion_args.data.block_size)
        self.f0_predictor_object = None
        if front_end is not None:
            self.hubert_model = getattr(front_end, "hubert_model", None)
            self.f0_predictor_object = getattr(front_end, "f0_predictor_object", None)
            for name in ("load_wav", "resample"):
                if getattr(front_end, name, None) is not None:
                    setattr(self, "_fe_" + name, getattr(front_end, name))
        else:
            self.hubert_model = utils.get_speech_encoder(self.speech_encoder, device=self.dev)
        if cluster_model_path and os.path.exists(cluster_model_path):       # :177-186
            if self.feature_retrieval:
                with open(cluster_model_path, "rb") as f:
                    self.cluster_model = pickle.load(f)
                self.big_npy = None
                self.now_spk_id = -1
            else:
                import cluster                      # the reference checkout's package (k-means on CPU), via sys.path
                self.cluster_model = cluster.get_cluster_model(cluster_model_path)
        else:
            self.feature_retrieval = False
        if self.shallow_diffusion:
            self.nsf_hifigan_enhance = False
        if self.nsf_hifigan_enhance:                                         # :190-192 — the reference's module, see svc_overlay
            from modules.enhancer import Enhancer
            self.enhancer = Enhancer("nsf-hifigan", "pretrain/nsf_hifigan/model", device=self.dev)

    def load_model(self, spk_mix_enable=False):
        model_kw = {k: v for k, v in self.hps_ms.model.items()}
        self.net_g_ms = SynthesizerTrn(self.hps_ms.data.filter_length // 2 + 1,
                                       self.hps_ms.train.segment_size // self.hps_ms.data.hop_length, **model_kw)
        utils.load_checkpoint(self.net_g_path, self.net_g_ms, None)
        # A "half" checkpoint (compress_model.py:21-48) is up-cast at load (utils.load_checkpoint: the fp32 masters then hold the
        # fp16 values exactly) and, as in the reference (:196-198), switches the model to half-precision inference: the generator's
        # 16-bit pipeline (SynthesizerTrn.half).  (plain and snake generators; the tiny template's odd stage widths have none and compute in fp32).
        self.dtype = torch.float32
        self.net_g_ms.float().eval().to(self.dev)
        self.half_mode = False
        if "half" in str(self.net_g_path) and os.environ.get("SVC_INFER_HALF", "1") != "0":
            try:
                self.net_g_ms.half()
                self.half_mode = True
            except NotImplementedError:
                pass
        # Opt-in precision mode of FP32 inference (no counterpart in the reference): SVC_INFER_SPLIT=1 runs the generator's convolutions
        # on the fp16 matrix instruction with every value carried as a hi + lo fp16 pair (SynthesizerTrn.split_f16): fp32-level output
        # (6e-7 from the fp32 kernels on a 10 s clip), 1.6x the clip rate.  Ignored for generators without a split form.
        self.split_mode = False
        if not self.half_mode and os.environ.get("SVC_INFER_SPLIT", "0") == "1":
            try:
                self.net_g_ms.split_f16()
                self.split_mode = True
            except NotImplementedError:
                pass
        if spk_mix_enable:
            self.net_g_ms.EnableCharacterMix(len(self.spk2id), self.dev)

    # -- front-ends ---------------------------------------------------------------------------------------
    def _need(self, name):
        obj = getattr(self, name, None)
        if obj is None:
            raise SvcFrontEndMissing(f"Svc.{name} is not set")
        return obj

    def _speaker_id(self, speaker):
        sid = self.spk2id.get(speaker) if hasattr(self.spk2id, "get") else None
        if not sid and type(speaker) is int:                                 # :281-284 (0 is falsy there too)
            n_spk = len(self.spk2id.__dict__) if hasattr(self.spk2id, "__dict__") else len(self.spk2id)
            if n_spk >= speaker:
                sid = speaker
        if sid is None:
            raise RuntimeError("The name you entered is not in the speaker list!")
        return int(sid)

    def _load(self, raw_path):
        """raw_path: a path, a file object (the reference's BytesIO wav), or an (array, sample_rate) pair."""
        if hasattr(self, "_fe_load_wav") and not isinstance(raw_path, tuple):
            return self._fe_load_wav(raw_path)
        if isinstance(raw_path, tuple):
            wav, sr = raw_path
            return np.asarray(wav, dtype=np.float32), int(sr)
        wav, sr = svc_audio.read_audio(raw_path)
        return wav[0], sr                                                    # `.numpy()[0]`: first channel (:275)

    def _resample(self, wav, src, dst):
        """[1, L] tensor at `src` Hz -> `dst` Hz on the GPU (svc_resample_sinc_f32)."""
        if src == dst:
            return wav
        if hasattr(self, "_fe_resample"):
            return self._fe_resample(wav, src, dst)
        cache = self.__dict__.setdefault("_resamplers", {})
        if (src, dst) not in cache:
            cache[(src, dst)] = svc_audio.Resampler(src, dst)
        return cache[(src, dst)](wav.to(self.dev))

    def _f0_and_wav16k(self, wav, tran, f0_filter, f0_predictor, cr_threshold=0.05):
        """The part of get_unit_f0 in front of the unit encoder (:206-222): f0 / uv from the configured predictor (CPU code of the
        reference), the 16 kHz wave for the encoder."""
        if self.f0_predictor_object is None or f0_predictor != getattr(self.f0_predictor_object, "name", f0_predictor):
            self.f0_predictor_object = utils.get_f0_predictor(f0_predictor, hop_length=self.hop_size,
                                                              sampling_rate=self.target_sample, device=self.dev,
                                                              threshold=cr_threshold)
        f0, uv = self.f0_predictor_object.compute_f0_uv(wav)
        if f0_filter and sum(f0) == 0:
            raise F0FilterException("No voice detected")
        f0 = torch.as_tensor(np.asarray(f0), dtype=torch.float32).to(self.dev)
        uv = torch.as_tensor(np.asarray(uv), dtype=torch.float32).to(self.dev)
        f0 = (f0 * 2 ** (tran / 12)).unsqueeze(0)
        uv = uv.unsqueeze(0)
        wav_t = torch.from_numpy(np.asarray(wav, dtype=np.float32)).to(self.dev)
        wav16k = self._resample(wav_t[None, :], self.target_sample, 16000)[0]
        return f0, uv, wav16k

    def get_unit_f0(self, wav, tran, cluster_infer_ratio, speaker, f0_filter, f0_predictor, cr_threshold=0.05, units=None):
        """`units`: the encoder's output for this wave when the caller already has it (batched chunks), else it is computed."""
        if units is None:
            f0, uv, wav16k = self._f0_and_wav16k(wav, tran, f0_filter, f0_predictor, cr_threshold)
            c = self._need("hubert_model").encoder(wav16k)
        else:
            c, f0, uv = units
        c = utils.repeat_expand_2d(c.squeeze(0), f0.shape[1], self.unit_interpolate_mode)
        if cluster_infer_ratio != 0:                                         # :227-253 (CPU retrieval of the reference)
            if self.feature_retrieval:
                speaker_id = self._speaker_id(speaker)
                feature_index = self.cluster_model[speaker_id]
                feat_np = np.ascontiguousarray(c.transpose(0, 1).cpu().numpy())
                if self.big_npy is None or self.now_spk_id != speaker_id:
                    self.big_npy = feature_index.reconstruct_n(0, feature_index.ntotal)
                    self.now_spk_id = speaker_id
                score, ix = feature_index.search(feat_np, k=8)
                weight = np.square(1 / score)
                weight /= weight.sum(axis=1, keepdims=True)
                npy = np.sum(self.big_npy[ix] * np.expand_dims(weight, axis=2), axis=1)
                mixed = cluster_infer_ratio * npy + (1 - cluster_infer_ratio) * feat_np
                c = torch.as_tensor(mixed, dtype=torch.float32).to(self.dev).transpose(0, 1)
            else:
                import cluster
                cluster_c = cluster.get_cluster_center_result(self.cluster_model, c.cpu().numpy().T, speaker).T
                cluster_c = torch.as_tensor(cluster_c, dtype=torch.float32).to(self.dev)
                c = cluster_infer_ratio * cluster_c + (1 - cluster_infer_ratio) * c
        return c.unsqueeze(0), f0, uv

    # -- the hot path -------------------------------------------------------------------------------------
    def infer_units(self, c, f0, uv, sid, auto_predict_f0=False, noice_scale=0.4, vol=None, seed=52468, noise=None, lengths=None):
        """(c [B,ssl,T], f0 [B,T], uv [B,T], sid) -> (audio [B,1,T*hop], f0): net_g_ms.infer (infer_tool.py:297)."""
        with torch.no_grad():
            call = lambda: self.net_g_ms.infer(c.to(self.dev), f0=f0.to(self.dev), g=sid.to(self.dev), uv=uv.to(self.dev),
                                               predict_f0=auto_predict_f0, noice_scale=noice_scale, vol=vol, seed=seed, noise=noise,
                                               lengths=lengths)
            out = call()
            if self.split_mode and self.net_g_ms.split_range_exceeded():
                # the split pipeline's range guard (include/svc_hip.h, RANGE): an activation of the generator left what two fp16
                # pieces can carry (|v| > 65504) — that result is not fp32-level.  Same call on the fp32 kernels (same seed, same
                # draws), and the model stays there: a checkpoint that does this once will do it again.
                self.range_fallbacks = getattr(self, "range_fallbacks", 0) + 1
                logging.getLogger("infer_tool").warning(
                    "SVC_INFER_SPLIT: an activation exceeded the fp16 range of the split pipeline; re-running on the fp32 kernels "
                    "and leaving split mode")
                self.net_g_ms.split_f16(False)
                self.split_mode = False
                out = call()
            return out

    def _load_at_target_rate(self, raw_path):
        wav, sr = self._load(raw_path)
        if sr != self.target_sample:
            wav = self._resample(torch.from_numpy(np.asarray(wav, dtype=np.float32))[None, :], sr,
                                 self.target_sample)[0].cpu().numpy()
        return wav

    def _features(self, speaker, tran, raw_path, cluster_infer_ratio, f0_filter, f0_predictor, cr_threshold, frame, spk_mix,
                  wav=None, units=None):
        """Everything of infer() in front of the synthesizer (:270-290): (wav, c, f0, uv, sid, n_frames).  `wav` / `units`:
        what a batched caller computed already (the wave at the target rate; (encoder output, f0, uv))."""
        if wav is None:
            wav = self._load_at_target_rate(raw_path)
        if spk_mix:
            c, f0, uv = self.get_unit_f0(wav, tran, 0, None, f0_filter, f0_predictor, cr_threshold=cr_threshold, units=units)
            n_frames = f0.size(1)
            sid = speaker[:, frame:frame + n_frames].transpose(0, 1)
        else:
            sid = torch.LongTensor([self._speaker_id(speaker)]).to(self.dev).unsqueeze(0)
            c, f0, uv = self.get_unit_f0(wav, tran, cluster_infer_ratio, speaker, f0_filter, f0_predictor,
                                         cr_threshold=cr_threshold, units=units)
            n_frames = f0.size(1)
        return wav, c.float(), f0.float(), uv.float(), sid, n_frames

    def _post(self, wav, audio, c, f0, sid, vol, k_step, second_encoding, enhancer_adaptive_key, loudness_envelope_adjustment):
        """Everything of infer() behind the synthesizer (:298-330) for ONE item; audio [L] on the device."""
        audio_mel = self.vocoder.extract(audio[None, :], self.target_sample) if self.shallow_diffusion else None
        if self.only_diffusion or self.shallow_diffusion:
            vol = self.volume_extractor.extract(audio[None, :])[None, :, None].to(self.dev) if vol is None else vol[:, :, None]
            if self.shallow_diffusion and second_encoding:
                audio16k = self._resample(audio[None, :], self.target_sample, 16000)[0]
                c = self._need("hubert_model").encoder(audio16k)
                c = utils.repeat_expand_2d(c.squeeze(0), f0.shape[1], self.unit_interpolate_mode).unsqueeze(0)
            f0 = f0[:, :, None]
            c = c.transpose(-1, -2)
            with torch.no_grad():
                audio_mel = self.diffusion_model(c, f0, vol, spk_id=sid, spk_mix_dict=None, gt_spec=audio_mel, infer=True,
                                                 infer_speedup=self.diffusion_args.infer.speedup,
                                                 method=self.diffusion_args.infer.method, k_step=k_step, use_tqdm=False)
                audio = self.vocoder.infer(audio_mel, f0).squeeze()
        if self.nsf_hifigan_enhance:
            audio, _ = self.enhancer.enhance(audio[None, :], self.target_sample, f0[:, :, None], self.hps_ms.data.hop_length,
                                             adaptive_key=enhancer_adaptive_key)
        if loudness_envelope_adjustment != 1:
            audio = utils.change_rms(wav, self.target_sample, audio, self.target_sample, loudness_envelope_adjustment)
        return audio

    def infer(self, speaker, tran, raw_path, cluster_infer_ratio=0, auto_predict_f0=False, noice_scale=0.4,
              f0_filter=False, f0_predictor="pm", enhancer_adaptive_key=0, cr_threshold=0.05, k_step=100, frame=0,
              spk_mix=False, second_encoding=False, loudness_envelope_adjustment=1):
        """infer_tool.py:256-331.  `raw_path`: a path / file object holding a wav (the reference's contract), or an
        `(float array, sample_rate)` pair (extension: what slice_inference hands over without the BytesIO round trip)."""
        wav, c, f0, uv, sid, n_frames = self._features(speaker, tran, raw_path, cluster_infer_ratio, f0_filter, f0_predictor,
                                                       cr_threshold, frame, spk_mix)
        start = time.time()
        vol = None
        if not self.only_diffusion:
            if self.vol_embedding:
                vol = self.volume_extractor.extract(torch.as_tensor(wav, dtype=torch.float32).to(self.dev)[None, :])[None, :].to(self.dev)
            audio, f0 = self.infer_units(c, f0, uv, sid, auto_predict_f0=auto_predict_f0, noice_scale=noice_scale, vol=vol)
            audio = audio[0, 0].data.float()
        else:
            audio = torch.as_tensor(wav, dtype=torch.float32).to(self.dev)
        audio = self._post(wav, audio, c, f0, sid, vol, k_step, second_encoding, enhancer_adaptive_key,
                           loudness_envelope_adjustment)
        print("vits use time:{}".format(time.time() - start))
        return audio, audio.shape[-1], n_frames

    def clear_empty(self):
        torch.cuda.empty_cache()

    def unload_model(self):
        self.net_g_ms = self.net_g_ms.to("cpu")
        del self.net_g_ms
        if hasattr(self, "enhancer"):
            self.enhancer.enhancer = self.enhancer.enhancer.to("cpu")
            del self.enhancer.enhancer
            del self.enhancer
        gc.collect()

    # -- slicing ------------------------------------------------------------------------------------------
    def _spk_mix_tensor(self, spk, audio_data, audio_sr, per_size, lg_size, pad_seconds):
        """Per-frame speaker weights from the `spkmix.py` tracks (:392-441): [n_speakers, total_frames], columns sum to 1."""
        assert len(self.spk2id) == len(spk)
        audio_length = 0
        for slice_tag, data in audio_data:
            aud_length = int(np.ceil(len(data) / audio_sr * self.target_sample))
            if slice_tag:
                audio_length += aud_length // self.hop_size
                continue
            datas = split_list_by_n(data, per_size, lg_size) if per_size != 0 else [data]
            for dat in datas:
                pad_len = int(audio_sr * pad_seconds)
                per_length = int(np.ceil(len(dat) / audio_sr * self.target_sample))
                audio_length += (per_length + 2 * pad_len) // self.hop_size
        audio_length += len(audio_data)
        mix = torch.zeros(size=(len(spk), audio_length)).to(self.dev)
        for i in range(len(spk)):
            last_end = None
            for begin_f, end_f, v0, v1 in spk[i]:
                if v1 < 0. or v0 < 0.:
                    raise RuntimeError("mix value must higer Than zero!")
                begin, end = int(audio_length * begin_f), int(audio_length * end_f)
                length = end - begin
                if length <= 0:
                    raise RuntimeError("begin Must lower Than end!")
                if last_end is not None and last_end != begin:
                    raise RuntimeError("[i]EndTime Must Equal [i+1]BeginTime!")
                last_end = end
                step = (v1 - v0) / length
                ramp = torch.zeros(length).to(self.dev) + v0 if step == 0. else torch.arange(v0, v1, step).to(self.dev)
                if len(ramp) < length:
                    ramp = torch.nn.functional.pad(ramp, [0, length - len(ramp)], mode="reflect").to(self.dev)
                mix[i][begin:end] = ramp[:length]
        total = torch.sum(mix, dim=0).unsqueeze(0).to(self.dev)
        empty = total[0] == 0.0
        total[0][empty] = 1.0
        mix[:, empty] = 1.0 / len(spk)
        mix = mix / total
        if not ((torch.sum(mix, dim=0) - 1.) < 0.0001).all():
            raise RuntimeError("sum(spk_mix_tensor) not equal 1")
        return mix

    def slice_inference(self, raw_audio_path, spk, tran, slice_db, cluster_infer_ratio, auto_predict_f0, noice_scale,
                        pad_seconds=0.5, clip_seconds=0, lg_num=0, lgr_num=0.75, f0_predictor="pm",
                        enhancer_adaptive_key=0, cr_threshold=0.05, k_step=100, use_spk_mix=False,
                        second_encoding=False, loudness_envelope_adjustment=1, batch_chunks=False, chunks=None):
        """infer_tool.py:356-496: slice at silences (inference/slicer.py), optionally force-clip every `clip_seconds`,
        pad, convert chunk by chunk, trim, cross-fade `lg_num` seconds between forced clips.  `chunks` (extension):
        [(is_silence, samples)] at the file's rate replaces the slicer.  `batch_chunks` (extension): see the module docstring."""
        if use_spk_mix and len(self.spk2id) == 1:
            spk = list(self.spk2id.keys())[0]
            use_spk_mix = False
        if chunks is not None:
            _, audio_sr = self._load(raw_audio_path)
            audio_data = chunks
        else:
            wav_path = Path(raw_audio_path).with_suffix(".wav")
            audio_data, audio_sr = slicer.chunks2audio(wav_path, slicer.cut(wav_path, db_thresh=slice_db))
        per_size = int(clip_seconds * audio_sr)
        lg_size = int(lg_num * audio_sr)
        lg_size_r = int(lg_size * lgr_num)
        lg_size_c_l = (lg_size - lg_size_r) // 2
        lg_size_c_r = lg_size - lg_size_r - lg_size_c_l
        lg = np.linspace(0, 1, lg_size_r) if lg_size != 0 else 0
        if use_spk_mix:
            spk = self._spk_mix_tensor(spk, audio_data, audio_sr, per_size, lg_size, pad_seconds)

        # ---- plan: one job per voiced (sub-)chunk, in output order -------------------------------------------
        jobs = []                                          # dict(seg, k, dat, per_length)
        for seg, (slice_tag, data) in enumerate(audio_data):
            print(f"#=====segment start, {round(len(data) / audio_sr, 3)}s======")
            length = int(np.ceil(len(data) / audio_sr * self.target_sample))
            if slice_tag:
                print("jump empty segment")
                jobs.append(dict(seg=seg, silence=length))
                continue
            datas = split_list_by_n(data, per_size, lg_size) if per_size != 0 else [data]
            for k, dat in enumerate(datas):
                per_length = int(np.ceil(len(dat) / audio_sr * self.target_sample)) if clip_seconds != 0 else length
                if clip_seconds != 0:
                    print(f"###=====segment clip start, {round(len(dat) / audio_sr, 3)}s======")
                pad_len = int(audio_sr * pad_seconds)
                dat = np.concatenate([np.zeros([pad_len]), dat, np.zeros([pad_len])])
                # the reference passes the chunk through an in-memory 16-bit wav (:462-464)
                jobs.append(dict(seg=seg, k=k, dat=svc_audio.pcm16_round_trip(dat), per_length=per_length))

        kw = dict(cluster_infer_ratio=cluster_infer_ratio, auto_predict_f0=auto_predict_f0, noice_scale=noice_scale,
                  f0_predictor=f0_predictor, enhancer_adaptive_key=enhancer_adaptive_key, cr_threshold=cr_threshold,
                  k_step=k_step, spk_mix=use_spk_mix, second_encoding=second_encoding,
                  loudness_envelope_adjustment=loudness_envelope_adjustment)
        if batch_chunks and not self.only_diffusion:
            self._run_jobs_batched(jobs, spk, tran, audio_sr, kw)
        else:
            global_frame = 0
            for j in jobs:
                if "silence" in j:
                    global_frame += j["silence"] // self.hop_size
                    continue
                out_audio, _, out_frame = self.infer(spk, tran, (j["dat"], audio_sr), frame=global_frame, **kw)
                global_frame += out_frame
                j["audio"] = out_audio.cpu().numpy()

        # ---- stitch (:466-478) -------------------------------------------------------------------------------
        audio = []
        for j in jobs:
            if "silence" in j:
                audio.extend(list(pad_array(np.zeros(j["silence"]), j["silence"])))
                continue
            pad_len = int(self.target_sample * pad_seconds)
            _audio = pad_array(j["audio"][pad_len:-pad_len], j["per_length"])
            if lg_size != 0 and j["k"] != 0:
                lg1 = audio[-(lg_size_r + lg_size_c_r):-lg_size_c_r] if lgr_num != 1 else audio[-lg_size:]
                lg2 = _audio[lg_size_c_l:lg_size_c_l + lg_size_r] if lgr_num != 1 else _audio[0:lg_size]
                lg_pre = lg1 * (1 - lg) + lg2 * lg
                audio = audio[0:-(lg_size_r + lg_size_c_r)] if lgr_num != 1 else audio[0:-lg_size]
                audio.extend(lg_pre)
                _audio = _audio[lg_size_c_l + lg_size_r:] if lgr_num != 1 else _audio[lg_size:]
            audio.extend(list(_audio))
        return np.array(audio)

    #: chunks whose frame counts fall into the same bucket (multiples of this many frames, ~0.37 s) share one synthesizer call
    BATCH_BUCKET_FRAMES = 32

    def _run_jobs_batched(self, jobs, spk, tran, audio_sr, kw):
        """Front-ends chunk by chunk (they are per-utterance models / CPU code), then ONE synthesizer call per BUCKET of chunks:
        chunks are grouped by frame count rounded up to BATCH_BUCKET_FRAMES, zero-padded to the bucket's longest item, and run
        with per-item lengths (SynthesizerTrn.infer(lengths=...): the attention / flow / encoder masks are the true lengths, as
        if each chunk ran alone).  Every item gets exactly the noise the serial loop would give it — the reference re-seeds per
        chunk (models.py:498-501), so the draws are made per item at its own length and padded.  The decoder has no mask: an
        item's samples within its receptive field of the padded tail differ from the serial run, but that region (< 0.25 s)
        lies inside the `pad_seconds` (0.5 s) of silence slice_inference adds to every chunk and trims again (:460,470).
        Fills j["audio"] exactly as the serial loop would."""
        # front-ends: f0 chunk by chunk (CPU predictor), then the unit encoder over ALL chunks at once — encoder_batch runs waves
        # of equal length (fixed-length clipping gives them) as one batch, the others one by one
        voiced = [j for j in jobs if "silence" not in j]
        pre = []
        for j in voiced:
            wav = self._load_at_target_rate((j["dat"], audio_sr))
            pre.append((wav,) + self._f0_and_wav16k(wav, tran, False, kw["f0_predictor"], kw["cr_threshold"]))
        enc = self._need("hubert_model")
        batch = getattr(enc, "encoder_batch", None)
        units = batch([p[3] for p in pre]) if batch is not None else [enc.encoder(p[3]) for p in pre]
        global_frame = 0
        feats = []
        it = iter(zip(pre, units))
        for j in jobs:
            if "silence" in j:
                global_frame += j["silence"] // self.hop_size
                continue
            (wav, f0_, uv_, w16), c_ = next(it)
            wav, c, f0, uv, sid, n_frames = self._features(spk, tran, None, kw["cluster_infer_ratio"], False, kw["f0_predictor"],
                                                           kw["cr_threshold"], global_frame, kw["spk_mix"], wav=wav,
                                                           units=(c_, f0_, uv_))
            global_frame += n_frames
            feats.append(dict(j=j, wav=wav, c=c, f0=f0, uv=uv, sid=sid, T=n_frames))
        groups = {}
        bucket = self.BATCH_BUCKET_FRAMES
        for f in feats:
            key = (-(-f["T"] // bucket), tuple(f["sid"].shape)) if not kw["spk_mix"] else (id(f),)   # speaker-mix tensors are per utterance
            groups.setdefault(key, []).append(f)
        start = time.time()
        net = self.net_g_ms
        for items in groups.values():
            B = len(items)
            T = max(f["T"] for f in items)
            ragged = any(f["T"] != T for f in items)
            padT = lambda t, f: torch.nn.functional.pad(t, (0, T - f["T"]))            # noqa: E731
            c = torch.cat([padT(f["c"], f) for f in items], 0)
            f0 = torch.cat([padT(f["f0"], f) for f in items], 0)
            uv = torch.cat([padT(f["uv"], f) for f in items], 0)
            vol = None
            if self.vol_embedding:
                vol = torch.cat([padT(self.volume_extractor.extract(torch.as_tensor(f["wav"], dtype=torch.float32).to(self.dev)[None, :])[None, :], f)
                                 for f in items], 0).to(self.dev)
            sid = items[0]["sid"] if kw["spk_mix"] else torch.cat([f["sid"] for f in items], 0)
            # the serial loop re-seeds per chunk (models.py:498-501): item b sees the draws of a B = 1 call at ITS length
            L = T * net.dec.upp
            enc_p, rand_ini, sine = [], [], []
            draws = {}
            for f in items:
                if f["T"] not in draws:
                    torch.manual_seed(52468)
                    Lf = f["T"] * net.dec.upp
                    draws[f["T"]] = (torch.randn(1, net.inter_channels, f["T"], device=self.dev), torch.rand(1, 9, device=self.dev),
                                     torch.randn(1, Lf, 9, device=self.dev))
                e, r, sn = draws[f["T"]]
                enc_p.append(padT(e, f))
                rand_ini.append(r)
                sine.append(torch.nn.functional.pad(sn, (0, 0, 0, L - sn.shape[1])))
            noise = dict(enc_p=torch.cat(enc_p, 0).contiguous(), rand_ini=torch.cat(rand_ini, 0).contiguous(),
                         sine=torch.cat(sine, 0).contiguous())
            lengths = torch.tensor([f["T"] for f in items], device=self.dev) if ragged else None
            audio, f0o = self.infer_units(c, f0, uv, sid, auto_predict_f0=kw["auto_predict_f0"], noice_scale=kw["noice_scale"],
                                          vol=vol, noise=noise, lengths=lengths)
            for b, f in enumerate(items):
                n = f["T"] * net.dec.upp
                a = self._post(f["wav"], audio[b, 0, :n].data.float(), f["c"], f0o[b:b + 1, :f["T"]], f["sid"],
                               None if vol is None else vol[b:b + 1, :f["T"]],
                               kw["k_step"], kw["second_encoding"], kw["enhancer_adaptive_key"], kw["loudness_envelope_adjustment"])
                f["j"]["audio"] = a.cpu().numpy()
        print("vits use time:{} ({} chunks in {} synthesizer calls)".format(time.time() - start, len(feats), len(groups)))


class RealTimeVC:
    """infer_tool.py:498-549: chunked conversion with a cross-faded overlap of `pre_len` samples between calls."""

    def __init__(self):
        self.last_chunk = None
        self.last_o = None
        self.chunk_len = 16000   # chunk length
        self.pre_len = 3840      # cross fade length, multiples of 640

    def process(self, svc_model, speaker_id, f_pitch_change, input_wav_path, cluster_infer_ratio=0, auto_predict_f0=False,
                noice_scale=0.4, f0_filter=False):
        """Input and output are 1-dimensional numpy waveform arrays (input_wav_path: a file object holding a wav)."""
        audio, sr = svc_audio.read_audio(input_wav_path)
        audio = audio[0]
        kw = dict(cluster_infer_ratio=cluster_infer_ratio, auto_predict_f0=auto_predict_f0, noice_scale=noice_scale,
                  f0_filter=f0_filter)
        if self.last_chunk is None:
            input_wav_path.seek(0)
            out, _, _ = svc_model.infer(speaker_id, f_pitch_change, input_wav_path, **kw)
            out = out.cpu().numpy()
            self.last_chunk = out[-self.pre_len:]
            self.last_o = out
            return out[-self.chunk_len:]
        audio = np.concatenate([self.last_chunk, audio])
        out, _, _ = svc_model.infer(speaker_id, f_pitch_change, svc_audio.wav_bytes(audio, sr), **kw)
        out = out.cpu().numpy()
        ret = _crossfade(self.last_o, out, self.pre_len)
        self.last_chunk = out[-self.pre_len:]
        self.last_o = out
        return ret[self.chunk_len:2 * self.chunk_len]


def _crossfade(s1, s2, fade_len):
    """maad.util.crossfade as the reference uses it (:546): linear fade of the last `fade_len` samples of s1 into the first
    `fade_len` of s2, the rest concatenated."""
    ramp = np.linspace(0.0, 1.0, fade_len)
    mid = s1[-fade_len:] * (1.0 - ramp) + s2[:fade_len] * ramp
    return np.concatenate([s1[:-fade_len], mid, s2[fade_len:]])
