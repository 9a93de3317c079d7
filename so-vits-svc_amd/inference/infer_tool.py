"""Boundary mirror of the reference's inference/infer_tool.py `Svc` (inference/infer_tool.py:116-496): same constructor
signature, attributes (`target_sample`, `hop_size`, `spk2id`, `dev`, `net_g_ms`, `hubert_model`) and `infer` /
`slice_inference` / `clear_empty` / `unload_model` methods, with the synthesizer running on libsvc_hip.so.

Shallow diffusion (`shallow_diffusion=True` / `only_diffusion=True`, infer_tool.py:163-181,278-304) runs on the engine too:
synthesizer -> `Vocoder.extract` (log-mel, rocFFT) -> `Unit2Mel` (WaveNet denoiser, DDIM / PNDM / DPM-Solver(++)) ->
`Vocoder.infer` (stand-alone NSF-HiFiGAN), see diffusion/{unit2mel,vocoder}.py.

Scope: the I/O glue around the hot path — wav decoding, the fairseq ContentVec unit
encoders (the HuBERT-soft encoder IS mirrored: vencoder/HubertSoft.py), the f0 predictors (parselmouth/pyworld/crepe),
k-means / faiss retrieval and the enhancer — is OUT OF SCOPE of this engine and their third-party dependencies are not in
this image.  They are therefore *injected*: `Svc(..., front_end=FrontEnd)` (or assigning `svc.hubert_model`, `svc.f0_predictor_object`,
`svc.load_wav`) supplies objects with the reference's own interfaces
    hubert_model.encoder(wav16k[T16]) -> [1, ssl_dim, T50]                 (vencoder/encoder.py:8-13)
    f0_predictor_object.compute_f0_uv(wav[T]) -> (f0[Tf], uv[Tf]) numpy     (modules/F0Predictor/F0Predictor.py:10-16)
    load_wav(path_or_file) -> (float32 numpy [T], sample_rate)
A missing front-end raises SvcFrontEndMissing; nothing here falls back to a CPU model.  `infer_units` is the entry point
below get_unit_f0 (infer_tool.py:297) and is what the parity tests and bench.py drive.
"""
import gc
import os
import time

import numpy as np
import torch

import utils
from models import SynthesizerTrn


class F0FilterException(Exception):
    pass


class SvcFrontEndMissing(RuntimeError):
    pass


def pad_array(arr, target_length):
    """Centre-crop / zero-pad to target_length (infer_tool.py:92-106)."""
    n = arr.shape[0]
    if n >= target_length:
        lo = (n - target_length) // 2
        return arr[lo:lo + target_length]
    lo = (target_length - n) // 2
    return np.pad(arr, (lo, target_length - n - lo), "constant")


def split_list_by_n(seq, n, pre=0):
    for i in range(0, len(seq), n):
        yield seq[i - pre if i - pre >= 0 else i: i + n]


repeat_expand_2d = utils.repeat_expand_2d      # utils.py:396-424 (the reference calls utils.repeat_expand_2d, infer_tool.py:240)


class Svc(object):
    def __init__(self, net_g_path, config_path, device=None, cluster_model_path="logs/44k/kmeans_10000.pt",
                 nsf_hifigan_enhance=False, diffusion_model_path="logs/44k/diffusion/model_0.pt",
                 diffusion_config_path="configs/diffusion.yaml", shallow_diffusion=False, only_diffusion=False,
                 spk_mix_enable=False, feature_retrieval=False, front_end=None):
        if nsf_hifigan_enhance:
            raise NotImplementedError("the NSF-HiFiGAN enhancer (modules/enhancer.py: resampling + key-shifted mel extraction) "
                                      "is outside the MI355X engine's scope (SURVEY.md §2 row 22)")
        if feature_retrieval or (cluster_model_path and os.path.exists(cluster_model_path)):
            raise NotImplementedError("k-means / faiss feature retrieval is out of scope (SURVEY.md §2 row 19)")
        self.net_g_path = net_g_path
        self.only_diffusion = only_diffusion
        self.shallow_diffusion = shallow_diffusion
        self.feature_retrieval = False
        self.nsf_hifigan_enhance = False
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("Svc: no GPU visible and the MI355X engine has no CPU fallback")
            self.dev = torch.device("cuda")
        else:
            self.dev = torch.device(device)
        self.vol_embedding = False
        self.net_g_ms = None
        if not self.only_diffusion:                                          # infer_tool.py:141-160
            self.hps_ms = utils.get_hparams_from_file(config_path, True)
            self.target_sample = self.hps_ms.data.sampling_rate
            self.hop_size = self.hps_ms.data.hop_length
            self.spk2id = self.hps_ms.spk
            self.unit_interpolate_mode = self.hps_ms.data.unit_interpolate_mode or "left"
            self.vol_embedding = bool(self.hps_ms.model.vol_embedding)
            self.speech_encoder = self.hps_ms.model.speech_encoder or "vec768l12"
        if self.shallow_diffusion or self.only_diffusion:                    # infer_tool.py:163-181
            if os.path.exists(diffusion_model_path) and os.path.exists(diffusion_config_path):
                from diffusion.unit2mel import load_model_vocoder
                self.diffusion_model, self.vocoder, self.diffusion_args = load_model_vocoder(
                    diffusion_model_path, self.dev, config_path=diffusion_config_path)
                if self.only_diffusion:
                    self.target_sample = self.diffusion_args.data.sampling_rate
                    self.hop_size = self.diffusion_args.data.block_size
                    self.spk2id = self.diffusion_args.spk
                    self.speech_encoder = self.diffusion_args.data.encoder
                    self.unit_interpolate_mode = self.diffusion_args.data.unit_interpolate_mode or "left"
                if spk_mix_enable:
                    raise NotImplementedError("speaker-mix tracks for the diffusion model (Unit2Mel.init_spkmix) are not mirrored")
            else:
                if self.only_diffusion:
                    raise FileNotFoundError(f"only_diffusion needs {diffusion_model_path} and {diffusion_config_path}")
                print("No diffusion model or config found. Shallow diffusion mode will False")
                self.shallow_diffusion = False
        if not self.only_diffusion:
            self.load_model(spk_mix_enable)
        self.hubert_model = getattr(front_end, "hubert_model", None)
        self.f0_predictor_object = getattr(front_end, "f0_predictor_object", None)
        self.load_wav = getattr(front_end, "load_wav", None)
        self.resample = getattr(front_end, "resample", None)
        self.volume_extractor = utils.Volume_Extractor(self.hop_size)

    def load_model(self, spk_mix_enable=False):
        model_kw = {k: v for k, v in self.hps_ms.model.items()}
        self.net_g_ms = SynthesizerTrn(self.hps_ms.data.filter_length // 2 + 1,
                                       self.hps_ms.train.segment_size // self.hps_ms.data.hop_length, **model_kw)
        utils.load_checkpoint(self.net_g_path, self.net_g_ms, None)
        # the engine computes in fp32: a "half" checkpoint (compress_model.py) is up-cast at load
        self.dtype = torch.float32
        self.net_g_ms.float().eval().to(self.dev)
        if spk_mix_enable:
            self.net_g_ms.EnableCharacterMix(len(self.spk2id), self.dev)

    # -- front-ends ---------------------------------------------------------------------------------------
    def _need(self, name):
        obj = getattr(self, name, None)
        if obj is None:
            raise SvcFrontEndMissing(f"Svc.{name} is not set: the unit encoder / f0 predictor / wav loader are outside "
                                     "the engine (SURVEY.md §2 rows 16-18); inject them via Svc(front_end=...)")
        return obj

    def _speaker_id(self, speaker):
        sid = self.spk2id.get(speaker) if hasattr(self.spk2id, "get") else None
        if sid is None and isinstance(speaker, int) and len(self.spk2id) >= speaker:
            sid = speaker
        if sid is None:
            raise RuntimeError("The name you entered is not in the speaker list!")
        return torch.LongTensor([int(sid)]).to(self.dev).unsqueeze(0)

    def get_unit_f0(self, wav, tran, cluster_infer_ratio, speaker, f0_filter, f0_predictor, cr_threshold=0.05):
        if cluster_infer_ratio != 0:
            raise NotImplementedError("cluster_infer_ratio != 0 needs the out-of-scope cluster model")
        f0, uv = self._need("f0_predictor_object").compute_f0_uv(wav)
        if f0_filter and sum(f0) == 0:
            raise F0FilterException("No voice detected")
        f0 = (torch.as_tensor(np.asarray(f0), dtype=torch.float32).to(self.dev) * 2 ** (tran / 12)).unsqueeze(0)
        uv = torch.as_tensor(np.asarray(uv), dtype=torch.float32).to(self.dev).unsqueeze(0)
        wav_t = torch.from_numpy(np.asarray(wav, dtype=np.float32)).to(self.dev)
        wav16k = self._need("resample")(wav_t[None, :], self.target_sample, 16000)[0]
        c = self._need("hubert_model").encoder(wav16k)
        c = repeat_expand_2d(c.squeeze(0), f0.shape[1], self.unit_interpolate_mode)
        return c.unsqueeze(0), f0, uv

    # -- the hot path -------------------------------------------------------------------------------------
    def infer_units(self, c, f0, uv, sid, auto_predict_f0=False, noice_scale=0.4, vol=None, seed=52468):
        """(c [B,ssl,T], f0 [B,T], uv [B,T], sid) -> (audio [B,1,T*hop], f0): net_g_ms.infer (infer_tool.py:297)."""
        with torch.no_grad():
            return self.net_g_ms.infer(c.to(self.dev), f0=f0.to(self.dev), g=sid.to(self.dev), uv=uv.to(self.dev),
                                       predict_f0=auto_predict_f0, noice_scale=noice_scale, vol=vol, seed=seed)

    def infer(self, speaker, tran, raw_path, cluster_infer_ratio=0, auto_predict_f0=False, noice_scale=0.4,
              f0_filter=False, f0_predictor="pm", enhancer_adaptive_key=0, cr_threshold=0.05, k_step=100, frame=0,
              spk_mix=False, second_encoding=False, loudness_envelope_adjustment=1):
        wav, sr = self._need("load_wav")(raw_path)
        if sr != self.target_sample:
            wav = self._need("resample")(torch.from_numpy(wav)[None, :], sr, self.target_sample)[0].cpu().numpy()
        if spk_mix:
            c, f0, uv = self.get_unit_f0(wav, tran, 0, None, f0_filter, f0_predictor, cr_threshold=cr_threshold)
            n_frames = f0.size(1)
            sid = speaker[:, frame:frame + n_frames].transpose(0, 1)
        else:
            sid = self._speaker_id(speaker)
            c, f0, uv = self.get_unit_f0(wav, tran, cluster_infer_ratio, speaker, f0_filter, f0_predictor,
                                         cr_threshold=cr_threshold)
            n_frames = f0.size(1)
        start = time.time()
        vol = None
        if not self.only_diffusion:
            if self.vol_embedding:
                vol = self.volume_extractor.extract(torch.as_tensor(wav, dtype=torch.float32).to(self.dev)[None, :])[None, :]
            audio, f0 = self.infer_units(c, f0, uv, sid, auto_predict_f0=auto_predict_f0, noice_scale=noice_scale, vol=vol)
            audio = audio[0, 0].data.float()
            audio_mel = self.vocoder.extract(audio[None, :], self.target_sample) if self.shallow_diffusion else None
        else:
            audio = torch.as_tensor(wav, dtype=torch.float32).to(self.dev)
            audio_mel = None
        if self.only_diffusion or self.shallow_diffusion:                    # infer_tool.py:287-304
            vol = self.volume_extractor.extract(audio[None, :])[None, :, None].to(self.dev) if vol is None else vol[:, :, None]
            if self.shallow_diffusion and second_encoding:
                audio16k = self._need("resample")(audio[None, :], self.target_sample, 16000)[0]
                c = self._need("hubert_model").encoder(audio16k)
                c = repeat_expand_2d(c.squeeze(0), f0.shape[1], self.unit_interpolate_mode).unsqueeze(0)
            f0 = f0[:, :, None]
            c = c.transpose(-1, -2)
            with torch.no_grad():
                audio_mel = self.diffusion_model(c, f0, vol, spk_id=sid, spk_mix_dict=None, gt_spec=audio_mel, infer=True,
                                                 infer_speedup=self.diffusion_args.infer.speedup,
                                                 method=self.diffusion_args.infer.method, k_step=k_step, use_tqdm=False)
                audio = self.vocoder.infer(audio_mel, f0).squeeze()
        if loudness_envelope_adjustment != 1:
            raise NotImplementedError("loudness_envelope_adjustment != 1 (utils.change_rms, librosa) is out of scope")
        print("vits use time:{}".format(time.time() - start))
        return audio, audio.shape[-1], n_frames

    def clear_empty(self):
        torch.cuda.empty_cache()

    def unload_model(self):
        self.net_g_ms = self.net_g_ms.to("cpu")
        del self.net_g_ms
        gc.collect()

    def slice_inference(self, raw_audio_path, spk, tran, slice_db, cluster_infer_ratio, auto_predict_f0, noice_scale,
                        pad_seconds=0.5, clip_seconds=0, lg_num=0, lgr_num=0.75, f0_predictor="pm",
                        enhancer_adaptive_key=0, cr_threshold=0.05, k_step=100, use_spk_mix=False,
                        second_encoding=False, loudness_envelope_adjustment=1, chunks=None):
        """infer_tool.py:356-496.  The silence slicer (inference/slicer.py, librosa RMS) is out of scope: `chunks` =
        [(is_silence, samples)] may be supplied by the caller, else the whole file is one voiced chunk.  The
        per-chunk padding, clipping and linear cross-fade bookkeeping follows the reference."""
        if use_spk_mix:
            raise NotImplementedError("per-frame speaker mixing tracks in slice_inference are not mirrored yet")
        wav, audio_sr = self._need("load_wav")(raw_audio_path)
        audio_data = chunks if chunks is not None else [(False, wav)]
        per_size = int(clip_seconds * audio_sr)
        lg_size = int(lg_num * audio_sr)
        lg_size_r = int(lg_size * lgr_num)
        lg_size_c_l = (lg_size - lg_size_r) // 2
        lg_size_c_r = lg_size - lg_size_r - lg_size_c_l
        lg = np.linspace(0, 1, lg_size_r) if lg_size != 0 else 0
        global_frame = 0
        audio = []
        for slice_tag, data in audio_data:
            length = int(np.ceil(len(data) / audio_sr * self.target_sample))
            if slice_tag:
                audio.extend(list(np.zeros(length)))
                global_frame += length // self.hop_size
                continue
            datas = split_list_by_n(data, per_size, lg_size) if per_size != 0 else [data]
            for k, dat in enumerate(datas):
                per_length = int(np.ceil(len(dat) / audio_sr * self.target_sample)) if clip_seconds != 0 else length
                pad_len = int(audio_sr * pad_seconds)
                dat = np.concatenate([np.zeros([pad_len]), dat, np.zeros([pad_len])]).astype(np.float32)
                out_audio, _, out_frame = self.infer(spk, tran, (dat, audio_sr), cluster_infer_ratio=cluster_infer_ratio,
                                                     auto_predict_f0=auto_predict_f0, noice_scale=noice_scale,
                                                     f0_predictor=f0_predictor, cr_threshold=cr_threshold,
                                                     frame=global_frame)
                global_frame += out_frame
                _audio = out_audio.cpu().numpy()
                pad_len = int(self.target_sample * pad_seconds)
                _audio = pad_array(_audio[pad_len:-pad_len], per_length)
                if lg_size != 0 and k != 0:
                    lg1 = audio[-(lg_size_r + lg_size_c_r):-lg_size_c_r] if lgr_num != 1 else audio[-lg_size:]
                    lg2 = _audio[lg_size_c_l:lg_size_c_l + lg_size_r] if lgr_num != 1 else _audio[0:lg_size]
                    lg_pre = np.asarray(lg1) * (1 - lg) + lg2 * lg
                    audio = audio[0:-(lg_size_r + lg_size_c_r)] if lgr_num != 1 else audio[0:-lg_size]
                    audio.extend(lg_pre)
                    _audio = _audio[lg_size_c_l + lg_size_r:] if lgr_num != 1 else _audio[lg_size:]
                audio.extend(list(_audio))
        return np.array(audio)
