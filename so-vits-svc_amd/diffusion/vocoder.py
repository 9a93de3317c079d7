"""MI355X-native mirror of diffusion/vocoder.py: `Vocoder` = NSF-HiFiGAN wrapper used by shallow diffusion
(inference/infer_tool.py:166-171,278,303) — `extract` (audio -> log-mel, vdecoder/nsf_hifigan/nvSTFT.py) and `infer`
(mel + f0 -> audio, vdecoder/nsf_hifigan/models.py), both on libsvc_hip.so.  Resampling to the vocoder's rate
(torchaudio Resample, :27-33) is outside the engine: the audio must already be at the vocoder's sampling rate (44.1 kHz
for the released so-vits-svc / NSF-HiFiGAN pair)."""
import torch

import svc_hip as S
from vdecoder.nsf_hifigan.models import load_config, load_model
from vdecoder.nsf_hifigan.nvSTFT import STFT


class Vocoder:
    def __init__(self, vocoder_type, vocoder_ckpt, device=None):
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("Vocoder: no GPU visible and the MI355X engine has no CPU fallback")
            device = "cuda"
        self.device = device
        if vocoder_type == "nsf-hifigan":
            self.vocoder = NsfHifiGAN(vocoder_ckpt, device=device)
        elif vocoder_type == "nsf-hifigan-log10":
            self.vocoder = NsfHifiGANLog10(vocoder_ckpt, device=device)
        else:
            raise ValueError(f" [x] Unknown vocoder: {vocoder_type}")
        self.resample_kernel = {}
        self.vocoder_sample_rate = self.vocoder.sample_rate()
        self.vocoder_hop_size = self.vocoder.hop_size()
        self.dimension = self.vocoder.dimension()

    def extract(self, audio, sample_rate, keyshift=0):
        if sample_rate != self.vocoder_sample_rate:
            raise NotImplementedError(f"resampling {sample_rate} -> {self.vocoder_sample_rate} Hz (torchaudio) is outside the engine")
        return self.vocoder.extract(audio, keyshift=keyshift)          # B, n_frames, bins

    def infer(self, mel, f0):
        f0 = f0[:, :mel.size(1), 0]                                    # B, n_frames
        return self.vocoder(mel, f0)


class NsfHifiGAN(torch.nn.Module):
    def __init__(self, model_path, device=None):
        super().__init__()
        self.device = device or "cuda"
        self.model_path = model_path
        self.model = None
        self.h = load_config(model_path)
        self.stft = STFT(self.h.sampling_rate, self.h.num_mels, self.h.n_fft, self.h.win_size, self.h.hop_size, self.h.fmin,
                         self.h.fmax)

    def sample_rate(self):
        return self.h.sampling_rate

    def hop_size(self):
        return self.h.hop_size

    def dimension(self):
        return self.h.num_mels

    def extract(self, audio, keyshift=0):
        with torch.no_grad():
            return self.stft.get_mel(audio, keyshift=keyshift).transpose(1, 2)     # B, n_frames, bins

    _scale = 1.0

    def forward(self, mel, f0):
        if self.model is None:
            print("| Load HifiGAN: ", self.model_path)
            self.model, self.h = load_model(self.model_path, device=self.device)
        with torch.no_grad():
            c = mel.float().transpose(1, 2).contiguous()
            if self._scale != 1.0:
                c = S.ew(S.EW_SCALE, c, alpha=self._scale)
            return self.model(c, f0)


class NsfHifiGANLog10(NsfHifiGAN):
    _scale = 0.434294
