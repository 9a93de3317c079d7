"""MI355X-native mirror of vdecoder/nsf_hifigan/nvSTFT.py: `STFT.get_mel` — the log-mel extractor of the NSF-HiFiGAN
vocoder wrapper (diffusion/vocoder.py:61-63; used by shallow diffusion, inference/infer_tool.py:278).

get_mel (:63-122) = reflect pad (win-hop)/2 -> hann STFT (center=False) -> sqrt(re^2+im^2+1e-9) -> Slaney mel basis ->
log(clamp(., 1e-5)): the same chain as modules/mel_processing.mel_spectrogram_torch with another epsilon, so it runs on
the same kernels (frame gather, batched rocFFT R2C + fused magnitude, MFMA GEMM, log-clamp).  Not mirrored: `keyshift`
/ `speed` (resized FFT for pitch augmentation in preprocessing and the enhancer), `load_wav_to_torch` (soundfile /
librosa), inputs shorter than one window (the reference switches to constant padding there)."""
import torch

from modules.mel_processing import spec_to_mel_torch, spectrogram_torch


class STFT:
    def __init__(self, sr=22050, n_mels=80, n_fft=1024, win_size=1024, hop_length=256, fmin=20, fmax=11025, clip_val=1e-5):
        self.target_sr, self.n_mels, self.n_fft, self.win_size, self.hop_length = sr, n_mels, n_fft, win_size, hop_length
        self.fmin, self.fmax, self.clip_val = fmin, fmax, clip_val
        if clip_val != 1e-5:
            raise NotImplementedError("clip_val != 1e-5 is never used by the reference")

    def get_mel(self, y, keyshift=0, speed=1, center=False):
        """y [B, L] in [-1, 1] -> log-mel [B, n_mels, frames]."""
        if keyshift != 0 or speed != 1 or center:
            raise NotImplementedError("keyshift / speed / center=True (pitch-augmented extraction) are not mirrored")
        pad_left = (self.win_size - self.hop_length) // 2
        if y.size(-1) <= max((self.win_size - self.hop_length + 1) // 2, self.win_size - y.size(-1) - pad_left):
            raise NotImplementedError("inputs shorter than one analysis window (constant padding branch, :101-104)")
        spec = spectrogram_torch(y, self.n_fft, self.target_sr, self.hop_length, self.win_size, center=False, eps=1e-9)
        return spec_to_mel_torch(spec, self.n_fft, self.n_mels, self.target_sr, self.fmin, self.fmax)

    def __call__(self, audiopath):
        raise NotImplementedError("file loading (soundfile / librosa resampling) is outside the engine: pass tensors to get_mel")


stft = STFT()
