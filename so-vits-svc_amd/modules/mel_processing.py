"""MI355X-native mirror of modules/mel_processing.py (spectrogram_torch / spec_to_mel_torch / mel_spectrogram_torch,
:40-83) for the mel-reconstruction loss of train.py:171-182,202.

  framing   reflect pad (n_fft-hop)/2, hop, hann window          -> svc_stft_frame_f32 (+ adjoint)
  transform ONE batched real-to-complex rocFFT over all B*frames windows  -> svc_rfft_forward_f32
            (backward = the complex-to-real rocFFT on the halved-interior spectrum gradient, its exact adjoint)
  |.|       sqrt(re^2 + im^2 + 1e-6) on the interleaved half spectrum -> svc_cmag_c_f32 (+ bwd)
  mel       Slaney filterbank product, log(clamp(., 1e-5))       -> svc_gemm_f32, svc_ew_f32

The Slaney mel basis is restated from librosa.filters.mel (librosa==0.9.1, requirements.txt:23; htk=False,
norm='slaney') because librosa is not vendored by the reference; tests cross-check it against
transformers.audio_utils.mel_filter_bank.  Parity of the basis against librosa itself is UNPINNED.
"""
import math

import numpy as np
import torch

import svc_autograd as A
import svc_hip as S

mel_basis = {}
hann_window = {}


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def librosa_mel_fn(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with htk=False, norm='slaney' -> [n_mels, n_fft//2+1] fp32."""
    if fmax is None:
        fmax = sr / 2.0
    n_bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0, sr / 2.0, n_bins)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, n_bins), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights.astype(np.float32)


def _window(win_size, device):
    key = f"{win_size}_{device}"
    if key not in hann_window:
        hann_window[key] = torch.hann_window(win_size).to(device=device, dtype=torch.float32)
    return hann_window[key]


def _melmat(n_fft, num_mels, sampling_rate, fmin, fmax, device):
    key = f"{fmax}_{n_fft}_{num_mels}_{sampling_rate}_{fmin}_{device}"
    if key not in mel_basis:
        mel_basis[key] = torch.from_numpy(librosa_mel_fn(sampling_rate, n_fft, num_mels, fmin, fmax)).to(device)
    return mel_basis[key]


def dynamic_range_compression_torch(x, C=1, clip_val=1e-5):
    if C != 1:
        raise NotImplementedError("C != 1 is never used by the reference")
    return A.log_clamp(x, clip_val)


def spectral_normalize_torch(magnitudes):
    return dynamic_range_compression_torch(magnitudes)


def spectrogram_torch(y, n_fft, sampling_rate, hop_size, win_size, center=False, n_frames=None, eps=1e-6, prepadded=False):
    """y [B, L] in [-1, 1] -> |STFT| [B, n_fft/2+1, frames] (reference :40-64).  n_frames (optional) limits the number of
    frames produced (data_utils.batch_spectrogram passes signals that already carry their right-hand extension);
    prepadded=True: y already holds the (n_fft-hop)/2 samples of context on both sides (no reflect padding is added)."""
    if center or win_size != n_fft:
        raise NotImplementedError("only center=False, win_size == n_fft is used by so-vits-svc")
    y = y.float()
    B, L = y.shape
    pad = 0 if prepadded else int((n_fft - hop_size) / 2)
    NF = (L + 2 * pad - n_fft) // hop_size + 1
    if n_frames is not None:
        NF = min(NF, int(n_frames))
    frames = A.stft_frames(y, _window(win_size, y.device), NF, n_fft, hop_size, pad)      # [B, NF, n_fft]
    mag = A.rfft_mag(frames, eps)                     # batched rocFFT R2C + fused magnitude -> [B, NF, bins]
    return mag.transpose(1, 2)


def spec_to_mel_torch(spec, n_fft, num_mels, sampling_rate, fmin, fmax):
    """spec [B, bins, T] -> log-mel [B, num_mels, T] (reference :67-76)."""
    M = _melmat(n_fft, num_mels, sampling_rate, fmin, fmax, spec.device)       # [mels, bins]
    x = spec.float().transpose(1, 2)                                            # [B, T, bins] (view)
    mel = A.gemm2d(x, M.t().contiguous())                                       # [B, T, mels]
    return spectral_normalize_torch(mel).transpose(1, 2)


def mel_spectrogram_torch(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False):
    spec = spectrogram_torch(y, n_fft, sampling_rate, hop_size, win_size, center)
    return spec_to_mel_torch(spec, n_fft, num_mels, sampling_rate, fmin, fmax)
