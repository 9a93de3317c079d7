"""Audio plumbing either side of the synthesizer for `inference.infer_tool.Svc`: decoding / encoding
of wav containers, the framewise RMS behind the silence slicer, and sample-rate conversion.

The reference does these with torchaudio / soundfile / librosa (inference/infer_tool.py:219-222,270-274,462-464,
inference/slicer.py:41,125).  None of the three is a dependency of this engine:

  read_audio        torchaudio.load (:271)  -> soundfile when it is importable, else scipy.io.wavfile (PCM / float wav)
  pcm16_round_trip  soundfile.write(BytesIO, format="wav") + torchaudio.load (:462-464, :271): the reference hands every
                    chunk to infer() through an in-memory 16-bit wav, i.e. x -> round(x * 32767) / 32768 (libsndfile's
                    float -> PCM_16 scaling, torchaudio's int16 normalisation).  UNPINNED: soundfile is absent here.
  frame_rms         librosa.feature.rms(y, frame_length, hop_length) of librosa 0.9.1 (requirements.txt:23: centred frames,
                    reflect padding).  UNPINNED: librosa is absent here.
  Resampler         torchaudio.transforms.Resample(orig, new) (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99): the
                    filter bank is built on the host in float64 like torchaudio's `_get_sinc_resample_kernel`; the
                    convolution runs in libsvc_hip.so (svc_resample_sinc_f32).  No CPU fallback.  UNPINNED against
                    torchaudio itself (absent here); the oracle restates the same published formula (oracle/audio_oracle.py).
"""
import io
import math

import numpy as np
import torch

import svc_hip as S


def read_audio(path_or_file):
    """-> (float32 numpy [channels, samples] in [-1, 1), sample_rate), torchaudio.load's convention."""
    try:
        import soundfile
        data, sr = soundfile.read(path_or_file, dtype="float32", always_2d=True)
        return np.ascontiguousarray(data.T), int(sr)
    except ImportError:
        pass
    from scipy.io import wavfile
    sr, data = wavfile.read(path_or_file)
    if data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    if x.ndim == 1:
        x = x[:, None]
    return np.ascontiguousarray(x.T), int(sr)


def write_wav(path_or_file, data, sr):
    """16-bit PCM wav (soundfile.write's default subtype for format='wav'); data float [-1, 1]."""
    pcm = np.clip(np.rint(np.asarray(data, dtype=np.float64) * 32767.0), -32768, 32767).astype(np.int16)
    from scipy.io import wavfile
    wavfile.write(path_or_file, int(sr), pcm)


def pcm16_round_trip(x):
    """What a float signal looks like after soundfile.write(..., format='wav') + torchaudio.load."""
    x = np.asarray(x, dtype=np.float64)
    return (np.clip(np.rint(x * 32767.0), -32768, 32767) / 32768.0).astype(np.float32)


def frame_rms(y, frame_length, hop_length):
    """librosa.feature.rms (0.9.1 defaults: center=True, pad_mode='reflect') -> [n_frames] float32."""
    y = np.asarray(y, dtype=np.float32)
    pad = frame_length // 2
    yp = np.pad(y, (pad, pad), mode="reflect")
    n = 1 + (len(yp) - frame_length) // hop_length
    # mean of squares per frame via a cumulative sum (float64): frames overlap 4x, a strided view would do 4x the work
    cs = np.concatenate([[0.0], np.cumsum(yp.astype(np.float64) ** 2)])
    starts = np.arange(n) * hop_length
    power = (cs[starts + frame_length] - cs[starts]) / frame_length
    return np.sqrt(np.maximum(power, 0.0)).astype(np.float32)


def sinc_resample_bank(orig, new, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio's `_get_sinc_resample_kernel` (sinc_interp_hann) for the reduced rates orig/new:
    -> (bank float32 [K, new] tap-major, width) with K = 2*width + orig."""
    base = min(orig, new) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base))
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t = np.clip(t * base, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(t == 0, 1.0, np.sin(t) / t)
    k = k * window * (base / orig)
    return np.ascontiguousarray(k.T.astype(np.float32)), width


class Resampler:
    """Callable like torchaudio.transforms.Resample(orig_freq, new_freq): waveform [..., L] -> [..., ceil(L*new/orig)]."""

    def __init__(self, orig_freq, new_freq):
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        g = math.gcd(self.orig_freq, self.new_freq)
        self.o, self.n = self.orig_freq // g, self.new_freq // g
        self._bank = {}

    def to(self, device):
        return self

    def __call__(self, wav):
        if self.o == self.n:
            return wav
        if not isinstance(wav, torch.Tensor):
            wav = torch.as_tensor(np.asarray(wav, dtype=np.float32))
        if not wav.is_cuda:
            if not torch.cuda.is_available():
                raise S.SvcError("Resampler needs a GPU: the MI355X engine has no CPU fallback")
            wav = wav.cuda()
        key = str(wav.device)
        if key not in self._bank:
            bank, width = sinc_resample_bank(self.o, self.n)
            self._bank[key] = (torch.from_numpy(bank).to(wav.device), width)
        bank, width = self._bank[key]
        shape = wav.shape
        y = S.resample_sinc(wav.reshape(-1, shape[-1]).float(), bank, self.o, self.n, width)
        return y.view(*shape[:-1], y.shape[-1])


def wav_bytes(data, sr):
    f = io.BytesIO()
    write_wav(f, data, sr)
    f.seek(0)
    return f
