"""MI355X-side mirror of data_utils.py (TextAudioSpeakerLoader :17-128, TextAudioCollate :131-186) — SURVEY.md §8f row 3.

Same on-disk formats as the reference's preprocessing (preprocess_hubert_f0.py:31-103): `x.wav` (int16 / float PCM at
the configured rate), `x.wav.soft.pt` (torch.save, units [1, ssl_dim, T50]), `x.wav.f0.npy` (object array (f0, uv)),
`x.spec.pt` ([n_fft/2+1, T]), `x.wav.vol.npy`; same item tuple `(c, f0, spec, audio_norm, spk, uv, volume)` and the same
collate output `(c, f0, spec, wav, spkids, lengths, uv, volume)` sorted by decreasing length, zero padded.

What is re-designed: the reference computes a missing / re-scaled (vol_aug) linear spectrogram with torch.stft INSIDE
the DataLoader worker (:60-66,98-103).  Here the worker only does I/O; such items carry `spec=None`, the collate returns
`spec_padded=None`, and `batch_spectrogram` computes the spectrogram of the whole padded batch on the GPU in one batched
rocFFT call (per-item reflect padding at each item's TRUE end, so the result equals the per-item computation), which is
what keeps 8 GPUs fed without an 8x CPU STFT load.  Everything else is host-side plumbing on torch CPU tensors.
"""
import os
import random

import numpy as np
import torch
import torch.utils.data

import utils


class TextAudioSpeakerLoader(torch.utils.data.Dataset):
    def __init__(self, audiopaths, hparams, all_in_mem: bool = False, vol_aug: bool = True):
        self.audiopaths = utils.load_filepaths_and_text(audiopaths)
        self.hparams = hparams
        self.max_wav_value = hparams.data.max_wav_value
        self.sampling_rate = hparams.data.sampling_rate
        self.filter_length = hparams.data.filter_length
        self.hop_length = hparams.data.hop_length
        self.win_length = hparams.data.win_length
        self.unit_interpolate_mode = hparams.data.unit_interpolate_mode
        self.use_sr = hparams.train.use_sr
        self.spec_len = hparams.train.max_speclen
        self.spk_map = hparams.spk
        self.vol_emb = hparams.model.vol_embedding
        self.vol_aug = hparams.train.vol_aug and vol_aug
        random.seed(1234)
        random.shuffle(self.audiopaths)
        self.all_in_mem = all_in_mem
        if self.all_in_mem:
            self.cache = [self.get_audio(p[0]) for p in self.audiopaths]

    def get_audio(self, filename):
        filename = filename.replace("\\", "/")
        audio, sampling_rate = utils.load_wav_to_torch(filename)
        if sampling_rate != self.sampling_rate:
            raise ValueError("Sample Rate not match. Expect {} but got {} from {}".format(
                self.sampling_rate, sampling_rate, filename))
        audio_norm = (audio / self.max_wav_value).unsqueeze(0)
        spec_filename = filename.replace(".wav", ".spec.pt")
        spec = torch.load(spec_filename) if os.path.exists(spec_filename) else None     # None -> computed on the GPU
        spk = filename.split("/")[-2]
        spk = torch.LongTensor([self.spk_map[spk]])
        f0, uv = np.load(filename + ".f0.npy", allow_pickle=True)
        f0 = torch.FloatTensor(np.array(f0, dtype=float))
        uv = torch.FloatTensor(np.array(uv, dtype=float))
        c = torch.load(filename + ".soft.pt")
        c = utils.repeat_expand_2d(c.squeeze(0), f0.shape[0], mode=self.unit_interpolate_mode)
        volume = torch.from_numpy(np.load(filename + ".vol.npy")).float() if self.vol_emb else None
        n_spec = spec.size(-1) if spec is not None else audio_norm.shape[1] // self.hop_length
        lmin = min(c.size(-1), n_spec)
        assert abs(c.size(-1) - n_spec) < 3, (c.size(-1), n_spec, f0.shape, filename)
        assert abs(audio_norm.shape[1] - lmin * self.hop_length) < 3 * self.hop_length
        c, f0, uv = c[:, :lmin], f0[:lmin], uv[:lmin]
        if spec is not None:
            spec = spec[:, :lmin]
        audio_norm = audio_norm[:, :lmin * self.hop_length]
        if volume is not None:
            volume = volume[:lmin]
        return c, f0, spec, audio_norm, spk, uv, volume

    def random_slice(self, c, f0, spec, audio_norm, spk, uv, volume):
        if random.choice([True, False]) and self.vol_aug and volume is not None:
            max_amp = float(torch.max(torch.abs(audio_norm))) + 1e-5
            max_shift = min(1, np.log10(1 / max_amp))
            log10_vol_shift = random.uniform(-1, max_shift)
            audio_norm = audio_norm * (10 ** log10_vol_shift)
            volume = volume * (10 ** log10_vol_shift)
            spec = None                                 # re-scaled audio: spectrogram recomputed on the GPU
        n = c.shape[1]
        if n > 800:
            start = random.randint(0, n - 800)
            end = start + 790
            c, f0, uv = c[:, start:end], f0[start:end], uv[start:end]
            if spec is not None:
                spec = spec[:, start:end]
            audio_norm = audio_norm[:, start * self.hop_length: end * self.hop_length]
            if volume is not None:
                volume = volume[start:end]
        return c, f0, spec, audio_norm, spk, uv, volume

    def __getitem__(self, index):
        if self.all_in_mem:
            return self.random_slice(*self.cache[index])
        return self.random_slice(*self.get_audio(self.audiopaths[index][0]))

    def __len__(self):
        return len(self.audiopaths)


class TextAudioCollate:
    def __call__(self, batch):
        batch = [b for b in batch if b is not None]
        input_lengths, ids_sorted_decreasing = torch.sort(torch.LongTensor([x[0].shape[1] for x in batch]), dim=0,
                                                          descending=True)
        max_c_len = max(x[0].size(1) for x in batch)
        max_wav_len = max(x[3].size(1) for x in batch)
        n = len(batch)
        lengths = torch.LongTensor(n)
        c_padded = torch.zeros(n, batch[0][0].shape[0], max_c_len)
        f0_padded = torch.zeros(n, max_c_len)
        have_spec = all(x[2] is not None for x in batch)
        spec_padded = torch.zeros(n, batch[0][2].shape[0], max_c_len) if have_spec else None
        wav_padded = torch.zeros(n, 1, max_wav_len)
        spkids = torch.LongTensor(n, 1)
        uv_padded = torch.zeros(n, max_c_len)
        volume_padded = torch.zeros(n, max_c_len)
        for i in range(n):
            row = batch[ids_sorted_decreasing[i]]
            c = row[0]
            c_padded[i, :, :c.size(1)] = c
            lengths[i] = c.size(1)
            f0_padded[i, :row[1].size(0)] = row[1]
            if have_spec:
                spec_padded[i, :, :row[2].size(1)] = row[2]
            wav_padded[i, :, :row[3].size(1)] = row[3]
            spkids[i, 0] = row[4]
            uv_padded[i, :row[5].size(0)] = row[5]
            if row[6] is not None and volume_padded is not None:
                volume_padded[i, :row[6].size(0)] = row[6]
            else:
                volume_padded = None
        return c_padded, f0_padded, spec_padded, wav_padded, spkids, lengths, uv_padded, volume_padded


def batch_spectrogram(wav_padded, lengths, n_fft, sampling_rate, hop_size, win_size):
    """Linear spectrogram of a zero-padded batch ON THE GPU, equal to spectrogram_torch of every item alone
    (modules/mel_processing.py:40-64 as the loader calls it, data_utils.py:60-66): wav_padded [B,1,L] (device),
    lengths [B] frames -> [B, n_fft/2+1, max(lengths)], zero beyond each item's length.
    The reflect padding of (n_fft-hop)/2 samples is applied at each item's TRUE end by a gather, then one batched
    framing + rocFFT + magnitude runs over the whole batch."""
    from modules.mel_processing import spectrogram_torch
    B, _, L = wav_padded.shape
    pad = int((n_fft - hop_size) / 2)
    dev = wav_padded.device
    T = int(lengths.max())
    Ls = (lengths.to(dev) * hop_size).view(B, 1)                                  # true sample counts
    n = torch.arange(T * hop_size + pad, device=dev).view(1, -1)
    idx = torch.where(n < Ls, n, 2 * (Ls - 1) - n).clamp_(0, L - 1)               # reflect about each item's last sample
    ext = torch.gather(wav_padded[:, 0], 1, idx)                                   # [B, T*hop + pad]
    # spectrogram_torch reflect-pads `pad` on both sides of what it is given: hand it the signal WITHOUT the right
    # extension's mirror image being re-reflected — T frames only read up to T*hop + pad samples of `ext`
    spec = spectrogram_torch(ext, n_fft, sampling_rate, hop_size, win_size, center=False, n_frames=T)
    keep = (torch.arange(T, device=dev).view(1, 1, T) < lengths.to(dev).view(B, 1, 1)).to(spec.dtype)
    return spec * keep
