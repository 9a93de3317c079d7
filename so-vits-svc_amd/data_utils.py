"""MI355X-side mirror of data_utils.py (TextAudioSpeakerLoader :17-128, TextAudioCollate :131-186) — SURVEY.md §8f row 3.

Same on-disk formats as the reference's preprocessing (preprocess_hubert_f0.py:31-103): `x.wav` (int16 / float PCM at
the configured rate), `x.wav.soft.pt` (torch.save, units [1, ssl_dim, T50]), `x.wav.f0.npy` (object array (f0, uv)),
`x.spec.pt` ([n_fft/2+1, T]), `x.wav.vol.npy`; same item tuple `(c, f0, spec, audio_norm, spk, uv, volume)` and the same
collate output `(c, f0, spec, wav, spkids, lengths, uv, volume)` sorted by decreasing length, zero padded.

What is re-designed: the reference computes a missing / re-scaled (vol_aug) linear spectrogram with torch.stft INSIDE
the DataLoader worker (:60-66,98-103).  Here the worker only does I/O; such an item carries a `SpecContext` in its `spec`
slot — exactly the samples the reference's STFT frames of that item read: the reference transforms the WHOLE (re-scaled)
utterance and slices frames afterwards (:105-115), so the frames at a crop edge see the real neighbouring audio and only the
utterance's true ends are reflect-padded — the collate returns a `SpecContextBatch` instead of `spec_padded`, and
`batch_spectrogram` turns it into the [B, bins, T] spectrogram on the GPU in one batched rocFFT call.  That keeps 8 GPUs fed
without an 8x CPU STFT load and equals the reference's per-item result.  (The reference also writes a missing `.spec.pt`
back to disk, :66; the engine does not.)  Everything else is host-side plumbing on torch CPU tensors.
"""
import os
import random

import numpy as np
import torch
import torch.utils.data

import utils


class SpecContext:
    """Stands in for a linear spectrogram that still has to be computed (on the GPU).  `full` [1, L]: the utterance the
    reference would transform; `ext` [n_frames*hop + (n_fft - hop)]: after cropping, the samples frames [start, end) of that
    transform read (reflect padding only where the utterance really ends)."""

    def __init__(self, full=None, ext=None, n_frames=None):
        self.full, self.ext, self.n_frames = full, ext, n_frames

    def crop(self, start, end, hop, n_fft, scale=1.0):
        pad = int((n_fft - hop) / 2)
        x = self.full * scale if scale != 1.0 else self.full
        padded = torch.nn.functional.pad(x.unsqueeze(0), (pad, pad), mode="reflect")[0, 0]
        need = (end - start) * hop + 2 * pad
        ext = padded[start * hop:start * hop + need]
        if ext.shape[0] < need:                                   # audio a few samples short of its frame count
            ext = torch.nn.functional.pad(ext, (0, need - ext.shape[0]))
        return SpecContext(ext=ext.contiguous(), n_frames=end - start)


class SpecContextBatch:
    """Collated `spec` slot of a minibatch in which at least one item still needs its spectrogram: ext [B, max_frames*hop +
    2*pad] (zero rows for items that came with a cached spectrogram), n_frames [B] (0 for those), cached [B, bins, T]
    (zero rows for the others; None when no item was cached).  `.cuda()` / `.to()` / `.pin_memory()` like a tensor."""

    def __init__(self, ext, n_frames, cached=None):
        self.ext, self.n_frames, self.cached = ext, n_frames, cached

    def _map(self, f):
        return SpecContextBatch(f(self.ext), self.n_frames, None if self.cached is None else f(self.cached))

    def to(self, *a, **k):
        return self._map(lambda t: t.to(*a, **k))

    def cuda(self, *a, **k):
        return self._map(lambda t: t.cuda(*a, **k))

    def pin_memory(self):
        return self._map(lambda t: t.pin_memory())

    def rows(self, lo, hi):
        return SpecContextBatch(self.ext[lo:hi], self.n_frames[lo:hi], None if self.cached is None else self.cached[lo:hi])


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
        # no cached spectrogram: computed on the GPU from the utterance as the reference transforms it (:62-64)
        spec = torch.load(spec_filename) if os.path.exists(spec_filename) else SpecContext(full=audio_norm)
        spk = filename.split("/")[-2]
        spk = torch.LongTensor([self.spk_map[spk]])
        f0, uv = np.load(filename + ".f0.npy", allow_pickle=True)
        f0 = torch.FloatTensor(np.array(f0, dtype=float))
        uv = torch.FloatTensor(np.array(uv, dtype=float))
        c = torch.load(filename + ".soft.pt")
        c = utils.repeat_expand_2d(c.squeeze(0), f0.shape[0], mode=self.unit_interpolate_mode)
        volume = torch.from_numpy(np.load(filename + ".vol.npy")).float() if self.vol_emb else None
        n_spec = spec.size(-1) if torch.is_tensor(spec) else audio_norm.shape[1] // self.hop_length
        lmin = min(c.size(-1), n_spec)
        assert abs(c.size(-1) - n_spec) < 3, (c.size(-1), n_spec, f0.shape, filename)
        assert abs(audio_norm.shape[1] - lmin * self.hop_length) < 3 * self.hop_length
        c, f0, uv = c[:, :lmin], f0[:lmin], uv[:lmin]
        if torch.is_tensor(spec):
            spec = spec[:, :lmin]
        audio_norm = audio_norm[:, :lmin * self.hop_length]
        if not torch.is_tensor(spec):
            spec.n_frames = lmin
        if volume is not None:
            volume = volume[:lmin]
        return c, f0, spec, audio_norm, spk, uv, volume

    def random_slice(self, c, f0, spec, audio_norm, spk, uv, volume):
        if random.choice([True, False]) and self.vol_aug and volume is not None:
            max_amp = float(torch.max(torch.abs(audio_norm))) + 1e-5
            max_shift = min(1, np.log10(1 / max_amp))
            log10_vol_shift = random.uniform(-1, max_shift)
            scale = 10 ** log10_vol_shift
            # re-scaled audio: the reference re-transforms the re-scaled utterance AFTER cutting it to lmin * hop samples
            # (data_utils.py:96-110), so the last frames see reflect padding, not trailing samples; done on the GPU here
            full = audio_norm * scale
            audio_norm = audio_norm * scale
            volume = volume * scale
            spec = SpecContext(full=full, n_frames=c.shape[1])
        n = c.shape[1]
        start, end = 0, n
        if n > 800:
            start = random.randint(0, n - 800)
            end = start + 790
            c, f0, uv = c[:, start:end], f0[start:end], uv[start:end]
            if torch.is_tensor(spec):
                spec = spec[:, start:end]
            audio_norm = audio_norm[:, start * self.hop_length: end * self.hop_length]
            if volume is not None:
                volume = volume[start:end]
        if isinstance(spec, SpecContext):
            spec = spec.crop(start, end, self.hop_length, self.filter_length)
        return c, f0, spec, audio_norm, spk, uv, volume

    def __getitem__(self, index):
        if self.all_in_mem:
            return self.random_slice(*self.cache[index])
        return self.random_slice(*self.get_audio(self.audiopaths[index][0]))

    def __len__(self):
        return len(self.audiopaths)


# padded frame counts of a bucketed batch; beyond the last: multiples of 128.  The entries below 320 serve data sets of short
# clips and the short last batch of an epoch (padding a 100-frame batch to 320 tripled its step time); with
# drop_last=False they can add captured shapes, which TrainStep's graph cache bounds (least-recently-used eviction).
FRAME_BUCKETS = (128, 192, 256, 320, 448, 576, 672, 736, 800)


def bucket_frames(n_frames, buckets=FRAME_BUCKETS):
    """Smallest bucket >= n_frames.  The loader's crops are at most 800 frames long (data_utils.py:112-118: utterances above
    800 frames are cut to 790) and a batch is as long as its longest item, so with B = 16 nearly every batch lands in the last
    one or two buckets; the lower ones serve small batches and short data sets."""
    for b in buckets:
        if n_frames <= b:
            return b
    return -(-n_frames // 128) * 128


class TextAudioCollate:
    """data_utils.py:131-186.  `buckets` (engine addition, None = the reference's behaviour): pad the frame axis to
    `bucket_frames(longest item)` instead of the longest item itself, and the waveform to that many hops — the padding is zeros
    and every consumer masks by `lengths`, so the batch means the same; what changes is that a run sees a handful of padded
    shapes instead of a new one per batch, which is what lets train.TrainStep replay the iteration from one hipGraph per shape."""

    def __init__(self, buckets=None, hop_length=None):
        self.buckets, self.hop = buckets, hop_length
        if buckets is not None and not hop_length:
            raise ValueError("TextAudioCollate(buckets=...) needs hop_length (the waveform is padded to bucket * hop samples)")

    def __call__(self, batch):
        batch = [b for b in batch if b is not None]
        input_lengths, ids_sorted_decreasing = torch.sort(torch.LongTensor([x[0].shape[1] for x in batch]), dim=0,
                                                          descending=True)
        max_c_len = max(x[0].size(1) for x in batch)
        max_wav_len = max(x[3].size(1) for x in batch)
        if self.buckets is not None:
            max_c_len = bucket_frames(max_c_len, self.buckets)
            max_wav_len = max(max_wav_len, max_c_len * self.hop)
        n = len(batch)
        lengths = torch.LongTensor(n)
        c_padded = torch.zeros(n, batch[0][0].shape[0], max_c_len)
        f0_padded = torch.zeros(n, max_c_len)
        have_spec = all(torch.is_tensor(x[2]) for x in batch)
        if have_spec:
            spec_padded = torch.zeros(n, batch[0][2].shape[0], max_c_len)
        else:       # vol_aug re-scales a random half of the items: cached and to-be-computed spectrograms share a batch
            ext_len = max(x[2].ext.shape[0] for x in batch if isinstance(x[2], SpecContext))
            if self.buckets is not None:          # frames * hop + 2 * pad: the same context margin on the bucketed frame count
                ctx = next(x[2] for x in batch if isinstance(x[2], SpecContext))
                ext_len = max(ext_len, max_c_len * self.hop + (ctx.ext.shape[0] - ctx.n_frames * self.hop))
            ext_padded = torch.zeros(n, ext_len)
            ext_frames = torch.zeros(n, dtype=torch.long)
            cached = [x[2] for x in batch if torch.is_tensor(x[2])]
            spec_cached = torch.zeros(n, cached[0].shape[0], max_c_len) if cached else None
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
            elif torch.is_tensor(row[2]):
                spec_cached[i, :, :row[2].size(1)] = row[2]
            else:
                ext_padded[i, :row[2].ext.shape[0]] = row[2].ext
                ext_frames[i] = row[2].n_frames
            wav_padded[i, :, :row[3].size(1)] = row[3]
            spkids[i, 0] = row[4]
            uv_padded[i, :row[5].size(0)] = row[5]
            if row[6] is not None and volume_padded is not None:
                volume_padded[i, :row[6].size(0)] = row[6]
            else:
                volume_padded = None
        if not have_spec:
            spec_padded = SpecContextBatch(ext_padded, ext_frames, spec_cached)
        return c_padded, f0_padded, spec_padded, wav_padded, spkids, lengths, uv_padded, volume_padded


def context_spectrogram(ctx, n_fft, sampling_rate, hop_size, win_size):
    """SpecContextBatch (on the device) -> [B, n_fft/2+1, max frames], zero beyond each item's frame count: one batched
    framing + rocFFT + magnitude over signals that already carry their (n_fft-hop)/2 context on both sides."""
    from modules.mel_processing import spectrogram_torch
    T = int(ctx.n_frames.max())
    spec = spectrogram_torch(ctx.ext, n_fft, sampling_rate, hop_size, win_size, center=False, n_frames=T, prepadded=True)
    dev = spec.device
    keep = (torch.arange(T, device=dev).view(1, 1, T) < ctx.n_frames.to(dev).view(-1, 1, 1)).to(spec.dtype)
    spec = spec * keep
    if ctx.cached is not None:          # rows that came with a cached spectrogram have n_frames = 0 (all-zero rows above)
        Tc = ctx.cached.shape[2]
        if Tc > T:
            spec = torch.nn.functional.pad(spec, (0, Tc - T))
        spec = spec.clone()
        spec[:, :, :Tc] += ctx.cached.to(dev)
    return spec


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
