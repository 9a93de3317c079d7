"""Mirror of diffusion/data_loaders.py: `AudioDataset` + `get_data_loaders` for `train_diff.py`.

Same on-disk formats next to every listed wav (`.f0.npy` object pair, `.vol.npy`, `.aug_vol.npy`, `.mel.npy`, `.aug_mel.npy`
object pair (mel, keyshift), `.soft.pt` [1, C, T50]), same item dict (`mel, f0, volume, units, spk_id, aug_shift, name,
name_ext`), same random crop of `duration` seconds and the same 50 % pitch-augmented choice (:177-246).  What differs:
  * the clip duration comes from the wav header (`librosa.get_duration`, :132, is the only librosa call; librosa is not a
    dependency of the engine),
  * `get_data_loaders(..., rank, world)` shards the TRAINING file list over data-parallel ranks (the reference's script is
    single-GPU; BASELINE configs[4] runs it on 8) — validation stays whole on every rank that asks for it.
"""
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from utils import repeat_expand_2d

from .logger.utils import traverse_dir  # noqa: F401  (re-exported: the reference module defines it too)


def audio_duration(path, sample_rate=None):
    """Seconds of audio in a wav file, from its header."""
    try:
        import soundfile
        info = soundfile.info(path)
        return info.frames / info.samplerate
    except ImportError:
        pass
    import wave
    try:
        with wave.open(path, "rb") as w:
            return w.getnframes() / w.getframerate()
    except wave.Error:                                  # float / extensible wav: let scipy parse it
        from scipy.io import wavfile
        sr, data = wavfile.read(path, mmap=True)
        return data.shape[0] / sr


class _RankSampler(torch.utils.data.Sampler):
    """A fresh permutation per epoch (the reference's `shuffle=True`), identical on every rank, rank r taking every
    world-th index of it; wrapped around so that all ranks see the same number of batches."""

    def __init__(self, n_items, rank, world, seed=0):
        self.n, self.rank, self.world, self.seed, self.epoch = n_items, rank, world, seed, 0

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.seed + self.epoch)
        self.epoch += 1
        order = torch.randperm(self.n, generator=g).tolist()
        per = -(-self.n // self.world)
        order = (order * (per * self.world // max(self.n, 1) + 1))[:per * self.world]
        return iter(order[self.rank::self.world])

    def __len__(self):
        return -(-self.n // self.world)


def get_data_loaders(args, whole_audio=False, rank=0, world=1):
    data_train = AudioDataset(filelists=args.data.training_files, waveform_sec=args.data.duration, hop_size=args.data.block_size,
                              sample_rate=args.data.sampling_rate, load_all_data=args.train.cache_all_data,
                              whole_audio=whole_audio, extensions=args.data.extensions, n_spk=args.model.n_spk, spk=args.spk,
                              device=args.train.cache_device, fp16=args.train.cache_fp16,
                              unit_interpolate_mode=args.data.unit_interpolate_mode, use_aug=True)
    on_cpu = args.train.cache_device == "cpu"
    workers = int(os.environ.get("SVC_LOADER_WORKERS", args.train.num_workers)) if on_cpu else 0
    sampler = _RankSampler(len(data_train), rank, world) if world > 1 else None
    loader_train = torch.utils.data.DataLoader(data_train, batch_size=args.train.batch_size if not whole_audio else 1,
                                               shuffle=sampler is None, sampler=sampler, num_workers=workers,
                                               persistent_workers=workers > 0, pin_memory=on_cpu)
    data_valid = AudioDataset(filelists=args.data.validation_files, waveform_sec=args.data.duration, hop_size=args.data.block_size,
                              sample_rate=args.data.sampling_rate, load_all_data=args.train.cache_all_data, whole_audio=True,
                              spk=args.spk, extensions=args.data.extensions, unit_interpolate_mode=args.data.unit_interpolate_mode,
                              n_spk=args.model.n_spk)
    loader_valid = torch.utils.data.DataLoader(data_valid, batch_size=1, shuffle=False, num_workers=0, pin_memory=True)
    return loader_train, loader_valid


class AudioDataset(Dataset):
    def __init__(self, filelists, waveform_sec, hop_size, sample_rate, spk, load_all_data=True, whole_audio=False,
                 extensions=["wav"], n_spk=1, device="cpu", fp16=False, use_aug=False, unit_interpolate_mode="left"):
        super().__init__()
        self.waveform_sec, self.sample_rate, self.hop_size = waveform_sec, sample_rate, hop_size
        self.filelists, self.whole_audio, self.use_aug = filelists, whole_audio, use_aug
        self.data_buffer, self.pitch_aug_dict = {}, {}
        self.unit_interpolate_mode = unit_interpolate_mode
        print(("Load all the data filelists:" if load_all_data else "Load the f0, volume data filelists:"), filelists)
        with open(filelists, "r") as f:
            self.paths = f.read().splitlines()
        for name_ext in self.paths:
            duration = audio_duration(name_ext, self.sample_rate)
            f0, _ = np.load(name_ext + ".f0.npy", allow_pickle=True)
            f0 = torch.from_numpy(np.array(f0, dtype=float)).float().unsqueeze(-1).to(device)
            volume = torch.from_numpy(np.load(name_ext + ".vol.npy")).float().unsqueeze(-1).to(device)
            aug_vol = torch.from_numpy(np.load(name_ext + ".aug_vol.npy")).float().unsqueeze(-1).to(device)
            if n_spk is not None and n_spk > 1:
                spk_name = name_ext.split("/")[-2]
                spk_id = spk[spk_name] if spk_name in spk else 0
                if spk_id < 0 or spk_id >= n_spk:
                    raise ValueError(" [x] Muiti-speaker traing error : spk_id must be a positive integer from 0 to n_spk-1 ")
            else:
                spk_id = 0
            spk_id = torch.LongTensor(np.array([spk_id])).to(device)
            aug_mel, keyshift = np.load(name_ext + ".aug_mel.npy", allow_pickle=True)
            self.pitch_aug_dict[name_ext] = keyshift
            entry = dict(duration=duration, f0=f0, volume=volume, aug_vol=aug_vol, spk_id=spk_id)
            if load_all_data:
                mel = torch.from_numpy(np.load(name_ext + ".mel.npy")).to(device)
                aug_mel = torch.from_numpy(np.array(aug_mel, dtype=float)).to(device)
                units = torch.load(name_ext + ".soft.pt").to(device)[0]
                units = repeat_expand_2d(units, f0.size(0), unit_interpolate_mode).transpose(0, 1)
                if fp16:
                    mel, aug_mel, units = mel.half(), aug_mel.half(), units.half()
                entry.update(mel=mel, aug_mel=aug_mel, units=units)
            self.data_buffer[name_ext] = entry

    def __getitem__(self, file_idx):
        name_ext = self.paths[file_idx]
        buf = self.data_buffer[name_ext]
        if buf["duration"] < (self.waveform_sec + 0.1):          # too short: skip to the next file (:168-169)
            return self.__getitem__((file_idx + 1) % len(self.paths))
        return self.get_data(name_ext, buf)

    def get_data(self, name_ext, buf):
        name = os.path.splitext(name_ext)[0]
        frame_resolution = self.hop_size / self.sample_rate
        duration = buf["duration"]
        waveform_sec = duration if self.whole_audio else self.waveform_sec
        idx_from = 0 if self.whole_audio else random.uniform(0, duration - waveform_sec - 0.1)
        start = int(idx_from / frame_resolution)
        n = int(waveform_sec / frame_resolution)
        aug_flag = random.choice([True, False]) and self.use_aug
        mel = buf.get("aug_mel" if aug_flag else "mel")
        if mel is None:                                          # not cached: the PLAIN mel, also for augmented items (:209-214)
            mel = torch.from_numpy(np.load(name_ext + ".mel.npy")[start:start + n]).float()
        else:
            mel = mel[start:start + n]
        f0 = buf.get("f0")
        aug_shift = self.pitch_aug_dict[name_ext] if aug_flag else 0
        f0_frames = 2 ** (aug_shift / 12) * f0[start:start + n]
        units = buf.get("units")
        if units is None:
            units = torch.load(name_ext + ".soft.pt")[0]
            units = repeat_expand_2d(units, f0.size(0), self.unit_interpolate_mode).transpose(0, 1)
        units = units[start:start + n]
        volume_frames = buf.get("aug_vol" if aug_flag else "volume")[start:start + n]
        aug_shift = torch.from_numpy(np.array([[aug_shift]])).float()
        return dict(mel=mel, f0=f0_frames, volume=volume_frames, units=units, spk_id=buf.get("spk_id"), aug_shift=aug_shift,
                    name=name, name_ext=name_ext)

    def __len__(self):
        return len(self.paths)
