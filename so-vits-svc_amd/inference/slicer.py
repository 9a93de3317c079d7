"""Mirror of inference/slicer.py (the silence slicer in front of `Svc.slice_inference`, inference/infer_tool.py:380-381):
same `Slicer`, `cut` and `chunks2audio` names, arguments and return formats, without the librosa / torchaudio imports
(svc_audio.frame_rms restates librosa 0.9.1's centred RMS, svc_audio.read_audio decodes the wav).  Host-side integer
bookkeeping over a few thousand RMS frames — not kernel work.

Semantics (inference/slicer.py:6-116): frames quieter than `threshold` dB open a silence; a silence is CUT OUT when it is
the leading one and longer than `max_sil_kept`, or when it lasted >= `min_interval` and the clip before it >= `min_length`;
the cut points are the quietest frames near its ends, so at most `max_sil_kept` of silence stays on either side.
"""
import numpy as np

import svc_audio


class Slicer:
    def __init__(self, sr: int, threshold: float = -40., min_length: int = 5000, min_interval: int = 300,
                 hop_size: int = 20, max_sil_kept: int = 5000):
        if not min_length >= min_interval >= hop_size:
            raise ValueError('The following condition must be satisfied: min_length >= min_interval >= hop_size')
        if not max_sil_kept >= hop_size:
            raise ValueError('The following condition must be satisfied: max_sil_kept >= hop_size')
        interval_samples = sr * min_interval / 1000
        self.threshold = 10 ** (threshold / 20.)
        self.hop_size = round(sr * hop_size / 1000)
        self.win_size = min(round(interval_samples), 4 * self.hop_size)
        self.min_length = round(sr * min_length / 1000 / self.hop_size)
        self.min_interval = round(interval_samples / self.hop_size)
        self.max_sil_kept = round(sr * max_sil_kept / 1000 / self.hop_size)

    def _apply_slice(self, waveform, begin, end):
        lo, hi = begin * self.hop_size, end * self.hop_size
        if waveform.ndim > 1:
            return waveform[:, lo:min(waveform.shape[1], hi)]
        return waveform[lo:min(waveform.shape[0], hi)]

    def _silence_tags(self, rms):
        """[(first_cut_frame, last_cut_frame)] of every silence that gets removed (slicer.py:42-93)."""
        keep = self.max_sil_kept

        def quietest(lo, hi):              # frame index of the minimum of rms[lo:hi+1]
            return int(rms[lo:hi + 1].argmin()) + lo

        tags = []
        start = None                       # first frame of the silence being tracked
        clip_start = 0
        for i, level in enumerate(rms):
            if level < self.threshold:
                if start is None:
                    start = i
                continue
            if start is None:
                continue
            length = i - start
            leading = start == 0 and i > keep
            middle = length >= self.min_interval and i - clip_start >= self.min_length
            if leading or middle:
                if length <= keep:
                    cut = quietest(start, i)
                    tags.append((0, cut) if start == 0 else (cut, cut))
                    clip_start = cut
                else:
                    left = quietest(start, start + keep)
                    right = quietest(i - keep, i)
                    if length <= 2 * keep:
                        mid = quietest(i - keep, start + keep)
                        left, right = min(left, mid), max(right, mid)
                        if start == 0:
                            right = quietest(i - keep, i)
                    tags.append((0, right) if start == 0 else (left, right))
                    clip_start = right
            start = None
        total = rms.shape[0]
        if start is not None and total - start >= self.min_interval:
            tags.append((quietest(start, min(total, start + keep)), total + 1))
        return tags

    def slice(self, waveform):
        samples = waveform.mean(axis=0) if waveform.ndim > 1 else waveform
        whole = {"0": {"slice": False, "split_time": f"0,{len(waveform)}"}}
        if samples.shape[0] <= self.min_length:
            return whole
        rms = svc_audio.frame_rms(samples, self.win_size, self.hop_size)
        tags = self._silence_tags(rms)
        if not tags:
            return whole
        n = waveform.shape[0]
        hop = self.hop_size
        spans = []                                              # (is_silence, "begin,end")
        if tags[0][0]:
            spans.append((False, f"0,{min(n, tags[0][0] * hop)}"))
        for k, (a, b) in enumerate(tags):
            if k:
                spans.append((False, f"{tags[k - 1][1] * hop},{min(n, a * hop)}"))
            spans.append((True, f"{a * hop},{min(n, b * hop)}"))
        if tags[-1][1] * hop < len(waveform):
            spans.append((False, f"{tags[-1][1] * hop},{len(waveform)}"))
        return {str(k): {"slice": sil, "split_time": span} for k, (sil, span) in enumerate(spans)}


def cut(audio_path, db_thresh=-30, min_len=5000):
    audio, sr = svc_audio.read_audio(audio_path)
    audio = audio.mean(axis=0) if audio.shape[0] > 1 else audio[0]          # librosa.load(mono=True)
    return Slicer(sr=sr, threshold=db_thresh, min_length=min_len).slice(audio)


def chunks2audio(audio_path, chunks):
    chunks = dict(chunks)
    audio, sr = svc_audio.read_audio(audio_path)
    audio = audio.mean(axis=0) if audio.shape[0] > 1 else audio[0]
    result = []
    for _, v in chunks.items():
        lo, hi = v["split_time"].split(",")
        if lo != hi:
            result.append((v["slice"], audio[int(lo):int(hi)]))
    return result, sr
