"""data_utils mirror (reference data_utils.py:17-186): on-disk formats, item tuple, collate.
CPU: synthetic dataset in the reference's formats -> loader + collate invariants, and (when /root/reference is mounted,
i.e. in the build container) equality with the reference's own TextAudioSpeakerLoader / TextAudioCollate on the same
files.  GPU: batch_spectrogram == per-item spectrogram for ragged lengths."""
import json
import os
import random
import sys
import types

import numpy as np
import pytest
import torch

SR, HOP, NFFT, SSL = 44100, 512, 2048, 24


def _make_dataset(root, n_items=5, with_spec=True, with_vol=False):
    from scipy.io.wavfile import write
    g = torch.Generator().manual_seed(3)
    lines = []
    for i in range(n_items):
        spk = "alice" if i % 2 == 0 else "bob"
        d = os.path.join(root, "dataset", spk)
        os.makedirs(d, exist_ok=True)
        T = 40 + 7 * i
        wav = ((torch.rand(T * HOP + 100 * (i % 2), generator=g) - 0.5) * 20000).to(torch.int16).numpy()
        p = os.path.join(d, f"u{i}.wav")
        write(p, SR, wav)
        torch.save(torch.randn(1, SSL, T // 2 + 1, generator=g), p + ".soft.pt")
        f0 = (100 + 200 * torch.rand(T, generator=g)).numpy()
        f0[:3] = 0
        np.save(p + ".f0.npy", np.asanyarray((f0, (f0 > 0).astype(float)), dtype=object), allow_pickle=True)
        if with_spec:
            torch.save(torch.rand(NFFT // 2 + 1, T + (i % 2), generator=g), p.replace(".wav", ".spec.pt"))
        if with_vol:
            np.save(p + ".vol.npy", torch.rand(T, generator=g).numpy())
        lines.append(p)
    fl = os.path.join(root, "train.txt")
    with open(fl, "w") as f:
        f.write("\n".join(lines) + "\n")
    conf = dict(train=dict(use_sr=True, max_speclen=512, vol_aug=with_vol, segment_size=8192),
                data=dict(max_wav_value=32768.0, sampling_rate=SR, filter_length=NFFT, hop_length=HOP, win_length=NFFT,
                          unit_interpolate_mode="nearest", training_files=fl),
                model=dict(vol_embedding=with_vol), spk=dict(alice=0, bob=1))
    cj = os.path.join(root, "config.json")
    with open(cj, "w") as f:
        json.dump(conf, f)
    return fl, cj


def _ours(fl, cj):
    import data_utils
    import utils
    hps = utils.get_hparams_from_file(cj)
    return data_utils.TextAudioSpeakerLoader(fl, hps), data_utils.TextAudioCollate()


def test_loader_and_collate_invariants(tmp_path):
    fl, cj = _make_dataset(str(tmp_path), with_vol=True)
    ds, collate = _ours(fl, cj)
    assert len(ds) == 5
    random.seed(7)
    items = [ds[i] for i in range(len(ds))]
    for c, f0, spec, wav, spk, uv, vol in items:
        T = c.shape[1]
        assert c.shape[0] == SSL and f0.shape == (T,) and uv.shape == (T,) and wav.shape == (1, T * HOP) and vol.shape == (T,)
        if torch.is_tensor(spec):
            assert spec.shape == (NFFT // 2 + 1, T)
        else:       # vol-augmented item: the samples its T frames read, context included
            assert spec.n_frames == T and spec.ext.shape == (T * HOP + (NFFT - HOP),)
        assert wav.abs().max() <= 1.0 * 10 and int(spk) in (0, 1)
    c_p, f0_p, spec_p, wav_p, spk_p, lengths, uv_p, vol_p = collate(items)
    assert list(lengths) == sorted(lengths.tolist(), reverse=True) and c_p.shape == (5, SSL, int(lengths[0]))
    assert wav_p.shape == (5, 1, int(lengths[0]) * HOP) and spk_p.shape == (5, 1) and vol_p.shape == f0_p.shape
    for i in range(5):
        L = int(lengths[i])
        assert c_p[i, :, L:].abs().sum() == 0 and f0_p[i, L:].abs().sum() == 0 and wav_p[i, 0, L * HOP:].abs().sum() == 0
    if torch.is_tensor(spec_p):
        assert spec_p.shape == (5, NFFT // 2 + 1, int(lengths[0]))
    else:
        n_ctx = sum(1 for it in items if not torch.is_tensor(it[2]))
        assert 0 < n_ctx and int((spec_p.n_frames > 0).sum()) == n_ctx
        assert spec_p.ext.shape == (5, int(spec_p.n_frames.max()) * HOP + (NFFT - HOP))
        assert (spec_p.cached is None) == (n_ctx == 5)
        import data_parallel
        half = data_parallel.shard_batch([c_p[:4], None, spec_p.rows(0, 4)], 1, 2)
        assert half[1] is None and half[0].shape[0] == 2 and half[2].ext.shape[0] == 2


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is only mounted in the build container")
def test_matches_reference_loader_and_collate(tmp_path):
    fl, cj = _make_dataset(str(tmp_path), with_spec=True, with_vol=False)
    ds, collate = _ours(fl, cj)
    random.seed(11)
    ours = collate([ds[i] for i in range(len(ds))])
    # the reference's own classes (utils.py imports faiss / librosa at the top: stub them)
    saved_path, saved_mods = list(sys.path), dict(sys.modules)
    try:
        for name in ("faiss", "librosa", "librosa.filters"):
            sys.modules[name] = types.ModuleType(name)
        sys.modules["librosa.filters"].mel = lambda **kw: None
        for m in ("utils", "data_utils", "modules", "modules.mel_processing", "modules.commons"):
            sys.modules.pop(m, None)
        sys.path.insert(0, "/root/reference")
        import data_utils as RD
        import utils as RU
        hps = RU.get_hparams_from_file(cj)
        rds = RD.TextAudioSpeakerLoader(fl, hps)
        random.seed(11)
        ref = RD.TextAudioCollate()([rds[i] for i in range(len(rds))])
    finally:
        sys.path[:] = saved_path
        for m in list(sys.modules):
            if m not in saved_mods:
                del sys.modules[m]
        sys.modules.update(saved_mods)
    assert len(ours) == len(ref) == 8
    for a, b in zip(ours, ref):
        assert (a is None) == (b is None)
        if a is not None:
            assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.gpu
def test_batch_spectrogram_equals_per_item(dev):
    from data_utils import batch_spectrogram
    from modules.mel_processing import spectrogram_torch
    g = torch.Generator().manual_seed(5)
    lengths = torch.tensor([37, 30, 23, 16])
    L = int(lengths.max()) * HOP
    wav = torch.zeros(4, 1, L)
    for b, n in enumerate(lengths.tolist()):
        wav[b, 0, :n * HOP] = torch.rand(n * HOP, generator=g) - 0.5
    spec = batch_spectrogram(wav.to(dev), lengths.to(dev), NFFT, SR, HOP, NFFT)
    assert spec.shape == (4, NFFT // 2 + 1, 37)
    for b, n in enumerate(lengths.tolist()):
        one = spectrogram_torch(wav[b, :, :n * HOP].to(dev), NFFT, SR, HOP, NFFT)[0]            # [bins, n]
        assert (spec[b, :, :n] - one).abs().max().item() <= 1e-4 * one.abs().max().item()
        assert spec[b, :, n:].abs().max().item() == 0 if n < 37 else True


@pytest.mark.gpu
def test_context_spectrogram_equals_reference_per_item_transform(dev, tmp_path):
    """Items without a cached .spec.pt and vol-augmented items: the reference transforms the WHOLE (re-scaled) utterance in
    the loader worker and slices frames afterwards (data_utils.py:62-66,105-115); the engine ships the samples those frames
    read and transforms them on the GPU.  Checked against the oracle's torch.stft restatement applied the reference's way."""
    import data_utils
    import utils
    from oracle import train_oracle as TO
    fl, cj = _make_dataset(str(tmp_path), n_items=4, with_spec=False, with_vol=True)
    # make two utterances long enough to be cropped (> 800 frames)
    from scipy.io.wavfile import write
    g = torch.Generator().manual_seed(9)
    for i, p in enumerate(open(fl).read().split()[:2]):
        T = 900 + 20 * i
        write(p, SR, ((torch.rand(T * HOP + 37, generator=g) - 0.5) * 20000).to(torch.int16).numpy())
        torch.save(torch.randn(1, SSL, T // 2 + 1, generator=g), p + ".soft.pt")
        f0 = (100 + 200 * torch.rand(T, generator=g)).numpy()
        np.save(p + ".f0.npy", np.asanyarray((f0, (f0 > 0).astype(float)), dtype=object), allow_pickle=True)
        np.save(p + ".vol.npy", torch.rand(T, generator=g).numpy())
    hps = utils.get_hparams_from_file(cj)
    ds = data_utils.TextAudioSpeakerLoader(fl, hps)
    random.seed(3)
    items = [ds[i] for i in range(len(ds))]
    assert all(isinstance(it[2], data_utils.SpecContext) for it in items)
    c_p, f0_p, spec_p, wav_p, spk_p, lengths, uv_p, vol_p = data_utils.TextAudioCollate()(items)
    spec = data_utils.context_spectrogram(spec_p.cuda(), NFFT, SR, HOP, NFFT)
    assert spec.shape == (4, NFFT // 2 + 1, int(lengths[0]))
    # reference semantics, item by item, from the files: same RNG stream -> same vol shifts / crops
    random.seed(3)
    order = torch.sort(torch.LongTensor([it[0].shape[1] for it in items]), descending=True)[1].tolist()
    refs = []
    for i in range(len(ds)):
        c, f0, sp, audio, spk, uv, vol = ds.get_audio(ds.audiopaths[i][0])
        full = sp.full
        T = c.shape[1]
        if random.choice([True, False]) and ds.vol_aug and vol is not None:
            max_amp = float(torch.max(torch.abs(audio))) + 1e-5
            # the reference re-transforms `audio_norm`, which get_audio has ALREADY cut to lmin * hop samples (data_utils.py:96-110)
            full = audio * (10 ** random.uniform(-1, min(1, np.log10(1 / max_amp))))
        ref = TO.spectrogram(full, NFFT, HOP, NFFT)[0][:, :T]
        if T > 800:
            s0 = random.randint(0, T - 800)
            ref = ref[:, s0:s0 + 790]
        refs.append(ref)
    for row, i in enumerate(order):
        n = refs[i].shape[1]
        assert n == int(lengths[row])
        err = (spec[row, :, :n].cpu() - refs[i]).abs().max().item()
        assert err <= 2e-4 * refs[i].abs().max().item(), (row, err)
        assert spec[row, :, n:].abs().max().item() == 0 if n < spec.shape[2] else True


def test_repeat_expand_2d_left_is_the_reference_sequential_fill():
    """utils.repeat_expand_2d('left') vectorised == the reference's frame loop (utils.py:402-416, restated here), including
    target_len < src_len where the reference's fill advances at most one source column per frame."""
    import utils as U

    def ref_loop(content, target_len):
        src_len = content.shape[-1]
        target = torch.zeros([content.shape[0], target_len], dtype=torch.float)
        temp = torch.arange(src_len + 1) * target_len / src_len
        cur = 0
        for i in range(target_len):
            if i < temp[cur + 1]:
                target[:, i] = content[:, cur]
            else:
                cur += 1
                target[:, i] = content[:, cur]
        return target
    g = torch.Generator().manual_seed(0)
    for src, tgt in [(50, 86), (499, 862), (7, 7), (1, 5), (3, 100), (100, 37), (862, 500), (10, 9), (33, 34), (250, 431)]:
        x = torch.randn(4, src, generator=g)
        assert torch.equal(U.repeat_expand_2d(x, tgt), ref_loop(x, tgt)), (src, tgt)
    assert U.repeat_expand_2d(torch.randn(3, 10), 25, mode="nearest").shape == (3, 25)


def test_bucketed_collate_is_the_plain_collate_zero_padded(tmp_path):
    """train.run's collate (hipGraph training needs a handful of padded shapes): frame axis padded to data_utils.FRAME_BUCKETS,
    waveform to bucket * hop, everything else — order, lengths, values — as the reference's collate gives it."""
    import data_utils
    fl, cj = _make_dataset(str(tmp_path), n_items=6, with_spec=True, with_vol=True)
    ds, plain = _ours(fl, cj)
    bucketed = data_utils.TextAudioCollate(buckets=data_utils.FRAME_BUCKETS, hop_length=HOP)
    random.seed(5)
    items = [ds[i] for i in range(6)]
    a, b = plain(items), bucketed(items)
    T, Tb = a[0].shape[2], b[0].shape[2]
    assert Tb == data_utils.bucket_frames(T) and Tb in data_utils.FRAME_BUCKETS and Tb >= T and b[3].shape[2] == Tb * HOP
    assert torch.equal(a[5], b[5]) and torch.equal(a[4], b[4])                       # lengths, speaker ids: same order
    for x, y in zip(a, b):
        if torch.is_tensor(x) and x.dim() >= 2 and x.shape[-1] in (T, a[3].shape[2]):
            n = x.shape[-1]
            assert torch.equal(x, y[..., :n]) and float(y[..., n:].abs().sum()) == 0.0
    # items without a cached spectrogram: the context rows grow with the bucket, by whole frames
    fl2, cj2 = _make_dataset(str(tmp_path / "b"), n_items=4, with_spec=False)
    ds2, plain2 = _ours(fl2, cj2)
    random.seed(6)
    items2 = [ds2[i] for i in range(4)]
    a2, b2 = plain2(items2), bucketed(items2)
    assert isinstance(b2[2], data_utils.SpecContextBatch) and torch.equal(a2[2].n_frames, b2[2].n_frames)
    Tb2 = data_utils.bucket_frames(int(a2[2].n_frames.max()))
    assert b2[2].ext.shape[1] == Tb2 * HOP + (NFFT - HOP) and torch.equal(a2[2].ext, b2[2].ext[:, :a2[2].ext.shape[1]])
    assert [data_utils.bucket_frames(n) for n in (1, 128, 129, 200, 320, 321, 790, 800, 801, 1000)] == [128, 128, 192, 256, 320, 448, 800, 800, 896, 1024]
    with pytest.raises(ValueError):
        data_utils.TextAudioCollate(buckets=(64,))
