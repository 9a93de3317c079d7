"""The drop-in boundary on the GPU:

* the CALL SEQUENCE of the reference's `inference_main.main` (inference_main.py:94-152: `Svc(...)` with its eleven
  positional arguments, `infer_tool.mkdir`, `fill_a_to_b`, `format_wav`, `svc_model.slice_inference(**kwarg)`,
  `svc_model.clear_empty()`) executed end to end on a synthetic wav, with the unit encoder and the f0 predictor replaced the way
  one would replace them for the reference itself — by patching the `utils.get_speech_encoder` / `utils.get_f0_predictor`
  factories `Svc` calls (infer_tool.py:166,207).  The silence slicer, wav decoding, 44.1 -> 16 kHz resampling and the
  synthesizer are the engine's.  Checked against a by-hand recomputation chunk by chunk and against the CPU oracle;
* chunk batching (`batch_chunks=True`) against the serial loop;
* the HIP sinc resampler against the oracle restatement of torchaudio's algorithm;
* `ContentVec768L12` / `ContentVec256L9` loaded from a fairseq-named checkpoint against the HuBERT oracle.

Nothing here reads /root/reference."""
import os

import numpy as np
import pytest
import torch

from oracle import audio_oracle as AO
from oracle import hubert_oracle as HO
from oracle import svc_oracle as O
from oracle import weights as W
from test_svc_gpu import _write_model

pytestmark = pytest.mark.gpu
SR, HOP = 44100, 512


class FakeEncoder:
    """vencoder.encoder.SpeechEncoder interface; units are a fixed random projection of the 16 kHz frames, so they depend
    on what the engine's resampler hands over."""

    def __init__(self, ssl_dim, dev):
        g = torch.Generator().manual_seed(77)
        self.proj = (torch.randn(ssl_dim, 320, generator=g) / 6.0).to(dev)
        self.hidden_dim = ssl_dim
        self.calls = 0

    def encoder(self, wav16k):
        self.calls += 1
        n = wav16k.shape[-1] // 320
        frames = wav16k[:n * 320].view(n, 320).t()
        return (self.proj @ frames).unsqueeze(0)                  # [1, ssl_dim, n]


class FakeF0:
    name = "pm"

    def __init__(self, hop_length, sampling_rate):
        self.hop = hop_length

    def compute_f0_uv(self, wav):
        n = len(wav) // self.hop
        t = np.arange(n)
        f0 = 200.0 + 80.0 * np.sin(t / 11.0)
        f0[(t // 17) % 4 == 0] = 0.0
        return f0, (f0 > 0).astype(np.float32)


def _song(seconds_voiced=(1.3, 0.9), gap=1.0, lead=0.0):
    g = np.random.default_rng(3)
    parts = [np.zeros(int(SR * lead))] if lead else []
    for i, s in enumerate(seconds_voiced):
        n = int(SR * s)
        parts.append(0.25 * np.sin(2 * np.pi * 220.0 * (i + 1) * np.arange(n) / SR) + 0.05 * g.standard_normal(n))
        if i + 1 < len(seconds_voiced):
            parts.append(1e-5 * g.standard_normal(int(SR * gap)))
    return np.concatenate(parts).astype(np.float32)


@pytest.fixture
def patched_factories(monkeypatch, dev):
    import utils
    made = {}

    def get_speech_encoder(name, device=None, **kw):
        made["enc"] = FakeEncoder(made["ssl_dim"], dev)
        made["enc_name"] = name
        return made["enc"]

    def get_f0_predictor(name, hop_length, sampling_rate, **kw):
        made["f0_args"] = (name, hop_length, sampling_rate, sorted(kw))
        return FakeF0(hop_length, sampling_rate)

    monkeypatch.setattr(utils, "get_speech_encoder", get_speech_encoder)
    monkeypatch.setattr(utils, "get_f0_predictor", get_f0_predictor)
    return made


def test_inference_main_call_sequence(dev, tmp_path, monkeypatch, patched_factories):
    import svc_audio
    import utils
    from inference import infer_tool
    from inference.infer_tool import Svc
    cfg = W.small_config()
    patched_factories["ssl_dim"] = cfg["ssl_dim"]
    monkeypatch.chdir(tmp_path)
    os.makedirs("logs/44k")
    net, ck, cj = _write_model("logs/44k", cfg, 9)
    conf = utils.get_hparams_from_file(cj)
    os.makedirs("raw")
    # the slicer only cuts after >= 5 s of voiced audio (min_len) and only REMOVES the middle of a silence longer than 2 x max_sil_kept = 10 s
    wav = _song(seconds_voiced=(5.4, 1.1), gap=11.0)
    svc_audio.write_wav("raw/song.wav", wav, SR)
    chunks_dict = infer_tool.read_temp(str(tmp_path / "chunks_temp.json"))           # inference_main.py:10
    assert chunks_dict == {}

    # ---- inference_main.py:94-152, argument for argument (args.* replaced by the parser's defaults / our paths) ----
    clean_names, trans, spk_list = ["song.wav"], [0], ["bob"]
    slice_db, wav_format, auto_predict_f0, cluster_infer_ratio, noice_scale, pad_seconds = -40, "wav", False, 0, 0.4, 0.5
    clip, lg, lgr, f0p, enhance, enhancer_adaptive_key, cr_threshold = 0, 0, 0.75, "pm", False, 0, 0.05
    diffusion_model_path, diffusion_config_path = "logs/44k/diffusion/model_0.pt", "logs/44k/diffusion/config.yaml"
    k_step, only_diffusion, shallow_diffusion, use_spk_mix, second_encoding, loudness_envelope_adjustment = 100, False, False, False, False, 1
    cluster_model_path, feature_retrieval, device = "", False, None
    svc_model = Svc(ck, cj, device, cluster_model_path, enhance, diffusion_model_path, diffusion_config_path, shallow_diffusion,
                    only_diffusion, use_spk_mix, feature_retrieval)
    assert patched_factories["enc_name"] == "vec768l12" and svc_model.hubert_model is patched_factories["enc"]
    infer_tool.mkdir(["raw", "results"])
    spk_mix_map = {0: [[0., 1., 1., 1.]]}
    if len(spk_mix_map) <= 1:
        use_spk_mix = False
    infer_tool.fill_a_to_b(trans, clean_names)
    results = {}
    for clean_name, tran in zip(clean_names, trans):
        raw_audio_path = f"raw/{clean_name}"
        if "." not in raw_audio_path:
            raw_audio_path += ".wav"
        infer_tool.format_wav(raw_audio_path)
        for spk in spk_list:
            kwarg = {"raw_audio_path": raw_audio_path, "spk": spk, "tran": tran, "slice_db": slice_db,
                     "cluster_infer_ratio": cluster_infer_ratio, "auto_predict_f0": auto_predict_f0, "noice_scale": noice_scale,
                     "pad_seconds": pad_seconds, "clip_seconds": clip, "lg_num": lg, "lgr_num": lgr, "f0_predictor": f0p,
                     "enhancer_adaptive_key": enhancer_adaptive_key, "cr_threshold": cr_threshold, "k_step": k_step,
                     "use_spk_mix": use_spk_mix, "second_encoding": second_encoding,
                     "loudness_envelope_adjustment": loudness_envelope_adjustment}
            audio = svc_model.slice_inference(**kwarg)
            res_path = f"results/{clean_name}_{tran}key_{spk}_sovits_{f0p}.{wav_format}"
            svc_audio.write_wav(res_path, audio, svc_model.target_sample)             # soundfile.write(...) in the reference
            svc_model.clear_empty()
            results[res_path] = audio
    # -------------------------------------------------------------------------------------------------------------
    assert patched_factories["f0_args"] == ("pm", HOP, SR, ["device", "threshold"])
    (path, audio), = results.items()
    assert os.path.exists(path) and isinstance(audio, np.ndarray) and np.isfinite(audio).all()
    assert abs(len(audio) - len(wav)) <= 2 * HOP

    # by hand: slicer -> per voiced chunk pad, 16-bit round trip, units / f0 through the same front-ends, SynthesizerTrn.infer
    from inference import slicer
    chunks = slicer.cut("raw/song.wav", db_thresh=slice_db)
    data, sr = slicer.chunks2audio("raw/song.wav", chunks)
    assert sr == SR and [t for t, _ in data] == [False, True, False], [(t, len(d)) for t, d in data]     # voiced / silent gap / voiced
    net = net.to(dev).eval()
    pos = 0
    checked_oracle = False
    for tag, d in data:
        seg = audio[pos:pos + len(d)]
        pos += len(d)
        if tag:
            assert np.abs(seg).max() == 0.0
            continue
        pad = int(SR * pad_seconds)
        x = svc_audio.pcm16_round_trip(np.concatenate([np.zeros(pad), d, np.zeros(pad)]))
        c, f0, uv = svc_model.get_unit_f0(x, 0, 0, "bob", False, "pm")
        o, _ = net.infer(c, f0, uv, g=torch.LongTensor([[1]]).to(dev), noice_scale=noice_scale)
        ref = infer_tool.pad_array(o[0, 0].cpu().numpy()[pad:-pad], len(d))
        assert np.array_equal(ref, seg)
        if not checked_oracle:          # and the engine's output for that chunk against the CPU oracle on the same features
            T = c.shape[2]
            torch.manual_seed(52468)
            noise = dict(enc_p=torch.randn(1, cfg["inter_channels"], T, device=dev), rand_ini=torch.rand(1, 9, device=dev),
                         sine=torch.randn(1, T * HOP, 9, device=dev))
            with torch.no_grad():
                oref, _ = O.synth_infer(W.make_state_dict(cfg, 9), cfg, c.cpu(), f0.cpu(), uv.cpu(), torch.LongTensor([[1]]),
                                        {k: v.cpu() for k, v in noise.items()}, noice_scale=noice_scale)
            err = (o.cpu() - oref).abs().max().item()
            assert err <= 2e-4 * max(oref.abs().max().item(), 1e-3) and (o.cpu() - oref).pow(2).mean().item() < 1e-4
            # the units really came from the engine's resampler: same thing with the oracle's resampler
            w16 = AO.resample(torch.from_numpy(x)[None], SR, 16000)[0]
            c_ref = patched_factories["enc"].encoder(w16.to(dev))
            c_ref = utils.repeat_expand_2d(c_ref[0], f0.shape[1], conf.data.unit_interpolate_mode)
            assert (c[0] - c_ref).abs().max().item() <= 1e-4 * max(1.0, c_ref.abs().max().item())
            checked_oracle = True
    assert pos == len(audio) and checked_oracle


def test_inference_main_on_the_tiny_template_two_5s_wavs(dev, tmp_path, monkeypatch, patched_factories):
    """BASELINE configs[0] as written (minus "CPU-only": the engine has no CPU path by design): configs_template/
    config_tiny_template.json at its REAL widths (filter 512, decoder 200/100/50/25/12, depthwise-separable WN, shared flow), ONE
    speaker, TWO synthetic 5 s 44.1 kHz wavs, driven through `inference_main.main`'s own loop (inference_main.py:94-152: every clean
    name x every speaker -> format_wav, slice_inference(**kwarg), write, clear_empty).  Both results are checked for length and
    against a by-hand SynthesizerTrn.infer on the same features; the first also against the CPU oracle of the tiny template."""
    import svc_audio
    from inference import infer_tool
    from inference.infer_tool import Svc
    cfg = W.tiny_config()
    patched_factories["ssl_dim"] = cfg["ssl_dim"]
    monkeypatch.chdir(tmp_path)
    os.makedirs("logs/44k")
    net, ck, cj = _write_model("logs/44k", cfg, 17)
    os.makedirs("raw")
    g = np.random.default_rng(5)
    wavs = {}
    for i, name in enumerate(("a.wav", "b.wav")):
        n = 5 * SR
        w = (0.25 * np.sin(2 * np.pi * (196.0 + 55.0 * i) * np.arange(n) / SR) + 0.04 * g.standard_normal(n)).astype(np.float32)
        svc_audio.write_wav(f"raw/{name}", w, SR)
        wavs[name] = w
    clean_names, trans, spk_list = ["a.wav", "b.wav"], [0], ["alice"]
    svc_model = Svc(ck, cj, None, "", False, "logs/44k/diffusion/model_0.pt", "logs/44k/diffusion/config.yaml", False, False, False, False)
    assert not svc_model.half_mode and not svc_model.split_mode
    infer_tool.mkdir(["raw", "results"])
    infer_tool.fill_a_to_b(trans, clean_names)
    assert trans == [0, 0]
    results = {}
    for clean_name, tran in zip(clean_names, trans):
        raw_audio_path = f"raw/{clean_name}"
        infer_tool.format_wav(raw_audio_path)
        for spk in spk_list:
            kwarg = {"raw_audio_path": raw_audio_path, "spk": spk, "tran": tran, "slice_db": -40, "cluster_infer_ratio": 0,
                     "auto_predict_f0": False, "noice_scale": 0.4, "pad_seconds": 0.5, "clip_seconds": 0, "lg_num": 0, "lgr_num": 0.75,
                     "f0_predictor": "pm", "enhancer_adaptive_key": 0, "cr_threshold": 0.05, "k_step": 100, "use_spk_mix": False,
                     "second_encoding": False, "loudness_envelope_adjustment": 1}
            audio = svc_model.slice_inference(**kwarg)
            res_path = f"results/{clean_name}_{tran}key_{spk}_sovits_pm.wav"
            svc_audio.write_wav(res_path, audio, svc_model.target_sample)
            svc_model.clear_empty()
            results[clean_name] = (res_path, audio)
    assert len(results) == 2
    net = net.to(dev).eval()
    pad = int(SR * 0.5)
    for k, name in enumerate(clean_names):
        path, audio = results[name]
        assert os.path.exists(path) and np.isfinite(audio).all() and abs(len(audio) - 5 * SR) <= 2 * HOP
        from inference import slicer
        data, sr = slicer.chunks2audio(f"raw/{name}", slicer.cut(f"raw/{name}", db_thresh=-40))
        assert sr == SR and [t for t, _ in data] == [False]                      # 5 s of tone: one voiced chunk, nothing to cut
        x = svc_audio.pcm16_round_trip(np.concatenate([np.zeros(pad), data[0][1], np.zeros(pad)]))
        c, f0, uv = svc_model.get_unit_f0(x, 0, 0, "alice", False, "pm")
        assert c.shape[1] == cfg["ssl_dim"] and abs(c.shape[2] - (5 * SR + 2 * pad) // HOP) <= 1          # T = 431 + the padding's frames
        o, _ = net.infer(c, f0, uv, g=torch.LongTensor([[0]]).to(dev), noice_scale=0.4)
        ref = infer_tool.pad_array(o[0, 0].cpu().numpy()[pad:-pad], len(audio))
        assert np.array_equal(ref, audio)
        if k == 0:
            T = c.shape[2]
            torch.manual_seed(52468)
            noise = dict(enc_p=torch.randn(1, cfg["inter_channels"], T, device=dev), rand_ini=torch.rand(1, 9, device=dev),
                         sine=torch.randn(1, T * HOP, 9, device=dev))
            with torch.no_grad():
                oref, _ = O.synth_infer(W.make_state_dict(cfg, 17), cfg, c.cpu(), f0.cpu(), uv.cpu(), torch.LongTensor([[0]]),
                                        {kk: v.cpu() for kk, v in noise.items()}, noice_scale=0.4)
            err = (o.cpu() - oref).abs().max().item()
            assert err <= 2e-4 * max(oref.abs().max().item(), 1e-3) and (o.cpu() - oref).pow(2).mean().item() < 1e-4


def test_slice_inference_batched_chunks_equal_serial(dev, tmp_path, monkeypatch, patched_factories):
    """Forced clipping (-cl) gives equal-length chunks: one B=n synthesizer call must reproduce the serial chunk loop."""
    import svc_audio
    from inference.infer_tool import Svc
    cfg = W.small_config()
    patched_factories["ssl_dim"] = cfg["ssl_dim"]
    monkeypatch.chdir(tmp_path)
    net, ck, cj = _write_model(str(tmp_path), cfg, 9)
    wav = _song(seconds_voiced=(3.1,), gap=0.0)
    svc_audio.write_wav("song.wav", wav, SR)
    svc = Svc(ck, cj, "cuda:0", "")
    calls = []
    orig = svc.net_g_ms.infer
    svc.net_g_ms.infer = lambda c, *a, **k: (calls.append(c.shape[0]), orig(c, *a, **k))[1]
    kw = dict(pad_seconds=0.3, clip_seconds=0.7, lg_num=0.1, lgr_num=0.75)
    serial = svc.slice_inference("song.wav", "alice", 0, -40, 0, False, 0.4, **kw)
    n_serial = len(calls)
    assert n_serial >= 4 and set(calls) == {1}
    calls.clear()
    batched = svc.slice_inference("song.wav", "alice", 0, -40, 0, False, 0.4, batch_chunks=True, **kw)
    assert len(calls) < n_serial and max(calls) >= 3               # the equal-length clips went through together
    assert serial.shape == batched.shape
    assert np.abs(serial - batched).max() <= 2e-5 * max(1.0, np.abs(serial).max())


def test_slice_inference_batches_unequal_silence_sliced_chunks(dev, tmp_path, monkeypatch, patched_factories):
    """the DEFAULT use — a song sliced at silences into chunks of different lengths (no forced clipping).
    batch_chunks=True groups chunks by frame-count bucket, pads to the bucket's longest item and runs the synthesizer with
    per-item lengths; every chunk must come out as the serial B=1 loop (infer_tool.py:446-495) produces it."""
    import svc_audio
    from inference.infer_tool import Svc
    cfg = W.small_config()
    patched_factories["ssl_dim"] = cfg["ssl_dim"]
    monkeypatch.chdir(tmp_path)
    net, ck, cj = _write_model(str(tmp_path), cfg, 9)
    voiced = (1.10, 1.27, 0.93, 1.31, 1.18, 0.71, 1.22)          # seven voiced stretches of different lengths, 1 s gaps
    wav = _song(seconds_voiced=voiced, gap=1.0)
    svc_audio.write_wav("song.wav", wav, SR)
    # the slicer's own segmentation of this song (it merges stretches shorter than its min_length) is exercised by the tests
    # above; here the segmentation is handed over explicitly (`chunks=`: [(is_silence, samples)], what slicer.chunks2audio
    # returns) so that the chunk list is exactly seven voiced chunks of seven different lengths between silences
    chunks, pos = [], 0
    for i, sec in enumerate(voiced):
        n = int(SR * sec)
        chunks.append((False, wav[pos:pos + n]))
        pos += n
        if i + 1 < len(voiced):
            g = int(SR * 1.0)
            chunks.append((True, wav[pos:pos + g]))
            pos += g
    svc = Svc(ck, cj, "cuda:0", "")
    calls = []
    orig = svc.net_g_ms.infer
    svc.net_g_ms.infer = lambda c, *a, **k: (calls.append((c.shape[0], c.shape[2], k.get("lengths") is not None)), orig(c, *a, **k))[1]
    serial = svc.slice_inference("song.wav", "alice", 0, -40, 0, False, 0.4, pad_seconds=0.5, chunks=chunks)
    n_serial = len(calls)
    lens = sorted({t for _, t, _ in calls})
    assert n_serial >= 6 and len(lens) >= 4 and all(b == 1 for b, _, _ in calls), calls
    calls.clear()
    batched = svc.slice_inference("song.wav", "alice", 0, -40, 0, False, 0.4, pad_seconds=0.5, batch_chunks=True, chunks=chunks)
    assert len(calls) < n_serial and any(b >= 2 and ragged for b, _, ragged in calls), calls     # unequal chunks went through together
    assert serial.shape == batched.shape
    assert np.abs(serial - batched).max() <= 2e-5 * max(1.0, np.abs(serial).max())


@pytest.mark.parametrize("src,dst,n", [(44100, 16000, 44100 * 3 + 17), (48000, 44100, 30011), (16000, 44100, 9001),
                                       (44100, 16000, 400)])
def test_resample_matches_oracle(dev, src, dst, n):
    import svc_audio
    g = torch.Generator().manual_seed(n)
    x = torch.randn(2, n, generator=g) * 0.3
    ref = AO.resample(x, src, dst)
    y = svc_audio.Resampler(src, dst)(x.to(dev))
    assert y.shape == ref.shape
    assert (y.cpu() - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())
    # a pure tone well below both Nyquist limits keeps its frequency and amplitude
    t = torch.arange(n) / src
    tone = torch.sin(2 * np.pi * 440.0 * t)[None]
    yt = svc_audio.Resampler(src, dst)(tone.to(dev))[0].cpu()
    td = torch.arange(yt.shape[0]) / dst
    lo = 100
    if yt.shape[0] > 4 * lo:
        assert (yt[lo:-lo] - torch.sin(2 * np.pi * 440.0 * td)[lo:-lo]).abs().max().item() < 5e-3


@pytest.mark.parametrize("cls_name,layer,proj", [("ContentVec768L12", 12, False), ("ContentVec256L9", 9, True)])
def test_contentvec_encoders_match_oracle(dev, tmp_path, cls_name, layer, proj):
    """vencoder/ContentVec768L12.py:23-37 / ContentVec256L9.py:23-38 with the checkpoint in fairseq's key layout."""
    import importlib
    sd = HO.make_state_dict(11)
    path = str(tmp_path / "checkpoint_best_legacy_500.pt")
    torch.save({"model": HO.to_fairseq_state_dict(sd), "cfg": None}, path)
    cls = getattr(importlib.import_module("vencoder." + cls_name), cls_name)
    enc = cls(vec_path=path, device=dev)
    assert enc.hidden_dim == (256 if proj else 768)
    g = torch.Generator().manual_seed(layer)
    for n in (16000, 7003):
        wav = 0.3 * torch.randn(n, generator=g)
        with torch.no_grad():
            ref = HO.encode(sd, wav[None, None], layer=layer, pad=0)
            if proj:
                ref = torch.nn.functional.linear(ref, sd["proj.weight"], sd["proj.bias"])
        c = enc.encoder(wav.to(dev))
        assert c.shape == (1, ref.shape[2], ref.shape[1])
        err = (c[0].t().cpu() - ref[0]).abs().max().item()
        assert err <= 2e-4 * max(1.0, ref.abs().max().item()), err
    # through the factory Svc uses
    import utils
    import vencoder.ContentVec768L12 as M
    assert utils._SPEECH_ENCODERS["vec768l12"] == "ContentVec768L12" and M.ContentVec768L12.OUTPUT_LAYER == 12


def test_slice_inference_batches_the_unit_encoder_over_equal_length_chunks(dev, tmp_path, monkeypatch, patched_factories):
    """with batch_chunks=True the engine's HuBERT-based unit encoders take the chunks' 16 kHz waves through
    `encoder_batch` — equal lengths (forced clipping) as ONE batch: every op of that stack is per item (GroupNorm(512, 512) over
    the item's own time axis), so the batch equals the serial loop to fp32 round-off.  Real ContentVec768L12 mirror on a synthetic
    fairseq-format checkpoint; the synthesizer is built for its 768-d units."""
    import svc_audio
    import utils
    from inference.infer_tool import Svc
    from vencoder.ContentVec768L12 import ContentVec768L12
    cfg = dict(W.small_config(), ssl_dim=768)
    patched_factories["ssl_dim"] = 768
    monkeypatch.chdir(tmp_path)
    ckpt = str(tmp_path / "checkpoint_best_legacy_500.pt")
    torch.save({"model": HO.to_fairseq_state_dict(HO.make_state_dict(4)), "cfg": None, "args": None}, ckpt)
    enc = ContentVec768L12(vec_path=ckpt, device=dev)
    monkeypatch.setattr(utils, "get_speech_encoder", lambda name, device=None, **kw: enc)
    net, ck, cj = _write_model(str(tmp_path), cfg, 9)
    svc_audio.write_wav("song.wav", _song(seconds_voiced=(3.1,), gap=0.0), SR)
    svc = Svc(ck, cj, "cuda:0", "")
    sizes = []
    orig = enc.model.encode
    enc.model.encode = lambda x, layer=None, lengths=None: (sizes.append(x.shape[0]), orig(x, layer=layer, lengths=lengths))[1]
    kw = dict(pad_seconds=0.3, clip_seconds=0.7, lg_num=0.1, lgr_num=0.75)
    serial = svc.slice_inference("song.wav", "alice", 0, -40, 0, False, 0.4, **kw)
    assert len(sizes) >= 4 and set(sizes) == {1}
    n_serial = len(sizes)
    sizes.clear()
    batched = svc.slice_inference("song.wav", "alice", 0, -40, 0, False, 0.4, batch_chunks=True, **kw)
    assert len(sizes) < n_serial and max(sizes) >= 3, sizes          # the equal-length clips went through the encoder together
    assert serial.shape == batched.shape
    assert np.abs(serial - batched).max() <= 5e-5 * max(1.0, np.abs(serial).max())
