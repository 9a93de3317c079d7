"""`inference.infer_tool.Svc`'s precision modes END TO END through the boundary (VERDICT r5 missing #4, weak #1 / #3):

* the reference's half trigger (inference/infer_tool.py:196-198: a checkpoint whose NAME holds "half" is loaded and `.half()`ed;
  compress_model.py:21-48 writes such a file: enc_q dropped, every tensor cast to fp16; utils.py:163 casts the model to the
  checkpoint's dtype): an fp16 checkpoint written the way compress_model.py writes it, named `..._half.pth`, driven through
  `Svc(...)` -> `infer` / `slice_inference`, against the same checkpoint through the fp32 path and against a by-hand
  `SynthesizerTrn.half().infer`;
* the split mode's range guard: `SVC_INFER_SPLIT=1` on a checkpoint whose generator activations leave the fp16 range — `Svc` must
  notice (the flag is raised inside the replayed graph), re-run on the fp32 kernels bit-identically to a plain fp32 `Svc`, and stay
  there; a normal checkpoint must never trip it."""
import os

import numpy as np
import pytest
import torch

from oracle import weights as W
from test_svc_gpu import HOP, SR, _FrontEnd, _write_model

pytestmark = pytest.mark.gpu


def _compress_like_the_reference(src_ckpt, dst_ckpt):
    """What compress_model.py:21-48 (`removeOptimizer(..., ishalf=True)`) leaves on disk: {'model': {k: v.half() for k not enc_q.*},
    'iteration': 0, 'optimizer': <fresh AdamW state>, 'learning_rate': 0.0001} — written here with torch.save from the engine-saved
    fp32 checkpoint (the reference script itself needs its own models.py on the path; the FORMAT is what the loader sees)."""
    sd = torch.load(src_ckpt, map_location="cpu")
    model = {k: (v.half() if torch.is_floating_point(v) else v) for k, v in sd["model"].items() if "enc_q" not in k}
    torch.save({"model": model, "iteration": 0, "optimizer": {"state": {}, "param_groups": []}, "learning_rate": 0.0001}, dst_ckpt)
    return model


def _wav(seconds, seed=1):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(int(SR * seconds), generator=g) - 0.5).numpy().astype(np.float32)


def test_compressed_fp16_checkpoint_with_the_trigger_word_in_its_name(dev, tmp_path, monkeypatch):
    assert "half" not in str(tmp_path)      # the reference's trigger is the word anywhere in the PATH (infer_tool.py:196)
    from inference.infer_tool import Svc
    monkeypatch.delenv("SVC_INFER_HALF", raising=False)
    monkeypatch.delenv("SVC_INFER_SPLIT", raising=False)
    cfg = W.full_config()
    net, ck, cj = _write_model(str(tmp_path), cfg, 41)
    ck_half = os.path.join(str(tmp_path), "G_100_half.pth")
    model16 = _compress_like_the_reference(ck, ck_half)
    assert all(v.dtype == torch.float16 for v in model16.values() if torch.is_floating_point(v)) and not any("enc_q" in k for k in model16)
    fe = _FrontEnd(cfg["ssl_dim"], dev)
    svc = Svc(ck_half, cj, device="cuda:0", cluster_model_path="", front_end=fe)
    assert svc.half_mode and svc.net_g_ms.dec.half_mode is True            # the NAME triggered it (infer_tool.py:196-198)
    p0 = next(svc.net_g_ms.parameters())
    assert p0.dtype == torch.float32                                       # fp32 masters holding the fp16 values exactly
    k0 = next(k for k in model16 if k.endswith("emb_g.weight"))
    assert torch.equal(svc.net_g_ms.state_dict()[k0].cpu(), model16[k0].float())
    wav = _wav(2.0)
    audio, n, n_frames = svc.infer("bob", 2, (wav, SR), noice_scale=0.4)
    assert n == audio.shape[-1] == n_frames * HOP and torch.isfinite(audio).all()
    # the same checkpoint WITHOUT the trigger word in its name: the fp32 path on the same (fp16-valued) weights
    ck_plain = os.path.join(str(tmp_path), "G_100_fp16vals.pth")
    os.replace(ck_half, ck_plain)
    svc32 = Svc(ck_plain, cj, device="cuda:0", cluster_model_path="", front_end=fe)
    assert not svc32.half_mode and not svc32.net_g_ms.dec.half_mode
    audio32, _, _ = svc32.infer("bob", 2, (wav, SR), noice_scale=0.4)
    mse = (audio - audio32).pow(2).mean().item()
    print(f"Svc on a compress_model-style fp16 checkpoint: half pipeline vs fp32 kernels on the same weights, MSE {mse:.3e}, "
          f"max {(audio - audio32).abs().max().item():.3e}")
    assert 0.0 < mse < 1e-4                                                # north_star's bar; and it really is another pipeline
    # by hand: the same model object class, .half(), same units / f0, default seed
    c, f0, uv = svc.get_unit_f0(wav, 2, 0, "bob", False, "pm")
    ref, _ = svc32.net_g_ms.half().infer(c, f0, uv, g=torch.LongTensor([[1]]).to(dev), noice_scale=0.4)
    assert torch.equal(ref[0, 0], audio)
    # slice_inference through the half pipeline: chunked + cross-faded, finite, right length
    fe._wavs["x.wav"] = (wav, SR)
    out = svc.slice_inference("x.wav", "alice", 0, -40, 0, False, 0.4, pad_seconds=0.2, clip_seconds=0.6, lg_num=0.1, chunks=[(False, wav)])
    assert abs(len(out) - len(wav)) <= 4 * HOP and np.isfinite(out).all()
    # SVC_INFER_HALF=0 keeps a half-named checkpoint in fp32 (the documented opt-out)
    os.replace(ck_plain, ck_half)
    monkeypatch.setenv("SVC_INFER_HALF", "0")
    svc_off = Svc(ck_half, cj, device="cuda:0", cluster_model_path="", front_end=fe)
    assert not svc_off.half_mode
    a_off, _, _ = svc_off.infer("bob", 2, (wav, SR), noice_scale=0.4)
    assert torch.equal(a_off, audio32)


def test_split_mode_range_guard_falls_back_to_fp32(dev, tmp_path, monkeypatch, caplog):
    from inference.infer_tool import Svc
    import utils
    cfg = W.full_config()
    net, ck, cj = _write_model(str(tmp_path), cfg, 43)
    fe = _FrontEnd(cfg["ssl_dim"], dev)
    wav = _wav(1.5, seed=3)
    monkeypatch.setenv("SVC_INFER_SPLIT", "1")
    # (1) a normal checkpoint: split mode runs, the flag is never raised, the result is fp32-level
    svc = Svc(ck, cj, device="cuda:0", cluster_model_path="", front_end=fe)
    assert svc.split_mode and svc.net_g_ms.dec.half_mode == "split"
    a_split, _, _ = svc.infer("bob", 0, (wav, SR), noice_scale=0.4)
    assert svc.split_mode and getattr(svc, "range_fallbacks", 0) == 0
    monkeypatch.setenv("SVC_INFER_SPLIT", "0")
    svc32 = Svc(ck, cj, device="cuda:0", cluster_model_path="", front_end=fe)
    a32, _, _ = svc32.infer("bob", 0, (wav, SR), noice_scale=0.4)
    assert (a_split - a32).abs().max().item() < 5e-6
    # (2) trained-checkpoint-like extreme: the first MRF stage's input driven out of range — ups.0's weight-norm gain x 3e5
    big = {k: v.clone() for k, v in net.state_dict().items()}
    big["dec.ups.0.weight_g"] = big["dec.ups.0.weight_g"] * 3e5
    net.load_state_dict(big)
    ck_big = os.path.join(str(tmp_path), "G_200.pth")
    utils.save_checkpoint(net, None, 1e-4, 200, ck_big)
    svc32b = Svc(ck_big, cj, device="cuda:0", cluster_model_path="", front_end=fe)
    a32b, _, _ = svc32b.infer("bob", 0, (wav, SR), noice_scale=0.4)
    assert torch.isfinite(a32b).all()
    monkeypatch.setenv("SVC_INFER_SPLIT", "1")
    svcb = Svc(ck_big, cj, device="cuda:0", cluster_model_path="", front_end=fe)
    assert svcb.split_mode
    with caplog.at_level("WARNING", logger="infer_tool"):
        ab, _, _ = svcb.infer("bob", 0, (wav, SR), noice_scale=0.4)
    assert svcb.range_fallbacks == 1 and not svcb.split_mode and not svcb.net_g_ms.dec.half_mode
    assert any("fp16 range" in r.message for r in caplog.records)
    assert torch.equal(ab, a32b)                                            # the re-run IS the fp32 path (same seed, same draws)
    ab2, _, _ = svcb.infer("bob", 0, (wav, SR), noice_scale=0.4)            # and it stays there
    assert torch.equal(ab2, a32b) and svcb.range_fallbacks == 1
