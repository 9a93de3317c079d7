"""The `inference.infer_tool.Svc` boundary (reference inference/infer_tool.py:116-127,256-340,356-496) on
the GPU: config json + checkpoint written in the reference's formats, injected front-ends standing in for the
out-of-scope unit encoder / f0 predictor / wav loader, `infer` against a direct SynthesizerTrn.infer with the same
seed, and `slice_inference`'s chunking / cross-fade bookkeeping."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import weights as W

pytestmark = pytest.mark.gpu
SR, HOP = 44100, 512


class _FrontEnd:
    """Deterministic stand-ins with the reference's own interfaces (vencoder.encoder / f0 predictor / load_audio)."""

    def __init__(self, ssl_dim, dev):
        outer = self

        class Enc:
            def encoder(self, wav16k):                      # [n] -> [1, ssl_dim, n/320]  (50 fps units)
                n = max(1, wav16k.shape[-1] // 320)
                g = torch.Generator().manual_seed(int(wav16k.shape[-1]))
                return torch.randn(1, ssl_dim, n, generator=g).to(dev)

        class F0:
            def compute_f0_uv(self, wav):
                n = len(wav) // HOP
                t = np.arange(n)
                f0 = 220.0 + 60.0 * np.sin(t / 9.0)
                f0[(t // 13) % 5 == 0] = 0.0
                return f0, (f0 > 0).astype(np.float32)

        self.hubert_model = Enc()
        self.f0_predictor_object = F0()
        self.load_wav = lambda path: path if isinstance(path, tuple) else outer._wavs[path]
        self.resample = lambda x, sr_in, sr_out: torch.nn.functional.interpolate(
            x[None].float(), size=int(x.shape[-1] * sr_out / sr_in), mode="linear", align_corners=False)[0]
        self._wavs = {}


def _write_model(tmp_path, cfg, seed):
    import models
    import utils
    kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
    net = models.SynthesizerTrn(cfg["spec_channels"], cfg["segment_size"], **kw)
    net.load_state_dict(W.make_state_dict(cfg, seed))
    ck = os.path.join(tmp_path, "G_100.pth")
    utils.save_checkpoint(net, None, 1e-4, 100, ck)
    conf = dict(train=dict(segment_size=cfg["segment_size"] * HOP),
                data=dict(sampling_rate=SR, filter_length=(cfg["spec_channels"] - 1) * 2, hop_length=HOP,
                          unit_interpolate_mode="nearest"),
                model=kw, spk={"alice": 0, "bob": 1})
    cj = os.path.join(tmp_path, "config.json")
    with open(cj, "w") as f:
        json.dump(conf, f)
    return net, ck, cj


def test_svc_infer_and_slice_inference(dev, tmp_path):
    from inference.infer_tool import Svc
    cfg = W.small_config()
    net, ck, cj = _write_model(str(tmp_path), cfg, 9)
    fe = _FrontEnd(cfg["ssl_dim"], dev)
    svc = Svc(ck, cj, device="cuda:0", cluster_model_path="", front_end=fe)
    assert svc.target_sample == SR and svc.hop_size == HOP and svc.spk2id["bob"] == 1
    assert len(svc.net_g_ms.state_dict()) == len(net.state_dict())

    g = torch.Generator().manual_seed(1)
    wav = (torch.rand(SR * 2, generator=g) - 0.5).numpy().astype(np.float32)          # 2 s
    audio, n, n_frames = svc.infer("bob", 2, (wav, SR), noice_scale=0.4)
    assert n == audio.shape[-1] == n_frames * HOP and n_frames == len(wav) // HOP
    assert torch.isfinite(audio).all()
    # same thing by hand: units/f0 through the front-ends, then SynthesizerTrn.infer with the default seed
    c, f0, uv = svc.get_unit_f0(wav, 2, 0, "bob", False, "pm")
    ref, _ = net.to(dev).eval().infer(c, f0, uv, g=torch.LongTensor([[1]]).to(dev), noice_scale=0.4)
    assert torch.equal(ref[0, 0], audio)
    with pytest.raises(RuntimeError):
        svc.infer("carol", 0, (wav, SR))

    # slice_inference: two voiced chunks around a silent one, clip_seconds splitting with cross-fades
    fe._wavs["x.wav"] = (wav, SR)
    chunks = [(False, wav[:SR]), (True, np.zeros(SR // 2, dtype=np.float32)), (False, wav[SR:])]
    out = svc.slice_inference("x.wav", "alice", 0, -40, 0, False, 0.4, pad_seconds=0.2, chunks=chunks)
    assert isinstance(out, np.ndarray) and abs(len(out) - (len(wav) + SR // 2)) <= 3 * HOP
    assert np.abs(out[SR + 100:SR + SR // 2 - 100]).max() == 0.0                        # the silent chunk stays silent
    out2 = svc.slice_inference("x.wav", "alice", 0, -40, 0, False, 0.4, pad_seconds=0.2, clip_seconds=0.6, lg_num=0.1,
                               chunks=[(False, wav)])
    assert abs(len(out2) - len(wav)) <= 4 * HOP and np.isfinite(out2).all()
    svc.clear_empty()
    svc.unload_model()


def _write_diffusion(tmp_path, ssl_dim, n_mels, method="dpm-solver++", speedup=10):
    """Vocoder checkpoint + config.json and diffusion checkpoint + yaml in the reference's layouts
    (vdecoder/nsf_hifigan/models.py:17-35, diffusion/unit2mel.py:22-58)."""
    import yaml
    from oracle import diffusion_oracle as DO
    from oracle import nsf_hifigan_oracle as NO
    vd = os.path.join(tmp_path, "voc")
    os.makedirs(vd, exist_ok=True)
    h = dict(NO.small_h(), num_mels=n_mels, upsample_rates=[8, 8, 8], upsample_kernel_sizes=[16, 16, 16], n_fft=2048,
             win_size=2048, hop_size=HOP, fmin=40, fmax=16000, sampling_rate=SR)
    with open(os.path.join(vd, "config.json"), "w") as f:
        json.dump(h, f)
    torch.save({"generator": NO.make_state_dict(h, 21)}, os.path.join(vd, "model"))
    c = dict(DO.small_cfg(), input_channel=ssl_dim, out_dims=n_mels, n_spk=2)
    dd = os.path.join(tmp_path, "diff")
    os.makedirs(dd, exist_ok=True)
    from diffusion.unit2mel import Unit2Mel
    u2m = Unit2Mel(c["input_channel"], c["n_spk"], False, c["out_dims"], c["n_layers"], c["n_chans"], c["n_hidden"],
                   c["timesteps"], c["k_step_max"])
    u2m.load_state_dict(DO.make_state_dict(c, 22), strict=False)      # + the schedule buffers, as a real checkpoint has them
    torch.save({"model": u2m.state_dict()}, os.path.join(dd, "model_0.pt"))
    args = dict(data=dict(encoder_out_channels=ssl_dim, sampling_rate=SR, block_size=HOP, encoder="stub", unit_interpolate_mode="nearest"),
                model=dict(n_spk=c["n_spk"], use_pitch_aug=False, n_layers=c["n_layers"], n_chans=c["n_chans"],
                           n_hidden=c["n_hidden"], timesteps=c["timesteps"], k_step_max=c["k_step_max"]),
                vocoder=dict(type="nsf-hifigan", ckpt=os.path.join(vd, "model")), infer=dict(speedup=speedup, method=method),
                spk={"alice": 0, "bob": 1})
    yp = os.path.join(dd, "config.yaml")
    with open(yp, "w") as f:
        yaml.safe_dump(args, f)
    return os.path.join(dd, "model_0.pt"), yp


def test_svc_shallow_and_only_diffusion(dev, tmp_path):
    """infer_tool.py:163-181,278-304 — synthesizer -> Vocoder.extract -> Unit2Mel (DPM-Solver++) -> Vocoder.infer wired
    through Svc; compared with the same pipeline composed by hand from the mirror modules (each has its own parity test)."""
    from inference.infer_tool import Svc
    cfg = W.small_config()
    net, ck, cj = _write_model(str(tmp_path), cfg, 9)
    dm, dy = _write_diffusion(str(tmp_path), cfg["ssl_dim"], 16)
    fe = _FrontEnd(cfg["ssl_dim"], dev)
    svc = Svc(ck, cj, device="cuda:0", cluster_model_path="", front_end=fe, shallow_diffusion=True,
              diffusion_model_path=dm, diffusion_config_path=dy)
    assert svc.shallow_diffusion and svc.vocoder.dimension == 16 and svc.diffusion_args.infer.method == "dpm-solver++"
    g = torch.Generator().manual_seed(1)
    wav = (0.5 * (torch.rand(SR, generator=g) - 0.5)).numpy().astype(np.float32)          # 1 s
    torch.manual_seed(11)
    audio, n, n_frames = svc.infer("bob", 0, (wav, SR), k_step=30)
    assert n == audio.shape[-1] == n_frames * HOP and torch.isfinite(audio).all()
    # by hand
    c, f0, uv = svc.get_unit_f0(wav, 0, 0, "bob", False, "pm")
    sid = torch.LongTensor([[1]]).to(dev)
    torch.manual_seed(11)
    a0, f0o = svc.net_g_ms.infer(c, f0, uv, g=sid, noice_scale=0.4)
    a0 = a0[0, 0].float()
    mel0 = svc.vocoder.extract(a0[None, :], SR)
    vol = svc.volume_extractor.extract(a0[None, :])[None, :, None].to(dev)
    mel1 = svc.diffusion_model(c.transpose(-1, -2), f0o[:, :, None], vol, spk_id=sid, gt_spec=mel0, infer=True, infer_speedup=10,
                               method="dpm-solver++", k_step=30, use_tqdm=False)
    ref = svc.vocoder.infer(mel1, f0o[:, :, None]).squeeze()
    assert torch.equal(ref, audio)
    assert (mel1 - mel0).abs().max() > 1e-3          # the diffusion stage did change the mel
    with pytest.raises(Exception):
        svc.infer("bob", 0, (wav, SR), k_step=1000)  # k_step > k_step_max (diffusion/unit2mel.py:131-132)

    only = Svc(None, None, device="cuda:0", cluster_model_path="", front_end=fe, only_diffusion=True,
               diffusion_model_path=dm, diffusion_config_path=dy)
    assert only.net_g_ms is None and only.target_sample == SR and only.hop_size == HOP and only.spk2id["bob"] == 1
    torch.manual_seed(12)
    a2, n2, nf2 = only.infer("alice", 0, (wav, SR))
    assert n2 == nf2 * HOP and torch.isfinite(a2).all()
