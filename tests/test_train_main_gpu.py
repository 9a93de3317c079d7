"""The training ENTRY POINT behind the reference's CLI (`svc_run.py train.py -c ... -m ...` used to define a few
functions and exit 0).  A synthetic dataset in the reference's on-disk formats (wav + .soft.pt + .f0.npy [+ .spec.pt]), a
small config, then the launcher in a subprocess exactly as a user types it: it must train (losses in logs/<model>/train.log),
evaluate, write G_<step>.pth / D_<step>.pth that utils.load_checkpoint reads back, and resume from them."""
import glob
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "so-vits-svc_amd")
sys.path.insert(0, HERE)


def _config(root, fl):
    import synthetic_data as W
    model = {k: v for k, v in W.small_config().items() if k not in ("spec_channels", "segment_size")}
    model.update(ssl_dim=24, n_speakers=2, p_dropout=0.1)
    cfg = dict(
        train=dict(log_interval=1, eval_interval=2, seed=1234, epochs=2, learning_rate=1e-4, betas=[0.8, 0.99], eps=1e-9,
                   batch_size=2, fp16_run=False, half_type="fp16", lr_decay=0.999875, segment_size=8192, init_lr_ratio=1,
                   warmup_epochs=1, c_mel=45, c_kl=1.0, use_sr=True, max_speclen=512, port="8017", keep_ckpts=2,
                   all_in_mem=False, vol_aug=False),
        data=dict(training_files=fl, validation_files=fl, max_wav_value=32768.0, sampling_rate=44100, filter_length=2048,
                  hop_length=512, win_length=2048, n_mel_channels=80, mel_fmin=0.0, mel_fmax=22050, unit_interpolate_mode="nearest"),
        model=model, spk=dict(alice=0, bob=1))
    p = os.path.join(root, "config.json")
    with open(p, "w") as f:
        json.dump(cfg, f)
    return p


def test_training_entry_point_trains_checkpoints_and_resumes(dev, tmp_path):
    from test_data_utils import _make_dataset
    root = str(tmp_path)
    fl, _ = _make_dataset(root, n_items=5, with_spec=False)      # no cached spectrograms: the loader hands SpecContext items over
    cj = _config(root, fl)
    env = dict(os.environ, SVC_LOADER_WORKERS="0")
    cmd = [sys.executable, os.path.join(PKG, "svc_run.py"), os.path.join(PKG, "train.py"), "-c", cj, "-m", "unit"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=root, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    mdir = os.path.join(root, "logs", "unit")
    log = open(os.path.join(mdir, "train.log")).read()
    assert log.count("Losses:") >= 4 and "====> Epoch: 2" in log, log[-1500:]
    gs = sorted(glob.glob(os.path.join(mdir, "G_*.pth")))
    ds = sorted(glob.glob(os.path.join(mdir, "D_*.pth")))
    assert gs and ds and os.path.exists(os.path.join(mdir, "config.json")), os.listdir(mdir)
    ck = torch.load(gs[-1], map_location="cpu")
    assert set(ck) >= {"model", "iteration", "optimizer", "learning_rate"} and any(k.endswith("weight_g") for k in ck["model"])
    assert all(torch.isfinite(v).all() for v in ck["model"].values() if v.is_floating_point())
    # resume: one more epoch from the newest checkpoint pair (global_step continues from the file name, train.py:99-100)
    with open(cj) as f:
        cfg = json.load(f)
    cfg["train"]["epochs"] = 3
    with open(cj, "w") as f:
        json.dump(cfg, f)
    r2 = subprocess.run(cmd, capture_output=True, text=True, cwd=root, timeout=900, env=env)
    assert r2.returncode == 0, (r2.stdout[-1500:], r2.stderr[-3000:])
    log2 = open(os.path.join(mdir, "train.log")).read()
    assert "Loaded checkpoint" in log2 + r2.stdout + r2.stderr or "====> Epoch: 3" in log2
    assert "====> Epoch: 3" in log2, log2[-1500:]
