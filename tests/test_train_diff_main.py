"""`svc_run.py train_diff.py -c diffusion.yaml` — the reference's shallow-diffusion training ENTRY POINT on the engine
(the launcher ran the reference's script, whose `from diffusion.solver import train` failed on the mirror).

CPU: the name surface the reference's `train_diff.py` imports resolves to the engine (ast walk, like tests/test_boundary_cpu.py);
`diffusion.data_loaders.AudioDataset` yields the reference's own items on the same files / `random` seed (reference class
imported with `librosa.get_duration` stubbed — its only librosa call); `Saver` / `load_model` round trip; the rank sampler.
GPU: the launcher, in a subprocess exactly as a user types it, trains, logs, validates (sampler + vocoder + RTF print), writes
`model_<step>.pt`, and a second invocation resumes from the newest one with the step counter and the lr schedule continued."""
import ast
import glob
import os
import random
import subprocess
import sys
import types

import numpy as np
import pytest
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "so-vits-svc_amd")
REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="no reference checkout on this machine")
SR, HOP, SSL, MELS = 44100, 512, 24, 16


def _make_dataset(root, n_items=5, seed=3):
    """wav + .f0.npy + .vol.npy + .aug_vol.npy + .mel.npy + .aug_mel.npy + .soft.pt per item (preprocess_hubert_f0.py:31-103)."""
    from scipy.io.wavfile import write
    g = torch.Generator().manual_seed(seed)
    lines = []
    for i in range(n_items):
        spk = "alice" if i % 2 == 0 else "bob"
        d = os.path.join(root, "dataset", spk)
        os.makedirs(d, exist_ok=True)
        T = 70 + 9 * i                                              # 0.81 .. 1.23 s
        wav = ((torch.rand(T * HOP, generator=g) - 0.5) * 20000).to(torch.int16).numpy()
        p = os.path.join(d, f"u{i}.wav")
        write(p, SR, wav)
        torch.save(torch.randn(1, SSL, T // 2 + 1, generator=g), p + ".soft.pt")
        f0 = (100 + 200 * torch.rand(T, generator=g)).numpy()
        f0[:3] = 0
        np.save(p + ".f0.npy", np.asanyarray((f0, (f0 > 0).astype(float)), dtype=object), allow_pickle=True)
        np.save(p + ".vol.npy", torch.rand(T, generator=g).numpy())
        np.save(p + ".aug_vol.npy", torch.rand(T, generator=g).numpy())
        np.save(p + ".mel.npy", (torch.randn(T, MELS, generator=g) * 2 - 5).numpy())
        np.save(p + ".aug_mel.npy", np.asanyarray(((torch.randn(T, MELS, generator=g) * 2 - 5).numpy(), float(i % 3 - 1)), dtype=object),
                allow_pickle=True)
        lines.append(p)
    fl = os.path.join(root, "train.txt")
    with open(fl, "w") as f:
        f.write("\n".join(lines) + "\n")
    vl = os.path.join(root, "val.txt")
    with open(vl, "w") as f:
        f.write("\n".join(lines[:2]) + "\n")
    return fl, vl


def _config(root, fl, vl, vocoder_ckpt, **train):
    from oracle import diffusion_oracle as DO
    c = DO.small_cfg()
    cfg = dict(
        data=dict(sampling_rate=SR, block_size=HOP, duration=0.5, encoder="stub", encoder_out_channels=SSL, training_files=fl,
                  validation_files=vl, extensions=["wav"], unit_interpolate_mode="nearest"),
        model=dict(type="Diffusion", n_layers=c["n_layers"], n_chans=c["n_chans"], n_hidden=c["n_hidden"], use_pitch_aug=True,
                   timesteps=c["timesteps"], k_step_max=0, n_spk=2),
        device="cuda", vocoder=dict(type="nsf-hifigan", ckpt=vocoder_ckpt), infer=dict(speedup=10, method="dpm-solver++"),
        env=dict(expdir=os.path.join(root, "logs", "diffusion"), gpu_id=0),
        train=dict(dict(num_workers=0, amp_dtype="fp32", batch_size=2, cache_all_data=True, cache_device="cpu", cache_fp16=False,
                        epochs=2, interval_log=1, interval_val=2, interval_force_save=4, lr=2e-4, decay_step=4, gamma=0.5,
                        weight_decay=0, save_opt=True), **train),
        spk=dict(alice=0, bob=1))
    p = os.path.join(root, "diffusion.yaml")
    with open(p, "w") as f:
        yaml.safe_dump(cfg, f)
    return p


def _vocoder(root):
    import json
    from oracle import nsf_hifigan_oracle as NO
    vd = os.path.join(root, "voc")
    os.makedirs(vd, exist_ok=True)
    h = dict(NO.small_h(), num_mels=MELS, upsample_rates=[8, 8, 8], upsample_kernel_sizes=[16, 16, 16], n_fft=2048, win_size=2048,
             hop_size=HOP, fmin=40, fmax=16000, sampling_rate=SR)
    with open(os.path.join(vd, "config.json"), "w") as f:
        json.dump(h, f)
    torch.save({"generator": NO.make_state_dict(h, 21)}, os.path.join(vd, "model"))
    return os.path.join(vd, "model")


# ---------------------------------------------------------------- CPU ----------------------------------------------------
@needs_ref
def test_every_name_train_diff_imports_resolves_to_the_engine():
    tree = ast.parse(open(os.path.join(REF, "train_diff.py")).read())
    want = []
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module and node.module.startswith("diffusion"):
            want += [(node.module, a.name) for a in node.names]
    assert ("diffusion.solver", "train") in want and ("diffusion.data_loaders", "get_data_loaders") in want
    import importlib
    for mod, name in want:
        m = importlib.import_module(mod)
        assert m.__file__.startswith(PKG), (mod, m.__file__)
        if not hasattr(m, name):
            importlib.import_module(mod + "." + name)           # `from diffusion.logger import utils`
    # what train_diff.py / solver.train call on those modules
    from diffusion.logger import utils as LU
    from diffusion.logger.saver import Saver
    import diffusion.solver as solver
    for n in ("load_config", "load_model", "get_network_paras_amount", "DotDict", "traverse_dir"):
        assert hasattr(LU, n), n
    for n in ("log_info", "log_value", "log_spec", "log_audio", "save_model", "delete_model", "global_step_increment",
              "get_interval_time", "get_total_time"):
        assert hasattr(Saver, n), n
    assert callable(solver.train) and callable(solver.test)
    assert os.path.exists(os.path.join(PKG, "train_diff.py"))        # what svc_run.py substitutes for the checkout's script


@needs_ref
def test_audio_dataset_equals_the_reference_class(tmp_path):
    import importlib.util
    fl, vl = _make_dataset(str(tmp_path))
    from diffusion import data_loaders as mine
    stub = types.ModuleType("librosa")
    stub.get_duration = lambda filename, sr=None: mine.audio_duration(filename)
    had = sys.modules.get("librosa")
    sys.modules["librosa"] = stub
    try:
        spec = importlib.util.spec_from_file_location("ref_diff_data_loaders", os.path.join(REF, "diffusion", "data_loaders.py"))
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
    finally:
        if had is None:
            del sys.modules["librosa"]
        else:
            sys.modules["librosa"] = had
    for all_data in (True, False):
        kw = dict(filelists=fl, waveform_sec=0.5, hop_size=HOP, sample_rate=SR, spk=dict(alice=0, bob=1), load_all_data=all_data,
                  n_spk=2, use_aug=True, unit_interpolate_mode="nearest")
        a, b = mine.AudioDataset(**kw), ref.AudioDataset(**kw)
        assert len(a) == len(b) == 5
        for i in range(5):
            random.seed(100 + i)
            x = a[i]
            random.seed(100 + i)
            y = b[i]
            assert set(x) == set(y)
            for k in x:
                if torch.is_tensor(x[k]):
                    assert x[k].dtype == y[k].dtype and torch.equal(x[k], y[k]), (all_data, i, k)
                else:
                    assert x[k] == y[k]
        assert x["mel"].shape == (43, MELS) and x["units"].shape == (43, SSL) and x["aug_shift"].shape == (1, 1)
    whole = mine.AudioDataset(filelists=vl, waveform_sec=0.5, hop_size=HOP, sample_rate=SR, spk={}, whole_audio=True, n_spk=1)
    assert whole[1]["mel"].shape[0] == 79 and int(whole[1]["spk_id"]) == 0


def test_saver_and_load_model_round_trip(tmp_path):
    from diffusion.logger import utils as LU
    from diffusion.logger.saver import Saver
    args = LU.DotDict(env=dict(expdir=str(tmp_path / "exp")), data=dict(sampling_rate=SR), train=dict(lr=1e-3))
    net = torch.nn.Linear(3, 2)
    opt = torch.optim.AdamW(net.parameters())
    assert LU.load_model(args.env.expdir, net, opt)[0] == 0           # fresh directory
    saver = Saver(args, initial_global_step=0)
    assert yaml.safe_load(open(tmp_path / "exp" / "config.yaml"))["train"]["lr"] == 1e-3
    saver.log_info({"model": 12345})
    saver.log_info("hello")
    assert open(saver.path_log_info).read() == "model: 12,345\nhello\n"
    for step in (2, 4, 10):
        saver.global_step = step
        net.weight.data.fill_(float(step))
        net(torch.ones(1, 3)).sum().backward()
        opt.step()
        saver.save_model(net, opt if step == 10 else None, postfix=f"{step}")
    saver.delete_model(postfix="2")
    assert sorted(os.listdir(tmp_path / "exp")) == ["config.yaml", "log_info.txt", "model_10.pt", "model_4.pt"]
    net2 = torch.nn.Linear(3, 2)
    opt2 = torch.optim.AdamW(net2.parameters())
    gs, _, _ = LU.load_model(args.env.expdir, net2, opt2)
    assert gs == 10 and torch.equal(net2.weight, net.weight) and len(opt2.state) == 2     # newest step, optimizer state restored
    assert saver.get_total_time().count(":") == 2 and saver.get_interval_time() >= 0


def test_rank_sampler_partitions_each_epoch():
    from diffusion.data_loaders import _RankSampler
    for n, world in ((10, 4), (3, 4), (8, 2)):
        samplers = [_RankSampler(n, r, world) for r in range(world)]
        for _ in range(2):
            shards = [list(s) for s in samplers]
            assert len({len(s) for s in shards}) == 1 and len(shards[0]) == len(samplers[0]) == -(-n // world)
            assert set(sum(shards, [])) == set(range(n))                 # every item seen; wrap-around only pads
        e0 = list(_RankSampler(n, 0, world))
        s = _RankSampler(n, 0, world)
        list(s)
        assert n < 4 or list(s) != e0                                    # reshuffled per epoch


def test_train_diff_entry_point_refuses_without_a_gpu(tmp_path):
    if torch.cuda.is_available():
        pytest.skip("GPU present: exercised end to end below")
    (tmp_path / "train_diff.py").write_text("print('REFERENCE LOOP')\n")
    fl, vl = _make_dataset(str(tmp_path), n_items=2)
    cj = _config(str(tmp_path), fl, vl, str(tmp_path / "voc" / "model"))
    r = subprocess.run([sys.executable, os.path.join(PKG, "svc_run.py"), "train_diff.py", "-c", cj], capture_output=True, text=True,
                       cwd=str(tmp_path), timeout=300)
    assert "REFERENCE LOOP" not in r.stdout and r.returncode != 0
    assert "trains on the GPU only" in r.stderr, r.stderr[-1500:]


# ---------------------------------------------------------------- GPU ----------------------------------------------------
@pytest.mark.gpu
def test_train_diff_entry_point_trains_validates_checkpoints_and_resumes(dev, tmp_path):
    root = str(tmp_path)
    fl, vl = _make_dataset(root)
    cj = _config(root, fl, vl, _vocoder(root))
    env = dict(os.environ, SVC_LOADER_WORKERS="0")
    (tmp_path / "train_diff.py").write_text("raise SystemExit('the checkout script ran')\n")   # the launcher must substitute the engine's
    cmd = [sys.executable, os.path.join(PKG, "svc_run.py"), "train_diff.py", "-c", cj]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=root, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    exp = os.path.join(root, "logs", "diffusion")
    log = open(os.path.join(exp, "log_info.txt")).read()
    assert "======= start training =======" in log and log.count("| loss: ") == 6, log[-1500:]      # 3 batches x 2 epochs
    assert log.count("--- <validation> ---") == 3 and "Real Time Factor" in r.stdout and "RTF:" in r.stdout
    assert "nan" not in log.lower()
    # interval_val 2, force_save 4: model_2 deleted when model_4 is written (2 % 4 != 0), model_4 kept, model_6 newest
    names = sorted(os.path.basename(p) for p in glob.glob(os.path.join(exp, "model_*.pt")))
    assert names == ["model_4.pt", "model_6.pt"], names
    ck = torch.load(os.path.join(exp, "model_6.pt"), map_location="cpu")
    assert ck["global_step"] == 6 and "optimizer" in ck and any(k.startswith("decoder.denoise_fn.") for k in ck["model"])
    assert all(torch.isfinite(v).all() for v in ck["model"].values() if v.is_floating_point())
    # lr: StepLR(step 4, gamma 0.5) -> 2e-4 for steps 1..3, 1e-4 from the 4th scheduler step on
    lrs = [float(l.split("| lr: ")[1].split(" |")[0]) for l in log.splitlines() if "| lr: " in l]
    assert lrs[:3] == [2e-4] * 3 and lrs[-1] == 1e-4, lrs
    # resume: the newest checkpoint, step counter and schedule continue (train_diff.py:53-60)
    cfg = yaml.safe_load(open(cj))
    cfg["train"]["epochs"] = 1
    with open(cj, "w") as f:
        yaml.safe_dump(cfg, f)
    r2 = subprocess.run(cmd, capture_output=True, text=True, cwd=root, timeout=900, env=env)
    assert r2.returncode == 0, (r2.stdout[-2000:], r2.stderr[-3000:])
    assert "restoring model from" in r2.stdout and "model_6.pt" in r2.stdout
    log2 = open(os.path.join(exp, "log_info.txt")).read()[len(log):]
    steps = [int(l.rsplit("step: ", 1)[1]) for l in log2.splitlines() if "| step: " in l]
    assert steps == [7, 8, 9], steps
    lrs2 = [float(l.split("| lr: ")[1].split(" |")[0]) for l in log2.splitlines() if "| lr: " in l]
    assert lrs2[0] == 1e-4 and lrs2[-1] == 5e-5, lrs2               # decayed again at the 8th scheduler step
    ck9 = torch.load(os.path.join(exp, "model_8.pt"), map_location="cpu")
    moved = max((ck9["model"][k] - ck["model"][k]).abs().max().item() for k in ck["model"] if ck["model"][k].is_floating_point())
    assert 0 < moved < 1.0


@pytest.mark.gpu
def test_solver_train_accepts_the_reference_scripts_torch_optimizer(dev, tmp_path):
    """An UNCHANGED train_diff.py hands `train` a torch.optim.AdamW + StepLR: same loop, fused optimizer underneath."""
    from diffusion import solver
    from diffusion.data_loaders import get_data_loaders
    from diffusion.logger import utils as LU
    from diffusion.unit2mel import Unit2Mel
    from diffusion.vocoder import Vocoder
    root = str(tmp_path)
    fl, vl = _make_dataset(root)
    args = LU.load_config(_config(root, fl, vl, _vocoder(root), interval_val=100, epochs=1))
    voc = Vocoder(args.vocoder.type, args.vocoder.ckpt, device="cuda")
    torch.manual_seed(0)
    model = Unit2Mel(SSL, 2, True, voc.dimension, args.model.n_layers, args.model.n_chans, args.model.n_hidden, 100, 0).to(dev)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    opt = torch.optim.AdamW(model.parameters())
    for g in opt.param_groups:
        g["initial_lr"], g["lr"], g["weight_decay"] = 2e-4, 2e-4, 0
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=2, gamma=0.5, last_epoch=-2)
    loaders = get_data_loaders(args)
    end = solver.train(args, 0, model, opt, sched, voc, *loaders)
    assert end == 3
    assert sched.optimizer.__class__.__name__ == "FusedAdamW" and abs(sched.optimizer.param_groups[0]["lr"] - 1e-4) < 1e-12
    after = model.state_dict()
    assert any((after[k] - before[k]).abs().max() > 0 for k in before if before[k].is_floating_point())
