"""The drop-in boundary on the host side: with this package AHEAD of a reference
checkout on sys.path, every module-level name the reference's `inference_main.py` and `train.py` touch resolves — the
hot-path modules to the MI355X engine, everything else (spkmix, cluster, modules.F0Predictor.*) to the checkout.

The name lists are not hand-written: the reference scripts are parsed (ast) for their `import` / `from ... import` statements
and for every `infer_tool.X` / `utils.X` / `commons.X` attribute access.  Third-party packages that are absent from this
image (soundfile, librosa, torchaudio, faiss, tensorboard) are stubbed in the child interpreter only.  Tests that need the
reference checkout skip when /root/reference is absent (the GPU box)."""
import ast
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "so-vits-svc_amd")
REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="no reference checkout on this machine")

STUBS = textwrap.dedent("""
    import sys, types
    for name in ("soundfile", "librosa", "librosa.filters", "torchaudio", "torchaudio.transforms", "faiss", "tensorboard",
                 "torch.utils.tensorboard"):
        try:
            __import__(name)
        except Exception:
            m = types.ModuleType(name)
            sys.modules[name] = m
            if "." in name:
                setattr(sys.modules[name.rsplit(".", 1)[0]], name.rsplit(".", 1)[1], m)
    if not hasattr(sys.modules["torch.utils.tensorboard"], "SummaryWriter"):
        sys.modules["torch.utils.tensorboard"].SummaryWriter = object
""")


def _script_surface(path):
    """(imports, attribute uses) of a reference script: [(module, [names] | None)], {alias: {attr, ...}}."""
    tree = ast.parse(open(path).read())
    imports, aliases = [], {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                imports.append((a.name, None))
                aliases[(a.asname or a.name).split(".")[0]] = a.name
        elif isinstance(node, ast.ImportFrom) and node.level == 0:
            imports.append((node.module, [a.name for a in node.names]))
            for a in node.names:
                aliases[a.asname or a.name] = node.module + "." + a.name
    uses = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in ("infer_tool", "utils", "commons"):
            uses.setdefault(aliases[node.value.id], set()).add(node.attr)
    return imports, uses


@needs_ref
@pytest.mark.parametrize("script", ["inference_main.py", "train.py"])
def test_every_name_the_reference_entry_point_touches_resolves(script):
    imports, uses = _script_surface(os.path.join(REF, script))
    assert imports and uses
    if script == "inference_main.py":
        assert {"read_temp", "mkdir", "fill_a_to_b", "format_wav"} <= uses["inference.infer_tool"]
    else:
        assert {"get_hparams", "get_logger", "check_git_hash", "summarize", "clean_checkpoints", "load_checkpoint",
                "latest_checkpoint_path", "save_checkpoint", "plot_spectrogram_to_numpy", "plot_data_to_numpy"} <= uses["utils"]
    ours = {"models", "utils", "data_utils", "inference", "inference.infer_tool", "modules.commons", "modules.losses",
            "modules.mel_processing"}
    code = STUBS + textwrap.dedent(f"""
        import importlib, json, os, sys
        sys.path[:0] = [{PKG!r}, {REF!r}]
        imports = {imports!r}
        uses = {{k: sorted(v) for k, v in {({k: sorted(v) for k, v in uses.items()})!r}.items()}}
        where = {{}}
        for mod, names in imports:
            m = importlib.import_module(mod)
            where[mod] = getattr(m, "__file__", None)
            for n in names or []:
                try:
                    getattr(m, n)
                except AttributeError:
                    importlib.import_module(mod + "." + n)      # `from inference import infer_tool`
        for mod, attrs in uses.items():
            m = importlib.import_module(mod)
            for a in attrs:
                assert getattr(m, a) is not None, (mod, a)
        # names the engine does not mirror fall through to the checkout
        import modules.F0Predictor.F0Predictor as F0P
        import cluster, spkmix
        where["modules.F0Predictor.F0Predictor"] = F0P.__file__
        where["cluster"] = cluster.__file__
        where["spkmix"] = spkmix.__file__
        print("WHERE=" + json.dumps(where))
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(ROOT), timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    where = json.loads(r.stdout.split("WHERE=")[1])
    for mod, f in where.items():
        if mod in ours:
            assert f and f.startswith(PKG), (mod, f)
    for mod in ("modules.F0Predictor.F0Predictor", "cluster", "spkmix"):
        assert where[mod].startswith(REF), (mod, where[mod])


def test_launcher_puts_the_engine_first(tmp_path):
    """`python svc_run.py <script>`: the script's own directory (a checkout look-alike with a models.py) must come AFTER
    the engine, which a plain `python <script>` + PYTHONPATH cannot arrange."""
    (tmp_path / "vdecoder").mkdir()
    (tmp_path / "models.py").write_text("raise RuntimeError('the reference models.py was imported')\n")
    (tmp_path / "spkmix.py").write_text("spk_mix_map = {0: [[0., 1., 1., 1.]]}\n")
    (tmp_path / "probe.py").write_text("import sys, models, utils, spkmix\nfrom inference import infer_tool\n"
                                       "print('MODELS=' + models.__file__)\nprint('SPKMIX=' + spkmix.__file__)\n"
                                       "print('ARGV=' + ','.join(sys.argv[1:]))\nassert __name__ == '__main__'\n")
    r = subprocess.run([sys.executable, os.path.join(PKG, "svc_run.py"), "probe.py", "-m", "x"], capture_output=True, text=True,
                       cwd=str(tmp_path), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "MODELS=" + os.path.join(PKG, "models.py") in r.stdout
    assert "SPKMIX=" + str(tmp_path / "spkmix.py") in r.stdout and "ARGV=-m,x" in r.stdout
    # and the naive recipe really does pick the checkout's file (why the launcher exists)
    r2 = subprocess.run([sys.executable, "probe.py"], capture_output=True, text=True, cwd=str(tmp_path), timeout=300,
                        env=dict(os.environ, PYTHONPATH=PKG))
    assert r2.returncode != 0 and "reference models.py was imported" in r2.stderr


def test_launcher_train_entry_point_starts_training(tmp_path):
    """`svc_run.py train.py -c ... -m ...` substitutes the engine's train.py for the checkout's — which must then
    actually RUN (round 2's had no __main__: the command printed nothing and exited 0).  Without a GPU the engine's main() has
    to refuse loudly, exactly like the reference's (`assert torch.cuda.is_available(), "CPU training is not allowed."`,
    train.py:37), after having parsed the reference's CLI and created logs/<model>/config.json."""
    (tmp_path / "train.py").write_text("print('REFERENCE LOOP')\n")
    cfg = dict(train=dict(port="8001"), data={}, model={})
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    r = subprocess.run([sys.executable, os.path.join(PKG, "svc_run.py"), "train.py", "-c", "cfg.json", "-m", "unit"],
                       capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert "REFERENCE LOOP" not in r.stdout
    if torch.cuda.is_available():
        pytest.skip("GPU present: the training entry point is exercised end to end by tests/test_train_gpu.py")
    assert r.returncode != 0, "the launcher returned success without training"
    assert "CPU training is not allowed" in r.stderr, r.stderr[-1500:]


def _reference_module(name, relpath):
    """Import a reference file under an alias with the absent third-party imports stubbed (test-side only)."""
    import importlib.util
    import types
    added = []
    for stub in ("soundfile", "librosa", "torchaudio", "faiss", "cluster"):
        if stub not in sys.modules:
            try:
                __import__(stub)
            except Exception:
                sys.modules[stub] = types.ModuleType(stub)
                added.append(stub)
    try:
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for stub in added:                      # the stubs stay bound inside `mod`, but must not leak into other tests
            del sys.modules[stub]
    return mod


@needs_ref
def test_slicer_matches_reference_state_machine():
    """inference/slicer.py: the mirror's cut decisions vs the REAL reference Slicer (its librosa.feature.rms call served by
    the same RMS restatement, so this pins the silence state machine, chunk table and chunk extraction)."""
    import svc_audio
    from inference import slicer as mine
    from oracle import audio_oracle as AO
    ref = _reference_module("_ref_slicer", "inference/slicer.py")
    ref.librosa.feature = type("F", (), {"rms": staticmethod(
        lambda y, frame_length, hop_length: AO.frame_rms(y, frame_length, hop_length)[None, :])})
    ref.librosa.to_mono = lambda w: w.mean(axis=0)
    sr = 16000
    rng = np.random.default_rng(0)
    for case in range(12):
        # voiced bursts separated by silences of assorted lengths (incl. leading / trailing / very long ones)
        parts = []
        if case % 3 == 0:
            parts.append(np.zeros(int(sr * rng.uniform(0.1, 7.0))))
        for _ in range(rng.integers(2, 6)):
            parts.append(0.3 * rng.standard_normal(int(sr * rng.uniform(0.3, 7.0))))
            parts.append(1e-4 * rng.standard_normal(int(sr * rng.choice([0.05, 0.2, 0.5, 1.5, 6.0, 11.0]))))
        if case % 2 == 0:
            parts.pop()
        wav = np.concatenate(parts).astype(np.float32)
        for kw in (dict(threshold=-40.0), dict(threshold=-30.0, min_length=2000, min_interval=200, max_sil_kept=1000)):
            a = mine.Slicer(sr=sr, **kw).slice(wav)
            b = ref.Slicer(sr=sr, **kw).slice(wav)
            assert a == b, (case, kw)
    assert np.abs(svc_audio.frame_rms(wav, 1200, 320) - AO.frame_rms(wav, 1200, 320)).max() < 1e-6


@needs_ref
def test_infer_tool_helpers_match_reference(tmp_path):
    from inference import infer_tool as mine
    # the reference module imports the whole model stack; only its pure helpers are wanted: exec their source in isolation
    src = open(os.path.join(REF, "inference/infer_tool.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in
            ("read_temp", "write_temp", "get_end_file", "get_md5", "fill_a_to_b", "mkdir", "pad_array", "split_list_by_n")]
    ns = {}
    exec("import os, json, time, hashlib\nimport numpy as np\n" + "\n".join(ast.get_source_segment(src, n) for n in keep), ns)
    for n in (0, 3, 10, 11):
        for tgt in (0, 5, 10, 17):
            arr = np.arange(n, dtype=np.float32)
            assert np.array_equal(mine.pad_array(arr, tgt), ns["pad_array"](arr, tgt)), (n, tgt)
    for n, pre in ((4, 0), (4, 2), (7, 3)):
        seq = list(range(23))
        assert [list(x) for x in mine.split_list_by_n(seq, n, pre)] == [list(x) for x in ns["split_list_by_n"](seq, n, pre)]
    a, b = [5], [1, 2, 3]
    a2 = [5]
    mine.fill_a_to_b(a, b)
    ns["fill_a_to_b"](a2, b)
    assert a == a2 == [5, 5, 5]
    assert mine.get_md5(b"abc") == ns["get_md5"](b"abc")
    f1, f2 = str(tmp_path / "a.json"), str(tmp_path / "b.json")
    assert mine.read_temp(f1) == ns["read_temp"](f2) == {}
    assert mine.read_temp(f1) == ns["read_temp"](f2) == {"info": "temp_dict"}
    open(f1, "w").write("{broken")
    assert mine.read_temp(f1) == {"info": "temp_dict"}
    d = tmp_path / "tree"
    (d / "x" / ".hid").mkdir(parents=True)
    for f in ("x/a.wav", "x/.b.wav", "x/.hid/c.wav", "d.wav", "e.txt"):
        (d / f).write_text("")
    assert sorted(mine.get_end_file(str(d), "wav")) == sorted(ns["get_end_file"](str(d), "wav"))
    mine.mkdir([str(tmp_path / "m1"), str(tmp_path / "m1")])
    assert (tmp_path / "m1").is_dir()


def test_utils_run_directory_helpers(tmp_path, monkeypatch):
    sys.path.insert(0, PKG)
    import utils
    cfg = tmp_path / "c.json"
    cfg.write_text(json.dumps(dict(train=dict(port="8001"), model=dict(a=1))))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(sys, "argv", ["train.py", "-c", str(cfg), "-m", "run1"])
    hps = utils.get_hparams()
    assert hps.model_dir == os.path.join("./logs", "run1") and hps.train.port == "8001"
    assert json.load(open(os.path.join(hps.model_dir, "config.json")))["model"]["a"] == 1
    assert utils.get_hparams(init=False).model.a == 1
    lg = utils.get_logger(hps.model_dir)
    lg.info("hello")
    assert "hello" in open(os.path.join(hps.model_dir, "train.log")).read()
    for i, n in enumerate([0, 100, 200, 300]):
        for p in "GD":
            f = os.path.join(hps.model_dir, f"{p}_{n}.pth")
            open(f, "w").write("x")
            os.utime(f, (1000 + i, 1000 + i))
    utils.clean_checkpoints(hps.model_dir, n_ckpts_to_keep=2, sort_by_time=True)
    assert sorted(f for f in os.listdir(hps.model_dir) if f.endswith(".pth")) == sorted(
        [f"{p}_{n}.pth" for p in "GD" for n in (0, 200, 300)])
    assert utils.latest_checkpoint_path(hps.model_dir, "G_*.pth").endswith("G_300.pth")

    class W_:
        def __init__(self):
            self.calls = []

        def __getattr__(self, name):
            return lambda *a, **k: self.calls.append((name, a[0]))
    w = W_()
    img = utils.plot_spectrogram_to_numpy(np.random.rand(20, 30))
    assert img.ndim == 3 and img.shape[2] == 3 and img.dtype == np.uint8
    assert utils.plot_data_to_numpy(np.arange(5.0), np.arange(5.0) ** 2).shape[2] == 3
    utils.summarize(w, 7, scalars={"a": 1.0}, images={"i": img}, audios={"au": np.zeros(4)})
    assert sorted(w.calls) == [("add_audio", "au"), ("add_image", "i"), ("add_scalar", "a")]
    with pytest.raises(Exception):
        utils.get_speech_encoder("no-such-encoder")
    with pytest.raises(Exception):
        utils.get_f0_predictor("no-such-predictor", 512, 44100)
    with pytest.raises(AttributeError):
        utils.definitely_not_a_name


def test_fairseq_checkpoint_maps_onto_the_hubert_mirror(tmp_path):
    """vencoder/ContentVec768L12.py:12-15 loads `checkpoint_best_legacy_500.pt` through fairseq; here a synthetic checkpoint
    with fairseq's HubertModel key names (and fairseq-only pickled objects in `cfg`) must load without fairseq and give back
    the same tensors under the mirror's names."""
    from oracle import hubert_oracle as HO
    from vencoder.hubert import hubert_model as HM
    sd = HO.make_state_dict(3)
    fs = HO.to_fairseq_state_dict(sd)
    assert len(fs) == 166 + 12 * 4       # q/k/v stored separately

    import types
    fake = types.ModuleType("fairseq_not_installed_cfg")
    exec("class HubertConfig:\n    pass\n", fake.__dict__)
    sys.modules["fairseq_not_installed_cfg"] = fake
    obj = fake.HubertConfig()
    obj.label_rate = 50
    path = str(tmp_path / "checkpoint_best_legacy_500.pt")
    torch.save({"model": fs, "cfg": {"model": obj}, "args": None}, path)
    del sys.modules["fairseq_not_installed_cfg"]                       # "fairseq" is gone at load time
    net = HM.load_fairseq_hubert(path)
    got = net.state_dict()
    for k, v in sd.items():
        if k != "label_embedding.weight":
            assert torch.equal(got[k], v), k


def test_resample_bank_and_pcm16():
    import svc_audio
    bank, width = svc_audio.sinc_resample_bank(441, 160)
    assert bank.shape == (2 * width + 441, 160) and width == 17
    assert np.abs(bank.sum(axis=0) - 1.0).max() < 2e-3              # unit DC gain in every output phase
    x = np.array([0.0, 0.5, -0.5, 1.0, -1.0, 1e-5])
    q = svc_audio.pcm16_round_trip(x)
    assert np.allclose(q, np.round(x * 32767) / 32768) and q.dtype == np.float32
    f = svc_audio.wav_bytes(x, 8000)
    a, sr = svc_audio.read_audio(f)
    assert sr == 8000 and a.shape == (1, 6) and np.abs(a[0] - q).max() < 1e-7
