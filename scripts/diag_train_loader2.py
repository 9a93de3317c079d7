"""Fine-grained host timeline of the loader-driven training loop (diagnostic): per step the time in next(loader), in the
host->device copies, in TrainStep.__call__, and the GPU time between replays.  usage: python scripts/diag_train_loader2.py"""
import json
import logging
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "so-vits-svc_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
import bench_extra as X  # noqa: E402
import synthetic_data as W  # noqa: E402
import train as TR  # noqa: E402
import utils  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
root = tempfile.mkdtemp(prefix="svc_diag_ds_")
try:
    fl = X.write_train_dataset(root, 96)
    cfg = W.full_config()
    h = bench.train_hps(cfg)
    h["train"] = dict(h["train"], use_sr=True, max_speclen=512, vol_aug=False, all_in_mem=os.environ.get("ALL_IN_MEM", "0") == "1",
                      log_interval=10 ** 9, eval_interval=10 ** 9, seed=1234, keep_ckpts=0, epochs=3)
    h["data"] = dict(h["data"], training_files=fl, validation_files=fl, max_wav_value=32768.0, unit_interpolate_mode="nearest")
    h["spk"] = {f"spk{i}": i for i in range(4)}
    hps = utils.HParams(**h)
    hps.model_dir = root
    net_g, net_d, og, od = TR.build(hps, dev)
    net_g.module.load_state_dict(W.make_train_state_dict(cfg, 1234))
    net_d.module.load_state_dict(W.make_mpd_state_dict(1235))
    net_g.train(); net_d.train()
    step = TR.TrainStep(hps, net_g, net_d, og, od).enable_graph(True)
    loader, _ = TR.make_loaders(hps, 0, 1, True)
    rows = []
    for epoch in range(3):
        it = iter(loader)
        torch.cuda.synchronize()
        te = time.perf_counter()
        while True:
            t0 = time.perf_counter()
            try:
                items = next(it)
            except StopIteration:
                break
            t1 = time.perf_counter()
            pinned = [bool(t.is_pinned()) for t in items if torch.is_tensor(t)]
            dev_items = TR._to_device(items, dev)
            t2 = time.perf_counter()
            step(dev_items)
            t3 = time.perf_counter()
            rows.append(dict(epoch=epoch, next_ms=round(1e3 * (t1 - t0), 2), h2d_ms=round(1e3 * (t2 - t1), 2), step_ms=round(1e3 * (t3 - t2), 2),
                             pinned=all(pinned), mb=round(sum(t.numel() * t.element_size() for t in items if torch.is_tensor(t)) / 1e6, 1)))
        torch.cuda.synchronize()
        print(f"epoch {epoch}: {1e3 * (time.perf_counter() - te) / len(loader):.1f} ms/step", file=sys.stderr)
    print(json.dumps(dict(env={k: os.environ.get(k) for k in ("SVC_TRAIN_SERIALIZE", "SVC_LOADER_WORKERS", "ALL_IN_MEM")}, rows=rows[6:])))
finally:
    shutil.rmtree(root, ignore_errors=True)
