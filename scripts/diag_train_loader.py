"""Where does `train_loader` (bench_extra.bench_train_loader) lose time against the fixed-batch training step?
usage: python scripts/diag_train_loader.py   (env: SVC_LOADER_WORKERS, SVC_TRAIN_SERIALIZE) -> one JSON line"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "so-vits-svc_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
import bench_extra as X  # noqa: E402
import synthetic_data as W  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
out = X.bench_train_loader(dev, bench.train_hps(W.full_config()), n_items=int(os.environ.get("N_ITEMS", "96")),
                           epochs=int(os.environ.get("EPOCHS", "3")))
out["env"] = {k: os.environ.get(k) for k in ("SVC_LOADER_WORKERS", "SVC_TRAIN_SERIALIZE")}
print(json.dumps(out))
