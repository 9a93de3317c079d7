"""Mirror of diffusion/logger/saver.py: the run directory of `train_diff.py` — `log_info.txt`, `config.yaml`,
`model_<step>.pt` checkpoints (`{'global_step', 'model'[, 'optimizer']}`, :102-127), TensorBoard scalars / spectrogram figures /
audio when tensorboard (and matplotlib) are installed, silently skipped otherwise.  Host-only."""
import datetime
import os
import time

import torch
import yaml


class _NullWriter:
    def __getattr__(self, name):
        return lambda *a, **k: None


def _writer(log_dir):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir)
    except Exception:      # noqa: BLE001 — tensorboard / protobuf missing or broken: the run still logs to log_info.txt
        return _NullWriter()


def _plain(obj):
    """DotDict tree -> plain dicts (yaml.dump refuses dict subclasses in safe mode and tags them in the default one)."""
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    return obj


class Saver(object):
    def __init__(self, args, initial_global_step=-1):
        self.expdir = args.env.expdir
        self.sample_rate = args.data.sampling_rate
        self.global_step = initial_global_step
        self.init_time = time.time()
        self.last_time = time.time()
        os.makedirs(self.expdir, exist_ok=True)
        self.path_log_info = os.path.join(self.expdir, "log_info.txt")
        self.writer = _writer(os.path.join(self.expdir, "logs"))
        with open(os.path.join(self.expdir, "config.yaml"), "w") as f:
            yaml.dump(_plain(args), f)

    def log_info(self, msg):
        if isinstance(msg, dict):
            msg = "\n".join(("{}: {:,}" if isinstance(v, int) else "{}: {}").format(k, v) for k, v in msg.items())
        print(msg)
        with open(self.path_log_info, "a") as fp:
            fp.write(msg + "\n")

    def log_value(self, values):
        for k, v in values.items():
            self.writer.add_scalar(k, v, self.global_step)

    def log_spec(self, name, spec, spec_out, vmin=-14, vmax=3.5):
        if isinstance(self.writer, _NullWriter):
            return
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except Exception:      # noqa: BLE001
            return
        cat = torch.cat([(spec_out - spec).abs() + vmin, spec, spec_out], -1)[0].float().cpu().numpy()
        fig = plt.figure(figsize=(12, 9))
        plt.pcolor(cat.T, vmin=vmin, vmax=vmax)
        plt.tight_layout()
        self.writer.add_figure(name, fig, self.global_step)
        plt.close(fig)

    def log_audio(self, audios):
        for k, v in audios.items():
            self.writer.add_audio(k, v, global_step=self.global_step, sample_rate=self.sample_rate)

    def get_interval_time(self, update=True):
        now = time.time()
        dt = now - self.last_time
        if update:
            self.last_time = now
        return dt

    def get_total_time(self, to_str=True):
        total = time.time() - self.init_time
        return str(datetime.timedelta(seconds=total))[:-5] if to_str else total

    def _path(self, name, postfix):
        return os.path.join(self.expdir, name + ("_" + postfix if postfix else "") + ".pt")

    def save_model(self, model, optimizer, name="model", postfix="", to_json=False):
        path_pt = self._path(name, postfix)
        print(" [*] model checkpoint saved: {}".format(path_pt))
        ckpt = {"global_step": self.global_step, "model": model.state_dict()}
        if optimizer is not None:
            ckpt["optimizer"] = optimizer.state_dict()
        torch.save(ckpt, path_pt)

    def delete_model(self, name="model", postfix=""):
        path_pt = self._path(name, postfix)
        if os.path.exists(path_pt):
            os.remove(path_pt)
            print(" [*] model checkpoint deleted: {}".format(path_pt))

    def global_step_increment(self):
        self.global_step += 1
