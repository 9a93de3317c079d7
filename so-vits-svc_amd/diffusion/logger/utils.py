"""Mirror of diffusion/logger/utils.py: the yaml config as a `DotDict` (:50-56,69-74), the run directory scan that finds the
newest `model_<step>.pt` (:106-131), the parameter count the training log opens with (:59-66).  Host-only; no device work.

`load_model` keeps the reference's contract — `(global_step, model, optimizer)`; `strict=False` on the weights; the
optimizer's state only when the checkpoint carries one (`train.save_opt`) — and works with `optim.FusedAdamW`, whose
`load_state_dict` moves the loaded moments into its flat arena."""
import json
import os

import torch
import yaml


def traverse_dir(root_dir, extensions, amount=None, str_include=None, str_exclude=None, is_pure=False, is_sort=False,
                 is_ext=True):
    """Files under `root_dir` with one of `extensions` (:8-46): full or root-relative paths, optionally without suffix."""
    out = []
    for root, _, files in os.walk(root_dir):
        for name in files:
            if not any(name.endswith("." + e) for e in extensions):
                continue
            full = os.path.join(root, name)
            path = full[len(root_dir) + 1:] if is_pure else full
            if amount is not None and len(out) == amount:
                return sorted(out) if is_sort else out
            if str_include is not None and str_include not in path:
                continue
            if str_exclude is not None and str_exclude in path:
                continue
            if not is_ext:
                path = path[:-(len(path.split(".")[-1]) + 1)]
            out.append(path)
    return sorted(out) if is_sort else out


class DotDict(dict):
    def __getattr__(*args):
        val = dict.get(*args)
        return DotDict(val) if type(val) is dict else val

    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__


def get_network_paras_amount(model_dict):
    return {name: sum(p.numel() for p in m.parameters() if p.requires_grad) for name, m in model_dict.items()}


def load_config(path_config):
    with open(path_config, "r") as f:
        return DotDict(yaml.safe_load(f))


def save_config(path_config, config):
    with open(path_config, "w") as f:
        yaml.dump(dict(config), f)


def to_json(path_params, path_json):
    params = torch.load(path_params, map_location="cpu")
    with open(path_json, "w") as f:
        json.dump({k: v.flatten().numpy().tolist() for k, v in params.items()}, f, indent="\t")


def convert_tensor_to_numpy(tensor, is_squeeze=True):
    if is_squeeze:
        tensor = tensor.squeeze()
    return tensor.detach().cpu().numpy()


def load_model(expdir, model, optimizer, name="model", postfix="", device="cpu"):
    """Newest `<expdir>/<name>_<step>.pt` (else `<name>_best.pt`) -> (global_step, model, optimizer); (0, ...) on a fresh
    directory (:106-131)."""
    prefix = os.path.join(expdir, name + ("_" + postfix if postfix == "" else postfix))
    found = traverse_dir(expdir, ["pt"], is_ext=False)
    global_step = 0
    if found:
        steps = [s[len(prefix):] for s in found]
        maxstep = max(int(s) if s.isdigit() else 0 for s in steps)
        path_pt = prefix + (str(maxstep) if maxstep >= 0 else "best") + ".pt"
        print(" [*] restoring model from", path_pt)
        ckpt = torch.load(path_pt, map_location=torch.device(device))
        global_step = ckpt["global_step"]
        model.load_state_dict(ckpt["model"], strict=False)
        if ckpt.get("optimizer") is not None and optimizer is not None:
            optimizer.load_state_dict(ckpt["optimizer"])
    return global_step, model, optimizer
