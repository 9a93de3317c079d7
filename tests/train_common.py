"""Shared set-up of the training-parity tests (CPU oracle pin + GPU HIP-vs-oracle): the exact case stored in
tests/golden/train_small.npz (generated from the real reference by tests/golden/make_golden_train.py)."""
import json
import os

import numpy as np
import torch

from oracle import mel as OM
from oracle import weights as W

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case():
    z = np.load(os.path.join(G, "train_small.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    cfg = W.train_config()
    data = meta["data"]
    cfg["spec_channels"] = data["n_fft"] // 2 + 1
    cfg.update(upsample_rates=meta["upsample_rates"], upsample_kernel_sizes=meta["upsample_kernel_sizes"])
    B, T, seed, hop = meta["B"], meta["T"], meta["seed"], data["hop"]
    sd_g = W.make_train_state_dict(cfg, seed)
    sd_d = W.make_mpd_state_dict(seed + 1)
    batch = W.make_train_batch(cfg, B, T, seed, hop=hop)
    noise = W.make_train_noise(cfg, B, T, batch[-1], seed + 2, hop=hop)
    mel_basis = torch.from_numpy(OM.mel_filterbank(data["sr"], data["n_fft"], data["n_mels"], data["fmin"], data["fmax"]))
    return dict(z=z, meta=meta, cfg=cfg, data=data, sd_g=sd_g, sd_d=sd_d, batch=batch, noise=noise, mel_basis=mel_basis)


LOSS_KEYS = ["loss_disc", "loss_gen", "loss_fm", "loss_mel", "loss_kl", "loss_lf0", "loss_gen_all"]


def load_dropout_case():
    """The p_dropout = 0.1 case of tests/golden/train_dropout_small.npz (REAL reference with injected dropout draws,
    tests/golden/make_golden_train_dropout.py): same batch / weights / noise as load_case() plus noise["dropout_u"]."""
    z = np.load(os.path.join(G, "train_dropout_small.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    cs = load_case()
    cfg = dict(cs["cfg"], p_dropout=meta["p_dropout"])
    noise = dict(cs["noise"])
    noise["dropout_u"] = W.make_dropout_draws(cfg, meta["B"], meta["T"], meta["seed"] + 3)
    assert len(noise["dropout_u"]) == meta["n_sites"]
    return dict(cs, z=z, meta=meta, cfg=cfg, noise=noise)


def load_transflow_case():
    """use_transformer_flow = True with p_dropout = 0.1 (tests/golden/train_transflow_small.npz, REAL reference with
    injected dropout draws, tests/golden/make_golden_transflow.py): its own weights (seed 23), same batch recipe."""
    z = np.load(os.path.join(G, "train_transflow_small.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    cfg = W.train_config()
    data = meta["data"]
    cfg["spec_channels"] = data["n_fft"] // 2 + 1
    cfg.update(upsample_rates=meta["upsample_rates"], upsample_kernel_sizes=meta["upsample_kernel_sizes"],
               p_dropout=meta["p_dropout"], use_transformer_flow=True, n_layers_trans_flow=meta["n_layers_trans_flow"])
    B, T, seed, hop = meta["B"], meta["T"], meta["seed"], data["hop"]
    sd_g = W.make_train_state_dict(cfg, seed)
    batch = W.make_train_batch(cfg, B, T, seed, hop=hop)
    noise = W.make_train_noise(cfg, B, T, batch[-1], seed + 2, hop=hop)
    noise["dropout_u"] = W.make_dropout_draws(cfg, B, T, seed + 3)
    assert len(noise["dropout_u"]) == meta["n_sites"]
    return dict(z=z, meta=meta, cfg=cfg, data=data, sd_g=sd_g, batch=batch, noise=noise)
