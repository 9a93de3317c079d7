"""Golden vector for the HuBERT-soft unit encoder from the REAL in-tree module (build container only).

Imports /root/reference/vencoder/hubert/hubert_model.py unmodified, loads the deterministic synthetic checkpoint
(oracle.hubert_oracle.make_state_dict; the real hubert-soft checkpoint is not shipped), runs HubertSoft.units on a seeded
1 s waveform, asserts that the oracle restatement reproduces it and stores wav + units.

usage: python tests/golden/make_golden_hubert.py
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from oracle import hubert_oracle as HO
    spec = importlib.util.spec_from_file_location("ref_hubert", "/root/reference/vencoder/hubert/hubert_model.py")
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    net = R.HubertSoft()
    ref_shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert ref_shapes == {k: tuple(v) for k, v in HO.param_shapes().items()}
    seed = 77
    sd = HO.make_state_dict(seed)
    net.load_state_dict(sd)
    net.eval()
    g = torch.Generator().manual_seed(seed)
    n = 16000
    t = torch.arange(n) / 16000.0
    wav = (0.3 * torch.sin(2 * torch.pi * 220 * t) + 0.1 * torch.randn(n, generator=g)).view(1, 1, n)
    u_ref = net.units(wav)
    with torch.no_grad():
        u = HO.units(sd, wav)
    d = (u - u_ref).abs().max().item()
    print(f"oracle vs reference: max|diff| {d:.3e}, max|ref| {u_ref.abs().max().item():.3e}, rms {u_ref.pow(2).mean().sqrt().item():.3e}, shape {tuple(u_ref.shape)}")
    assert d <= 2e-5 * max(1.0, u_ref.abs().max().item())
    np.savez_compressed(os.path.join(HERE, "hubert_soft_1s.npz"), wav=wav.numpy(), units=u_ref.numpy(),
                        meta=json.dumps(dict(seed=seed, n=n)))
    print("wrote hubert_soft_1s.npz")


if __name__ == "__main__":
    main()
