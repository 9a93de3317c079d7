"""Golden vectors for shallow-diffusion TRAINING from the REAL reference modules (build container only):
Unit2Mel.forward(infer=False) + torch.optim.AdamW/StepLR as train_diff.py:55-60 / diffusion/solver.py:116-147 run them.
usage: python tests/golden/make_golden_diffusion_train.py"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def make_batches(c, seed, B, T, n):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        f0 = 100 + 300 * torch.rand(B, T, 1, generator=g)
        f0[:, 3:6] = 0
        out.append(dict(units=torch.randn(B, T, c["input_channel"], generator=g), f0=f0, volume=torch.rand(B, T, 1, generator=g),
                        spk_id=torch.randint(0, c["n_spk"], (B, 1), generator=g),
                        gt=-6 + 3 * torch.randn(B, T, c["out_dims"], generator=g),
                        t=torch.randint(0, c["k_step_max"], (B,), generator=g),
                        noise=torch.randn(B, 1, c["out_dims"], T, generator=g)))
    return out


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from oracle import diffusion_oracle as DO
    for name in ("librosa", "librosa.filters", "soundfile", "torchaudio", "torchaudio.transforms"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["librosa.filters"].mel = lambda **k: None
    sys.modules["torchaudio.transforms"].Resample = object
    sys.path.insert(0, "/root/reference")
    from diffusion.unit2mel import Unit2Mel
    c = DO.small_cfg()
    seed, B, T, N, LR = 47, 3, 30, 4, 2e-3
    net = Unit2Mel(c["input_channel"], c["n_spk"], c["use_pitch_aug"], c["out_dims"], c["n_layers"], c["n_chans"],
                   c["n_hidden"], c["timesteps"], c["k_step_max"])
    sd = DO.make_state_dict(c, seed)
    net.load_state_dict(sd, strict=False)
    net.train()
    opt = torch.optim.AdamW(net.parameters())
    for pg in opt.param_groups:                       # train_diff.py:57-60
        pg["initial_lr"] = LR
        pg["lr"] = LR
        pg["weight_decay"] = 0
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=100000, gamma=0.5, last_epoch=-1)
    batches = make_batches(c, seed, B, T, N)
    losses, grads0 = [], None
    orig = torch.randint, torch.randn_like
    for bt in batches:
        torch.randint = lambda *a, **k: bt["t"]
        torch.randn_like = lambda x, **k: bt["noise"]
        try:
            opt.zero_grad()
            loss = net(bt["units"], bt["f0"], bt["volume"], bt["spk_id"], aug_shift=None, gt_spec=bt["gt"], infer=False,
                       k_step=net.k_step_max)
            loss.backward()
        finally:
            torch.randint, torch.randn_like = orig
        if grads0 is None:
            grads0 = {k: p.grad.clone() for k, p in net.named_parameters()}
        opt.step()
        sched.step()
        losses.append(float(loss))
    final = {k: v.detach() for k, v in net.named_parameters()}
    o_losses, o_g0, o_final = DO.train_loop(sd, c, batches, lr=LR)
    print("reference losses", losses)
    print("oracle    losses", o_losses)
    assert np.allclose(losses, o_losses, rtol=2e-5)
    gd = max((grads0[k] - o_g0[k]).abs().max().item() / max(1e-6, grads0[k].abs().max().item()) for k in o_g0)
    pd = max((final[k] - o_final[k]).abs().max().item() for k in o_final)
    print(f"oracle vs reference: max rel grad diff {gd:.3e}, max final-param diff {pd:.3e}")
    assert gd < 2e-4 and pd < 2e-4
    keys = sorted(o_g0)
    np.savez_compressed(os.path.join(HERE, "diffusion_train_small.npz"), losses=np.array(losses, np.float64),
                        **{"g0/" + k: grads0[k].numpy() for k in keys}, **{"final/" + k: final[k].numpy() for k in keys},
                        meta=json.dumps(dict(seed=seed, B=B, T=T, N=N, lr=LR)))
    print("wrote diffusion_train_small.npz")


if __name__ == "__main__":
    main()
