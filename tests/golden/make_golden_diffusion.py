"""Golden vectors for the shallow-diffusion model from the REAL reference modules (build container only).
usage: python tests/golden/make_golden_diffusion.py"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


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
    seed, B, T = 31, 2, 30
    net = Unit2Mel(c["input_channel"], c["n_spk"], c["use_pitch_aug"], c["out_dims"], c["n_layers"], c["n_chans"],
                   c["n_hidden"], c["timesteps"], c["k_step_max"])
    sd = DO.make_state_dict(c, seed)
    learn = {k: tuple(v.shape) for k, v in net.state_dict().items() if k in sd}
    assert learn == {k: tuple(v.shape) for k, v in sd.items()} and len(learn) == len(DO.param_shapes(c))
    net.load_state_dict(sd, strict=False)
    net.eval()
    g = torch.Generator().manual_seed(seed)
    units = torch.randn(B, T, c["input_channel"], generator=g)
    f0 = (100 + 300 * torch.rand(B, T, 1, generator=g))
    f0[:, 3:6] = 0
    volume = torch.rand(B, T, 1, generator=g)
    spk_id = torch.tensor([[0], [2]])
    gt = -6 + 3 * torch.randn(B, T, c["out_dims"], generator=g)
    x_T = torch.randn(B, 1, c["out_dims"], T, generator=g)
    K = 12
    steps = [torch.randn(B, 1, c["out_dims"], T, generator=g) for _ in range(K)]
    out = {}

    def run(method, speedup, gt_spec, k_step, queue, nb=B):
        q = [t[:nb].clone() if t.dim() == 4 else t.clone() for t in queue]
        orig_randn, orig_randn_like = torch.randn, torch.randn_like
        torch.randn = lambda *a, **k: q.pop(0)
        torch.randn_like = lambda t, **k: q.pop(0)
        try:
            with torch.no_grad():
                return net(units[:nb], f0[:nb], volume[:nb], spk_id=spk_id[:nb], gt_spec=None if gt_spec is None else gt_spec[:nb],
                           infer=True, infer_speedup=speedup, method=method, k_step=k_step, use_tqdm=False)
        finally:
            torch.randn, torch.randn_like = orig_randn, orig_randn_like

    cond = DO.condition(sd, c, units, f0, volume, spk_id)
    cases = [("ddim_full", "ddim", 10, None, None, [x_T], None), ("pndm_full", "pndm", 10, None, None, [x_T], None),
             ("ddim_shallow", "ddim", 5, gt, 40, [x_T], None), ("naive_shallow", None, 1, gt, K, [x_T] + steps, steps)]
    for name, method, speedup, gts, k_step, queue, sn in cases:
        nb = 1 if method == "pndm" else B      # the reference's PLMS step does `max(t - interval, 0)` on a [B] tensor (:189): B = 1 only
        ref = run(method, speedup, gts, k_step, queue, nb)
        with torch.no_grad():
            mine = DO.sample(sd, c, cond[:nb], method, speedup, gt_spec=None if gts is None else gts[:nb], k_step=k_step,
                             x_T=x_T[:nb], step_noise=sn)
        d = (mine - ref).abs().max().item()
        print(f"[{name}] oracle vs reference: max|diff| {d:.3e}, max|ref| {ref.abs().max().item():.3e}, shape {tuple(ref.shape)}")
        assert d <= 5e-5 * max(1.0, ref.abs().max().item())
        out["mel_" + name] = ref.numpy()
    np.savez_compressed(os.path.join(HERE, "diffusion_small.npz"), units=units.numpy(), f0=f0.numpy(), volume=volume.numpy(),
                        spk_id=spk_id.numpy(), gt=gt.numpy(), x_T=x_T.numpy(), steps=torch.stack(steps).numpy(), **out,
                        meta=json.dumps(dict(seed=seed, B=B, T=T, K=K)))
    print("wrote diffusion_small.npz")


if __name__ == "__main__":
    main()
