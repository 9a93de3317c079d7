"""Golden vectors for the DPM-Solver / DPM-Solver++ samplers (the reference's defaults) and UniPC from the REAL reference modules:
Unit2Mel -> GaussianDiffusion.forward(method='dpm-solver' / 'dpm-solver++') -> diffusion/dpm_solver_pytorch.py.
Inputs are those of diffusion_small.npz.  usage: python tests/golden/make_golden_diffusion_dpm.py"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CASES = [("dpm_full", "dpm-solver", 10, False, None), ("dpmpp_full", "dpm-solver++", 10, False, None),
         ("dpm_shallow", "dpm-solver", 5, True, 40), ("dpmpp_shallow", "dpm-solver++", 5, True, 40),
         ("dpmpp_shallow3", "dpm-solver++", 10, True, 30),
         # UniPC (diffusion/diffusion.py:339-371 -> diffusion/uni_pc.py, variant bh2, multistep order 2)
         ("unipc_full", "unipc", 10, False, None), ("unipc_shallow", "unipc", 5, True, 40), ("unipc_shallow3", "unipc", 10, True, 30)]


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
    z = np.load(os.path.join(HERE, "diffusion_small.npz"))
    meta = json.loads(str(z["meta"]))
    c = DO.small_cfg()
    net = Unit2Mel(c["input_channel"], c["n_spk"], c["use_pitch_aug"], c["out_dims"], c["n_layers"], c["n_chans"],
                   c["n_hidden"], c["timesteps"], c["k_step_max"])
    sd = DO.make_state_dict(c, meta["seed"])
    net.load_state_dict(sd, strict=False)
    net.eval()
    t = lambda k: torch.from_numpy(z[k])
    units, f0, volume, spk_id, gt, x_T = t("units"), t("f0"), t("volume"), t("spk_id"), t("gt"), t("x_T")
    cond = DO.condition(sd, c, units, f0, volume, spk_id)
    out = {}
    for name, method, speedup, shallow, k_step in CASES:
        orig = torch.randn, torch.randn_like
        torch.randn = lambda *a, **k: x_T.clone()
        torch.randn_like = lambda x, **k: x_T.clone()
        try:
            with torch.no_grad():
                ref = net(units, f0, volume, spk_id=spk_id, gt_spec=gt if shallow else None, infer=True, infer_speedup=speedup,
                          method=method, k_step=k_step if shallow else 300, use_tqdm=False)
        finally:
            torch.randn, torch.randn_like = orig
        with torch.no_grad():
            mine = DO.sample(sd, c, cond, method, speedup, gt_spec=gt if shallow else None, k_step=k_step, x_T=x_T)
        d = (mine - ref).abs().max().item()
        print(f"[{name}] oracle vs reference: max|diff| {d:.3e}, max|ref| {ref.abs().max().item():.3e}, shape {tuple(ref.shape)}")
        assert d <= 5e-5 * max(1.0, ref.abs().max().item())
        out["mel_" + name] = ref.numpy()
    np.savez_compressed(os.path.join(HERE, "diffusion_dpm_small.npz"), **out)
    print("wrote diffusion_dpm_small.npz")


if __name__ == "__main__":
    main()
