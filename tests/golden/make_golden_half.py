"""Golden vectors of the reference's HALF-PRECISION inference (inference/infer_tool.py:196-198: `net_g_ms.half()`), from the REAL
reference run on CPU (build container only; see make_golden.py): the full template at T = 24 (the case of infer_full_T24.npz: same
weights, inputs and injected noise), every parameter and input cast to fp16 exactly as Svc does (:198, :289-291).

Stored: the reference's fp16 output and its distance to the reference's own fp32 output — the reference's half mode is NOT close to
its fp32 mode (the harmonic source integrates its phase in fp16), which is the yardstick the engine's 16-bit pipeline is held to:
closer to the fp32 result than the reference's half mode is, and inside north_star's waveform bar (MSE < 1e-4).

usage: python tests/golden/make_golden_half.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import NoiseInjector, build_ref_model, import_reference  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    models, utils = import_reference()
    from oracle import weights as W
    z = np.load(os.path.join(HERE, "infer_full_T24.npz"))
    meta = json.loads(str(z["meta"]))
    cfg = W.full_config()
    sd = W.make_state_dict(cfg, meta["seed"])
    net = build_ref_model(models, cfg, sd)
    t = lambda k: torch.from_numpy(z[k])
    c, f0, uv, sid = t("c"), t("f0"), t("uv"), t("sid")
    noise = [t("noise_enc_p"), t("noise_rand_ini"), t("noise_sine"), None]
    with NoiseInjector(list(noise)), torch.no_grad():
        o32, _ = net.infer(c, f0, uv, g=sid, noice_scale=meta["noice_scale"])
    assert (o32 - t("o")).abs().max().item() < 1e-6
    neth = net.half()
    with NoiseInjector([n.half() if n is not None else None for n in noise]), torch.no_grad():
        oh, _ = neth.infer(c.half(), f0.half(), uv.half(), g=sid, noice_scale=meta["noice_scale"])
    assert oh.dtype == torch.float16
    d = oh.float() - o32
    rep = dict(mse_half_vs_fp32=d.pow(2).mean().item(), max_half_vs_fp32=d.abs().max().item(), max_ref=o32.abs().max().item())
    print("[full_T24 half] reference .half() vs reference fp32:", rep)
    np.savez_compressed(os.path.join(HERE, "infer_full_T24_half.npz"), o_half=oh.numpy(), meta=json.dumps(dict(meta, **rep)))


if __name__ == "__main__":
    main()
