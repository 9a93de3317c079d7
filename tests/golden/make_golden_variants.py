"""Golden vectors for the model VARIANTS from the REAL reference (build container only; see make_golden.py):
  snake_T40   small config with vocoder_name="nsf-snake-hifigan" (vdecoder/hifiganwithsnake, SnakeAlias activations)
  tiny_T40    small config with the tiny template's switches (configs_template/config_tiny_template.json:
              use_depthwise_conv, flow_share_parameter, odd decoder widths 100/50/25/12/6)
  tinyfull_T24  configs_template/config_tiny_template.json:42-71 at its REAL widths (filter 512, upsample_initial_channel
              400 -> decoder 200/100/50/25/12 channels, depthwise WN, one WN shared by the four flows): BASELINE configs[0]
  mixvol_T40  small config with vol_embedding=True, infer(..., vol=...) after EnableCharacterMix(4) with a per-frame
              speaker-mix matrix g [T, 4] (models.py:456-461,505-509,517)

usage: python tests/golden/make_golden_variants.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, run_case  # noqa: E402


def run_mixvol(models):
    import json
    import numpy as np
    from make_golden import NoiseInjector, build_ref_model
    from oracle import svc_oracle as O
    from oracle import weights as W
    cfg = W.small_config()
    cfg["vol_embedding"] = True
    seed, T, S = 15, 40, cfg["n_speakers"]
    sd = W.make_state_dict(cfg, seed)
    net = build_ref_model(models, cfg, sd)
    net.EnableCharacterMix(S, torch.device("cpu"))
    c, f0, uv, _ = W.make_inputs(cfg, 1, T, seed)
    gen = torch.Generator().manual_seed(seed)
    mix = torch.softmax(2.0 * torch.randn(T, S, generator=gen), dim=1)
    vol = torch.rand(1, T, generator=gen)
    noise = W.make_noise(cfg, 1, T, seed + 1)
    with NoiseInjector([noise["enc_p"], noise["rand_ini"], noise["sine"], None]), torch.no_grad():
        o_ref, _ = net.infer(c, f0, uv, g=mix, noice_scale=0.4, vol=vol)
    with torch.no_grad():
        o, _ = O.synth_infer(sd, cfg, c, f0, uv, None, noise, noice_scale=0.4, vol=vol, g_mix=mix)
    d = (o - o_ref).abs().max().item()
    print(f"[mixvol_T40] oracle vs reference max|diff| {d:.3e}  max|ref| {o_ref.abs().max().item():.3e}")
    assert d <= 2e-5 * max(o_ref.abs().max().item(), 1e-3)
    np.savez_compressed(os.path.join(HERE, "infer_mixvol_T40.npz"), c=c.numpy(), f0=f0.numpy(), uv=uv.numpy(),
                        mix=mix.numpy(), vol=vol.numpy(), noise_enc_p=noise["enc_p"].numpy(),
                        noise_rand_ini=noise["rand_ini"].numpy(), noise_sine=noise["sine"].numpy(), o=o_ref.numpy(),
                        meta=json.dumps(dict(seed=seed, T=T, S=S, noice_scale=0.4)))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    models, utils = import_reference()
    from oracle import weights as W
    snake = W.small_config()
    snake["vocoder_name"] = "nsf-snake-hifigan"
    run_case(models, "snake_T40", snake, B=2, T=40, seed=13)
    run_case(models, "tiny_T40", W.small_tiny_config(), B=2, T=40, seed=14)
    run_case(models, "tinyfull_T24", W.tiny_config(), B=2, T=24, seed=16)
    run_mixvol(models)


if __name__ == "__main__":
    main()
