"""Golden vector for the stand-alone NSF-HiFiGAN vocoder from the REAL reference module (build container only).
usage: python tests/golden/make_golden_nsf_hifigan.py"""
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
    from oracle import nsf_hifigan_oracle as NO
    sys.path.insert(0, "/root/reference")
    import vdecoder.nsf_hifigan.models as R
    from vdecoder.nsf_hifigan.env import AttrDict
    h = NO.small_h()
    seed, B, T = 23, 2, 24
    sd = NO.make_state_dict(h, seed)
    net = R.Generator(AttrDict(dict(h)))
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v) for k, v in NO.param_shapes(h).items()}
    net.load_state_dict(sd)
    net.eval()
    g = torch.Generator().manual_seed(seed)
    mel = torch.randn(B, h["num_mels"], T, generator=g)
    f0 = 100 + 300 * torch.rand(B, T, generator=g)
    f0[:, 5:9] = 0
    f0[1, 15:] = 880.0
    upp = int(np.prod(h["upsample_rates"]))
    rand_ini = torch.rand(B, 9, generator=g)
    noise = torch.randn(B, T * upp, 9, generator=g)
    orig_rand, orig_randn_like = torch.rand, torch.randn_like
    torch.rand = lambda *a, **k: rand_ini.clone()
    torch.randn_like = lambda t, **k: noise.clone()
    try:
        with torch.no_grad():
            y_ref = net(mel, f0)
    finally:
        torch.rand, torch.randn_like = orig_rand, orig_randn_like
    with torch.no_grad():
        y = NO.generator(sd, h, mel, f0, rand_ini, noise)
    d = (y - y_ref).abs().max().item()
    print(f"oracle vs reference: max|diff| {d:.3e}, max|ref| {y_ref.abs().max().item():.3e}, rms {y_ref.pow(2).mean().sqrt().item():.3e}")
    assert d <= 2e-5 * max(y_ref.abs().max().item(), 1e-3)
    np.savez_compressed(os.path.join(HERE, "nsf_hifigan_small.npz"), mel=mel.numpy(), f0=f0.numpy(), rand_ini=rand_ini.numpy(),
                        noise=noise.numpy(), y=y_ref.numpy(), meta=json.dumps(dict(seed=seed, B=B, T=T)))
    print("wrote nsf_hifigan_small.npz")


if __name__ == "__main__":
    main()
