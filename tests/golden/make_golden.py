"""Generate the committed golden vectors by running the REAL reference (imported from /root/reference).

Runs only in the build container (the GPU box has no /root/reference).  It
  1. imports the unmodified reference modules (stubbing the absent faiss/librosa imports of utils.py:12-13),
  2. checks oracle.weights.param_shapes against the reference state_dict key-for-key / shape-for-shape,
  3. loads the deterministic synthetic checkpoint, runs SynthesizerTrn.infer with the RNG draws replaced by our
     explicit noise tensors (draw order: models.py:160, vdecoder/hifigan/models.py:147, :266, :319),
  4. checks the oracle restatement against the reference on the same inputs and
  5. writes small .npz fixtures (inputs, noise, reference outputs + intermediates) and the key list.

usage: python tests/golden/make_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def import_reference():
    for name in ("faiss", "librosa", "librosa.filters"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["librosa.filters"].mel = lambda **kw: None
    sys.modules["librosa"].filters = sys.modules["librosa.filters"]
    sys.path.insert(0, REF)
    import models  # noqa
    import utils  # noqa
    return models, utils


class NoiseInjector:
    """Replaces torch.randn_like / torch.rand with a queue of preset tensors (checked by shape)."""

    def __init__(self, queue):
        self.queue = list(queue)
        self.orig_randn_like = torch.randn_like
        self.orig_rand = torch.rand

    def __enter__(self):
        inj = self

        def randn_like(t, **kw):
            n = inj.queue.pop(0)
            if n is None:
                return inj.orig_randn_like(t, **kw)
            assert tuple(n.shape) == tuple(t.shape), (n.shape, t.shape)
            return n.clone()

        def rand(*size, **kw):
            n = inj.queue.pop(0)
            assert tuple(n.shape) == tuple(size), (n.shape, size)
            return n.clone()

        torch.randn_like = randn_like
        torch.rand = rand
        return self

    def __exit__(self, *a):
        torch.randn_like = self.orig_randn_like
        torch.rand = self.orig_rand


def build_ref_model(models, cfg, sd):
    kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
    net = models.SynthesizerTrn(cfg["spec_channels"], cfg["segment_size"], **kw)
    ref_sd = net.state_dict()
    missing = set(ref_sd) - set(sd)
    extra = set(sd) - set(ref_sd)
    assert not missing and not extra, (sorted(missing)[:5], sorted(extra)[:5])
    for k in ref_sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), (k, ref_sd[k].shape, sd[k].shape)
    net.load_state_dict(sd)
    net.eval()
    return net


def run_case(models, name, cfg, B, T, seed, predict_f0=False, noice_scale=0.4):
    from oracle import svc_oracle as O
    from oracle import weights as W
    sd = W.make_state_dict(cfg, seed)
    net = build_ref_model(models, cfg, sd)
    c, f0, uv, sid = W.make_inputs(cfg, B, T, seed)
    noise = W.make_noise(cfg, B, T, seed + 1)
    L = noise["sine"].shape[1]
    cap = {}
    hooks = [
        net.enc_p.register_forward_hook(lambda m, i, o: cap.__setitem__("z_p", o[0].detach().clone())),
        net.flow.register_forward_hook(lambda m, i, o: cap.__setitem__("z", o.detach().clone())),
        net.dec.m_source.register_forward_hook(
            lambda m, i, o: cap.__setitem__("har", o[0].detach().transpose(1, 2).clone())),
    ]
    queue = [noise["enc_p"], noise["rand_ini"], noise["sine"], None]
    with NoiseInjector(queue), torch.no_grad():
        o_ref, f0_ref = net.infer(c, f0, uv, g=sid, noice_scale=noice_scale, predict_f0=predict_f0)
    for h in hooks:
        h.remove()
    with torch.no_grad():
        out = O.synth_infer(sd, cfg, c, f0, uv, sid, noise, noice_scale=noice_scale, predict_f0=predict_f0,
                            return_all=True)
    rep = {}
    for k, ref in (("o", o_ref), ("z_p", cap["z_p"]), ("z", cap["z"]), ("har", cap["har"]), ("f0", f0_ref)):
        d = (out[k] - ref).abs().max().item()
        rep[k] = (d, ref.abs().max().item(), ref.pow(2).mean().sqrt().item())
    print(f"[{name}] oracle vs reference  (max|diff|, max|ref|, rms ref):")
    for k, v in rep.items():
        print(f"    {k:4s} {v[0]:.3e}  {v[1]:.3e}  {v[2]:.3e}")
    assert rep["o"][0] <= 2e-5 * max(rep["o"][1], 1e-3), "oracle does not reproduce the reference"
    np.savez_compressed(
        os.path.join(HERE, f"infer_{name}.npz"),
        c=c.numpy(), f0=f0.numpy(), uv=uv.numpy(), sid=sid.numpy(),
        noise_enc_p=noise["enc_p"].numpy(), noise_rand_ini=noise["rand_ini"].numpy(),
        noise_sine=noise["sine"].numpy().astype(np.float32),
        o=o_ref.numpy(), f0_out=f0_ref.numpy(), z_p=cap["z_p"].numpy(), z=cap["z"].numpy(), har=cap["har"].numpy(),
        meta=json.dumps(dict(cfg=name, B=B, T=T, seed=seed, predict_f0=predict_f0, noice_scale=noice_scale)))
    return net


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    models, utils = import_reference()
    from oracle import svc_oracle as O
    from oracle import weights as W

    full, small = W.full_config(), W.small_config()
    # key lists (751 keys for the full template)
    shapes = W.param_shapes(full)
    with open(os.path.join(HERE, "state_dict_keys_full.json"), "w") as f:
        json.dump({k: list(v) for k, v in shapes.items()}, f, indent=0, sort_keys=True)
    print("full template keys:", len(shapes), "params:", sum(int(np.prod(s)) for s in shapes.values()))

    # f0_to_coarse golden: dense sweep incl. the clamp edges (utils.py:69-80)
    f0 = torch.cat([torch.zeros(3), torch.linspace(1, 1500, 4000), torch.tensor([50., 1100., 1099.9, 49.9])])
    np.savez_compressed(os.path.join(HERE, "f0_to_coarse.npz"), f0=f0.numpy(), coarse=utils.f0_to_coarse(f0).numpy())
    assert torch.equal(utils.f0_to_coarse(f0), O.f0_to_coarse(f0))

    run_case(models, "small_T40", small, B=2, T=40, seed=11)
    run_case(models, "small_T40_predf0", small, B=1, T=40, seed=12, predict_f0=True)
    run_case(models, "full_T24", full, B=1, T=24, seed=1234)


if __name__ == "__main__":
    main()
