"""Training LOOP parity on MI355X: train.TrainStep (HIP forward/backward, flat-arena FusedAdamW, reference step order)
against the committed vectors of the REAL reference's loop (tests/golden/train_loop_small.npz: 3 iterations of
train.py:150-213 with torch.optim.AdamW) — losses 1e-3 relative at every iteration, sampled parameters 5e-5 absolute
after the last (AdamW moves each by ~lr per step = 6e-4 in total, so this resolves the update direction of every entry
whose gradient is above fp32 noise).  Plus FusedAdamW against torch.optim.AdamW on the GPU."""
import json
import os

import numpy as np
import pytest
import torch

from train_common import G, LOSS_KEYS, load_case

pytestmark = pytest.mark.gpu


def _hps(cs, lr):
    d = cs["data"]
    cfg = cs["cfg"]
    model = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
    return dict(data=dict(filter_length=d["n_fft"], hop_length=d["hop"], win_length=d["win"], n_mel_channels=d["n_mels"],
                          sampling_rate=d["sr"], mel_fmin=d["fmin"], mel_fmax=d["fmax"]),
                train=dict(segment_size=cfg["segment_size"] * d["hop"], learning_rate=lr, betas=[0.8, 0.99], eps=1e-9,
                           c_mel=45.0, c_kl=1.0, fp16_run=False),
                model=model)


@pytest.mark.parametrize("graph", [False, True])
def test_train_loop_matches_reference_loop(dev, graph):
    """graph=True: the whole iteration replayed from one hipGraph (TrainStep.enable_graph) must give the same losses and
    parameters — incl. the device-side AdamW step counter / bias correction advancing across replays."""
    import train as T
    cs = load_case()
    z = np.load(os.path.join(G, "train_loop_small.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    hps = _hps(cs, meta["lr"])
    net_g, net_d, optim_g, optim_d = T.build(hps, dev)
    net_g.module.load_state_dict(cs["sd_g"], strict=True)
    net_d.module.load_state_dict(cs["sd_d"], strict=True)
    optim_g.arena.check_views()                      # load_state_dict copies in place: the arena views survive
    net_g.train()
    net_d.train()
    step = T.TrainStep(hps, net_g, net_d, optim_g, optim_d).enable_graph(graph)
    c, f0, uv, spec, y, sid, lengths = [t.to(dev) for t in cs["batch"]]
    noise = {k: v.to(dev) for k, v in cs["noise"].items()}
    items = (c, f0, spec, y, sid, lengths, uv, None)
    for it in range(meta["n_iter"]):
        out = step(items, noise=noise)
        for k in LOSS_KEYS:
            ref = float(z[f"it{it}.{k}"])
            got = float(out[k])
            assert abs(got - ref) <= 1e-3 * max(1.0, abs(ref)), (it, k, got, ref)
    pg = dict(net_g.module.named_parameters())
    pd = dict(net_d.module.named_parameters())
    for name in z.files:
        if name.startswith("param_g."):
            assert np.abs(pg[name[8:]].detach().cpu().numpy() - z[name]).max() <= 5e-5, name
        if name.startswith("param_d."):
            assert np.abs(pd[name[8:]].detach().cpu().numpy() - z[name]).max() <= 5e-5, name
    # optimizer state round-trips through torch's state_dict format (utils.save_checkpoint / load_checkpoint)
    import copy
    sd = copy.deepcopy(optim_g.state_dict())
    assert len(sd["state"]) == len(optim_g.arena.params) and float(sd["state"][0]["step"]) == meta["n_iter"]
    m0 = optim_g.exp_avg.clone()
    optim_g.exp_avg.zero_()
    optim_g.load_state_dict(sd)
    assert torch.equal(optim_g.exp_avg, m0) and set(optim_g._steps) == {meta["n_iter"]}


def test_fused_adamw_matches_torch(dev):
    from optim import FusedAdamW
    torch.manual_seed(3)
    shapes = [(33, 7, 5), (129,), (1, 9, 96), (64, 1, 1)]
    ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    rs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt = FusedAdamW(ps, lr=1e-3, betas=(0.8, 0.99), eps=1e-9)
    ref = torch.optim.AdamW(rs, lr=1e-3, betas=(0.8, 0.99), eps=1e-9)
    for it in range(4):
        opt.zero_grad()
        ref.zero_grad()
        gs = [torch.randn(s, device=dev) * (10.0 ** (it - 2)) for s in shapes]
        skip = 1 if it == 2 else None            # one parameter without a gradient at iteration 2: must be left alone
        for i, (p, r, g) in enumerate(zip(ps, rs, gs)):
            if i == skip:
                continue
            (p * g).sum().backward()
            r.grad = g.clone()
        v0 = [p._version for p in ps]
        opt.step()
        ref.step()
        for i, (p, r) in enumerate(zip(ps, rs)):
            assert torch.allclose(p.detach(), r.detach(), rtol=2e-6, atol=2e-7), (it, i)
            if i != skip:
                assert p._version > v0[i]
    # lr schedulers drive it like any torch optimizer (train.py:111-114)
    sch = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.5)
    sch.step()
    assert abs(opt.param_groups[0]["lr"] - 5e-4) < 1e-12
