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


@pytest.mark.parametrize("graph", [False, True])
def test_nonfinite_losses_are_caught_without_a_sync_per_step(dev, graph):
    """TrainStep's device-side guard (svc_nonfinite_guard_f32, one thread per iteration, captured with it): finite iterations leave
    the counters clean; once a weight is poisoned the replayed iterations keep running without any host read-back, and the next
    periodic check raises with the device's own count of what it saw."""
    import train as T
    cs = load_case()
    hps = _hps(cs, 2e-4)
    net_g, net_d, optim_g, optim_d = T.build(hps, dev)
    net_g.module.load_state_dict(cs["sd_g"], strict=True)
    net_d.module.load_state_dict(cs["sd_d"], strict=True)
    net_g.train()
    net_d.train()
    step = T.TrainStep(hps, net_g, net_d, optim_g, optim_d).enable_graph(graph)
    step.finite_every = 1000                      # no periodic read-back inside this test: only the forced ones below
    c, f0, uv, spec, y, sid, lengths = [t.to(dev) for t in cs["batch"]]
    noise = {k: v.to(dev) for k, v in cs["noise"].items()}
    items = (c, f0, spec, y, sid, lengths, uv, None)
    for _ in range(3):
        step(items, noise=noise)
    step.check_finite(force=True)                 # clean so far (warm-up and capture launches included)
    n_before = int(step._guard[2])
    with torch.no_grad():      # (the prior encoder's projection: its NaN reaches loss_kl directly; a NaN in the waveform would be
        #                         swallowed by the mel's log(max(x, 1e-5)) clamp — fmax returns the non-NaN operand)
        net_g.module.enc_p.proj.bias.fill_(float("nan"))
        torch.autograd.graph.increment_version(net_g.module.enc_p.proj.bias)
    out = step(items, noise=noise)                # runs to completion: nothing synchronises on the losses
    step(items, noise=noise)
    assert not all(bool(torch.isfinite(v)) for v in out.values() if torch.is_tensor(v))
    with pytest.raises(FloatingPointError) as e:
        step.check_finite(force=True)
    assert "non-finite" in str(e.value)
    assert int(step._guard[2]) == n_before + 2 and int(step._guard[0]) == 0     # counted 2 more launches; cleared after raising


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


def test_training_graph_replays_back_to_back_equal_eager(dev, monkeypatch):
    """replays of the whole-iteration hipGraph enqueued back to back — NO host wait between them (the
    default since round 3, see train.TrainStep._serialize_replays), a host synchronisation in the middle of the sequence (the
    pattern that gave bimodal losses in round 2), different noise every iteration — must reproduce the same number of EAGER
    iterations: losses of every iteration and the final parameters."""
    import train as T
    cs = load_case()
    hps = _hps(cs, 2e-4)
    c, f0, uv, spec, y, sid, lengths = [t.to(dev) for t in cs["batch"]]
    items = (c, f0, spec, y, sid, lengths, uv, None)
    n_it = 6
    noises = []
    for i in range(n_it):
        g = torch.Generator().manual_seed(500 + i)
        noises.append({k: (v + 0.05 * torch.randn(v.shape, generator=g)).to(dev) if v.is_floating_point() else v.to(dev)
                       for k, v in cs["noise"].items()})

    def run(graph):
        net_g, net_d, og, od = T.build(hps, dev)
        net_g.module.load_state_dict(cs["sd_g"], strict=True)
        net_d.module.load_state_dict(cs["sd_d"], strict=True)
        net_g.train()
        net_d.train()
        step = T.TrainStep(hps, net_g, net_d, og, od).enable_graph(graph)
        if graph:
            snaps = (og.snapshot(), od.snapshot())
            step(items, noise=noises[0])                   # warm-up + capture + first replay ...
            torch.cuda.synchronize()
            og.restore(snaps[0])                           # ... undone (parameters, moments, device step counters): start over
            od.restore(snaps[1])
        outs = []
        for i in range(n_it):
            outs.append(step(items, noise=noises[i]))      # device scalars: nothing here waits for the GPU
            if i == 2:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        flat = torch.cat([net_g.module.state_dict()[k].flatten().float() for k in sorted(cs["sd_g"]) if cs["sd_g"][k].is_floating_point()])
        return [{k: float(v) for k, v in o.items() if torch.is_tensor(v)} for o in outs], flat.cpu()

    monkeypatch.setenv("SVC_TRAIN_SERIALIZE", "0")        # the host wait between replays is the default; this test pins the path without it
    le, pe = run(False)
    lg, pg = run(True)
    for i, (a, b) in enumerate(zip(le, lg)):
        for k in LOSS_KEYS:
            assert abs(a[k] - b[k]) <= 2e-3 * max(1.0, abs(a[k])), (i, k, a[k], b[k])
    d = (pe - pg).abs()
    # wgrad atomics + Adam's lr * sign(g) steps: isolated elements may differ by a couple of lr (see test_data_parallel_gpu)
    assert d.max().item() <= 2.5 * 2e-4 * n_it and d.mean().item() <= 0.02 * 2e-4, (d.max().item(), d.mean().item())
