"""Data-parallel TRAINING with the real HIP modules, world_size 2.  The box has one GPU and RCCL
refuses two ranks per device, so both ranks share cuda:0 and talk over gloo: a functional check of everything except
the transport — parameter broadcast on the flat arena, autograd-hook driven bucket all-reduce issued from the backward
thread, the frozen-discriminator generator step, FusedAdamW on the averaged gradients.  After two iterations on
DIFFERENT per-rank items both ranks must hold bit-identical generator and discriminator parameters."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, graph=False, backend="gloo", capture=False, env=None):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.update(env or {})
        if capture:
            os.environ["SVC_DP_CAPTURE_COLLECTIVES"] = "1"
        root = os.path.dirname(HERE)
        for p in (root, os.path.join(root, "so-vits-svc_amd"), HERE):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch.distributed as dist
        if backend == "nccl":                              # one rank per GPU over RCCL (needs >= `world` devices)
            dev = torch.device("cuda", rank)
            torch.cuda.set_device(dev)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dev = torch.device("cuda:0")
            torch.cuda.set_device(0)
        import train as T
        from test_train_loop_gpu import _hps
        from train_common import load_case
        cs = load_case()
        hps = _hps(cs, 2e-4)
        torch.manual_seed(100 + rank)                      # different init per rank: the broadcast must fix it
        net_g, net_d, og, od = T.build(hps, dev)
        if rank == 0:
            net_g.module.load_state_dict(cs["sd_g"])
            net_d.module.load_state_dict(cs["sd_d"])
        net_g.reducer.broadcast_parameters(0)
        net_d.reducer.broadcast_parameters(0)
        net_g.train()
        net_d.train()
        step = T.TrainStep(hps, net_g, net_d, og, od).enable_graph(bool(graph))
        net_g.reducer.time_exposed = net_d.reducer.time_exposed = True      # opt-in (bench.py does the same)
        c, f0, uv, spec, y, sid, lengths = [t.to(dev) for t in cs["batch"]]
        noise = {k: v.to(dev) for k, v in cs["noise"].items()}
        sl = slice(rank, rank + 1)                          # rank r trains on item r of the 2-item batch
        items = (c[sl], f0[sl], spec[sl], y[sl], sid[sl], lengths[sl], uv[sl], None)
        nz = {k: v[sl].contiguous() for k, v in noise.items()}
        for _ in range(2 if graph is False else 3):
            out = step(items, noise=nz)
        if graph:
            assert any(k[0] == "dp" for k in step._graphs), "the data-parallel hipGraph path was not taken"
        flat_dev = torch.cat([net_g.arena.param.detach(), net_d.arena.param.detach()])
        parts = [torch.empty_like(flat_dev) for _ in range(world)]
        dist.all_gather(parts, flat_dev)                    # (device tensors: valid for gloo and nccl alike)
        parts = [t.cpu() for t in parts]
        flat = flat_dev.cpu()
        same = all(torch.equal(parts[0], p) for p in parts)
        fin = all(torch.isfinite(v).all().item() for v in out.values() if torch.is_tensor(v))
        stats = (dict(net_g.reducer.stats, mode=getattr(step, "dp_mode", "eager"),
                      exposed_ms=net_g.reducer.exposed_ms() if graph else 0.0), dict(net_d.reducer.stats))
        dist.destroy_process_group()
        # (by value, as a numpy array: a torch tensor travels through the queue as a shared-memory handle that dies with this process)
        q.put((rank, "ok" if (same and fin) else f"same={same} finite={fin}", stats, flat.numpy() if rank == 0 else None))
    except Exception:      # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc(), None, None))


def _run_two_ranks(graph, backend="gloo", capture=False, world=2, env=None, timeout=600):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, graph, backend, capture, env)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=timeout) for _ in procs]
    except Exception:      # noqa: BLE001 — queue.Empty: a rank hangs (e.g. a mis-captured collective): never leave it on the GPU
        for p in procs:
            if p.is_alive():
                p.kill()
        raise
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    for rank, msg, stats, _ in res:
        assert msg == "ok", f"rank {rank}: {msg}"
    return [(r, m, st, torch.from_numpy(f) if f is not None and not torch.is_tensor(f) else f) for r, m, st, f in res]


def test_two_rank_training_keeps_parameters_identical():
    for rank, msg, stats, _ in _run_two_ranks(False):
        g, d = stats
        assert g["backward_passes"] == 2 and g["launches"] >= 2 and d["backward_passes"] == 2     # D: only the D steps


def test_two_rank_training_graph_segments():
    """train.TrainStep with a process group AND enable_graph(): each phase of the iteration is a sequence of hipGraphs cut at
    the gradient-bucket boundaries DURING the captured backward pass (real capture_end / capture_begin inside the reducer's
    hooks), a bucket's all-reduce issued behind the graph that completed it, AdamW after the last.  Ranks stay bit-identical,
    the un-captured warm-up leaves no trace (3 replayed iterations == 3 training steps: compared against 3 eager data-parallel
    iterations)."""
    res_g = _run_two_ranks(True)
    g_stats = res_g[0][2][0]
    assert g_stats["mode"].startswith("split graphs"), g_stats
    assert g_stats["backward_passes"] == 0 and g_stats["launches"] % 3 == 0, g_stats     # no collective outside the 3 replays
    flat_g = next(f for r, _, _, f in res_g if f is not None)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker3_eager, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg, _, _ in res:
        assert msg == "ok", f"rank {rank}: {msg}"
    flat_e = torch.as_tensor(next(f for r, _, _, f in res if f is not None))
    # Not bit-comparable: the weight-gradient kernels combine their time splits with fp32 atomics, and in its first steps
    # Adam moves every element by ~lr * sign(g) — an element whose gradient is ~0 can flip direction between two runs
    # (|difference| up to 2 * lr per step, isolated elements).  A skipped / doubled step would shift EVERY element by ~lr.
    d = (flat_g - flat_e).abs()
    lr, steps = 2e-4, 3
    stats = dict(max=d.max().item(), mean=d.mean().item(), frac_gt_1e5=(d > 1e-5).float().mean().item())
    out_dir = os.path.join(os.path.dirname(HERE), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "dp_graph_vs_eager.txt"), "w") as f:
            f.write(repr(stats) + "\n")
    assert stats["max"] <= 2.5 * lr * steps and stats["mean"] <= 0.02 * lr and stats["frac_gt_1e5"] <= 0.02, stats


def _worker3_eager(rank, world, port, q):
    _worker(rank, world, port, q, graph=None)


def _worker_diffusion(rank, world, port, q, graph=False):
    """BASELINE configs[4] (train_diff.py under data parallelism): each rank trains the shallow-diffusion model on its half
    of a 4-item batch; the averaged-gradient result must equal single-process training on the whole batch.  graph=True: the
    form `diffusion.solver.train` runs by default — graph[zero_grad, forward, backward] -> all-reduce(arena) -> AdamW."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        root = os.path.dirname(HERE)
        for p in (root, os.path.join(root, "so-vits-svc_amd"), HERE, os.path.join(HERE, "golden")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        torch.cuda.set_device(0)
        from data_parallel import DataParallel
        from diffusion import solver
        from diffusion.unit2mel import Unit2Mel
        from make_golden_diffusion_train import make_batches
        from oracle import diffusion_oracle as DO
        c = DO.small_cfg()
        batches = make_batches(c, 5, 4, 30, 2)

        def fresh(seed):
            net = Unit2Mel(c["input_channel"], c["n_spk"], c["use_pitch_aug"], c["out_dims"], c["n_layers"], c["n_chans"],
                           c["n_hidden"], c["timesteps"], c["k_step_max"])
            net.load_state_dict(DO.make_state_dict(c, seed), strict=False)
            return net.to(dev).train()

        def run(net_or_dp, sl):
            opt = solver.build_optimizer(net_or_dp.module if isinstance(net_or_dp, DataParallel) else net_or_dp, lr=2e-3)
            step = solver.TrainStep(net_or_dp, opt).enable_graph(graph and isinstance(net_or_dp, DataParallel))
            for bt in batches:
                d = {k: v[sl].to(dev) for k, v in bt.items()}
                step(dict(units=d["units"], f0=d["f0"], volume=d["volume"], spk_id=d["spk_id"], mel=d["gt"]),
                     noise=dict(t=d["t"], noise=d["noise"].contiguous()))
            mod = net_or_dp.module if isinstance(net_or_dp, DataParallel) else net_or_dp
            return torch.cat([p.detach().reshape(-1) for p in mod.parameters()]).cpu()

        net = fresh(7 + rank)                               # different init per rank: DataParallel's broadcast must fix it
        if rank == 0:
            net = fresh(7)
        dp = DataParallel(net)
        flat = run(dp, slice(2 * rank, 2 * rank + 2))
        if graph and dp.reducer.stats["reduce_all_calls"] != len(batches):
            raise AssertionError(f"graph mode did not reduce between replays: {dp.reducer.stats}")
        parts = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(parts, flat)
        same = torch.equal(parts[0], parts[1])
        msg = "ok" if same else "ranks diverged"
        dist.destroy_process_group()
        if rank == 0 and same:
            ref = run(fresh(7), slice(0, 4))
            err = (ref - flat).abs().max().item()
            if err > 2e-5:
                msg = f"data-parallel result differs from full-batch training by {err:.3e}"
        q.put((rank, msg, None))
    except Exception:      # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc(), None))


@pytest.mark.parametrize("graph", [False, True])
def test_two_rank_diffusion_training_equals_full_batch(graph):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_diffusion, args=(r, 2, port, q, graph)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg, _ in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_split_graph_iteration_over_rccl_world1():
    """What ONE GPU can exercise of the default N > 1 mode over the real transport: process group "nccl" (RCCL) at world size 1
    with SVC_DP_FORCE=1 — reducers built, hipGraph captures cut inside the backward pass under ProcessGroupNCCL's watchdog
    thread (capture_error_mode thread_local), every bucket's all-reduce really issued on RCCL's stream between the graph
    replays.  Three replayed iterations must equal three eager hook-driven iterations of the same process group to the weight
    gradients' atomics noise, and the two monolithic graphs of SVC_DP_SPLIT=0 likewise."""
    env = {"SVC_DP_FORCE": "1"}
    split = _run_two_ranks(True, backend="nccl", world=1, env=env)
    st = split[0][2][0]
    assert st["mode"].startswith("split graphs") and st["launches"] > 0, st
    eager = _run_two_ranks(None, backend="nccl", world=1, env=env)
    mono = _run_two_ranks(True, backend="nccl", world=1, env=dict(env, SVC_DP_SPLIT="0"))
    assert mono[0][2][0]["mode"].startswith("two graphs"), mono[0][2][0]
    lr, steps = 2e-4, 3
    for other in (eager, mono):
        d = (split[0][3] - other[0][3]).abs()
        assert d.max().item() <= 2.5 * lr * steps and d.mean().item() <= 0.02 * lr, (d.max().item(), d.mean().item())
    out_dir = os.path.join(os.path.dirname(HERE), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "dp_split_rccl_world1.txt"), "w") as f:
            f.write(repr(dict(split=split[0][2], mono=mono[0][2])) + "\n")


def test_captured_collectives_over_rccl_world1():
    """SVC_DP_CAPTURE_COLLECTIVES=1 on what ONE GPU can exercise of it: process group "nccl" at world size 1 with
    SVC_DP_FORCE=1 — the autograd hooks fire DURING the two captures, so every bucket's ncclAllReduce is recorded on RCCL's stream
    inside the hipGraphs and replayed with them; nothing is issued between the replays.  Three replayed iterations must equal the
    split-graph mode's (same kernels, same gradients) to the weight gradients' atomics noise.  The worker is killed if it does not
    answer (a mis-captured collective hangs instead of raising)."""
    env = {"SVC_DP_FORCE": "1"}
    cap = _run_two_ranks(True, backend="nccl", world=1, env=env, capture=True, timeout=300)
    st = cap[0][2][0]
    assert st["mode"].startswith("collectives captured"), st
    split = _run_two_ranks(True, backend="nccl", world=1, env=env)
    lr, steps = 2e-4, 3
    d = (split[0][3] - cap[0][3]).abs()
    assert d.max().item() <= 2.5 * lr * steps and d.mean().item() <= 0.02 * lr, (d.max().item(), d.mean().item())


needs_two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")


@needs_two_gpus
def test_two_rank_training_over_rccl():
    """the N > 1 path over the REAL transport (backend "nccl" = RCCL over xGMI), one rank per GPU: eager
    bucket-overlapped mode, the default two-graph mode, and the captured-collective mode (all-reduces inside the hipGraphs,
    overlapped with the replayed backward).  Every mode must leave both ranks bit-identical; the two graph modes run the same
    kernels on the same gradients and must agree with each other to fp32 atomics' noise.  Skipped on one-GPU boxes — it runs
    the moment the suite sees a multi-GPU node."""
    _run_two_ranks(False, backend="nccl")
    two = next(f for r, _, _, f in _run_two_ranks(True, backend="nccl") if f is not None)      # split graphs, overlapped buckets
    cap = next(f for r, _, _, f in _run_two_ranks(True, backend="nccl", capture=True) if f is not None)
    d = (two - cap).abs()
    assert d.max().item() <= 2.5 * 2e-4 * 3 and d.mean().item() <= 0.02 * 2e-4, (d.max().item(), d.mean().item())
