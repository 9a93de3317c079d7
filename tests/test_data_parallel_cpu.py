"""CPU suite, world_size 2 over gloo: the host logic of data-parallel training — ParamArena views,
arena-slice buckets, backward-overlapped all-reduce, the end-of-backward flush for unused parameters, no_sync,
parameter broadcast, minibatch sharding.  The reducer is device-agnostic host code (the same class drives RCCL on the
GPU); the modules here are plain torch CPU modules standing in for the HIP ones, which cannot run without a GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(24, 40)
        self.b = nn.Linear(40, 40)
        self.unused = nn.Linear(8, 8)          # never used in forward: exercises the end-of-backward flush
        self.c = nn.Linear(40, 3)

    def forward(self, x):
        return self.c(torch.tanh(self.b(torch.relu(self.a(x)))))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import data_parallel as DP
        from optim import ParamArena
        torch.manual_seed(100 + rank)                   # DIFFERENT init per rank: broadcast must fix it
        net = Net()
        torch.manual_seed(7)
        ref = Net()                                     # same on every rank: the full-batch single-process reference
        gX = torch.randn(8, 24)
        gY = torch.randn(8, 3)
        # rank 0's weights are the truth after broadcast
        src = [p.detach().clone() for p in net.parameters()]
        dp = DP.DataParallel(net, bucket_bytes=4 * 1024, first_bucket_bytes=1024)     # tiny buckets -> several
        assert dp.reducer is not None and len(dp.reducer.buckets) >= 3
        objs = [None] * world
        dist.all_gather_object(objs, [p.detach().clone() for p in net.parameters()])
        for a, b in zip(objs[0], objs[1]):
            assert torch.equal(a, b)
        if rank == 0:
            for a, b in zip(src, net.parameters()):
                assert torch.equal(a, b.detach())
        ref.load_state_dict(net.state_dict())
        # views into the arena
        arena = dp.arena
        assert isinstance(arena, ParamArena)
        for p, o in zip(arena.params, arena.offsets):
            assert p.data_ptr() == arena.param.data_ptr() + 4 * o
            assert p.grad.data_ptr() == arena.grad.data_ptr() + 4 * o

        # sharded minibatch: mean-loss over the local shard, all-reduce(mean) == full-batch gradient
        x, y = DP.shard_batch([gX, gY], rank, world)
        loss = ((dp(x) - y) ** 2).mean()
        loss.backward()
        ((ref(gX) - gY) ** 2).mean().backward()
        for (n, p), (_, r) in zip(net.named_parameters(), ref.named_parameters()):
            if n.startswith("unused"):
                assert p.grad.abs().max().item() == 0
                continue
            assert torch.allclose(p.grad, r.grad, rtol=1e-5, atol=1e-7), n
        st = dict(dp.reducer.stats)
        assert st["backward_passes"] == 1 and st["launches"] >= 3
        touched = [n for (n, _), t in zip(net.named_parameters(), arena.touched) if t]
        assert not any(n.startswith("unused") for n in touched)

        # second backward without zero_grad accumulates (avg(g1) + avg(g2)) like DDP
        loss = ((dp(x) - y) ** 2).mean()
        loss.backward()
        for (n, p), (_, r) in zip(net.named_parameters(), ref.named_parameters()):
            if not n.startswith("unused"):
                assert torch.allclose(p.grad, 2 * r.grad, rtol=1e-5, atol=1e-7), n

        # no_sync: local gradients only, no communication
        arena.zero_grad()
        before = dp.reducer.stats["launches"]
        with dp.no_sync():
            ((dp(x) - y) ** 2).mean().backward()
        assert dp.reducer.stats["launches"] == before
        local = Net()
        local.load_state_dict(net.state_dict())
        ((local(x) - y) ** 2).mean().backward()
        for (n, p), (_, r) in zip(net.named_parameters(), local.named_parameters()):
            if not n.startswith("unused"):
                assert torch.allclose(p.grad, r.grad, rtol=1e-5, atol=1e-7), n

        # ... followed by reduce_all(): every bucket of the flat buffer all-reduced back to back == the full-batch gradient
        # again (the path train.TrainStep takes between two hipGraph replays, where the per-bucket hooks do not run)
        dp.reducer.reduce_all()
        assert dp.reducer.stats["launches"] == before + len(dp.reducer.buckets)
        for (n, p), (_, r) in zip(net.named_parameters(), ref.named_parameters()):
            if not n.startswith("unused"):
                assert torch.allclose(p.grad, r.grad, rtol=1e-5, atol=1e-7), n

        # frozen-parameter pass (the D pass of the generator step): no gradients, no communication
        arena.zero_grad()
        before = dp.reducer.stats["launches"]
        xin = x.clone().requires_grad_(True)
        with DP.no_param_grads(net):
            out = net(xin).sum()
        out.backward()
        assert xin.grad is not None and dp.reducer.stats["launches"] == before
        assert all(p.requires_grad for p in net.parameters())
        assert not any(arena.touched)
        q.put((rank, "ok"))
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_gradreducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_shard_indices_partition():
    import data_parallel as DP
    n, world = 103, 4
    shards = [DP.shard_indices(n, r, world, epoch=3) for r in range(world)]
    assert all(len(s) == n // world for s in shards)
    flat = sorted(i for s in shards for i in s)
    assert len(set(flat)) == len(flat) and set(flat) <= set(range(n))
    assert shards != [DP.shard_indices(n, r, world, epoch=4) for r in range(world)]
    padded = [DP.shard_indices(n, r, world, drop_last=False, shuffle=False) for r in range(world)]
    assert all(len(s) == 26 for s in padded) and set(i for s in padded for i in s) == set(range(n))


def test_arena_touched_runs_and_foreign_grad():
    from optim import ParamArena
    ps = [nn.Parameter(torch.randn(5, 3)), nn.Parameter(torch.randn(70)), nn.Parameter(torch.randn(2))]
    vals = [p.detach().clone() for p in ps]
    a = ParamArena(ps)
    for p, v in zip(ps, vals):
        assert torch.equal(p.detach(), v)
    assert a.offsets == [0, 64, 192] and a.numel == 256
    # gather mode (no listener): autograd stores the produced gradients as they are; collect() moves them into the arena
    assert a.gather and all(p.grad is None for p in ps)
    (ps[0].sum() * 2 + ps[2].sum()).backward()
    assert a.touched == [True, False, True]
    assert a.touched_runs() == [(0, 64, None), (192, 256, None)]
    assert ps[0].grad.data_ptr() != a.grad.data_ptr() and ps[1].grad is None
    a.collect()
    assert torch.equal(a.grad[:15], torch.full((15,), 2.0)) and torch.equal(a.grad[192:194], torch.full((2,), 1.0))
    a.zero_grad()
    assert not any(a.touched) and all(p.grad is None for p in ps)
    # view mode (a listener, i.e. the data-parallel reducer, is attached): p.grad IS the arena slice
    seen = []
    (ps[0].sum() * 5).backward()
    a.add_listener(seen.append)
    assert not a.gather and ps[0].grad.data_ptr() == a.grad.data_ptr() and torch.equal(a.grad[:15], torch.full((15,), 5.0))
    assert a.grad[64:134].abs().max().item() == 0
    a.zero_grad()
    (ps[0].sum() * 2 + ps[2].sum()).backward()
    assert sorted(seen) == [0, 2] and torch.equal(a.grad[:15], torch.full((15,), 2.0))
    # a foreign .grad (module.zero_grad(set_to_none=True)) is moved back into the arena on the next accumulate
    ps[1].grad = None
    (ps[1] * 3).sum().backward()
    assert ps[1].grad.data_ptr() == a.grad.data_ptr() + 4 * 64 and torch.equal(a.grad[64:134], torch.full((70,), 3.0))
    a.zero_grad()
    assert not any(a.touched) and a.grad.abs().max().item() == 0


def test_param_arena_is_released_with_its_optimizer():
    """the id(param) -> arena registry kept every arena (flat param / grad / Adam buffers) alive for the life
    of the process and made a parameter un-registrable once its optimizer was gone."""
    import gc
    import weakref
    import optim
    lin = torch.nn.Linear(8, 8)
    a1 = optim.ParamArena(lin.parameters())
    assert optim.arena_for(lin.parameters()) is a1
    with pytest.raises(Exception):
        optim.ParamArena(lin.parameters())              # still owned by a live arena
    w = lin.weight.detach().clone()
    ref = weakref.ref(a1)
    a1.release()
    assert torch.equal(lin.weight, w) and lin.weight.data_ptr() != a1.param.data_ptr()
    del a1
    gc.collect()
    assert ref() is None                                 # nothing else holds the flat buffers
    a2 = optim.ParamArena(lin.parameters())              # the parameters can join a new arena
    ref2 = weakref.ref(a2)
    del a2                                               # dropped WITHOUT release(): must not stay reachable either
    gc.collect()
    assert ref2() is None
    a3 = optim.ParamArena(lin.parameters())
    assert optim.arena_for(lin.parameters()) is a3


# ---- world size 4: the control flow of train.TrainStep._call_graph_dp (two graphs per iteration, all-reduces between them) -------
class _FakeGraph:
    """Stands in for torch.cuda.CUDAGraph on CPU: 'capture' remembers the segment that ran inside it, replay() re-runs it on the
    same (static) input tensors — which is what a replayed hipGraph does."""
    capturing = None

    def __init__(self):
        self.fn = None

    def pool(self):
        return "pool"

    def capture_begin(self, *pool, capture_error_mode="global"):
        assert _FakeGraph.capturing is None, "nested capture"
        _FakeGraph.capturing = self

    def capture_end(self):
        assert _FakeGraph.capturing is self
        _FakeGraph.capturing = None

    def replay(self):
        if self.fn is not None:          # a graph begun inside a gradient hook holds the REST of a backward pass: the fake's first
            self.fn()                    # graph re-runs the whole segment, the later ones have nothing left to do


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return lambda *a, **k: None


def _dp4_worker(rank, world, port, q, scenario="same_shapes"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import contextlib
        import data_parallel as DP
        import svc_hip as S
        import train as TR
        from optim import arena_for

        # -- device calls stubbed: streams / events / graphs are no-ops or _FakeGraph; the optimizers are plain SGD over the SAME
        #    ParamArena class the fused AdamW uses (touched / collect / zero_grad bookkeeping is the real one)
        torch.cuda.Stream = _Dummy
        torch.cuda.Event = _Dummy
        torch.cuda.current_stream = lambda *a, **k: _Dummy()
        torch.cuda.stream = lambda s: contextlib.nullcontext()
        torch.cuda.synchronize = lambda *a, **k: None
        torch.cuda.CUDAGraph = _FakeGraph

        @contextlib.contextmanager
        def fake_capture(graph, pool=None):
            _FakeGraph.capturing = graph
            try:
                yield
            finally:
                _FakeGraph.capturing = None
        S.graph_capture = fake_capture
        S.capture_stream = contextlib.nullcontext
        S.capture_error_mode = lambda: "global"

        class SGD:
            def __init__(self, params, lr=0.05):
                self.arena, self.lr, self.steps = arena_for(list(params)), lr, 0

            def zero_grad(self):
                self.arena.zero_grad()

            def snapshot(self):
                return dict(param=self.arena.param.clone(), steps=self.steps)

            def restore(self, snap, device=True):
                self.steps = snap["steps"]
                if device:
                    with torch.no_grad():
                        self.arena.param.copy_(snap["param"])

            def step(self):
                a = self.arena
                a.collect()
                with torch.no_grad():
                    for s, e, _ in a.touched_runs():
                        a.param[s:e] -= self.lr * a.grad[s:e]
                self.steps += 1

        class Step(TR.TrainStep):
            log = []

            def _dense_spec(self, items):          # (the real one densifies the loader's spectrogram slot)
                return items

            def _replayed(self, seg, *a):          # a replayed hipGraph runs the kernels only: no Python autograd hooks fire
                with self.net_g.reducer.no_sync(), self.net_d.reducer.no_sync():
                    return seg(*a)

            def _seg_d(self, items, noise=None):
                if _FakeGraph.capturing is not None:
                    _FakeGraph.capturing.fn = lambda: self._replayed(self._seg_d, items, noise)
                x, y = items[:2]
                y_hat = self.net_g(x)
                loss_d = ((self.net_d(y) - 1) ** 2).mean() + (self.net_d(y_hat.detach()) ** 2).mean()
                self.optim_d.zero_grad()
                loss_d.backward()
                self.ctx = dict(y_hat=y_hat, loss_disc=loss_d.detach())
                Step.log.append("D")
                return self.ctx

            def _seg_g(self, ctx):
                if _FakeGraph.capturing is not None:
                    _FakeGraph.capturing.fn = lambda: self._replayed(self._seg_g, self.ctx)
                dmod = self.net_d.module
                with DP.no_param_grads(dmod):
                    loss_g = ((dmod(self.ctx["y_hat"]) - 1) ** 2).mean()
                self.optim_g.zero_grad()
                loss_g.backward()
                Step.log.append("G")
                return dict(loss_disc=self.ctx["loss_disc"], loss_gen=loss_g.detach())

        torch.manual_seed(50 + rank)                      # different init per rank: the constructor's broadcast fixes it
        G = nn.Sequential(nn.Linear(6, 16), nn.Tanh(), nn.Linear(16, 5))
        D = nn.Sequential(nn.Linear(5, 12), nn.Tanh(), nn.Linear(12, 1))
        D.register_buffer("weight_u", torch.randn(12))    # a spectral-norm style buffer: rank 0's must reach everyone
        og, od = SGD(G.parameters()), SGD(D.parameters())
        net_g = DP.DataParallel(G, bucket_bytes=256, first_bucket_bytes=128)
        net_d = DP.DataParallel(D, bucket_bytes=256, first_bucket_bytes=128)
        assert len(net_g.reducer.buckets) >= 2 and net_g.reducer.world == 4
        bufs = [None] * world
        dist.all_gather_object(bufs, D.weight_u.clone())
        assert all(torch.equal(bufs[0], b) for b in bufs), "buffers were not broadcast from rank 0"
        hps = dict(data=dict(filter_length=8, n_mel_channels=2, sampling_rate=8, hop_length=2, win_length=8, mel_fmin=0, mel_fmax=4),
                   train=dict(segment_size=4, c_mel=1, c_kl=1, fp16_run=False))
        step = Step(hps, net_g, net_d, og, od).enable_graph(True)
        p0 = [p.detach().clone() for p in list(G.parameters()) + list(D.parameters())]
        gen = torch.Generator().manual_seed(3)
        if scenario == "same_shapes":
            # uneven shards: rank r holds r + 1 items (every rank captures its own shapes; only the arena is communicated)
            shards = [(torch.randn(r + 1, 6, generator=gen), torch.randn(r + 1, 5, generator=gen)) for r in range(world)]
            n_it = 3
            sched = [[shards[r]] * n_it for r in range(world)]
            for it in range(n_it):
                out = step(sched[rank][it])
            assert step.dp_mode.startswith("split graphs") and len(step._graphs) == 1
            # warm-up (2 eager iterations, undone) + capture + 3 replayed iterations, D segment always before G
            assert Step.log == ["D", "G"] * (2 + 1 + n_it), Step.log
            for red in (net_g.reducer, net_d.reducer):
                # no collective in the warm-up or the capture: exactly one all-reduce per bucket and replayed iteration, none through
                # the autograd hooks
                assert red.stats["reduce_all_calls"] == 0 and red.stats["backward_passes"] == 0, red.stats
                assert red.stats["launches"] == n_it * len(red.buckets), red.stats
            prog_d, prog_g = next(iter(step._graphs.values()))[:2]
            for prog, red in ((prog_d, net_d.reducer), (prog_g, net_g.reducer)):
                assert [x for o, x in prog if o == "reduce"] == list(range(len(red.buckets)))      # canonical order on every rank
                assert prog[0][0] == "graph" and sum(1 for o, _ in prog if o == "graph") >= 2        # the backward pass WAS cut
                first_reduce = [o for o, _ in prog].index("reduce")
                assert "graph" in [o for o, _ in prog[first_reduce:]], "no graph left to overlap the first all-reduce with"
        else:
            # every rank meets ITS batch shapes in ITS order (rank-local shards pad to different buckets), with a
            # graph cap of 2 — so on the same iteration one rank replays, another warms up + captures, a third is past the cap and
            # launches eagerly (rank 1: eviction off) or evicts (the others).  The collective sequence must still line up.
            os.environ["SVC_TRAIN_GRAPH_EVICT"] = "0" if rank == 1 else "1"
            step.max_graphs = 2
            sizes = {0: [2, 2, 3, 2, 4, 3], 1: [1, 2, 3, 4, 1, 2], 2: [3, 3, 3, 3, 3, 3], 3: [4, 1, 4, 2, 2, 1]}
            n_it = len(sizes[0])
            batches = {n: (torch.randn(n, 6, generator=gen), torch.randn(n, 5, generator=gen)) for n in (1, 2, 3, 4)}
            sched = [[batches[n] for n in sizes[r]] for r in range(world)]
            for it in range(n_it):
                out = step(sched[rank][it])
            for red in (net_g.reducer, net_d.reducer):
                assert red.stats["backward_passes"] == 0, red.stats            # the hook-driven path never ran (warm-ups are no_sync)
                assert red.stats["launches"] == n_it * len(red.buckets), red.stats
            if rank == 1:
                assert step.eager_fallbacks == 2 and step.graph_evictions == 0 and len(step._graphs) == 2
            elif rank == 2:
                assert step.eager_fallbacks == 0 and step.graph_evictions == 0 and len(step._graphs) == 1
            else:
                assert step.eager_fallbacks == 0 and step.graph_evictions >= 1 and len(step._graphs) == 2
        # every rank ends with the same parameters ...
        flat = torch.cat([p.detach().flatten() for p in list(G.parameters()) + list(D.parameters())])
        allp = [None] * world
        dist.all_gather_object(allp, flat)
        assert all(torch.equal(allp[0], a) for a in allp)
        # ... the ones a single process gets by averaging the four ranks' gradients by hand
        Gr = nn.Sequential(nn.Linear(6, 16), nn.Tanh(), nn.Linear(16, 5))
        Dr = nn.Sequential(nn.Linear(5, 12), nn.Tanh(), nn.Linear(12, 1))
        with torch.no_grad():
            for p, v in zip(list(Gr.parameters()) + list(Dr.parameters()), p0):
                p.copy_(v)
        for it in range(n_it):
            gd = [torch.zeros_like(p) for p in Dr.parameters()]
            for r in range(world):
                x, y = sched[r][it]
                loss = ((Dr(y) - 1) ** 2).mean() + (Dr(Gr(x).detach()) ** 2).mean()
                for a, g_ in zip(gd, torch.autograd.grad(loss, list(Dr.parameters()))):
                    a += g_ / world
            with torch.no_grad():
                for p, g_ in zip(Dr.parameters(), gd):
                    p -= 0.05 * g_
            gg = [torch.zeros_like(p) for p in Gr.parameters()]
            for r in range(world):
                x, y = sched[r][it]
                loss = ((Dr(Gr(x)) - 1) ** 2).mean()
                for a, g_ in zip(gg, torch.autograd.grad(loss, list(Gr.parameters()))):
                    a += g_ / world
            with torch.no_grad():
                for p, g_ in zip(Gr.parameters(), gg):
                    p -= 0.05 * g_
        ref = torch.cat([p.detach().flatten() for p in list(Gr.parameters()) + list(Dr.parameters())])
        assert torch.allclose(flat, ref, rtol=1e-5, atol=1e-6), (flat - ref).abs().max().item()
        assert torch.isfinite(out["loss_gen"]) and not S.wgrad_slab.active
        q.put((rank, "ok"))
    except Exception:      # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["same_shapes", "rank_local_shapes"])
def test_split_graph_data_parallel_iteration_world4_gloo(scenario):
    """train.TrainStep._call_graph_dp at world size 4 with the device calls stubbed: warm-up
    under no_sync, each phase captured as a SEQUENCE of graphs cut at the gradient-bucket boundaries by the reducer's hooks, then
    per iteration graph -> all-reduce(bucket 0) ‖ graph -> ... -> wait -> step for D and for G; buffers broadcast from rank 0; all
    ranks bit-identical and equal to hand-averaged single-process training.  `rank_local_shapes`: the ranks meet different batch
    shapes on the same iteration (replay / warm-up + capture / eviction / eager fallback mixed) and still issue the same
    collective sequence."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp4_worker, args=(r, 4, port, q, scenario)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
