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
