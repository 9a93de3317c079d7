"""Data-parallel training over RCCL/xGMI.

Reference: `DDP(net_g, device_ids=[rank])`, `DDP(net_d, device_ids=[rank])` (train.py:57,89-90) — torch's C++ reducer
with 25 MB buckets of gathered gradient copies; and, because the reference builds its DataLoader without a
DistributedSampler, every rank trains on the SAME minibatch.  Both are re-designed here:

  * **One process per GPU, minibatch sharded across ranks** (`shard_indices` is the DistributedSampler the reference
    omits), the only exchange being one gradient all-reduce (mean) per optimizer per step.
  * **Buckets are contiguous slices of the optimizer's flat gradient arena** (optim.ParamArena): no gather/scatter
    copies, bucket boundaries chosen in bytes.  A bucket's all-reduce is issued from the autograd hook of the LAST of its
    parameters to receive a gradient, so it runs on RCCL's stream while the rest of backward is still computing;
    `backward()` returns with every bucket waited on (same contract as DDP: grads are averaged when backward returns).
    Buckets are laid out in REVERSE parameter order (the order backward produces gradients), and sized for xGMI:
    ring all-reduce is per-link bound (≈153 GB/s × 7 links per GPU), so few large buckets (default 64 MiB → 4 for the
    210 MB generator) beat many small ones; the first bucket is smaller (8 MiB) so that communication starts early.
  * The parameter broadcast at construction is ONE collective on the arena.
  * In the generator step the discriminator's parameters take no gradient at all (train.py's G step back-propagates
    through `net_d` into `y_hat`; the reference computes AND all-reduces the discriminator's weight gradients there
    only to zero them next iteration): `no_param_grads(net_d)` switches them off, skipping wgrad kernels and the
    wasted 187 MB all-reduce.

The same code runs over `gloo` on CPU tensors (tests/test_data_parallel_cpu.py, world_size 2).
"""
import contextlib
import os

import torch
import torch.distributed as dist
from torch import nn

from optim import arena_for


def shard_indices(n_items, rank, world, epoch=0, shuffle=True, seed=1234, drop_last=True):
    """Indices of this rank's shard of an epoch (DistributedSampler semantics: same permutation on every rank, rank r
    takes items r, r+world, ...)."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        order = torch.randperm(n_items, generator=g).tolist()
    else:
        order = list(range(n_items))
    if drop_last:
        order = order[:n_items - n_items % world]
    elif order:
        pad = (-len(order)) % world
        order = order + (order * (pad // len(order) + 1))[:pad]       # wrap around as often as needed (n_items < world)
    return order[rank::world]


def shard_batch(items, rank, world):
    """Slice every tensor of a global minibatch [B, ...] into this rank's B/world rows (B must divide)."""
    out = []
    for t in items:
        if t is None:                 # TextAudioCollate: volume_padded without vol_embedding (data_utils.py:186)
            out.append(None)
            continue
        if not torch.is_tensor(t):    # data_utils.SpecContextBatch
            n_all = t.ext.shape[0]
            if n_all % world:
                raise ValueError(f"global batch {n_all} does not divide over {world} ranks")
            k = n_all // world
            out.append(t.rows(rank * k, (rank + 1) * k))
            continue
        B = t.shape[0]
        if B % world:
            raise ValueError(f"global batch {B} does not divide over {world} ranks")
        n = B // world
        out.append(t[rank * n:(rank + 1) * n])
    return out


@contextlib.contextmanager
def no_param_grads(module):
    """Run a forward whose backward must flow to the INPUT only (the D pass of the generator step)."""
    ps = [p for p in module.parameters() if p.requires_grad]
    for p in ps:
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p in ps:
            p.requires_grad_(True)


class GradReducer:
    """Bucketed, backward-overlapped all-reduce (mean) over a ParamArena's flat gradient buffer."""

    def __init__(self, arena, process_group=None, bucket_bytes=64 << 20, first_bucket_bytes=8 << 20):
        self.arena = arena
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.backend = dist.get_backend(process_group)
        # world size 1 normally short-circuits every collective; SVC_DP_FORCE=1 issues them anyway (dry run of the RCCL calls)
        self.single = self.world == 1 and os.environ.get("SVC_DP_FORCE", "0") != "1"
        n = len(arena.params)
        # walk parameters in reverse (≈ gradient production order); a bucket is a contiguous [start,end) of the arena
        self.buckets = []           # dicts(start, end, members)
        self.bucket_of = [0] * n
        cur, limit = None, first_bucket_bytes
        for i in range(n - 1, -1, -1):
            s, e = arena.span(i)
            if cur is not None and (cur["end"] - s) * 4 > limit and cur["members"]:
                self.buckets.append(cur)
                cur, limit = None, bucket_bytes
            if cur is None:
                cur = dict(start=s, end=e, members=[])
            cur["start"] = s
            cur["members"].append(i)
            self.bucket_of[i] = len(self.buckets)
        if cur is not None:
            self.buckets.append(cur)
        self.enabled = True
        self._pending = None        # per-bucket count of members still waiting for a gradient in this backward
        self._works = []
        self._launched = None
        self.stats = dict(reduced_bytes=0, launches=0, backward_passes=0, reduce_all_calls=0, wait_all_calls=0)
        # exposed-communication timing is OPT-IN (bench.py / the GPU tests set `time_exposed`): a training run of hundreds of
        # thousands of iterations must not collect two hipEvents per wait that nobody drains
        self.time_exposed = os.environ.get("SVC_DP_TIME_EXPOSED", "0") == "1"
        self._exposed = []          # (start, end) event pairs around the waits on the compute stream, while time_exposed
        self._exposed_ms = 0.0      # pairs already folded into a running sum (the list is drained every _EXPOSED_KEEP pairs)
        self._seg_cb = None         # segmented(): callback(bucket) instead of a collective (hipGraph capture of a backward pass)
        self.seg_next = 0           # ... first bucket (index order) not yet handed to the callback
        arena.add_listener(self._on_grad)

    # -- construction-time parameter sync ----------------------------------------------------------------------
    def broadcast_parameters(self, src=0):
        dist.broadcast(self.arena.param, src=src, group=self.pg)
        torch.autograd.graph.increment_version(self.arena.params)

    # -- backward-time ---------------------------------------------------------------------------------------------
    def _begin(self):
        self._pending = [len(b["members"]) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._works = []
        self.stats["backward_passes"] += 1
        torch.autograd.Variable._execution_engine.queue_callback(self._finalize)

    def _on_grad(self, i):
        if self._seg_cb is not None:
            self._on_grad_segmented(i)
            return
        if not self.enabled or self.single:
            return
        if self._pending is None:
            self._begin()
        b = self.bucket_of[i]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        bk = self.buckets[b]
        buf = self.arena.grad[bk["start"]:bk["end"]]
        if self.backend == "nccl":
            w = dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)
        else:       # gloo has no AVG: sum, then scale when the work completes
            w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._works.append((w, buf))
        if self._launched is not None:
            self._launched[b] = True
        self.stats["reduced_bytes"] += buf.numel() * 4
        self.stats["launches"] += 1

    def _finalize(self):
        """End of this backward pass (autograd engine callback): reduce buckets that hold at least one fresh gradient
        but were not completed (parameters unused this pass — identical on every rank because every rank runs the
        same graph), then make the compute stream wait for every outstanding all-reduce."""
        if self._pending is None:
            return
        for b, bk in enumerate(self.buckets):
            if not self._launched[b] and self._pending[b] < len(bk["members"]):
                self._launch(b)
        for w, buf in self._works:
            w.wait()
            if self.backend != "nccl":
                buf.mul_(1.0 / self.world)
        self._works = []
        self._pending = None

    # -- bucket-ordered mode: the replayed-graph iteration (train.TrainStep._call_graph_dp) ---------------------------------
    # Every rank issues the SAME collectives in the SAME order on every iteration — bucket 0, 1, ..., n-1 of an optimizer —
    # whatever it does locally (replays split graphs, captures a new batch shape, or launches eagerly past the graph cap), so
    # ranks whose shards produce different padded shapes on an iteration stay in step.
    @contextlib.contextmanager
    def segmented(self, on_ready):
        """Run a backward pass (being captured into hipGraphs) WITHOUT collectives: `on_ready(b)` is called from the gradient
        hook that completes bucket b — strictly in bucket index order (a bucket that completes early waits for its
        predecessors), on the thread that runs backward.  The caller ends the current capture there, notes "all-reduce
        bucket b", and begins the next graph; buckets not completed when backward returns (`seg_next` .. n-1: parameters without
        a gradient in this pass) are reduced after the last graph."""
        self._seg_cb, self._pending, self.seg_next = on_ready, None, 0
        try:
            yield self
        finally:
            self._seg_cb, self._pending = None, None

    def _on_grad_segmented(self, i):
        if self._pending is None:
            self._pending = [len(b["members"]) for b in self.buckets]
        b = self.bucket_of[i]
        self._pending[b] -= 1
        while self.seg_next < len(self.buckets) and self._pending[self.seg_next] == 0:
            nb = self.seg_next
            self.seg_next += 1
            self._seg_cb(nb)
        if self._pending[b] == 0 and b >= self.seg_next and not self.__dict__.get("_stall_logged"):
            # bucket b is complete but an EARLIER bucket still waits for a gradient (a parameter without one in this pass): every
            # bucket from seg_next on is then reduced after the last graph — correct, identical on all ranks, but without overlap
            self.__dict__["_stall_logged"] = True
            import logging
            missing = [i2 for i2 in self.buckets[self.seg_next]["members"]]
            logging.getLogger("train").warning(
                "GradReducer: bucket %d completed before bucket %d (%d of its %d parameters still without a gradient): buckets are "
                "released in index order, so all-reduces from bucket %d on lose their overlap with this backward pass",
                b, self.seg_next, self._pending[self.seg_next], len(missing), self.seg_next)

    def launch_bucket(self, b):
        """Asynchronous all-reduce (mean) of bucket b, ordered after the work queued on the current stream so far (RCCL's
        stream waits for an event recorded here); the current stream is NOT blocked: what is enqueued next overlaps it."""
        if self.single:
            return
        self._launch(b)

    _EXPOSED_KEEP = 64

    def _timing(self, time_it):
        on = self.time_exposed if time_it is None else time_it
        return bool(on) and self.arena.grad.is_cuda

    def _note_exposed(self, ev):
        self._exposed.append(ev)
        if len(self._exposed) > self._EXPOSED_KEEP and ev[0].query():
            # bounded: fold the pairs that have completed into the running sum (no device synchronisation)
            keep = []
            for a, b in self._exposed:
                if b.query():
                    self._exposed_ms += a.elapsed_time(b)
                else:
                    keep.append((a, b))
            self._exposed = keep

    def wait_all(self, time_it=None):
        """The current stream waits for every all-reduce issued by launch_bucket since the last call.  The time the stream
        stalls here is the iteration's EXPOSED communication (`exposed_ms()`)."""
        if self.single:
            return
        self.stats["wait_all_calls"] += 1
        ev = None
        if self._timing(time_it):
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w, buf in self._works:
            w.wait()
            if self.backend != "nccl":
                buf.mul_(1.0 / self.world)
        self._works = []
        self._launched = None
        if ev is not None:
            ev[1].record()
            self._note_exposed(ev)

    def reduce_all(self, time_it=None):
        """All-reduce (mean) of the whole flat gradient buffer, ordered after the work already queued on the current stream:
        the reduction used BETWEEN hipGraph replays (train.TrainStep with a process group) — the per-bucket path above is
        driven by Python autograd hooks, which do not run when a captured backward is replayed.  The same contiguous buckets
        are issued back to back as asynchronous collectives (RCCL pipelines them on its stream), then the compute stream
        waits for all of them.  Nothing overlaps with compute here, so the time between the two events recorded on the
        compute stream IS the exposed communication time of the iteration (`exposed_ms()`); at 0.6 GB of gradients per
        iteration that is a few ms over xGMI — the eager launch path it replaces costs ~100 ms of host time."""
        if self.single:
            return
        self.stats["reduce_all_calls"] += 1
        ev = None
        if self._timing(time_it):
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        works = []
        for bk in self.buckets:
            buf = self.arena.grad[bk["start"]:bk["end"]]
            op = dist.ReduceOp.AVG if self.backend == "nccl" else dist.ReduceOp.SUM
            works.append((dist.all_reduce(buf, op=op, group=self.pg, async_op=True), buf))
            self.stats["reduced_bytes"] += buf.numel() * 4
            self.stats["launches"] += 1
        for w, buf in works:
            w.wait()
            if self.backend != "nccl":
                buf.mul_(1.0 / self.world)
        if ev is not None:
            ev[1].record()
            self._note_exposed(ev)

    def exposed_ms(self):
        """Total time the compute stream spent waiting for all-reduces since the last call (synchronises the device).  Only
        measured while `time_exposed` is set (bench.py, tests; SVC_DP_TIME_EXPOSED=1): 0.0 otherwise."""
        if self._exposed:
            torch.cuda.synchronize()
        total = self._exposed_ms + sum(a.elapsed_time(b) for a, b in self._exposed)
        self._exposed, self._exposed_ms = [], 0.0
        return total

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation without communication (DDP.no_sync)."""
        old, self.enabled = self.enabled, False
        try:
            yield
        finally:
            self.enabled = old


class DataParallel(nn.Module):
    """Drop-in for `torch.nn.parallel.DistributedDataParallel(module, device_ids=[rank])` (train.py:89-90): exposes
    `.module` (used as `net_g.module.infer`, train.py:297; `utils.save_checkpoint` strips it, utils.py:189-193), forwards
    calls, and keeps gradients synchronised through a GradReducer on the module's ParamArena."""

    def __init__(self, module, device_ids=None, process_group=None, bucket_bytes=64 << 20, first_bucket_bytes=8 << 20,
                 broadcast=True, **_ignored):
        super().__init__()
        self.module = module
        params = [p for p in module.parameters() if p.requires_grad]
        self.arena = arena_for(params)
        self.reducer = None
        # SVC_DP_FORCE=1: build the reducer at world size 1 too — a functional dry run of the multi-rank path (RCCL calls between
        # the graph replays, bucket events) on a single-GPU box
        force = os.environ.get("SVC_DP_FORCE", "0") == "1"
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(process_group) > 1 or force):
            self.reducer = GradReducer(self.arena, process_group, bucket_bytes, first_bucket_bytes)
            if broadcast:
                self.reducer.broadcast_parameters(0)
                # ... and the module's buffers, like DDP's constructor (`_sync_module_states`): the power-iteration vectors of
                # spectral-norm discriminators (`weight_u` / `weight_v`, models.py:170,205) start from rank 0's everywhere.  They
                # then evolve identically on every rank — each forward advances them from the (synchronised) weights by the same
                # deterministic kernel — so DDP's per-forward re-broadcast (`broadcast_buffers`) is not needed.
                for buf in module.buffers():
                    if torch.is_tensor(buf) and buf.numel():
                        dist.broadcast(buf, src=0, group=process_group)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def no_sync(self):
        return self.reducer.no_sync() if self.reducer is not None else contextlib.nullcontext()
