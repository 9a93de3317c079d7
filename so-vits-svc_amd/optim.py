"""Flat-arena parameter storage + fused AdamW.

Reference: `torch.optim.AdamW(net.parameters(), lr, betas=(0.8, 0.99), eps=1e-9)` for G and D (train.py:79-88), stepped
through a GradScaler (train.py:191-213), with `ExponentialLR` per epoch (train.py:111-114) and the optimizer state saved
in every checkpoint (utils.py:189-199).  The reference runs a Python loop over 751 (G) / ~180 (D) tensors per step.

MI355X design: 99 M fp32 parameters are 0.4 GB — nothing next to 288 GB of HBM — so all parameters of one optimizer live
in ONE contiguous arena (`ParamArena`): `p.data` of every parameter is a view into the flat `param` buffer, gradients
live in the flat `grad` buffer (as `p.grad` views when a data-parallel reducer is attached, else gathered there by one
multi-tensor copy per step), Adam moments are two more flat buffers.  Consequences:
  * the optimizer step is ONE `svc_adamw_f32` launch over the arena (HBM-bound: 5 reads + 3 writes of 4 B / element),
  * `zero_grad` is one memset (view mode) or free (gather mode: `p.grad = None`),
  * the data-parallel gradient all-reduce (data_parallel.py) works on contiguous slices of `grad` — buckets need no
    gather/scatter copies and the initial parameter broadcast is one collective.
`FusedAdamW` is a `torch.optim.Optimizer` (param_groups / state_dict / lr schedulers keep working unchanged), with
torch.optim.AdamW's exact update rule; parameters that received no gradient since the last `zero_grad` are skipped,
like torch does for `p.grad is None`.
"""
import weakref

import torch

import svc_hip as S

_ALIGN = 64          # elements: every parameter starts on a 256-byte boundary (coalesced float4 access, RCCL alignment)
_ARENA_ATTR = "_svc_arena"   # on each parameter: (weakref to its arena, index).  Weak: an arena (0.8 GB of flat buffers for G)
                             # lives exactly as long as its optimizer / reducer hold it, and a parameter whose arena is gone
                             # can join a new one (rebuilt optimizers in one process: test suites, notebooks, Svc reloads)


def _arena_of(p):
    ent = getattr(p, _ARENA_ATTR, None)
    if ent is None:
        return None
    arena = ent[0]()
    return None if arena is None or arena.released else (arena, ent[1])


class ParamArena:
    """Contiguous fp32 storage for a list of parameters (all on one device)."""

    def __init__(self, params):
        params = list(params)
        if not params:
            raise ValueError("ParamArena needs at least one parameter")
        dev = params[0].device
        for p in params:
            if p.device != dev or p.dtype != torch.float32:
                raise S.SvcError("ParamArena: all parameters must be fp32 on one device "
                                 f"(got {p.dtype} on {p.device} vs {dev})")
            if _arena_of(p) is not None:
                raise S.SvcError("ParamArena: parameter already belongs to another (live) arena; release() it first")
        self.params = params
        self.device = dev
        self.released = False
        self.offsets, off = [], 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        self.param = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        self.touched = [False] * len(params)          # gradient accumulated since the last zero_grad
        self._listeners = []                          # callables(index) fired from the post-accumulate hook
        self._hooks = []
        # Two ways for gradients to reach `grad`:
        #   gather (default, single process): `p.grad` is None before backward, autograd stores each produced gradient
        #     tensor as is (no per-parameter accumulate kernel: ~1100 launches per training iteration) and step() moves
        #     them into the arena with ONE multi-tensor copy;
        #   views (a listener is attached: the data-parallel reducer needs each gradient in its bucket slice the moment
        #     it is produced): `p.grad` IS the arena slice and autograd accumulates into it in place.
        self.gather = True
        self._gviews = []
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(params, self.offsets)):
                view = self.param[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                self._gviews.append(self.grad[o:o + p.numel()].view(p.shape))
                if p.grad is not None:
                    self.touched[i] = True
                setattr(p, _ARENA_ATTR, (weakref.ref(self), i))
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    def _make_hook(self, i):
        ref = weakref.ref(self)          # the hook must not keep the arena alive through its parameters

        def hook(p):
            arena = ref()
            if arena is not None and not arena.released:
                arena._on_grad(i, p)
        return hook

    def release(self):
        """Detach from the parameters: hooks removed, every parameter gets its own storage back (a copy of its current
        value) and may join another arena.  The flat buffers are freed once the last reference to the arena goes."""
        if self.released:
            return
        self.released = True
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._listeners = []
        with torch.no_grad():
            for p in self.params:
                p.data = p.data.clone()
                if p.grad is not None:
                    p.grad = None
                if hasattr(p, _ARENA_ATTR):
                    delattr(p, _ARENA_ATTR)

    def _on_grad(self, i, p):
        self.touched[i] = True
        if self.gather:
            return
        o = self.offsets[i]
        if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
            # someone replaced .grad (e.g. zero_grad(set_to_none=True) on the module): move it back into the arena
            gview = self._gviews[i]
            if p.grad is not None:
                gview.copy_(p.grad)
            p.grad = gview
        for fn in self._listeners:
            fn(i)

    def add_listener(self, fn):
        """Attach a per-gradient callback (the data-parallel reducer): switches the arena to view mode."""
        self._listeners.append(fn)
        if self.gather:
            self.gather = False
            with torch.no_grad():
                for i, p in enumerate(self.params):
                    gview = self._gviews[i]
                    if p.grad is not None and p.grad.data_ptr() != gview.data_ptr():
                        gview.copy_(p.grad)
                    elif p.grad is None:
                        gview.zero_()
                    p.grad = gview

    def collect(self):
        """gather mode: move the gradients autograd produced into the arena — one multi-tensor copy."""
        if not self.gather:
            return
        dst, src = [], []
        for i, p in enumerate(self.params):
            if self.touched[i] and p.grad is not None and p.grad.data_ptr() != self._gviews[i].data_ptr():
                dst.append(self._gviews[i])
                src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)

    def zero_grad(self):
        self.touched = [False] * len(self.params)
        if self.gather:
            for p in self.params:
                p.grad = None
            return
        self.grad.zero_()
        for i, p in enumerate(self.params):
            if p.grad is None or p.grad.data_ptr() != self._gviews[i].data_ptr():
                p.grad = self._gviews[i]

    def span(self, i):
        """(start, end) element range of parameter i including its alignment padding."""
        end = self.offsets[i + 1] if i + 1 < len(self.params) else self.numel
        return self.offsets[i], end

    def touched_runs(self, keys=None):
        """Maximal runs (start, end, key) (elements) of consecutive parameters that received a gradient (and share
        keys[i], when given)."""
        runs, cur = [], None
        for i, t in enumerate(self.touched):
            if t:
                s, e = self.span(i)
                k = keys[i] if keys is not None else None
                if cur is not None and cur[1] == s and cur[2] == k:
                    cur[1] = e
                else:
                    cur = [s, e, k]
                    runs.append(cur)
            else:
                cur = None
        return [tuple(r) for r in runs]

    def check_views(self):
        """Re-attach parameters whose storage was moved away (module.to()/.half() after construction is an error)."""
        for p, o in zip(self.params, self.offsets):
            if p.data_ptr() != self.param.data_ptr() + 4 * o:
                raise S.SvcError("ParamArena: a parameter's storage was replaced after the arena was built "
                                 "(move the module to its device/dtype BEFORE constructing the optimizer)")


def arena_for(params):
    """The arena holding exactly `params` (created on first use)."""
    params = list(params)
    hit = _arena_of(params[0])
    if hit is not None:
        arena = hit[0]
        if len(arena.params) == len(params) and all(a is b for a, b in zip(arena.params, params)):
            return arena
        raise S.SvcError("parameters already belong to a different arena")
    return ParamArena(params)


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW(params, lr, betas, eps, weight_decay=0.01) semantics, one HIP launch per step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        params = list(params)
        if params and isinstance(params[0], dict):
            raise S.SvcError("FusedAdamW takes one flat parameter list (the reference uses a single group per net)")
        params = [p for p in params if p.requires_grad]
        for p in params:
            if not p.is_cuda:
                raise S.SvcError("FusedAdamW needs CUDA/ROCm parameters: the MI355X engine has no CPU fallback")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.arena = arena_for(params)
        n = self.arena.numel
        self.exp_avg = torch.zeros(n, device=self.arena.device, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=self.arena.device, dtype=torch.float32)
        self._steps = [0] * len(params)     # per-parameter step counts (torch semantics: a skipped parameter lags)
        self._gstep = 0                     # number of step() calls that updated something
        self._hyper = {}                    # lag -> dict(buf=device float[7], step=host mirror of buf[5], host=(lr, ...))
        self._captured_plan = None
        self.grad_scale = 1.0          # multiplied into the gradient inside the kernel (GradScaler's 1/scale)
        for i, p in enumerate(params):
            o = self.arena.offsets[i]
            self.state[p] = dict(step=torch.tensor(0.0),
                                 exp_avg=self.exp_avg[o:o + p.numel()].view(p.shape),
                                 exp_avg_sq=self.exp_avg_sq[o:o + p.numel()].view(p.shape))

    def zero_grad(self, set_to_none=True):
        self.arena.zero_grad()

    def release(self):
        """Give the parameters their own storage back and drop the arena (call before discarding the optimizer when the
        model lives on, e.g. to build a new optimizer over the same parameters)."""
        self.arena.release()

    # -- hyper-parameters in device memory (svc_adamw_f32 reads them there: hipGraph-safe) ----------------------------
    def _host_hyper(self):
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        return (float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]), float(self.grad_scale))

    def _hyper_for(self, lag, step_before):
        """Device float[7] for the parameters whose step count lags the optimizer's by `lag` (0 for all of them unless
        some parameter was skipped once); its step field holds `step_before` and is advanced on the device."""
        ent = self._hyper.get(lag)
        hh = self._host_hyper()
        if ent is None or ent["step"] != step_before or ent["host"] != hh:
            lr, b1, b2, eps, wd, gs = hh
            buf = ent["buf"] if ent is not None else torch.empty(7, device=self.arena.device, dtype=torch.float32)
            if torch.cuda.is_current_stream_capturing():
                raise S.SvcError("FusedAdamW: hyper-parameters changed inside a hipGraph capture; call sync_hyper() "
                                 "before capturing / replaying")
            buf.copy_(torch.tensor([lr, b1, b2, eps, wd, float(step_before), gs], dtype=torch.float32))
            ent = dict(buf=buf, step=step_before, host=hh)
            self._hyper[lag] = ent
        return ent

    def sync_hyper(self):
        """Push lr / betas / eps / weight_decay / grad_scale changes (lr scheduler, GradScaler) to the device copies.
        step() does this itself; a captured training step must call it before every replay."""
        hh = self._host_hyper()
        for lag, ent in self._hyper.items():
            if ent["host"] != hh:
                lr, b1, b2, eps, wd, gs = hh
                ent["buf"].copy_(torch.tensor([lr, b1, b2, eps, wd, float(ent["step"]), gs], dtype=torch.float32))
                ent["host"] = hh

    def note_replayed_step(self):
        """Host bookkeeping for one replay of a hipGraph that captured step(): the device-side step counters advanced
        by themselves; mirror that in the per-parameter step counts / hyper mirrors (same parameters as at capture)."""
        if self._captured_plan is None:
            raise S.SvcError("FusedAdamW.note_replayed_step: no step() was captured")
        touched, lags = self._captured_plan
        self._steps = [n + 1 if t else n for n, t in zip(self._steps, touched)]
        self._gstep += 1
        for lag in lags:
            self._hyper[lag]["step"] += 1
        torch.autograd.graph.increment_version([p for p, t in zip(self.arena.params, touched) if t])

    def snapshot(self):
        """Everything step() changes (used to run un-counted warm-up iterations before a hipGraph capture)."""
        return dict(param=self.arena.param.clone(), m=self.exp_avg.clone(), v=self.exp_avg_sq.clone(),
                    steps=list(self._steps), gstep=self._gstep, hsteps={k: e["step"] for k, e in self._hyper.items()})

    def restore(self, snap, device=True):
        self._steps = list(snap["steps"])
        self._gstep = snap["gstep"]
        for lag, ent in self._hyper.items():
            ent["step"] = snap["hsteps"].get(lag, self._gstep - lag)
        if device:
            with torch.no_grad():
                self.arena.param.copy_(snap["param"])
                self.exp_avg.copy_(snap["m"])
                self.exp_avg_sq.copy_(snap["v"])
                for lag, ent in self._hyper.items():
                    lr, b1, b2, eps, wd, gs = ent["host"]
                    ent["buf"].copy_(torch.tensor([lr, b1, b2, eps, wd, float(ent["step"]), gs], dtype=torch.float32))
            torch.autograd.graph.increment_version(self.arena.params)

    @torch.no_grad()
    def grads_finite(self):
        """True when every gradient produced since the last zero_grad is finite (what GradScaler.unscale_ / step look at,
        train.py:193-195,211-212).  Reads one device scalar: a host synchronisation, like the reference's `found_inf.item()`."""
        a = self.arena
        a.collect()
        runs = a.touched_runs()
        if not runs:
            return True
        ok = torch.stack([torch.isfinite(a.grad[s:e]).all() for s, e, _ in runs]).all()
        return bool(ok)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        a = self.arena
        a.check_views()
        if not any(a.touched):
            return loss
        a.collect()
        self._gstep += 1
        self._steps = [n + 1 if t else n for n, t in zip(self._steps, a.touched)]
        lags = [self._gstep - n for n in self._steps]
        runs = a.touched_runs(lags)              # one run == one launch; normally a single run over the whole arena
        used = []
        for lag in sorted(set(r[2] for r in runs)):
            ent = self._hyper_for(lag, self._gstep - lag - 1)
            S.adamw_advance(ent["buf"])
            ent["step"] += 1
            used.append(lag)
        for s, e, lag in runs:
            S.adamw_step(a.param[s:e], a.grad[s:e], self.exp_avg[s:e], self.exp_avg_sq[s:e], self._hyper[lag]["buf"])
        if torch.cuda.is_current_stream_capturing():
            self._captured_plan = (list(a.touched), used)
        # the kernel wrote the arena behind torch's back: bump the version counters so that cached packed weights
        # (svc_nn._PackedMixin, keyed on `_version`) are rebuilt and autograd's saved-tensor checks stay valid
        torch.autograd.graph.increment_version([p for p, t in zip(a.params, a.touched) if t])
        return loss

    # -- checkpoint compatibility with torch.optim.AdamW (utils.py:155-199 saves/loads optimizer.state_dict()) --------
    def state_dict(self):
        for p, n in zip(self.arena.params, self._steps):
            self.state[p]["step"] = torch.tensor(float(n))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        with torch.no_grad():
            for i, p in enumerate(self.arena.params):
                st = self.state.get(p)
                if not st:
                    continue
                o = self.arena.offsets[i]
                for name, flat in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
                    view = flat[o:o + p.numel()].view(p.shape)
                    if st[name].data_ptr() != view.data_ptr():
                        view.copy_(st[name])
                        st[name] = view
                self._steps[i] = int(float(st["step"]))
        self._gstep = max(self._steps) if self._steps else 0
        self._hyper = {}


class LossScaler:
    """torch.cuda.amp.GradScaler's rule (train.py:143 `GradScaler(enabled=hps.train.fp16_run)`, :192-213) for the fp16 mode: the
    loss is multiplied by `scale` before backward, the optimizer divides the gradients by it again (FusedAdamW.grad_scale: inside
    the one AdamW launch) and skips its step when a gradient is not finite; `update()` — once per iteration, after both
    optimizers — halves the scale if either of them skipped, doubles it after `growth_interval` clean iterations.  Defaults are
    GradScaler's (65536, x2, x0.5, 2000).  Host-side: one device->host flag per optimizer step, as in the reference."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.scale, self.growth_factor, self.backoff_factor, self.growth_interval = float(init_scale), growth_factor, backoff_factor, growth_interval
        self._good, self._found_inf, self.skipped = 0, False, 0

    def step(self, optimizer):
        if not optimizer.grads_finite():
            self._found_inf = True
            self.skipped += 1
            return False
        optimizer.grad_scale = 1.0 / self.scale
        optimizer.step()
        return True

    def update(self):
        if self._found_inf:
            self.scale *= self.backoff_factor
            self._good = 0
        else:
            self._good += 1
            if self._good == self.growth_interval:
                self.scale *= self.growth_factor
                self._good = 0
        self._found_inf = False

    def state_dict(self):
        return dict(scale=self.scale, growth_tracker=self._good)

    def load_state_dict(self, d):
        self.scale, self._good = float(d["scale"]), int(d.get("growth_tracker", 0))
