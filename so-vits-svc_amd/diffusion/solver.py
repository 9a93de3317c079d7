"""MI355X-native mirror of the training step of diffusion/solver.py (`train`, :93-147) and the optimizer set-up of
train_diff.py:55-60: one iteration = zero_grad -> Unit2Mel(infer=False) loss -> backward -> AdamW step -> StepLR step,
fp32.  Re-designed rather than mirrored: `optim.FusedAdamW` over a flat arena (one launch), optional whole-iteration
hipGraph replay, `data_parallel.DataParallel` when a process group is up (the reference's train_diff.py is single-GPU).
Out of scope: Saver / TensorBoard / validation audio (:13-90,174-200), fp16/bf16 autocast (`amp_dtype` must be fp32)."""
import torch

from data_parallel import DataParallel
import svc_hip as S
from optim import FusedAdamW


def build_optimizer(model, lr, weight_decay=0.0, gamma=0.5, decay_step=100000, initial_global_step=0):
    """train_diff.py:55-60 — AdamW defaults with lr / weight_decay from the config, StepLR state from the global step."""
    opt = FusedAdamW(model.parameters(), lr=lr * gamma ** max((initial_global_step - 2) // decay_step, 0), betas=(0.9, 0.999),
                     eps=1e-8, weight_decay=weight_decay)
    for pg in opt.param_groups:
        pg["initial_lr"] = lr
    return opt


class TrainStep:
    def __init__(self, model, optimizer, gamma=0.5, decay_step=100000, initial_global_step=0, amp_dtype="fp32"):
        if amp_dtype != "fp32":
            raise NotImplementedError("amp_dtype fp16/bf16 is not implemented: the MI355X engine trains in fp32")
        self.model, self.opt = model, optimizer
        self.gamma, self.decay_step = gamma, decay_step
        # torch's StepLR(optimizer, step_size, gamma, last_epoch=initial_global_step - 2) (train_diff.py:60): the constructor
        # performs one step() (last_epoch -> initial_global_step - 1), then every scheduler.step() increments
        # last_epoch and multiplies the CURRENT lr by gamma when last_epoch is a non-zero multiple of step_size
        self.sched_epoch = initial_global_step - 2
        self._sched_step()                                          # the constructor's step (it CAN decay: last_epoch 3 -> 4 at step 4)
        self.use_graph = False
        self._graphs = {}
        self.plan_sets = S.PlanSets()

    def enable_graph(self, on=True):
        self.use_graph = bool(on)
        if not on:
            self._graphs.clear()
        return self

    def _body(self, data, noise):
        S.wgrad_slab.active = True       # weight / bias gradients of the iteration accumulate into one pre-zeroed slab
        try:
            S.wgrad_slab.reset()
            self.opt.zero_grad()
            self.plan_sets.enter("fwd", self.model.parameters())     # every conv weight of the pass prepared in one launch
            try:
                loss = self.model(data["units"].float(), data["f0"], data["volume"], data["spk_id"], aug_shift=data.get("aug_shift"),
                                  gt_spec=data["mel"].float(), infer=False, k_step=getattr(self._mod(), "k_step_max", None), noise=noise)
            finally:
                self.plan_sets.leave("fwd")
            loss.backward()
            self.opt.step()
        finally:
            S.wgrad_slab.active = False  # also when the step raises: later backward passes must not get views of this slab
        return loss.detach()

    def _mod(self):
        return self.model.module if isinstance(self.model, DataParallel) else self.model

    def _sched_step(self):                       # lr_scheduler.StepLR.step() (solver.py:147), chainable form
        self.sched_epoch += 1
        if self.sched_epoch != 0 and self.sched_epoch % self.decay_step == 0:
            for pg in self.opt.param_groups:
                pg["lr"] = pg["lr"] * self.gamma

    def __call__(self, data, noise=None):
        """data: dict(units [B,T,n_unit], f0 [B,T,1], volume [B,T,1], spk_id [B,1], mel [B,T,M], aug_shift [B,1,1] | None)
        on the device; noise: optional dict(t [B] long, noise [B,1,M,T]).  Returns the loss (0-dim device tensor)."""
        if not self.use_graph:
            loss = self._body(data, noise)
            self._sched_step()
            return loss
        keys = sorted(k for k, v in data.items() if torch.is_tensor(v))
        nkeys = sorted(noise) if noise else []
        flat = [data[k] for k in keys] + [noise[k] for k in nkeys]
        sig = tuple((k, tuple(t.shape), str(t.dtype)) for k, t in zip(keys + nkeys, flat))
        ent = self._graphs.get(sig)
        if ent is None:
            static = [t.clone() for t in flat]
            sdata = dict(zip(keys, static[:len(keys)]))
            snoise = dict(zip(nkeys, static[len(keys):])) if nkeys else None
            snap = self.opt.snapshot()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._body(sdata, snoise)
            torch.cuda.current_stream().wait_stream(side)
            self.opt.restore(snap)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with S.graph_capture(graph):
                out = self._body(sdata, snoise)
            self.opt.restore(snap, device=False)
            ent = (graph, static, out)
            self._graphs[sig] = ent
        graph, static, out = ent
        for s, t in zip(static, flat):
            s.copy_(t, non_blocking=True)
        self.opt.sync_hyper()
        graph.replay()
        self.opt.note_replayed_step()
        self._sched_step()
        return out.clone()
