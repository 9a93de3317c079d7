"""MI355X-native mirror of diffusion/solver.py: `train` (:93-199: the loop behind `train_diff.py`), `test` (:13-90: the
validation pass) and, underneath, `TrainStep` = one iteration (zero_grad -> Unit2Mel(infer=False) loss -> backward -> AdamW
step -> StepLR step, :116-147) with the optimizer set-up of train_diff.py:55-60.  fp32.
Re-designed rather than mirrored: `optim.FusedAdamW` over a flat arena (one launch), whole-iteration hipGraph replay (the
crops of one run all have `duration` seconds: one graph, plus one for a short last batch), `data_parallel.DataParallel`
when a process group is up (the reference's train_diff.py is single-GPU), rank 0 logs / validates / saves.
`train()` takes the reference's argument list; an `torch.optim.AdamW` handed in by an unchanged `train_diff.py` is replaced
by a FusedAdamW with the same hyper-parameters and state, and its StepLR keeps driving the learning rate.
`train.amp_dtype: bf16 / fp16` run the model call with 16-bit matrix operands (svc_hip.mma_mode); fp16 adds the GradScaler rule
(optim.LossScaler) and launches eagerly."""
import os
import time

import numpy as np
import torch

from data_parallel import DataParallel
import svc_hip as S
from optim import FusedAdamW, LossScaler


def build_optimizer(model, lr, weight_decay=0.0, gamma=0.5, decay_step=100000, initial_global_step=0):
    """train_diff.py:55-60 — AdamW defaults with lr / weight_decay from the config, StepLR state from the global step."""
    opt = FusedAdamW(model.parameters(), lr=lr * gamma ** max((initial_global_step - 2) // decay_step, 0), betas=(0.9, 0.999),
                     eps=1e-8, weight_decay=weight_decay)
    for pg in opt.param_groups:
        pg["initial_lr"] = lr
    return opt


class TrainStep:
    def __init__(self, model, optimizer, gamma=0.5, decay_step=100000, initial_global_step=0, amp_dtype="fp32", scheduler=None):
        # solver.py:107-115,127-131: `amp_dtype` bf16 / fp16 runs the model call under torch.autocast.  bf16: the engine's form of
        # that region — svc_hip.mma_mode: 16-bit operands on the matrix pipe for the convolutions and their gradients, fp32
        # accumulation / storage / master weights (bf16 needs no GradScaler: nothing is stored in it).
        if amp_dtype not in ("fp32", "bf16", "fp16"):
            raise ValueError(" [x] Unknown amp_dtype: " + str(amp_dtype))
        self.mma = {"fp32": S.MMA_F32, "bf16": S.MMA_BF16, "fp16": S.MMA_F16}[amp_dtype]
        # fp16 operands need the reference's GradScaler (solver.py:105,134-137): scale the loss, skip the step on overflow — a host
        # decision per step, so this mode launches eagerly (no whole-iteration hipGraph)
        self.scaler = LossScaler() if amp_dtype == "fp16" else None
        self.model, self.opt = model, optimizer
        self.gamma, self.decay_step = gamma, decay_step
        self.scheduler = scheduler           # a torch lr_scheduler already attached to `optimizer`: it replaces the built-in StepLR
        if scheduler is not None:
            optimizer._opt_called = True     # replayed steps never call optimizer.step(): silence torch's call-order warning
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

    def _fwd_bwd(self, data, noise):
        S.wgrad_slab.active = True       # weight / bias gradients of the iteration accumulate into one pre-zeroed slab
        S.wgrad_slab.reset()
        self.opt.zero_grad()
        self.plan_sets.enter("fwd", self.model.parameters())     # every conv weight of the pass prepared in one launch
        try:
            with S.mma_mode(self.mma):
                loss = self.model(data["units"].float(), data["f0"], data["volume"], data["spk_id"], aug_shift=data.get("aug_shift"),
                                  gt_spec=data["mel"].float(), infer=False, k_step=getattr(self._mod(), "k_step_max", None), noise=noise)
        finally:
            self.plan_sets.leave("fwd")
        (loss * self.scaler.scale if self.scaler is not None else loss).backward()
        return loss.detach()

    def _body(self, data, noise):
        try:
            loss = self._fwd_bwd(data, noise)
            if self.scaler is None:
                self.opt.step()
            else:
                self.scaler.step(self.opt)
                self.scaler.update()
        finally:
            S.wgrad_slab.active = False  # also when the step raises: later backward passes must not get views of this slab
        return loss

    def _reducer(self):
        return getattr(self.model, "reducer", None)

    def _mod(self):
        return self.model.module if isinstance(self.model, DataParallel) else self.model

    def _sched_step(self):                       # lr_scheduler.StepLR.step() (solver.py:147), chainable form
        if self.scheduler is not None:
            if getattr(self, "_sched_started", False):   # the scheduler's constructor already took its first step
                self.scheduler.step()
            self._sched_started = True
            return
        self.sched_epoch += 1
        if self.sched_epoch != 0 and self.sched_epoch % self.decay_step == 0:
            for pg in self.opt.param_groups:
                pg["lr"] = pg["lr"] * self.gamma

    def __call__(self, data, noise=None):
        """data: dict(units [B,T,n_unit], f0 [B,T,1], volume [B,T,1], spk_id [B,1], mel [B,T,M], aug_shift [B,1,1] | None)
        on the device; noise: optional dict(t [B] long, noise [B,1,M,T]).  Returns the loss (0-dim device tensor)."""
        if not self.use_graph or self.scaler is not None:
            loss = self._body(data, noise)
            self._sched_step()
            return loss
        keys = sorted(k for k, v in data.items() if torch.is_tensor(v))
        nkeys = sorted(noise) if noise else []
        flat = [data[k] for k in keys] + [noise[k] for k in nkeys]
        sig = tuple((k, tuple(t.shape), str(t.dtype)) for k, t in zip(keys + nkeys, flat))
        red = self._reducer()
        ent = self._graphs.get(sig)
        if ent is None:
            static = [t.clone() for t in flat]
            sdata = dict(zip(keys, static[:len(keys)]))
            snoise = dict(zip(nkeys, static[len(keys):])) if nkeys else None
            snap = self.opt.snapshot()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            import contextlib
            # warm-up WITHOUT collectives: ranks read their own shards, so only this rank may be meeting this shape now (its
            # results are discarded anyway); the replay below issues the one bucket-ordered reduction every rank issues per step
            with torch.cuda.stream(side), (red.no_sync() if red is not None else contextlib.nullcontext()):
                for _ in range(2):
                    self._body(sdata, snoise)
            torch.cuda.current_stream().wait_stream(side)
            self.opt.restore(snap)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            if red is None:
                with S.graph_capture(graph):
                    out = self._body(sdata, snoise)
                touched = None
                self.opt.restore(snap, device=False)
            else:
                # data parallel: the autograd hooks that launch the bucket all-reduces do not run in a replay and collectives
                # are not captured -> graph[zero_grad, forward, backward] -> all-reduce(arena) -> AdamW (one eager launch)
                try:
                    with red.no_sync(), S.graph_capture(graph):
                        out = self._fwd_bwd(sdata, snoise)
                finally:
                    S.wgrad_slab.active = False
                touched = list(self.opt.arena.touched)
            ent = (graph, static, out, touched)
            self._graphs[sig] = ent
        graph, static, out, touched = ent
        for s, t in zip(static, flat):
            s.copy_(t, non_blocking=True)
        if touched is None:
            self.opt.sync_hyper()
            graph.replay()
            self.opt.note_replayed_step()
        else:
            graph.replay()
            red.reduce_all()
            self.opt.arena.touched = list(touched)
            self.opt.step()
        self._sched_step()
        return out.clone()


# ---- the loop behind `svc_run.py train_diff.py -c configs/diffusion.yaml` (reference diffusion/solver.py:13-199) -------------
def _fused(optimizer, scheduler):
    """An unchanged train_diff.py builds torch.optim.AdamW (+ StepLR): same hyper-parameters and state on the flat arena."""
    if isinstance(optimizer, FusedAdamW):
        return optimizer
    g = optimizer.param_groups[0]
    params = [p for grp in optimizer.param_groups for p in grp["params"]]
    fused = FusedAdamW(params, lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"])
    if optimizer.state:
        fused.load_state_dict(optimizer.state_dict())
    for pg in fused.param_groups:
        pg["lr"], pg["weight_decay"] = g["lr"], g["weight_decay"]
        if "initial_lr" in g:
            pg["initial_lr"] = g["initial_lr"]
    if scheduler is not None:
        scheduler.optimizer = fused               # torch schedulers read / write optimizer.param_groups only
    return fused


def _rank_world():
    import torch.distributed as dist
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def _load_audio(path, sr, like):
    """`librosa.load(path, sr=sr)` + `to_mono` of the validation log (:80-83) without librosa: wav decode, channel mean,
    windowed-sinc resample on the device when the file's rate differs."""
    import svc_audio
    data, file_sr = svc_audio.read_audio(path)
    audio = torch.from_numpy(data.mean(0)).unsqueeze(0).to(like)
    if file_sr != sr:
        audio = svc_audio.Resampler(file_sr, sr)(audio)
    return audio


def test(args, model, vocoder, loader_test, saver):
    """Reference :13-90 — every validation file: sample the mel (`infer.method`, `infer.speedup`; shallow from the ground truth
    when the model is a shallow one), vocode, print the real-time factor; the loss is the mean over `batch_size` random
    (t, noise) draws per file; spectrogram figure + audio go to the saver."""
    print(" [*] testing...")
    model.eval()
    test_loss, rtf_all = 0.0, []
    num_batches = len(loader_test)
    with torch.no_grad():
        for bidx, data in enumerate(loader_test):
            fn = data["name"][0].split("/")[-1]
            speaker = data["name"][0].split("/")[-2] if "/" in data["name"][0] else ""
            print("--------")
            print("{}/{} - {}".format(bidx, num_batches, fn))
            for k in data.keys():
                if not k.startswith("name"):
                    data[k] = data[k].to(args.device)
            print(">>", data["name"][0])
            st = time.time()
            mel = model(data["units"], data["f0"], data["volume"], data["spk_id"],
                        gt_spec=None if model.k_step_max == model.timesteps else data["mel"], infer=True,
                        infer_speedup=args.infer.speedup, method=args.infer.method, k_step=model.k_step_max, use_tqdm=False)
            signal = vocoder.infer(mel, data["f0"])
            torch.cuda.synchronize()
            run_time = time.time() - st
            song_time = signal.shape[-1] / args.data.sampling_rate
            rtf = run_time / song_time
            print("RTF: {}  | {} / {}".format(rtf, run_time, song_time))
            rtf_all.append(rtf)
            losses = [model(data["units"], data["f0"], data["volume"], data["spk_id"], gt_spec=data["mel"], infer=False,
                            k_step=model.k_step_max) for _ in range(args.train.batch_size)]
            test_loss += float(torch.stack([l.detach() for l in losses]).sum())
            saver.log_spec(f"{speaker}_{fn}.wav", data["mel"], mel)
            audio = _load_audio(data["name_ext"][0], args.data.sampling_rate, signal)
            saver.log_audio({f"{speaker}_{fn}_gt.wav": audio, f"{speaker}_{fn}_pred.wav": signal})
    test_loss /= args.train.batch_size
    test_loss /= max(num_batches, 1)
    print(" [test_loss] test_loss:", test_loss)
    print(" Real Time Factor", np.mean(rtf_all) if rtf_all else float("nan"))
    return test_loss


def train(args, initial_global_step, model, optimizer, scheduler, vocoder, loader_train, loader_test):
    """Reference :93-199 with the same argument list, log lines, checkpoint names and intervals.  The batch loop body is
    `TrainStep` (HIP forward + backward, fused AdamW) replayed from a hipGraph unless SVC_TRAIN_GRAPH=0; the loss is read back
    only on logging steps (the reference's `torch.isnan(loss)` check, one host sync per step, is made there)."""
    from .logger import utils
    from .logger.saver import Saver
    rank, world = _rank_world()
    saver = Saver(args, initial_global_step=initial_global_step) if rank == 0 else None
    info = (lambda m: saver.log_info(m)) if rank == 0 else (lambda m: None)
    info("--- model size ---")
    info(utils.get_network_paras_amount({"model": model}))
    if args.train.amp_dtype not in ("fp32", "fp16", "bf16"):
        raise ValueError(" [x] Unknown amp_dtype: " + str(args.train.amp_dtype))
    optimizer = _fused(optimizer, scheduler)
    net = DataParallel(model) if world > 1 and not isinstance(model, DataParallel) else model
    step = TrainStep(net, optimizer, gamma=args.train.gamma, decay_step=args.train.decay_step,
                     initial_global_step=initial_global_step, amp_dtype=args.train.amp_dtype, scheduler=scheduler)
    step.enable_graph(os.environ.get("SVC_TRAIN_GRAPH", "1") == "1")
    core = net.module if isinstance(net, DataParallel) else net
    # rank 0 validates and checkpoints alone (reference :170-189) while the others would run into the next step's all-reduce and
    # sit there until RCCL's watchdog (10 min) aborts the job on a long validation set: they wait at a barrier of a side group
    # with a long timeout instead
    val_group = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        val_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(hours=12))
    num_batches = len(loader_train)
    core.train()
    info("======= start training =======")
    info("epoch|batch_idx/num_batches|output_dir|batch/s|lr|time|step")
    global_step = initial_global_step
    max_steps = int(os.environ.get("SVC_TRAIN_DIFF_MAX_STEPS", "0"))        # 0 = run all `train.epochs` (tests / smoke runs bound it)
    for epoch in range(args.train.epochs):
        for batch_idx, data in enumerate(loader_train):
            global_step += 1
            if saver is not None:
                saver.global_step_increment()
            batch = {}
            for k, v in data.items():          # float64 (.aug_mel.npy) / half (cache_fp16) items -> fp32: one graph shape, as the
                if not k.startswith("name"):   # reference's `.float()` at the model call
                    v = v.to(args.device, non_blocking=True)
                    batch[k] = v.float() if v.is_floating_point() else v
            loss = step(batch)
            if global_step % args.train.interval_log == 0:
                val = float(loss)
                if val != val:
                    raise ValueError(" [x] nan loss ")
                if rank == 0:
                    lr = optimizer.param_groups[0]["lr"]
                    info("epoch: {} | {:3d}/{:3d} | {} | batch/s: {:.2f} | lr: {:.6} | loss: {:.3f} | time: {} | step: {}".format(
                        epoch, batch_idx, num_batches, args.env.expdir, args.train.interval_log / saver.get_interval_time(), lr,
                        val, saver.get_total_time(), saver.global_step))
                    saver.log_value({"train/loss": val})
                    saver.log_value({"train/lr": lr})
            if global_step % args.train.interval_val == 0 and rank == 0:
                saver.save_model(core, optimizer if args.train.save_opt else None, postfix=f"{saver.global_step}")
                last_val_step = saver.global_step - args.train.interval_val
                if last_val_step % args.train.interval_force_save != 0:
                    saver.delete_model(postfix=f"{last_val_step}")
                test_loss = test(args, core, vocoder, loader_test, saver)
                info(" --- <validation> --- \nloss: {:.3f}. ".format(test_loss))
                saver.log_value({"validation/loss": test_loss})
                core.train()
            if global_step % args.train.interval_val == 0 and val_group is not None:
                import torch.distributed as dist
                dist.barrier(group=val_group)
            if max_steps and global_step - initial_global_step >= max_steps:
                return global_step
    return global_step
