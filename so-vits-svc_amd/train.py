"""MI355X-native mirror of the reference's training step (train.py:44-213): model/optimizer construction as in `run`
(train.py:74-90) and the body of `train_and_evaluate`'s batch loop (train.py:150-213) — D step, G step, fp32.

What is re-designed rather than mirrored:
  * optimizers are `optim.FusedAdamW` over a flat parameter arena (one launch per step),
  * `data_parallel.DataParallel` instead of torch DDP: arena-slice buckets all-reduced over RCCL while backward runs,
  * in the G step the discriminator runs with its parameters frozen (`no_param_grads`): no D weight gradients are
    computed or all-reduced there, and the real branch (whose feature maps are detached by `feature_loss`) runs
    without a tape,
  * no `.item()` host syncs inside the step: the losses come back as device scalars.
`main()` / `run()` / `train_and_evaluate()` / `evaluate()` at the end of the file are the entry point behind the reference's CLI
(`svc_run.py train.py -c ... -m ...`): same logs/<model> layout, checkpoints, epoch / warm-up / ExponentialLR bookkeeping.
`fp16_run` runs the reference's autocast regions with 16-bit matrix operands (svc_hip.mma_mode): `half_type: bf16` on
v_mfma_f32_32x32x16_bf16, `half_type: fp16` on v_mfma_f32_32x32x16_f16 with the GradScaler rule (LossScaler; eager launches).
"""
import torch
import torch.distributed as dist

import os

import models
import modules.commons as commons
import svc_autograd as A
import svc_hip as S
from data_parallel import DataParallel, no_param_grads
from modules.losses import discriminator_loss, feature_loss, generator_loss, kl_loss
from modules.mel_processing import mel_spectrogram_torch, spec_to_mel_torch
from optim import FusedAdamW, LossScaler


def _get(h, name):
    return h[name] if isinstance(h, dict) else getattr(h, name)


def _model_kwargs(hm):
    d = dict(hm) if isinstance(hm, dict) else {k: v for k, v in hm.items()}
    return d


def build(hps, device):
    """train.py:74-90: nets on `device`, AdamW for each, data-parallel wrappers (no-ops at world size 1)."""
    data, train, model = _get(hps, "data"), _get(hps, "train"), _get(hps, "model")
    net_g = models.SynthesizerTrn(_get(data, "filter_length") // 2 + 1,
                                  _get(train, "segment_size") // _get(data, "hop_length"),
                                  **_model_kwargs(model)).to(device)
    use_sn = model.get("use_spectral_norm", False) if isinstance(model, dict) else getattr(model, "use_spectral_norm", False)
    net_d = models.MultiPeriodDiscriminator(use_sn).to(device)
    kw = dict(lr=_get(train, "learning_rate"), betas=tuple(_get(train, "betas")), eps=_get(train, "eps"))
    optim_g = FusedAdamW(net_g.parameters(), **kw)
    optim_d = FusedAdamW(net_d.parameters(), **kw)
    net_g = DataParallel(net_g)
    net_d = DataParallel(net_d)
    return net_g, net_d, optim_g, optim_d


def _sn_buffers(net_d):
    """The power-iteration vectors of spectral-norm discriminators (`weight_u` / `weight_v`, models.py:170,205): advanced in
    place by every training-mode forward, so the warm-up and capture runs of a graph must hand them back untouched."""
    mod = net_d.module if hasattr(net_d, "module") else net_d
    return [b for n, b in mod.named_buffers() if n.endswith("weight_u") or n.endswith("weight_v")]


class TrainStep:
    """One iteration of train.py:150-213 on a minibatch already resident on the device."""

    def __init__(self, hps, net_g, net_d, optim_g, optim_d):
        self.hps = hps
        self.net_g, self.net_d, self.optim_g, self.optim_d = net_g, net_d, optim_g, optim_d
        d, t = _get(hps, "data"), _get(hps, "train")
        self.n_fft, self.n_mels, self.sr = _get(d, "filter_length"), _get(d, "n_mel_channels"), _get(d, "sampling_rate")
        self.hop, self.win = _get(d, "hop_length"), _get(d, "win_length")
        self.fmin, self.fmax = _get(d, "mel_fmin"), _get(d, "mel_fmax")
        self.segment_size = _get(t, "segment_size")
        self.c_mel, self.c_kl = _get(t, "c_mel"), _get(t, "c_kl")
        # train.py:114,143,166,187,198: `fp16_run` wraps the generator forward, the mel of y_hat and both discriminator forwards
        # in torch.autocast(dtype=half_type) and computes every loss in fp32.  The engine's form of that region is
        # svc_hip.mma_mode: convolutions (forward, input- and weight-gradient) take bf16 operands on v_mfma_f32_32x32x16_bf16
        # with fp32 accumulation; tensors and master weights stay fp32 — so no GradScaler is needed (bf16 has fp32's exponent
        # range, and nothing is stored in it), the mel / attention products / element-wise maths keep fp32, and a shape without
        # a 16-bit kernel runs in fp32.  `half_type: fp16`: the same with fp16 operands plus the GradScaler rule (LossScaler).
        self.mma, self.scaler = S.MMA_F32, None
        if _get(t, "fp16_run"):
            half = t.get("half_type", "fp16") if isinstance(t, dict) else getattr(t, "half_type", "fp16")
            if half == "bf16":
                self.mma = S.MMA_BF16
            else:
                # fp16 operands (v_mfma_f32_32x32x16_f16) need the reference's loss scaling: a gradient below 6e-8 rounds to zero
                # as an fp16 operand.  The scaler's skip-on-overflow is a host decision per optimizer step (as in the reference:
                # GradScaler reads found_inf back), so this mode launches eagerly — no whole-iteration hipGraph.
                self.mma, self.scaler = S.MMA_F16, LossScaler()

        self.use_graph = False
        import collections
        self._graphs = collections.OrderedDict()      # batch shape -> captured iteration, least recently used first
        self.graph_evictions, self._cap_logged = 0, False
        self.dp_ordered = False       # eager launches with the replayed form's collective order (a rank whose capture failed)
        self._guard, self._since_check = None, 0
        self.finite_every = max(1, int(os.environ.get("SVC_TRAIN_FINITE_EVERY", "200")))
        # every captured shape keeps its own ~10 GB tape alive: bound the count (bucketed batches need 2-4; a data set of
        # odd batch sizes falls back to eager launches for the shapes beyond the cap instead of growing without limit)
        # default: every frame bucket of the loader (data_utils.FRAME_BUCKETS) + a few for the short last batch of an epoch, so
        # that a bucketed data set never cycles through evictions (a miss costs two warm-ups + a capture + the tape re-allocation)
        try:
            from data_utils import FRAME_BUCKETS
            dflt = len(FRAME_BUCKETS) + 3
        except Exception:      # noqa: BLE001 — TrainStep is usable without the loader module
            dflt = 12
        self.max_graphs = int(os.environ.get("SVC_TRAIN_GRAPH_MAX", str(dflt)))
        self.eager_fallbacks = 0
        self.plan_sets = S.PlanSets()     # one-launch weight preparation per forward pass (G, D, D again after its step)

    def enable_graph(self, on=True):
        """Replay the whole iteration (forward, both backwards, both optimizer steps: ~7000 launches) from ONE hipGraph
        per input shape — the eager step is bound by the host's launch rate, not by the GPU.  Random draws then come
        from the device generator (commons.DEVICE_RNG).  Bucket / pad T in the data pipeline to bound the graph count."""
        self.use_graph = bool(on)
        if not on:
            self._graphs.clear()
        return self

    def __call__(self, items, noise=None):
        """items = (c, f0, spec, y, spk, lengths, uv, volume) as the reference's collate returns them (train.py:151);
        returns a dict of 0-dim device tensors."""
        out = self._dispatch(items, noise)
        self.check_finite()
        return out

    def _dispatch(self, items, noise=None):
        if self.use_graph and self.scaler is None:
            items = self._dense_spec(items)
            if all(r is not None for r in self._reducers()):
                return self._call_graph_dp(items, noise)
            return self._call_graph(items, noise)
        if self.dp_ordered and all(r is not None for r in self._reducers()):
            return self._step_body_dp_eager(items, noise)
        return self._step_body(items, noise)

    def _dense_spec(self, items):
        """The graph paths key on tensor shapes and copy every item into static buffers.  A loader batch whose `spec` slot is
        a data_utils.SpecContextBatch (items without a cached .spec.pt, or vol-augmented ones: a random half of the items
        with vol_aug) or None is turned into the dense [B, F, T] spectrogram HERE — eagerly, outside any capture: the frame
        count comes from the host (`int(n_frames.max())`) and would otherwise be frozen into the captured graph."""
        c, f0, spec, y, spk, lengths, uv, volume = items
        if torch.is_tensor(spec):
            return items
        if spec is None:
            from data_utils import batch_spectrogram
            spec = batch_spectrogram(y, lengths, self.n_fft, self.sr, self.hop, self.win)
        else:
            from data_utils import context_spectrogram
            spec = context_spectrogram(spec.to(y.device), self.n_fft, self.sr, self.hop, self.win)
        if spec.shape[2] < c.shape[2]:            # keep the frame axis the collate's (graph key = padded batch shape)
            spec = torch.nn.functional.pad(spec, (0, c.shape[2] - spec.shape[2]))
        return (c, f0, spec, y, spk, lengths, uv, volume)

    def _call_graph(self, items, noise=None):
        nkeys = sorted(noise) if noise is not None else []
        items = list(items) + [noise[k] for k in nkeys]
        key = tuple((tuple(t.shape), str(t.dtype)) if t is not None else None for t in items) + tuple(nkeys)
        ent = self._graph_get(key)
        n_in = len(items) - len(nkeys)
        if ent is None and self._graph_cap_reached():
            self.eager_fallbacks += 1
            return self._step_body(items[:n_in], dict(zip(nkeys, items[n_in:])) if nkeys else None)
        if ent is None:
            commons.DEVICE_RNG = True
            static = [t.clone() if t is not None else None for t in items]
            run = lambda: self._step_body(static[:n_in], dict(zip(nkeys, static[n_in:])) if nkeys else None)
            snaps = (self.optim_g.snapshot(), self.optim_d.snapshot())
            sn = _sn_buffers(self.net_d)
            sn_saved = [b.clone() for b in sn]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):                       # warm-up: weight packs, kernel attributes, allocator pools
                    run()
            torch.cuda.current_stream().wait_stream(side)
            self.optim_g.restore(snaps[0])               # ... without counting as training steps
            self.optim_d.restore(snaps[1])
            for b, v in zip(sn, sn_saved):
                b.copy_(v)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with S.graph_capture(graph):
                out = run()
            self.optim_g.restore(snaps[0], device=False)   # capture ran the host bookkeeping but no kernels
            self.optim_d.restore(snaps[1], device=False)
            ent = (graph, static, out)
            self._graphs[key] = ent
        graph, static, out = ent
        self._serialize_replays()
        for s, t in zip(static, items):
            if s is not None:
                s.copy_(t, non_blocking=True)
        self.optim_g.sync_hyper()
        self.optim_d.sync_hyper()
        graph.replay()
        self.optim_d.note_replayed_step()
        self.optim_g.note_replayed_step()
        res = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in out.items()}
        self._mark_replay()
        return res

    def _serialize_replays(self):
        """Wait on the host for the previous replay of the iteration graph before enqueueing the next one (default ON;
        SVC_TRAIN_SERIALIZE=0 opts out).  Round 2 needed this: with a mid-run synchronize (diag SYNC=at2 / at5) the iterations
        after it came out bimodal / NaN when the next replay was enqueued behind a running one (profiles/r02_o_*).  Round 3 re-ran
        those reproducers WITHOUT the wait — 35 runs (profiles/r03p_*, r03q_*, r03r_replay_diag.txt) — and every run reproduced
        the same loss trajectory; the round-2 traces predate that round's LayerNorm / grouped-conv kernel rewrites, but no
        stand-alone reproducer was ever found, i.e. the root cause is NOT known.  The wait costs one graph-launch latency per
        ~100 ms iteration (the host work of the next batch — loader, collate, input copies — still overlaps the running replay:
        the wait sits after it), a silent divergence costs a training run: the guard stays until the round-2 failure is
        explained.  tests/test_train_loop_gpu.py pins back-to-back replays against eager iterations with the wait off."""
        ev = self.__dict__.get("_replay_done")
        if ev is not None and os.environ.get("SVC_TRAIN_SERIALIZE", "1") == "1":
            ev.synchronize()

    def _mark_replay(self):
        ev = self.__dict__.get("_replay_done")
        if ev is None:
            ev = self.__dict__["_replay_done"] = torch.cuda.Event()
        ev.record()

    def _step_body(self, items, noise=None):
        try:
            ctx = self._seg_d(items, noise)
            if self.scaler is None:
                self.optim_d.step()
                out = self._seg_g(ctx)
                self.optim_g.step()
            else:                                    # train.py:192-213: scale -> backward -> unscale -> step (or skip) -> update
                self.scaler.step(self.optim_d)
                out = self._seg_g(ctx)
                self.scaler.step(self.optim_g)
                self.scaler.update()
                out["loss_scale"] = self.scaler.scale
        finally:
            # gradients consumed (or the step raised): later backward passes in this process must not receive views of a
            # slab that the next reset() zeroes
            S.wgrad_slab.active = False
        return out

    def _seg_d(self, items, noise=None):
        """Generator forward + the discriminator step up to (and including) its backward (train.py:151-194)."""
        # weight / bias gradients of this iteration accumulate into one pre-zeroed slab (one memset instead of ~585)
        S.wgrad_slab.active = True
        S.wgrad_slab.reset()
        c, f0, spec, y, spk, lengths, uv, volume = items
        net_g, net_d = self.net_g, self.net_d
        seg_frames = self.segment_size // self.hop
        if spec is None:          # no spectrogram supplied at all: STFT of the padded batch on the GPU
            from data_utils import batch_spectrogram
            spec = batch_spectrogram(y, lengths, self.n_fft, self.sr, self.hop, self.win)
        elif not torch.is_tensor(spec):   # loader items without a cached .spec.pt / vol-augmented audio (data_utils.SpecContextBatch)
            from data_utils import context_spectrogram
            spec = context_spectrogram(spec.to(y.device), self.n_fft, self.sr, self.hop, self.win)
        mel = spec_to_mel_torch(spec, self.n_fft, self.n_mels, self.sr, self.fmin, self.fmax)            # :158-164
        kw = dict(noise=noise) if noise is not None else {}
        self.plan_sets.enter("g", net_g.parameters())
        try:
            with S.mma_mode(self.mma):                                                                    # :166 autocast region
                y_hat, ids_slice, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), pred_lf0, norm_lf0, lf0 = net_g(
                    c, f0, uv, spec, g=spk, c_lengths=lengths, spec_lengths=lengths, vol=volume, **kw)    # :167-169
        finally:
            self.plan_sets.leave("g")
        y_mel = commons.slice_segments(mel, ids_slice, seg_frames)                                        # :171
        y_hat_mel = mel_spectrogram_torch(y_hat.squeeze(1), self.n_fft, self.n_mels, self.sr, self.hop, self.win,
                                          self.fmin, self.fmax)                                           # :172-181
        y = commons.slice_segments(y, ids_slice * self.hop, self.segment_size)                            # :182

        # ---- discriminator step (:184-195) ----
        self.plan_sets.enter("d", net_d.parameters())
        try:
            with S.mma_mode(self.mma):                                                                    # :185, still inside :166's region
                y_d_hat_r, y_d_hat_g, _, _ = net_d(y, y_hat.detach())
        finally:
            self.plan_sets.leave("d")
        loss_disc, _, _ = discriminator_loss(y_d_hat_r, y_d_hat_g)
        self.optim_d.zero_grad()
        (loss_disc * self.scaler.scale if self.scaler is not None else loss_disc).backward()
        return dict(y=y, y_hat=y_hat, y_mel=y_mel, y_hat_mel=y_hat_mel, z_p=z_p, logs_q=logs_q, m_p=m_p, logs_p=logs_p,
                    z_mask=z_mask, pred_lf0=pred_lf0, lf0=lf0, loss_disc=loss_disc)

    def _seg_g(self, ctx):
        """The generator step up to (and including) its backward (train.py:198-211): D frozen, real branch tape-free."""
        net_g, net_d = self.net_g, self.net_d
        gmod = net_g.module if hasattr(net_g, "module") else net_g
        dmod = net_d.module if hasattr(net_d, "module") else net_d
        y, y_hat, y_mel, y_hat_mel = ctx["y"], ctx["y_hat"], ctx["y_mel"], ctx["y_hat_mel"]
        self.plan_sets.enter("d_gen", dmod.parameters())      # D's weights as its optimizer step just left them
        try:
            with no_param_grads(dmod), S.mma_mode(self.mma):                                              # :198-200
                _, y_d_hat_g, fmap_r, fmap_g = dmod.forward_gen_step(y, y_hat)
        finally:
            self.plan_sets.leave("d_gen")
        loss_mel = A.sum_abs_diff(y_mel, y_hat_mel) / y_mel.numel() * self.c_mel                           # :202
        loss_kl = kl_loss(ctx["z_p"], ctx["logs_q"], ctx["m_p"], ctx["logs_p"], ctx["z_mask"]) * self.c_kl  # :203
        loss_fm = feature_loss(fmap_r, fmap_g)
        loss_gen, _ = generator_loss(y_d_hat_g)
        if gmod.use_automatic_f0_prediction:
            loss_lf0 = A.sum_sq_diff(ctx["pred_lf0"], ctx["lf0"]) / ctx["lf0"].numel()                    # :206
        else:
            loss_lf0 = 0
        loss_gen_all = loss_gen + loss_fm + loss_mel + loss_kl + loss_lf0
        self.optim_g.zero_grad()
        (loss_gen_all * self.scaler.scale if self.scaler is not None else loss_gen_all).backward()
        loss_disc = ctx["loss_disc"]
        out = dict(loss_disc=loss_disc.detach(), loss_gen=loss_gen.detach(), loss_fm=loss_fm.detach(),
                   loss_mel=loss_mel.detach(), loss_kl=loss_kl.detach(),
                   loss_lf0=loss_lf0.detach() if torch.is_tensor(loss_lf0) else loss_lf0,
                   loss_gen_all=loss_gen_all.detach())
        self._guard_losses(out)
        return out

    # -- non-finite guard ------------------------------------------------------------------------------------------------------
    def _guard_losses(self, out):
        """One one-thread kernel per iteration (captured with the rest of it) counts non-finite values among the seven losses
        into sticky device counters; the host looks at them every SVC_TRAIN_FINITE_EVERY iterations (default 200) — the replayed
        iteration never synchronises on a loss, so without this a divergence (or a recurrence of the round-2 replay corruption,
        DESIGN §5 (d)) would run on silently for hours."""
        vals = [v for v in out.values() if torch.is_tensor(v) and v.is_cuda and v.dtype == torch.float32 and v.dim() == 0]
        if not vals:
            return
        if self._guard is None or self._guard.device != vals[0].device:
            self._guard = torch.zeros(3, dtype=torch.int32, device=vals[0].device)
        S.nonfinite_guard(vals[:8], self._guard)

    def check_finite(self, force=False):
        """Raise FloatingPointError if an iteration since the last check produced a non-finite loss.  Called from __call__ every
        `finite_every` iterations (one device read-back); `force` reads now."""
        self._since_check += 1
        if self._guard is None or (not force and self._since_check < self.finite_every):
            return
        self._since_check = 0
        bad, last, n = self._guard.tolist()
        if bad:
            self._guard[:2].zero_()          # (the launch counter keeps running)
            raise FloatingPointError(f"training diverged: {bad} non-finite loss value(s), last in iteration {last} of {n} "
                                     f"(counted on the device by svc_nonfinite_guard_f32; SVC_TRAIN_FINITE_EVERY={self.finite_every})")

    # -- data-parallel hipGraph mode ---------------------------------------------------------------------------------------
    def _reducers(self):
        return getattr(self.net_g, "reducer", None), getattr(self.net_d, "reducer", None)

    def _call_graph_dp(self, items, noise=None):
        """With a process group the iteration is replayed as a SEQUENCE of hipGraphs per optimizer, split where a gradient bucket
        becomes complete, with that bucket's all-reduce issued right behind the graph that completed it:
            D phase: graph[G forward, D forward, D backward up to bucket 0] -> all-reduce(b0) ‖ graph[... up to bucket 1] ->
                     all-reduce(b1) ‖ ... -> wait -> AdamW(D)
            G phase: graph[D forward (frozen), losses, G backward up to bucket 0] -> all-reduce(b0) ‖ ... -> wait -> AdamW(G)
        Each all-reduce runs on RCCL's stream behind an event of the compute stream, which goes straight on to the next graph:
        communication overlaps the rest of that backward pass (north_star's schedule: train.py:57,89-90 = DDP's bucketed
        overlap), and only the LAST bucket's transfer is exposed.  The split points come from the reducer's gradient hooks
        firing DURING capture (`GradReducer.segmented`: the backward pass runs on the capturing thread —
        `torch.autograd.set_multithreading_enabled(False)` — so a hook can end one capture and begin the next); nothing is
        captured of the collectives themselves.  SVC_DP_SPLIT=0 keeps the two monolithic graphs with both reductions between
        them (round-2 form); SVC_DP_CAPTURE_COLLECTIVES=1 records the collectives inside two graphs instead.

        Rank consistency: ranks read different shards, so on one iteration rank A may replay a cached shape while
        rank B meets a new one (warm-up + capture) and rank C is past the graph cap (eager launches).  All three issue the
        same collectives — D's buckets 0..n-1, then G's — because (1) warm-up iterations run under `no_sync` (their results
        are discarded anyway), (2) a capture records no collective and is followed by a replay of what it captured, (3) the
        eager fallback runs its two segments under `no_sync` and reduces with `reduce_all` (same buckets, same order)."""
        red_g, red_d = self._reducers()
        nkeys = sorted(noise) if noise is not None else []
        items = list(items) + [noise[k] for k in nkeys]
        key = ("dp",) + tuple((tuple(t.shape), str(t.dtype)) if t is not None else None for t in items) + tuple(nkeys)
        ent = self._graph_get(key)
        n_in = len(items) - len(nkeys)
        if ent is None and self._graph_cap_reached():
            self.eager_fallbacks += 1
            return self._step_body_dp_eager(items[:n_in], dict(zip(nkeys, items[n_in:])) if nkeys else None)
        if ent is None:
            commons.DEVICE_RNG = True
            static = [t.clone() if t is not None else None for t in items]
            s_items = static[:n_in]
            s_noise = dict(zip(nkeys, static[n_in:])) if nkeys else None
            snaps = (self.optim_g.snapshot(), self.optim_d.snapshot())
            sn = _sn_buffers(self.net_d)
            sn_saved = [b.clone() for b in sn]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), red_g.no_sync(), red_d.no_sync():
                for _ in range(2):                       # warm-up WITHOUT collectives: only this rank may be here (see above)
                    self._step_body(s_items, s_noise)
            torch.cuda.current_stream().wait_stream(side)
            self.optim_g.restore(snaps[0])               # also invalidates every packed-weight cache (version bump)
            self.optim_d.restore(snaps[1])
            for b, v in zip(sn, sn_saved):
                b.copy_(v)
            torch.cuda.synchronize()
            captured = os.environ.get("SVC_DP_CAPTURE_COLLECTIVES", "0") == "1" and red_g.backend == "nccl"
            split = not captured and os.environ.get("SVC_DP_SPLIT", "1") == "1"
            if split:
                prog_d, ctx, pool = self._capture_split(lambda: self._seg_d(s_items, s_noise), red_d, None)
                touched_d = list(self.optim_d.arena.touched)
                # what optim_d.step() does to the host view of the weights: the G segment must re-pack D's weights
                torch.autograd.graph.increment_version(self.optim_d.arena.params)
                prog_g, out, _ = self._capture_split(lambda: self._seg_g(ctx), red_g, pool)
                touched_g = list(self.optim_g.arena.touched)
                torch.autograd.graph.increment_version(self.optim_g.arena.params)
                self.dp_mode = (f"split graphs ({sum(1 for o, _ in prog_d if o == 'graph')} + "
                                f"{sum(1 for o, _ in prog_g if o == 'graph')}), bucket all-reduces overlapped with the backward passes")
            else:
                # SVC_DP_CAPTURE_COLLECTIVES=1 (opt-in, RCCL only): let the autograd hooks fire DURING capture, so the per-bucket
                # all-reduces are recorded on RCCL's stream inside the two graphs.  Not the default: collectives inside hipGraphs
                # have never run on this code base's hardware (single-GPU boxes only), a mis-captured collective hangs instead of
                # raising, and every rank must then capture on the SAME iteration (a capture executes no collective).
                import contextlib
                guard = contextlib.ExitStack()
                if not captured:
                    guard.enter_context(red_g.no_sync())     # hooks must not launch collectives inside a capture
                    guard.enter_context(red_d.no_sync())
                with guard:
                    g1 = torch.cuda.CUDAGraph()
                    with S.graph_capture(g1):
                        ctx = self._seg_d(s_items, s_noise)
                    touched_d = list(self.optim_d.arena.touched)
                    torch.autograd.graph.increment_version(self.optim_d.arena.params)
                    g2 = torch.cuda.CUDAGraph()
                    with S.graph_capture(g2, pool=g1.pool()):
                        out = self._seg_g(ctx)
                    touched_g = list(self.optim_g.arena.touched)
                    torch.autograd.graph.increment_version(self.optim_g.arena.params)
                tail = lambda red: [] if captured else [("reduce", b) for b in range(len(red.buckets))]
                prog_d, prog_g = [("graph", g1)] + tail(red_d), [("graph", g2)] + tail(red_g)
                self.dp_mode = "collectives captured in the graphs (bucket-overlapped)" if captured else \
                    "two graphs, bucketed all-reduce between them (exposed)"
            ent = (prog_d, prog_g, static, out, touched_d, touched_g, ctx)
            self._graph_put(key, ent)
        prog_d, prog_g, static, out, touched_d, touched_g, _ = ent
        self._serialize_replays()
        for s, t in zip(static, items):
            if s is not None:
                s.copy_(t, non_blocking=True)
        try:
            self._run_program(prog_d, red_d)
            self.optim_d.arena.touched = list(touched_d)
            self.optim_d.step()
            self._run_program(prog_g, red_g)
            self.optim_g.arena.touched = list(touched_g)
            self.optim_g.step()
        finally:
            S.wgrad_slab.active = False
        res = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in out.items()}
        self._mark_replay()
        return res

    @staticmethod
    def _run_program(prog, red):
        for op, x in prog:
            if op == "graph":
                x.replay()
            else:
                red.launch_bucket(x)
        red.wait_all()

    def _capture_split(self, fn, red, pool):
        """Capture `fn` (a forward + backward segment) into a sequence of hipGraphs sharing one memory pool, cut wherever
        `red` reports a completed gradient bucket -> ([("graph", g) | ("reduce", bucket), ...], fn's result, pool)."""
        prog = []
        st = dict(g=None, pool=pool)
        mode = S.capture_error_mode()

        def begin():
            g = torch.cuda.CUDAGraph()
            args = () if st["pool"] is None else (st["pool"],)
            g.capture_begin(*args, capture_error_mode=mode)
            st["g"] = g

        def end():
            g, st["g"] = st["g"], None
            g.capture_end()
            prog.append(("graph", g))
            if st["pool"] is None:
                st["pool"] = g.pool()

        def ready(b):
            end()
            prog.append(("reduce", b))
            begin()

        with S.capture_stream(), torch.autograd.set_multithreading_enabled(False), red.segmented(ready):
            begin()
            try:
                out = fn()
            finally:
                if st["g"] is not None:
                    end()
            nb = red.seg_next
        prog += [("reduce", b) for b in range(nb, len(red.buckets))]     # buckets without a (complete set of) gradient(s)
        return prog, out, st["pool"]

    def _step_body_dp_eager(self, items, noise=None):
        """The iteration launched eagerly (past the graph cap) with the SAME collective sequence as the replayed form: both
        segments under no_sync, each followed by the bucket-ordered reduction."""
        red_g, red_d = self._reducers()
        try:
            with red_g.no_sync(), red_d.no_sync():
                ctx = self._seg_d(items, noise)
            red_d.reduce_all()
            # fp16 operands (LossScaler): the reduced gradients are still scaled — unscale, skip on overflow, update, exactly as
            # _step_body does (the mean of the ranks' scaled gradients is inf / nan on every rank alike, so all ranks skip together)
            if self.scaler is None:
                self.optim_d.step()
            else:
                self.scaler.step(self.optim_d)
            with red_g.no_sync(), red_d.no_sync():
                out = self._seg_g(ctx)
            red_g.reduce_all()
            if self.scaler is None:
                self.optim_g.step()
            else:
                self.scaler.step(self.optim_g)
                self.scaler.update()
                out["loss_scale"] = self.scaler.scale
        finally:
            S.wgrad_slab.active = False
        return out

    # -- graph cache: least-recently-used eviction at the cap ---------------------------------------------------------------------
    def _graph_get(self, key):
        ent = self._graphs.get(key)
        if ent is not None:
            self._graphs.move_to_end(key)
        return ent

    def _graph_cap_reached(self):
        """Every captured shape keeps its own ~10 GB tape alive, so the count is bounded (SVC_TRAIN_GRAPH_MAX).  At the cap the
        least recently used graph is dropped (SVC_TRAIN_GRAPH_EVICT=0: launch unseen shapes eagerly from then on); either way the
        event is logged once and counted (`graph_evictions` / `eager_fallbacks`), ADVICE r4 medium."""
        if len(self._graphs) < self.max_graphs:
            return False
        evict = os.environ.get("SVC_TRAIN_GRAPH_EVICT", "1") == "1"
        if not self._cap_logged:
            self._cap_logged = True
            import logging
            logging.getLogger("train").warning(
                "TrainStep: %d batch shapes captured (SVC_TRAIN_GRAPH_MAX) — %s; bucket the loader's frame counts or raise the cap",
                len(self._graphs), "dropping the least recently used graph for each new shape" if evict
                else "unseen shapes launch eagerly from now on")
        if not evict:
            return True
        self._graphs.popitem(last=False)
        self.graph_evictions += 1
        if self.graph_evictions in (10, 100) or self.graph_evictions % 1000 == 0:
            import logging
            logging.getLogger("train").warning(
                "TrainStep: %d graph evictions so far (each costs two warm-up iterations, a capture and the tape's re-allocation): "
                "the data set cycles through more than SVC_TRAIN_GRAPH_MAX=%d padded shapes", self.graph_evictions, self.max_graphs)
        return False

    def _graph_put(self, key, ent):
        self._graphs[key] = ent


def init_distributed(rank, world, device):
    """train.py:57: one process per GPU; backend "nccl" is RCCL on ROCm."""
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)


# ---- the entry point: `python svc_run.py train.py -c configs/config.json -m 44k` (reference train.py:35-340) -----------------
# Same CLI, same logs/<model>/{config.json, train.log, G_<step>.pth, D_<step>.pth} layout, same checkpoint dicts
# (utils.save_checkpoint), same epoch / warm-up / ExponentialLR bookkeeping as the reference's main() / run() /
# train_and_evaluate() / evaluate().  What differs is below the loop: TrainStep (HIP forward + backward, fused AdamW, RCCL
# reducer) instead of DDP + autocast + GradScaler, and every rank reads ITS shard of the file list (the reference gives every
# rank the same batches — SURVEY.md §2a).
global_step = 0


class _ShardSampler(torch.utils.data.Sampler):
    """File order as the reference's loader (shuffle=False, train.py:66-67), rank r taking items r, r + world, ..."""

    def __init__(self, n_items, rank, world):
        from data_parallel import shard_indices
        self.idx = shard_indices(n_items, rank, world, shuffle=False, drop_last=False)

    def __iter__(self):
        return iter(self.idx)

    def __len__(self):
        return len(self.idx)


class _NullWriter:
    """Stands in for torch.utils.tensorboard.SummaryWriter when tensorboard is not installed: the run still logs to train.log."""

    def __getattr__(self, name):
        return lambda *a, **k: None


def _writer(log_dir):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir=log_dir)
    except Exception:      # noqa: BLE001 — tensorboard / protobuf missing or broken: not a reason to stop training
        return _NullWriter()


def _to_device(items, device):
    out = []
    for t in items:
        out.append(t if t is None else (t.to(device, non_blocking=True) if hasattr(t, "to") else t))
    return out


def main():
    """train.py:35-46: one process per GPU of the node."""
    import utils
    assert torch.cuda.is_available(), "CPU training is not allowed."
    hps = utils.get_hparams()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:          # already one process per GPU (torch.distributed.run)
        # RANK = position in the job (process group, data shard, rank-0 duties); LOCAL_RANK = the GPU on this node
        run(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), hps, local_rank=int(os.environ.get("LOCAL_RANK", os.environ["RANK"])))
        return
    n_gpus = torch.cuda.device_count()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(hps.train.port)
    if n_gpus == 1:
        run(0, 1, hps)
    else:
        import torch.multiprocessing as mp
        mp.spawn(run, nprocs=n_gpus, args=(n_gpus, hps,))


def make_loaders(hps, rank, n_gpus, use_graph):
    """train.py:61-72: the training loader (this rank's shard of the file list) and, on rank 0, the validation loader.
    The iteration is replayed from one hipGraph per padded batch shape (SVC_TRAIN_GRAPH=0: eager launches, ~100 ms of host
    time per iteration).  The reference pads a batch to its longest item (data_utils.py:131-185) — a new shape nearly every
    batch — so with graphs on the collate pads to a few bucketed frame counts instead (data_utils.FRAME_BUCKETS; zeros, masked
    by `lengths`).  Workers persist across epochs (the reference re-forks them every epoch)."""
    import multiprocessing
    from data_utils import FRAME_BUCKETS, TextAudioCollate, TextAudioSpeakerLoader
    from torch.utils.data import DataLoader
    collate_fn = TextAudioCollate(buckets=FRAME_BUCKETS, hop_length=hps.data.hop_length) if use_graph else TextAudioCollate()
    all_in_mem = hps.train.all_in_mem
    train_dataset = TextAudioSpeakerLoader(hps.data.training_files, hps, all_in_mem=all_in_mem)
    num_workers = 0 if all_in_mem else (5 if multiprocessing.cpu_count() > 4 else multiprocessing.cpu_count())
    num_workers = int(os.environ.get("SVC_LOADER_WORKERS", num_workers))
    sampler = _ShardSampler(len(train_dataset), rank, n_gpus) if n_gpus > 1 else None
    train_loader = DataLoader(train_dataset, num_workers=num_workers, shuffle=False, sampler=sampler, pin_memory=True,
                              batch_size=hps.train.batch_size, collate_fn=collate_fn, persistent_workers=num_workers > 0)
    eval_loader = None
    if rank == 0:
        eval_dataset = TextAudioSpeakerLoader(hps.data.validation_files, hps, all_in_mem=all_in_mem, vol_aug=False)
        eval_loader = DataLoader(eval_dataset, num_workers=min(1, num_workers), shuffle=False, batch_size=1, pin_memory=False,
                                 drop_last=False, collate_fn=TextAudioCollate())
    return train_loader, eval_loader


def run(rank, n_gpus, hps, local_rank=None):
    """train.py:49-147.  `rank` is the global rank; `local_rank` (default: the same, one node) selects the device."""
    global global_step
    import utils
    logger = writer = writer_eval = None
    if rank == 0:
        logger = utils.get_logger(hps.model_dir)
        logger.info(hps)
        utils.check_git_hash(hps.model_dir)
        writer = _writer(hps.model_dir)
        writer_eval = _writer(os.path.join(hps.model_dir, "eval"))
    device = torch.device("cuda", rank if local_rank is None else local_rank)
    torch.cuda.set_device(device)
    if n_gpus > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", init_method="env://", world_size=n_gpus, rank=rank, device_id=device)
    torch.manual_seed(hps.train.seed)
    use_graph = os.environ.get("SVC_TRAIN_GRAPH", "1") == "1"
    train_loader, eval_loader = make_loaders(hps, rank, n_gpus, use_graph)

    net_g, net_d, optim_g, optim_d = build(hps, device)

    skip_optimizer = False
    try:
        _, _, _, epoch_str = utils.load_checkpoint(utils.latest_checkpoint_path(hps.model_dir, "G_*.pth"), net_g, optim_g,
                                                   skip_optimizer)
        _, _, _, epoch_str = utils.load_checkpoint(utils.latest_checkpoint_path(hps.model_dir, "D_*.pth"), net_d, optim_d,
                                                   skip_optimizer)
        epoch_str = max(epoch_str, 1)
        name = utils.latest_checkpoint_path(hps.model_dir, "D_*.pth")
        global_step = int(name[name.rfind("_") + 1:name.rfind(".")]) + 1
    except Exception:      # noqa: BLE001 — same policy as train.py:106-109: no (readable) checkpoint = fresh start
        print("load old checkpoint failed...")
        epoch_str = 1
        global_step = 0
    for opt in (optim_g, optim_d):
        for group in opt.param_groups:
            group.setdefault("initial_lr", hps.train.learning_rate)
    warmup_epoch = hps.train.warmup_epochs
    scheduler_g = torch.optim.lr_scheduler.ExponentialLR(optim_g, gamma=hps.train.lr_decay, last_epoch=epoch_str - 2)
    scheduler_d = torch.optim.lr_scheduler.ExponentialLR(optim_d, gamma=hps.train.lr_decay, last_epoch=epoch_str - 2)
    step = TrainStep(hps, net_g, net_d, optim_g, optim_d)
    step.enable_graph(use_graph)

    for epoch in range(epoch_str, hps.train.epochs + 1):
        if epoch <= warmup_epoch:                                                     # train.py:126-131
            for opt in (optim_g, optim_d):
                for group in opt.param_groups:
                    group["lr"] = hps.train.learning_rate / warmup_epoch * epoch
        train_and_evaluate(rank, epoch, hps, step, [train_loader, eval_loader], logger, [writer, writer_eval], device)
        scheduler_g.step()
        scheduler_d.step()


def train_and_evaluate(rank, epoch, hps, step, loaders, logger, writers, device):
    """train.py:136-275: one epoch; logging / evaluation / checkpoints on rank 0 at the reference's intervals."""
    global global_step
    import time
    import utils
    train_loader, eval_loader = loaders
    writer, writer_eval = writers
    net_g, net_d, optim_g, optim_d = step.net_g, step.net_d, step.optim_g, step.optim_d
    net_g.train()
    net_d.train()
    t_epoch = time.time()
    for batch_idx, items in enumerate(train_loader):
        out = step(_to_device(items, device))
        if global_step % hps.train.eval_interval == 0:
            # every rank, same global_step, BEFORE rank 0 evaluates / writes G_<step>.pth: the periodic guard read counts calls
            # since construction (not global_step), so after a resume a checkpoint could otherwise be written up to
            # finite_every - 1 iterations into a divergence, and `keep_ckpts` rotation could delete the last good one; and all
            # ranks raise on the same iteration instead of leaving peers blocked in a collective
            step.check_finite(force=True)
        if rank == 0:
            if global_step % hps.train.log_interval == 0:
                lr = optim_g.param_groups[0]["lr"]
                losses = [out["loss_disc"], out["loss_gen"], out["loss_fm"], out["loss_mel"], out["loss_kl"]]
                vals = [float(x) for x in losses]
                logger.info("Train Epoch: {} [{:.0f}%]".format(epoch, 100. * batch_idx / len(train_loader)))
                logger.info(f"Losses: {vals}, step: {global_step}, lr: {lr}, reference_loss: {sum(vals)}")
                scalars = {"loss/g/total": float(out["loss_gen_all"]), "loss/d/total": vals[0], "learning_rate": lr,
                           "loss/g/fm": vals[2], "loss/g/mel": vals[3], "loss/g/kl": vals[4], "loss/g/lf0": float(out["loss_lf0"])}
                utils.summarize(writer=writer, global_step=global_step, scalars=scalars)
            if global_step % hps.train.eval_interval == 0:
                evaluate(hps, net_g, eval_loader, writer_eval, device)
                utils.save_checkpoint(net_g, optim_g, hps.train.learning_rate, epoch,
                                      os.path.join(hps.model_dir, "G_{}.pth".format(global_step)))
                utils.save_checkpoint(net_d, optim_d, hps.train.learning_rate, epoch,
                                      os.path.join(hps.model_dir, "D_{}.pth".format(global_step)))
                keep_ckpts = getattr(hps.train, "keep_ckpts", 0)
                if keep_ckpts > 0:
                    utils.clean_checkpoints(path_to_models=hps.model_dir, n_ckpts_to_keep=keep_ckpts, sort_by_time=True)
        global_step += 1
    if rank == 0:
        logger.info(f"====> Epoch: {epoch}, cost {format(time.time() - t_epoch, '.2f')} s")


def evaluate(hps, generator, eval_loader, writer_eval, device):
    """train.py:278-335: the first item of every validation batch through SynthesizerTrn.infer; audio and the mel L1 distance
    go to the eval writer."""
    import utils
    gen = generator.module if hasattr(generator, "module") else generator
    gen.eval()
    d = hps.data
    audio, dist_sum, n = {}, 0.0, 0
    with torch.no_grad():
        for batch_idx, items in enumerate(eval_loader):
            c, f0, spec, y, spk, _, uv, volume = _to_device(items, device)
            vol = volume[:1] if volume is not None else None
            y_hat, _ = gen.infer(c[:1], f0[:1], uv[:1], g=spk[:1], vol=vol)
            y_hat_mel = mel_spectrogram_torch(y_hat.squeeze(1).float(), d.filter_length, d.n_mel_channels, d.sampling_rate,
                                              d.hop_length, d.win_length, d.mel_fmin, d.mel_fmax)
            y_mel = mel_spectrogram_torch(y[:1].squeeze(1).float(), d.filter_length, d.n_mel_channels, d.sampling_rate,
                                          d.hop_length, d.win_length, d.mel_fmin, d.mel_fmax)
            T = min(y_mel.shape[-1], y_hat_mel.shape[-1])
            dist_sum += float((y_mel[..., :T] - y_hat_mel[..., :T]).abs().mean())
            n += 1
            audio.update({f"gen/audio_{batch_idx}": y_hat[0], f"gt/audio_{batch_idx}": y[0]})
    utils.summarize(writer=writer_eval, global_step=global_step, audios=audio, audio_sampling_rate=d.sampling_rate,
                    scalars={"eval/mel_l1": dist_sum / max(n, 1)})
    gen.train()


if __name__ == "__main__":
    main()
