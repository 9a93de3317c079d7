#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X so-vits-svc engine (contract: see the task brief / DESIGN.md §4).

Metric (BASELINE.json): 44.1 kHz audio samples/sec (whole job), inference.
Workload at N=1 (BASELINE.json configs[1]): configs_template/config_template.json model, single speaker id,
ContentVec-768 units, NSF-HiFiGAN decoder, one 10.01 s clip (T = 862 frames -> 441,344 samples), B = 1, fp32,
synthetic inputs (c ~ N(0,1), f0 ~ U(100,400) with 10 % unvoiced runs) and seeded random weights (no checkpoint
exists in the reference tree).  A "step" = one SynthesizerTrn.infer call on that clip with the inputs already
resident in HBM; the RNG draws (3 torch.randn/rand fills) are inside the step, as in the reference.

N > 1: utterances are independent ("replicas only"): every rank runs the same step on its own clip,
no data-path collective; value = N * samples * steps / max-over-ranks time (weak scaling).

Training (BASELINE.json: "train steps/sec at 1/2/4/8 MI355X", configs[2]): `--mode train` makes the step one
iteration of train.py:150-213 (D step + G step, fp32, FusedAdamW) on B=16 items per GPU, T~U{300..790} frames padded
to the max, segment_size 8192, full template model + MultiPeriodDiscriminator; N>1 = minibatch sharded data-parallel
(global batch 16*N, weak scaling) with the bucketed RCCL gradient all-reduce overlapped with backward.  The default
mode ("both") prints the inference line with the training result attached under "train".

Extra objects:
  roofline     — dominant kernel family (conv1d_mfma, the fused fp32-MFMA conv that carries the MRF ResBlocks):
                 algorithmic FLOP per launch / mean launch duration, durations from hipEvents recorded around every
                 launch by libsvc_hip's profiler in a separate EAGER pass over the same steps (events cannot be
                 recorded inside a replayed hipGraph); peak = 157.3 TFLOP/s fp32 matrix (MI355X_MICROARCH.md).
  cpu_baseline — the CPU oracle (a torch-CPU fp32 restatement of the reference modules, validated against the real
                 reference in tests/golden) timed on this host on the same clip: kind "port".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3
T_FRAMES = 862          # 10.01 s at hop 512 / 44.1 kHz
HOP = 512


def build_model(dev):
    import models
    import synthetic_data as W        # deterministic synthetic checkpoint + inputs (data generator, not the oracle)
    cfg = W.full_config()
    kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
    net = models.SynthesizerTrn(cfg["spec_channels"], cfg["segment_size"], **kw)
    net.load_state_dict(W.make_state_dict(cfg, 1234))
    net = net.to(dev).eval()
    return net, cfg, W


def cpu_baseline(cfg, W, inputs, max_seconds=45.0, runs=5):
    """Time the CPU oracle on the same 10 s clip (bounded: per thread count 1 warm-up + up to `runs` timed runs inside
    ~max_seconds of host time in all; the sample string says what was taken).  torch's intra-op pool over-subscribes these
    convolutions on a many-core host (128 threads measured SLOWER than 8 here), so a few thread counts are tried and the best
    median is reported with the thread count that gave it (`cores`).  The unmodified reference cannot be timed on the GPU box
    (no /root/reference there): kind = "port"; `reference` carries the build container's timing of the REAL modules next to
    the port on the same cores (scripts/time_reference_cpu.py), labelled with the machine it comes from."""
    from oracle import svc_oracle as O
    c, f0, uv, sid = inputs
    sd = W.make_state_dict(cfg, 1234)
    noise = W.make_noise(cfg, c.shape[0], c.shape[2], seed=99)
    n = c.shape[0] * c.shape[2] * HOP
    all_threads = torch.get_num_threads()
    cands = sorted({t for t in (all_threads, 64, 32, 16) if 1 <= t <= all_threads}, reverse=True)
    t_start, tried, best = time.perf_counter(), {}, None
    with torch.no_grad():
        for nt in cands:
            if tried and time.perf_counter() - t_start > 0.75 * max_seconds:
                break
            torch.set_num_threads(nt)
            O.synth_infer(sd, cfg, c, f0, uv, sid, noise, noice_scale=0.4)      # warm-up (oneDNN primitive cache)
            times = []
            budget = max_seconds / len(cands)
            while len(times) < runs and (not times or sum(times) + times[-1] < budget):
                t0 = time.perf_counter()
                O.synth_infer(sd, cfg, c, f0, uv, sid, noise, noice_scale=0.4)
                times.append(time.perf_counter() - t0)
            med = sorted(times)[len(times) // 2]
            tried[nt] = (round(med, 3), len(times))
            if best is None or med < best[0]:
                best = (med, nt, len(times))
    torch.set_num_threads(all_threads)
    med, nt, nruns = best
    out = dict(value=n / med, unit="samples/s", cores=nt, kind="port",
               sample=f"oracle.synth_infer on the same {n}-sample clip, median of {nruns} runs after 1 warm-up at {nt} threads "
                      f"({med:.2f} s/clip, RTF {med / (n / 44100):.3f}); thread counts tried (median s, runs): {tried}")
    ref = os.path.join(ROOT, "profiles", "cpu_reference_build_container.json")
    if os.path.exists(ref):
        try:
            rj = json.load(open(ref))
            out["reference"] = rj
            # the label the port's number needs: how the port compares with the UNMODIFIED reference where both could be timed
            # (same machine, same cores, same clip) — and what this box's figure becomes under that ratio
            ratio = rj["port_on_same_cores"]["value"] / rj["value"]
            out["port_vs_reference"] = dict(
                ratio=round(ratio, 3), reference_estimate_on_this_box=out["value"] / ratio,
                note=f"port throughput / reference throughput = {ratio:.3f}, both measured in the build container on {rj['cores']} cores "
                     "(scripts/time_reference_cpu.py -> profiles/cpu_reference_build_container.json); the estimate divides this box's port "
                     "figure by it — the reference itself cannot run here (no /root/reference on the GPU box)")
        except Exception:      # noqa: BLE001
            pass
    return out


TRAIN_B = 16
TRAIN_SEG = 8192


def train_hps(cfg, bf16=False):
    """bf16: False (fp32), True / "bf16" (fp16_run + half_type bf16) or "fp16" (fp16_run + half_type fp16)."""
    model = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
    amp = bool(bf16)
    tr = dict(segment_size=TRAIN_SEG, learning_rate=1e-4, betas=[0.8, 0.99], eps=1e-9, c_mel=45, c_kl=1.0,
              fp16_run=amp, half_type=("fp16" if bf16 == "fp16" else "bf16") if amp else "fp16", batch_size=TRAIN_B)
    return dict(data=dict(filter_length=2048, hop_length=HOP, win_length=2048, n_mel_channels=80, sampling_rate=44100,
                          mel_fmin=0.0, mel_fmax=22050), train=tr, model=model)


def make_train_items(cfg, B, seed):
    """T~U{300..790} padded to max, spec=|N(0,1)|, y~U(-0.5,0.5), 4 speakers."""
    g = torch.Generator().manual_seed(seed)
    lengths = torch.randint(300, 791, (B,), generator=g)
    T = int(lengths.max())
    c = torch.randn(B, cfg["ssl_dim"], T, generator=g)
    f0 = 100 + 300 * torch.rand(B, T, generator=g)
    for b in range(B):
        for s0 in torch.randint(0, T - 8, (max(1, T // 80),), generator=g).tolist():
            f0[b, s0:s0 + 8] = 0
    uv = (f0 > 0).float()
    spec = torch.randn(B, cfg["spec_channels"], T, generator=g).abs()
    y = (torch.rand(B, 1, T * HOP, generator=g) - 0.5)
    spk = torch.randint(0, 4, (B, 1), generator=g)
    return (c, f0, spec, y, spk, lengths, uv, None), T


def cpu_baseline_train(cfg, hps, items_cpu, max_items=4):
    """One iteration of the CPU oracle's training loop (oracle.train_oracle.gan_train_loop: torch-CPU autograd + AdamW in
    the reference's order) on a BOUNDED sample of the workload: the first `max_items` items of the same minibatch (the
    full B=16 iteration is ~4 TFLOP — minutes on host cores).  steps/s is reported for the full batch by scaling with
    items (work is linear in the batch)."""
    import synthetic_data as W
    from oracle import mel as OM
    from oracle import train_oracle as TO
    c, f0, spec, y, spk, lengths, uv, _ = [t[:max_items] if t is not None else None for t in items_cpu]
    T = int(lengths.max())
    c, f0, spec, uv, y = c[:, :, :T], f0[:, :T], spec[:, :, :T], uv[:, :T], y[:, :, :T * HOP]
    d = hps["data"]
    data = dict(n_fft=d["filter_length"], hop=d["hop_length"], win=d["win_length"], n_mels=d["n_mel_channels"],
                sr=d["sampling_rate"], fmin=d["mel_fmin"], fmax=d["mel_fmax"])
    ocfg = dict(cfg)                      # p_dropout 0.1 active, like the timed HIP iteration
    sd_g = W.make_train_state_dict(cfg, 1234)
    sd_d = W.make_mpd_state_dict(1235)
    noise = W.make_train_noise(ocfg, max_items, T, lengths, 7, hop=HOP)
    noise["dropout_u"] = W.make_dropout_draws(ocfg, max_items, T, 8)
    mb = torch.from_numpy(OM.mel_filterbank(data["sr"], data["n_fft"], data["n_mels"], data["fmin"], data["fmax"]))
    t0 = time.perf_counter()
    TO.gan_train_loop(sd_g, sd_d, ocfg, data, (c, f0, uv, spec, y, spk, lengths), noise, mb, 1)
    cold = time.perf_counter() - t0
    t0 = time.perf_counter()          # second iteration: oneDNN primitive caches / allocator pools warm (bounded: ~2 x 15 s)
    TO.gan_train_loop(sd_g, sd_d, ocfg, data, (c, f0, uv, spec, y, spk, lengths), noise, mb, 1)
    dt = time.perf_counter() - t0
    return dict(value=(max_items / TRAIN_B) / dt, unit="steps/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle.gan_train_loop, warm iteration on the first {max_items} of the {TRAIN_B} items "
                       f"(T={T} frames): {dt:.1f} s (cold first iteration {cold:.1f} s); steps/s scaled by {max_items}/{TRAIN_B} "
                       "(work is linear in the batch)")


PEAK_BF16_MFMA_TFLOPS = 2500.0


def run_train(args, dev, rank, world, dist, bf16=False):
    """Time K training iterations; returns the result dict (rank 0) or None.  bf16: the reference's `fp16_run: true,
    half_type: bf16` configuration (bf16 matrix operands inside the autocast regions) — reported under its own key, never as
    the headline."""
    import svc_hip as S
    import synthetic_data as W
    import train as TR
    cfg = W.full_config()
    if os.environ.get("SVC_BENCH_PDROP") is not None:      # determinism experiments only
        cfg["p_dropout"] = float(os.environ["SVC_BENCH_PDROP"])
    hps = train_hps(cfg, bf16=bf16)
    torch.manual_seed(1234)
    net_g, net_d, optim_g, optim_d = TR.build(hps, dev)
    net_g.module.load_state_dict(W.make_train_state_dict(cfg, 1234))
    net_d.module.load_state_dict(W.make_mpd_state_dict(1235))
    net_g.train()
    net_d.train()
    step_fn = TR.TrainStep(hps, net_g, net_d, optim_g, optim_d)
    # N = 1: the whole iteration is ONE hipGraph.  N > 1: each phase (D, G) is a sequence of hipGraphs cut at the gradient-bucket
    # boundaries of its backward pass, a bucket's all-reduce issued behind the graph that completed it and overlapped with the
    # next graph (train.TrainStep._call_graph_dp; SVC_DP_SPLIT=0: two monolithic graphs, reductions between them);
    # SVC_TRAIN_GRAPH=0 selects the eager, hook-driven bucket-overlapped path instead.
    use_graph = (not args.no_graph) and os.environ.get("SVC_TRAIN_GRAPH", "1") != "0"
    step_fn.enable_graph(use_graph)
    for net in (net_g, net_d):                     # exposed-communication timing is opt-in (two hipEvents per wait)
        if getattr(net, "reducer", None) is not None:
            net.reducer.time_exposed = True
    items_cpu, T = make_train_items(cfg, TRAIN_B, 4321 + rank)
    items = tuple(t.to(dev) if t is not None else None for t in items_cpu)
    torch.manual_seed(99 + rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    steps, warm = args.train_steps, args.train_warmup
    last = None
    if use_graph and dist is not None:
        try:                                       # first call = warm-up + capture; every rank takes the same branch
            last = step_fn(items)
        except Exception as e:                     # noqa: BLE001 — e.g. a collective runtime that cannot coexist with capture
            print(f"bench.py: data-parallel hipGraph capture failed ({type(e).__name__}: {e}); falling back to eager launches",
                  file=sys.stderr)
            use_graph = False
            step_fn.enable_graph(False)
            step_fn.dp_ordered = True              # keep issuing the bucket-ordered collectives the other ranks' replays issue
            # a capture issues no collective before its first replay, so the failed call left this rank one iteration behind its
            # peers (who are in, or past, D's buckets 0..n-1 then G's): run that iteration now in the same collective order
            last = step_fn(items)
    trace = [] if os.environ.get("SVC_BENCH_TRACE") else None      # determinism experiments: per-iteration losses
    for _ in range(warm):
        last = step_fn(items)
        if trace is not None:
            trace.append(last)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step_fn(items)
        if trace is not None:
            trace.append(last)
    barrier()
    elapsed = time.perf_counter() - t0
    if trace is not None and rank == 0:
        print("TRACE " + " ".join(f"{float(l['loss_disc']):.4f}/{float(l['loss_kl']):.3f}/{float(l['loss_mel']):.3f}" for l in trace),
              file=sys.stderr)
    per_rank_ms = None
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        parts = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(parts, tt)                 # every rank's own clock over the same K iterations (value uses the MAX)
        per_rank_ms = [round(p.item() * 1e3 / steps, 3) for p in parts]
        elapsed = max(p.item() for p in parts)
    if rank != 0:
        return None
    fams = roof = None
    if not args.no_roofline:
        import contextlib
        step_fn.enable_graph(False)
        S.prof_enable(True)
        S.prof_reset()
        # rank 0 only: this extra (untimed) iteration must not communicate — the other ranks have already left
        with contextlib.ExitStack() as es:
            for net in (net_g, net_d):
                if getattr(net, "reducer", None) is not None:
                    es.enter_context(net.reducer.no_sync())
            step_fn(items)
        torch.cuda.synchronize()
        rep = S.prof_report()
        S.prof_enable(False)
        tot = sum(v["ms"] for v in rep.values())
        fams = {k: dict(ms_per_step=round(v["ms"], 3), calls=v["calls"],
                        tflops=round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else 0.0)
                for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])}
        fams["_kernel_ms_total"] = round(tot, 3)
        # the training step's roofline: every MFMA family of the iteration (forward + dgrad convs, weight gradients, attention
        # products) against the fp32-MFMA peak — algorithmic FLOP of their launches / the sum of their hipEvent durations
        mf = {k: v for k, v in rep.items() if v["flop"] > 0 and v["ms"] > 0}
        mflop, mms = sum(v["flop"] for v in mf.values()), sum(v["ms"] for v in mf.values())
        dom = max(mf.items(), key=lambda kv: kv[1]["ms"]) if mf else None
        roof = dict(bound="mfma", unit="TFLOP/s", peak=PEAK_FP32_MFMA_TFLOPS,
                    kernel="+".join(sorted(mf)), achieved=round(mflop / (mms * 1e-3) / 1e12, 2) if mms else 0.0,
                    frac=round(mflop / (mms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if mms else 0.0,
                    flop_per_step=mflop, mfma_kernel_ms_per_step=round(mms, 3), launches_per_step=int(sum(v["calls"] for v in rep.values())),
                    dominant=dict(kernel=dom[0], ms_per_step=round(dom[1]["ms"], 3),
                                  tflops=round(dom[1]["flop"] / (dom[1]["ms"] * 1e-3) / 1e12, 2)) if dom else None,
                    whole_step=dict(tflops=round(mflop / (1e-3 * 1e3 * elapsed / steps) / 1e12, 2),
                                    frac=round(mflop / (elapsed / steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                    note="the same FLOP over ms_per_step (hipGraph replay): element-wise / copy launches and gaps included"),
                    traffic=None, note="per-launch hipEvent durations from one eager iteration after the timed region; launches_per_step "
                                       "counts the library's kernels only (torch element-wise / copy launches are in the rocprof summary)")
        # frames padded to T count as work in flop_per_step: say how much of it the items' own frames are.  The frame-proportional
        # part of the iteration (pre + prior encoder + F0 decoder + posterior encoder + flow: SURVEY §8a, 402 GFLOP forward at
        # B 16 x T 400 = 62.8 MFLOP per item-frame, x3 with the two backward products) shrinks with the masked frames; the
        # generator's 8192-sample segments and the discriminators do not depend on T.
        lens = items_cpu[5].float()
        frac = float(lens.mean()) / T
        fp = 3 * 62.8e6 * TRAIN_B * T
        roof["padding"] = dict(padded_frames=T, mean_item_frames=round(float(lens.mean()), 1), item_frame_fraction=round(frac, 4),
                               frame_proportional_flop=fp, flop_per_step_item_frames_only=mflop - fp * (1 - frac),
                               frac_item_frames_only=round((mflop - fp * (1 - frac)) / (mms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if mms else 0.0,
                               note="flop_per_step counts frames padded to T as work; the *_item_frames_only figures remove the padded share of "
                                    "the frame-proportional networks (estimate from SURVEY §8a's per-network FLOP)")
        if not args.no_pmc and world == 1 and bf16 in (False, True, "bf16") and os.environ.get("SVC_BENCH_PMC", "1") != "0":
            torch.cuda.synchronize()
            live = collect_pmc_traffic(timeout_s=300, mode="train", extra_args=("--bf16",) if bf16 else ())
            if live is not None:
                roof["traffic"], roof["traffic_all_kernels"], roof["traffic_source"] = live
    red = None
    if getattr(net_g, "reducer", None) is not None:
        rg, rd = net_g.reducer, net_d.reducer
        # every iteration reduces each network's gradients once: per-iteration figures = totals / number of reductions
        # (reduce_all calls between graph replays, or hooked backward passes in the eager mode; warm-up iterations included)
        def n_iter(r):      # one reduction of a network's gradients per iteration, whichever of the three forms issued it
            return max(r.stats["wait_all_calls"] + r.stats["reduce_all_calls"] + r.stats["backward_passes"], 1)

        def per_it(r, key):
            return r.stats[key] / n_iter(r)
        n_red = n_iter(rg)
        red = dict(backend=rg.backend, ranks=world,
                   mode=getattr(step_fn, "dp_mode", "eager launches, per-bucket all-reduce overlapped with backward (autograd hooks)"),
                   bytes_per_iter=per_it(rg, "reduced_bytes") + per_it(rd, "reduced_bytes"),
                   launches_per_iter=per_it(rg, "launches") + per_it(rd, "launches"),
                   buckets=dict(g=len(rg.buckets), d=len(rd.buckets)),
                   exposed_ms=(rg.exposed_ms() + rd.exposed_ms()) / n_red, bytes=per_it(rg, "reduced_bytes") + per_it(rd, "reduced_bytes"),
                   launches=per_it(rg, "launches") + per_it(rd, "launches"),
                   note="per iteration; exposed_ms = time the compute stream stalled on all-reduces (hipEvents around the waits)")
        red["exposed_ms_per_iter"] = red["exposed_ms"]
    cpu = None
    if world == 1 and not args.no_cpu_baseline and not bf16:
        cpu = cpu_baseline_train(cfg, hps, items_cpu)
    if bf16 and roof is not None:
        # the step mixes bf16-operand launches (the LDS-DMA tilings of the batched convolutions, the 128 x 64 weight-gradient
        # kernel) with fp32 ones (unaligned / narrow shapes, attention products): the fraction is quoted against the bf16 peak
        roof["peak"] = PEAK_BF16_MFMA_TFLOPS
        roof["frac"] = round(roof["achieved"] / PEAK_BF16_MFMA_TFLOPS, 4)
        roof["whole_step"]["frac"] = round(roof["whole_step"]["tflops"] / PEAK_BF16_MFMA_TFLOPS, 4)
        roof["note"] += "; bf16-operand and fp32 launches share these family rows, peak = dense bf16 MFMA"
        roof["bf16_launches"] = dict(conv=S.lib().svc_debug_bf16(-1), wgrad=S.tlib().svc_debug_wgrad_bf16_launches())
    if bf16:
        h = "fp16" if bf16 == "fp16" else "bf16"
        tag, dt, cfgtag = f", fp16_run + half_type {h}", f"{h} matrix operands, f32 accumulate / storage", "fp16_run half_type=" + h
    else:
        tag, dt, cfgtag = "", "f32", "fp32"
    return dict(metric="train steps/sec (train.py D+G iteration)" + tag,
                value=steps / elapsed, unit="steps/s",
                ms_per_step=1e3 * elapsed / steps, steps=steps, warmup=warm, n_gpus=world, scaling="weak",
                dtype=dt,
                items_per_s=world * TRAIN_B * steps / elapsed,
                config=dict(workload="BASELINE configs[2]: config_template.json model + MultiPeriodDiscriminator, "
                                     f"batch_size={TRAIN_B} per GPU, segment_size={TRAIN_SEG}, T padded to {T} frames, "
                                     f"4 speakers, {cfgtag}, FusedAdamW(lr 1e-4, betas (0.8,0.99), eps 1e-9)",
                            global_batch=TRAIN_B * world, frames=T,
                            launch="eager (fp16: the GradScaler rule decides every optimizer step on the host)" if bf16 == "fp16" else
                            (("hipGraph replay of the whole iteration" if world == 1 else
                              "hipGraphs cut at the gradient-bucket boundaries, bucket all-reduces overlapped with the backward passes, AdamW per phase") if use_graph else "eager"),
                            parallelism=f"dp{world} (sharded minibatch, bucketed RCCL all-reduce)" if world > 1 else "single GPU",
                            p_dropout=cfg["p_dropout"]),
                losses={k: round(float(v), 4) for k, v in last.items()},
                roofline=roof if fams is not None else None, families=fams, allreduce=red, per_rank_ms_per_step=per_rank_ms,
                cpu_baseline=cpu)


def collect_pmc_traffic(timeout_s=200, mode="infer", extra_args=(), fam_ok=None):
    """roofline.traffic, collected IN this run: two separate `rocprofv3 --pmc` passes (FETCH_SIZE, then WRITE_SIZE — the TCC block
    cannot hold both, MI355X_MICROARCH.md counter table) around a short eager infer of this same script (one stream, launches back to
    back), summarised per kernel family by scripts/pmc_summary.py: read = FETCH_SIZE KiB x 1024 x 2 (gfx950 tallies 128-byte
    requests at 64 B), write = WRITE_SIZE KiB x 1024 (uncalibrated), Infinity-Cache hits included -> an upper bound on HBM bytes.
    Returns (bytes per launch of the conv1d_mfma family, bytes per clip, note) or None when rocprofv3 is absent / a pass fails."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import pmc_summary as PS
    steps, warm = (3, 1) if mode == "infer" else (1, 1)
    env = dict(os.environ, SVC_MRF_STREAMS="0", TMPDIR="/tmp")
    per = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(td, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "run", "--", sys.executable, os.path.abspath(__file__),
                   "--mode", mode, "--steps", str(steps), "--warmup", str(warm), "--no-cpu-baseline", "--no-graph", "--no-roofline",
                   "--no-extras", "--no-host-io", "--no-steady", "--no-pmc"] + list(extra_args)
            try:
                r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=timeout_s)
            except Exception:      # noqa: BLE001 — a hung / missing profiler must not take the bench line down
                return None
            if r.returncode != 0:
                return None
            per[counter], _ = PS.load(d, counter)
    if mode == "train":
        # one iteration = every launch of it: MFMA families (dense convs forward + dgrad, weight gradients, attention GEMMs) and all
        mf = lambda k: PS.family(k) in ("conv1d_mfma", "conv1d_wgrad", "conv1d_wgrad_small") or PS.family(k).startswith("gemm_f32")
        it = steps + warm
        rdm = sum(v for k, (v, n) in per["FETCH_SIZE"].items() if mf(k)) * 1024 * 2
        wrm = sum(v for k, (v, n) in per["WRITE_SIZE"].items() if mf(k)) * 1024
        total = sum(v for v, _ in per["FETCH_SIZE"].values()) * 1024 * 2 + sum(v for v, _ in per["WRITE_SIZE"].values()) * 1024
        nl = sum(n for k, (v, n) in per["FETCH_SIZE"].items() if mf(k))
        if nl == 0:
            return None
        return (rdm + wrm) / it, total / it, (
            f"collected in this run: two separate rocprofv3 --pmc passes (FETCH_SIZE x1024 x2 for gfx950's 128-byte requests, WRITE_SIZE "
            f"x1024 uncalibrated; Infinity-Cache hits included) over {it} eager iterations of this script; traffic = bytes per iteration of "
            f"the MFMA families ({nl // it} launches), {total / it / 1e9:.1f} GB per iteration over all kernels")
    if fam_ok is None:
        fam_ok = lambda f: f == "conv1d_mfma"
    rd = sum(v for k, (v, n) in per["FETCH_SIZE"].items() if fam_ok(PS.family(k))) * 1024 * 2
    wr = sum(v for k, (v, n) in per["WRITE_SIZE"].items() if fam_ok(PS.family(k))) * 1024
    n = sum(n for k, (v, n) in per["FETCH_SIZE"].items() if fam_ok(PS.family(k)))
    if n == 0:
        return None
    total = sum(v for v, _ in per["FETCH_SIZE"].values()) * 1024 * 2 + sum(v for v, _ in per["WRITE_SIZE"].values()) * 1024
    return (rd + wr) / n, total / (steps + warm), (
        f"collected in this run: two separate rocprofv3 --pmc passes (FETCH_SIZE x1024 x2 for gfx950's 128-byte requests, WRITE_SIZE "
        f"x1024 uncalibrated; Infinity-Cache hits included) over a {steps + warm}-clip eager infer of this script, {n} conv launches; "
        f"{total / (steps + warm) / 1e9:.2f} GB per clip over all kernels")


def run_inflight(args, dev):
    """`--mode inflight` (child of the default run): N independent B = 1 clips per hipGraph replay, fp32 (N = 2, 4) and the
    half-precision and split modes (N = 4); prints one JSON object."""
    import bench_extra as X
    net, cfg, W = build_model(dev)
    c, f0, uv, sid = [t.to(dev) for t in W.make_inputs(cfg, 1, T_FRAMES, seed=1234)]
    samples = T_FRAMES * HOP
    out = dict(note="N independent B = 1 clips (own inputs / noise / outputs) replayed as parallel branches of one hipGraph "
                    "(SynthesizerTrn.infer_many): throughput of a chunk stream or a request queue — one clip alone leaves most CUs "
                    "idle through its encoder + flow section; every clip bit-identical to its single replay; per-clip LATENCY is the "
                    "headline's ms_per_step")

    def leg(tag, counts):
        net.enable_graph(True)
        one, _ = net.infer(c, f0, uv, g=sid, noice_scale=0.4)
        for nfl in counts:
            many = lambda: net.infer_many([(c, f0, uv, sid)] * nfl, noice_scale=0.4)
            same = all(torch.equal(o, one) for o, _ in many())
            dt = X._timeit(many, max(args.steps // nfl, 4), warm=3) / nfl
            out[f"{tag}{nfl}"] = dict(clips_in_flight=nfl, ms_per_clip=round(1e3 * dt, 4), samples_per_s=samples / dt,
                                      outputs_equal_single_clip=bool(same))
        net.enable_graph(False)
    leg("f32_", [int(v) for v in os.environ.get("SVC_BENCH_IN_FLIGHT", "2,4").split(",")])
    net.half()
    leg("half_", [4])
    net.float()
    net.split_f16()
    leg("split_", [4])
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--mode", choices=["infer", "train", "both", "inflight"], default="both")
    ap.add_argument("--train-steps", type=int, default=None)
    ap.add_argument("--train-warmup", type=int, default=None)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--bf16", action="store_true", help="--mode train only: the fp16_run + half_type bf16 configuration")
    ap.add_argument("--split", action="store_true", help="--mode infer only: time SynthesizerTrn.split_f16() (the generator on the split pipeline: hi + lo "
                    "fp16 planes, three fp16 MFMA per product, fp32-level output) instead of the fp32-MFMA path; labelled as such — profiling aid, not the headline")
    ap.add_argument("--half", action="store_true", help="--mode infer only: time SynthesizerTrn.half() (the reference's half-precision mode; never the headline: "
                    "the default run reports it under infer_half)")
    ap.add_argument("--fp16", action="store_true", help="--mode train only: fp16_run + half_type fp16 (GradScaler rule: eager launches)")
    ap.add_argument("--no-host-io", action="store_true", help="skip the PCIe-inclusive pass (profiling runs: keeps the step count exact)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra objects (device, e2e, snake_b8, diffusion_*)")
    ap.add_argument("--no-pmc", action="store_true", help="do not re-run a short infer under rocprofv3 --pmc for roofline.traffic (the stamped profiles/pmc_conv1d_mfma.json is quoted when it matches the kernel sources)")
    ap.add_argument("--no-steady", action="store_true", help="skip the 200-replay steady_state look (profiling runs: keeps the step count exact)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher — one rank per GPU under torch.distributed.run, exactly the
        # command line the driver uses; rank 0's JSON line is the child's stdout
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the MI355X engine has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    # one rank per GPU; SVC_DIST_BACKEND=gloo + fewer GPUs than ranks (a 1-GPU box) folds the ranks onto the available
    # devices — a functional check of the multi-rank path, not a scaling measurement (RCCL refuses two ranks per device)
    backend = os.environ.get("SVC_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if os.environ.get("SVC_CONV_CFG"):           # tuning aid (A/B of kernel variants on one box): svc_debug_set_conv_cfg code
        import svc_hip
        svc_hip.tlib().svc_debug_set_conv_cfg(int(os.environ["SVC_CONV_CFG"]))
    dist = None
    # SVC_DP_FORCE=1 at N=1: initialise the process group anyway so the data-parallel code path (reducer, two-graph iteration,
    # RCCL all-reduces between the replays) runs against the real collective library on a single GPU — functional dry run only
    if world > 1 or os.environ.get("SVC_DP_FORCE", "0") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.mode == "inflight":
        run_inflight(args, dev)
        return
    if args.mode == "train":
        args.train_steps = args.train_steps or args.steps
        args.train_warmup = args.warmup if args.train_warmup is None else args.train_warmup
        res = run_train(args, dev, rank, world, dist, bf16="fp16" if args.fp16 else args.bf16)
        if rank == 0:
            res.update(higher_is_better=True, vs_baseline=None, data="synthetic")
            print(json.dumps(res))
        if dist is not None:
            dist.destroy_process_group()
        return
    args.train_steps = args.train_steps or min(args.steps, 8)
    args.train_warmup = 2 if args.train_warmup is None else args.train_warmup

    import svc_hip as S
    net, cfg, W = build_model(dev)
    if args.half:
        net.half()
        args.no_extras = True
    if args.split:
        net.split_f16()
        args.no_extras = True

    B = args.batch
    cpu_in = W.make_inputs(cfg, B, T_FRAMES, seed=1234 + rank)
    c, f0, uv, sid = [t.to(dev) for t in cpu_in]
    net.enable_graph(not args.no_graph)

    def step():
        return net.infer(c, f0, uv, g=sid, noice_scale=0.4)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()

    samples_per_step = B * T_FRAMES * HOP
    value = world * samples_per_step * args.steps / elapsed

    # ---- a longer look at the same step (K = 20 steps are 0.15 s): >= 200 more replays, reported beside `value`, never as it ----
    steady = None
    if rank == 0 and world == 1 and not args.no_steady:
        ns = max(200, args.steps)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(ns):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / ns
        steady = dict(steps=ns, ms_per_step=round(1e3 * dt, 4), samples_per_s=samples_per_step / dt)

    # ---- N clips in flight (beside `value`, never as it): N independent B = 1 clips as parallel branches of ONE hipGraph
    # (SynthesizerTrn.infer_many).  Measured in a CHILD process (`--mode inflight`): a graph with forked branches is the one
    # construct that has crashed this runtime at capture time during development (nested forks, hipStreamEndCapture) — an extra
    # must not be able to take the headline line down with it ----
    inflight = None
    if rank == 0 and world == 1 and not args.no_steady and not args.no_graph:
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--mode", "inflight", "--steps", str(max(args.steps, 24))],
                               capture_output=True, text=True, timeout=300)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            inflight = json.loads(line[-1]) if r.returncode == 0 and line else dict(error=f"child exited {r.returncode}: {r.stderr[-300:]}")
        except Exception as e:      # noqa: BLE001
            inflight = dict(error=f"{type(e).__name__}: {e}")

    # ---- PCIe-inclusive rate (reported beside `value`, never as it): units / f0 / uv start in pinned host memory and the
    # waveform ends in pinned host memory, one clip at a time, synchronised per clip (what a caller holding host buffers sees) --
    host_io = None
    if rank == 0 and not args.no_host_io:
        hin = [t.pin_memory() for t in cpu_in[:3]]
        o0, _ = step()
        oh = torch.empty(o0.shape, dtype=o0.dtype).pin_memory()
        nio = max(3, min(args.steps, 10))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(nio):
            ch, fh, uh = [t.to(dev, non_blocking=True) for t in hin]
            o, _ = net.infer(ch, fh, uh, g=sid, noice_scale=0.4)
            oh.copy_(o, non_blocking=True)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / nio
        host_io = dict(ms_per_step=round(1e3 * dt, 4), samples_per_s=samples_per_step / dt,
                       h2d_bytes=int(sum(t.numel() * t.element_size() for t in hin)), d2h_bytes=int(oh.numel() * oh.element_size()),
                       note="pinned host -> HBM -> pinned host around every clip, synchronised per clip")

    # ---- roofline: per-launch hipEvent durations of every kernel family, eager pass over the same step ----
    roof = None
    if rank == 0 and not args.no_roofline:
        net.enable_graph(False)
        # the three MRF ResBlock chains of a decoder stage run on concurrent HIP streams in the timed region; a launch's
        # hipEvent duration is only meaningful when launches do not overlap, so this profiling pass serialises them
        import vdecoder.hifigan.models as _gen
        mrf_streams_was = _gen._MRF_STREAMS
        _gen._MRF_STREAMS = False
        step()
        torch.cuda.synchronize()
        S.prof_enable(True)
        S.prof_reset()
        nprof = min(args.steps, 5)
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        rep = S.prof_report()
        S.prof_enable(False)
        _gen._MRF_STREAMS = mrf_streams_was
        fam = max(rep.items(), key=lambda kv: kv[1]["ms"])
        name, r = fam
        achieved = r["flop"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0
        # HBM traffic needs rocprofv3 --pmc passes AROUND a process (MI355X_MICROARCH.md): when rocprofv3 is on the box this run
        # re-executes a short eager infer of this script under it, twice (collect_pmc_traffic), and quotes what it measured;
        # otherwise the committed PMC summary of an earlier builder-side run is quoted — only if it was taken on these kernel
        # sources, and labelled as such; null when neither exists.
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_conv1d_mfma.json")
        live = None
        if not args.no_pmc and world == 1 and os.environ.get("SVC_BENCH_PMC", "1") != "0":
            torch.cuda.synchronize()
            live = collect_pmc_traffic()
        if live is not None:
            traffic, _, traffic_source = live
        elif os.path.exists(pmc):
            try:
                import importlib.util
                spec = importlib.util.spec_from_file_location("svc_build", os.path.join(ROOT, "so-vits-svc_amd", "csrc", "build.py"))
                bld = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(bld)
                pj = json.load(open(pmc))
                if pj.get("csrc_sha") == bld.source_hash():
                    traffic = pj.get("hbm_bytes_per_launch")
                    traffic_source = ("profiles/pmc_conv1d_mfma.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                      f"command on the builder's box, same kernel sources (csrc_sha {pj['csrc_sha']}), "
                                      f"{pj.get('hbm_bytes_per_step', 0) / 1e9:.2f} GB per clip; NOT collected in this run")
                else:       # a summary of OTHER kernel sources says nothing about this build: no figure rather than a stale one
                    traffic_source = (f"profiles/pmc_conv1d_mfma.json was collected on csrc_sha {pj.get('csrc_sha')}, this build is "
                                      f"{bld.source_hash()}: not quoted")
            except Exception as e:      # noqa: BLE001
                traffic, traffic_source = None, f"profiles/pmc_conv1d_mfma.json unreadable ({type(e).__name__})"
        roof = dict(bound="mfma", kernel=name, achieved=round(achieved, 2), peak=PEAK_FP32_MFMA_TFLOPS,
                    unit="TFLOP/s", frac=round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), traffic=traffic,
                    traffic_source=traffic_source,
                    note="per-launch durations from a serialised eager pass (the timed region overlaps the three MRF chains of a "
                         "stage on concurrent streams: ms_per_step < sum of launch durations)",
                    launches_per_step=r["calls"] / nprof, avg_launch_us=round(1e3 * r["ms"] / r["calls"], 2),
                    flop_per_launch=r["flop"] / r["calls"],
                    families={k: dict(ms_per_step=round(v["ms"] / nprof, 4), calls=v["calls"] // nprof,
                                      tflops=round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else 0.0)
                              for k, v in rep.items()})

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, W, cpu_in)

    # ---- extra objects: the box, and the other BASELINE configs (bench_extra.py); one GPU, rank 0, bounded legs ----
    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:
        import bench_extra as X
        extras["device"] = X.guarded(X.device_info, dev)
        net.enable_graph(not args.no_graph)
        extras["e2e"] = X.guarded(X.bench_e2e, dev, net, (c, f0, uv, sid), T_FRAMES)
        ih = X.guarded(X.bench_infer_half, dev, net, (c, f0, uv, sid), T_FRAMES)
        if isinstance(ih, dict) and "ms_per_step" in ih:
            ih["speedup_vs_f32"] = round(1e3 * elapsed / args.steps / ih["ms_per_step"], 3)
        extras["infer_half"] = ih
        isp = X.guarded(X.bench_infer_split, dev, net, (c, f0, uv, sid), T_FRAMES)
        if isinstance(isp, dict) and "ms_per_step" in isp:
            isp["speedup_vs_f32_mfma"] = round(1e3 * elapsed / args.steps / isp["ms_per_step"], 3)
        extras["infer_split"] = isp
        # roofline.traffic of the two 16-bit-instruction legs: the same in-run PMC passes as the headline's, over their own launches
        if not args.no_pmc and os.environ.get("SVC_BENCH_PMC", "1") != "0":
            for key, flag, sfx in (("infer_half", "--half", "_h"), ("infer_split", "--split", "_hl")):
                leg = extras.get(key)
                if not (isinstance(leg, dict) and isinstance(leg.get("roofline"), dict)):
                    continue
                torch.cuda.synchronize()
                live = X.guarded(collect_pmc_traffic, 200, "infer", (flag,), lambda f, sfx=sfx: f.endswith(sfx))
                if isinstance(live, tuple):
                    leg["roofline"]["traffic"], leg["roofline"]["traffic_all_kernels_per_clip"], leg["roofline"]["traffic_source"] = live

    train_res = None
    if args.mode == "both":
        del net
        torch.cuda.empty_cache()
        if rank == 0 and world == 1 and not args.no_extras:
            extras["snake_b8"] = X.guarded(X.bench_snake_b8, dev)
            dres = X.guarded(X.bench_diffusion, dev)
            if isinstance(dres, tuple):
                extras["diffusion_train"], extras["diffusion_infer"] = dres
            else:
                extras["diffusion_train"] = extras["diffusion_infer"] = dres
        train_res = run_train(args, dev, rank, world, dist)
        if rank == 0 and world == 1 and not args.no_extras and train_res is not None:
            torch.cuda.empty_cache()
            # the reference's reduced-precision mode (fp16_run + half_type bf16), under its own key
            tb = X.guarded(run_train, args, dev, rank, world, dist, True)
            if isinstance(tb, dict) and "ms_per_step" in tb:
                tb["speedup_vs_f32"] = round(train_res["ms_per_step"] / tb["ms_per_step"], 3)
            train_res["train_bf16"] = tb
            torch.cuda.empty_cache()
            # the same iteration driven through the entry point's loader loop (files on disk -> DataLoader -> bucketed collate)
            tl = X.guarded(X.bench_train_loader, dev, train_hps(cfg))
            if isinstance(tl, dict) and "bucketed_graph" in tl:
                tl["vs_train_ms_per_step"] = round(tl["bucketed_graph"]["ms_per_step_without_epoch_start"] / train_res["ms_per_step"], 3)
                tl["vs_note"] = ("bucketed_graph.ms_per_step_without_epoch_start / train.ms_per_step; the loader's batches are padded "
                                 f"to {tl['bucketed_graph'].get('padded_frames')} frames, the fixed bench batch to {train_res['config']['frames']}")
            train_res["train_loader"] = tl

    if rank == 0:
        out = dict(metric="44.1kHz audio samples/sec (inference, SynthesizerTrn.split_f16().infer)" if args.split else
                   "44.1kHz audio samples/sec (inference, SynthesizerTrn.infer)", value=value,
                   unit="samples/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="f32 values as hi + lo fp16 planes in the generator, 3 fp16 MFMA per product, f32 accumulate (NOT the headline mode)" if args.split else "f32",
                   data="synthetic",
                   rtf=(elapsed / args.steps) / (samples_per_step / 44100.0),
                   config=dict(workload="BASELINE configs[1]: config_template.json, 1 speaker id, ContentVec768 units, "
                                        "NSF-HiFiGAN, one 10.01 s clip per step (T=862 frames, 441344 samples)",
                               batch=B, frames=T_FRAMES, samples_per_step=samples_per_step,
                               launch="hipGraph replay" if not args.no_graph else "eager",
                               parallelism=f"replicas x{world}" if world > 1 else "single GPU"),
                   steady_state=steady, clips_in_flight=inflight, roofline=roof, cpu_baseline=cpu, host_io=host_io, train=train_res, **extras)

        # LAST key of the line: the leg results again, compact — a log tail that cuts the long line still holds every figure
        def _ms(d, *path):
            for k in path:
                d = d.get(k) if isinstance(d, dict) else None
            return round(d, 3) if isinstance(d, (int, float)) else None
        out["summary"] = dict(
            infer_f32_ms=round(out["ms_per_step"], 3), infer_f32_roofline_frac=_ms(roof, "frac"),
            infer_split_ms=_ms(extras, "infer_split", "ms_per_step"), infer_half_ms=_ms(extras, "infer_half", "ms_per_step"),
            e2e_ms=_ms(extras, "e2e", "e2e_ms"),
            train_f32_ms=_ms(train_res, "ms_per_step"), train_f32_whole_step_frac=_ms(train_res, "roofline", "whole_step", "frac"),
            train_bf16_ms=_ms(train_res, "train_bf16", "ms_per_step"),
            train_loader_ms=_ms(train_res, "train_loader", "bucketed_graph", "ms_per_step_without_epoch_start"),
            train_per_rank_ms=train_res.get("per_rank_ms_per_step") if isinstance(train_res, dict) else None,
            train_allreduce=({k: train_res["allreduce"].get(k) for k in ("mode", "exposed_ms", "bytes", "launches")}
                             if isinstance(train_res, dict) and isinstance(train_res.get("allreduce"), dict) else None),
            snake_b8_ms=_ms(extras, "snake_b8", "ms_per_step"), diffusion_train_ms=_ms(extras, "diffusion_train", "ms_per_step"),
            cpu_baseline_samples_per_s=_ms(cpu, "value"), n_gpus=world)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
