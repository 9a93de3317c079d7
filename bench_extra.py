"""Extra objects of bench.py's JSON line (rank 0, one GPU): the other BASELINE.json configs and the box's identity.

  device           what this box is (name, CUs, clocks / power cap when rocm-smi answers) + one calibration launch — the
                   128-channel k=11 MRF conv in a hipGraph — so a run-to-run difference can be attributed to the box
  e2e              configs[1] end to end: 16 kHz wave -> ContentVec768L12 units (HuBERT-base stack) -> SynthesizerTrn.infer
  infer_half       configs[1] in the reference's half-precision mode (`net_g_ms.half()`): the generator's 16-bit pipeline
  snake_b8         configs[3]: nsf-snake-hifigan decoder, 30 s clips (T = 2584 frames), batch 8
  diffusion_train  configs[4]: WaveNet unit2mel training step, 20 x 512, B = 48 crops of 172 frames, fp32
  diffusion_infer  the same model sampling a 10 s clip: 100 DDIM steps (timesteps 1000, speedup 10)

Random-init weights of the named architectures (no checkpoint exists in the reference tree), synthetic inputs.  Nothing here
imports oracle/.  Every leg is bounded (a few seconds) and failure-tolerant: a leg that raises reports {"error": ...} instead of
taking the headline line down with it."""
import os
import subprocess
import time

import torch

HOP = 512


def _timeit(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def _graphed(fn, warm=2):
    """Capture fn() into a hipGraph (after `warm` eager calls); returns (replay, outputs of the captured call)."""
    import svc_hip as S
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with S.graph_capture(g):
        out = fn()
    return g.replay, out


def _families(fn, n=3):
    """Per-family hipEvent times of n eager calls — with the three MRF chains on ONE stream: an event pair around a launch also times
    whatever runs beside it on the other streams (round 5's `infer_split.families` were taken with the streams on and read 27 % above the
    rocprof durations of the same launches)."""
    import svc_hip as S
    import vdecoder.hifigan.models as _gen
    was = _gen._MRF_STREAMS
    _gen._MRF_STREAMS = False
    try:
        S.prof_enable(True)
        S.prof_reset()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        rep = S.prof_report()
        S.prof_enable(False)
    finally:
        _gen._MRF_STREAMS = was
    return {k: dict(ms_per_step=round(v["ms"] / n, 4), calls=v["calls"] // n,
                    tflops=round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else 0.0)
            for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])}


def device_info(dev):
    import svc_hip as S
    name, cus = S.device_info()
    props = torch.cuda.get_device_properties(dev)
    info = dict(name=name, cus=cus, hbm_gb=round(props.total_memory / 2 ** 30, 1), torch=torch.__version__,
                hip=getattr(torch.version, "hip", None))
    try:      # best effort: clocks and power cap as rocm-smi reports them right now (idle values; DVFS moves them under load)
        out = subprocess.run(["rocm-smi", "--showclocks", "--showmaxpower", "--showpower", "-d", str(dev.index or 0)],
                             capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "Max Graphics Package Power",
                                                                             "Average Graphics Package Power", "Socket Power"))]
        info["rocm_smi"] = keep[:8]
    except Exception as e:      # noqa: BLE001
        info["rocm_smi"] = f"unavailable ({type(e).__name__})"
    # calibration: ten launches of the decoder's 128-channel k = 11, dilation 5 conv on a 10 s clip's stage length, in one graph
    x = torch.randn(1, 128, 55168, device=dev)
    w = torch.randn(128, 128, 11, device=dev) / (128 * 11) ** 0.5
    wp = S.pack_conv1d_weight(w)
    b = torch.zeros(128, device=dev)
    out = torch.empty_like(x)
    n = 10

    def launches():
        for _ in range(n):
            S.conv1d(x, wp, 128, 11, bias=b, dil=5, pad_left=25, res=x, res_mode=1, out=out)
    replay, _ = _graphed(launches, warm=1)
    dt = _timeit(replay, 5, warm=2) / n
    info["calibration"] = dict(kernel="svc_conv1d_f32 128->128 k=11 d=5 T=55168 (+residual), 10 launches per graph replay",
                               us_per_launch=round(dt * 1e6, 1), tflops=round(2.0 * 128 * 128 * 11 * 55168 / dt / 1e12, 1))
    # launch-latency probe: 1000 empty one-wave kernels as nodes of one graph.  The calibration above (190 us launches) cannot see what
    # moved when a box runs every launch-heavy leg ~10 % slow at an identical calibration (two boxes of round 5); this can.
    try:
        nn_ = 1000

        def empties():
            for _ in range(nn_):
                S.check(S.lib().svc_debug_empty_kernel(S.stream_ptr()), "empty kernel")
        replay_e, _ = _graphed(empties, warm=1)
        de = _timeit(replay_e, 10, warm=3) / nn_
        info["launch_probe"] = dict(what=f"{nn_} empty one-wave kernels as nodes of one hipGraph (dependent chain on one stream)",
                                    us_per_node=round(de * 1e6, 3))
    except Exception as e:      # noqa: BLE001
        info["launch_probe"] = f"unavailable ({type(e).__name__}: {e})"
    # clocks / power WHILE the GPU works: the SMI read above is an idle value
    try:
        for _ in range(400):
            replay()
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "-d", str(dev.index or 0)], capture_output=True, text=True, timeout=20).stdout
        torch.cuda.synchronize()
        info["rocm_smi_under_load"] = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power"))][:8]
    except Exception as e:      # noqa: BLE001
        info["rocm_smi_under_load"] = f"unavailable ({type(e).__name__})"
    return info


def bench_e2e(dev, net, inputs, frames):
    """wav16k -> units -> infer.  `net` is the headline SynthesizerTrn (graph mode on), `inputs` its (c, f0, uv, sid)."""
    import utils
    from vencoder.ContentVec768L12 import ContentVec768L12
    from vencoder.hubert import hubert_model as HM
    torch.manual_seed(7)
    enc = ContentVec768L12(device=dev, model=HM.Hubert())       # random-init HuBERT-base stack (no checkpoint in the tree)
    n16 = int(round(frames * HOP / 44100 * 16000))
    wav = 0.3 * torch.randn(n16, device=dev)
    c, f0, uv, sid = inputs

    def units():
        return enc.encoder(wav)

    u = units()
    t_units_eager = _timeit(units, 5)
    fam = _families(units)
    replay, u_static = _graphed(units)
    t_units = _timeit(replay, 10)

    def expand():
        return utils.repeat_expand_2d(u_static.squeeze(0), frames, "left").unsqueeze(0)
    t_expand = _timeit(expand, 5)

    def whole():
        replay()
        return net.infer(expand(), f0, uv, g=sid, noice_scale=0.4)
    t_whole = _timeit(whole, 5)
    n = frames * HOP
    return dict(workload=f"10.01 s clip: {n16} samples @16 kHz -> ContentVec768L12 (HuBERT-base, 12 layers, layer-12 output "
                         f"{tuple(u.shape)}) -> repeat_expand to {frames} frames -> SynthesizerTrn.infer",
                unit_encoder_ms=round(1e3 * t_units, 3), unit_encoder_eager_ms=round(1e3 * t_units_eager, 3),
                repeat_expand_ms=round(1e3 * t_expand, 3), e2e_ms=round(1e3 * t_whole, 3), e2e_samples_per_s=n / t_whole,
                note="unit encoder and synthesizer each replayed from a hipGraph; repeat_expand_2d (utils.py:396-424) is a device "
                     "gather through an index cached per (source, target) length pair",
                unit_encoder_families=fam)


def bench_infer_half(dev, net, inputs, frames, steps=20):
    """The headline clip through the reference's half-precision inference mode (inference/infer_tool.py:196-198 `net_g_ms.half()`
    on a compress_model.py checkpoint) = SynthesizerTrn.half(): the generator's 16-bit pipeline (fp16 activations in HBM and LDS,
    fp16 weights, v_mfma_f32_32x32x16_f16 with fp32 accumulation).  Own key, own roofline against the dense 16-bit MFMA peak;
    never the headline.  The waveform's distance to the fp32 path on the same noise is measured here, on the spot."""
    import svc_hip as S
    c, f0, uv, sid = inputs
    noise = dict(enc_p=torch.randn(1, net.inter_channels, frames, device=dev), rand_ini=torch.rand(1, 9, device=dev),
                 sine=torch.randn(1, frames * net.dec.upp, 9, device=dev))
    was_graph = net.use_graph
    net.enable_graph(False)
    o32, _ = net.infer(c, f0, uv, g=sid, noice_scale=0.4, noise=noise)
    net.half()
    try:
        oh, _ = net.infer(c, f0, uv, g=sid, noice_scale=0.4, noise=noise)
        mse = (oh - o32).pow(2).mean().item()
        mx = (oh - o32).abs().max().item()
        fams = _families(lambda: net.infer(c, f0, uv, g=sid, noice_scale=0.4), n=3)
        net.enable_graph(True)
        step = lambda: net.infer(c, f0, uv, g=sid, noice_scale=0.4)
        dt = _timeit(step, steps, warm=3)
    finally:
        net.float()
        net.enable_graph(was_graph)
    samples = frames * HOP
    h = {k: v for k, v in fams.items() if k.endswith("_h")}
    hms = sum(v["ms_per_step"] for v in h.values())
    hflop = sum(v["tflops"] * v["ms_per_step"] * 1e9 for v in h.values())
    peak = 2500.0
    ach = hflop / (hms * 1e-3) / 1e12 if hms > 0 else 0.0
    return dict(metric="44.1kHz audio samples/sec (inference, SynthesizerTrn.half().infer)", value=samples / dt, unit="samples/s",
                ms_per_step=round(1e3 * dt, 4), steps=steps, dtype="fp16 activations + weights in the generator and the flow (one fused "
                "fp16 kernel per coupling layer), f32 accumulate; encoder / harmonic source f32", launch="hipGraph replay",
                waveform_vs_f32_path=dict(mse=mse, max_abs=mx, bar="north_star: MSE < 1e-4"),
                roofline=dict(bound="mfma", kernel="+".join(sorted(h)), achieved=round(ach, 1), peak=peak, unit="TFLOP/s",
                              frac=round(ach / peak, 4), kernel_ms_per_step=round(hms, 4), traffic=None,
                              note="16-bit conv launches of the generator (serialised eager pass, hipEvents per launch) against "
                                   "the dense f16 MFMA peak"),
                families=fams)


def bench_infer_split(dev, net, inputs, frames, steps=20, oracle_err=None):
    """The headline clip with the generator on the SPLIT pipeline (SynthesizerTrn.split_f16(); csrc/conv1d_hl.hip): fp32 inference
    whose generator convolutions run on the fp16 matrix instruction at fp32-level precision — every value a hi and a lo fp16 plane
    (22 mantissa bits), every product three fp16 instructions with fp32 accumulation, where the fp32 instruction needs sixteen.
    Own key, never the headline (the headline is the fp32-MFMA path).  The waveform's distance to the fp32-MFMA path on the same noise
    is measured here; the FLOPs counted are the convolutions' (one multiply-add per product, not three), against the fp32 MFMA peak —
    a fraction above 1 is what the split buys."""
    c, f0, uv, sid = inputs
    noise = dict(enc_p=torch.randn(1, net.inter_channels, frames, device=dev), rand_ini=torch.rand(1, 9, device=dev),
                 sine=torch.randn(1, frames * net.dec.upp, 9, device=dev))
    was_graph = net.use_graph
    net.enable_graph(False)
    o32, _ = net.infer(c, f0, uv, g=sid, noice_scale=0.4, noise=noise)
    net.split_f16()
    try:
        oh, _ = net.infer(c, f0, uv, g=sid, noice_scale=0.4, noise=noise)
        mse = (oh - o32).pow(2).mean().item()
        mx = (oh - o32).abs().max().item()
        fams = _families(lambda: net.infer(c, f0, uv, g=sid, noice_scale=0.4), n=3)
        net.enable_graph(True)
        step = lambda: net.infer(c, f0, uv, g=sid, noice_scale=0.4)
        dt = _timeit(step, steps, warm=3)
    finally:
        net.float()
        net.enable_graph(was_graph)
    samples = frames * HOP
    h = {k: v for k, v in fams.items() if k.endswith("_hl")}
    hms = sum(v["ms_per_step"] for v in h.values())
    hflop = sum(v["tflops"] * v["ms_per_step"] * 1e9 for v in h.values())
    ach = hflop / (hms * 1e-3) / 1e12 if hms > 0 else 0.0
    return dict(metric="44.1kHz audio samples/sec (inference, SynthesizerTrn.split_f16().infer)", value=samples / dt, unit="samples/s",
                ms_per_step=round(1e3 * dt, 4), steps=steps,
                dtype="f32 values as hi + lo fp16 planes in the generator and the flow (22 mantissa bits), 3 fp16 MFMA per product, f32 "
                      "accumulate; encoder / harmonic source f32", launch="hipGraph replay",
                waveform_vs_f32_mfma_path=dict(mse=mse, max_abs=mx, max_abs_waveform=o32.abs().max().item()),
                roofline=dict(bound="mfma", kernel="+".join(sorted(h)), achieved=round(3 * ach, 1), peak=2500.0,
                              unit="TFLOP/s", frac=round(3 * ach / 2500.0, 4),
                              delivered=round(ach, 1), frac_delivered_vs_f32_mfma_peak=round(ach / 157.3, 4),
                              kernel_ms_per_step=round(hms, 4), traffic=None,
                              note="split launches (generator convolutions + the fused flow couplings; serialised eager pass, hipEvents "
                                   "per launch).  achieved / frac = fp16 matrix instructions ISSUED (three per delivered product) against "
                                   "the dense f16 MFMA peak they run on; delivered = the convolutions' own FLOPs, whose ratio to the "
                                   "fp32 MFMA peak (157.3) is what the split buys over the fp32 instruction"),
                families=fams)


def bench_snake_b8(dev, steps=3):
    import models
    import synthetic_data as W
    cfg = W.full_config()
    cfg["vocoder_name"] = "nsf-snake-hifigan"
    kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
    net = models.SynthesizerTrn(cfg["spec_channels"], cfg["segment_size"], **kw)
    net.load_state_dict(W.make_state_dict(cfg, 77))
    net = net.to(dev).eval()
    B, T = 8, 2584
    c, f0, uv, sid = [t.to(dev) for t in W.make_inputs(cfg, B, T, seed=21)]
    net.enable_graph(True)

    def step():
        return net.infer(c, f0, uv, g=sid, noice_scale=0.4)
    dt = _timeit(step, steps, warm=2)
    n = B * T * HOP
    # the same batch in the reference's half-precision mode (SynthesizerTrn.half(): generator + SnakeAlias sites on blocked fp16)
    half = None
    try:
        noise = dict(enc_p=torch.randn(B, net.inter_channels, T, device=dev), rand_ini=torch.rand(B, 9, device=dev),
                     sine=torch.randn(B, T * net.dec.upp, 9, device=dev))
        net.enable_graph(False)
        o32, _ = net.infer(c, f0, uv, g=sid, noice_scale=0.4, noise=noise)
        net.half()
        oh, _ = net.infer(c, f0, uv, g=sid, noice_scale=0.4, noise=noise)
        mse = (oh - o32).pow(2).mean().item()
        del oh
        net.enable_graph(True)
        dth = _timeit(step, steps, warm=2)
        half = dict(ms_per_step=round(1e3 * dth, 3), samples_per_s=n / dth, speedup_vs_f32=round(dt / dth, 3), waveform_mse_vs_f32_path=mse,
                    note="SynthesizerTrn.half(): fp16 activations + weights in the generator (conv1d_h, snake_alias_h), f32 accumulate")
        # ... and on the split pipeline (SynthesizerTrn.split_f16(): hi + lo fp16 planes, three fp16 MFMA per product, fp32-level output)
        try:
            net.float()
            net.split_f16()
            net.enable_graph(False)
            osp, _ = net.infer(c, f0, uv, g=sid, noice_scale=0.4, noise=noise)
            mxs = (osp - o32).abs().max().item()
            del osp
            net.enable_graph(True)
            dts = _timeit(step, steps, warm=2)
            split = dict(ms_per_step=round(1e3 * dts, 3), samples_per_s=n / dts, speedup_vs_f32_mfma=round(dt / dts, 3),
                         waveform_max_abs_vs_f32_mfma_path=mxs,
                         note="SynthesizerTrn.split_f16(): conv1d_hl / snake_alias_hl (resblock pairs unfused: a SnakeAlias sits between the convs)")
        except Exception as e:      # noqa: BLE001
            split = dict(error=f"{type(e).__name__}: {e}")
        half["split"] = split
        del o32, noise
    except Exception as e:      # noqa: BLE001
        half = dict(error=f"{type(e).__name__}: {e}")
    del net
    torch.cuda.empty_cache()
    return dict(workload="BASELINE configs[3]: nsf-snake-hifigan (SnakeAlias activations), batch 8 x 30.0 s clips "
                         f"(T={T} frames, {T * HOP} samples each), fp32, hipGraph replay",
                ms_per_step=round(1e3 * dt, 3), samples_per_s=n / dt, rtf=dt / (n / 44100.0), steps=steps, half=half)


def bench_diffusion(dev):
    from diffusion import solver
    from diffusion.unit2mel import Unit2Mel
    torch.manual_seed(0)
    L, C, H, M = 20, 512, 256, 128
    net = Unit2Mel(768, 1, False, M, L, C, H, 1000, 1000).to(dev)
    torch.nn.init.normal_(net.decoder.denoise_fn.output_projection.weight, std=0.02)
    # ---- training step (configs_template/diffusion_template.yaml: batch 48, 2 s crops = 172 frames)
    net.train()
    B, T = 48, 172
    step = solver.TrainStep(net, solver.build_optimizer(net, lr=1e-4))
    data = dict(units=torch.randn(B, T, 768, device=dev), f0=200 + 100 * torch.rand(B, T, 1, device=dev),
                volume=torch.rand(B, T, 1, device=dev), spk_id=torch.zeros(B, 1, dtype=torch.long, device=dev),
                mel=-6 + 2 * torch.randn(B, T, M, device=dev))
    step.enable_graph(True)
    dt = _timeit(lambda: step(data), 8, warm=3)
    flop_fwd = 2.0 * B * T * (M * C + L * (C * 2 * C * 3 + H * 2 * C + C * 2 * C) + C * C + C * M + 768 * H)
    train = dict(workload=f"BASELINE configs[4]: WaveNet unit2mel {L} x {C}, {M} mels, 768-d units, batch {B} x {T} frames, fp32, "
                          "FusedAdamW, whole iteration replayed from one hipGraph",
                 ms_per_step=round(1e3 * dt, 3), steps_per_s=round(1.0 / dt, 3), tflops=round(3 * flop_fwd / dt / 1e12, 1))
    step.enable_graph(False)
    # ---- sampling a 10 s clip: 100 DDIM steps
    net.eval()
    Ti = 862
    units = torch.randn(1, Ti, 768, device=dev)
    f0 = 100 + 300 * torch.rand(1, Ti, 1, device=dev)
    vol = torch.rand(1, Ti, 1, device=dev)
    spk = torch.zeros(1, 1, dtype=torch.long, device=dev)

    def sample():
        with torch.no_grad():
            return net(units, f0, vol, spk_id=spk, infer=True, infer_speedup=10, method="ddim", use_tqdm=False)
    t_eager = _timeit(sample, 2, warm=1)
    infer = dict(workload="10 s clip (T=862 frames): Unit2Mel conditioning + 100 DDIM denoiser steps (timesteps 1000, speedup 10)",
                 ms_per_clip_eager=round(1e3 * t_eager, 2))
    try:
        replay, _ = _graphed(sample, warm=1)
        t_graph = _timeit(replay, 3, warm=1)
        infer.update(ms_per_clip=round(1e3 * t_graph, 2), launch="hipGraph replay of the whole 100-step sampler")
    except Exception as e:      # noqa: BLE001 — a sampler that syncs with the host cannot be captured: report the eager figure
        infer.update(ms_per_clip=infer["ms_per_clip_eager"], launch=f"eager (capture failed: {type(e).__name__}: {str(e)[:120]})")
    del net, step
    torch.cuda.empty_cache()
    return train, infer


def write_train_dataset(root, n_items, seed=4321, n_spk=4, ssl_dim=768, n_fft=2048):
    """BASELINE configs[2]'s synthetic set ON DISK in the reference's formats (preprocess_hubert_f0.py:31-103): per item a
    44.1 kHz int16 wav, `.soft.pt` units at 50 fps, `.f0.npy` (f0, uv), a cached `.spec.pt`; T ~ U{300..790} frames."""
    import numpy as np
    from scipy.io.wavfile import write
    g = torch.Generator().manual_seed(seed)
    lines = []
    for i in range(n_items):
        d = os.path.join(root, "dataset", f"spk{i % n_spk}")
        os.makedirs(d, exist_ok=True)
        T = int(torch.randint(300, 791, (1,), generator=g))
        p = os.path.join(d, f"u{i}.wav")
        write(p, 44100, ((torch.rand(T * HOP, generator=g) - 0.5) * 32767).to(torch.int16).numpy())
        torch.save(torch.randn(1, ssl_dim, T // 2 + 1, generator=g), p + ".soft.pt")
        f0 = (100 + 300 * torch.rand(T, generator=g)).numpy()
        for s0 in torch.randint(0, T - 8, (max(1, T // 80),), generator=g).tolist():
            f0[s0:s0 + 8] = 0
        np.save(p + ".f0.npy", np.asanyarray((f0, (f0 > 0).astype(float)), dtype=object), allow_pickle=True)
        torch.save(torch.randn(n_fft // 2 + 1, T, generator=g).abs(), p.replace(".wav", ".spec.pt"))
        lines.append(p)
    fl = os.path.join(root, "train.txt")
    with open(fl, "w") as f:
        f.write("\n".join(lines) + "\n")
    return fl


class _TimedLoader:
    """Wraps a DataLoader: records how long the training loop waited for each batch (host time not hidden behind the GPU)."""

    def __init__(self, loader):
        self.loader, self.waits = loader, []

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        it = iter(self.loader)
        while True:
            t0 = time.perf_counter()
            try:
                batch = next(it)
            except StopIteration:
                return
            self.waits.append(time.perf_counter() - t0)
            yield batch


def bench_train_loader(dev, hps_dict, n_items=192, epochs=3):
    """The training number THROUGH the entry point's loop: train.make_loaders (reference-format files,
    DataLoader workers, collate) -> train.train_and_evaluate -> TrainStep, on a synthetic data set of `n_items` utterances with
    T ~ U{300..790}: once as `svc_run.py train.py` runs by default (length-bucketed collate, one hipGraph per padded shape) and
    once with SVC_TRAIN_GRAPH=0 (the reference's pad-to-longest collate, eager launches).  The first epoch (graph captures,
    worker start-up, page cache) is not timed."""
    import logging
    import shutil
    import tempfile
    import synthetic_data as W
    import train as TR
    import utils
    root = tempfile.mkdtemp(prefix="svc_bench_ds_")
    try:
        fl = write_train_dataset(root, n_items)
        cfg = W.full_config()
        h = dict(hps_dict)
        h["train"] = dict(h["train"], use_sr=True, max_speclen=512, vol_aug=False, all_in_mem=False, log_interval=10 ** 9,
                          eval_interval=10 ** 9, seed=1234, keep_ckpts=0, epochs=epochs)
        h["data"] = dict(h["data"], training_files=fl, validation_files=fl, max_wav_value=32768.0, unit_interpolate_mode="nearest")
        h["spk"] = {f"spk{i}": i for i in range(4)}
        hps = utils.HParams(**h)
        hps.model_dir = root
        logger = logging.getLogger("svc_bench_train_loader")
        logger.addHandler(logging.NullHandler())
        logger.propagate = False
        out = {}
        for name, use_graph in (("bucketed_graph", True), ("eager", False)):
            torch.manual_seed(1234)
            net_g, net_d, optim_g, optim_d = TR.build(hps, dev)
            net_g.module.load_state_dict(W.make_train_state_dict(cfg, 1234))
            net_d.module.load_state_dict(W.make_mpd_state_dict(1235))
            step = TR.TrainStep(hps, net_g, net_d, optim_g, optim_d).enable_graph(use_graph)
            loaders = TR.make_loaders(hps, 0, 1, use_graph)
            loaders = [_TimedLoader(loaders[0]), loaders[1]]
            TR.global_step = 1                       # nothing logs / evaluates / saves at these intervals
            times = []
            for epoch in range(1, epochs + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                TR.train_and_evaluate(0, epoch, hps, step, loaders, logger, [TR._NullWriter(), TR._NullWriter()], dev)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            n_steps = len(loaders[0])
            n_timed = n_steps * (epochs - 1)
            dt = sum(times[1:]) / n_timed
            waits = loaders[0].waits
            first = [waits[e * n_steps] for e in range(1, epochs)]          # the first batch of an epoch: nothing is prefetched yet
            steady = (sum(times[1:]) - sum(first)) / n_timed
            out[name] = dict(ms_per_step=round(1e3 * dt, 2), steps_per_s=round(1.0 / dt, 3), first_epoch_s=round(times[0], 2),
                             timed_steps=n_timed, steps_per_epoch=n_steps,
                             epoch_start_wait_ms=round(1e3 * sum(first) / len(first), 1),
                             ms_per_step_without_epoch_start=round(1e3 * steady, 2),
                             loader_wait_ms_per_step=round(1e3 * (sum(waits[n_steps:]) - sum(first)) / n_timed, 2))
            if use_graph:
                out[name]["graphs"] = len(step._graphs)
                out[name]["padded_frames"] = sorted({k[0][0][2] for k in step._graphs})
                out[name]["eager_fallbacks"] = step.eager_fallbacks
            del step, net_g, net_d, optim_g, optim_d, loaders
            torch.cuda.empty_cache()
        out["workload"] = (f"train.make_loaders + train.train_and_evaluate on {n_items} synthetic utterances on disk (wav, .soft.pt, "
                           f".f0.npy, .spec.pt; T~U{{300..790}}), batch {hps.train.batch_size}, DataLoader workers as train.run sets them, "
                           f"{epochs - 1} timed epochs after one warm-up epoch; `epoch_start_wait_ms` = the wait for an epoch's first batch "
                           "(16 items read by one worker, nothing prefetched across the epoch boundary — the reference's loader, "
                           "which also re-forks its workers there, pays the same): amortised over a real data set's epochs, "
                           "not over these 12-step ones")
        return out
    finally:
        shutil.rmtree(root, ignore_errors=True)


def guarded(fn, *a, **k):
    try:
        return fn(*a, **k)
    except Exception as e:      # noqa: BLE001
        import traceback
        return dict(error=f"{type(e).__name__}: {str(e)[:300]}", where=traceback.format_exc().strip().splitlines()[-3:])
