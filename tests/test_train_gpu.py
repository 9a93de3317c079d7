"""Training-graph parity on MI355X: the HIP SynthesizerTrn.forward + MultiPeriodDiscriminator + mel + losses and their
gradients against (a) the committed vectors of the REAL reference (tests/golden/train_small.npz) and (b) the CPU oracle's
autograd on the same seeded inputs / injected noise.  fp32 throughout.  Tolerances: losses 1e-4 relative; per-parameter
gradient L2 norms 2e-3 relative; full gradients 1e-3 of the tensor's max magnitude (fp32 atomics / fmaf-chain order)."""
import numpy as np
import pytest
import torch

from train_common import LOSS_KEYS, load_case, load_dropout_case, load_transflow_case

pytestmark = pytest.mark.gpu


def _build(cs, dev):
    import models
    cfg = cs["cfg"]
    kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
    net_g = models.SynthesizerTrn(cfg["spec_channels"], cfg["segment_size"], **kw)
    net_g.load_state_dict(cs["sd_g"], strict=True)
    net_d = models.MultiPeriodDiscriminator()
    net_d.load_state_dict(cs["sd_d"], strict=True)
    return net_g.to(dev).train(), net_d.to(dev).train()


def _step(cs, net_g, net_d, dev):
    """train.py:167-207 with the mirror modules (same code shape as the reference's loop body)."""
    import modules.commons as commons
    from modules.losses import discriminator_loss, feature_loss, generator_loss, kl_loss
    from modules.mel_processing import mel_spectrogram_torch, spec_to_mel_torch
    import svc_autograd as A
    d = cs["data"]
    c, f0, uv, spec, y, sid, lengths = [t.to(dev) for t in cs["batch"]]
    noise = {k: v.to(dev) for k, v in cs["noise"].items()}
    seg, hop = cs["cfg"]["segment_size"], d["hop"]
    mel = spec_to_mel_torch(spec, d["n_fft"], d["n_mels"], d["sr"], d["fmin"], d["fmax"])
    y_hat, ids_slice, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), pred_lf0, norm_lf0, lf0 = net_g(
        c, f0, uv, spec, g=sid, c_lengths=lengths, spec_lengths=lengths, noise=noise)
    y_mel = commons.slice_segments(mel, ids_slice, seg)
    y_hat_mel = mel_spectrogram_torch(y_hat.squeeze(1), d["n_fft"], d["n_mels"], d["sr"], hop, d["win"], d["fmin"], d["fmax"])
    y_seg = commons.slice_segments(y, ids_slice * hop, seg * hop)
    rs, gs, _, _ = net_d(y_seg, y_hat.detach())
    loss_disc, _, _ = discriminator_loss(rs, gs)
    rs, gs, fr, fg = net_d(y_seg, y_hat)
    loss_mel = A.sum_abs_diff(y_mel, y_hat_mel) / y_mel.numel() * 45.0
    loss_kl = kl_loss(z_p, logs_q, m_p, logs_p, z_mask) * 1.0
    loss_fm = feature_loss(fr, fg)
    loss_gen, _ = generator_loss(gs)
    loss_lf0 = A.sum_sq_diff(pred_lf0, lf0) / lf0.numel()
    loss_gen_all = loss_gen + loss_fm + loss_mel + loss_kl + loss_lf0
    return dict(loss_disc=loss_disc, loss_gen=loss_gen, loss_fm=loss_fm, loss_mel=loss_mel, loss_kl=loss_kl,
                loss_lf0=loss_lf0, loss_gen_all=loss_gen_all, y_hat=y_hat)


def _bench_like_items(dev, n=2):
    """-> f(fp16_run, half_type) -> (hps dict, items on the device): the first `n` items of bench.py's training batch on the
    full template (what tests that drive train.TrainStep as an object need)."""
    import bench
    import synthetic_data as W
    cfg = W.full_config()
    (c, f0, spec, y, spk, lengths, uv, _), _ = bench.make_train_items(cfg, bench.TRAIN_B, 4321)
    T = int(lengths[:n].max())
    items = tuple(t.to(dev) if t is not None else None for t in
                  (c[:n, :, :T], f0[:n, :T], spec[:n, :, :T], y[:n, :, :T * bench.HOP], spk[:n], lengths[:n], uv[:n, :T], None))

    def make(fp16_run, half_type):
        hps = bench.train_hps(cfg)
        hps["train"] = dict(hps["train"], fp16_run=fp16_run, half_type=half_type)
        return hps, items
    return make


def test_training_step_matches_reference_and_oracle(dev):
    cs = load_case()
    z = cs["z"]
    net_g, net_d = _build(cs, dev)
    out = _step(cs, net_g, net_d, dev)
    for k in LOSS_KEYS:
        ref = float(z["loss." + k])
        got = float(out[k])
        assert abs(got - ref) <= 1e-4 * max(1.0, abs(ref)), (k, got, ref)
    yh = out["y_hat"].detach().cpu().numpy()
    assert np.abs(yh - z["y_hat"]).max() <= 2e-4 * max(1.0, np.abs(z["y_hat"]).max())

    # D step gradients
    out["loss_disc"].backward(retain_graph=True)
    gd = {k: p.grad.detach().cpu() for k, p in net_d.named_parameters()}
    report = []
    for k, n in zip([str(k) for k in z["gnorm_d_keys"]], z["gnorm_d"]):
        report.append(("D", k, abs(gd[k].norm().item() - n) / max(n, 1e-6), n))
    for name in z.files:
        if name.startswith("grad_d."):
            g = gd[name[7:]].numpy()
            assert np.abs(g - z[name]).max() <= GRAD_ELEM_TOL * max(np.abs(z[name]).max(), 1e-6), name
    net_d.zero_grad()
    # G step gradients
    out["loss_gen_all"].backward()
    gg = {k: p.grad.detach().cpu() for k, p in net_g.named_parameters() if p.grad is not None}
    for k, n in zip([str(k) for k in z["gnorm_g_keys"]], z["gnorm_g"]):
        if k.endswith("conv_k.bias"):
            continue
        assert k in gg, k
        report.append(("G", k, abs(gg[k].norm().item() - n) / max(n, 1e-5), n))
    for name in z.files:
        if name.startswith("grad_g."):
            g = gg[name[7:]].numpy()
            assert np.abs(g - z[name]).max() <= GRAD_ELEM_TOL * max(np.abs(z[name]).max(), 1e-6), name
    _check_grad_norms(report)


# Gradient tolerances of an fp32 path against the fp32 reference (2e-3 on the norm would let a systematic
# 0.1 % error of one conv's wgrad through).  Every parameter's gradient NORM agrees with the reference's to GRAD_NORM_TOL; the
# tensors in GRAD_NORM_LOOSE are the measured exceptions, each with its reason.  What the fp32 round-off consists of here: the
# time reduction of a weight gradient runs as 256 partial sums combined by atomics (order varies run to run), and the backward
# chain re-associates every convolution's reduction (MFMA groups of 2 / 4 channels), where the CPU reference runs oneDNN's
# blocked GEMMs — relative differences of ~1e-6 per op compound over the ~100-layer path, and cancel less in tensors whose
# gradient is a small difference of large terms.
GRAD_NORM_TOL = 1e-4          # measured worst: 7.7e-5 (dec.m_source.l_linear.bias), every other tensor <= 1.5e-5 (profiles/r05d_grad_norm_errors.txt)
GRAD_ELEM_TOL = 1e-3          # sampled elements, relative to the tensor's largest: dominated by the same effects on small entries
GRAD_NORM_LOOSE = {}          # name suffix -> (tolerance, reason): no tensor needs one (662 of 662 inside GRAD_NORM_TOL)


def _check_grad_norms(report):
    import os
    report = sorted(report, key=lambda r: -r[2])
    path = os.environ.get("SVC_GRAD_REPORT")
    if path:
        with open(path, "w") as f:
            f.write("# |norm(grad HIP) - norm(grad reference)| / norm(reference), tests/golden/train_small.npz, worst first\n")
            for net, k, e, n in report:
                f.write(f"{net} {k:60s} rel_err {e:.3e}  ref_norm {n:.3e}\n")
    bad = []
    for net, k, e, n in report:
        tol = GRAD_NORM_TOL
        for suffix, (t, _why) in GRAD_NORM_LOOSE.items():
            if k.endswith(suffix):
                tol = t
        if e > tol:
            bad.append((net, k, f"{e:.2e}", f"{n:.2e}"))
    assert not bad, bad[:12]


def test_training_forward_with_vol_embedding_matches_oracle(dev):
    """vol_embedding=True (models.py:469): HIP training forward + the gradients of emb_vol / pre against the oracle's
    torch-CPU autograd on the same injected noise."""
    import models
    from oracle import train_oracle as TO
    from oracle import weights as W
    cs = load_case()
    cfg = dict(cs["cfg"], vol_embedding=True)
    sd = W.make_train_state_dict(cfg, 31)
    kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
    net = models.SynthesizerTrn(cfg["spec_channels"], cfg["segment_size"], **kw)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).train()
    c, f0, uv, spec, y, sid, lengths = cs["batch"]
    g = torch.Generator().manual_seed(5)
    vol = torch.rand(c.shape[0], c.shape[2], generator=g)
    sg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = TO.synth_forward(sg, cfg, c, f0, uv, spec, sid, lengths, lengths, cs["noise"], vol=vol)
    out = net(c.to(dev), f0.to(dev), uv.to(dev), spec.to(dev), g=sid.to(dev), c_lengths=lengths.to(dev),
              spec_lengths=lengths.to(dev), vol=vol.to(dev), noise={k: v.to(dev) for k, v in cs["noise"].items()})
    yh, yr = out[0].detach().cpu(), ref[0].detach()
    assert (yh - yr).abs().max().item() <= 2e-4 * max(1.0, yr.abs().max().item())
    import svc_autograd as A
    # y_hat depends on the posterior path only; the prior statistics m_p carry the emb_vol / pre gradients
    mh, mr = out[3][2], ref[3][2]
    assert (mh.detach().cpu() - mr.detach()).abs().max().item() <= 2e-4 * max(1.0, mr.abs().max().item())
    (A.sum_sq(mh) / mh.numel()).backward()
    mr.pow(2).mean().backward()
    for k in ("emb_vol.weight", "emb_vol.bias", "pre.weight"):
        gh = dict(net.named_parameters())[k].grad.cpu()
        gr = sg[k].grad
        assert (gh - gr).abs().max().item() <= 2e-3 * max(gr.abs().max().item(), 1e-6), k


def test_snake_decoder_training_matches_oracle(dev):
    """vocoder_name="nsf-snake-hifigan" in the TRAINING graph: y_hat and the gradients of decoder parameters (incl. the
    SnakeAlias alpha / beta of several sites) against the oracle's torch-CPU autograd on the same injected noise."""
    import models
    import svc_autograd as A
    from oracle import train_oracle as TO
    from oracle import weights as W
    cs = load_case()
    cfg = dict(cs["cfg"], vocoder_name="nsf-snake-hifigan")
    sd = W.make_train_state_dict(cfg, 41)
    kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
    net = models.SynthesizerTrn(cfg["spec_channels"], cfg["segment_size"], **kw)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).train()
    c, f0, uv, spec, y, sid, lengths = cs["batch"]
    sg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith(".filter") else v.clone())
          for k, v in sd.items()}
    ref = TO.synth_forward(sg, cfg, c, f0, uv, spec, sid, lengths, lengths, cs["noise"])
    out = net(c.to(dev), f0.to(dev), uv.to(dev), spec.to(dev), g=sid.to(dev), c_lengths=lengths.to(dev),
              spec_lengths=lengths.to(dev), noise={k: v.to(dev) for k, v in cs["noise"].items()})
    yh, yr = out[0], ref[0]
    assert (yh.detach().cpu() - yr.detach()).abs().max().item() <= 2e-4 * max(1.0, yr.abs().max().item())
    (A.sum_sq(yh) / yh.numel()).backward()
    yr.pow(2).mean().backward()
    named = dict(net.named_parameters())
    for k in ("dec.snakes.0.act.alpha", "dec.snakes.2.act.beta", "dec.resblocks.0.activations.1.act.alpha",
              "dec.resblocks.7.activations.4.act.beta", "dec.snake_post.act.alpha", "dec.conv_pre.weight_v",
              "dec.ups.1.weight_g", "enc_q.pre.weight"):
        gh, gr = named[k].grad.cpu(), sg[k].grad
        assert (gh - gr).abs().max().item() <= 2e-3 * max(gr.abs().max().item(), 1e-6), k


def test_tiny_template_training_matches_oracle(dev):
    """use_depthwise_conv + flow_share_parameter (config_tiny_template) in the TRAINING graph: z_p / y_hat and gradients of
    the depthwise / pointwise / shared-WN parameters against the oracle's torch-CPU autograd."""
    import models
    import svc_autograd as A
    from oracle import train_oracle as TO
    from oracle import weights as W
    cs = load_case()
    cfg = dict(cs["cfg"], use_depthwise_conv=True, flow_share_parameter=True)
    sd = W.make_train_state_dict(cfg, 43)
    kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
    net = models.SynthesizerTrn(cfg["spec_channels"], cfg["segment_size"], **kw)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).train()
    c, f0, uv, spec, y, sid, lengths = cs["batch"]
    # the shared WN appears under five prefixes in the state_dict: build the oracle's leaf tensors once per storage
    leaves = {}
    sg = {}
    for k, v in sd.items():
        key = v.data_ptr()
        if key not in leaves:
            leaves[key] = v.clone().requires_grad_(True)
        sg[k] = leaves[key]
    ref = TO.synth_forward(sg, cfg, c, f0, uv, spec, sid, lengths, lengths, cs["noise"])
    out = net(c.to(dev), f0.to(dev), uv.to(dev), spec.to(dev), g=sid.to(dev), c_lengths=lengths.to(dev),
              spec_lengths=lengths.to(dev), noise={k: v.to(dev) for k, v in cs["noise"].items()})
    zh, zr = out[3][1], ref[3][1]                       # z_p = flow(z): exercises enc_q (depthwise WN) and the shared flow WN
    assert (zh.detach().cpu() - zr.detach()).abs().max().item() <= 2e-4 * max(1.0, zr.abs().max().item())
    assert (out[0].detach().cpu() - ref[0].detach()).abs().max().item() <= 2e-4 * max(1.0, ref[0].abs().max().item())
    (A.sum_sq(zh) / zh.numel()).backward()
    zr.pow(2).mean().backward()
    named = net.state_dict(keep_vars=True)          # lists the shared WN under all of its aliases
    for k in ("flow.wn.in_layers.0.depth_conv.weight_v", "flow.wn.in_layers.2.point_conv.weight_g",
              "flow.wn.in_layers.1.depth_conv.bias", "flow.wn.res_skip_layers.3.weight_v",
              "enc_q.enc.in_layers.5.depth_conv.weight_g", "enc_q.enc.in_layers.0.point_conv.weight_v", "enc_q.pre.weight"):
        gh, gr = named[k].grad.cpu(), sg[k].grad
        assert (gh - gr).abs().max().item() <= 2e-3 * max(gr.abs().max().item(), 1e-6), k


def test_tiny_template_true_widths_training_step_matches_oracle(dev):
    """config_tiny_template.json:42-71 at its real widths in the TRAINING graph (hidden 192, filter 512, decoder 200/100/50/25/12,
    spec 1025, segment 16 frames = 8192 samples): forward vectors and the gradients of a loss that reaches every branch — y_hat
    through the odd-width MRF stages, z_p through the depthwise posterior WN and the shared flow WN, pred_lf0 through the F0
    decoder — against the oracle's torch-CPU autograd."""
    import models
    import svc_autograd as A
    from oracle import train_oracle as TO
    from oracle import weights as W
    cfg = dict(W.tiny_config(), p_dropout=0.0)
    B, T, seed = 2, 40, 61
    sd = W.make_train_state_dict(cfg, seed)
    kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
    net = models.SynthesizerTrn(cfg["spec_channels"], cfg["segment_size"], **kw)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).train()
    c, f0, uv, spec, y, sid, lengths = W.make_train_batch(cfg, B, T, seed)
    noise = W.make_train_noise(cfg, B, T, lengths, seed + 2)
    leaves, sg = {}, {}
    for k, v in sd.items():                              # the shared WN appears under five prefixes: one leaf per storage
        key = v.data_ptr()
        if key not in leaves:
            leaves[key] = v.clone().requires_grad_(True)
        sg[k] = leaves[key]
    ref = TO.synth_forward(sg, cfg, c, f0, uv, spec, sid, lengths, lengths, noise)
    out = net(c.to(dev), f0.to(dev), uv.to(dev), spec.to(dev), g=sid.to(dev), c_lengths=lengths.to(dev),
              spec_lengths=lengths.to(dev), noise={k: v.to(dev) for k, v in noise.items()})
    yh, yr = out[0], ref[0]
    zh, zr = out[3][1], ref[3][1]
    ph, pr = out[4], ref[4]
    assert yh.shape == yr.shape == (B, 1, 16 * 512)
    for name, h, r in (("y_hat", yh, yr), ("z_p", zh, zr), ("pred_lf0", ph, pr), ("m_p", out[3][2], ref[3][2])):
        err = (h.detach().cpu() - r.detach()).abs().max().item()
        assert err <= 2e-4 * max(1.0, r.detach().abs().max().item()), (name, err)
    (A.sum_sq(yh) / yh.numel() + A.sum_sq(zh) / zh.numel() + A.sum_sq(ph) / ph.numel()).backward()
    (yr.pow(2).mean() + zr.pow(2).mean() + pr.pow(2).mean()).backward()
    named = net.state_dict(keep_vars=True)
    checked = 0
    for k, p in named.items():
        gr = sg[k].grad
        if gr is None or not p.requires_grad or k.endswith("conv_k.bias"):      # softmax is shift-invariant: d/d(k bias) == 0
            continue
        assert p.grad is not None, k
        gh = p.grad.cpu()
        # norm-wise 2e-3 (the check that catches a wrong tap / offset / scale), element-wise 2e-2 of the tensor's largest entry: the
        # decoder's weight and bias gradients under this loss are sums of 16 k signed terms that cancel to ~1e-4 of their absolute
        # sum, so single elements carry fp32 noise of several 1e-3 of the largest one (measured worst: 7.2e-3 on one element of
        # dec.resblocks.3.convs1.0.weight_v, 2.6e-3 on one of its bias)
        assert (gh - gr).abs().max().item() <= 2e-2 * max(gr.abs().max().item(), 1e-6), k
        assert (gh - gr).norm().item() <= 2e-3 * max(gr.norm().item(), 1e-6), k
        checked += 1
    # every decoder stage (odd widths), the depthwise / pointwise pairs and the shared WN are among the checked tensors
    for k in ("dec.resblocks.14.convs2.2.weight_v", "dec.resblocks.9.convs1.0.weight_g", "dec.ups.4.weight_v", "dec.ups.0.weight_g",
              "dec.noise_convs.3.weight", "flow.wn.in_layers.0.depth_conv.weight_v", "flow.wn.in_layers.3.point_conv.weight_g",
              "enc_q.enc.in_layers.15.depth_conv.weight_v", "enc_q.pre.weight", "f0_decoder.proj.weight"):
        assert named[k].grad is not None and sg[k].grad is not None, k
    assert checked > 400, checked


def test_training_forward_with_dropout_matches_reference(dev):
    """p_dropout = 0.1 (configs_template/config_template.json:49): the HIP training graph with the keep decisions made
    inside svc_attn_softmax_fwd_f32 (attention probabilities, modules/attentions.py:232) and SVC_EW_DROPOUT (attention /
    FFN outputs and FFN hidden activations) on injected uniform draws, against the REAL reference's vectors: prior
    statistics, pred_lf0, loss_kl, loss_lf0 and the gradients of loss_kl + loss_lf0."""
    import models
    import svc_autograd as A
    from modules.losses import kl_loss
    cs = load_dropout_case()
    z = cs["z"]
    cfg = cs["cfg"]
    kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
    net = models.SynthesizerTrn(cfg["spec_channels"], cfg["segment_size"], **kw)
    net.load_state_dict(cs["sd_g"], strict=True)
    net = net.to(dev).train()
    c, f0, uv, spec, y, sid, lengths = [t.to(dev) for t in cs["batch"]]
    noise = {k: ([u.to(dev) for u in v] if isinstance(v, list) else v.to(dev)) for k, v in cs["noise"].items()}
    out = net(c, f0, uv, spec, g=sid, c_lengths=lengths, spec_lengths=lengths, noise=noise)
    y_hat, ids_slice, z_mask, (zq, z_p, m_p, logs_p, m_q, logs_q), pred_lf0, norm_lf0, lf0 = out
    assert np.abs(m_p.detach().cpu().numpy() - z["m_p"]).max() <= 2e-4 * max(1.0, np.abs(z["m_p"]).max())
    assert np.abs(logs_p.detach().cpu().numpy() - z["logs_p"]).max() <= 2e-4 * max(1.0, np.abs(z["logs_p"]).max())
    assert np.abs(pred_lf0.detach().cpu().numpy() - z["pred_lf0"]).max() <= 2e-4 * max(1.0, np.abs(z["pred_lf0"]).max())
    loss_kl = kl_loss(z_p, logs_q, m_p, logs_p, z_mask)
    loss_lf0 = A.sum_sq_diff(pred_lf0, lf0) / lf0.numel()
    assert abs(float(loss_kl) - float(z["loss_kl"])) <= 1e-4 * max(1.0, abs(float(z["loss_kl"])))
    assert abs(float(loss_lf0) - float(z["loss_lf0"])) <= 1e-4 * max(1.0, abs(float(z["loss_lf0"])))
    (loss_kl + loss_lf0).backward()
    gg = {k: p.grad.detach().cpu() for k, p in net.named_parameters() if p.grad is not None}
    for k, n in zip([str(k) for k in z["gnorm_keys"]], z["gnorm"]):
        if k.endswith("conv_k.bias"):
            continue
        assert k in gg, k
        assert abs(gg[k].norm().item() - n) <= 2e-3 * max(n, 1e-5), (k, gg[k].norm().item(), n)
    for name in z.files:
        if name.startswith("grad."):
            g = gg[name[5:]].numpy()
            assert np.abs(g - z[name]).max() <= 1e-3 * max(np.abs(z[name]).max(), 1e-6), name
    # without injected draws the module draws its own masks (torch.rand on the device): the statistics differ
    n2 = dict(noise)
    n2.pop("dropout_u")
    with torch.no_grad():
        m2 = net(c, f0, uv, spec, g=sid, c_lengths=lengths, spec_lengths=lengths, noise=n2)[3][2]
    assert (m2 - m_p).abs().max().item() > 1e-3


def test_transformer_flow_training_matches_reference(dev):
    """use_transformer_flow = True (models.py:438-439: TransformerCouplingBlock, FFT(isflow=True) coupling networks) in the
    HIP training graph with p_dropout = 0.1 and injected draws, against the REAL reference's vectors: z_p = flow(z), the
    prior statistics, loss_kl + loss_lf0 and their gradients (norms of all, the flow's cond_pre / cond_layer / attention /
    FFN parameters in full)."""
    import models
    import svc_autograd as A
    from modules.losses import kl_loss
    cs = load_transflow_case()
    z = cs["z"]
    cfg = cs["cfg"]
    kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
    net = models.SynthesizerTrn(cfg["spec_channels"], cfg["segment_size"], **kw)
    net.load_state_dict(cs["sd_g"], strict=True)
    net = net.to(dev).train()
    c, f0, uv, spec, y, sid, lengths = [t.to(dev) for t in cs["batch"]]
    noise = {k: ([u.to(dev) for u in v] if isinstance(v, list) else v.to(dev)) for k, v in cs["noise"].items()}
    out = net(c, f0, uv, spec, g=sid, c_lengths=lengths, spec_lengths=lengths, noise=noise)
    y_hat, ids_slice, z_mask, (zq, z_p, m_p, logs_p, m_q, logs_q), pred_lf0, norm_lf0, lf0 = out
    assert np.abs(z_p.detach().cpu().numpy() - z["z_p"]).max() <= 2e-4 * max(1.0, np.abs(z["z_p"]).max())
    assert np.abs(m_p.detach().cpu().numpy() - z["m_p"]).max() <= 2e-4 * max(1.0, np.abs(z["m_p"]).max())
    loss_kl = kl_loss(z_p, logs_q, m_p, logs_p, z_mask)
    loss_lf0 = A.sum_sq_diff(pred_lf0, lf0) / lf0.numel()
    assert abs(float(loss_kl) - float(z["loss_kl"])) <= 1e-4 * max(1.0, abs(float(z["loss_kl"])))
    assert abs(float(loss_lf0) - float(z["loss_lf0"])) <= 1e-4 * max(1.0, abs(float(z["loss_lf0"])))
    (loss_kl + loss_lf0).backward()
    gg = {k: p.grad.detach().cpu() for k, p in net.named_parameters() if p.grad is not None}
    for k, n in zip([str(k) for k in z["gnorm_keys"]], z["gnorm"]):
        if k.endswith("conv_k.bias"):
            continue
        assert k in gg, k
        assert abs(gg[k].norm().item() - n) <= 2e-3 * max(n, 1e-5), (k, gg[k].norm().item(), n)
    for name in z.files:
        if name.startswith("grad."):
            g = gg[name[5:]].numpy()
            assert np.abs(g - z[name]).max() <= 1e-3 * max(np.abs(z[name]).max(), 1e-6), name


def test_attention_dropout_op_matches_torch(dev):
    """svc_autograd.attention with injected dropout draws vs a plain torch fp32 restatement of
    modules/attentions.py:207-239 (window 4, padding mask), forward and all five gradients."""
    import math
    import svc_autograd as A
    g = torch.Generator().manual_seed(3)
    B, H, dk, T, w, p = 2, 2, 24, 37, 4, 0.25
    q, k, v = [torch.randn(B, H * dk, T, generator=g).to(dev).requires_grad_(True) for _ in range(3)]
    ek, ev = [(torch.randn(1, 2 * w + 1, dk, generator=g) * 0.2).to(dev).requires_grad_(True) for _ in range(2)]
    u = torch.rand(B, H, T, T, generator=g).to(dev)
    lens = torch.tensor([T, T - 9])
    mask = (torch.arange(T)[None] < lens[:, None]).float().to(dev)
    out = A.attention(q, k, v, H, ek, ev, w, mask, 1, drop_u=u, p_drop=p)
    go = torch.randn(B, H * dk, T, generator=g).to(dev)
    out.backward(go)
    got = [t.grad.clone() for t in (q, k, v, ek, ev)]

    def ref(q, k, v, ek, ev):
        qh = q.view(B, H, dk, T).transpose(2, 3) / math.sqrt(dk)
        kh = k.view(B, H, dk, T).transpose(2, 3)
        vh = v.view(B, H, dk, T).transpose(2, 3)
        sc = qh @ kh.transpose(-2, -1)
        rel = qh @ ek[0].t()                                       # [B,H,T,2w+1]
        idx = torch.arange(T, device=dev)
        band = idx[None, :] - idx[:, None]                        # j - i
        inb = band.abs() <= w
        ii = idx[:, None].expand(T, T)[inb]
        jj = idx[None, :].expand(T, T)[inb]
        rr = (band + w)[inb]
        relfull = torch.zeros_like(sc)
        relfull[:, :, ii, jj] = rel[:, :, ii, rr]
        sc = sc + relfull
        am = mask[:, None, :, None] * mask[:, None, None, :]
        sc = sc.masked_fill(am == 0, -1e4)
        pa = torch.softmax(sc, -1) * ((u >= p).float() * (1.0 / (1.0 - p)))
        o = pa @ vh
        pb = torch.zeros(B, H, T, 2 * w + 1, device=dev)
        pb[:, :, ii, rr] = pa[:, :, ii, jj]
        o = o + pb @ ev[0]
        return o.transpose(2, 3).contiguous().view(B, H * dk, T)

    rq, rk, rv, rek, rev = [t.detach().clone().requires_grad_(True) for t in (q, k, v, ek, ev)]
    ro = ref(rq, rk, rv, rek, rev)
    assert (out - ro).abs().max().item() <= 2e-5 * max(1.0, ro.abs().max().item())
    ro.backward(go)
    for a, b, name in zip(got, (rq, rk, rv, rek, rev), "q k v emb_k emb_v".split()):
        assert (a - b.grad).abs().max().item() <= 2e-4 * max(1.0, b.grad.abs().max().item()), name


def test_training_step_at_the_benchmarked_shapes(dev):
    """BASELINE configs[2] shapes as bench.py builds them — full template (p_dropout 0.1 ACTIVE), segment_size 8192, hop 512,
    n_fft 2048, 80 mels, items of 300..790 frames from bench.make_train_items — on the first FOUR items of the benchmark's
    minibatch (the CPU oracle needs ~10 s per item for one step; the losses are batch means, so the comparison has to run on
    the same items): the seven losses, y_hat and the gradient norms of a parameter sample against the oracle's CPU autograd,
    every random draw (incl. the 48 dropout sites) injected."""
    import bench
    import models
    import svc_autograd as A
    import modules.commons as commons
    from modules.losses import discriminator_loss, feature_loss, generator_loss, kl_loss
    from modules.mel_processing import mel_spectrogram_torch, spec_to_mel_torch
    from oracle import mel as OM
    from oracle import train_oracle as TO
    from oracle import weights as W
    cfg = W.full_config()
    assert cfg["p_dropout"] == 0.1
    hps = bench.train_hps(cfg)
    (c, f0, spec, y, spk, lengths, uv, _), _ = bench.make_train_items(cfg, bench.TRAIN_B, 4321)
    n = 4
    lengths = lengths[:n]
    T = int(lengths.max())
    c, f0, spec, uv, y, spk = c[:n, :, :T], f0[:n, :T], spec[:n, :, :T], uv[:n, :T], y[:n, :, :T * bench.HOP], spk[:n]
    assert T >= 600
    d = hps["data"]
    data = dict(n_fft=d["filter_length"], hop=d["hop_length"], win=d["win_length"], n_mels=d["n_mel_channels"],
                sr=d["sampling_rate"], fmin=d["mel_fmin"], fmax=d["mel_fmax"])
    sd_g = W.make_train_state_dict(cfg, 1234)
    sd_d = W.make_mpd_state_dict(1235)
    noise = W.make_train_noise(cfg, n, T, lengths, 7, hop=bench.HOP)
    noise["dropout_u"] = W.make_dropout_draws(cfg, n, T, 8)
    mb = torch.from_numpy(OM.mel_filterbank(data["sr"], data["n_fft"], data["n_mels"], data["fmin"], data["fmax"]))
    probe_g = ["pre.weight", "enc_p.enc_.attn_layers.3.conv_q.weight", "enc_p.enc_.ffn_layers.5.conv_2.weight",
               "f0_decoder.decoder.self_attn_layers.2.conv_v.weight", "flow.flows.2.enc.in_layers.1.weight_v",
               "dec.ups.1.weight_v", "dec.resblocks.7.convs1.1.weight_v", "dec.conv_post.weight_v", "enc_q.enc.in_layers.9.weight_v"]
    probe_d = ["discriminators.0.convs.3.weight_v", "discriminators.2.convs.4.weight_v", "discriminators.5.convs.2.weight_v"]
    sg = {k: v.clone().requires_grad_(k in probe_g) for k, v in sd_g.items()}
    sdd = {k: v.clone().requires_grad_(k in probe_d) for k, v in sd_d.items()}
    ref = TO.gan_step_losses(sg, sdd, cfg, data, (c, f0, uv, spec, y, spk, lengths), noise, mb)
    rg_d = torch.autograd.grad(ref["loss_disc"], [sdd[k] for k in probe_d], retain_graph=True)
    rg_g = torch.autograd.grad(ref["loss_gen_all"], [sg[k] for k in probe_g])

    kw = {k: v for k, v in cfg.items() if k not in ("spec_channels", "segment_size")}
    net_g = models.SynthesizerTrn(cfg["spec_channels"], hps["train"]["segment_size"] // bench.HOP, **kw)
    net_g.load_state_dict(sd_g, strict=True)
    net_d = models.MultiPeriodDiscriminator()
    net_d.load_state_dict(sd_d, strict=True)
    net_g, net_d = net_g.to(dev).train(), net_d.to(dev).train()
    nz = {k: ([u.to(dev) for u in v] if isinstance(v, list) else v.to(dev)) for k, v in noise.items()}
    cD, f0D, uvD, specD, yD, spkD, lenD = [t.to(dev) for t in (c, f0, uv, spec, y, spk, lengths)]
    seg, hop = cfg["segment_size"], bench.HOP
    mel = spec_to_mel_torch(specD, data["n_fft"], data["n_mels"], data["sr"], data["fmin"], data["fmax"])
    y_hat, ids_slice, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), pred_lf0, norm_lf0, lf0 = net_g(
        cD, f0D, uvD, specD, g=spkD, c_lengths=lenD, spec_lengths=lenD, noise=nz)
    y_mel = commons.slice_segments(mel, ids_slice, seg)
    y_hat_mel = mel_spectrogram_torch(y_hat.squeeze(1), data["n_fft"], data["n_mels"], data["sr"], hop, data["win"], data["fmin"], data["fmax"])
    y_seg = commons.slice_segments(yD, ids_slice * hop, seg * hop)
    rs, gs, _, _ = net_d(y_seg, y_hat.detach())
    loss_disc, _, _ = discriminator_loss(rs, gs)
    rs, gs, fr, fg = net_d(y_seg, y_hat)
    out = dict(loss_disc=loss_disc, loss_mel=A.sum_abs_diff(y_mel, y_hat_mel) / y_mel.numel() * 45.0,
               loss_kl=kl_loss(z_p, logs_q, m_p, logs_p, z_mask) * 1.0, loss_fm=feature_loss(fr, fg),
               loss_gen=generator_loss(gs)[0], loss_lf0=A.sum_sq_diff(pred_lf0, lf0) / lf0.numel())
    out["loss_gen_all"] = out["loss_gen"] + out["loss_fm"] + out["loss_mel"] + out["loss_kl"] + out["loss_lf0"]
    assert y_hat.shape == (n, 1, 8192)
    yr = ref["y_hat"].detach()
    assert (y_hat.detach().cpu() - yr).abs().max().item() <= 2e-4 * max(1.0, yr.abs().max().item())
    assert (y_hat.detach().cpu() - yr).pow(2).mean().item() < 1e-4
    for k in LOSS_KEYS:
        r = float(ref[k])
        assert abs(float(out[k]) - r) <= 2e-4 * max(1.0, abs(r)), (k, float(out[k]), r)
    out["loss_disc"].backward(retain_graph=True)
    pd = dict(net_d.named_parameters())
    for k, g in zip(probe_d, rg_d):
        assert abs(pd[k].grad.norm().item() - g.norm().item()) <= 5e-4 * max(g.norm().item(), 1e-6), ("D", k)
    net_d.zero_grad()
    out["loss_gen_all"].backward()
    pg = dict(net_g.named_parameters())
    for k, g in zip(probe_g, rg_g):
        assert abs(pg[k].grad.norm().item() - g.norm().item()) <= 5e-4 * max(g.norm().item(), 1e-6), ("G", k, pg[k].grad.norm().item(), g.norm().item())
