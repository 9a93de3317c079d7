"""Deterministic synthetic checkpoints and inputs for the SynthesizerTrn path — a DATA generator (no reference algorithm
lives here): used by the oracle and the tests (through `oracle.weights`, which re-exports it) and by bench.py /
__graft_entry__.smoke() to build the weights and inputs they run the HIP engine on.

The reference ships no trained checkpoint, and its 52 M-parameter state_dict is too big to commit,
so parity tests and bench.py build the SAME weights on any machine from (config, seed): every tensor is drawn from
its own torch.Generator seeded with crc32(name) ^ seed, so the values do not depend on iteration order, on the
reference being importable, or on the device.  `param_shapes` restates the reference's state_dict layout
(models.py:344-454); tests/golden/make_golden.py asserts it equals the real
SynthesizerTrn(...).state_dict() key-for-key and shape-for-shape, and commits the key list as a golden.

Scales are fan-in normalised (not the reference's init) so activations stay O(1) through the whole stack: the
reference's default init (N(0, 0.01) decoder convs, zero `post`) yields ~1e-3-amplitude output, which would make
an absolute MSE < 1e-4 parity bound vacuous.
"""
import math
import re
import zlib

import torch


def full_config():
    """model section of configs_template/config_template.json (:42-71) + the data fields the path needs."""
    return dict(
        spec_channels=1025, segment_size=16,
        inter_channels=192, hidden_channels=192, filter_channels=768, n_heads=2, n_layers=6, kernel_size=3,
        p_dropout=0.1, resblock="1", resblock_kernel_sizes=[3, 7, 11],
        resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], upsample_rates=[8, 8, 2, 2, 2],
        upsample_initial_channel=512, upsample_kernel_sizes=[16, 16, 4, 4, 4], n_layers_q=3,
        n_layers_trans_flow=3, n_flow_layer=4, use_spectral_norm=False, gin_channels=768, ssl_dim=768,
        n_speakers=200, vocoder_name="nsf-hifigan", speech_encoder="vec768l12", speaker_embedding=False,
        vol_embedding=False, use_depthwise_conv=False, flow_share_parameter=False,
        use_automatic_f0_prediction=True, use_transformer_flow=False, sampling_rate=44100)


def tiny_config():
    """model section of configs_template/config_tiny_template.json (:42-71): filter_channels 512,
    upsample_initial_channel 400 (decoder channels 200/100/50/25/12), depthwise-separable WN in_layers, one WN shared by
    the four flows."""
    c = full_config()
    c.update(filter_channels=512, upsample_initial_channel=400, use_depthwise_conv=True, flow_share_parameter=True)
    return c


def small_tiny_config():
    """small_config with the tiny template's structural switches (odd decoder widths 100/50/25/12/6, depthwise WN,
    shared flow WN): a few-MB state_dict for committed goldens."""
    c = small_config()
    c.update(upsample_initial_channel=200, use_depthwise_conv=True, flow_share_parameter=True)
    return c


def small_config():
    """A channel-reduced config with the SAME structure (5 upsample stages x 3 MRF kernels x 3 dilations, 2 heads,
    window-4 attention, 4 flows) whose whole state_dict is a few MB: used for committed end-to-end goldens."""
    c = full_config()
    c.update(inter_channels=64, hidden_channels=64, filter_channels=128, n_layers=2, upsample_initial_channel=128,
             gin_channels=32, ssl_dim=48, n_speakers=4, spec_channels=65)
    return c


def param_shapes(cfg, include_enc_q=True, include_f0_decoder=True):
    h, inter, filt = cfg["hidden_channels"], cfg["inter_channels"], cfg["filter_channels"]
    gin, k, nl, nh = cfg["gin_channels"], cfg["kernel_size"], cfg["n_layers"], cfg["n_heads"]
    kc = h // nh
    P = {}

    def conv(name, cout, cin, ks, wn=False, bias=True, transposed=False):
        shape = (cin, cout, ks) if transposed else (cout, cin, ks)
        if bias:
            P[name + ".bias"] = (cout,)
        if wn:
            P[name + ".weight_g"] = (shape[0], 1, 1)
            P[name + ".weight_v"] = shape
        else:
            P[name + ".weight"] = shape

    def attn_stack(prefix, attn_name, norm_a, norm_b, window, n_layers=None, k=k):
        for i in range(nl if n_layers is None else n_layers):
            a = f"{prefix}.{attn_name}.{i}"
            if window:
                P[a + ".emb_rel_k"] = (1, 9, kc)
                P[a + ".emb_rel_v"] = (1, 9, kc)
            for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
                conv(f"{a}.{n}", h, h, 1)
            P[f"{prefix}.{norm_a}.{i}.gamma"] = (h,)
            P[f"{prefix}.{norm_a}.{i}.beta"] = (h,)
            conv(f"{prefix}.ffn_layers.{i}.conv_1", filt, h, k)
            conv(f"{prefix}.ffn_layers.{i}.conv_2", h, filt, k)
            P[f"{prefix}.{norm_b}.{i}.gamma"] = (h,)
            P[f"{prefix}.{norm_b}.{i}.beta"] = (h,)

    def wn_block(prefix, n_layers):
        conv(prefix + ".cond_layer", 2 * h * n_layers, gin, 1, wn=True)
        for i in range(n_layers):
            if cfg.get("use_depthwise_conv"):     # Depthwise_Separable_Conv1D (modules/DSConv.py:5-27), weight-normed
                conv(f"{prefix}.in_layers.{i}.depth_conv", h, 1, 5, wn=True)
                conv(f"{prefix}.in_layers.{i}.point_conv", 2 * h, h, 1, wn=True)
            else:
                conv(f"{prefix}.in_layers.{i}", 2 * h, h, 5, wn=True)
            conv(f"{prefix}.res_skip_layers.{i}", 2 * h if i < n_layers - 1 else h, h, 1, wn=True)

    P["emb_g.weight"] = (cfg["n_speakers"], gin)
    if cfg.get("vol_embedding"):
        P["emb_vol.weight"] = (h, 1)
        P["emb_vol.bias"] = (h,)
    conv("pre", h, cfg["ssl_dim"], 5)
    conv("enc_p.proj", 2 * inter, h, 1)
    P["enc_p.f0_emb.weight"] = (256, h)
    attn_stack("enc_p.enc_", "attn_layers", "norm_layers_1", "norm_layers_2", window=True)

    # decoder (vdecoder/hifigan/models.py:323-361)
    P["dec.m_source.l_linear.weight"] = (1, 9)
    P["dec.m_source.l_linear.bias"] = (1,)
    ups, uks, c0 = cfg["upsample_rates"], cfg["upsample_kernel_sizes"], cfg["upsample_initial_channel"]
    conv("dec.conv_pre", c0, inter, 7, wn=True)
    nk = len(cfg["resblock_kernel_sizes"])
    ch = c0
    for i, (u, ks) in enumerate(zip(ups, uks)):
        cin, ch = c0 // (2 ** i), c0 // (2 ** (i + 1))
        conv(f"dec.ups.{i}", ch, cin, ks, wn=True, transposed=True)
        if i + 1 < len(ups):
            s = int(math.prod(ups[i + 1:]))
            conv(f"dec.noise_convs.{i}", ch, 1, s * 2)
        else:
            conv(f"dec.noise_convs.{i}", ch, 1, 1)
        for j, (kk, dd) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            rb = f"dec.resblocks.{i * nk + j}"
            for m in range(len(dd)):
                if cfg["resblock"] == "1":
                    conv(f"{rb}.convs1.{m}", ch, ch, kk, wn=True)
                    conv(f"{rb}.convs2.{m}", ch, ch, kk, wn=True)
                else:
                    conv(f"{rb}.convs.{m}", ch, ch, kk, wn=True)
    conv("dec.conv_post", 1, ch, 7, wn=True)
    conv("dec.cond", c0, gin, 1)
    if cfg.get("vocoder_name") == "nsf-snake-hifigan":
        # SnakeAlias sites (vdecoder/hifiganwithsnake/models.py:62-65,95-99,367,374): per site alpha/beta [C] parameters
        # and the two [1,1,12] `filter` buffers (alias/resample.py:20, alias/filter.py:83)
        def snake(name, c):
            P[name + ".act.alpha"] = (c,)
            P[name + ".act.beta"] = (c,)
            P[name + ".upsample.filter"] = (1, 1, 12)
            P[name + ".downsample.lowpass.filter"] = (1, 1, 12)
        for i in range(len(ups)):
            snake(f"dec.snakes.{i}", c0 // (2 ** i))
            chn = c0 // (2 ** (i + 1))
            for j, dd in enumerate(cfg["resblock_dilation_sizes"]):
                nact = 2 * len(dd) if cfg["resblock"] == "1" else len(dd)
                for a in range(nact):
                    snake(f"dec.resblocks.{i * nk + j}.activations.{a}", chn)
        snake("dec.snake_post", ch)

    if include_enc_q:
        conv("enc_q.pre", h, cfg["spec_channels"], 1)
        wn_block("enc_q.enc", 16)
        conv("enc_q.proj", 2 * inter, h, 1)

    nfl = cfg.get("n_flow_layer", 4)
    if cfg.get("use_transformer_flow"):
        # TransformerCouplingBlock (models.py:54-92, built at :438-439): n_flow_layer couplings, each a conditioned FFT
        # (attentions.FFT(isflow=True), modules/attentions.py:24-28) of n_layers_trans_flow layers, kernel size 5
        ntf = cfg.get("n_layers_trans_flow", 3)

        def flow_net(prefix, _):
            conv(prefix + ".cond_pre", 2 * h, h, 1)
            conv(prefix + ".cond_layer", 2 * h * ntf, gin, 1, wn=True)
            attn_stack(prefix, "self_attn_layers", "norm_layers_0", "norm_layers_1", window=False, n_layers=ntf, k=5)
        n_flows = nfl
    else:
        flow_net, n_flows = wn_block, 4
    if cfg.get("flow_share_parameter"):
        flow_net("flow.wn", nfl)
    for f in range(n_flows):
        fp = f"flow.flows.{2 * f}"
        conv(fp + ".pre", h, inter // 2, 1)
        flow_net(fp + ".enc", nfl)      # with flow_share_parameter these keys alias flow.wn.* (same module registered twice)
        conv(fp + ".post", inter // 2, h, 1)

    if include_f0_decoder and cfg.get("use_automatic_f0_prediction", True):
        conv("f0_decoder.prenet", h, h, 3)
        attn_stack("f0_decoder.decoder", "self_attn_layers", "norm_layers_0", "norm_layers_1", window=False)
        conv("f0_decoder.proj", 1, h, 1)
        conv("f0_decoder.f0_prenet", h, 1, 3)
        conv("f0_decoder.cond", h, gin, 1)
    P["emb_uv.weight"] = (2, h)
    return P


def snake_filter():
    """The 12-tap Kaiser-sinc half-band low-pass of SnakeAlias (alias/filter.py:29-58 with cutoff 0.25, half_width 0.3,
    kernel_size 12), restated: beta from the Kaiser attenuation formula, unit DC gain."""
    half = 6
    atten = 2.285 * (half - 1) * math.pi * (4 * 0.3) + 7.95
    beta = 0.1102 * (atten - 8.7) if atten > 50 else (0.5842 * (atten - 21) ** 0.4 + 0.07886 * (atten - 21) if atten >= 21 else 0.0)
    w = torch.kaiser_window(12, beta=beta, periodic=False)
    t = torch.arange(-half, half) + 0.5
    h = 2 * 0.25 * w * torch.sinc(2 * 0.25 * t)
    return h / h.sum()


def _gen(name, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def make_tensor(name, shape, seed, all_shapes=None):
    g = _gen(name, seed)
    r = lambda *s: torch.randn(*s, generator=g)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "filter":
        return snake_filter().view(shape)
    if leaf == "alpha" or (leaf == "beta" and ".act." in name):
        return 0.3 * r(*shape)          # log-scale: e^alpha in ~[0.5, 2]
    if leaf == "bias":
        return 0.05 * r(*shape)
    if leaf == "gamma":
        return 1.0 + 0.1 * r(*shape)
    if leaf == "beta":
        return 0.1 * r(*shape)
    if leaf in ("emb_rel_k", "emb_rel_v"):
        return r(*shape) * shape[-1] ** -0.5
    if leaf == "weight_v":
        return 0.1 * r(*shape)
    if leaf == "weight_g":
        vshape = all_shapes[name[:-1] + "v"]
        gain = 0.6 if ".convs" in name else 1.0
        if ".ups." in name:  # ConvTranspose1d: [Cin, Cout, K], norm per input channel; stride = K/2 on this path
            cin, cout, ks = vshape
            gain = gain * math.sqrt(cout * (ks // 2) / cin)
        if "conv_post" in name:
            gain = 0.5
        return gain * (1.0 + 0.1 * r(*shape)).abs()
    if leaf == "weight":
        if len(shape) == 3:
            fan_in = shape[1] * shape[2]
            gain = 0.3 if name.endswith("post.weight") else 1.0   # flow `post` (zero-init in the reference)
            if "noise_convs" in name:
                gain = 0.5
            return gain * r(*shape) / math.sqrt(fan_in)
        if name.endswith("l_linear.weight"):
            return r(*shape) * 0.6
        if name.startswith("emb_") or ".f0_emb" in name:
            return 0.5 * r(*shape)
        return r(*shape) / math.sqrt(shape[-1])
    raise KeyError(name)


def make_state_dict(cfg, seed=1234, **kw):
    shapes = param_shapes(cfg, **kw)
    sd = {n: make_tensor(n, s, seed, shapes) for n, s in shapes.items()}
    if cfg.get("flow_share_parameter"):
        # models.py:37,42: one WN registered as flow.wn AND as every flow.flows.N.enc -> the state_dict lists it five times
        for n in list(sd):
            m = re.match(r"flow\.flows\.\d+\.enc\.(.*)", n)
            if m:
                sd[n] = sd["flow.wn." + m.group(1)]
    return sd


def make_inputs(cfg, B, T, seed=1234, unvoiced_frac=0.1):
    """Synthetic (c, f0, uv, sid) as SURVEY.md §8d: c~N(0,1), f0~U(100,400) with unvoiced (=0) runs."""
    g = torch.Generator().manual_seed(seed)
    c = torch.randn(B, cfg["ssl_dim"], T, generator=g)
    f0 = 100 + 300 * torch.rand(B, T, generator=g)
    # unvoiced runs of ~8 frames
    nrun = max(1, int(T * unvoiced_frac / 8))
    for b in range(B):
        starts = torch.randint(0, max(T - 8, 1), (nrun,), generator=g)
        for s in starts.tolist():
            f0[b, s:s + 8] = 0
    uv = (f0 > 0).float()
    sid = torch.randint(0, cfg["n_speakers"], (B, 1), generator=g)
    return c, f0, uv, sid


def make_noise(cfg, B, T, seed=4321):
    g = torch.Generator().manual_seed(seed)
    L = T * int(math.prod(cfg["upsample_rates"]))
    return dict(enc_p=torch.randn(B, cfg["inter_channels"], T, generator=g),
                rand_ini=torch.rand(B, 9, generator=g),
                sine=torch.randn(B, L, 9, generator=g))


# ---------------------------------------------------------------------------------------------------------------
# MultiPeriodDiscriminator (models.py:165-252): fixed architecture, 46,747,132 parameters
# ---------------------------------------------------------------------------------------------------------------
def mpd_param_shapes():
    P = {}

    def nconv(name, shape):
        P[name + ".bias"] = (shape[0],)
        P[name + ".weight_g"] = (shape[0],) + (1,) * (len(shape) - 1)
        P[name + ".weight_v"] = shape
    s = "discriminators.0"
    for i, sh in enumerate([(16, 1, 15), (64, 4, 41), (256, 4, 41), (1024, 4, 41), (1024, 4, 41), (1024, 1024, 5)]):
        nconv(f"{s}.convs.{i}", sh)
    nconv(f"{s}.conv_post", (1, 1024, 3))
    for d in range(1, 6):
        s = f"discriminators.{d}"
        for i, (a, b) in enumerate([(1, 32), (32, 128), (128, 512), (512, 1024), (1024, 1024)]):
            nconv(f"{s}.convs.{i}", (b, a, 5, 1))
        nconv(f"{s}.conv_post", (1, 1024, 3, 1))
    return P


def make_mpd_state_dict(seed=4321):
    shapes = mpd_param_shapes()
    sd = {}
    for name, shape in shapes.items():
        gen = _gen("mpd." + name, seed)
        if name.endswith(".bias"):
            sd[name] = torch.randn(shape, generator=gen) * 0.05
        elif name.endswith(".weight_v"):
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            sd[name] = torch.randn(shape, generator=gen) * (1.0 / math.sqrt(fan_in))
        else:   # weight_g: around the norm of the matching v so the effective weight keeps the fan-in scale
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=gen)
    return sd


def make_mpd_sn_state_dict(seed=4321):
    """MultiPeriodDiscriminator(use_spectral_norm=True): torch.nn.utils.spectral_norm's keys — weight_orig (the weight_v tensor
    of make_mpd_state_dict), bias, and unit-norm weight_u [Cout] / weight_v [rest] buffers."""
    base = make_mpd_state_dict(seed)
    sd = {}
    for name, t in base.items():
        if name.endswith(".bias"):
            sd[name] = t
        elif name.endswith(".weight_v"):
            pre = name[:-len(".weight_v")]
            sd[pre + ".weight_orig"] = t
            gen = _gen("mpd_sn." + pre, seed)
            u = torch.randn(t.shape[0], generator=gen)
            v = torch.randn(t.numel() // t.shape[0], generator=gen)
            sd[pre + ".weight_u"] = u / u.norm()
            sd[pre + ".weight_v"] = v / v.norm()
    return sd


def train_config():
    """small_config with dropout off (so the training graph is deterministic given the injected noise)."""
    c = small_config()
    c.update(p_dropout=0.0, segment_size=8)
    return c


def make_train_batch(cfg, B, T, seed, hop=512):
    """(c, f0, uv, spec, y, sid, lengths) for one training step; lengths vary so the padding masks are exercised."""
    gen = torch.Generator().manual_seed(seed)
    c = torch.randn(B, cfg["ssl_dim"], T, generator=gen)
    f0 = torch.rand(B, T, generator=gen) * 300 + 100
    f0[:, : max(1, T // 10)] = 0
    f0[0, T // 2: T // 2 + 3] = 0
    uv = (f0 > 0).float()
    spec = torch.randn(B, cfg["spec_channels"], T, generator=gen).abs()
    y = (torch.rand(B, 1, T * hop, generator=gen) - 0.5)
    sid = torch.randint(0, cfg["n_speakers"], (B, 1), generator=gen)
    lengths = torch.tensor([T - 3 * i for i in range(B)], dtype=torch.long)
    return c, f0, uv, spec, y, sid, lengths


def make_train_noise(cfg, B, T, lengths, seed, hop=512):
    gen = torch.Generator().manual_seed(seed)
    seg = cfg["segment_size"]
    ids_max = (lengths - seg + 1).float()
    ids_rand = torch.rand(B, generator=gen)
    ids = (ids_rand * ids_max).long()
    return dict(ids_rand=ids_rand, f0_factor=torch.rand(B, 1, generator=gen) * 0.4 + 0.8,
                enc_p=torch.randn(B, cfg["inter_channels"], T, generator=gen),
                enc_q=torch.randn(B, cfg["inter_channels"], T, generator=gen),
                ids_slice=ids, rand_ini=torch.rand(B, 9, generator=gen),
                sine=torch.randn(B, seg * hop, 9, generator=gen))


def make_dropout_draws(cfg, B, T, seed):
    """Uniform [0,1) draws for every active nn.Dropout site of one SynthesizerTrn.forward, in the reference's call order
    (models.py:476 f0_decoder, then :477 enc_p; per attention layer: attention probabilities [B,H,T,T]
    (modules/attentions.py:232), attention output (:51/:100), FFN hidden (:344), FFN output (:55/:104))."""
    gen = torch.Generator().manual_seed(seed)
    H, C, Fc = cfg["n_heads"], cfg["hidden_channels"], cfg["filter_channels"]
    out = []
    sites = ((B, H, T, T), (B, C, T), (B, Fc, T), (B, C, T))
    for _stack in ("f0_decoder", "enc_p"):
        for _ in range(cfg["n_layers"]):
            for shape in sites:
                out.append(torch.rand(shape, generator=gen))
    if cfg.get("use_transformer_flow"):       # models.py:482: the flow's FFT stacks come last (spec frames == unit frames)
        for _ in range(cfg.get("n_flow_layer", 4) * cfg.get("n_layers_trans_flow", 3)):
            for shape in sites:
                out.append(torch.rand(shape, generator=gen))
    return out


def make_train_state_dict(cfg, seed):
    """make_state_dict with the log-variance projections damped (x0.1): with O(1) random `proj` weights exp(-2*logs_p)
    in the KL term (modules/losses.py:52-54) reaches 1e6 and every other loss term / gradient drowns in it."""
    sd = make_state_dict(cfg, seed)
    inter = cfg["inter_channels"]
    for p in ("enc_p.proj", "enc_q.proj"):
        for suffix in (".weight", ".bias"):
            t = sd[p + suffix].clone()
            t[inter:] = t[inter:] * 0.1
            sd[p + suffix] = t
    return sd
