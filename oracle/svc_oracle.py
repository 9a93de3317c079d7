"""CPU oracle for the so-vits-svc SynthesizerTrn hot path.

TEST INFRASTRUCTURE ONLY.  This is a functional, torch-CPU fp32 restatement of the reference's algorithm: it is
what tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg check the HIP path against (or time beside
it).  Nothing under so-vits-svc_amd/ may import it, and it is never a fallback for the product path.

Pinning: every function here is checked against the real reference (imported from /root/reference in the build
container by tests/golden/make_golden.py) and the resulting vectors are committed under tests/golden/; the
`-m "not gpu"` suite re-checks the oracle against those vectors.  Citations are path:line under /root/reference.

All functions take the reference's own `state_dict` (a {name: tensor} mapping with weight_g/weight_v pairs kept,
SURVEY.md §8b) and explicit noise tensors instead of drawing from the global RNG (precedent:
onnxexport/model_onnx_speaker_mix.py:334).
"""
import math

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # vdecoder/hifigan/models.py:14, modules/modules.py:14


# ------------------------------------------------------------------------------------------------------------
# small helpers
# ------------------------------------------------------------------------------------------------------------
def weight_of(sd, prefix):
    """Effective conv weight.  torch.nn.utils.weight_norm (dim=0): w = v * (g / ||v||), the norm taken over all
    dims except 0 — for Conv1d that is per output channel, for ConvTranspose1d per INPUT channel
    (vdecoder/hifigan/models.py:340-342)."""
    if prefix + ".weight" in sd:
        return sd[prefix + ".weight"]
    v = sd[prefix + ".weight_v"]
    g = sd[prefix + ".weight_g"]
    norm = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return v * (g / norm)


def conv1d(x, sd, prefix, **kw):
    return F.conv1d(x, weight_of(sd, prefix), sd.get(prefix + ".bias"), **kw)


def get_padding(kernel_size, dilation=1):  # modules/commons.py:33-34
    return int((kernel_size * dilation - dilation) / 2)


def sequence_mask(lengths, max_len):  # modules/commons.py:144-148
    return torch.arange(max_len, dtype=lengths.dtype)[None, :] < lengths[:, None]


def f0_to_coarse(f0):
    """utils.py:69-80: mel-scale f0 -> integer bin in [1, 255]."""
    f0_bin, f0_max, f0_min = 256, 1100.0, 50.0
    mel_min = 1127 * math.log(1 + f0_min / 700)
    mel_max = 1127 * math.log(1 + f0_max / 700)
    f0_mel = 1127 * (1 + f0 / 700).log()
    a = (f0_bin - 2) / (mel_max - mel_min)
    b = mel_min * a - 1.
    f0_mel = torch.where(f0_mel > 0, f0_mel * a - b, f0_mel)
    c = torch.round(f0_mel).long()
    c = c * (c > 0)
    c = c + ((c < 1) * 1)
    c = c * (c < f0_bin)
    c = c + ((c >= f0_bin) * (f0_bin - 1))
    return c


def normalize_f0(lf0, x_mask, uv, factor=None):
    """utils.py:31-45 with the random scale made explicit (`factor` [B,1]; None -> 1 as at inference)."""
    uv_sum = torch.sum(uv, dim=1, keepdim=True)
    uv_sum[uv_sum == 0] = 9999
    means = torch.sum(lf0[:, 0, :] * uv, dim=1, keepdim=True) / uv_sum
    if factor is None:
        factor = torch.ones(lf0.shape[0], 1)
    return (lf0 - means.unsqueeze(-1)) * factor.unsqueeze(-1) * x_mask


def layer_norm_c(x, gamma, beta, eps=1e-5):  # modules/modules.py:23-35 (LN over the channel dim of [B,C,T])
    return F.layer_norm(x.transpose(1, -1), (x.shape[1],), gamma, beta, eps).transpose(1, -1)


# ------------------------------------------------------------------------------------------------------------
# attention stack (modules/attentions.py)
# ------------------------------------------------------------------------------------------------------------
class DropSeq:
    """nn.Dropout(p) in training mode with the uniform draws explicit: site k keeps where us[k] >= p and scales by 1/(1-p)
    (torch's dropout multiplies by a Bernoulli(1-p) mask pre-divided by 1-p).  `us` is consumed in call order; p == 0 or
    us is None = identity (eval mode)."""

    def __init__(self, p, us):
        self.p = float(p)
        self.us = list(us) if us is not None else None

    def __call__(self, x):
        if self.us is None or self.p <= 0:
            return x
        u = self.us.pop(0)
        assert tuple(u.shape) == tuple(x.shape), (u.shape, x.shape)
        keep = (u >= self.p).to(x.dtype) * (1.0 / (1.0 - self.p))
        return x * keep


_NO_DROP = DropSeq(0.0, None)


def multi_head_attention(x, sd, prefix, attn_mask, n_heads, window_size=None, drop=_NO_DROP):
    """MultiHeadAttention.forward/attention, modules/attentions.py:198-239, self-attention case.

    The reference realises window-`w` relative positions by padding emb_rel_k/v to 2T-1 rows and skewing
    (:259-303).  Restated in the equivalent banded form: with r = j - i, |r| <= w,
        scores[i,j] += (q_i / sqrt(d)) . E_k[r + w]       out_i += sum_r p[i, i+r] * E_v[r + w]
    (E shared across heads, heads_share=True :161).  Masked scores are set to -1e4 (:231), not -inf."""
    q = conv1d(x, sd, prefix + ".conv_q")
    k = conv1d(x, sd, prefix + ".conv_k")
    v = conv1d(x, sd, prefix + ".conv_v")
    e_k = sd[prefix + ".emb_rel_k"][0] if window_size is not None else None
    e_v = sd[prefix + ".emb_rel_v"][0] if window_size is not None else None
    out = attention_core(q, k, v, attn_mask, n_heads, e_k, e_v, window_size, drop)
    return conv1d(out, sd, prefix + ".conv_o")


def attention_core(q, k, v, attn_mask, n_heads, e_k=None, e_v=None, window_size=None, drop=_NO_DROP):
    """MultiHeadAttention.attention, modules/attentions.py:207-239, on projected q,k,v [B, H*dk, T]."""
    b, d, t = q.shape
    kc = d // n_heads
    q = q.view(b, n_heads, kc, t).transpose(2, 3)
    k = k.view(b, n_heads, kc, t).transpose(2, 3)
    v = v.view(b, n_heads, kc, t).transpose(2, 3)
    qs = q / math.sqrt(kc)
    scores = torch.matmul(qs, k.transpose(-2, -1))
    if window_size is not None:
        rel = torch.matmul(qs, e_k.t())              # [b,h,t,2w+1]
        idx = torch.arange(t)
        for m in range(2 * window_size + 1):
            r = m - window_size
            i = idx[(idx + r >= 0) & (idx + r < t)]
            scores[:, :, i, i + r] = scores[:, :, i, i + r] + rel[:, :, i, m]
    if attn_mask is not None:
        scores = scores.masked_fill(attn_mask == 0, -1e4)
    p = drop(F.softmax(scores, dim=-1))          # :232 p_attn = self.drop(p_attn)
    out = torch.matmul(p, v)
    if window_size is not None:
        idx = torch.arange(t)
        for m in range(2 * window_size + 1):
            r = m - window_size
            i = idx[(idx + r >= 0) & (idx + r < t)]
            out[:, :, i, :] = out[:, :, i, :] + p[:, :, i, i + r].unsqueeze(-1) * e_v[m]
    return out.transpose(2, 3).contiguous().view(b, d, t)


def ffn(x, x_mask, sd, prefix, kernel_size, causal=False, drop=_NO_DROP):  # modules/attentions.py:337-363
    if kernel_size == 1:
        pad = (0, 0)
    elif causal:
        pad = (kernel_size - 1, 0)
    else:
        pad = ((kernel_size - 1) // 2, kernel_size // 2)
    h = conv1d(F.pad(x * x_mask, pad), sd, prefix + ".conv_1")
    h = drop(torch.relu(h))
    h = conv1d(F.pad(h * x_mask, pad), sd, prefix + ".conv_2")
    return h * x_mask


def attn_encoder(x, x_mask, sd, prefix, n_layers, n_heads, kernel_size, window_size=4, drop=_NO_DROP):
    """attentions.Encoder.forward, modules/attentions.py:95-107 (post-LN; `drop`: the DropSeq of a training pass)."""
    attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    x = x * x_mask
    for i in range(n_layers):
        y = drop(multi_head_attention(x, sd, f"{prefix}.attn_layers.{i}", attn_mask, n_heads, window_size, drop))
        x = layer_norm_c(x + y, sd[f"{prefix}.norm_layers_1.{i}.gamma"], sd[f"{prefix}.norm_layers_1.{i}.beta"])
        y = drop(ffn(x, x_mask, sd, f"{prefix}.ffn_layers.{i}", kernel_size, drop=drop))
        x = layer_norm_c(x + y, sd[f"{prefix}.norm_layers_2.{i}.gamma"], sd[f"{prefix}.norm_layers_2.{i}.beta"])
    return x * x_mask


def fft_decoder(x, x_mask, sd, prefix, n_layers, n_heads, kernel_size, drop=_NO_DROP, g=None):
    """attentions.FFT.forward, modules/attentions.py:43-70: causal self-attention, causal FFN.  With `g` (isflow=True,
    :24-28,49-61): g -> weight-normed cond_layer once, and in front of every layer the SHARED 1x1 `cond_pre` (H -> 2H)
    followed by fused_add_tanh_sigmoid_multiply with that layer's 2H slice of the conditioning."""
    t = x.shape[2]
    causal = torch.tril(torch.ones(t, t)).unsqueeze(0).unsqueeze(0)
    if g is not None:
        g = conv1d(g, sd, prefix + ".cond_layer")
    x = x * x_mask
    for i in range(n_layers):
        if g is not None:
            hidden = x.shape[1]
            x_in = conv1d(x, sd, prefix + ".cond_pre") + g[:, i * 2 * hidden:(i + 1) * 2 * hidden, :]
            x = torch.tanh(x_in[:, :hidden]) * torch.sigmoid(x_in[:, hidden:])        # modules/commons.py:129-136
        y = drop(multi_head_attention(x, sd, f"{prefix}.self_attn_layers.{i}", causal, n_heads, None, drop))
        x = layer_norm_c(x + y, sd[f"{prefix}.norm_layers_0.{i}.gamma"], sd[f"{prefix}.norm_layers_0.{i}.beta"])
        y = drop(ffn(x, x_mask, sd, f"{prefix}.ffn_layers.{i}", kernel_size, causal=True, drop=drop))
        x = layer_norm_c(x + y, sd[f"{prefix}.norm_layers_1.{i}.gamma"], sd[f"{prefix}.norm_layers_1.{i}.beta"])
    return x * x_mask


def text_encoder(x, x_mask, f0_coarse, sd, cfg, noise, noice_scale=1.0, prefix="enc_p", drop=_NO_DROP):
    """TextEncoder.forward, models.py:155-162."""
    x = x + sd[prefix + ".f0_emb.weight"][f0_coarse].transpose(1, 2)
    x = attn_encoder(x * x_mask, x_mask, sd, prefix + ".enc_", cfg["n_layers"], cfg["n_heads"], cfg["kernel_size"], drop=drop)
    stats = conv1d(x, sd, prefix + ".proj") * x_mask
    m, logs = torch.split(stats, cfg["inter_channels"], dim=1)
    z = (m + noise * torch.exp(logs) * noice_scale) * x_mask
    return z, m, logs


def f0_decoder(x, norm_f0, x_mask, g, sd, cfg, prefix="f0_decoder", drop=_NO_DROP):
    """F0Decoder.forward, models.py:328-336."""
    x = x + conv1d(g, sd, prefix + ".cond")
    x = x + conv1d(norm_f0, sd, prefix + ".f0_prenet", padding=1)
    x = conv1d(x, sd, prefix + ".prenet", padding=1) * x_mask
    x = fft_decoder(x * x_mask, x_mask, sd, prefix + ".decoder", cfg["n_layers"], cfg["n_heads"],
                    cfg["kernel_size"], drop=drop)
    return conv1d(x, sd, prefix + ".proj") * x_mask


# ------------------------------------------------------------------------------------------------------------
# WaveNet block + coupling flows (modules/modules.py)
# ------------------------------------------------------------------------------------------------------------
def wn(x, x_mask, g, sd, prefix, hidden, kernel_size, dilation_rate, n_layers):
    """WN.forward, modules/modules.py:110-138 (p_dropout=0); in_layers are dense Conv1d's or, with
    use_depthwise_conv (modules/modules.py:16-20,95), depthwise-separable pairs."""
    output = torch.zeros_like(x)
    if g is not None:
        g = conv1d(g, sd, prefix + ".cond_layer")
    for i in range(n_layers):
        dilation = dilation_rate ** i
        padding = int((kernel_size * dilation - dilation) / 2)
        ip = f"{prefix}.in_layers.{i}"
        if ip + ".depth_conv.weight_v" in sd:
            # use_depthwise_conv: Depthwise_Separable_Conv1D (modules/DSConv.py:5-27) = depthwise k-tap conv + 1x1 conv
            d = F.conv1d(x, weight_of(sd, ip + ".depth_conv"), sd[ip + ".depth_conv.bias"], dilation=dilation,
                         padding=padding, groups=x.shape[1])
            x_in = conv1d(d, sd, ip + ".point_conv")
        else:
            x_in = conv1d(x, sd, ip, dilation=dilation, padding=padding)
        if g is not None:
            x_in = x_in + g[:, i * 2 * hidden:(i + 1) * 2 * hidden, :]
        acts = torch.tanh(x_in[:, :hidden]) * torch.sigmoid(x_in[:, hidden:])  # modules/commons.py:129-136
        rs = conv1d(acts, sd, f"{prefix}.res_skip_layers.{i}")
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * x_mask
            output = output + rs[:, hidden:]
        else:
            output = output + rs
    return output * x_mask


def coupling_layer(x, x_mask, g, sd, prefix, wn_prefix, cfg, reverse):
    """ResidualCouplingLayer.forward with mean_only=True, modules/modules.py:288-307."""
    half = cfg["inter_channels"] // 2
    x0, x1 = torch.split(x, [half, half], 1)
    h = conv1d(x0, sd, prefix + ".pre") * x_mask
    h = wn(h, x_mask, g, sd, wn_prefix, cfg["hidden_channels"], 5, 1, cfg.get("n_flow_layer", 4))
    m = conv1d(h, sd, prefix + ".post") * x_mask
    logs = torch.zeros_like(m)
    if not reverse:
        x1 = m + x1 * torch.exp(logs) * x_mask
    else:
        x1 = (x1 - m) * torch.exp(-logs) * x_mask
    return torch.cat([x0, x1], 1)


def transformer_coupling_layer(x, x_mask, g, sd, prefix, enc_prefix, cfg, reverse, drop=_NO_DROP):
    """TransformerCouplingLayer.forward with mean_only=True, modules/modules.py:337-356: the coupling network is the
    conditioned FFT (attentions.FFT(isflow=True)) instead of WN."""
    half = cfg["inter_channels"] // 2
    x0, x1 = torch.split(x, [half, half], 1)
    h = conv1d(x0, sd, prefix + ".pre") * x_mask
    h = fft_decoder(h, x_mask, sd, enc_prefix, cfg.get("n_layers_trans_flow", 3), cfg["n_heads"], 5, drop=drop, g=g)
    m = conv1d(h, sd, prefix + ".post") * x_mask
    if not reverse:
        x1 = m + x1 * x_mask
    else:
        x1 = (x1 - m) * x_mask
    return torch.cat([x0, x1], 1)


def flow(x, x_mask, g, sd, cfg, reverse, prefix="flow", drop=_NO_DROP):
    """ResidualCouplingBlock.forward, models.py:45-52: flows = [coupling, Flip] * n_flows, Flip = torch.flip over
    channels (modules/modules.py:232-239).  Note models.py:441 passes n_flow_layer as the WN depth (positional n_layers);
    n_flows stays at its default 4.  With use_transformer_flow (models.py:438-439, TransformerCouplingBlock :54-92) the
    couplings are TransformerCouplingLayers, n_flow_layer IS the number of flows, and the shared network (if any) is
    `flow.wn` as well."""
    trans = cfg.get("use_transformer_flow", False)
    n_flows = cfg.get("n_flow_layer", 4) if trans else 4
    share = cfg.get("flow_share_parameter", False)
    order = range(n_flows) if not reverse else reversed(range(n_flows))

    def couple(x, cp, rev):
        enc_prefix = f"{prefix}.wn" if share else cp + ".enc"
        if trans:
            return transformer_coupling_layer(x, x_mask, g, sd, cp, enc_prefix, cfg, rev, drop)
        return coupling_layer(x, x_mask, g, sd, cp, enc_prefix, cfg, rev)

    for i in order:
        cp = f"{prefix}.flows.{2 * i}"
        if not reverse:
            x = torch.flip(couple(x, cp, False), [1])
        else:
            x = couple(torch.flip(x, [1]), cp, True)
    return x


def posterior_encoder(spec, spec_mask, g, sd, cfg, noise, prefix="enc_q"):
    """Encoder.forward (posterior), models.py:117-125: pre 1x1, WN(k5, 16 layers), proj, reparameterise."""
    x = conv1d(spec, sd, prefix + ".pre") * spec_mask
    x = wn(x, spec_mask, g, sd, prefix + ".enc", cfg["hidden_channels"], 5, 1, 16)
    stats = conv1d(x, sd, prefix + ".proj") * spec_mask
    m, logs = torch.split(stats, cfg["inter_channels"], dim=1)
    z = (m + noise * torch.exp(logs)) * spec_mask
    return z, m, logs


# ------------------------------------------------------------------------------------------------------------
# NSF-HiFiGAN generator (vdecoder/hifigan/models.py)
# ------------------------------------------------------------------------------------------------------------
def sine_source(f0_up, sd, rand_ini, noise_sine, sampling_rate, prefix="dec.m_source", harmonic_num=8,
                sine_amp=0.1, noise_std=0.003):
    """SineGen._f02sine/forward + SourceModuleHnNSF.forward, vdecoder/hifigan/models.py:138-166,250-271,307-320.
    f0_up: [B, L, 1] (already nearest-upsampled); rand_ini: [B, 9] with column 0 ignored; noise_sine: [B, L, 9].
    Returns har_source [B, 1, L]."""
    harm = torch.arange(1, harmonic_num + 2, dtype=torch.float32).view(1, 1, -1)
    fn = f0_up * harm
    rad = (fn / sampling_rate) % 1
    ri = rand_ini.clone()
    ri[:, 0] = 0
    rad[:, 0, :] = rad[:, 0, :] + ri
    tmp_over_one = torch.cumsum(rad, 1) % 1
    over_idx = (tmp_over_one[:, 1:, :] - tmp_over_one[:, :-1, :]) < 0   # padDiff(...) < 0, models.py:100-101,161
    shift = torch.zeros_like(rad)
    shift[:, 1:, :] = over_idx * -1.0
    sines = torch.sin(torch.cumsum(rad + shift, dim=1) * 2 * math.pi) * sine_amp
    uv = (f0_up > 0).float()
    noise_amp = uv * noise_std + (1 - uv) * sine_amp / 3
    sine_waves = sines * uv + noise_amp * noise_sine
    merged = torch.tanh(F.linear(sine_waves, sd[prefix + ".l_linear.weight"], sd[prefix + ".l_linear.bias"]))
    return merged.transpose(1, 2)


def resblock1(x, sd, prefix, kernel_size, dilations):  # vdecoder/hifigan/models.py:60-67
    for j, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = conv1d(xt, sd, f"{prefix}.convs1.{j}", dilation=d, padding=get_padding(kernel_size, d))
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = conv1d(xt, sd, f"{prefix}.convs2.{j}", dilation=1, padding=get_padding(kernel_size, 1))
        x = xt + x
    return x


def resblock2(x, sd, prefix, kernel_size, dilations):  # vdecoder/hifigan/models.py:88-93
    for j, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = conv1d(xt, sd, f"{prefix}.convs.{j}", dilation=d, padding=get_padding(kernel_size, d))
        x = xt + x
    return x


def snake_alias(x, sd, prefix):
    """SnakeAlias.forward, vdecoder/hifiganwithsnake/alias/act.py:125-130: UpSample1d (alias/resample.py:38-54) ->
    SnakeBeta log-scale (alias/act.py:79-92) -> DownSample1d / LowPassFilter1d (alias/filter.py:93-110)."""
    C = x.shape[1]
    fu = sd[prefix + ".upsample.filter"].expand(C, -1, -1)
    fd = sd[prefix + ".downsample.lowpass.filter"].expand(C, -1, -1)
    x = F.pad(x, (5, 5), mode="replicate")
    x = 2 * F.conv_transpose1d(x, fu, stride=2, groups=C)
    x = x[..., 15:-15]
    a = torch.exp(sd[prefix + ".act.alpha"])[None, :, None]
    b = torch.exp(sd[prefix + ".act.beta"])[None, :, None]
    x = x + (1.0 / (b + 1e-9)) * torch.sin(x * a) ** 2
    x = F.pad(x, (5, 6), mode="replicate")
    return F.conv1d(x, fd, stride=2, groups=C)


def resblock1_snake(x, sd, prefix, kernel_size, dilations):  # vdecoder/hifiganwithsnake/models.py:71-77
    for j, d in enumerate(dilations):
        xt = snake_alias(x, sd, f"{prefix}.activations.{2 * j}")
        xt = conv1d(xt, sd, f"{prefix}.convs1.{j}", dilation=d, padding=get_padding(kernel_size, d))
        xt = snake_alias(xt, sd, f"{prefix}.activations.{2 * j + 1}")
        xt = conv1d(xt, sd, f"{prefix}.convs2.{j}", dilation=1, padding=get_padding(kernel_size, 1))
        x = xt + x
    return x


def resblock2_snake(x, sd, prefix, kernel_size, dilations):  # vdecoder/hifiganwithsnake/models.py:101-106
    for j, d in enumerate(dilations):
        xt = snake_alias(x, sd, f"{prefix}.activations.{j}")
        xt = conv1d(xt, sd, f"{prefix}.convs.{j}", dilation=d, padding=get_padding(kernel_size, d))
        x = xt + x
    return x


def generator(x, f0, g, sd, cfg, rand_ini, noise_sine, prefix="dec", return_source=False):
    """hifigan.Generator.forward, vdecoder/hifigan/models.py:366-394 (and the snake variant,
    vdecoder/hifiganwithsnake/models.py:380-413, when cfg["vocoder_name"] == "nsf-snake-hifigan").
    x [B,inter,T], f0 [B,T], g [B,gin,1]."""
    snake = cfg.get("vocoder_name") == "nsf-snake-hifigan"
    ups = cfg["upsample_rates"]
    upp = int(math.prod(ups))
    f0_up = f0[:, None].repeat_interleave(upp, dim=2).transpose(1, 2)   # nn.Upsample(nearest), :369
    har = sine_source(f0_up, sd, rand_ini, noise_sine, cfg.get("sampling_rate", 44100), prefix + ".m_source")
    x = conv1d(x, sd, prefix + ".conv_pre", padding=3)
    x = x + conv1d(g, sd, prefix + ".cond")
    nk = len(cfg["resblock_kernel_sizes"])
    if snake:
        rb = resblock1_snake if cfg["resblock"] == "1" else resblock2_snake
    else:
        rb = resblock1 if cfg["resblock"] == "1" else resblock2
    for i, (u, k) in enumerate(zip(ups, cfg["upsample_kernel_sizes"])):
        x = snake_alias(x, sd, f"{prefix}.snakes.{i}") if snake else F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, weight_of(sd, f"{prefix}.ups.{i}"), sd[f"{prefix}.ups.{i}.bias"], stride=u,
                               padding=(k - u + 1) // 2)
        if i + 1 < len(ups):
            s = int(math.prod(ups[i + 1:]))
            xs = conv1d(har, sd, f"{prefix}.noise_convs.{i}", stride=s, padding=(s + 1) // 2)
        else:
            xs = conv1d(har, sd, f"{prefix}.noise_convs.{i}")
        x = x + xs
        acc = None
        for j, (kk, dd) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            r = rb(x, sd, f"{prefix}.resblocks.{i * nk + j}", kk, dd)
            acc = r if acc is None else acc + r
        x = acc / nk
    x = snake_alias(x, sd, prefix + ".snake_post") if snake else F.leaky_relu(x)   # default slope 0.01, :390
    x = conv1d(x, sd, prefix + ".conv_post", padding=3)
    x = torch.tanh(x)
    return (x, har) if return_source else x


# ------------------------------------------------------------------------------------------------------------
# whole-path inference (models.py:496-532)
# ------------------------------------------------------------------------------------------------------------
def synth_infer(sd, cfg, c, f0, uv, sid, noise, noice_scale=0.35, predict_f0=False, vol=None, return_all=False,
                g_mix=None):
    """SynthesizerTrn.infer.  c [B,ssl,T], f0 [B,T], uv [B,T], sid [B,1] int64.
    g_mix [T, S] (B = 1): the character-mix branch models.py:505-509 after EnableCharacterMix(S) (:456-461) — a
    per-frame convex mix of the first S speaker embeddings, i.e. a time-varying g [1, gin, T].
    noise: dict(enc_p [B,inter,T], rand_ini [B,9], sine [B,L,9]) — the reference's draw order is
    randn_like (models.py:160), rand (vdecoder/hifigan/models.py:147), randn_like (:266), randn_like (:319, unused)."""
    B, _, T = c.shape
    if g_mix is not None:
        S_ = g_mix.shape[1]
        g = (g_mix @ sd["emb_g.weight"][:S_]).t().unsqueeze(0)       # [1, gin, T]
    else:
        g = sd["emb_g.weight"][sid].transpose(1, 2)                  # models.py:513
    x_mask = torch.ones(B, 1, T)
    volp = 0
    if vol is not None and cfg.get("vol_embedding", False):
        volp = F.linear(vol[:, :, None], sd["emb_vol.weight"], sd["emb_vol.bias"]).transpose(1, 2)
    x = conv1d(c, sd, "pre", padding=2) * x_mask + sd["emb_uv.weight"][uv.long()].transpose(1, 2) + volp
    if predict_f0 and cfg.get("use_automatic_f0_prediction", True):
        lf0 = 2595. * torch.log10(1. + f0.unsqueeze(1) / 700.) / 500
        norm_lf0 = normalize_f0(lf0, x_mask, uv)
        pred_lf0 = f0_decoder(x, norm_lf0, x_mask, g, sd, cfg)
        f0 = (700 * (torch.pow(10, pred_lf0 * 500 / 2595) - 1)).squeeze(1)
    z_p, m_p, logs_p = text_encoder(x, x_mask, f0_to_coarse(f0), sd, cfg, noise["enc_p"], noice_scale)
    z = flow(z_p, x_mask, g, sd, cfg, reverse=True)
    o, har = generator(z * x_mask, f0, g, sd, cfg, noise["rand_ini"], noise["sine"], return_source=True)
    if return_all:
        return dict(o=o, f0=f0, x=x, z_p=z_p, m_p=m_p, logs_p=logs_p, z=z, har=har)
    return o, f0
