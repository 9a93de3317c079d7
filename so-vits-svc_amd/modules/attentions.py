"""MI355X-native mirror of the reference's modules/attentions.py (Encoder, FFT, MultiHeadAttention, FFN).

Per layer the reference launches ~25 aten ops (4 projections, 2 T x T matmuls + 2 padded relative-position matmuls
with pad/reshape skews, softmax, masked_fill, 2 convs, 2 transposed layer norms...).  Here a layer is 7 kernels:
fused qkv projection (one MFMA conv, Cout = 3C), flash-style attention with the window-4 relative terms inside,
output projection, residual+LayerNorm, FFN conv1 (+mask+ReLU), FFN conv2 (+masks), residual+LayerNorm (+final mask).
"""
import math
import os

import torch
from torch import nn

import svc_autograd as A
import svc_hip as S
from svc_nn import Conv1d, _no_grad_guard, mask2d, training_call
from modules.modules import LayerNorm

MASK_NONE, MASK_PADDING, MASK_CAUSAL = 0, 1, 2
# training-time epilogue fusions (FFN: ReLU and masks in the conv epilogues); SVC_FUSED_TRAIN=0: one autograd op per reference op
FUSED_TRAIN = os.environ.get("SVC_FUSED_TRAIN", "1") != "0"


class MultiHeadAttention(nn.Module):
    def __init__(self, channels, out_channels, n_heads, p_dropout=0., window_size=None, heads_share=True,
                 block_length=None, proximal_bias=False, proximal_init=False):
        super().__init__()
        assert channels % n_heads == 0
        self.channels = channels
        self.out_channels = out_channels
        self.n_heads = n_heads
        self.p_dropout = p_dropout
        self.window_size = window_size
        self.heads_share = heads_share
        self.block_length = block_length
        self.proximal_bias = proximal_bias
        self.proximal_init = proximal_init
        self.attn = None
        if block_length is not None or proximal_bias or not heads_share:
            raise NotImplementedError("block_length / proximal_bias / per-head relative embeddings are never enabled "
                                      "on the so-vits-svc path (modules/attentions.py:33,91)")
        self.k_channels = channels // n_heads
        self.conv_q = Conv1d(channels, channels, 1)
        self.conv_k = Conv1d(channels, channels, 1)
        self.conv_v = Conv1d(channels, channels, 1)
        self.conv_o = Conv1d(channels, out_channels, 1)
        if window_size is not None:
            rel_stddev = self.k_channels ** -0.5
            self.emb_rel_k = nn.Parameter(torch.randn(1, window_size * 2 + 1, self.k_channels) * rel_stddev)
            self.emb_rel_v = nn.Parameter(torch.randn(1, window_size * 2 + 1, self.k_channels) * rel_stddev)
        nn.init.xavier_uniform_(self.conv_q.weight)
        nn.init.xavier_uniform_(self.conv_k.weight)
        nn.init.xavier_uniform_(self.conv_v.weight)
        if proximal_init:
            with torch.no_grad():
                self.conv_k.weight.copy_(self.conv_q.weight)
                self.conv_k.bias.copy_(self.conv_q.bias)

    # fused q|k|v projection weight, cached on the three parameter versions
    def _qkv_packed(self):
        ps = (self.conv_q.weight, self.conv_k.weight, self.conv_v.weight, self.conv_q.bias, self.conv_k.bias,
              self.conv_v.bias)
        key = tuple((p.data_ptr(), p._version, str(p.device)) for p in ps)
        hit = self.__dict__.get("_qkv_cache")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                w = torch.cat([self.conv_q.weight, self.conv_k.weight, self.conv_v.weight], 0)
                b = torch.cat([self.conv_q.bias, self.conv_k.bias, self.conv_v.bias], 0)
                self.__dict__["_qkv_cache"] = (key, (S.pack_conv1d_weight(w), b.contiguous()))
        return self.__dict__["_qkv_cache"][1]

    def _mask_mode(self, attn_mask, t):
        """Classify an explicit [B,1,T,T] reference-style mask (host sync; Encoder/FFT pass hints instead)."""
        if attn_mask is None:
            return MASK_NONE, None
        am = attn_mask
        if bool((am != 0).all()):
            return MASK_NONE, None
        tril = torch.tril(torch.ones(t, t, device=am.device))
        if am.shape[0] == 1 and bool(((am[0, 0] != 0) == (tril != 0)).all()):
            return MASK_CAUSAL, None
        vec = torch.diagonal(am[:, 0], dim1=-2, dim2=-1).contiguous().float()
        outer = vec.unsqueeze(1) * vec.unsqueeze(2)
        if bool(((am[:, 0] != 0) == (outer != 0)).all()):
            return MASK_PADDING, vec
        raise NotImplementedError("attention mask is neither all-ones, causal nor an outer product of a padding mask")

    def forward_train(self, x, mask_mode, mask_vec, draws=None):
        """Reference modules/attentions.py:198-239: separate q/k/v projections, attention as GEMMs + masked softmax, and
        `p_attn = self.drop(p_attn)` (:232) fused into the softmax kernel (`draws`: the DropoutDraws of this pass;
        without one the module draws for itself when it is in training mode with p_dropout > 0)."""
        win = self.window_size or 0
        if draws is None:
            draws = DropoutDraws(self.p_dropout, self.training)
        B, C, T = x.shape
        u = draws.u((B, self.n_heads, T, T), x.device) if draws.active else None
        if QKV_FUSED_TRAIN and self.conv_q.kernel_size == 1 and not self.conv_q.is_weight_norm and self.conv_q.bias is not None:
            # the three 1 x 1 projections as ONE 3C-row convolution whose rows are ordered (head, {q, k, v}, d): x has one consumer
            # (no accumulation of three input gradients), one input-gradient and one weight-gradient launch instead of three each,
            # and the attention op reads / writes q, k, v and their gradients in place of that tensor (svc_autograd._AttentionQKV).
            # The stack is an index copy of the three parameters; its backward hands each its rows of the fused gradient.
            w = A.stack_qkv(self.conv_q.weight, self.conv_k.weight, self.conv_v.weight, self.n_heads)       # [3C, C, 1]
            b = A.stack_qkv(self.conv_q.bias, self.conv_k.bias, self.conv_v.bias, self.n_heads)             # [3C]
            qkv = A.conv1d(x, w, b)
            att = A.attention_qkv(qkv, self.n_heads, self.emb_rel_k if win else None, self.emb_rel_v if win else None, win,
                                  mask_vec, mask_mode, drop_u=u, p_drop=draws.p)
            return self.conv_o.forward_train(att)
        q = self.conv_q.forward_train(x)
        k = self.conv_k.forward_train(x)
        v = self.conv_v.forward_train(x)
        att = A.attention(q, k, v, self.n_heads, self.emb_rel_k if win else None, self.emb_rel_v if win else None, win,
                          mask_vec, mask_mode, drop_u=u, p_drop=draws.p)
        return self.conv_o.forward_train(att)

    def forward(self, x, c, attn_mask=None, mask_mode=None, mask_vec=None):
        if c is not x:
            raise NotImplementedError("only self-attention is used on the so-vits-svc path")
        if training_call(self.conv_q.weight) or (torch.is_grad_enabled() and x.requires_grad):
            if mask_mode is None:
                mask_mode, mask_vec = self._mask_mode(attn_mask, x.shape[2])
            return self.forward_train(x, mask_mode, mask_vec)
        _no_grad_guard(self.conv_q.weight)
        B, C, T = x.shape
        if mask_mode is None:
            mask_mode, mask_vec = self._mask_mode(attn_mask, T)
        wp, b = self._qkv_packed()
        qkv = S.conv1d(x, wp, 3 * C, 1, bias=b)
        win = self.window_size or 0
        att = S.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], self.n_heads,
                          emb_rel_k=self.emb_rel_k[0] if win else None, emb_rel_v=self.emb_rel_v[0] if win else None,
                          window=win, mask=mask_vec, mask_mode=mask_mode)
        return self.conv_o.run(att)


class FFN(nn.Module):
    def __init__(self, in_channels, out_channels, filter_channels, kernel_size, p_dropout=0., activation=None,
                 causal=False):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.filter_channels = filter_channels
        self.kernel_size = kernel_size
        self.p_dropout = p_dropout
        self.activation = activation
        self.causal = causal
        if activation == "gelu":
            raise NotImplementedError("gelu FFN is never enabled on the so-vits-svc path")
        self.conv_1 = Conv1d(in_channels, filter_channels, kernel_size)
        self.conv_2 = Conv1d(filter_channels, out_channels, kernel_size)

    def forward_train(self, x, x_mask, drop=None):
        """Reference modules/attentions.py:337-345."""
        k = self.kernel_size
        if k % 2 == 0 and not self.causal:
            raise NotImplementedError("even FFN kernel sizes with 'same' padding are not on the so-vits-svc path")
        if drop is None:
            drop = DropoutDraws(self.p_dropout, self.training)
        if FUSED_TRAIN and self.conv_1.fused_train_ok() and self.conv_2.fused_train_ok():
            # ReLU and the `* x_mask` in front of conv_2 (:343-345) in conv_1's epilogue — the mask commutes with the dropout
            # between them (both multiply element-wise, the mask by 0 / 1) —, the final mask in conv_2's
            h = self.conv_1.forward_train(A.mul_bcast(x, x_mask), causal=self.causal, padding=(k - 1) // 2, mask=x_mask,
                                          post_act=S.ACT_RELU)
            h = drop(h)
            return self.conv_2.forward_train(h, causal=self.causal, padding=(k - 1) // 2, mask=x_mask)
        h = self.conv_1.forward_train(A.mul_bcast(x, x_mask), causal=self.causal, padding=(k - 1) // 2)
        h = A.relu(h)
        h = drop(h)                                                             # :344
        h = self.conv_2.forward_train(A.mul_bcast(h, x_mask), causal=self.causal, padding=(k - 1) // 2)
        return A.mul_bcast(h, x_mask)

    def forward(self, x, x_mask, x_is_masked=False):
        """`x_is_masked`: the caller already multiplied x by x_mask (the LayerNorm epilogue does); the `* x_mask` in front
        of conv_2 (reference :341) then moves into conv_1's epilogue and neither conv needs a pre-mask on its input — which
        is what lets them take the vectorised / LDS-DMA staging paths."""
        if training_call(self.conv_1.weight) or (torch.is_grad_enabled() and x.requires_grad):
            return self.forward_train(x, x_mask)
        k = self.kernel_size
        pad_l = 0 if k == 1 else (k - 1 if self.causal else (k - 1) // 2)
        m = mask2d(x_mask)
        T = x.shape[2]
        if x_is_masked:
            h = self.conv_1.run(x, pad_left=pad_l, Tout=T, post_act=S.ACT_RELU, mask=m)
            return self.conv_2.run(h, pad_left=pad_l, Tout=T, mask=m)
        h = self.conv_1.run(x, premask=m, pad_left=pad_l, Tout=T, post_act=S.ACT_RELU)
        return self.conv_2.run(h, premask=m, pad_left=pad_l, Tout=T, mask=m)


class Encoder(nn.Module):
    def __init__(self, hidden_channels, filter_channels, n_heads, n_layers, kernel_size=1, p_dropout=0., window_size=4,
                 **kwargs):
        super().__init__()
        self.hidden_channels = hidden_channels
        self.filter_channels = filter_channels
        self.n_heads = n_heads
        self.n_layers = n_layers
        self.kernel_size = kernel_size
        self.p_dropout = p_dropout
        self.window_size = window_size
        self.attn_layers = nn.ModuleList()
        self.norm_layers_1 = nn.ModuleList()
        self.ffn_layers = nn.ModuleList()
        self.norm_layers_2 = nn.ModuleList()
        for _ in range(n_layers):
            self.attn_layers.append(MultiHeadAttention(hidden_channels, hidden_channels, n_heads, p_dropout=p_dropout,
                                                       window_size=window_size))
            self.norm_layers_1.append(LayerNorm(hidden_channels))
            self.ffn_layers.append(FFN(hidden_channels, hidden_channels, filter_channels, kernel_size,
                                       p_dropout=p_dropout))
            self.norm_layers_2.append(LayerNorm(hidden_channels))

    def forward_train(self, x, x_mask, dropout_u=None):
        """Reference modules/attentions.py:95-107.  `dropout_u`: optional list of injected uniform draws, one per dropout
        site in the reference's order (parity tests); consumed from the front."""
        return _encoder_forward_train(self, x, x_mask, self.attn_layers, self.norm_layers_1, self.norm_layers_2, MASK_PADDING,
                                      dropout_u)

    def forward(self, x, x_mask, full_mask=False):
        """`full_mask=True` promises x_mask is all ones (inference, models.py:503) and skips the padding mask."""
        if training_call(self.norm_layers_1[0].gamma) or (torch.is_grad_enabled() and x.requires_grad):
            return self.forward_train(x, x_mask)
        m = mask2d(x_mask)
        x = S.copy_bct(x, mask=m)
        mode = MASK_NONE if full_mask else MASK_PADDING
        # every LayerNorm applies x_mask in its epilogue (the reference masks only at the very end, :107: frames outside
        # the mask never influence frames inside it, so the valid frames are unchanged) — the FFN input is then masked
        for i in range(self.n_layers):
            y = self.attn_layers[i](x, x, mask_mode=mode, mask_vec=None if full_mask else m)
            x = self.norm_layers_1[i](x, residual=y, x_mask=x_mask)
            y = self.ffn_layers[i](x, x_mask, x_is_masked=True)
            x = self.norm_layers_2[i](x, residual=y, x_mask=x_mask)
        return x


# training: q / k / v projections as one convolution (SVC_QKV_FUSED_TRAIN=0: three, one autograd op per reference op — A/B switch)
QKV_FUSED_TRAIN = __import__("os").environ.get("SVC_QKV_FUSED_TRAIN", "1") != "0"


class DropoutDraws:
    """The uniform draws behind every nn.Dropout site of one forward pass.  The Bernoulli keep decision is u >= p, made
    INSIDE the consuming HIP kernel (svc_attn_softmax_fwd_f32 for the attention probabilities, SVC_EW_DROPOUT for the
    activation sites): the only torch call is the draw itself (torch.rand on the device, like every other random tensor
    on the path).  `injected`: a list of draws consumed from the front in the reference's call order (parity tests)."""

    _SEED = {}          # device -> int64[1] running counter (advanced once per DropoutDraws that draws from it)
    HASHED = __import__("os").environ.get("SVC_DROPOUT_HASH", "1") != "0"     # 0: torch.rand tensors (round-4 form, A/B switch)

    def __init__(self, p, training, injected=None):
        self.p = float(p) if training else 0.0
        self.injected = injected
        self.seed, self.site = None, 0

    @property
    def active(self):
        return self.p > 0.0

    def u(self, shape, device):
        if self.injected is not None:
            t = self.injected.pop(0)
            if tuple(t.shape) != tuple(shape):
                raise S.SvcError(f"injected dropout draw {tuple(t.shape)} does not match the site's shape {tuple(shape)}")
            return t.to(device=device, dtype=torch.float32).contiguous()
        if not self.HASHED:
            return torch.rand(shape, device=device)
        # production: no tensor of draws — the consuming kernels evaluate u(seed, site, element) themselves (900 MB of torch.rand
        # output per iteration otherwise, 48 launches).  This stack's seed = a snapshot of the device counter, advanced by this
        # stack only: it stays put between the sites' forward and backward kernels, and a replayed hipGraph re-executes the
        # advance + snapshot, so every replay drops different elements.  (torch.manual_seed does not reach it: seeded runs that
        # must reproduce torch's draws inject them or set SVC_DROPOUT_HASH=0.)
        if self.seed is None:
            ctr = DropoutDraws._SEED.get(device)
            if ctr is None:
                ctr = DropoutDraws._SEED[device] = torch.randint(1 << 40, (1,), dtype=torch.int64).to(device)
            ctr.add_(0x632BE59BD9B4E019 % (1 << 62))
            self.seed = ctr.clone()
        self.site += 1
        return S.HashDraw(self.seed, self.site)

    def __call__(self, x):
        if not self.active:
            return x
        return A.dropout(x, self.u(tuple(x.shape), x.device), self.p)


def _encoder_forward_train(self, x, x_mask, attn_layers, norm_a, norm_b, mask_mode, dropout_u=None, pre_layer=None):
    """Encoder / FFT training forward (modules/attentions.py:43-70,95-107) with all four dropout sites of a layer in the
    reference's order: attention probabilities (:232), attention output (:51/:100), FFN hidden (:344), FFN output.
    `pre_layer(i, x)`: the conditioning gate in front of each layer of a flow FFT (:49-56)."""
    m = x_mask[:, 0].contiguous() if x_mask.dim() == 3 else x_mask
    draws = DropoutDraws(self.p_dropout, self.training, dropout_u)
    x = A.mul_bcast(x, x_mask)
    for i in range(self.n_layers):
        if pre_layer is not None:
            x = pre_layer(i, x)
        y = attn_layers[i].forward_train(x, mask_mode, m if mask_mode == MASK_PADDING else None, draws=draws)
        y = draws(y)
        x = norm_a[i](A.add(x, y))
        y = self.ffn_layers[i].forward_train(x, x_mask, drop=draws)
        y = draws(y)
        x = norm_b[i](A.add(x, y))
    return A.mul_bcast(x, x_mask)


class FFT(nn.Module):
    def __init__(self, hidden_channels, filter_channels, n_heads, n_layers=1, kernel_size=1, p_dropout=0.,
                 proximal_bias=False, proximal_init=True, isflow=False, **kwargs):
        super().__init__()
        self.hidden_channels = hidden_channels
        self.filter_channels = filter_channels
        self.n_heads = n_heads
        self.n_layers = n_layers
        self.kernel_size = kernel_size
        self.p_dropout = p_dropout
        self.proximal_bias = proximal_bias
        self.proximal_init = proximal_init
        self.isflow = isflow
        if isflow:
            # modules/attentions.py:24-28: conditioning of a coupling network (use_transformer_flow): the speaker embedding
            # goes through a weight-normed 1x1 `cond_layer` once, and ONE shared 1x1 `cond_pre` (H -> 2H) runs in front of
            # every layer, gated by that layer's slice of the conditioning (fused_add_tanh_sigmoid_multiply)
            self.gin_channels = kwargs["gin_channels"]
            self.cond_pre = Conv1d(hidden_channels, 2 * hidden_channels, 1)
            self.cond_layer = Conv1d(self.gin_channels, 2 * hidden_channels * n_layers, 1, weight_norm=True)
        self.self_attn_layers = nn.ModuleList()
        self.norm_layers_0 = nn.ModuleList()
        self.ffn_layers = nn.ModuleList()
        self.norm_layers_1 = nn.ModuleList()
        for _ in range(n_layers):
            self.self_attn_layers.append(MultiHeadAttention(hidden_channels, hidden_channels, n_heads,
                                                            p_dropout=p_dropout, proximal_bias=proximal_bias,
                                                            proximal_init=proximal_init))
            self.norm_layers_0.append(LayerNorm(hidden_channels))
            self.ffn_layers.append(FFN(hidden_channels, hidden_channels, filter_channels, kernel_size,
                                       p_dropout=p_dropout, causal=True))
            self.norm_layers_1.append(LayerNorm(hidden_channels))

    def _check_g(self, g):
        if (g is not None) != self.isflow:
            raise S.SvcError("FFT: the conditioning `g` is given exactly when the module was built with isflow=True")

    def forward_train(self, x, x_mask, dropout_u=None, g=None):
        """Reference modules/attentions.py:43-70; with `g` the per-layer conditioning gate of :49-56."""
        self._check_g(g)
        pre = None
        if g is not None:
            H = self.hidden_channels
            gcs = A.chunk_channels(self.cond_layer.forward_train(g), self.n_layers, views=True)  # L x [B, 2H, 1]: one backward buffer
            pre = lambda i, x: A.gate(A.add_bcast(self.cond_pre.forward_train(x), gcs[i]))
        return _encoder_forward_train(self, x, x_mask, self.self_attn_layers, self.norm_layers_0, self.norm_layers_1,
                                      MASK_CAUSAL, dropout_u, pre_layer=pre)

    def forward(self, x, x_mask, g=None, dropout_u=None):
        if training_call(self.norm_layers_0[0].gamma) or (torch.is_grad_enabled() and x.requires_grad):
            return self.forward_train(x, x_mask, dropout_u=dropout_u, g=g)
        self._check_g(g)
        m = mask2d(x_mask)
        x = S.copy_bct(x, mask=m)
        H = self.hidden_channels
        gc = self.cond_layer(g) if g is not None else None
        for i in range(self.n_layers):
            if gc is not None:      # cond_pre + conditioning + tanh*sigmoid gate: one MFMA conv with the gate epilogue
                x = self.cond_pre.run(x, cond=gc[:, i * 2 * H:(i + 1) * 2 * H], epi=S.EPI_GATE)
            y = self.self_attn_layers[i](x, x, mask_mode=MASK_CAUSAL)
            x = self.norm_layers_0[i](x, residual=y, x_mask=x_mask)
            y = self.ffn_layers[i](x, x_mask, x_is_masked=True)
            x = self.norm_layers_1[i](x, residual=y, x_mask=x_mask)
        return x
