"""MI355X-native mirror of the reference's modules/modules.py: same class names, constructor signatures and
parameter names (state_dict compatible); every forward is a sequence of libsvc_hip.so kernels.

    LayerNorm                modules/modules.py:23-35
    WN                       modules/modules.py:73-146   -> 2 fused kernels per layer (conv+cond+gate, res/skip)
    ResidualCouplingLayer    modules/modules.py:260-307  -> pre, WN, post with the affine update in the epilogue
    Flip                     modules/modules.py:232-239  -> folded into channel strides by ResidualCouplingBlock
    ResBlock1 / ResBlock2    modules/modules.py:149-218  (masked variants; the decoder uses vdecoder.hifigan's)
"""
import os

import torch
from torch import nn

import svc_autograd as A
import svc_hip as S
from modules.commons import get_padding, init_weights
from svc_nn import Conv1d, DepthwiseSeparableConv1d, mask2d, training_call

LRELU_SLOPE = 0.1
# training-time WN layers with fused epilogues (WN.forward_train); SVC_WN_FUSED=0: one autograd op per reference op
WN_FUSED = os.environ.get("SVC_WN_FUSED", "1") != "0"

_use_depthwise_conv = False


def set_Conv1dModel(use_depthwise_conv):
    """Reference modules/modules.py:16-20 switches WN/ResBlock to Depthwise_Separable_Conv1D (tiny config)."""
    global _use_depthwise_conv
    _use_depthwise_conv = bool(use_depthwise_conv)


def _conv(cin, cout, k, **kw):
    """The reference's `Conv1dModel` (modules/modules.py:16-20)."""
    if _use_depthwise_conv:
        return DepthwiseSeparableConv1d(cin, cout, k, **kw)
    return Conv1d(cin, cout, k, **kw)


class LayerNorm(nn.Module):
    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.channels = channels
        self.eps = eps
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))

    def forward(self, x, residual=None, x_mask=None):
        """LN over the channel dim of [B,C,T]; `residual` (added first) and `x_mask` (applied last) are fusions."""
        if training_call(self.gamma, self.beta) or (torch.is_grad_enabled() and x.requires_grad):
            if residual is not None:
                x = A.add(x, residual)
            y = A.layer_norm(x, self.gamma, self.beta, self.eps)
            return A.mul_bcast(y, x_mask) if x_mask is not None else y
        return S.add_layernorm(x.contiguous(), None if residual is None else residual.contiguous(), self.gamma,
                               self.beta, mask=mask2d(x_mask), eps=self.eps)


class WN(nn.Module):
    def __init__(self, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0):
        super().__init__()
        assert kernel_size % 2 == 1
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size,
        self.dilation_rate = dilation_rate
        self.n_layers = n_layers
        self.gin_channels = gin_channels
        self.p_dropout = p_dropout
        if p_dropout != 0:
            raise NotImplementedError("WN dropout is not used on the so-vits-svc path (p_dropout=0)")
        self.in_layers = nn.ModuleList()
        self.res_skip_layers = nn.ModuleList()
        if gin_channels != 0:
            self.cond_layer = Conv1d(gin_channels, 2 * hidden_channels * n_layers, 1, weight_norm=True)
        for i in range(n_layers):
            dilation = dilation_rate ** i
            padding = int((kernel_size * dilation - dilation) / 2)
            self.in_layers.append(_conv(hidden_channels, 2 * hidden_channels, kernel_size, dilation=dilation,
                                        padding=padding, weight_norm=True))
            rs = 2 * hidden_channels if i < n_layers - 1 else hidden_channels
            self.res_skip_layers.append(Conv1d(hidden_channels, rs, 1, weight_norm=True))

    def forward_train(self, x, x_mask, g=None):
        """Reference modules/modules.py:110-138, one autograd op per reference op."""
        H = self.hidden_channels
        gc = self.cond_layer.forward_train(g) if g is not None else None          # [B, 2H*L, 1|T]
        # the per-layer conditioning rows as ONE op: its backward writes the L gradients into one buffer (a Python slice per
        # layer costs a zero-fill + copy + accumulate launch each in torch's slice backward: 36 WN layers per iteration)
        gcs = A.chunk_channels(gc, self.n_layers, views=True) if gc is not None else None
        output = None
        # fused form (default; SVC_WN_FUSED=0 = one autograd op per reference op): the conditioning add in in_layer's epilogue,
        # res / skip / mask in res_skip_layer's (the inference path's epilogues, with their adjoints in svc_autograd)
        fused = WN_FUSED and all(l.fused_train_ok() for l in self.res_skip_layers) and \
            all(getattr(l, "fused_train_ok", lambda: False)() for l in self.in_layers)
        for i in range(self.n_layers):
            last = i == self.n_layers - 1
            if fused:
                cond = gcs[i] if gcs is not None else None
                acts = A.gate(self.in_layers[i].forward_train(x, cond=cond))
                x, output = self.res_skip_layers[i].forward_train_res_skip(acts, x, output, x_mask, last)
                continue
            x_in = self.in_layers[i].forward_train(x)
            if gcs is not None:
                x_in = A.add_bcast(x_in, gcs[i])
            acts = A.gate(x_in)
            rs = self.res_skip_layers[i].forward_train(acts)
            if not last:
                res, skip = A.chunk_channels(rs, 2)     # one gradient buffer in the backward (no zero-fill + add per half)
                x = A.mul_bcast(A.add(x, res), x_mask)
            else:
                skip = rs
            output = skip if output is None else A.add(output, skip)
        if fused:
            return output
        return A.mul_bcast(output, x_mask)

    def forward(self, x, x_mask, g=None, gc=None, **kwargs):
        """`gc` (inference only): cond_layer(g) computed by the caller ahead of time (it depends on the speaker embedding alone:
        SynthesizerTrn._infer_body runs it on a side stream under the encoder instead of on the flow's critical path)."""
        if training_call(*self.res_skip_layers[0].parameters()) or (torch.is_grad_enabled() and x.requires_grad):
            return self.forward_train(x, x_mask, g=g)
        H = self.hidden_channels
        B, _, T = x.shape
        m = mask2d(x_mask)
        if gc is None:
            gc = self.cond_layer(g) if g is not None else None   # [B, 2H*L, 1|T]
        output = torch.empty((B, H, T), device=x.device, dtype=torch.float32)
        xcur = x
        xbuf = None
        for i in range(self.n_layers):
            cond = gc[:, i * 2 * H:(i + 1) * 2 * H] if gc is not None else None
            acts = self.in_layers[i].run(xcur, cond=cond, epi=S.EPI_GATE)
            last = i == self.n_layers - 1
            if not last:
                if xbuf is None:
                    xbuf = torch.empty((B, H, T), device=x.device, dtype=torch.float32)
                self.res_skip_layers[i].run(acts, epi=S.EPI_RES_SKIP, res=xcur, out=xbuf, out2=output, skip_from=H,
                                            mask=m, beta=1.0 if i > 0 else 0.0)
                xcur = xbuf
            else:
                # all rows are skip rows; res_mode=1 applies the final `output * x_mask`
                self.res_skip_layers[i].run(acts, epi=S.EPI_RES_SKIP, res=xcur, out=xcur if xbuf is not None else acts,
                                            out2=output, skip_from=0, mask=m, beta=1.0 if i > 0 else 0.0,
                                            res_mode=1)
        return output

    def remove_weight_norm(self):
        if self.gin_channels != 0:
            self.cond_layer.remove_weight_norm()
        for l in self.in_layers:
            l.remove_weight_norm()
        for l in self.res_skip_layers:
            l.remove_weight_norm()


class Flip(nn.Module):
    def forward(self, x, *args, reverse=False, **kwargs):
        if torch.is_grad_enabled() and x.requires_grad:
            y = torch.flip(x, [1])                       # pure data movement (channel reversal)
            return (y, torch.zeros(x.size(0), dtype=x.dtype, device=x.device)) if not reverse else y
        y = S.copy_bct(S.flip_view(x))
        if not reverse:
            return y, torch.zeros(x.size(0), dtype=x.dtype, device=x.device)
        return y


class ResidualCouplingLayer(nn.Module):
    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=0, gin_channels=0,
                 mean_only=False, wn_sharing_parameter=None):
        assert channels % 2 == 0, "channels should be divisible by 2"
        super().__init__()
        self.channels = channels
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.dilation_rate = dilation_rate
        self.n_layers = n_layers
        self.half_channels = channels // 2
        self.mean_only = mean_only
        if not mean_only:
            raise NotImplementedError("only mean_only=True couplings exist on the so-vits-svc path (models.py:40)")
        self.pre = Conv1d(self.half_channels, hidden_channels, 1)
        self.enc = WN(hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=p_dropout,
                      gin_channels=gin_channels) if wn_sharing_parameter is None else wn_sharing_parameter
        self.post = Conv1d(hidden_channels, self.half_channels * (2 - mean_only), 1)
        with torch.no_grad():
            self.post.weight.zero_()
            self.post.bias.zero_()

    def apply_inplace(self, view, x_mask, g=None, reverse=False, gc=None):
        """Coupling update on a [B,C,T] view (tensor or FlipView) IN PLACE: x1 <- m + x1*mask  /  (x1 - m)*mask."""
        half = self.half_channels
        if isinstance(view, S.FlipView):
            x0, x1 = view.narrow_c(0, half), view.narrow_c(half, half)
        else:
            x0, x1 = view[:, :half], view[:, half:]
        m = mask2d(x_mask)
        h = self.pre.run(x0, mask=m)
        h = self.enc(h, x_mask, g=g, gc=gc) if gc is not None else self.enc(h, x_mask, g=g)
        # stats = post(h) * mask ; reverse: x1 = (x1 - stats) * mask ; forward: x1 = stats + x1 * mask   (logs == 0)
        self.post.run(h, mask=m, res=x1, res_mode=2 if reverse else 3, out=x1)

    def forward_train(self, x, x_mask, g=None, reverse=False, dropout_u=None):
        """Reference modules/modules.py:288-307 with mean_only=True (logs == 0).  `dropout_u`: injected dropout draws for
        a transformer coupling network (WN has no dropout on this path)."""
        half = self.half_channels
        x0, x1 = A.chunk_channels(x, 2, views=True)         # (one gradient buffer in the backward instead of two zero-filled slices + an add)
        h = A.mul_bcast(self.pre.forward_train(x0), x_mask)
        if dropout_u is not None:
            h = self.enc.forward_train(h, x_mask, g=g, dropout_u=dropout_u)
        else:
            h = self.enc.forward_train(h, x_mask, g=g)
        m = A.mul_bcast(self.post.forward_train(h), x_mask)
        if not reverse:
            x1n = A.add(m, A.mul_bcast(x1, x_mask))
            return torch.cat([x0, x1n], 1), torch.zeros(x.size(0), dtype=x.dtype, device=x.device)
        x1n = A.mul_bcast(A.add(x1, m, 1.0, -1.0), x_mask)
        return torch.cat([x0, x1n], 1)

    def forward(self, x, x_mask, g=None, reverse=False, dropout_u=None):
        if training_call(self.pre.weight) or (torch.is_grad_enabled() and x.requires_grad):
            return self.forward_train(x, x_mask, g=g, reverse=reverse, dropout_u=dropout_u)
        y = S.copy_bct(x)
        self.apply_inplace(y, x_mask, g=g, reverse=reverse)
        if not reverse:
            return y, torch.zeros(x.size(0), dtype=x.dtype, device=x.device)  # logdet = sum(logs) = 0
        return y


class TransformerCouplingLayer(ResidualCouplingLayer):
    """Reference modules/modules.py:309-356 (use_transformer_flow): the same mean-only coupling with the conditioned FFT
    (attentions.FFT(isflow=True)) as the coupling network; the in-place update, the Flip views and the autograd form are
    ResidualCouplingLayer's."""

    def __init__(self, channels, hidden_channels, kernel_size, n_layers, n_heads, p_dropout=0, filter_channels=0,
                 mean_only=False, wn_sharing_parameter=None, gin_channels=0):
        assert channels % 2 == 0, "channels should be divisible by 2"
        nn.Module.__init__(self)
        from modules import attentions
        self.channels = channels
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.n_layers = n_layers
        self.half_channels = channels // 2
        self.mean_only = mean_only
        if not mean_only:
            raise NotImplementedError("only mean_only=True couplings exist on the so-vits-svc path (models.py:82)")
        self.pre = Conv1d(self.half_channels, hidden_channels, 1)
        self.enc = attentions.FFT(hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout, isflow=True,
                                  gin_channels=gin_channels) if wn_sharing_parameter is None else wn_sharing_parameter
        self.post = Conv1d(hidden_channels, self.half_channels * (2 - mean_only), 1)
        with torch.no_grad():
            self.post.weight.zero_()
            self.post.bias.zero_()


class ResBlock1(nn.Module):
    """Masked ResBlock (modules/modules.py:149-191); not used by `dec` (which has its own, unmasked)."""

    def __init__(self, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.convs1 = nn.ModuleList([_conv(channels, channels, kernel_size, dilation=d,
                                           padding=get_padding(kernel_size, d), weight_norm=True) for d in dilation])
        self.convs2 = nn.ModuleList([_conv(channels, channels, kernel_size, dilation=1,
                                           padding=get_padding(kernel_size, 1), weight_norm=True) for _ in dilation])
        self.convs1.apply(init_weights)
        self.convs2.apply(init_weights)

    def forward(self, x, x_mask=None):
        m = mask2d(x_mask)
        for c1, c2 in zip(self.convs1, self.convs2):
            xt = c1.run(x, pre_slope=LRELU_SLOPE, premask=m)
            x = c2.run(xt, pre_slope=LRELU_SLOPE, premask=m, res=x, res_mode=1)
        if m is not None:
            x = S.copy_bct(x, mask=m)
        return x

    def remove_weight_norm(self):
        for l in list(self.convs1) + list(self.convs2):
            l.remove_weight_norm()


class ResBlock2(nn.Module):
    def __init__(self, channels, kernel_size=3, dilation=(1, 3)):
        super().__init__()
        self.convs = nn.ModuleList([_conv(channels, channels, kernel_size, dilation=d,
                                          padding=get_padding(kernel_size, d), weight_norm=True) for d in dilation])
        self.convs.apply(init_weights)

    def forward(self, x, x_mask=None):
        m = mask2d(x_mask)
        for c in self.convs:
            x = c.run(x, pre_slope=LRELU_SLOPE, premask=m, res=x, res_mode=1)
        if m is not None:
            x = S.copy_bct(x, mask=m)
        return x

    def remove_weight_norm(self):
        for l in self.convs:
            l.remove_weight_norm()
