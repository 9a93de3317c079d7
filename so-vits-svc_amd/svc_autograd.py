"""torch.autograd glue for the TRAINING graph: every Function's forward AND backward is one or a few libsvc_hip.so
kernels (svc_hip.py); torch contributes the tape, tensor storage and pure index reshapes of weights.  Nothing here
falls back to torch arithmetic.

Lowering of the convolution family (reference: every nn.Conv1d / ConvTranspose1d / Conv2d((k,1)) on the training path,
models.py:165-227, modules/*, vdecoder/hifigan/models.py):
  dense stride-1 conv      -> svc_conv1d_f32                          dgrad: same kernel on the transposed/flipped weight
                                                                      wgrad: svc_conv1d_wgrad_f32, dbias: svc_reduce_bct
  stride-s conv            -> svc_decimate_f32 (s phases as channels) + dense conv on a re-indexed weight
  ConvTranspose1d (stride u) -> dense conv producing the u output phases as channels + interleave (adjoint of decimate)
  Conv2d((k,1),(s,1)) on [B,C,T/p,p] -> decimate by p (columns -> batch) then the strided conv above
  grouped conv (DiscriminatorS) -> svc_gconv1d_{fwd,dgrad,wgrad}
"""
import torch
from torch.autograd import Function

import svc_hip as S


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _Conv1dDense(Function):
    """y = conv1d(x, w, bias, stride=1, padding=pad, dilation=dil); w is the explicit [Cout,Cin,KS] weight."""

    @staticmethod
    def forward(ctx, x, w, bias, pad, dil):
        x = _c(x)
        Cout, Cin, KS = w.shape
        Tin = x.shape[2]
        Tout = Tin + 2 * pad - dil * (KS - 1)
        wp = S.pack_conv1d_weight(w.detach())
        y = S.conv1d(x, wp, Cout, KS, bias=bias, dil=dil, pad_left=pad, Tout=Tout)
        ctx.save_for_backward(x, w)
        ctx.cfg = (pad, dil, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        pad, dil, has_bias = ctx.cfg
        dy = _c(dy)
        Cout, Cin, KS = w.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = S.pack_conv1d_weight_T(w)
            dx = S.conv1d(dy, wt, Cin, KS, dil=dil, pad_left=dil * (KS - 1) - pad, Tout=x.shape[2])
        if ctx.needs_input_grad[1]:
            dw = S.conv1d_wgrad(dy, x, KS, dil, pad)
        if has_bias and ctx.needs_input_grad[2]:
            db = S.reduce_bct(dy, 0)
        return dx, dw, db, None, None


class _Decimate(Function):
    @staticmethod
    def forward(ctx, x, s, off, Q, lp):
        ctx.cfg = (x.shape[1], x.shape[2], s, off, lp)
        return S.decimate(x, s, off, Q, lp)

    @staticmethod
    def backward(ctx, dy):
        Cc, T, s, off, lp = ctx.cfg
        return S.decimate_bwd(dy, Cc, T, s, off, lp), None, None, None, None


class _Interleave(Function):
    """Adjoint of decimate: x [B, s*C, Q] -> y [B, C, T],  y[b,c,q*s + r + off] = x[b, r*C + c, q]."""

    @staticmethod
    def forward(ctx, x, Cc, T, s, off):
        ctx.cfg = (s, off, x.shape[2])
        return S.decimate_bwd(x, Cc, T, s, off, None)

    @staticmethod
    def backward(ctx, dy):
        s, off, Q = ctx.cfg
        return S.decimate(dy, s, off, Q, None), None, None, None, None


class _GConv1d(Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, groups):
        x = _c(x)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, groups, bias is not None)
        return S.gconv1d_fwd(x, w, bias, stride, pad, groups)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, groups, has_bias = ctx.cfg
        dy = _c(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = S.gconv1d_dgrad(dy, w, x.shape[1], x.shape[2], stride, pad, groups)
        if ctx.needs_input_grad[1]:
            dw = S.gconv1d_wgrad(dy, x, w.shape[2], stride, pad, groups)
        if has_bias and ctx.needs_input_grad[2]:
            db = S.reduce_bct(dy, 0)
        return dx, dw, db, None, None, None


class _WeightNorm(Function):
    @staticmethod
    def forward(ctx, v, g):
        w, norm = S.weight_norm_fwd(v, g.reshape(-1))
        ctx.save_for_backward(v, g, norm)
        return w

    @staticmethod
    def backward(ctx, dw):
        v, g, norm = ctx.saved_tensors
        dv, dg = S.weight_norm_bwd(v, g.reshape(-1), norm, dw)
        return dv, dg.reshape(g.shape)


class _EwUnary(Function):
    @staticmethod
    def forward(ctx, x, op, bop, alpha, save_out):
        y = S.ew(op, x, alpha=alpha)
        ctx.save_for_backward(y if save_out else x)
        ctx.cfg = (bop, alpha)
        return y

    @staticmethod
    def backward(ctx, dy):
        (s,) = ctx.saved_tensors
        bop, alpha = ctx.cfg
        return S.ew(bop, _c(dy), s, alpha=alpha), None, None, None, None


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b, alpha, beta):
        ctx.cfg = (alpha, beta)
        return S.ew(S.EW_ADD, a, b, alpha=alpha, beta=beta)

    @staticmethod
    def backward(ctx, dy):
        alpha, beta = ctx.cfg
        dy = _c(dy)
        da = dy if alpha == 1.0 else S.ew(S.EW_SCALE, dy, alpha=alpha)
        db = dy if beta == 1.0 else S.ew(S.EW_SCALE, dy, alpha=beta)
        return (da if ctx.needs_input_grad[0] else None), (db if ctx.needs_input_grad[1] else None), None, None


class _MulBcast(Function):
    """y = x * side (side broadcast over the dims where it has extent 1); gradient flows to x only unless
    side requires grad (then it is reduced back over the broadcast dims)."""

    @staticmethod
    def forward(ctx, x, side):
        ctx.save_for_backward(x, side)
        return S.ew_bct(S.EW_MUL, x, side)

    @staticmethod
    def backward(ctx, dy):
        x, side = ctx.saved_tensors
        dy = _c(dy)
        dx = S.ew_bct(S.EW_MUL, dy, side) if ctx.needs_input_grad[0] else None
        ds = None
        if ctx.needs_input_grad[1]:
            full = S.ew(S.EW_MUL, dy, _c(x))
            ds = _reduce_to(full, side.shape)
        return dx, ds


class _AddBcast(Function):
    """y = x + side (side broadcast)."""

    @staticmethod
    def forward(ctx, x, side):
        ctx.sshape = tuple(side.shape)
        return S.ew_bct(S.EW_ADD, x, side, alpha=1.0, beta=1.0)

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        ds = _reduce_to(dy, ctx.sshape) if ctx.needs_input_grad[1] else None
        return (dy if ctx.needs_input_grad[0] else None), ds


def _reduce_to(full, shape):
    """Sum a [B,C,T] tensor down to a broadcastable `shape` using the HIP reductions."""
    B, Cc, T = full.shape
    sb, sc, st = shape
    if (sb, sc, st) == (B, Cc, T):
        return full
    if st == 1 and sc == Cc and sb == B:
        return S.reduce_bct(full, 1)
    if st == 1 and sc == Cc and sb == 1:
        return S.reduce_bct(full, 0).view(1, Cc, 1)
    if sc == 1 and st == T and sb == B:
        return S.reduce_c(full)
    raise S.SvcError(f"unsupported broadcast reduction {tuple(full.shape)} -> {shape}")


class _Gate(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.save_for_backward(x)
        return S.gate_fwd(x)

    @staticmethod
    def backward(ctx, d):
        (x,) = ctx.saved_tensors
        return S.gate_bwd(x, d)


# ---------------------------------------------------------------------------------------------------------------
# functional API
# ---------------------------------------------------------------------------------------------------------------
def weight_norm(v, g):
    return _WeightNorm.apply(v, g)


def leaky_relu(x, slope):
    return _EwUnary.apply(x, S.EW_LRELU, S.EW_LRELU_BWD, float(slope), False)


def relu(x):
    return _EwUnary.apply(x, S.EW_RELU, S.EW_RELU_BWD, 0.0, False)


def tanh(x):
    return _EwUnary.apply(x, S.EW_TANH, S.EW_TANH_BWD, 1.0, True)


def add(a, b, alpha=1.0, beta=1.0):
    return _Add.apply(a, b, float(alpha), float(beta))


def mul_bcast(x, side):
    return _MulBcast.apply(x, side)


def add_bcast(x, side):
    return _AddBcast.apply(x, side)


def gate(x):
    return _Gate.apply(x)


def conv1d(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    """F.conv1d semantics on [B,Cin,T] with an explicit weight [Cout, Cin/groups, KS]."""
    Cout, Cg, KS = w.shape
    if groups != 1:
        if dilation != 1:
            raise S.SvcError("grouped conv with dilation is not on the so-vits-svc path")
        return _GConv1d.apply(x, w, bias, stride, padding, groups)
    if stride == 1:
        return _Conv1dDense.apply(x, w, bias, padding, dilation)
    if dilation != 1:
        raise S.SvcError("strided conv with dilation is not on the so-vits-svc path")
    # stride-s conv: input position t*s + k - pad = (t + m)*s + r with k - pad = s*m + r, r in [0,s)
    s = stride
    Tin = x.shape[2]
    Tout = (Tin + 2 * padding - KS) // s + 1
    m_min = (0 - padding) // s
    m_max = (KS - 1 - padding) // s
    KSd = m_max - m_min + 1
    shift = -(s * m_min + padding)               # >= 0: zeros in front so that k + shift = s*(m - m_min) + r
    wpad = torch.nn.functional.pad(w, (shift, s * KSd - KS - shift))
    wd = wpad.view(Cout, Cg, KSd, s).permute(0, 3, 1, 2).reshape(Cout, s * Cg, KSd)   # index reshapes only
    Q = (Tin + s - 1) // s
    xd = _Decimate.apply(x, s, 0, Q, None)
    y = _Conv1dDense.apply(xd, wd, bias, -m_min, 1)
    # dense conv over Q samples yields Q + 2*(-m_min) - (KSd-1) outputs; keep the first Tout
    return y[:, :, :Tout] if y.shape[2] != Tout else y


def conv_transpose1d(x, w, bias=None, stride=1, padding=0):
    """F.conv_transpose1d semantics; w [Cin, Cout, KS]."""
    Cin, Cout, KS = w.shape
    u = stride
    Tin = x.shape[2]
    Lout = (Tin - 1) * u - 2 * padding + KS
    M = (KS + u - 1) // u
    # y[co, q*u + p - pad] = sum_ci sum_m x[ci, q-m] W[ci,co,p+m*u]: dense conv with the phases as output channels
    wpad = torch.nn.functional.pad(w, (0, M * u - KS))
    wd = wpad.view(Cin, Cout, M, u).flip(2).permute(3, 1, 0, 2).reshape(u * Cout, Cin, M)
    yq = _Conv1dDense.apply(x, wd, None, M - 1, 1)          # [B, u*Cout, Tin + M - 1]
    y = _Interleave.apply(yq, Cout, Lout, u, -padding)
    if bias is not None:
        y = add_bcast(y, bias.view(1, -1, 1))
    return y
