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
import math

import torch
from torch.autograd import Function

import svc_hip as S


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _Conv1dDense(Function):
    """y = conv1d(x, w, bias, stride=1, padding=pad, dilation=dil)[..., :tout]; w is the explicit [Cout,Cin,KS] weight.
    `exact`: produce exactly `tout` columns even beyond the natural output length (x counts as zero-extended) — the padded
    row layout of DiscriminatorP, where the columns past the logical length are don't-care and get masked by the caller."""

    @staticmethod
    def forward(ctx, x, w, bias, pad, dil, tout=None, exact=False):
        x = _c(x)
        Cout, Cin, KS = w.shape
        Tin = x.shape[2]
        Tout = Tin + 2 * pad - dil * (KS - 1)
        if tout is not None:
            Tout = tout if exact else min(Tout, tout)
        wp = S.pack_conv1d_weight(w.detach())
        wp.d4_ok = False               # (a pack that lives for one call: no second pack for the short-sequence kernel)
        ctx.mma = S.current_mma()      # the operand format of this forward is also the one of its backward (an autocast region's rule)
        y = S.conv1d(x, wp, Cout, KS, bias=bias, dil=dil, pad_left=pad, Tout=Tout, mma=ctx.mma)
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
            dx = S.conv1d(dy, wt, Cin, KS, dil=dil, pad_left=dil * (KS - 1) - pad, Tout=x.shape[2], mma=ctx.mma)
        want_db = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if want_db:      # bias gradient from the dy tiles the wgrad kernel stages anyway
                db, zeroed = S.wgrad_zeros((Cout,), dy.device)
                if not zeroed and S.wgrad_slab.active:
                    db.zero_()
            dw = S.conv1d_wgrad(dy, x, KS, dil, pad, dbias=db, mma=ctx.mma)
        elif want_db:
            db = S.reduce_bct(dy, 0)
        return dx, dw, db, None, None, None, None


class _ConvPlanned(Function):
    """_Conv1dDense with the weight side folded in: the inputs are the PARAMETERS (v and, for weight-normed layers, g) and a
    svc_hip.ConvWeightPlan that maps them to the dense-conv operands.  Forward = plan.prepare (one launch: weight-norm scale,
    index map of a strided / transposed layout, forward AND dgrad packing) + the MFMA conv; backward = dgrad on the operand
    prepared in the forward, wgrad in the dense layout, plan.grad (one launch back to dv, dg).  Replaces weight_norm_fwd +
    F.pad / permute / copy + pack (+ pack_T, the adjoint copies and weight_norm_bwd in the backward): 4-9 launches -> 2."""

    @staticmethod
    def forward(ctx, x, v, g, bias, plan, pad, dil, tout, exact, cond=None, res=None, mask=None, post_act=0, post_slope=0.0):
        """Epilogue fusions (each one launch and one pass over the output less than the separate op; the kernel's order:
        bias + cond, activation, mask, residual — include/svc_hip.h):
          cond [B, Od, 1|T]  added before the activation (WN's `x_in + g_l`, modules/modules.py:123-128)
          post_act           S.ACT_RELU: FFN's `torch.relu(conv_1(x))` (modules/attentions.py:342); S.ACT_LRELU (post_slope > 0):
                             `F.leaky_relu(c1(..))` in front of a ResBlock's second conv (vdecoder/hifigan/models.py:64-65)
          mask [B, 1, T]     `y * x_mask` after a conv
          res [B, Od, T]     `conv(x) + res` (the ResBlock sums, vdecoder/hifigan/models.py:66,92); not with post_act"""
        x = _c(x)
        vd = v.detach()
        gd = g.detach().reshape(-1) if g is not None else None
        wp, _ = plan.prepare(vd, gd)
        Tin = x.shape[2]
        Tout = Tin + 2 * pad - dil * (plan.Kd - 1)
        if tout is not None:
            Tout = tout if exact else min(Tout, tout)
        # 1x1 conv of a [B, C, 1] tensor (the speaker-conditioning convs `cond_layer(g)`, modules/modules.py:96-97,114): as B rows
        # of ONE column each launch fills 1/128 of a tile (0.3 TFLOP/s, 150 us per call at B = 16: profiles/r03w_*).  The batch
        # becomes the column axis instead: [1, C, B] — the same GEMM in one tile row.
        ctx.batch_cols = bool(Tin == 1 and Tout == 1 and plan.Kd == 1 and pad == 0 and x.shape[0] > 1)
        ctx.mma = S.current_mma()
        if ctx.batch_cols:
            x = x.squeeze(2).t().contiguous().unsqueeze(0)                       # [1, Cin, B]
            y = S.conv1d(x, wp, plan.Od, 1, bias=bias, mma=ctx.mma)               # [1, Od, B]
            y = y.squeeze(0).t().contiguous().unsqueeze(2)                       # [B, Od, 1]
        else:
            y = S.conv1d(x, wp, plan.Od, plan.Kd, bias=bias, dil=dil, pad_left=pad, Tout=Tout, mma=ctx.mma, cond=cond,
                         mask=None if mask is None else mask.detach(), post_act=post_act, post_slope=post_slope,
                         res=None if res is None else _c(res), res_mode=0 if res is None else 1)
        fused = cond is not None or res is not None or mask is not None or post_act != 0
        if fused and ctx.batch_cols:
            raise S.SvcError("conv1d_planned: epilogue fusions with a [B, C, 1] input are not on the path")
        if post_act not in (0, S.ACT_RELU, S.ACT_LRELU) or (post_act and res is not None) or (post_act == S.ACT_LRELU and not
                                                                                            (post_slope > 0.0 and mask is None)):
            raise S.SvcError("conv1d_planned: the fused activation is ReLU or leaky ReLU (slope > 0, no mask), without a residual")
        ctx.cond_shape = tuple(cond.shape) if cond is not None else None
        ctx.epi = (res is not None, mask is not None, post_act, float(post_slope))
        ctx.save_for_backward(x, v, g, y if post_act else None, mask.detach() if (mask is not None and not post_act) else None)
        ctx.plan = plan
        ctx.cfg = (pad, dil, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, v, g, y_act, m = ctx.saved_tensors
        plan = ctx.plan
        pad, dil, has_bias = ctx.cfg
        dy = _c(dy)
        has_res, has_mask, post_act, post_slope = ctx.epi
        d_res = dy if (has_res and ctx.needs_input_grad[10]) else None
        if post_act == S.ACT_LRELU:
            dy = S.ew(S.EW_LRELU_BWD, dy, y_act, alpha=post_slope)      # sign(y) == sign(v) for slope > 0
        elif post_act:
            dy = S.ew(S.EW_RELU_BWD, dy, y_act)         # y = relu(v) * mask: y > 0 <=> v > 0 inside the mask
        elif has_mask:
            dy = S.ew_bct(S.EW_MUL, dy, m)
        if ctx.batch_cols:                                                       # x was saved as [1, Cin, B]
            dy = dy.squeeze(2).t().contiguous().unsqueeze(0)                     # [1, Od, B]
        dx = dv = dg = db = None
        if ctx.needs_input_grad[0]:
            dx = S.conv1d(dy, plan.wt, plan.Id, plan.Kd, dil=dil, pad_left=dil * (plan.Kd - 1) - pad, Tout=x.shape[2], mma=ctx.mma)
            if ctx.batch_cols:
                dx = dx.squeeze(0).t().contiguous().unsqueeze(2)                 # [B, Cin, 1]
        want_db = has_bias and ctx.needs_input_grad[3]
        if ctx.needs_input_grad[1]:
            if want_db:
                db, zeroed = S.wgrad_zeros((plan.Od,), dy.device)
                if not zeroed and S.wgrad_slab.active:
                    db.zero_()
            dwd = S.conv1d_wgrad(dy, x, plan.Kd, dil, pad, dbias=db, mma=ctx.mma)
            gd = g.detach().reshape(-1) if g is not None else None
            dv, dgf = plan.grad(v.detach(), gd, dwd)
            dv = dv.view(v.shape)
            if g is not None:
                dg = dgf.view(g.shape)
        elif want_db:
            db = S.reduce_bct(dy, 0)
        dc = _reduce_to(dy, ctx.cond_shape) if ctx.cond_shape is not None and ctx.needs_input_grad[9] else None
        return dx, dv, dg, db, None, None, None, None, None, dc, d_res, None, None, None


class _WNResSkip(Function):
    """One WN layer's `res_skip_layers[i](acts)` with what follows it (modules/modules.py:130-137) in the conv's epilogue —
    the inference path's SVC_EPI_RES_SKIP launch:
        x_new  = (x + rs[:, :H]) * x_mask          (not the last layer)
        output = output + rs[:, H:]                (in place; the last layer: output = (output + rs) * x_mask, :138)
    instead of chunk + add + mask-multiply + add (three element-wise launches forward).  Backward: the gradient of the conv
    output is [dx_new * x_mask ; d_output] written into ONE buffer (two launches instead of the chunk / mask / add adjoints),
    then dgrad / wgrad / plan.grad as _ConvPlanned.  `output` is None for the first layer."""

    @staticmethod
    def forward(ctx, acts, x, output, x_mask, v, g, bias, plan, last):
        acts = _c(acts)
        B, H, T = acts.shape
        vd = v.detach()
        gd = g.detach().reshape(-1) if g is not None else None
        wp, _ = plan.prepare(vd, gd)
        first = output is None
        if first:
            output = torch.empty((B, H, T), device=acts.device, dtype=torch.float32)
        else:
            ctx.mark_dirty(output)
        ctx.mma = S.current_mma()
        m = x_mask.detach()
        if last:
            # every row is a skip row; res_mode 1 applies the final `output * x_mask`
            S.conv1d(acts, wp, H, 1, bias=bias, epi=S.EPI_RES_SKIP, res=acts, out=acts, out2=output, skip_from=0, mask=m,
                     beta=0.0 if first else 1.0, res_mode=1, mma=ctx.mma)
            x_new = None
        else:
            x = _c(x)
            x_new = torch.empty_like(x)
            S.conv1d(acts, wp, 2 * H, 1, bias=bias, epi=S.EPI_RES_SKIP, res=x, out=x_new, out2=output, skip_from=H, mask=m,
                     beta=0.0 if first else 1.0, mma=ctx.mma)
        ctx.save_for_backward(acts, m, v, g)
        ctx.plan = plan
        ctx.cfg = (last, first, bias is not None, H)
        if last:
            return output
        return x_new, output

    @staticmethod
    def backward(ctx, *grads):
        acts, m, v, g = ctx.saved_tensors
        plan = ctx.plan
        last, first, has_bias, H = ctx.cfg
        B, _, T = acts.shape
        if last:
            (dout,) = grads
            d_rs = S.ew_bct(S.EW_MUL, _c(dout), m)                  # gradient of (output + rs) * x_mask
            d_x, d_prev = None, (None if first else d_rs)
        else:
            dxn, dout = grads
            d_rs = torch.empty((B, 2 * H, T), device=acts.device, dtype=torch.float32)
            if dxn is None:
                d_rs[:, :H].zero_()
            else:
                S.ew_bct(S.EW_MUL, dxn if dxn.stride(2) == 1 else _c(dxn), m, out=d_rs[:, :H])   # (a batch-strided view is fine)
            if dout is None:
                d_rs[:, H:].zero_()
            else:
                S.copy_bct(_c(dout), out=d_rs[:, H:])
            d_x = d_rs[:, :H] if ctx.needs_input_grad[1] else None
            d_prev = None if first else dout
        d_acts = dv = dg = db = None
        if ctx.needs_input_grad[0]:
            d_acts = S.conv1d(d_rs, plan.wt, plan.Id, 1, Tout=T, mma=ctx.mma)
        want_db = has_bias and ctx.needs_input_grad[6]
        if ctx.needs_input_grad[4]:
            if want_db:
                db, zeroed = S.wgrad_zeros((plan.Od,), acts.device)
                if not zeroed and S.wgrad_slab.active:
                    db.zero_()
            dwd = S.conv1d_wgrad(d_rs, acts, 1, 1, 0, dbias=db, mma=ctx.mma)
            gd = g.detach().reshape(-1) if g is not None else None
            dv, dgf = plan.grad(v.detach(), gd, dwd)
            dv = dv.view(v.shape)
            dg = dgf.view(g.shape) if g is not None else None
        elif want_db:
            db = S.reduce_bct(d_rs, 0)
        return d_acts, d_x, d_prev, None, dv, dg, db, None, None


def wn_res_skip(acts, x, output, x_mask, plan, v, g, bias, last):
    """See _WNResSkip: returns (x_new, output) — x_new is None for the last layer."""
    if last:
        return None, _WNResSkip.apply(acts, x, output, x_mask, v, g, bias, plan, True)
    return _WNResSkip.apply(acts, x, output, x_mask, v, g, bias, plan, False)


class _Decimate(Function):
    @staticmethod
    def forward(ctx, x, s, off, Q, lp, inner=1):
        ctx.cfg = (x.shape[1], x.shape[2], s, off, lp, inner)
        return S.decimate(x, s, off, Q, lp, inner)

    @staticmethod
    def backward(ctx, dy):
        Cc, T, s, off, lp, inner = ctx.cfg
        return S.decimate_bwd(dy, Cc, T, s, off, lp, inner), None, None, None, None, None


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


class _ChunkC(Function):
    """x [B, n*C, T] -> n contiguous [B, C, T] chunks; the backward writes the n gradients into ONE buffer (torch's own
    slice backward would zero-fill and add a full-size tensor per chunk)."""

    @staticmethod
    def forward(ctx, x, n, views):
        B, Ct, T = x.shape
        C = Ct // n
        ctx.cfg = (n, C)
        if views:       # channel-slice VIEWS (what a Python slice gives; consumers take strided [B, C, T] operands): no launch
            return tuple(x.narrow(1, i * C, C) for i in range(n))
        outs = tuple(S.copy_bct(x[:, i * C:(i + 1) * C]) for i in range(n))
        return outs

    @staticmethod
    def backward(ctx, *gs):
        n, C = ctx.cfg
        g0 = next(g for g in gs if g is not None)
        B, _, T = g0.shape
        if all(g is not None and g.is_contiguous() for g in gs) and (n > 2 or g0.numel() < (1 << 16)):
            return torch.cat(gs, 1), None, None       # many small pieces (per-layer conditioning rows): ONE launch
        dx = torch.empty((B, n * C, T), device=g0.device, dtype=torch.float32)
        for i, g in enumerate(gs):
            if g is None:
                dx[:, i * C:(i + 1) * C].zero_()
            else:
                S.copy_bct(g, out=dx[:, i * C:(i + 1) * C])
        return dx, None, None


class _SplitBatch(Function):
    """(x[:n], x[n:]) along dim 0 as views; the backward joins the two gradients with ONE cat (torch's slice backward: a zero-fill,
    a copy and an accumulate per half) — the real / generated halves of the discriminators' batched pass (models.py:246-250)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n, ctx.shape = n, x.shape
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, g0, g1):
        n, shape = ctx.n, ctx.shape
        ref = g0 if g0 is not None else g1
        if g0 is None:
            g0 = torch.zeros((n,) + tuple(shape[1:]), device=ref.device, dtype=ref.dtype)
        if g1 is None:
            g1 = torch.zeros((shape[0] - n,) + tuple(shape[1:]), device=ref.device, dtype=ref.dtype)
        return torch.cat([g0, g1], 0), None


def split_batch(x, n):
    return _SplitBatch.apply(x, n)


class _StackQKV(Function):
    """The fused q / k / v projection weight (or bias) of a training attention layer: three [H*dk, ...] parameters -> one tensor whose
    rows are ordered (head, {q, k, v}, d).  Backward: ONE permuting copy of the fused gradient to [3, H, dk, ...], whose three slices
    are contiguous — unbound views of torch.stack's own backward are strided, and AccumulateGrad clones every strided gradient
    (six clone launches per layer)."""

    @staticmethod
    def forward(ctx, wq, wk, wv, H):
        ctx.H, ctx.shape = H, wq.shape
        dk = wq.shape[0] // H
        rest = tuple(wq.shape[1:])
        return torch.stack([w.reshape((H, dk) + rest) for w in (wq, wk, wv)], 1).reshape((3 * wq.shape[0],) + rest)

    @staticmethod
    def backward(ctx, g):
        H, shape = ctx.H, ctx.shape
        dk = shape[0] // H
        g3 = g.reshape((H, 3, dk) + tuple(shape[1:])).transpose(0, 1).contiguous()       # [3, H, dk, ...]: one launch
        return g3[0].reshape(shape), g3[1].reshape(shape), g3[2].reshape(shape), None


def stack_qkv(wq, wk, wv, n_heads):
    return _StackQKV.apply(wq, wk, wv, n_heads)


def chunk_channels(x, n, views=False):
    """x [B, n*C, T] -> n [B, C, T] chunks whose gradients meet in ONE buffer in the backward (torch's own slice backward zero-fills
    and accumulates a full-size tensor per chunk).  views=True: the chunks are channel-slice views of x (no forward launch)."""
    return _ChunkC.apply(x, n, views)


# ---- packed batches: [B, C, T] items laid end to end in ONE row of length Lp, `gap` zero columns after each item ----------
# A stride-1 "same" conv with halo <= gap over the packed row equals the per-item zero-padded conv as long as the gap columns
# of its INPUT are zero; its MFMA column tiles then run over B*(T+gap) columns instead of B x ceil(T/128) x 128 (short
# items: T = 172 fills 67 % of two 128-wide tiles).  The [B, C, T] window of a packed row is a strided view, so the existing
# strided copy / broadcast / reduction kernels do the packing and the gap masking.
def packed_len(B, T, gap):
    return (B * (T + gap) + 3) // 4 * 4


def _pview(p, B, T, gap):
    return p.as_strided((B, p.shape[1], T), (T + gap, p.shape[2], 1), p.storage_offset())


class _Pack(Function):
    @staticmethod
    def forward(ctx, x, gap):
        B, C, T = x.shape
        ctx.cfg = (B, T, gap)
        p = torch.zeros((1, C, packed_len(B, T, gap)), device=x.device, dtype=torch.float32)
        S.copy_bct(x, out=_pview(p, B, T, gap))
        return p

    @staticmethod
    def backward(ctx, dp):
        B, T, gap = ctx.cfg
        return S.copy_bct(_pview(_c(dp), B, T, gap)), None


class _Unpack(Function):
    @staticmethod
    def forward(ctx, p, B, T, gap):
        ctx.cfg = (B, T, gap, p.shape[2])
        return S.copy_bct(_pview(_c(p), B, T, gap))

    @staticmethod
    def backward(ctx, dx):
        B, T, gap, Lp = ctx.cfg
        dp = torch.zeros((1, dx.shape[1], Lp), device=dx.device, dtype=torch.float32)
        S.copy_bct(_c(dx), out=_pview(dp, B, T, gap))
        return dp, None, None, None


class _PackedAddItem(Function):
    """y = packed(x_b + side_b) with the gap columns ZERO (the mask a following conv with a halo needs); side [B, C, 1]."""

    @staticmethod
    def forward(ctx, p, side, B, T, gap):
        ctx.cfg = (B, T, gap)
        y = torch.zeros_like(p)
        S.ew_bct(S.EW_ADD, _pview(_c(p), B, T, gap), side, alpha=1.0, beta=1.0, out=_pview(y, B, T, gap))
        return y

    @staticmethod
    def backward(ctx, dy):
        B, T, gap = ctx.cfg
        dyv = _pview(_c(dy), B, T, gap)
        dp = None
        if ctx.needs_input_grad[0]:
            dp = torch.zeros_like(dy)
            S.copy_bct(dyv, out=_pview(dp, B, T, gap))
        ds = S.reduce_bct(dyv, 1) if ctx.needs_input_grad[1] else None
        return dp, ds, None, None, None


def pack_items(x, gap):
    return _Pack.apply(x, gap)


def unpack_items(p, B, T, gap):
    return _Unpack.apply(p, B, T, gap)


def packed_add_item(p, side, B, T, gap):
    return _PackedAddItem.apply(p, side, B, T, gap)


# ---------------------------------------------------------------------------------------------------------------
# functional API
# ---------------------------------------------------------------------------------------------------------------
class _SpectralNorm(Function):
    """torch.nn.utils.spectral_norm's compute_weight (one power iteration in training mode, in place on the u / v buffers;
    gradient through sigma with u, v held constant)."""

    @staticmethod
    def forward(ctx, W, u, v, power_iteration, eps):
        Wd = W.detach()
        w, sigma = S.spectral_norm_fwd(Wd, u, v, power_iteration, eps)
        # torch clones u, v after the iteration so that the tensors saved for backward survive the next in-place update
        ctx.save_for_backward(W, u.clone(), v.clone(), sigma)
        return w

    @staticmethod
    def backward(ctx, g):
        W, u, v, sigma = ctx.saved_tensors
        return S.spectral_norm_bwd(W.detach(), u, v, sigma, g), None, None, None, None


def spectral_norm(weight_orig, u, v, training, eps=1e-12):
    return _SpectralNorm.apply(weight_orig, u, v, bool(training), eps)


def weight_norm(v, g):
    return _WeightNorm.apply(v, g)


def leaky_relu(x, slope):
    return _EwUnary.apply(x, S.EW_LRELU, S.EW_LRELU_BWD, float(slope), False)


class _LReluRes(Function):
    """(leaky_relu(x), x) for an x that feeds `lrelu -> conv` AND a residual add (a ResBlock pair's input, vdecoder/hifigan/
    models.py:60-67): as ONE node with two outputs, so that the two gradients meet in ONE launch — leaky_relu'(x) * g_act + g_res
    (svc_lrelu_bwd_add_f32) — instead of the leaky-ReLU backward plus the autograd engine's own accumulation add."""

    @staticmethod
    def forward(ctx, x, slope):
        x = _c(x)
        ctx.save_for_backward(x)
        ctx.slope = slope
        return S.ew(S.EW_LRELU, x, alpha=slope), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gr):
        (x,) = ctx.saved_tensors
        if ga is None:
            return gr, None
        if gr is None:
            return S.ew(S.EW_LRELU_BWD, _c(ga), x, alpha=ctx.slope), None
        return S.lrelu_bwd_add(ga, x, gr, ctx.slope), None


def leaky_relu_res(x, slope):
    return _LReluRes.apply(x, float(slope))


class _LReluTail(Function):
    """leaky_relu on the first `valid` columns of every row and zero on the padded tail, in both directions
    (svc_lrelu_tail_{fwd,bwd}_f32); the saved tensor is the OUTPUT (sign(y) == sign(x) for slope > 0)."""

    @staticmethod
    def forward(ctx, x, slope, valid):
        y = S.lrelu_tail_fwd(_c(x), valid, slope)
        ctx.save_for_backward(y)
        ctx.cfg = (slope, valid)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        slope, valid = ctx.cfg
        return S.lrelu_tail_bwd(y, dy, valid, slope), None, None


def leaky_relu_tail(x, slope, valid):
    return _LReluTail.apply(x, slope, valid)


def relu(x):
    return _EwUnary.apply(x, S.EW_RELU, S.EW_RELU_BWD, 0.0, False)


def tanh(x):
    return _EwUnary.apply(x, S.EW_TANH, S.EW_TANH_BWD, 1.0, True)


def mish(x):
    """x * tanh(softplus(x)) (diffusion/wavenet.py:76)."""
    return _EwUnary.apply(x, S.EW_MISH, S.EW_MISH_BWD, 0.0, False)


def add(a, b, alpha=1.0, beta=1.0):
    return _Add.apply(a, b, float(alpha), float(beta))


class _Scale(Function):
    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return S.ew(S.EW_SCALE, x, alpha=alpha)

    @staticmethod
    def backward(ctx, dy):
        return S.ew(S.EW_SCALE, _c(dy), alpha=ctx.alpha), None


def scale(x, alpha):
    return _Scale.apply(x, float(alpha))


def mul_bcast(x, side):
    return _MulBcast.apply(x, side)


def add_bcast(x, side):
    return _AddBcast.apply(x, side)


def gate(x):
    return _Gate.apply(x)


def align_blocks(h, inner):
    """Smallest block count >= h whose row length h*inner is a multiple of 4 floats (16-byte rows)."""
    m = 4 // math.gcd(inner, 4)
    return (h + m - 1) // m * m


def conv1d(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1, inner=1, lp=None, out_blocks=None):
    """F.conv1d semantics on [B,Cin,T] with an explicit weight [Cout, Cin/groups, KS].
    inner > 1: x is [B, Cin, H*inner] — H blocks of `inner` time-contiguous samples — and the convolution runs over the
    BLOCK index (this is Conv2d((KS,1),(stride,1)) on the [B,Cin,H,inner] view, models.py:171-177); stride/padding are
    in blocks.  lp: length the input is (virtually) reflect-padded to on the right (models.py:185-189).
    out_blocks: produce exactly this many output blocks (>= the logical count; x counts as zero-extended) and decimate to
    a 16-byte-aligned row length — the padded row layout of DiscriminatorP: with zero tails on x the first logical-count
    blocks are the convolution's result and the rest is don't-care (the caller masks it, svc_autograd.leaky_relu_tail)."""
    Cout, Cg, KS = w.shape
    if groups != 1:
        if dilation != 1 or inner != 1 or out_blocks is not None:
            raise S.SvcError("grouped conv with dilation is not on the so-vits-svc path")
        return _GConv1d.apply(x, w, bias, stride, padding, groups)

    def dense(xx, pad, dil, tout=None, exact=False, strided=None):
        if strided is None:
            return _Conv1dDense.apply(xx, w, bias, pad, dil, tout, exact)
        KSd, shift = strided
        wpad = torch.nn.functional.pad(w, (shift, stride * KSd - KS - shift))
        wd = wpad.view(Cout, Cg, KSd, stride).permute(0, 3, 1, 2).reshape(Cout, stride * Cg, KSd)   # index reshapes only
        return _Conv1dDense.apply(xx, wd, bias, pad, dil, tout, exact)

    return _conv1d_lowered(dense, x, KS, stride, padding, dilation, inner, lp, out_blocks)


def strided_geometry(KS, s, padding):
    """Stride-s conv as a dense conv over the s input phases: input position t*s + k - pad = (t + m)*s + r with
    k - pad = s*m + r, r in [0,s).  -> (KSd taps of the dense conv, shift: zeros in front of the kernel so that
    k + shift = s*(m - m_min) + r, m_min)."""
    m_min = (0 - padding) // s
    m_max = (KS - 1 - padding) // s
    return m_max - m_min + 1, -(s * m_min + padding), m_min


def _conv1d_lowered(dense, x, KS, stride, padding, dilation, inner, lp, out_blocks):
    """The activation side shared by conv1d() and conv1d_planned(): `dense(x, pad, dil, tout, exact, strided)` runs the dense
    stride-1 convolution (strided = (KSd, shift) when the weight has to take the phase-decimated layout)."""
    if stride == 1:
        if lp is not None and lp != x.shape[2]:
            raise S.SvcError("reflect padding is folded into the decimation of a strided conv only")
        if out_blocks is not None:
            return dense(x, padding * inner, dilation * inner, out_blocks * inner, True)
        return dense(x, padding * inner, dilation * inner)
    if dilation != 1:
        raise S.SvcError("strided conv with dilation is not on the so-vits-svc path")
    s = stride
    Tin = (x.shape[2] if lp is None else lp) // inner          # in blocks
    Tout = (Tin + 2 * padding - KS) // s + 1
    KSd, shift, m_min = strided_geometry(KS, s, padding)
    Q = (Tin + s - 1) // s
    if out_blocks is not None:
        # blocks past the signal decimate to zeros (svc_decimate_f32 zero-fills beyond lp), so rounding Q up only appends a
        # zero tail; the adjoint never reads it
        xd = _Decimate.apply(x, s, 0, align_blocks(Q, inner), lp, inner)
        return dense(xd, -m_min * inner, inner, out_blocks * inner, True, (KSd, shift))
    xd = _Decimate.apply(x, s, 0, Q, lp, inner)
    # dense conv (dilation `inner`) over the Q blocks; only the first Tout blocks are produced
    return dense(xd, -m_min * inner, inner, Tout * inner, False, (KSd, shift))


def conv_plan(weight_shape, stride=1, padding=0, transposed=False):
    """svc_hip.ConvWeightPlan for a module's convolution: weight [Cout, Cin, KS(,1)] (nn.Conv1d / Conv2d((k,1))) or, with
    transposed=True, [Cin, Cout, KS] (nn.ConvTranspose1d)."""
    R, C2, KS = weight_shape[0], weight_shape[1], weight_shape[2]
    P = S.ConvWeightPlan
    if transposed:
        return P(P.TRANSPOSED, R, C2, KS, s=stride)
    if stride == 1:
        return P(P.DENSE, R, C2, KS)
    KSd, shift, _ = strided_geometry(KS, stride, padding)
    return P(P.STRIDED, R, C2, KS, s=stride, shift=shift, Kd=KSd)


def conv1d_planned(x, plan, v, g=None, bias=None, stride=1, padding=0, dilation=1, inner=1, lp=None, out_blocks=None,
                   causal=False, cond=None, res=None, mask=None, post_act=0, post_slope=0.0):
    """conv1d() on a module's parameters through its ConvWeightPlan (groups == 1): v is weight / weight_v, g weight_g or
    None.  causal: left padding (K-1)*dilation only, output length == input length."""
    KS = plan.K
    if causal:
        if stride != 1 or inner != 1:
            raise S.SvcError("causal padding with a stride is not on the so-vits-svc path")
        return _ConvPlanned.apply(x, v, g, bias, plan, (KS - 1) * dilation, dilation, x.shape[2], False, cond, res, mask, post_act, post_slope)

    if cond is not None or res is not None or mask is not None or post_act:
        if stride != 1 or inner != 1 or lp is not None or out_blocks is not None:
            raise S.SvcError("conv1d_planned: the fused epilogues exist for plain stride-1 convolutions")
        return _ConvPlanned.apply(x, v, g, bias, plan, padding, dilation, None, False, cond, res, mask, post_act, post_slope)

    def dense(xx, pad, dil, tout=None, exact=False, strided=None):
        return _ConvPlanned.apply(xx, v, g, bias, plan, pad, dil, tout, exact)

    return _conv1d_lowered(dense, x, KS, stride, padding, dilation, inner, lp, out_blocks)


def conv_transpose1d_planned(x, plan, v, g=None, bias=None, stride=1, padding=0):
    """conv_transpose1d() on a module's parameters through its (TRANSPOSED) ConvWeightPlan; v [Cin, Cout, KS]."""
    Cin, Cout, KS = plan.R, plan.C2, plan.K
    M = plan.Kd
    Lout = (x.shape[2] - 1) * stride - 2 * padding + KS
    yq = _ConvPlanned.apply(x, v, g, None, plan, M - 1, 1, None, False)          # [B, u*Cout, Tin + M - 1]
    y = _Interleave.apply(yq, Cout, Lout, stride, -padding)
    if bias is not None:
        y = add_bcast(y, bias.view(1, -1, 1))
    return y


def conv1d_causal(x, w, bias=None, dilation=1):
    """F.conv1d(F.pad(x, ((K-1)*dilation, 0)), w, bias, dilation=dilation): output length == input length."""
    KS = w.shape[2]
    return _Conv1dDense.apply(x, w, bias, (KS - 1) * dilation, dilation, x.shape[2])


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


# ---------------------------------------------------------------------------------------------------------------
# LayerNorm / attention / embedding / reparam / NSF source
# ---------------------------------------------------------------------------------------------------------------
class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = _c(x)
        y, mean, rstd = S.layernorm_fwd(x, gamma, beta, eps)
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dx, dg, db = S.layernorm_bwd(x, gamma, dy, mean, rstd)
        return dx, dg, db, None


def layer_norm(x, gamma, beta, eps=1e-5):
    return _LayerNorm.apply(x, gamma, beta, eps)


class _Attention(Function):
    """MultiHeadAttention.attention (modules/attentions.py:207-239) on projected q,k,v [B, H*dk, T], training form:
    scores / probabilities are materialised per head ([B*H, T, T]) and every product runs on svc_gemm_f32.  With
    `drop_u` (uniform draws [B,H,T,T]) the probabilities are dropped as the reference does (:232) inside the softmax
    kernel: P (for the softmax backward) and Pd = P * keep (for the AV and relative-value products) are both kept."""

    @staticmethod
    def forward(ctx, q, k, v, emb_k, emb_v, mask, n_heads, window, mask_mode, drop_u, p_drop):
        q, k, v = _c(q), _c(k), _c(v)
        B, Cc, T = q.shape
        H = n_heads
        dk = Cc // H
        BH = B * H
        sc = dk ** -0.5
        qs = (dk * T, 1, T)      # A[m=i, k=d] = q[bh, d, i]
        P = S.gemm(q, k, qs, (dk * T, T, 1), BH, T, T, dk, alpha=sc)
        rel = None
        if window:
            ek = _c(emb_k.view(-1, dk))
            ev = _c(emb_v.view(-1, dk))
            rel = S.gemm(q, ek, qs, (0, 1, dk), BH, T, 2 * window + 1, dk, alpha=sc)
        hashed = isinstance(drop_u, S.HashDraw)      # keep decisions made inside the kernels from (seed, site, element)
        if drop_u is not None and not hashed:
            drop_u = _c(drop_u)
            if drop_u.numel() != P.numel():
                raise S.SvcError(f"attention dropout draws {tuple(drop_u.shape)} do not match [B,H,T,T] = {(B, H, T, T)}")
        Pd = S.attn_softmax_fwd(P, rel, mask, B, H, T, window, mask_mode, drop_u, p_drop)
        out = torch.empty_like(q)
        S.gemm(v, Pd, (dk * T, T, 1), (T * T, 1, T), BH, dk, T, T, out=out, c_strides=(dk * T, T, 1))
        pband = None
        if window:
            pband = S.band_gather(Pd, BH * T, T, window)
            S.gemm(ev, pband, (0, 1, dk), (T * (2 * window + 1), 1, 2 * window + 1), BH, dk, T, 2 * window + 1, out=out,
                   c_strides=(dk * T, T, 1), beta=1.0)
        ctx.save_for_backward(q, k, v, P, pband, emb_k, emb_v, None if hashed else drop_u, Pd if drop_u is not None else None, mask)
        ctx.cfg = (B, H, dk, T, window, p_drop, mask_mode)
        ctx.hash_draw = drop_u if hashed else None
        return out

    @staticmethod
    def backward(ctx, dO):
        q, k, v, P, pband, emb_k, emb_v, drop_u, Pd, mask = ctx.saved_tensors
        B, H, dk, T, window, p_drop, mask_mode = ctx.cfg
        if ctx.hash_draw is not None:
            drop_u = ctx.hash_draw
        if Pd is None:
            Pd = P
        BH = B * H
        sc = dk ** -0.5
        nrel = 2 * window + 1
        dO = _c(dO)
        dV = torch.empty_like(v)
        S.gemm(dO, Pd, (dk * T, T, 1), (T * T, T, 1), BH, dk, T, T, out=dV, c_strides=(dk * T, T, 1))
        dP = S.gemm(dO, v, (dk * T, 1, T), (dk * T, T, 1), BH, T, T, dk)
        dEk = dEv = None
        if window:
            ek = _c(emb_k.view(-1, dk))
            ev = _c(emb_v.view(-1, dk))
            dpband = S.gemm(dO, ev, (dk * T, 1, T), (0, 1, dk), BH, T, nrel, dk)
            S.band_scatter_add(dP, dpband, BH * T, T, window)
            dEv_b = S.gemm(pband, dO, (T * nrel, 1, nrel), (dk * T, 1, T), BH, nrel, dk, T, split_k_atomic=True)      # [BH, nrel, dk]
            dEv = S.reduce_bct(dEv_b.view(BH, nrel * dk, 1), 0).view(emb_v.shape)
        S.attn_softmax_bwd(P, dP, B, H, T, drop_u, p_drop, mask, mask_mode)          # dP(d) -> dS in place
        dS = dP
        dQ = torch.empty_like(q)
        S.gemm(k, dS, (dk * T, T, 1), (T * T, 1, T), BH, dk, T, T, out=dQ, c_strides=(dk * T, T, 1), alpha=sc)
        dK = torch.empty_like(k)
        S.gemm(q, dS, (dk * T, T, 1), (T * T, T, 1), BH, dk, T, T, out=dK, c_strides=(dk * T, T, 1), alpha=sc)
        if window:
            drel = S.band_gather(dS, BH * T, T, window)
            S.gemm(ek, drel, (0, 1, dk), (T * nrel, 1, nrel), BH, dk, T, nrel, out=dQ, c_strides=(dk * T, T, 1), alpha=sc,
                   beta=1.0)
            dEk_b = S.gemm(drel, q, (T * nrel, 1, nrel), (dk * T, 1, T), BH, nrel, dk, T, alpha=sc, split_k_atomic=True)
            dEk = S.reduce_bct(dEk_b.view(BH, nrel * dk, 1), 0).view(emb_k.shape)
        return dQ, dK, dV, dEk, dEv, None, None, None, None, None, None


def attention(q, k, v, n_heads, emb_rel_k=None, emb_rel_v=None, window=0, mask=None, mask_mode=0, drop_u=None, p_drop=0.0):
    return _Attention.apply(q, k, v, emb_rel_k, emb_rel_v, mask, n_heads, window or 0, mask_mode, drop_u, float(p_drop))


class _AttentionQKV(Function):
    """_Attention on ONE fused projection output qkv [B, 3*H*dk, T] whose rows are ordered (head, {q, k, v}, d): head h of item b has
    its q / k / v blocks at ((b*H + h)*3 + {0, 1, 2}) * dk * T — one batch stride for the GEMMs, and the backward writes dQ, dK, dV
    straight into ONE [B, 3C, T] gradient buffer, which is the fused conv's dy.  With the q / k / v projections as one 3C-row conv
    (modules/attentions.MultiHeadAttention.forward_train) a layer saves two forward convs, two input-gradient convs, two
    weight-gradient launches and the two accumulation adds of x's three consumers."""

    @staticmethod
    def forward(ctx, qkv, emb_k, emb_v, mask, n_heads, window, mask_mode, drop_u, p_drop):
        qkv = _c(qkv)
        B, C3, T = qkv.shape
        H = n_heads
        dk = C3 // (3 * H)
        BH, BS = B * H, 3 * dk * T
        sc = dk ** -0.5
        v5 = qkv.view(BH, 3, dk, T)
        q, k, v = v5[:, 0], v5[:, 1], v5[:, 2]
        qs = (BS, 1, T)          # A[m=i, k=d] = q[bh, d, i]
        P = S.gemm(q, k, qs, (BS, T, 1), BH, T, T, dk, alpha=sc)
        rel = None
        if window:
            ek = _c(emb_k.view(-1, dk))
            ev = _c(emb_v.view(-1, dk))
            rel = S.gemm(q, ek, qs, (0, 1, dk), BH, T, 2 * window + 1, dk, alpha=sc)
        hashed = isinstance(drop_u, S.HashDraw)
        if drop_u is not None and not hashed:
            drop_u = _c(drop_u)
            if drop_u.numel() != P.numel():
                raise S.SvcError(f"attention dropout draws {tuple(drop_u.shape)} do not match [B,H,T,T] = {(B, H, T, T)}")
        Pd = S.attn_softmax_fwd(P, rel, mask, B, H, T, window, mask_mode, drop_u, p_drop)
        out = torch.empty((B, H * dk, T), device=qkv.device, dtype=torch.float32)
        S.gemm(v, Pd, (BS, T, 1), (T * T, 1, T), BH, dk, T, T, out=out, c_strides=(dk * T, T, 1))
        pband = None
        if window:
            pband = S.band_gather(Pd, BH * T, T, window)
            S.gemm(ev, pband, (0, 1, dk), (T * (2 * window + 1), 1, 2 * window + 1), BH, dk, T, 2 * window + 1, out=out,
                   c_strides=(dk * T, T, 1), beta=1.0)
        ctx.save_for_backward(qkv, P, pband, emb_k, emb_v, None if hashed else drop_u, Pd if drop_u is not None else None, mask)
        ctx.cfg = (B, H, dk, T, window, p_drop, mask_mode)
        ctx.hash_draw = drop_u if hashed else None
        return out

    @staticmethod
    def backward(ctx, dO):
        qkv, P, pband, emb_k, emb_v, drop_u, Pd, mask = ctx.saved_tensors
        B, H, dk, T, window, p_drop, mask_mode = ctx.cfg
        if ctx.hash_draw is not None:
            drop_u = ctx.hash_draw
        if Pd is None:
            Pd = P
        BH, BS = B * H, 3 * dk * T
        sc = dk ** -0.5
        nrel = 2 * window + 1
        dO = _c(dO)
        v5 = qkv.view(BH, 3, dk, T)
        q, k, v = v5[:, 0], v5[:, 1], v5[:, 2]
        dqkv = torch.empty_like(qkv)
        d5 = dqkv.view(BH, 3, dk, T)
        dQ, dK, dV = d5[:, 0], d5[:, 1], d5[:, 2]
        S.gemm(dO, Pd, (dk * T, T, 1), (T * T, T, 1), BH, dk, T, T, out=dV, c_strides=(BS, T, 1))
        dP = S.gemm(dO, v, (dk * T, 1, T), (BS, T, 1), BH, T, T, dk)
        dEk = dEv = None
        if window:
            ek = _c(emb_k.view(-1, dk))
            ev = _c(emb_v.view(-1, dk))
            dpband = S.gemm(dO, ev, (dk * T, 1, T), (0, 1, dk), BH, T, nrel, dk)
            S.band_scatter_add(dP, dpband, BH * T, T, window)
            dEv_b = S.gemm(pband, dO, (T * nrel, 1, nrel), (dk * T, 1, T), BH, nrel, dk, T, split_k_atomic=True)      # [BH, nrel, dk]
            dEv = S.reduce_bct(dEv_b.view(BH, nrel * dk, 1), 0).view(emb_v.shape)
        S.attn_softmax_bwd(P, dP, B, H, T, drop_u, p_drop, mask, mask_mode)          # dP(d) -> dS in place
        dS = dP
        S.gemm(k, dS, (BS, T, 1), (T * T, 1, T), BH, dk, T, T, out=dQ, c_strides=(BS, T, 1), alpha=sc)
        S.gemm(q, dS, (BS, T, 1), (T * T, T, 1), BH, dk, T, T, out=dK, c_strides=(BS, T, 1), alpha=sc)
        if window:
            drel = S.band_gather(dS, BH * T, T, window)
            S.gemm(ek, drel, (0, 1, dk), (T * nrel, 1, nrel), BH, dk, T, nrel, out=dQ, c_strides=(BS, T, 1), alpha=sc, beta=1.0)
            dEk_b = S.gemm(drel, q, (T * nrel, 1, nrel), (BS, 1, T), BH, nrel, dk, T, alpha=sc, split_k_atomic=True)
            dEk = S.reduce_bct(dEk_b.view(BH, nrel * dk, 1), 0).view(emb_k.shape)
        return dqkv, dEk, dEv, None, None, None, None, None, None


def attention_qkv(qkv, n_heads, emb_rel_k=None, emb_rel_v=None, window=0, mask=None, mask_mode=0, drop_u=None, p_drop=0.0):
    """attention() on a fused [B, 3C, T] projection whose rows are ordered (head, {q, k, v}, d) — see _AttentionQKV."""
    return _AttentionQKV.apply(qkv, emb_rel_k, emb_rel_v, mask, n_heads, window or 0, mask_mode, drop_u, float(p_drop))


class _Dropout(Function):
    """nn.Dropout(p) with the uniform draws `u` explicit: y = x * (u >= p ? 1/(1-p) : 0), one svc_ew_f32 launch each way."""

    @staticmethod
    def forward(ctx, x, u, p):
        ctx.save_for_backward(u)
        ctx.p = p
        return S.ew(S.EW_DROPOUT, x, u, alpha=p)

    @staticmethod
    def backward(ctx, dy):
        (u,) = ctx.saved_tensors
        return S.ew(S.EW_DROPOUT, dy, u, alpha=ctx.p), None, None


class _DropoutRng(Function):
    """nn.Dropout(p) with the keep decisions of a svc_hip.HashDraw: one svc_dropout_rng_f32 launch each way, no tensor of draws."""

    @staticmethod
    def forward(ctx, x, draw, p):
        ctx.draw, ctx.p = draw, p
        return S.dropout_rng(x, draw, p)

    @staticmethod
    def backward(ctx, dy):
        return S.dropout_rng(_c(dy), ctx.draw, ctx.p), None, None


def dropout(x, u, p):
    if isinstance(u, S.HashDraw):
        return _DropoutRng.apply(x, u, float(p))
    return _Dropout.apply(x, _c(u), float(p))


class _Embed(Function):
    @staticmethod
    def forward(ctx, idx, W):
        ctx.save_for_backward(idx)
        ctx.n = W.shape[0]
        return S.embed_fwd(idx, W)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return None, S.embed_bwd(idx, dy, ctx.n)


def embedding_bct(idx, W):
    """nn.Embedding(idx).transpose(1,2): idx [B,T] int64 -> [B,C,T]."""
    return _Embed.apply(idx, W)


class _Reparam(Function):
    @staticmethod
    def forward(ctx, stats, noise, mask, scale):
        stats = _c(stats)
        ctx.save_for_backward(stats, noise, mask)
        ctx.scale = scale
        return S.reparam(stats, noise, mask=mask, scale=scale)

    @staticmethod
    def backward(ctx, dz):
        stats, noise, mask = ctx.saved_tensors
        return S.reparam_bwd(stats, noise, mask, dz, ctx.scale), None, None, None


def reparam(stats, noise, mask, scale=1.0):
    return _Reparam.apply(stats, noise, mask, float(scale))


class _NsfSource(Function):
    @staticmethod
    def forward(ctx, f0, rand_ini, noise, lin_w, lin_b, upp, sr, sine_amp, noise_std):
        har, waves = S.nsf_source_train(f0, rand_ini, noise, lin_w.view(-1), lin_b.view(-1), upp, sr, sine_amp, noise_std)
        ctx.save_for_backward(waves, har)
        ctx.shapes = (lin_w.shape, lin_b.shape)
        return har

    @staticmethod
    def backward(ctx, dhar):
        waves, har = ctx.saved_tensors
        dw, db = S.nsf_linear_bwd(waves, har, dhar)
        return None, None, None, dw.view(ctx.shapes[0]), db.view(ctx.shapes[1]), None, None, None, None


def nsf_source(f0, rand_ini, noise, lin_w, lin_b, upp, sr, sine_amp=0.1, noise_std=0.003):
    return _NsfSource.apply(f0, rand_ini, noise, lin_w, lin_b, upp, float(sr), sine_amp, noise_std)


# ---------------------------------------------------------------------------------------------------------------
# scalar losses: sums computed by HIP reductions; the (0-dim) results are combined by the caller's scalar arithmetic
# ---------------------------------------------------------------------------------------------------------------
def _bscale(t, g):
    """t * g with g a 0-dim / 1-element device tensor (no host sync)."""
    flat = t.reshape(1, 1, -1)
    return S.ew_bct(S.EW_MUL, flat, g.reshape(1, 1, 1).float()).view(t.shape)


class _SumAbsDiff(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        ctx.save_for_backward(a, b)
        return S.f64_to_f32(S.reduce_scalar(S.RED_ABS_DIFF, a, b)).view(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        sgn = S.ew(S.EW_SIGN_MUL, S.ew(S.EW_ADD, a, b, alpha=1.0, beta=-1.0), alpha=1.0)
        da = _bscale(sgn, g)
        return (da if ctx.needs_input_grad[0] else None), (S.ew(S.EW_SCALE, da, alpha=-1.0) if ctx.needs_input_grad[1] else None)


class _SumSqDiff(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        ctx.save_for_backward(a, b)
        return S.f64_to_f32(S.reduce_scalar(S.RED_SQ_DIFF, a, b)).view(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = _bscale(S.ew(S.EW_ADD, a, b, alpha=2.0, beta=-2.0), g)
        return (da if ctx.needs_input_grad[0] else None), (S.ew(S.EW_SCALE, da, alpha=-1.0) if ctx.needs_input_grad[1] else None)


class _SumSq(Function):
    """sum(a^2) (one_minus=False) or sum((1-a)^2) (one_minus=True)."""

    @staticmethod
    def forward(ctx, a, one_minus):
        a = _c(a)
        ctx.save_for_backward(a)
        ctx.one_minus = one_minus
        return S.f64_to_f32(S.reduce_scalar(S.RED_SQ_ONE_MINUS if one_minus else S.RED_SQ, a)).view(())

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        d = S.ew(S.EW_SCALE, a, alpha=2.0, beta=-2.0) if ctx.one_minus else S.ew(S.EW_SCALE, a, alpha=2.0)
        return _bscale(d, g), None


class _KLSums(Function):
    """returns a 2-vector [sum(kl*mask), sum(mask)] (modules/losses.py:43-58)."""

    @staticmethod
    def forward(ctx, z_p, logs_q, m_p, logs_p, mask):
        z_p, logs_q, m_p, logs_p, mask = _c(z_p), _c(logs_q), _c(m_p), _c(logs_p), _c(mask)
        ctx.save_for_backward(z_p, m_p, logs_p, mask)
        return S.f64_to_f32(S.kl_fwd(z_p, logs_q, m_p, logs_p, mask))

    @staticmethod
    def backward(ctx, g):
        z_p, m_p, logs_p, mask = ctx.saved_tensors
        dz, dlq, dm, dlp = S.kl_bwd(z_p, m_p, logs_p, mask, _c(g[:1].float()))
        return dz, dlq, dm, dlp, None


class _WeightedSums(Function):
    """total = sum_i w_i * red_i(a_i[, b_i]) for a list of reduction terms (ops of svc_reduce_scalar_f64), accumulated on the
    device into ONE float64 cell (per_term=False) or one cell per term plus the total (per_term=True: the second output holds
    the weighted terms, for logging; it carries no gradient).  One zero-fill + one reduction launch per term + one or three
    conversions, instead of (zero-fill, reduce, convert, divide, add) per term in the caller's scalar arithmetic — the
    feature-matching loss alone has 41 terms (modules/losses.py:4-12)."""

    @staticmethod
    def forward(ctx, spec, per_term, *tensors):
        n = len(spec)
        ts, k = [], 0
        for op, w, nin in spec:
            ts.append(tuple(_c(t) for t in tensors[k:k + nin]))
            k += nin
        dev = tensors[0].device
        acc = torch.zeros(n + 1 if per_term else 1, device=dev, dtype=torch.float64)
        for i, ((op, w, nin), tt) in enumerate(zip(spec, ts)):
            S.reduce_scalar(op, tt[0], tt[1] if nin > 1 else None, scale=w, acc=acc[i + 1:i + 2] if per_term else acc)
        ctx.spec = spec
        ctx.save_for_backward(*[t for tt in ts for t in tt])
        if not per_term:
            vals = torch.empty(0, device=dev)
            total = S.f64_to_f32(acc).view(())
        else:
            vals = S.f64_to_f32(acc[1:])
            S.reduce_scalar(S.RED_SUM, vals, acc=acc[0:1])
            total = S.f64_to_f32(acc[0:1]).view(())
        ctx.mark_non_differentiable(vals)
        return total, vals

    @staticmethod
    def backward(ctx, g, _gv):
        saved = ctx.saved_tensors
        grads, k = [], 0
        for j, (op, w, nin) in enumerate(ctx.spec):
            a = saved[k]
            b = saved[k + 1] if nin > 1 else None
            need_a = ctx.needs_input_grad[2 + k]
            need_b = nin > 1 and ctx.needs_input_grad[3 + k]
            da = db = None
            if need_a or need_b:
                if op == S.RED_ABS_DIFF:
                    d = S.ew(S.EW_SIGN_MUL, S.ew(S.EW_ADD, a, b, alpha=1.0, beta=-1.0), alpha=w)
                elif op == S.RED_SQ_DIFF:
                    d = S.ew(S.EW_ADD, a, b, alpha=2.0 * w, beta=-2.0 * w)
                elif op == S.RED_SQ:
                    d = S.ew(S.EW_SCALE, a, alpha=2.0 * w)
                elif op == S.RED_SQ_ONE_MINUS:
                    d = S.ew(S.EW_SCALE, a, alpha=2.0 * w, beta=-2.0 * w)
                else:
                    raise S.SvcError(f"weighted_sums: no gradient for reduction op {op}")
                da = _bscale(d, g)
                if need_b:
                    db = S.ew(S.EW_SCALE, da, alpha=-1.0)
                if not need_a:
                    da = None
            grads.append(da)
            if nin > 1:
                grads.append(db)
            k += nin
        return (None, None) + tuple(grads)


def weighted_sums(terms, per_term=False):
    """terms: [(op, weight, a[, b])] with op one of S.RED_ABS_DIFF / RED_SQ_DIFF / RED_SQ / RED_SQ_ONE_MINUS.
    -> total (0-dim) or, with per_term, (total, [N] weighted terms without gradient)."""
    spec = tuple((t[0], float(t[1]), len(t) - 2) for t in terms)
    flat = [x for t in terms for x in t[2:]]
    total, vals = _WeightedSums.apply(spec, per_term, *flat)
    return (total, vals) if per_term else total


def sum_abs_diff(a, b):
    return _SumAbsDiff.apply(a, b)


def sum_sq_diff(a, b):
    return _SumSqDiff.apply(a, b)


def sum_sq(a):
    return _SumSq.apply(a, False)


def sum_sq_one_minus(a):
    return _SumSq.apply(a, True)


def kl_sums(z_p, logs_q, m_p, logs_p, mask):
    return _KLSums.apply(z_p, logs_q, m_p, logs_p, mask)


# ---------------------------------------------------------------------------------------------------------------
# STFT magnitude and mel (modules/mel_processing.py:40-83)
# ---------------------------------------------------------------------------------------------------------------
class _Gemm2D(Function):
    """Y[r, n] = sum_k X[r, k] * W[k, n]  for a fixed (non-trainable) basis W; rows r may be any leading shape."""

    @staticmethod
    def forward(ctx, x, W):
        x = _c(x)
        K, N = W.shape
        R = x.numel() // K
        ctx.save_for_backward(W)
        ctx.xshape = x.shape
        y = S.gemm(x, W, (0, K, 1), (0, N, 1), 1, R, N, K)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        (W,) = ctx.saved_tensors
        K, N = W.shape
        dy = _c(dy)
        R = dy.numel() // N
        dx = S.gemm(dy, W, (0, N, 1), (0, 1, N), 1, R, K, N)      # dy [R,N] x W^T [N,K]
        return dx.view(ctx.xshape), None


class _StftFrames(Function):
    @staticmethod
    def forward(ctx, y, win, NF, nfft, hop, pad):
        ctx.save_for_backward(win)
        ctx.cfg = (y.shape[1], hop, pad)
        return S.stft_frame(y, win, NF, nfft, hop, pad)

    @staticmethod
    def backward(ctx, d):
        (win,) = ctx.saved_tensors
        L, hop, pad = ctx.cfg
        return S.stft_frame_bwd(d, win, L, hop, pad), None, None, None, None, None


class _CMag(Function):
    @staticmethod
    def forward(ctx, re, im, eps):
        re, im = _c(re), _c(im)
        mag = S.cmag(re, im, eps)
        ctx.save_for_backward(re, im, mag)
        return mag

    @staticmethod
    def backward(ctx, d):
        re, im, mag = ctx.saved_tensors
        dre, dim = S.cmag_bwd(re, im, mag, d)
        return dre, dim, None


class _SnakeAlias(Function):
    """SnakeAlias (vdecoder/hifiganwithsnake/alias/act.py:125-130): fused up2 -> snake -> down2, forward and backward."""

    @staticmethod
    def forward(ctx, x, alpha, beta, taps):
        x = _c(x)
        ctx.save_for_backward(x, alpha, beta)
        ctx.taps = taps
        return S.snake_alias(x, alpha.detach(), beta.detach(), taps)

    @staticmethod
    def backward(ctx, dy):
        x, alpha, beta = ctx.saved_tensors
        dx, da, db = S.snake_alias_bwd(x, _c(dy), alpha.detach(), beta.detach(), ctx.taps)
        return dx, da, db, None


def snake_alias(x, alpha, beta, taps):
    return _SnakeAlias.apply(x, alpha, beta, tuple(taps))


class _RfftMag(Function):
    """|rFFT(frames)| with the 1e-6 floor (modules/mel_processing.py:61-63): rocFFT forward + fused magnitude; backward
    = magnitude gradient + rocFFT complex-to-real (the adjoint)."""

    @staticmethod
    def forward(ctx, frames, eps):
        shp = frames.shape
        z, mag = S.rfft_mag(frames.reshape(-1, shp[-1]), eps)
        ctx.save_for_backward(z, mag)
        ctx.n = shp[-1]
        ctx.shp = shp
        return mag.view(*shp[:-1], mag.shape[-1])

    @staticmethod
    def backward(ctx, d):
        z, mag = ctx.saved_tensors
        dx = S.rfft_mag_bwd(z, mag, _c(d).reshape(mag.shape), ctx.n)
        return dx.view(ctx.shp), None


def rfft_mag(frames, eps):
    return _RfftMag.apply(frames, float(eps))


class _LogClamp(Function):
    @staticmethod
    def forward(ctx, x, lo):
        x = _c(x)
        ctx.save_for_backward(x)
        ctx.lo = lo
        return S.ew(S.EW_LOG_CLAMP, x, alpha=lo)

    @staticmethod
    def backward(ctx, d):
        (x,) = ctx.saved_tensors
        return S.ew(S.EW_LOG_CLAMP_BWD, _c(d), x, alpha=ctx.lo), None


def gemm2d(x, W):
    return _Gemm2D.apply(x, W)


def stft_frames(y, win, NF, nfft, hop, pad):
    return _StftFrames.apply(y, win, NF, nfft, hop, pad)


def cmag(re, im, eps):
    return _CMag.apply(re, im, eps)


def log_clamp(x, lo):
    return _LogClamp.apply(x, float(lo))
