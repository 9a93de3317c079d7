"""ctypes binding of libsvc_hip.so (C-ABI declared in include/svc_hip.h).

This is the only place Python touches the native library.  There is NO fallback: if the shared object is missing
or a call fails, an exception is raised (the product path must fail loudly without the HIP extension).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsvc_hip.so")

EPI_PLAIN, EPI_GATE, EPI_RES_SKIP = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_TANH, ACT_LRELU = 0, 1, 2, 3

_f32p = C.c_void_p


class SvcError(RuntimeError):
    pass


class Conv1dArgs(C.Structure):
    _fields_ = [
        ("x", _f32p), ("w", _f32p), ("bias", _f32p), ("cond", _f32p), ("mask", _f32p), ("premask", _f32p),
        ("res", _f32p), ("y", _f32p), ("y2", _f32p),
        ("x_bs", C.c_longlong), ("x_cs", C.c_longlong), ("y_bs", C.c_longlong), ("y_cs", C.c_longlong),
        ("res_bs", C.c_longlong), ("res_cs", C.c_longlong), ("y2_bs", C.c_longlong), ("y2_cs", C.c_longlong),
        ("cond_bs", C.c_longlong), ("cond_cs", C.c_longlong), ("cond_ts", C.c_longlong),
        ("mask_bs", C.c_longlong), ("premask_bs", C.c_longlong),
        ("B", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("Tin", C.c_int), ("Tout", C.c_int),
        ("KS", C.c_int), ("dil", C.c_int), ("pad_left", C.c_int), ("CoutP", C.c_int),
        ("epi", C.c_int), ("post_act", C.c_int), ("res_mode", C.c_int), ("skip_from", C.c_int),
        ("pre_slope", C.c_float), ("post_slope", C.c_float), ("beta", C.c_float), ("out_div", C.c_float),
    ]


_lib = None


def lib():
    """Load (once) and return the native library; raises SvcError when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SvcError(
                f"{LIB_PATH} not found: build it with `python so-vits-svc_amd/csrc/build.py` "
                "(or __graft_entry__.build()); there is no CPU/PyTorch fallback")
        L = C.CDLL(LIB_PATH)
        L.svc_last_error.restype = C.c_char_p
        L.svc_abi_version.restype = C.c_int
        L.svc_device_info.argtypes = [C.c_char_p, C.c_int]
        L.svc_prof_enable.argtypes = [C.c_int]
        L.svc_prof_report.argtypes = [C.c_char_p, C.c_int]
        L.svc_pack_conv1d_weight.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_void_p]
        L.svc_pack_convt1d_weight.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.svc_conv1d_f32.argtypes = [C.POINTER(Conv1dArgs), C.c_void_p]
        _lib = L
    return _lib


EXPORTS = [
    "svc_last_error", "svc_abi_version", "svc_device_info", "svc_prof_enable", "svc_prof_reset", "svc_prof_report",
    "svc_pack_conv1d_weight", "svc_pack_convt1d_weight", "svc_conv1d_f32",
]


def check(rc, what=""):
    if rc != 0:
        raise SvcError(f"{what} failed ({rc}): {lib().svc_last_error().decode()}")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SvcError("svc_hip ops need tensors on the GPU (cuda:N on ROCm); there is no CPU fallback")
        if t is not None and t.dtype != torch.float32:
            raise SvcError(f"svc_hip ops are fp32; got {t.dtype}")


def round_up(a, m):
    return (a + m - 1) // m * m


# --------------------------------------------------------------------------------------------------------------
# weight packing
# --------------------------------------------------------------------------------------------------------------
def pack_conv1d_weight(v, g=None, gate_half=0):
    """v: [Cout, Cin, KS] (weight or weight_v); g: weight_g ([Cout,1,1]) or None -> packed [Cin, KS, CoutP]."""
    require_gpu(v, g)
    v = v.contiguous()
    Cout, Cin, KS = v.shape
    CoutP = Cout if gate_half else round_up(Cout, 32)
    dst = torch.empty((Cin, KS, CoutP), device=v.device, dtype=torch.float32)
    gg = g.contiguous().view(-1) if g is not None else None
    check(lib().svc_pack_conv1d_weight(ptr(v), ptr(gg), ptr(dst), Cout, Cin, KS, CoutP, gate_half, stream_ptr()),
          "pack_conv1d_weight")
    return dst


def pack_convt1d_weight(v, g=None):
    """v: [Cin, Cout, KS] ConvTranspose1d weight(_v); g: [Cin,1,1] or None -> packed [Cin, KS, CoutP]."""
    require_gpu(v, g)
    v = v.contiguous()
    Cin, Cout, KS = v.shape
    CoutP = round_up(Cout, 32)
    dst = torch.empty((Cin, KS, CoutP), device=v.device, dtype=torch.float32)
    gg = g.contiguous().view(-1) if g is not None else None
    check(lib().svc_pack_convt1d_weight(ptr(v), ptr(gg), ptr(dst), Cin, Cout, KS, CoutP, stream_ptr()),
          "pack_convt1d_weight")
    return dst


# --------------------------------------------------------------------------------------------------------------
# conv1d
# --------------------------------------------------------------------------------------------------------------
def _bct_strides(t):
    """(batch_stride, channel_stride) of a [B,C,T] tensor/view whose time stride is 1."""
    if t.dim() != 3 or (t.shape[2] > 1 and t.stride(2) != 1):
        raise SvcError(f"expected a [B,C,T] tensor with contiguous time, got shape {tuple(t.shape)} "
                       f"stride {t.stride()}")
    return t.stride(0), t.stride(1)


def conv1d(x, wp, Cout, KS, *, bias=None, dil=1, pad_left=0, Tout=None, pre_slope=1.0, premask=None, cond=None,
           mask=None, post_act=ACT_NONE, post_slope=0.0, res=None, res_mode=0, out=None, beta=0.0, out_div=1.0,
           epi=EPI_PLAIN, out2=None, skip_from=0):
    """Fused conv1d (see include/svc_hip.h).  x/res/out are [B,C,T] views (channel stride may be negative: use
    flip_view()); wp is a packed weight from pack_conv1d_weight; cond is [B,C,1|T]; mask/premask are [B,1,T]."""
    require_gpu(x, wp, bias, cond, mask, premask, res, out, out2)
    B, Cin, Tin = x.shape
    if wp.shape[0] != Cin or wp.shape[1] != KS:
        raise SvcError(f"packed weight {tuple(wp.shape)} does not match Cin={Cin} KS={KS}")
    if Tout is None:
        Tout = Tin
    a = Conv1dArgs()
    a.x, a.w, a.bias = ptr(x), ptr(wp), ptr(bias)
    a.x_bs, a.x_cs = _bct_strides(x)
    out_ch = Cout // 2 if epi == EPI_GATE else (skip_from if epi == EPI_RES_SKIP else Cout)
    if out is None:
        out = torch.empty((B, out_ch, Tout), device=x.device, dtype=torch.float32)
    a.y = ptr(out)
    a.y_bs, a.y_cs = _bct_strides(out)
    if cond is not None:
        a.cond = ptr(cond)
        a.cond_bs, a.cond_cs = cond.stride(0), cond.stride(1)
        a.cond_ts = cond.stride(2) if cond.shape[2] > 1 else 0
    if mask is not None:
        a.mask = ptr(mask)
        a.mask_bs = mask.stride(0)
    if premask is not None:
        a.premask = ptr(premask)
        a.premask_bs = premask.stride(0)
    if res is not None:
        a.res = ptr(res)
        a.res_bs, a.res_cs = _bct_strides(res)
    if out2 is not None:
        a.y2 = ptr(out2)
        a.y2_bs, a.y2_cs = _bct_strides(out2)
    a.B, a.Cin, a.Cout, a.Tin, a.Tout = B, Cin, Cout, Tin, Tout
    a.KS, a.dil, a.pad_left, a.CoutP = KS, dil, pad_left, wp.shape[2]
    a.epi, a.post_act, a.res_mode, a.skip_from = epi, post_act, res_mode, skip_from
    a.pre_slope, a.post_slope, a.beta, a.out_div = pre_slope, post_slope, beta, out_div
    check(lib().svc_conv1d_f32(C.byref(a), stream_ptr()), "conv1d")
    return out


def flip_view(x):
    """Channel-reversed view of a [B,C,T] tensor (negative channel stride) as a raw-pointer carrying wrapper."""
    return FlipView(x)


class FlipView:
    """Duck-typed [B,C,T] view with reversed channels: element (b,c,t) = base[b, C-1-c, t].  torch has no
    negative strides, so this carries the pointer/strides by hand for the C-ABI."""

    def __init__(self, base, c0=None, c1=None):
        self.base = base
        B, Cb, T = base.shape
        self.c0 = Cb - 1 if c0 is None else c0  # base channel of view channel 0
        n = Cb if c1 is None else c1
        self.shape = (B, n, T)
        self.is_cuda = base.is_cuda
        self.dtype = base.dtype
        self.device = base.device

    def dim(self):
        return 3

    def stride(self, i=None):
        st = (self.base.stride(0), -self.base.stride(1), self.base.stride(2))
        return st if i is None else st[i]

    def data_ptr(self):
        return self.base.data_ptr() + 4 * self.c0 * self.base.stride(1)

    def narrow_c(self, start, length):
        """Sub-range of view channels [start, start+length)."""
        return FlipView(self.base, self.c0 - start, length)


# --------------------------------------------------------------------------------------------------------------
# profiling
# --------------------------------------------------------------------------------------------------------------
def prof_enable(on=True):
    check(lib().svc_prof_enable(1 if on else 0))


def prof_reset():
    check(lib().svc_prof_reset())


def prof_report():
    buf = C.create_string_buffer(1 << 16)
    n = lib().svc_prof_report(buf, len(buf))
    out = {}
    for line in buf.raw[:max(n, 0)].decode().splitlines():
        name, calls, ms, flop, byt = line.split()
        out[name] = dict(calls=int(calls), ms=float(ms), flop=float(flop), bytes=float(byt))
    return out


def device_info():
    buf = C.create_string_buffer(256)
    cus = lib().svc_device_info(buf, 256)
    return buf.value.decode(), cus
