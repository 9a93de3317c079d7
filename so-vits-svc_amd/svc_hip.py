"""ctypes binding of libsvc_hip.so (C-ABI declared in include/svc_hip.h).

This is the only place Python touches the native library.  There is NO fallback: if the shared object is missing
or a call fails, an exception is raised (the product path must fail loudly without the HIP extension).
"""
import ctypes as C
import math
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVC_HIP_LIB") or os.path.join(_HERE, "libsvc_hip.so")   # (override: kernel A/B builds)

EPI_PLAIN, EPI_GATE, EPI_RES_SKIP = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_TANH, ACT_LRELU, ACT_GELU = 0, 1, 2, 3, 4

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
        ("n_phase", C.c_int), ("y_ts", C.c_int), ("y_t0", C.c_int), ("y_len", C.c_int),
        ("w_phase_stride", C.c_longlong),
        ("pre_slope", C.c_float), ("post_slope", C.c_float), ("beta", C.c_float), ("out_div", C.c_float),
        ("mma", C.c_int),
        ("w_d4", _f32p),
    ]


class ConvT1dArgs(C.Structure):
    _fields_ = [
        ("x", _f32p), ("w", _f32p), ("bias", _f32p), ("res", _f32p), ("y", _f32p),
        ("x_bs", C.c_longlong), ("x_cs", C.c_longlong), ("y_bs", C.c_longlong), ("y_cs", C.c_longlong),
        ("res_bs", C.c_longlong), ("res_cs", C.c_longlong),
        ("B", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("Tin", C.c_int), ("Tout", C.c_int),
        ("KS", C.c_int), ("stride", C.c_int), ("padding", C.c_int), ("CoutP", C.c_int),
        ("pre_slope", C.c_float),
        ("w_d4", _f32p),
    ]


class Conv1dDirectArgs(C.Structure):
    _fields_ = [
        ("x", _f32p), ("w", _f32p), ("bias", _f32p), ("mask", _f32p), ("res", _f32p), ("y", _f32p),
        ("x_bs", C.c_longlong), ("x_cs", C.c_longlong), ("y_bs", C.c_longlong), ("y_cs", C.c_longlong),
        ("res_bs", C.c_longlong), ("res_cs", C.c_longlong), ("mask_bs", C.c_longlong),
        ("B", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("Tin", C.c_int), ("Tout", C.c_int),
        ("KS", C.c_int), ("dil", C.c_int), ("stride", C.c_int), ("pad_left", C.c_int), ("CoutP", C.c_int),
        ("post_act", C.c_int),
        ("pre_slope", C.c_float), ("post_slope", C.c_float),
    ]


class ResblockPairArgs(C.Structure):
    _fields_ = [("x", _f32p), ("w1", _f32p), ("b1", _f32p), ("w2", _f32p), ("b2", _f32p), ("y", _f32p),
                ("x_bs", C.c_longlong), ("x_cs", C.c_longlong), ("y_bs", C.c_longlong), ("y_cs", C.c_longlong),
                ("B", C.c_int), ("C", C.c_int), ("T", C.c_int), ("KS", C.c_int), ("dil1", C.c_int), ("CP", C.c_int),
                ("slope", C.c_float), ("beta", C.c_float), ("out_div", C.c_float)]


class Resblock16Args(C.Structure):
    _fields_ = [("x", _f32p), ("y", _f32p), ("w1", _f32p * 3), ("b1", _f32p * 3), ("w2", _f32p * 3), ("b2", _f32p * 3),
                ("x_bs", C.c_longlong), ("x_cs", C.c_longlong), ("y_bs", C.c_longlong), ("y_cs", C.c_longlong),
                ("B", C.c_int), ("T", C.c_int), ("KS", C.c_int), ("n_pairs", C.c_int), ("dil", C.c_int * 3), ("CP", C.c_int),
                ("slope", C.c_float), ("beta", C.c_float), ("out_div", C.c_float)]


class Conv1dHArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("bias", _f32p), ("res", C.c_void_p), ("y", C.c_void_p),
                ("B", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("Tin", C.c_int), ("Tq", C.c_int), ("Ty", C.c_int),
                ("KS", C.c_int), ("dil", C.c_int), ("pad_left", C.c_int),
                ("u", C.c_int), ("y_t0", C.c_int), ("RP", C.c_int), ("post_act", C.c_int),
                ("pre_slope", C.c_float), ("post_slope", C.c_float), ("beta", C.c_float), ("out_div", C.c_float),
                ("acc_scale", C.c_float)]


COUPLING_MAX_LAYERS = 8


class CouplingArgs(C.Structure):
    _fields_ = [("x", _f32p), ("x_bs", C.c_longlong), ("x_cs", C.c_longlong), ("mask", _f32p),
                ("cond", _f32p), ("cond_bs", C.c_longlong), ("cond_cs", C.c_longlong), ("cond_ts", C.c_int),
                ("w_pre", C.c_void_p), ("b_pre", _f32p),
                ("w_in", C.c_void_p * COUPLING_MAX_LAYERS), ("b_in", _f32p * COUPLING_MAX_LAYERS),
                ("w_rs", C.c_void_p * COUPLING_MAX_LAYERS), ("b_rs", _f32p * COUPLING_MAX_LAYERS),
                ("w_post", C.c_void_p), ("b_post", _f32p),
                ("s_pre", C.c_float), ("s_in", C.c_float * COUPLING_MAX_LAYERS), ("s_rs", C.c_float * COUPLING_MAX_LAYERS),
                ("s_post", C.c_float),
                ("B", C.c_int), ("T", C.c_int), ("channels", C.c_int), ("hidden", C.c_int), ("kernel_size", C.c_int),
                ("n_layers", C.c_int), ("reverse", C.c_int), ("planes", C.c_int)]


class AttentionArgs(C.Structure):
    _fields_ = [
        ("q", _f32p), ("k", _f32p), ("v", _f32p), ("emb_rel_k", _f32p), ("emb_rel_v", _f32p), ("mask", _f32p),
        ("out", _f32p),
        ("q_bs", C.c_longlong), ("q_cs", C.c_longlong), ("k_bs", C.c_longlong), ("k_cs", C.c_longlong),
        ("v_bs", C.c_longlong), ("v_cs", C.c_longlong), ("o_bs", C.c_longlong), ("o_cs", C.c_longlong),
        ("mask_bs", C.c_longlong),
        ("B", C.c_int), ("H", C.c_int), ("dk", C.c_int), ("T", C.c_int), ("window", C.c_int), ("mask_mode", C.c_int),
        ("ws", C.c_void_p), ("ws_bytes", C.c_longlong),
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
        if L.svc_abi_version() != ABI_VERSION:      # argument structs are read in full: a stale library must not be driven
            raise SvcError(f"{LIB_PATH} has ABI version {L.svc_abi_version()}, this binding was written against {ABI_VERSION} "
                           "(include/svc_hip.h SVC_ABI_VERSION): rebuild with `python so-vits-svc_amd/csrc/build.py`")
        L.svc_device_info.argtypes = [C.c_char_p, C.c_int]
        L.svc_prof_enable.argtypes = [C.c_int]
        L.svc_debug_empty_kernel.argtypes = [C.c_void_p]
        L.svc_prof_report.argtypes = [C.c_char_p, C.c_int]
        L.svc_pack_conv1d_weight.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_void_p]
        L.svc_pack_convt1d_weight.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p]
        L.svc_conv1d_f32.argtypes = [C.POINTER(Conv1dArgs), C.c_void_p]
        L.svc_pack_conv1d_d4.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.svc_pack_conv1d_d4_floats.argtypes = [C.c_int, C.c_int, C.c_int]
        L.svc_pack_conv1d_d4_floats.restype = C.c_longlong
        L.svc_conv1d_wants_d4.argtypes = [C.POINTER(Conv1dArgs)]
        L.svc_debug_bf16.argtypes = [C.c_int]
        L.svc_conv_transpose1d_f32.argtypes = [C.POINTER(ConvT1dArgs), C.c_void_p]
        L.svc_conv1d_direct_f32.argtypes = [C.POINTER(Conv1dDirectArgs), C.c_void_p]
        L.svc_resblock_pair_f32.argtypes = [C.POINTER(ResblockPairArgs), C.c_void_p]
        L.svc_nsf_source_scratch_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
        L.svc_nsf_source_scratch_bytes.restype = C.c_longlong
        L.svc_nsf_source_f32.argtypes = [_f32p] * 7 + [C.c_int] * 4 + [C.c_float] * 3 + [C.c_void_p]
        L.svc_f0_to_coarse.argtypes = [_f32p, C.c_void_p, C.c_longlong, C.c_void_p]
        L.svc_prenet_embed_f32.argtypes = [_f32p] * 11 + [C.c_int] * 3 + [C.c_void_p]
        L.svc_add_layernorm_f32.argtypes = [_f32p] * 6 + [C.c_int] * 3 + [C.c_float, C.c_void_p]
        L.svc_reparam_f32.argtypes = [_f32p] * 4 + [C.c_int] * 3 + [C.c_float, C.c_void_p]
        L.svc_attention_f32.argtypes = [C.POINTER(AttentionArgs), C.c_void_p]
        L.svc_pack_conv1d_h.argtypes = [_f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.svc_conv1d_h.argtypes = [C.POINTER(Conv1dHArgs), C.c_void_p]
        L.svc_snake_alias_h.argtypes = [C.c_void_p, C.c_void_p, _f32p, _f32p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.svc_resblock_pair_h.argtypes = [C.c_void_p, C.c_void_p, _f32p, C.c_void_p, _f32p, C.c_void_p] + [C.c_int] * 6 + \
            [C.c_float] * 3 + [C.c_void_p]
        L.svc_cvt_to_h.argtypes = [_f32p, _f32p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, C.c_int, C.c_int,
                                   C.c_int, C.c_void_p]
        L.svc_cvt_from_h.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.svc_conv_post_h.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                      C.c_void_p]
        for sfx in ("hl",):     # the split pipeline's entry points mirror the 16-bit ones
            getattr(L, "svc_pack_conv1d_" + sfx).argtypes = [_f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                                             C.c_void_p]
            getattr(L, "svc_conv1d_" + sfx).argtypes = L.svc_conv1d_h.argtypes
            getattr(L, "svc_cvt_to_" + sfx).argtypes = L.svc_cvt_to_h.argtypes
            getattr(L, "svc_cvt_from_" + sfx).argtypes = L.svc_cvt_from_h.argtypes
            getattr(L, "svc_conv_post_" + sfx).argtypes = L.svc_conv_post_h.argtypes
            getattr(L, "svc_resblock_pair_" + sfx).argtypes = [C.c_void_p, C.c_void_p, _f32p, C.c_void_p, _f32p, C.c_void_p] + \
                [C.c_int] * 6 + [C.c_float] * 5 + [C.c_void_p]
            getattr(L, "svc_snake_alias_" + sfx).argtypes = L.svc_snake_alias_h.argtypes
        L.svc_hl_range_flag.argtypes = [C.c_void_p]
        L.svc_resblock16_f32.argtypes = [C.POINTER(Resblock16Args), C.c_void_p]
        L.svc_coupling_fused_h.argtypes = [C.POINTER(CouplingArgs), C.c_void_p]
        L.svc_attention_ws_bytes.argtypes = [C.POINTER(AttentionArgs)]
        L.svc_attention_ws_bytes.restype = C.c_longlong
        L.svc_f0_norm_lf0_f32.argtypes = [_f32p] * 6 + [C.c_int] * 3 + [C.c_void_p]
        L.svc_lf0_to_f0_f32.argtypes = [_f32p, _f32p, C.c_longlong, C.c_void_p]
        L.svc_copy_bct_f32.argtypes = [_f32p] * 3 + [C.c_longlong] * 5 + [C.c_int] * 3 + [C.c_void_p]
        L.svc_snake_alias_f32.argtypes = [_f32p] * 4 + [C.POINTER(C.c_float)] + [C.c_longlong] * 4 + [C.c_int] * 3 + [C.c_void_p]
        L.svc_sinusoidal_emb_f32.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_void_p]
        L.svc_nsf_source_exact_f32.argtypes = [_f32p] * 7 + [C.c_int] * 4 + [C.c_float] * 3 + [C.c_void_p]
        L.svc_channel_norm_gelu_f32.argtypes = [_f32p] * 4 + [C.c_int] * 3 + [C.c_float, C.c_int, C.c_void_p]
        L.svc_channel_norm_gelu_len_f32.argtypes = [_f32p] * 3 + [C.c_void_p, _f32p] + [C.c_int] * 3 + [C.c_float, C.c_int, C.c_void_p]
        L.svc_resample_sinc_f32.argtypes = [_f32p] * 3 + [C.c_longlong] * 2 + [C.c_int] * 7 + [C.c_void_p]
        L.svc_snake_alias_bwd_f32.argtypes = [_f32p] * 4 + [C.POINTER(C.c_float)] + [_f32p] * 3 + [C.c_longlong] * 6 + \
            [C.c_int] * 3 + [C.c_void_p]
        L.svc_posconv_pack_f32.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.svc_posconv_f32.argtypes = [_f32p] * 4 + [C.c_int] * 6 + [C.c_void_p]
        L.svc_debug_set_conv_strip.argtypes = [C.c_int]
        if os.environ.get("SVC_CONV_STRIP"):     # A/B switch of the strip kernel (csrc/conv1d_strip.hip), read once at load
            L.svc_debug_set_conv_strip(int(os.environ["SVC_CONV_STRIP"]))
        _lib = L
    return _lib


ABI_VERSION = 6      # include/svc_hip.h SVC_ABI_VERSION (tests/test_abi_cpu.py keeps the two and the struct layouts in step)

EXPORTS = [
    "svc_last_error", "svc_abi_version", "svc_device_info", "svc_debug_empty_kernel", "svc_prof_enable", "svc_prof_reset", "svc_prof_report",
    "svc_pack_conv1d_weight", "svc_pack_convt1d_weight", "svc_conv1d_f32",
    "svc_debug_bf16", "svc_debug_wgrad_bf16_launches",
    "svc_conv_transpose1d_f32", "svc_pack_conv1d_d4", "svc_pack_conv1d_d4_floats", "svc_conv1d_wants_d4",
    "svc_conv1d_direct_f32", "svc_resblock_pair_f32", "svc_resblock16_f32", "svc_nsf_source_scratch_bytes", "svc_nsf_source_f32", "svc_f0_to_coarse",
    "svc_prenet_embed_f32", "svc_add_layernorm_f32", "svc_reparam_f32", "svc_attention_f32", "svc_attention_ws_bytes", "svc_pack_conv1d_h", "svc_conv1d_h", "svc_resblock_pair_h", "svc_snake_alias_h", "svc_debug_set_conv_h", "svc_cvt_to_h", "svc_cvt_from_h", "svc_conv_post_h", "svc_hl_range_flag", "svc_coupling_fused_h", "svc_debug_set_coupling_fused", "svc_pack_conv1d_hl", "svc_conv1d_hl", "svc_debug_set_conv_hl", "svc_resblock_pair_hl", "svc_snake_alias_hl", "svc_cvt_to_hl", "svc_cvt_from_hl", "svc_conv_post_hl", "svc_debug_set_attention_waves", "svc_posconv_pack_f32", "svc_posconv_f32", "svc_copy_bct_f32", "svc_f0_norm_lf0_f32", "svc_lf0_to_f0_f32",
    "svc_resample_sinc_f32", "svc_snake_alias_f32", "svc_snake_alias_bwd_f32", "svc_channel_norm_gelu_f32", "svc_channel_norm_gelu_len_f32", "svc_nsf_source_exact_f32", "svc_sinusoidal_emb_f32",
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
    dst.d4_ok = True       # a fresh tensor nobody rewrites in place: a lane-linear second pack may be derived from it (_d4_of)
    return dst


def pack_convt1d_weight(v, g, stride):
    """v: [Cin, Cout, KS] ConvTranspose1d weight(_v); g: [Cin,1,1] or None -> polyphase [stride, Cin, M, CoutP]."""
    require_gpu(v, g)
    v = v.contiguous()
    Cin, Cout, KS = v.shape
    CoutP = round_up(Cout, 32)
    M = (KS + stride - 1) // stride
    dst = torch.empty((stride, Cin, M, CoutP), device=v.device, dtype=torch.float32)
    gg = g.contiguous().view(-1) if g is not None else None
    check(lib().svc_pack_convt1d_weight(ptr(v), ptr(gg), ptr(dst), Cin, Cout, KS, CoutP, stride, stream_ptr()),
          "pack_convt1d_weight")
    dst.d4_ok = True
    return dst


# The second, lane-linear pack of a convolution's weights (svc_conv1d_args.w_d4): what the register-fed short-sequence kernel reads
# with 16-byte loads.  Made on first use by a short launch (the encoder / flow / pre convolutions of one utterance, the
# phases-as-rows ConvTranspose1d stages) and kept as an attribute of the standard pack, i.e. for as long as that one lives.
D4_MAX_COLS = 16384        # launches of more columns never take the register-fed kernel (conv1d_dispatch)
_D4 = os.environ.get("SVC_CONV_DIRECT4", "1") != "0"


def conv1d_d4(wp3):
    """wp3: a standard pack viewed [Cin, KS, CoutP] -> its lane-linear pack, or None where the kernel has no use for one."""
    Cin, KS, CoutP = wp3.shape
    if not _D4 or KS not in (1, 2, 3, 5, 7) or Cin % 32 or CoutP % 32:
        return None
    n = lib().svc_pack_conv1d_d4_floats(Cin, KS, CoutP)
    dst = torch.empty((n,), device=wp3.device, dtype=torch.float32)
    check(lib().svc_pack_conv1d_d4(ptr(wp3), ptr(dst), Cin, KS, CoutP, stream_ptr()), "pack_conv1d_d4")
    return dst


def _d4_of(wp, view3, cols, dil=1, args=None):
    """The cached lane-linear pack of `wp` for a launch of `cols` columns.  Made on the first launch that would READ one
    (`args`: the launch's Conv1dArgs — svc_conv1d_wants_d4 runs the dispatch rule without launching; the unit encoder's 768 <-> 3072
    projections on 500 frames take an LDS-staged tiling and get no second pack)."""
    if cols >= D4_MAX_COLS or dil != 1 or not _D4 or not getattr(wp, "d4_ok", False):
        return None      # (operand buffers that are rewritten in place — the training plans' — never carry the mark)
    d4 = getattr(wp, "d4", False)
    if d4 is False:
        if args is not None and not lib().svc_conv1d_wants_d4(C.byref(args)):
            return None                                   # asked again at the next launch: another length may take the other kernel
        d4 = wp.d4 = conv1d_d4(view3)
    return d4


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
           epi=EPI_PLAIN, out2=None, skip_from=0, mma=None):
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
    a.n_phase, a.y_ts, a.y_t0, a.y_len, a.w_phase_stride = 1, 1, 0, Tout, 0
    a.mma = _MMA if mma is None else mma
    a.w_d4 = ptr(_d4_of(wp, wp, B * Tout, dil, a))
    check(lib().svc_conv1d_f32(C.byref(a), stream_ptr()), "conv1d")
    return out



# ---- matrix-pipe operand format of the convolutions (svc_conv1d_args.mma): the engine's form of the reference's autocast region
MMA_F32, MMA_BF16, MMA_F16 = 0, 1, 2
_MMA = MMA_F32


def current_mma():
    return _MMA


class mma_mode:
    """`with mma_mode(MMA_BF16): ...` — convolutions (and, through svc_autograd, their input / weight gradients) issued inside
    the block take bf16 operands on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, where the shape has such a kernel: what
    `torch.autocast(dtype=torch.bfloat16)` is to the reference's training step (train.py:166,187,198).  Tensors stay fp32.
    Autograd ops record the mode of their forward and use it in their backward, whatever thread that runs on."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        global _MMA
        self.prev, _MMA = _MMA, self.mode
        return self

    def __exit__(self, *a):
        global _MMA
        _MMA = self.prev
        return False


def conv_transpose1d(x, wp, Cout, KS, stride, padding, *, bias=None, pre_slope=1.0, res=None, out=None):
    """ConvTranspose1d via polyphase MFMA sub-convolutions; wp from pack_convt1d_weight."""
    require_gpu(x, wp, bias, res, out)
    B, Cin, Tin = x.shape
    Tout = (Tin - 1) * stride - 2 * padding + KS
    if out is None:
        out = torch.empty((B, Cout, Tout), device=x.device, dtype=torch.float32)
    a = ConvT1dArgs()
    a.x, a.w, a.bias, a.res, a.y = ptr(x), ptr(wp), ptr(bias), ptr(res), ptr(out)
    a.x_bs, a.x_cs = _bct_strides(x)
    a.y_bs, a.y_cs = _bct_strides(out)
    if res is not None:
        a.res_bs, a.res_cs = _bct_strides(res)
    a.B, a.Cin, a.Cout, a.Tin, a.Tout = B, Cin, Cout, Tin, Tout
    a.KS, a.stride, a.padding, a.CoutP = KS, stride, padding, wp.shape[3]
    a.pre_slope = pre_slope
    if stride & (stride - 1) == 0 and 1 < stride <= 16 and Cin >= 256:   # phases-as-rows layout [Cin][M][stride*CoutP] (svc::convt_rows_layout); the narrow x2 stages do not take the pack (conv1d_mfma.hip)
        M = (KS + stride - 1) // stride
        a.w_d4 = ptr(_d4_of(wp, wp.view(Cin, M, stride * wp.shape[3]), 0))
    check(lib().svc_conv_transpose1d_f32(C.byref(a), stream_ptr()), "conv_transpose1d")
    return out


def conv1d_direct(x, wp, Cout, KS, *, bias=None, stride=1, dil=1, pad_left=0, Tout=None, pre_slope=1.0,
                  post_act=ACT_NONE, post_slope=0.0, mask=None, res=None, out=None):
    """VALU direct conv (noise_convs, conv_post, Cin=1 / Cout=1 layers); wp from pack_conv1d_weight."""
    require_gpu(x, wp, bias, mask, res, out)
    B, Cin, Tin = x.shape
    if Tout is None:
        Tout = (Tin + 2 * pad_left - dil * (KS - 1) - 1) // stride + 1
    if out is None:
        out = torch.empty((B, Cout, Tout), device=x.device, dtype=torch.float32)
    a = Conv1dDirectArgs()
    a.x, a.w, a.bias, a.mask, a.res, a.y = ptr(x), ptr(wp), ptr(bias), ptr(mask), ptr(res), ptr(out)
    a.x_bs, a.x_cs = _bct_strides(x)
    a.y_bs, a.y_cs = _bct_strides(out)
    if res is not None:
        a.res_bs, a.res_cs = _bct_strides(res)
    if mask is not None:
        a.mask_bs = mask.stride(0)
    a.B, a.Cin, a.Cout, a.Tin, a.Tout = B, Cin, Cout, Tin, Tout
    a.KS, a.dil, a.stride, a.pad_left, a.CoutP, a.post_act = KS, dil, stride, pad_left, wp.shape[2], post_act
    a.pre_slope, a.post_slope = pre_slope, post_slope
    check(lib().svc_conv1d_direct_f32(C.byref(a), stream_ptr()), "conv1d_direct")
    return out


RESBLOCK_PAIR_BUILT = (16, 32)          # channel counts svc_resblock_pair_f32 is instantiated for
# ... and the ones the decoder routes to it: measured on MI355X (profiles/r02_f_resblock_pair_fused_vs_two_launches.txt) the
# fused pair wins at C = 16 (HBM-bound stage: 50/69/88 us vs 68/82/99 us for K = 3/7/11) and still loses at C = 32, where the
# two-launch path already runs at 67-83 TFLOP/s and the fused kernel's column tiles do not fill its 8 waves (52-58 TFLOP/s)
RESBLOCK_PAIR_CHANNELS = (16,)
RESBLOCK_PAIR_KERNELS = (3, 7, 11)


def resblock_pair(x, w1p, b1, w2p, b2, KS, dil1, *, slope=0.1, out=None, beta=0.0, out_div=1.0):
    """y = conv2(lrelu(conv1(lrelu(x)) + b1)) + b2 + x in one launch (C in RESBLOCK_PAIR_CHANNELS, see include/svc_hip.h);
    w1p / w2p: packed weights of the two convs."""
    require_gpu(x, w1p, b1, w2p, b2, out)
    B, Cc, T = x.shape
    if out is None:
        out = torch.empty((B, Cc, T), device=x.device, dtype=torch.float32)
    a = ResblockPairArgs()
    a.x, a.w1, a.b1, a.w2, a.b2, a.y = ptr(x), ptr(w1p), ptr(b1), ptr(w2p), ptr(b2), ptr(out)
    a.x_bs, a.x_cs = _bct_strides(x)
    a.y_bs, a.y_cs = _bct_strides(out)
    if tuple(w1p.shape) != tuple(w2p.shape) or w1p.shape[0] != Cc or w1p.shape[1] != KS:
        raise SvcError(f"resblock_pair: packed weights {tuple(w1p.shape)} / {tuple(w2p.shape)} do not match C={Cc} KS={KS}")
    a.B, a.C, a.T, a.KS, a.dil1, a.CP = B, Cc, T, KS, dil1, w1p.shape[2]
    a.slope, a.beta, a.out_div = slope, beta, out_div
    check(lib().svc_resblock_pair_f32(C.byref(a), stream_ptr()), "resblock_pair")
    return out


RESBLOCK16 = os.environ.get("SVC_RESBLOCK16", "1") != "0"      # whole 16-channel ResBlock1 in one launch (0: one launch per pair, A/B)


def resblock16(x, pairs, KS, dils, *, slope=0.1, out=None, beta=0.0, out_div=1.0):
    """All dilation pairs of a 16-channel ResBlock1 in ONE launch (svc_resblock16_f32): pairs = [(w1p, b1, w2p, b2), ...] packed as for
    resblock_pair; out = (block(x) + beta * out) / out_div.  Returns None when the dilations do not fit the kernel's tile (the caller
    keeps the pair launches)."""
    B, Cc, T = x.shape
    if Cc != 16 or len(dils) != len(pairs):
        raise SvcError("resblock16: 16 channels, one dilation per pair")
    if tuple(int(d) for d in dils) != (1, 3, 5) or KS not in (3, 7, 11):
        return None
    if out is None:
        out = torch.empty((B, Cc, T), device=x.device, dtype=torch.float32)
    a = Resblock16Args()
    a.x, a.y = ptr(x), ptr(out)
    for j, (w1p, b1, w2p, b2) in enumerate(pairs):
        require_gpu(x, w1p, b1, w2p, b2, out)
        if tuple(w1p.shape) != tuple(w2p.shape) or w1p.shape[0] != 16 or w1p.shape[1] != KS:
            raise SvcError(f"resblock16: packed weights {tuple(w1p.shape)} / {tuple(w2p.shape)} do not match C=16 KS={KS}")
        a.w1[j], a.b1[j], a.w2[j], a.b2[j] = w1p.data_ptr(), (b1.data_ptr() if b1 is not None else None), w2p.data_ptr(), \
            (b2.data_ptr() if b2 is not None else None)
        a.dil[j] = int(dils[j])
    a.x_bs, a.x_cs = _bct_strides(x)
    a.y_bs, a.y_cs = _bct_strides(out)
    a.B, a.T, a.KS, a.n_pairs, a.CP = B, T, KS, len(pairs), pairs[0][0].shape[2]
    a.slope, a.beta, a.out_div = slope, beta, out_div
    check(lib().svc_resblock16_f32(C.byref(a), stream_ptr()), "resblock16")
    return out


# --------------------------------------------------------------------------------------------------------------
# 16-bit decoder pipeline (csrc/conv1d_h.hip): fp16 activations in the blocked layout [B, C/8, T, 8]
# --------------------------------------------------------------------------------------------------------------
def _hptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _require_gpu_h(*tensors):
    """require_gpu for the 16-bit pipeline's calls: every tensor on the GPU, fp16 (blocked activations / packs) or fp32 (bias,
    dense weights, plain tensors being converted)."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SvcError("svc_hip ops need tensors on the GPU (cuda:N on ROCm); there is no CPU fallback")
        if t is not None and t.dtype not in (torch.float16, torch.float32):
            raise SvcError(f"svc_hip 16-bit ops take fp16 / fp32 tensors; got {t.dtype}")


def _check_h(t, what, split=None):
    """A blocked fp16 activation: [B, C/8, T, 8], or [2, B, C/8, T, 8] (hi plane, lo plane) on the split pipeline."""
    ok = t.dtype == torch.float16 and t.is_contiguous() and t.shape[-1] == 8 and \
        (t.dim() == 4 or (t.dim() == 5 and t.shape[0] == 2))
    if ok and split is not None:
        ok = (t.dim() == 5) == bool(split)
    if not ok:
        raise SvcError(f"{what}: expected a contiguous fp16 [B, C/8, T, 8] (or split [2, B, C/8, T, 8]) tensor, got {t.dtype} {tuple(t.shape)}")


def is_split(t):
    """True for a tensor of the split (hi / lo plane) pipeline (csrc/conv1d_hl.hip)."""
    return t.dim() == 5


def pack_conv1d_h(w, u=1, split=False):
    """Dense fp32 weight (weight norm folded) -> the fp16 operand pack of svc_conv1d_h.  u == 1: Conv1d [Cout, Cin, K];
    u > 1: ConvTranspose1d [Cin, Cout, K] with stride u (phases as rows).  split: the [2, ...] hi / lo pack of svc_conv1d_hl."""
    _require_gpu_h(w)
    w = w.detach().float().contiguous()
    if u > 1:
        Cin, Cout, K = w.shape
        taps, R = (K + u - 1) // u, u * Cout
    else:
        Cout, Cin, K = w.shape
        taps, R = K, Cout
    RP = round_up(R, 128)
    dst = torch.empty(((2,) if split else ()) + (Cin // 16, taps, RP, 16), device=w.device, dtype=torch.float16)
    if split:
        # per-tensor power-of-two scale (exact): max |w| -> [2^13, 2^14), so every weight within 2^-17 of the largest keeps its 22 bits
        # whatever the tensor's magnitude; the convolution multiplies its accumulators by 1 / scale (`acc_scale`, read off the pack).
        # One host read per pack (packs are cached per parameter version, never built inside a capture).
        amax = float(w.abs().max())
        if not math.isfinite(amax):
            raise SvcError("pack_conv1d_h(split=True): the weight holds inf / nan")
        scale = 2.0 ** (14 - math.frexp(amax)[1]) if amax > 0.0 else 1.0
        scale = min(max(scale, 2.0 ** -100), 2.0 ** 100)
        check(lib().svc_pack_conv1d_hl(ptr(w), _hptr(dst), Cout, Cin, K, u, RP, scale, stream_ptr()), "pack_conv1d_hl")
        dst.acc_scale = 1.0 / scale
        return dst
    check(lib().svc_pack_conv1d_h(ptr(w), _hptr(dst), Cout, Cin, K, u, RP, stream_ptr()), "pack_conv1d_h")
    return dst


def _acc_scale(wp):
    return float(getattr(wp, "acc_scale", 1.0))


def hl_range_flag(flag):
    """Register (or, with None, withdraw) the int32 device word into which the split pipeline's launches of THIS host thread report a
    value outside the fp16 range (include/svc_hip.h, RANGE).  The pointer is taken at launch time: launches captured into a hipGraph
    keep reporting into the same tensor, which the caller must keep alive."""
    if flag is not None and not (flag.is_cuda and flag.dtype == torch.int32 and flag.numel() >= 1 and flag.is_contiguous()):
        raise SvcError("hl_range_flag: expected an int32 tensor on the GPU")
    check(lib().svc_hl_range_flag(flag.data_ptr() if flag is not None else None), "hl_range_flag")


def conv1d_h(x, wp, Cout, *, bias=None, dil=1, pad_left=0, Tout=None, pre_slope=1.0, post_slope=None, res=None, out=None, beta=0.0,
             out_div=1.0):
    """Conv1d on the 16-bit pipeline: x / res / out are blocked fp16 [B, C/8, T, 8]; wp from pack_conv1d_h.  With split tensors
    ([2, B, C/8, T, 8] and a split pack) the same call runs the split pipeline (svc_conv1d_hl)."""
    _require_gpu_h(x, wp, bias, res, out)
    _check_h(x, "conv1d_h")
    sp = is_split(x)
    B, CB, Tin, _ = x.shape[-4:]
    KS = wp.shape[-3]
    if wp.shape[-4] * 16 != CB * 8 or (wp.dim() == 5) != sp:
        raise SvcError(f"conv1d_h: packed weight {tuple(wp.shape)} does not match the input {tuple(x.shape)}")
    if Tout is None:
        Tout = Tin
    if out is None:
        out = torch.empty(((2,) if sp else ()) + (B, Cout // 8, Tout, 8), device=x.device, dtype=torch.float16)
    _check_h(out, "conv1d_h out", sp)
    a = Conv1dHArgs()
    a.x, a.w, a.bias, a.res, a.y = _hptr(x), _hptr(wp), ptr(bias), _hptr(res), _hptr(out)
    if res is not None:
        _check_h(res, "conv1d_h res", sp)
    a.B, a.Cin, a.Cout, a.Tin, a.Tq, a.Ty = B, CB * 8, Cout, Tin, Tout, Tout
    a.KS, a.dil, a.pad_left, a.u, a.y_t0, a.RP = KS, dil, pad_left, 1, 0, wp.shape[-2]
    a.post_act = ACT_LRELU if post_slope is not None else ACT_NONE
    a.pre_slope, a.post_slope, a.beta, a.out_div = pre_slope, post_slope or 0.0, beta, out_div
    a.acc_scale = _acc_scale(wp)
    check((lib().svc_conv1d_hl if sp else lib().svc_conv1d_h)(C.byref(a), stream_ptr()), "conv1d_h")
    return out


RESBLOCK_PAIR_H_MAX_C = 128
RESBLOCK_PAIR_HL_MAX_C = int(os.environ.get("SVC_PAIR_HL_MAX_C", "128"))   # split planes (the 128-channel form: one eight-wave workgroup per CU)


def resblock_pair_h(x, w1p, b1, w2p, b2, dil1, *, slope=0.1, out=None, beta=0.0, out_div=1.0):
    """out = (beta * out + conv2(lrelu(conv1(lrelu(x)) + b1)) + b2 + x) / out_div on blocked fp16 tensors, one launch
    (svc_resblock_pair_h); w1p / w2p from pack_conv1d_h, both with the same tap count."""
    _require_gpu_h(x, w1p, b1, w2p, b2, out)
    _check_h(x, "resblock_pair_h")
    sp = is_split(x)
    B, CB, T, _ = x.shape[-4:]
    KS = w1p.shape[-3]
    if tuple(w1p.shape) != tuple(w2p.shape) or w1p.shape[-4] * 16 != CB * 8 or (w1p.dim() == 5) != sp:
        raise SvcError(f"resblock_pair_h: packed weights {tuple(w1p.shape)} / {tuple(w2p.shape)} do not match the input {tuple(x.shape)}")
    if sp and CB * 8 > RESBLOCK_PAIR_HL_MAX_C:
        raise SvcError(f"resblock_pair_h: the split form is built for up to {RESBLOCK_PAIR_HL_MAX_C} channels")
    if out is None:
        out = torch.empty_like(x)
    _check_h(out, "resblock_pair_h out", sp)
    if out.data_ptr() == x.data_ptr():
        raise SvcError("resblock_pair_h: x and out may not alias (neighbouring workgroups read x's halo)")
    if sp:
        check(lib().svc_resblock_pair_hl(_hptr(x), _hptr(w1p), ptr(b1), _hptr(w2p), ptr(b2), _hptr(out), B, CB * 8, T, KS, dil1,
                                         w1p.shape[-2], slope, beta, out_div, _acc_scale(w1p), _acc_scale(w2p), stream_ptr()),
              "resblock_pair_hl")
        return out
    check(lib().svc_resblock_pair_h(_hptr(x), _hptr(w1p), ptr(b1), _hptr(w2p), ptr(b2), _hptr(out), B, CB * 8, T, KS, dil1, w1p.shape[-2],
                                    slope, beta, out_div, stream_ptr()), "resblock_pair_h")
    return out


def snake_alias_h(xh, alpha, beta, taps, out=None):
    """SnakeAlias on a blocked fp16 tensor (svc_snake_alias_h); out may not be xh (a tile reads its neighbours' halo)."""
    _require_gpu_h(xh, alpha, beta, out)
    _check_h(xh, "snake_alias_h")
    sp = is_split(xh)
    B, CB, T, _ = xh.shape[-4:]
    if alpha.numel() < CB * 8:       # zero-padded channels (svc_nn.Conv1d.packed_h): x = 0 there and snake(0) = 0 for any parameters
        padc = CB * 8 - alpha.numel()
        alpha = torch.nn.functional.pad(alpha.detach().float(), (0, padc))
        beta = torch.nn.functional.pad(beta.detach().float(), (0, padc))
    if out is None:
        out = torch.empty_like(xh)
    _check_h(out, "snake_alias_h out", sp)
    if out.data_ptr() == xh.data_ptr():
        raise SvcError("snake_alias_h: in-place use is not supported")
    tp = (C.c_float * 12)(*[float(v) for v in taps])
    fn = lib().svc_snake_alias_hl if sp else lib().svc_snake_alias_h
    check(fn(_hptr(xh), _hptr(out), ptr(alpha), ptr(beta), tp, B, CB * 8, T, stream_ptr()), "snake_alias_h")
    return out


def conv_transpose1d_h(x, wp, Cout, K, stride, padding, *, bias=None, pre_slope=1.0, res=None, out=None):
    """ConvTranspose1d on the 16-bit pipeline (phases as rows); wp from pack_conv1d_h(w, u=stride)."""
    _require_gpu_h(x, wp, bias, res, out)
    _check_h(x, "conv_transpose1d_h")
    sp = is_split(x)
    B, CB, Tin, _ = x.shape[-4:]
    M = wp.shape[-3]
    if (wp.dim() == 5) != sp:
        raise SvcError("conv_transpose1d_h: split input needs a split pack (and the other way round)")
    Lout = (Tin - 1) * stride - 2 * padding + K
    if out is None:
        out = torch.empty(((2,) if sp else ()) + (B, Cout // 8, Lout, 8), device=x.device, dtype=torch.float16)
    _check_h(out, "conv_transpose1d_h out", sp)
    if res is not None:
        _check_h(res, "conv_transpose1d_h res", sp)
    a = Conv1dHArgs()
    a.x, a.w, a.bias, a.res, a.y = _hptr(x), _hptr(wp), ptr(bias), _hptr(res), _hptr(out)
    a.B, a.Cin, a.Cout, a.Tin, a.Ty = B, CB * 8, Cout, Tin, Lout
    a.Tq = (Lout - 1 + padding) // stride + 1
    a.KS, a.dil, a.pad_left, a.u, a.y_t0, a.RP = M, 1, M - 1, stride, -padding, wp.shape[-2]
    a.post_act, a.pre_slope, a.post_slope, a.beta, a.out_div = ACT_NONE, pre_slope, 0.0, 0.0, 1.0
    a.acc_scale = _acc_scale(wp)
    check((lib().svc_conv1d_hl if sp else lib().svc_conv1d_h)(C.byref(a), stream_ptr()), "conv_transpose1d_h")
    return out


def to_h(x, add=None, out=None, split=False, pad16=False):
    """fp32 [B, C, T] (time-contiguous view) (+ add) -> blocked fp16 [B, C/8, T, 8]; split: -> [2, B, C/8, T, 8], hi and lo planes
    with hi + lo = the fp32 value to 22 bits."""
    _require_gpu_h(x, add, out)
    B, Cc, T = x.shape
    if pad16 and Cc % 16:
        # channel counts that are not multiples of 16 (the tiny template's decoder) travel zero-padded through the 16-bit pipeline
        # (svc_nn.Conv1d.packed_h): pad the fp32 source here — data movement only, and only on that compatibility path
        padc = (-Cc) % 16
        x = torch.nn.functional.pad(x.contiguous() if isinstance(x, torch.Tensor) else copy_bct(x), (0, 0, 0, padc))
        if add is not None:
            add = torch.nn.functional.pad(add.contiguous(), (0, 0, 0, padc))
        Cc += padc
    if out is None:
        out = torch.empty(((2,) if split else ()) + (B, Cc // 8, T, 8), device=x.device, dtype=torch.float16)
    _check_h(out, "to_h out", split)
    xb, xc = _bct_strides(x)
    ab, ac = _bct_strides(add) if add is not None else (0, 0)
    fn = lib().svc_cvt_to_hl if split else lib().svc_cvt_to_h
    check(fn(ptr(x), ptr(add), _hptr(out), xb, xc, ab, ac, B, Cc, T, stream_ptr()), "cvt_to_h")
    return out


def from_h(xh):
    """Blocked fp16 [B, C/8, T, 8] -> fp32 [B, C, T]."""
    _require_gpu_h(xh)
    _check_h(xh, "from_h")
    B, CB, T, _ = xh.shape[-4:]
    out = torch.empty((B, CB * 8, T), device=xh.device, dtype=torch.float32)
    fn = lib().svc_cvt_from_hl if is_split(xh) else lib().svc_cvt_from_h
    check(fn(_hptr(xh), ptr(out), B, CB * 8, T, stream_ptr()), "cvt_from_h")
    return out


def conv_post_h(xh, w, bias, KS, pad, pre_slope=0.01, act=None):
    """leaky_relu -> Conv1d(C -> 1) -> act on a blocked fp16 input; fp32 arithmetic and output [B, 1, T]."""
    _require_gpu_h(xh, w, bias)
    _check_h(xh, "conv_post_h")
    B, CB, T, _ = xh.shape[-4:]
    out = torch.empty((B, 1, T), device=xh.device, dtype=torch.float32)
    w = w.detach().float().contiguous()
    if w.shape[0] < CB * 8:      # zero-padded channels of the blocked tensor: zero weight rows
        w = torch.nn.functional.pad(w, (0, 0, 0, CB * 8 - w.shape[0])).contiguous()
    fn = lib().svc_conv_post_hl if is_split(xh) else lib().svc_conv_post_h
    check(fn(_hptr(xh), ptr(w), ptr(bias), ptr(out), B, CB * 8, T, KS, pad, pre_slope,
             ACT_TANH if act is None else act, stream_ptr()), "conv_post_h")
    return out


def coupling_fused_h(view, mask, cond, pre, ins, rss, post, *, reverse, split):
    """One ResidualCouplingLayer (mean_only) on the flow's fp32 working buffer, in place, ONE launch (svc_coupling_fused_h).
    view: the [B, channels, T] buffer or a FlipView of it; mask [B, T] or None; cond = cond_layer(g) [B, 2 H L, 1 | T] or None;
    pre / post: (pack, bias); ins / rss: per-layer lists of (pack, bias) — packs from pack_conv1d_h(dense weight, split=split)."""
    B, Cc, T = view.shape
    a = CouplingArgs()
    a.x = view.data_ptr()
    a.x_bs, a.x_cs = view.stride(0), view.stride(1)
    if view.stride(2) != 1 and T > 1:
        raise SvcError("coupling_fused_h: the buffer must be time-contiguous")
    a.mask = ptr(mask)
    if cond is not None:
        require_gpu(cond)
        a.cond = ptr(cond)
        a.cond_bs, a.cond_cs = cond.stride(0), cond.stride(1)
        a.cond_ts = 0 if cond.shape[2] == 1 else 1
        if a.cond_ts and (cond.stride(2) != 1 or cond.shape[2] != T):
            raise SvcError("coupling_fused_h: a per-frame conditioning tensor must be [B, C, T] with contiguous time")
    L = len(ins)
    if L != len(rss) or not 1 <= L <= COUPLING_MAX_LAYERS:
        raise SvcError("coupling_fused_h: bad layer count")
    want = 5 if split else 4
    for wp, _ in [pre, post] + list(ins) + list(rss):
        if wp.dim() != want or wp.dtype != torch.float16 or not wp.is_cuda:
            raise SvcError("coupling_fused_h: weight packs must come from pack_conv1d_h(..., split=%s)" % bool(split))
    a.w_pre, a.b_pre, a.s_pre = _hptr(pre[0]), ptr(pre[1]), _acc_scale(pre[0])
    a.w_post, a.b_post, a.s_post = _hptr(post[0]), ptr(post[1]), _acc_scale(post[0])
    for l in range(L):
        require_gpu(ins[l][1], rss[l][1])
        a.w_in[l], a.b_in[l], a.s_in[l] = ins[l][0].data_ptr(), ins[l][1].data_ptr(), _acc_scale(ins[l][0])
        a.w_rs[l], a.b_rs[l], a.s_rs[l] = rss[l][0].data_ptr(), rss[l][1].data_ptr(), _acc_scale(rss[l][0])
    a.B, a.T, a.channels, a.hidden = B, T, Cc, ins[0][0].shape[-4] * 16
    a.kernel_size, a.n_layers, a.reverse, a.planes = ins[0][0].shape[-3], L, 1 if reverse else 0, 2 if split else 1
    check(lib().svc_coupling_fused_h(C.byref(a), stream_ptr()), "coupling_fused_h")


def nsf_source(f0, rand_ini, noise, lin_w, lin_b, upp, sampling_rate, sine_amp=0.1, noise_std=0.003, out=None,
               scratch=None):
    """f0 [B,T], rand_ini [B,H], noise [B,T*upp,H], lin_w [1,H]|[H], lin_b [1] -> har_source [B,1,T*upp]."""
    require_gpu(f0, rand_ini, noise, lin_w, lin_b, out)
    B, T = f0.shape
    H = rand_ini.shape[1]
    L = T * upp
    if tuple(noise.shape) != (B, L, H):
        raise SvcError(f"nsf_source: noise shape {tuple(noise.shape)} != {(B, L, H)}")
    if out is None:
        out = torch.empty((B, 1, L), device=f0.device, dtype=torch.float32)
    nbytes = lib().svc_nsf_source_scratch_bytes(B, T, H)
    if scratch is None or scratch.numel() * scratch.element_size() < nbytes:
        scratch = torch.empty((nbytes + 7) // 8, device=f0.device, dtype=torch.float64)
    check(lib().svc_nsf_source_f32(ptr(f0.contiguous()), ptr(rand_ini.contiguous()), ptr(noise.contiguous()),
                                   ptr(lin_w.contiguous()), ptr(lin_b.contiguous()), ptr(out), ptr(scratch), B, T,
                                   upp, H, float(sampling_rate), sine_amp, noise_std, stream_ptr()), "nsf_source")
    return out


def sinusoidal_emb(t, dim):
    """SinusoidalPosEmb (diffusion/wavenet.py:16-28): t [B] -> [B, dim]."""
    require_gpu(t)
    t = t.float().contiguous()
    out = torch.empty((t.shape[0], dim), device=t.device, dtype=torch.float32)
    check(lib().svc_sinusoidal_emb_f32(ptr(t), ptr(out), t.shape[0], dim, stream_ptr()), "sinusoidal_emb")
    return out


def resample_sinc(x, kern, orig, new, width, Lout=None):
    """x [B, Lin] -> [B, Lout] through the [K, new] filter bank `kern` (see include/svc_hip.h); orig/new already reduced."""
    require_gpu(x, kern)
    x = x.contiguous()
    B, Lin = x.shape
    K = kern.shape[0]
    if tuple(kern.shape) != (2 * width + orig, new):
        raise SvcError(f"resample_sinc: kernel bank {tuple(kern.shape)} != {(2 * width + orig, new)}")
    full = -(-Lin * new // orig)
    Lout = full if Lout is None else Lout
    y = torch.empty((B, Lout), device=x.device, dtype=torch.float32)
    check(lib().svc_resample_sinc_f32(ptr(x), ptr(kern.contiguous()), ptr(y), x.stride(0), y.stride(0), B, Lin, Lout, orig, new,
                                      K, width, stream_ptr()), "resample_sinc")
    return y


def nsf_source_exact(f0, rand_ini, noise, lin_w, lin_b, upp, sampling_rate, sine_amp=0.1, noise_std=0.003):
    """vdecoder/nsf_hifigan source module (double-precision phase): same shapes as nsf_source."""
    require_gpu(f0, rand_ini, noise, lin_w, lin_b)
    B, T = f0.shape
    H = rand_ini.shape[1]
    L = T * upp
    if tuple(noise.shape) != (B, L, H):
        raise SvcError(f"nsf_source_exact: noise shape {tuple(noise.shape)} != {(B, L, H)}")
    out = torch.empty((B, 1, L), device=f0.device, dtype=torch.float32)
    scratch = torch.empty(B * H * T, device=f0.device, dtype=torch.float64)
    check(lib().svc_nsf_source_exact_f32(ptr(f0.contiguous()), ptr(rand_ini.contiguous()), ptr(noise.contiguous()),
                                         ptr(lin_w.contiguous()), ptr(lin_b.contiguous()), ptr(out), ptr(scratch), B, T,
                                         upp, H, float(sampling_rate), sine_amp, noise_std, stream_ptr()), "nsf_source_exact")
    return out


def f0_to_coarse(f0):
    require_gpu(f0)
    f0c = f0.contiguous()
    out = torch.empty(f0c.shape, device=f0.device, dtype=torch.int64)
    check(lib().svc_f0_to_coarse(ptr(f0c), ptr(out), f0c.numel(), stream_ptr()), "f0_to_coarse")
    return out


def prenet_embed(xin, uv, f0, emb_uv, f0_emb, mask=None, vol=None, vol_w=None, vol_b=None):
    """Returns (x, x_enc): x = xin + emb_uv[uv] (+ vol); x_enc = (x + f0_emb[coarse(f0)]) * mask."""
    require_gpu(xin, uv, f0, emb_uv, f0_emb, mask, vol, vol_w, vol_b)
    B, Cc, T = xin.shape
    xin = xin.contiguous()
    x = torch.empty_like(xin)
    x_enc = torch.empty_like(xin)
    check(lib().svc_prenet_embed_f32(ptr(xin), ptr(uv.contiguous()), ptr(f0.contiguous()), ptr(emb_uv.contiguous()),
                                     ptr(f0_emb.contiguous()), ptr(mask), ptr(vol), ptr(vol_w), ptr(vol_b), ptr(x),
                                     ptr(x_enc), B, Cc, T, stream_ptr()), "prenet_embed")
    return x, x_enc


def add_layernorm(x, r, gamma, beta, mask=None, eps=1e-5, out=None):
    require_gpu(x, r, gamma, beta, mask, out)
    B, Cc, T = x.shape
    if not x.is_contiguous() or (r is not None and not r.is_contiguous()):
        raise SvcError("add_layernorm needs contiguous [B,C,T] tensors")
    if out is None:
        out = torch.empty_like(x)
    check(lib().svc_add_layernorm_f32(ptr(x), ptr(r), ptr(gamma), ptr(beta), ptr(mask), ptr(out), B, Cc, T, eps,
                                      stream_ptr()), "add_layernorm")
    return out


def reparam(stats, noise, mask=None, scale=1.0, out=None):
    require_gpu(stats, noise, mask, out)
    B, C2, T = stats.shape
    Cc = C2 // 2
    if not stats.is_contiguous() or not noise.is_contiguous():
        raise SvcError("reparam needs contiguous tensors")
    if out is None:
        out = torch.empty((B, Cc, T), device=stats.device, dtype=torch.float32)
    check(lib().svc_reparam_f32(ptr(stats), ptr(noise), ptr(mask), ptr(out), B, Cc, T, scale, stream_ptr()),
          "reparam")
    return out


def attention(q, k, v, n_heads, *, emb_rel_k=None, emb_rel_v=None, window=0, mask=None, mask_mode=0, out=None):
    """q,k,v: [B, H*dk, T] views (time contiguous).  Returns [B, H*dk, T]."""
    require_gpu(q, k, v, emb_rel_k, emb_rel_v, mask, out)
    B, Cc, T = q.shape
    dk = Cc // n_heads
    if out is None:
        out = torch.empty((B, Cc, T), device=q.device, dtype=torch.float32)
    a = AttentionArgs()
    a.q, a.k, a.v, a.out = ptr(q), ptr(k), ptr(v), ptr(out)
    a.emb_rel_k, a.emb_rel_v, a.mask = ptr(emb_rel_k), ptr(emb_rel_v), ptr(mask)
    a.q_bs, a.q_cs = _bct_strides(q)
    a.k_bs, a.k_cs = _bct_strides(k)
    a.v_bs, a.v_cs = _bct_strides(v)
    a.o_bs, a.o_cs = _bct_strides(out)
    if mask is not None:
        a.mask_bs = mask.stride(0)
    a.B, a.H, a.dk, a.T, a.window, a.mask_mode = B, n_heads, dk, T, window, mask_mode
    nws = lib().svc_attention_ws_bytes(C.byref(a))       # > 0: a short sequence whose keys are split over workgroups too
    ws = None
    if nws > 0:
        ws = torch.empty(nws // 4, device=q.device, dtype=torch.float32)
        a.ws, a.ws_bytes = ptr(ws), nws
    check(lib().svc_attention_f32(C.byref(a), stream_ptr()), "attention")
    return out


def f0_norm_lf0(f0, uv, mask=None, factor=None, input_is_lf0=False):
    """f0, uv: [B,T]; mask [B,1,T]|[B,T]; factor [B]|[B,1] -> (lf0 [B,1,T], norm_lf0 [B,1,T])."""
    require_gpu(f0, uv, mask, factor)
    B, T = f0.shape
    lf0 = torch.empty((B, 1, T), device=f0.device, dtype=torch.float32)
    norm = torch.empty((B, 1, T), device=f0.device, dtype=torch.float32)
    check(lib().svc_f0_norm_lf0_f32(ptr(f0.contiguous()), ptr(uv.contiguous()),
                                    ptr(mask.contiguous() if mask is not None else None),
                                    ptr(factor.contiguous() if factor is not None else None), ptr(lf0), ptr(norm), B, T,
                                    1 if input_is_lf0 else 0, stream_ptr()), "f0_norm_lf0")
    return lf0, norm


def lf0_to_f0(lf0):
    require_gpu(lf0)
    x = lf0.contiguous()
    out = torch.empty_like(x)
    check(lib().svc_lf0_to_f0_f32(ptr(x), ptr(out), x.numel(), stream_ptr()), "lf0_to_f0")
    return out


def copy_bct(x, out=None, mask=None):
    """out[b,c,t] = x[b,c,t] * mask[b,t]; x/out may be FlipViews or channel slices."""
    require_gpu(x, out, mask)
    B, Cc, T = x.shape
    if out is None:
        out = torch.empty((B, Cc, T), device=x.device, dtype=torch.float32)
    xb, xc = _bct_strides(x)
    yb, yc = _bct_strides(out)
    check(lib().svc_copy_bct_f32(ptr(x), ptr(out), ptr(mask), xb, xc, yb, yc, mask.stride(0) if mask is not None else 0,
                                 B, Cc, T, stream_ptr()), "copy_bct")
    return out


def snake_alias(x, alpha, beta, taps, out=None):
    """y = DownSample1d(SnakeBeta(UpSample1d(x))) (vdecoder/hifiganwithsnake/alias/act.py:125-130), one kernel.
    taps: sequence of the 12 filter taps (host floats)."""
    require_gpu(x, alpha, beta, out)
    B, Cc, T = x.shape
    if out is None:
        out = torch.empty((B, Cc, T), device=x.device, dtype=torch.float32)
    xb, xc = _bct_strides(x)
    yb, yc = _bct_strides(out)
    tp = (C.c_float * 12)(*[float(v) for v in taps])
    check(lib().svc_snake_alias_f32(ptr(x), ptr(out), ptr(alpha), ptr(beta), tp, xb, xc, yb, yc, B, Cc, T, stream_ptr()),
          "snake_alias")
    return out


def channel_norm_gelu(x, gamma, beta, eps=1e-5, gelu=True, lengths=None):
    """GroupNorm(C, C) over time + GELU (vencoder/hubert/hubert_model.py:76,87).  `lengths` (int32 [B] on the device): items of
    different lengths zero-padded to T — statistics over each item's own length, zeros beyond it."""
    require_gpu(x, gamma, beta)
    x = x.contiguous()
    B, Cc, T = x.shape
    y = torch.empty_like(x)
    if lengths is not None:
        if lengths.dtype != torch.int32 or not lengths.is_cuda or lengths.numel() != B:
            raise SvcError("channel_norm_gelu: lengths must be an int32 [B] device tensor")
        check(lib().svc_channel_norm_gelu_len_f32(ptr(x), ptr(gamma), ptr(beta), C.c_void_p(lengths.data_ptr()), ptr(y), B, Cc, T, eps,
                                                  1 if gelu else 0, stream_ptr()), "channel_norm_gelu_len")
        return y
    check(lib().svc_channel_norm_gelu_f32(ptr(x), ptr(gamma), ptr(beta), ptr(y), B, Cc, T, eps, 1 if gelu else 0, stream_ptr()),
          "channel_norm_gelu")
    return y


def posconv_pack(v, g=None, groups=16):
    """HuBERT positional-conv weight v [C, C/groups, KS] (+ weight_norm dim=2 gain g [KS]) -> the MFMA kernel's packed layout."""
    require_gpu(v, g)
    v = v.contiguous().float()
    Cc, cg, KS = v.shape
    dst = torch.empty((groups, cg, KS, 64), device=v.device, dtype=torch.float32)
    gg = g.reshape(-1).contiguous().float() if g is not None else None
    check(lib().svc_posconv_pack_f32(ptr(v), ptr(gg), ptr(dst), Cc, KS, groups, stream_ptr()), "posconv_pack")
    return dst


def posconv(x, wpacked, bias, pad=64):
    """y = x + gelu(grouped_conv1d(x, w, bias, padding=pad)[..., :T]) (vencoder/hubert/hubert_model.py:116-129), one MFMA kernel."""
    require_gpu(x, wpacked, bias)
    x = x.contiguous()
    B, Cc, T = x.shape
    G, cg, KS, _ = wpacked.shape
    y = torch.empty_like(x)
    check(lib().svc_posconv_f32(ptr(x), ptr(wpacked), ptr(bias), ptr(y), B, Cc, T, KS, pad, G, stream_ptr()), "posconv")
    return y


def snake_alias_bwd(x, dy, alpha, beta, taps):
    """(dx, dalpha, dbeta) of snake_alias."""
    require_gpu(x, dy, alpha, beta)
    B, Cc, T = x.shape
    dx = torch.empty((B, Cc, T), device=x.device, dtype=torch.float32)
    da = torch.empty(Cc, device=x.device, dtype=torch.float32)
    db = torch.empty(Cc, device=x.device, dtype=torch.float32)
    xb, xc = _bct_strides(x)
    gb, gc = _bct_strides(dy)
    tp = (C.c_float * 12)(*[float(v) for v in taps])
    check(lib().svc_snake_alias_bwd_f32(ptr(x), ptr(dy), ptr(alpha), ptr(beta), tp, ptr(dx), ptr(da), ptr(db), xb, xc, gb, gc,
                                        Cc * T, T, B, Cc, T, stream_ptr()), "snake_alias_bwd")
    return dx, da, db


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


# --------------------------------------------------------------------------------------------------------------
# training-path entry points (include/svc_hip.h, "TRAINING path")
# --------------------------------------------------------------------------------------------------------------
(EW_ADD, EW_MUL, EW_LRELU, EW_LRELU_BWD, EW_TANH, EW_TANH_BWD, EW_RELU, EW_RELU_BWD, EW_EXP, EW_LOG_CLAMP,
 EW_LOG_CLAMP_BWD, EW_SCALE, EW_SIGMOID, EW_SQUARE, EW_SIGN_MUL, EW_DIV, EW_GELU, EW_MISH, EW_CLAMP, EW_MISH_BWD,
 EW_DROPOUT) = range(21)
RED_SUM, RED_ABS_DIFF, RED_SQ_DIFF, RED_SQ_ONE_MINUS, RED_SQ, RED_KL = range(6)


class WgradArgs(C.Structure):
    _fields_ = [("A", _f32p), ("Bm", _f32p), ("G", _f32p),
                ("a_bs", C.c_longlong), ("a_cs", C.c_longlong), ("b_bs", C.c_longlong), ("b_cs", C.c_longlong),
                ("B", C.c_int), ("Ca", C.c_int), ("Cb", C.c_int), ("TA", C.c_int), ("TB", C.c_int), ("KS", C.c_int),
                ("dil", C.c_int), ("pad", C.c_int), ("accumulate", C.c_int), ("dbias", _f32p), ("mma", C.c_int)]


class GemmArgs(C.Structure):
    _fields_ = [("A", _f32p), ("B", _f32p), ("C", _f32p),
                ("a_bs", C.c_longlong), ("a_ms", C.c_longlong), ("a_ks", C.c_longlong),
                ("b_bs", C.c_longlong), ("b_ks", C.c_longlong), ("b_ns", C.c_longlong),
                ("c_bs", C.c_longlong), ("c_ms", C.c_longlong), ("c_ns", C.c_longlong),
                ("batch", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("alpha", C.c_float), ("beta", C.c_float), ("split_k_atomic", C.c_int)]


class ConvWeightArgs(C.Structure):
    _fields_ = [("v", _f32p), ("g", _f32p), ("wp", _f32p), ("wt", _f32p), ("norm", _f32p)] + \
               [(n, C.c_int) for n in ("kind", "R", "C2", "K", "Od", "Id", "Kd", "OdP", "IdP", "s", "shift")]


TRAIN_EXPORTS = [
    "svc_conv_weight_prep_f32", "svc_conv_weight_prep_multi_f32", "svc_conv_weight_prep_blocks", "svc_conv_weight_grad_f32",
    "svc_weight_norm_fwd_f32", "svc_weight_norm_bwd_f32", "svc_pack_conv1d_weight_T", "svc_conv1d_wgrad_f32",
    "svc_gemm_f32", "svc_reduce_bct_f32", "svc_reduce_c_f32", "svc_ew_f32", "svc_ew_bct_f32", "svc_gate_fwd_f32",
    "svc_gate_bwd_f32", "svc_decimate_f32", "svc_decimate_bwd_f32", "svc_gconv1d_fwd_f32", "svc_gconv1d_dgrad_f32",
    "svc_gconv1d_wgrad_f32", "svc_reduce_scalar_f64", "svc_f64_to_f32", "svc_adamw_f32", "svc_adamw_advance", "svc_debug_set_conv_cfg",
    "svc_debug_set_wgrad_target", "svc_debug_set_conv_strip", "svc_debug_set_gconv_version", "svc_nonfinite_guard_f32",
    "svc_lrelu_bwd_add_f32",
]
EXPORTS += TRAIN_EXPORTS
_train_bound = False


def tlib():
    """lib() with the training entry points' argtypes declared."""
    global _train_bound
    L = lib()
    if not _train_bound:
        i, f, ll, vp = C.c_int, C.c_float, C.c_longlong, C.c_void_p
        L.svc_weight_norm_fwd_f32.argtypes = [_f32p] * 4 + [i, i, vp]
        L.svc_weight_norm_bwd_f32.argtypes = [_f32p] * 6 + [i, i, vp]
        L.svc_pack_conv1d_weight_T.argtypes = [_f32p, _f32p, i, i, i, i, vp]
        L.svc_conv1d_wgrad_f32.argtypes = [C.POINTER(WgradArgs), vp]
        L.svc_conv_weight_prep_f32.argtypes = [C.POINTER(ConvWeightArgs), vp]
        L.svc_conv_weight_grad_f32.argtypes = [C.POINTER(ConvWeightArgs), _f32p, _f32p, _f32p, vp]
        L.svc_conv_weight_prep_multi_f32.argtypes = [C.POINTER(ConvWeightArgs), vp, vp, vp, i, vp]
        L.svc_conv_weight_prep_blocks.argtypes = [i, i, i]
        L.svc_gemm_f32.argtypes = [C.POINTER(GemmArgs), vp]
        L.svc_nonfinite_guard_f32.argtypes = [vp, i, vp, vp]
        L.svc_reduce_bct_f32.argtypes = [_f32p, _f32p, ll, ll, i, i, i, i, f, vp]
        L.svc_reduce_c_f32.argtypes = [_f32p, _f32p, _f32p, i, i, i, vp]
        L.svc_ew_f32.argtypes = [i, _f32p, _f32p, _f32p, ll, f, f, vp]
        L.svc_lrelu_bwd_add_f32.argtypes = [_f32p, _f32p, _f32p, _f32p, ll, f, vp]
        L.svc_ew_bct_f32.argtypes = [i, _f32p, _f32p, _f32p] + [ll] * 7 + [i, i, i, f, f, vp]
        L.svc_gate_fwd_f32.argtypes = [_f32p, _f32p, i, i, i, vp]
        L.svc_gate_bwd_f32.argtypes = [_f32p, _f32p, _f32p, i, i, i, vp]
        L.svc_decimate_f32.argtypes = [_f32p, _f32p, i, i, i, i, i, i, i, i, vp]
        L.svc_decimate_bwd_f32.argtypes = [_f32p, _f32p, i, i, i, i, i, i, i, i, vp]
        L.svc_gconv1d_fwd_f32.argtypes = [_f32p] * 4 + [i] * 9 + [vp]
        L.svc_gconv1d_dgrad_f32.argtypes = [_f32p] * 3 + [i] * 9 + [vp]
        L.svc_gconv1d_wgrad_f32.argtypes = [_f32p] * 3 + [i] * 9 + [vp]
        L.svc_reduce_scalar_f64.argtypes = [i, _f32p, _f32p, _f32p, _f32p, ll, vp, C.c_double, vp]
        L.svc_f64_to_f32.argtypes = [vp, _f32p, i, vp]
        L.svc_adamw_f32.argtypes = [_f32p] * 4 + [ll, _f32p, vp]
        L.svc_adamw_advance.argtypes = [_f32p, vp]
        L.svc_debug_set_conv_cfg.argtypes = [i]
        L.svc_debug_set_wgrad_target.argtypes = [i]
        L.svc_debug_set_gconv_version.argtypes = [i]
        if os.environ.get("SVC_GCONV_VERSION"):
            L.svc_debug_set_gconv_version(int(os.environ["SVC_GCONV_VERSION"]))
        if os.environ.get("SVC_WGRAD_TARGET"):       # tuning aid (A/B on one box)
            L.svc_debug_set_wgrad_target(int(os.environ["SVC_WGRAD_TARGET"]))
        if os.environ.get("SVC_WGRAD_SMALL_TARGET"):
            L.svc_debug_set_wgrad_target(-int(os.environ["SVC_WGRAD_SMALL_TARGET"]))
        _train_bound = True
    return L


def ew(op, a, b=None, alpha=1.0, beta=0.0, out=None):
    require_gpu(a, b, out)
    a = a.contiguous()
    if b is not None:
        b = b.contiguous()
        if b.shape != a.shape:
            raise SvcError(f"ew: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
    if out is None:
        out = torch.empty_like(a)
    if a.numel():
        check(tlib().svc_ew_f32(op, ptr(a), ptr(b), ptr(out), a.numel(), alpha, beta, stream_ptr()), "ew")
    return out


def lrelu_bwd_add(dy, x, r, slope):
    """leaky_relu'(x) * dy + r in one launch (all three contiguous, same shape)."""
    require_gpu(dy, x, r)
    dy, x, r = dy.contiguous(), x.contiguous(), r.contiguous()
    out = torch.empty_like(x)
    check(tlib().svc_lrelu_bwd_add_f32(ptr(dy), ptr(x), ptr(r), ptr(out), x.numel(), float(slope), stream_ptr()), "lrelu_bwd_add")
    return out


def ew_bct(op, x, side, alpha=1.0, beta=0.0, out=None):
    """out[b,c,t] = op(x[b,c,t], side broadcast); side is [B|1, C|1, T|1]."""
    require_gpu(x, side, out)
    B, Cc, T = x.shape
    if out is None:
        out = torch.empty((B, Cc, T), device=x.device, dtype=torch.float32)
    sb = side.stride(0) if side.shape[0] > 1 else 0
    sc = side.stride(1) if side.shape[1] > 1 else 0
    st = side.stride(2) if side.shape[2] > 1 else 0
    xb, xc = _bct_strides(x)
    yb, yc = _bct_strides(out)
    check(tlib().svc_ew_bct_f32(op, ptr(x), ptr(side), ptr(out), xb, xc, sb, sc, st, yb, yc, B, Cc, T, alpha, beta,
                                stream_ptr()), "ew_bct")
    return out


def reduce_bct(x, mode, out=None, beta=0.0):
    """mode 0: [C] = sum over (b,t);  mode 1: [B,C,1] = sum over t."""
    require_gpu(x, out)
    B, Cc, T = x.shape
    xb, xc = _bct_strides(x)
    if out is None:
        out = torch.empty((Cc,) if mode == 0 else (B, Cc, 1), device=x.device, dtype=torch.float32)
    check(tlib().svc_reduce_bct_f32(ptr(x), ptr(out), xb, xc, B, Cc, T, mode, beta, stream_ptr()), "reduce_bct")
    return out


def reduce_c(x, w=None):
    require_gpu(x, w)
    x = x.contiguous()
    B, Cc, T = x.shape
    out = torch.empty((B, 1, T), device=x.device, dtype=torch.float32)
    check(tlib().svc_reduce_c_f32(ptr(x), ptr(w), ptr(out), B, Cc, T, stream_ptr()), "reduce_c")
    return out


def weight_norm_fwd(v, g):
    require_gpu(v, g)
    v = v.contiguous()
    rows = v.shape[0]
    cols = v.numel() // rows
    w = torch.empty_like(v)
    norm = torch.empty(rows, device=v.device, dtype=torch.float32)
    check(tlib().svc_weight_norm_fwd_f32(ptr(v), ptr(g.contiguous()), ptr(w), ptr(norm), rows, cols, stream_ptr()),
          "weight_norm_fwd")
    return w, norm


def weight_norm_bwd(v, g, norm, dw):
    require_gpu(v, g, norm, dw)
    v = v.contiguous()
    rows = v.shape[0]
    cols = v.numel() // rows
    dv = torch.empty_like(v)
    dg = torch.empty_like(g)
    check(tlib().svc_weight_norm_bwd_f32(ptr(v), ptr(g.contiguous()), ptr(norm), ptr(dw.contiguous()), ptr(dv), ptr(dg),
                                         rows, cols, stream_ptr()), "weight_norm_bwd")
    return dv, dg


def pack_conv1d_weight_T(w):
    """w [Cout,Cin,KS] -> dgrad packing [Cout, KS, CinP]."""
    require_gpu(w)
    w = w.contiguous()
    Cout, Cin, KS = w.shape
    CinP = round_up(Cin, 32)
    dst = torch.empty((Cout, KS, CinP), device=w.device, dtype=torch.float32)
    check(tlib().svc_pack_conv1d_weight_T(ptr(w), ptr(dst), Cout, Cin, KS, CinP, stream_ptr()), "pack_conv1d_T")
    return dst


class ConvWeightPlan:
    """The weight side of one training convolution: the index map parameter -> dense-conv weight (svc_conv_weight_args in
    include/svc_hip.h) and the two persistent operand buffers it fills.  `prepare(v, g)` is ONE launch producing the forward
    operand, the dgrad operand and the weight-norm row norms; `grad(v, g, dwd)` ONE launch mapping the dense-layout weight
    gradient back to (dv, dg).  The buffers are allocated zero-filled on first use and only their mapped entries are ever
    rewritten, so they are valid hipGraph operands (fixed addresses) and the padding stays zero."""

    DENSE, STRIDED, TRANSPOSED = 0, 1, 2

    def __init__(self, kind, R, C2, K, s=1, shift=0, Kd=None):
        self.kind, self.R, self.C2, self.K, self.s, self.shift = kind, R, C2, K, s, shift
        if kind == self.DENSE:
            self.Od, self.Id, self.Kd = R, C2, K
        elif kind == self.STRIDED:
            self.Od, self.Id, self.Kd = R, s * C2, Kd
        else:
            self.Od, self.Id, self.Kd = s * C2, R, (K + s - 1) // s
        self.OdP, self.IdP = round_up(self.Od, 32), round_up(self.Id, 32)
        self.wp = self.wt = self.norm = None

    def _args(self, v, g):
        a = ConvWeightArgs()
        a.v, a.g, a.wp, a.wt, a.norm = ptr(v), ptr(g), ptr(self.wp), ptr(self.wt), ptr(self.norm if g is not None else None)
        a.kind, a.R, a.C2, a.K, a.s, a.shift = self.kind, self.R, self.C2, self.K, self.s, self.shift
        a.Od, a.Id, a.Kd, a.OdP, a.IdP = self.Od, self.Id, self.Kd, self.OdP, self.IdP
        return a

    def _check(self, v, g):
        require_gpu(v, g)
        if v.numel() != self.R * self.C2 * self.K or not v.is_contiguous():
            raise SvcError(f"ConvWeightPlan: parameter {tuple(v.shape)} does not match the plan [{self.R},{self.C2},{self.K}]")
        if g is not None and (g.numel() != self.R or not g.is_contiguous()):
            raise SvcError("ConvWeightPlan: weight_g must hold one contiguous value per row")

    def _buffers(self):
        """Addresses of the persistent operand buffers (what a PlanSets device table bakes in)."""
        return None if self.wp is None else (self.wp.data_ptr(), self.wt.data_ptr(), self.norm.data_ptr())

    def _alloc(self, device):
        if self.wp is None or self.wp.device != device:
            self.__dict__.pop("_fresh", None)     # new buffers are unfilled: no open bracket's token may vouch for them
            self.wp = torch.zeros((self.Id, self.Kd, self.OdP), device=device, dtype=torch.float32)
            self.wt = torch.zeros((self.Od, self.Kd, self.IdP), device=device, dtype=torch.float32)
            self.norm = torch.empty((self.R,), device=device, dtype=torch.float32)

    def prepare(self, v, g=None):
        """-> (wp [Id,Kd,OdP], wt [Od,Kd,IdP]) for v (and g) as they are now."""
        self._check(v, g)
        fresh = self.__dict__.get("_fresh")
        if fresh is not None and fresh == (v.data_ptr(), g.data_ptr() if g is not None else 0) and self.wp is not None:
            return self.wp, self.wt      # filled by the open PlanSets bracket's one launch from exactly these tensors
        self._alloc(v.device)
        check(tlib().svc_conv_weight_prep_f32(C.byref(self._args(v, g)), stream_ptr()), "conv_weight_prep")
        if PlanSets.recording is not None:
            PlanSets.recording.note(self, v, g)
        return self.wp, self.wt

    def grad(self, v, g, dwd):
        """dwd [Od,Id,Kd] -> (dv like v, dg like g or None); uses the row norms of the last prepare()."""
        self._check(v, g)
        require_gpu(dwd)
        if tuple(dwd.shape) != (self.Od, self.Id, self.Kd) or not dwd.is_contiguous():
            raise SvcError(f"ConvWeightPlan.grad: dwd {tuple(dwd.shape)} is not contiguous [{self.Od},{self.Id},{self.Kd}]")
        if g is None and self.kind == self.DENSE:
            return dwd.view(v.shape), None
        dv = torch.empty_like(v)
        dg = torch.empty_like(g) if g is not None else None
        check(tlib().svc_conv_weight_grad_f32(C.byref(self._args(v, g)), ptr(dwd), ptr(dv), ptr(dg), stream_ptr()),
              "conv_weight_grad")
        return dv, dg


class PlanSets:
    """Weight preparation for a whole group of ConvWeightPlans in two launches — row norms, operand packs — instead of one or
    two per plan (svc_conv_weight_prep_multi_f32).

    A training loop brackets a forward pass whose convolution weights are parameters with `enter(tag, params)` /
    `leave(tag)`.  The first bracketed pass records which plans were prepared from which parameters; from then on
    `enter` prepares them all together and hands each plan a token naming the parameter storage it was prepared from,
    so the pass's own `plan.prepare(v, g)` calls return the already filled operands (any other tensor prepares as usual).
    `leave` withdraws the tokens: operands are never reused outside the bracket, i.e. across an optimizer step — the caller
    must not update the parameters inside one.
    Plans fed from computed tensors (spectral-norm weights) are not parameters' storage and stay on their own launch."""

    recording = None      # the instance whose bracket is open and recording (one at a time)

    def __init__(self):
        self.sets, self.rec, self.ptrs, self.enabled = {}, None, None, os.environ.get("SVC_PLAN_SETS", "1") != "0"
        self._retired = []   # tables of replaced sets: a captured graph may still point at them

    def note(self, plan, v, g):
        if self.rec is None:
            return
        if v.data_ptr() in self.ptrs and (g is None or g.data_ptr() in self.ptrs) and all(e[0] is not plan for e in self.rec):
            self.rec.append((plan, v, g))

    def enter(self, tag, params):
        if not self.enabled:
            return
        ent = self.sets.get(tag)
        live = {p.data_ptr() for p in params} if ent is not None else None
        # still valid: the recorded tensors sit where they were AND are storage of the parameters handed in now (a module that
        # replaced its Parameter objects would otherwise be prepared from the old ones — harmless, the tokens would not match,
        # but wasted)
        # ... and every plan still owns the operand buffers whose addresses the device table holds (a plan re-allocates them on
        # a device change; the multi kernel would scatter into the freed ones)
        if ent is not None and all(v.data_ptr() == pv and (g.data_ptr() if g is not None else 0) == pg and pl.wp is not None
                                   and pv in live and (pg == 0 or pg in live) and pl._buffers() == bufs
                                   for (pl, v, g), (pv, pg), bufs in zip(ent["items"], ent["ptrs"], ent["bufs"])):
            check(tlib().svc_conv_weight_prep_multi_f32(ent["host"], ent["dev"].data_ptr(), ent["rows"][0].data_ptr(),
                                                        ent["rows"][1].data_ptr(), len(ent["items"]), stream_ptr()),
                  "conv_weight_prep_multi")
            for (pl, v, g), key in zip(ent["items"], ent["ptrs"]):
                pl._fresh = key
            return
        if torch.cuda.is_current_stream_capturing():
            return               # building a table copies host memory: not inside a capture (this pass prepares plan by plan)
        old = self.sets.pop(tag, None)
        if old is not None:
            self._retired.append((old["dev"], old["rows"], old["host"]))
        self.rec, self.ptrs = [], (live if live is not None else {p.data_ptr() for p in params})
        PlanSets.recording = self

    def leave(self, tag):
        if not self.enabled:
            return
        if self.rec is not None:
            items, self.rec, self.ptrs = self.rec, None, None
            PlanSets.recording = None
            if items:
                host = (ConvWeightArgs * len(items))(*[pl._args(v, g) for pl, v, g in items])
                dev = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(items[0][1].device)
                rstart, bstart, ra, ba = [], [], 0, 0
                for pl, _, _ in items:       # prefix sums of rows (norm launch) and of scatter workgroups (svc_hip.h)
                    rstart.append(ra)
                    bstart.append(ba)
                    ra += pl.R
                    ba += tlib().svc_conv_weight_prep_blocks(pl.R, pl.C2, pl.K)
                rows = torch.tensor([rstart, bstart], dtype=torch.int32).to(dev.device)
                self.sets[tag] = dict(items=items, host=host, dev=dev, rows=rows,
                                      ptrs=[(v.data_ptr(), g.data_ptr() if g is not None else 0) for _, v, g in items],
                                      bufs=[pl._buffers() for pl, _, _ in items])
            return
        ent = self.sets.get(tag)
        if ent is not None:
            for pl, _, _ in ent["items"]:
                pl.__dict__.pop("_fresh", None)



class ZeroSlab:
    """Pre-zeroed storage for the atomically accumulated outputs of one training iteration (weight / bias gradients of
    svc_conv1d_wgrad_f32): `take` hands out zero-filled views, `reset` re-zeroes what was handed out with ONE memset per
    chunk.  Replaces one hipMemsetAsync per gradient tensor (585 per iteration of the B=16 GAN step, 2.9 ms of 5 us
    launches, profiles/r01_k_train_aten_op_census.txt).  Off unless a training loop turns it on (train.TrainStep): a view
    stays valid only until the next reset, i.e. until the optimizer has consumed the gradients of the iteration."""

    CHUNK = 96 * 1024 * 1024     # floats per chunk (384 MB: the 99 M parameters of G + D fit in two)

    def __init__(self):
        self.chunks, self.used, self.cur, self.active = [], [], 0, False

    def take(self, shape, device):
        n = 1
        for d in shape:
            n *= int(d)
        need = (n + 63) // 64 * 64
        while True:
            if self.cur < len(self.chunks):
                c = self.chunks[self.cur]
                if c.device == torch.device(device) and self.used[self.cur] + need <= c.numel():
                    o = self.used[self.cur]
                    self.used[self.cur] = o + need
                    return c[o:o + n].view(*shape)
                self.cur += 1
                continue
            self.chunks.append(torch.zeros(max(need, self.CHUNK), device=device, dtype=torch.float32))
            self.used.append(0)

    def reset(self):
        for c, u in zip(self.chunks, self.used):
            if u:
                c[:u].zero_()
        self.used = [0] * len(self.chunks)
        self.cur = 0


wgrad_slab = ZeroSlab()


_SLAB_ON = os.environ.get("SVC_WGRAD_SLAB", "1") != "0"


def wgrad_zeros(shape, device):
    """Zero-initialised gradient buffer: a slab view when the slab is active (then no per-tensor memset is needed)."""
    if wgrad_slab.active and _SLAB_ON:
        return wgrad_slab.take(shape, device), True
    return torch.empty(tuple(shape), device=device, dtype=torch.float32), False


def conv1d_wgrad(A, Bm, KS, dil, pad, out=None, accumulate=False, dbias=None, mma=None):
    """G[ca,cb,k] = sum_{b,t} A[b,ca,t] * Bm[b,cb,t + k*dil - pad]; dbias (optional [Ca] buffer) also receives
    sum_{b,t} A[b,ca,t].  Without `out` the result lands in the pre-zeroed gradient slab when that is active; a dbias
    passed together with a slab-backed output must itself be zero-initialised (wgrad_zeros)."""
    require_gpu(A, Bm, out, dbias)
    B, Ca, TA = A.shape
    _, Cb, TB = Bm.shape
    if out is None:
        out, zeroed = wgrad_zeros((Ca, Cb, KS), A.device)
        accumulate = accumulate or zeroed
    a = WgradArgs()
    a.A, a.Bm, a.G = ptr(A), ptr(Bm), ptr(out)
    a.a_bs, a.a_cs = _bct_strides(A)
    a.b_bs, a.b_cs = _bct_strides(Bm)
    a.B, a.Ca, a.Cb, a.TA, a.TB, a.KS, a.dil, a.pad, a.accumulate = B, Ca, Cb, TA, TB, KS, dil, pad, 1 if accumulate else 0
    a.dbias = ptr(dbias)
    a.mma = _MMA if mma is None else mma
    check(tlib().svc_conv1d_wgrad_f32(C.byref(a), stream_ptr()), "conv1d_wgrad")
    return out


def gemm(A, Bmat, a_strides, b_strides, batch, M, N, K, out=None, c_strides=None, alpha=1.0, beta=0.0, split_k_atomic=False):
    """C[b,m,n] = alpha*sum_k A[b,m,k]*B[b,k,n] + beta*C.  a_strides = (bs, ms, ks), b_strides = (bs, ks, ns),
    c_strides = (bs, ms, ns) (default contiguous [batch,M,N]).  A/Bmat/out are tensors used as base pointers.
    split_k_atomic: allow the thin-M split-reduction kernel (fp32 atomics: not bit-reproducible) — gradient products only."""
    require_gpu(A, Bmat, out)
    if out is None:
        out = torch.empty((batch, M, N), device=A.device, dtype=torch.float32)
    if c_strides is None:
        c_strides = (M * N, N, 1)
    a = GemmArgs()
    a.A, a.B, a.C = ptr(A), ptr(Bmat), ptr(out)
    a.a_bs, a.a_ms, a.a_ks = a_strides
    a.b_bs, a.b_ks, a.b_ns = b_strides
    a.c_bs, a.c_ms, a.c_ns = c_strides
    a.batch, a.M, a.N, a.K, a.alpha, a.beta = batch, M, N, K, alpha, beta
    a.split_k_atomic = 1 if split_k_atomic else 0
    check(tlib().svc_gemm_f32(C.byref(a), stream_ptr()), "gemm")
    return out


def gate_fwd(x):
    require_gpu(x)
    x = x.contiguous()
    B, C2, T = x.shape
    out = torch.empty((B, C2 // 2, T), device=x.device, dtype=torch.float32)
    check(tlib().svc_gate_fwd_f32(ptr(x), ptr(out), B, C2 // 2, T, stream_ptr()), "gate_fwd")
    return out


def gate_bwd(x, dacts):
    require_gpu(x, dacts)
    x = x.contiguous()
    B, C2, T = x.shape
    din = torch.empty_like(x)
    check(tlib().svc_gate_bwd_f32(ptr(x), ptr(dacts.contiguous()), ptr(din), B, C2 // 2, T, stream_ptr()), "gate_bwd")
    return din


def decimate(x, s, off, Q, lp=None, inner=1):
    """y[b, r*C + c, q*inner + j] = xpad[b, c, (q*s + r + off)*inner + j] -> [B, s*C, Q*inner]."""
    require_gpu(x)
    x = x.contiguous()
    B, Cc, T = x.shape
    y = torch.empty((B, s * Cc, Q * inner), device=x.device, dtype=torch.float32)
    check(tlib().svc_decimate_f32(ptr(x), ptr(y), B, Cc, T, s, inner, off, Q, T if lp is None else lp, stream_ptr()),
          "decimate")
    return y


def decimate_bwd(dy, Cc, T, s, off, lp=None, inner=1):
    require_gpu(dy)
    dy = dy.contiguous()
    B, sC, QW = dy.shape
    dx = torch.empty((B, Cc, T), device=dy.device, dtype=torch.float32)
    check(tlib().svc_decimate_bwd_f32(ptr(dy), ptr(dx), B, Cc, T, s, inner, off, QW // inner, T if lp is None else lp,
                                      stream_ptr()), "decimate_bwd")
    return dx


def gconv1d_fwd(x, w, bias, stride, pad, groups):
    require_gpu(x, w, bias)
    x, w = x.contiguous(), w.contiguous()
    B, Cin, Tin = x.shape
    Cout, Cg, KS = w.shape
    Tout = (Tin + 2 * pad - KS) // stride + 1
    y = torch.empty((B, Cout, Tout), device=x.device, dtype=torch.float32)
    check(tlib().svc_gconv1d_fwd_f32(ptr(x), ptr(w), ptr(bias), ptr(y), B, Cin, Cout, Tin, Tout, KS, stride, pad, groups,
                                     stream_ptr()), "gconv1d_fwd")
    return y


def gconv1d_dgrad(dy, w, Cin, Tin, stride, pad, groups):
    require_gpu(dy, w)
    dy, w = dy.contiguous(), w.contiguous()
    B, Cout, Tout = dy.shape
    KS = w.shape[2]
    dx = torch.empty((B, Cin, Tin), device=dy.device, dtype=torch.float32)
    check(tlib().svc_gconv1d_dgrad_f32(ptr(dy), ptr(w), ptr(dx), B, Cin, Cout, Tin, Tout, KS, stride, pad, groups,
                                       stream_ptr()), "gconv1d_dgrad")
    return dx


def gconv1d_wgrad(dy, x, KS, stride, pad, groups):
    require_gpu(dy, x)
    dy, x = dy.contiguous(), x.contiguous()
    B, Cout, Tout = dy.shape
    _, Cin, Tin = x.shape
    dw = torch.empty((Cout, Cin // groups, KS), device=dy.device, dtype=torch.float32)
    check(tlib().svc_gconv1d_wgrad_f32(ptr(dy), ptr(x), ptr(dw), B, Cin, Cout, Tin, Tout, KS, stride, pad, groups,
                                       stream_ptr()), "gconv1d_wgrad")
    return dw


def reduce_scalar(op, a, b=None, c=None, d=None, scale=1.0, acc=None):
    """acc (float64 [1] device tensor) += scale * sum f(a,b,c,d); returns acc."""
    require_gpu(a, b, c, d)
    ts = [t.contiguous() if t is not None else None for t in (a, b, c, d)]
    if acc is None:
        acc = torch.zeros(1, device=a.device, dtype=torch.float64)
    check(tlib().svc_reduce_scalar_f64(op, ptr(ts[0]), ptr(ts[1]), ptr(ts[2]), ptr(ts[3]), ts[0].numel(),
                                       C.c_void_p(acc.data_ptr()), float(scale), stream_ptr()), "reduce_scalar")
    return acc


def nonfinite_guard(scalars, counter):
    """counter (int32[3] on the device, zeroed once): [0] += non-finite values among the 0-dim fp32 device tensors `scalars`
    (at most 8), [1] = launch number that last saw one, [2] += 1.  No host sync; capturable."""
    require_gpu(*scalars)
    if counter.dtype != torch.int32 or counter.numel() < 3 or not counter.is_cuda:
        raise SvcError("nonfinite_guard: counter must be an int32[3] device tensor")
    arr = (C.c_void_p * len(scalars))(*[t.data_ptr() for t in scalars])
    check(tlib().svc_nonfinite_guard_f32(arr, len(scalars), C.c_void_p(counter.data_ptr()), stream_ptr()), "nonfinite_guard")


def f64_to_f32(acc):
    out = torch.empty(acc.shape, device=acc.device, dtype=torch.float32)
    check(tlib().svc_f64_to_f32(C.c_void_p(acc.data_ptr()), ptr(out), acc.numel(), stream_ptr()), "f64_to_f32")
    return out


def adamw_step(p, g, m, v, hyper):
    """One fused AdamW update of the flat buffers; hyper = device float[7] (lr, b1, b2, eps, wd, step, grad_scale)."""
    require_gpu(p, g, m, v, hyper)
    check(tlib().svc_adamw_f32(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(hyper), stream_ptr()), "adamw")


def adamw_advance(hyper):
    check(tlib().svc_adamw_advance(ptr(hyper), stream_ptr()), "adamw_advance")


# ---- second batch of training entry points (layernorm / attention pieces / embedding / reparam / nsf / kl / stft) ----
FFT_EXPORTS = ["svc_rfft_plan_create", "svc_rfft_plan_destroy", "svc_rfft_forward_f32", "svc_rfft_inverse_f32",
               "svc_cmag_c_f32", "svc_cmag_c_bwd_f32"]
EXPORTS += FFT_EXPORTS
_fft_bound = False
_fft_plans = {}


def _fftlib():
    global _fft_bound
    L = lib()
    if not _fft_bound:
        i, ll, vp = C.c_int, C.c_longlong, C.c_void_p
        L.svc_rfft_plan_create.argtypes = [i, i, C.POINTER(vp), C.POINTER(ll)]
        L.svc_rfft_plan_destroy.argtypes = [vp]
        L.svc_rfft_forward_f32.argtypes = [vp, _f32p, _f32p, vp, vp]
        L.svc_rfft_inverse_f32.argtypes = [vp, _f32p, _f32p, vp, vp]
        L.svc_cmag_c_f32.argtypes = [_f32p, _f32p, ll, C.c_float, vp]
        L.svc_cmag_c_bwd_f32.argtypes = [_f32p, _f32p, _f32p, _f32p, ll, i, vp]
        _fft_bound = True
    return L


def _rfft_plan(n, batch, device):
    """(plan handle, work buffer) for batched length-n real FFTs, created once per (n, batch, device)."""
    key = (n, batch, str(device))
    ent = _fft_plans.get(key)
    if ent is None:
        plan, wb = C.c_void_p(), C.c_longlong()
        check(_fftlib().svc_rfft_plan_create(n, batch, C.byref(plan), C.byref(wb)), "rfft_plan_create")
        work = torch.empty(max(int(wb.value), 16), device=device, dtype=torch.uint8)
        ent = (plan, work)
        _fft_plans[key] = ent
    return ent


def rfft_mag(frames, eps):
    """frames [R, n] -> (z [R, n/2+1, 2] interleaved half spectrum, mag [R, n/2+1] = sqrt(re^2+im^2+eps))."""
    require_gpu(frames)
    frames = frames.contiguous()
    R, n = frames.shape
    plan, work = _rfft_plan(n, R, frames.device)
    z = torch.empty((R, n // 2 + 1, 2), device=frames.device, dtype=torch.float32)
    check(_fftlib().svc_rfft_forward_f32(plan, ptr(frames), ptr(z), ptr(work), stream_ptr()), "rfft_forward")
    mag = torch.empty((R, n // 2 + 1), device=frames.device, dtype=torch.float32)
    check(_fftlib().svc_cmag_c_f32(ptr(z), ptr(mag), mag.numel(), eps, stream_ptr()), "cmag_c")
    return z, mag


def rfft_mag_bwd(z, mag, dmag, n):
    """Gradient of rfft_mag w.r.t. the frames: [R, n]."""
    require_gpu(z, mag, dmag)
    dmag = dmag.contiguous()
    R, bins = mag.shape
    plan, work = _rfft_plan(n, R, z.device)
    gz = torch.empty_like(z)
    check(_fftlib().svc_cmag_c_bwd_f32(ptr(z), ptr(mag), ptr(dmag), ptr(gz), mag.numel(), bins, stream_ptr()), "cmag_c_bwd")
    dx = torch.empty((R, n), device=z.device, dtype=torch.float32)
    check(_fftlib().svc_rfft_inverse_f32(plan, ptr(gz), ptr(dx), ptr(work), stream_ptr()), "rfft_inverse")
    return dx


TRAIN_EXPORTS2 = [
    "svc_layernorm_fwd_f32", "svc_layernorm_bwd_f32", "svc_attn_softmax_fwd_f32", "svc_attn_softmax_bwd_f32",
    "svc_attn_softmax_fwd_rng_f32", "svc_attn_softmax_bwd_rng_f32", "svc_dropout_rng_f32", "svc_band_gather_f32", "svc_band_scatter_add_f32", "svc_embed_fwd_f32", "svc_embed_bwd_f32", "svc_reparam_bwd_f32",
    "svc_nsf_source_train_f32", "svc_nsf_linear_fwd_f32", "svc_nsf_linear_bwd_f32", "svc_kl_fwd_f64", "svc_kl_bwd_f32",
    "svc_stft_frame_f32", "svc_stft_frame_bwd_f32", "svc_dft_basis_f32", "svc_cmag_f32", "svc_cmag_bwd_f32",
    "svc_lrelu_tail_fwd_f32", "svc_lrelu_tail_bwd_f32", "svc_spectral_norm_fwd_f32", "svc_spectral_norm_bwd_f32",
]
EXPORTS += TRAIN_EXPORTS2
_train2_bound = False


def t2lib():
    global _train2_bound
    L = tlib()
    if not _train2_bound:
        i, f, ll, vp = C.c_int, C.c_float, C.c_longlong, C.c_void_p
        L.svc_layernorm_fwd_f32.argtypes = [_f32p] * 6 + [i, i, i, f, vp]
        L.svc_layernorm_bwd_f32.argtypes = [_f32p] * 8 + [i, i, i, vp]
        L.svc_attn_softmax_fwd_f32.argtypes = [_f32p] * 3 + [i] * 5 + [_f32p, C.c_float, _f32p, vp]
        L.svc_attn_softmax_bwd_f32.argtypes = [_f32p] * 2 + [i] * 3 + [_f32p, C.c_float, _f32p, i, vp]
        L.svc_attn_softmax_fwd_rng_f32.argtypes = [_f32p] * 3 + [i] * 5 + [vp, i, C.c_float, _f32p, vp]
        L.svc_attn_softmax_bwd_rng_f32.argtypes = [_f32p] * 2 + [i] * 3 + [vp, i, C.c_float, _f32p, i, vp]
        L.svc_dropout_rng_f32.argtypes = [_f32p, _f32p, ll, vp, i, C.c_float, vp]
        L.svc_band_gather_f32.argtypes = [_f32p, _f32p, ll, i, i, vp]
        L.svc_lrelu_tail_fwd_f32.argtypes = [_f32p, _f32p, ll, i, i, f, vp]
        L.svc_spectral_norm_fwd_f32.argtypes = [_f32p] * 6 + [i, i, i, f, vp]
        L.svc_spectral_norm_bwd_f32.argtypes = [_f32p] * 6 + [vp, i, i, vp]
        L.svc_lrelu_tail_bwd_f32.argtypes = [_f32p, _f32p, _f32p, ll, i, i, f, vp]
        L.svc_band_scatter_add_f32.argtypes = [_f32p, _f32p, ll, i, i, vp]
        L.svc_embed_fwd_f32.argtypes = [vp, _f32p, _f32p, i, i, i, vp]
        L.svc_embed_bwd_f32.argtypes = [vp, _f32p, _f32p, i, i, i, i, vp]
        L.svc_reparam_bwd_f32.argtypes = [_f32p] * 5 + [i, i, i, f, vp]
        L.svc_nsf_source_train_f32.argtypes = [_f32p] * 8 + [i] * 4 + [f] * 3 + [vp]
        L.svc_nsf_linear_fwd_f32.argtypes = [_f32p] * 4 + [ll, i, vp]
        L.svc_nsf_linear_bwd_f32.argtypes = [_f32p] * 5 + [ll, i, vp]
        L.svc_kl_fwd_f64.argtypes = [_f32p] * 5 + [vp, i, i, i, vp]
        L.svc_kl_bwd_f32.argtypes = [_f32p] * 9 + [i, i, i, vp]
        L.svc_stft_frame_f32.argtypes = [_f32p] * 3 + [i] * 6 + [vp]
        L.svc_stft_frame_bwd_f32.argtypes = [_f32p] * 3 + [i] * 6 + [vp]
        L.svc_dft_basis_f32.argtypes = [_f32p, _f32p, i, i, vp]
        L.svc_cmag_f32.argtypes = [_f32p] * 3 + [ll, f, vp]
        L.svc_cmag_bwd_f32.argtypes = [_f32p] * 6 + [ll, vp]
        _train2_bound = True
    return L


def layernorm_fwd(x, gamma, beta, eps):
    require_gpu(x, gamma, beta)
    x = x.contiguous()
    B, Cc, T = x.shape
    y = torch.empty_like(x)
    mean = torch.empty((B, T), device=x.device, dtype=torch.float32)
    rstd = torch.empty((B, T), device=x.device, dtype=torch.float32)
    check(t2lib().svc_layernorm_fwd_f32(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), B, Cc, T, eps,
                                        stream_ptr()), "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(x, gamma, dy, mean, rstd):
    require_gpu(x, gamma, dy, mean, rstd)
    B, Cc, T = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty_like(gamma)
    db = torch.empty_like(gamma)
    check(t2lib().svc_layernorm_bwd_f32(ptr(x), ptr(gamma), ptr(dy.contiguous()), ptr(mean), ptr(rstd), ptr(dx), ptr(dg),
                                        ptr(db), B, Cc, T, stream_ptr()), "layernorm_bwd")
    return dx, dg, db


class HashDraw:
    """The uniform draws of one dropout site as a counter-based function u(seed, site, element) evaluated INSIDE the consuming
    kernels (svc_attn_softmax_{fwd,bwd}_rng_f32, svc_dropout_rng_f32) instead of a torch.rand tensor: `seed` is an int64[1] device
    tensor that stays untouched between the site's forward and backward, `site` numbers the site within its encoder call."""

    def __init__(self, seed, site):
        if seed.dtype != torch.int64 or not seed.is_cuda or seed.numel() != 1:
            raise SvcError("HashDraw: seed must be an int64[1] device tensor")
        self.seed, self.site = seed, int(site)

    def ptr(self):
        return C.c_void_p(self.seed.data_ptr())


def dropout_rng(x, draw, p, out=None):
    """y = x * keep / (1 - p) with the keep decisions of `draw` (HashDraw): forward on x, backward on dy."""
    require_gpu(x, out)
    x = x.contiguous()
    if out is None:
        out = torch.empty_like(x)
    check(t2lib().svc_dropout_rng_f32(ptr(x), ptr(out), x.numel(), draw.ptr(), draw.site, float(p), stream_ptr()), "dropout_rng")
    return out


def attn_softmax_fwd(S_, rel, mask, B, H, T, window, mask_mode, drop_u=None, p_drop=0.0):
    """In place scores -> probabilities; with drop_u (uniform draws, same shape, or a HashDraw) also returns the dropped
    probabilities P * (u >= p ? 1/(1-p) : 0) (modules/attentions.py:232), else returns S_ itself."""
    if isinstance(drop_u, HashDraw):
        Pd = torch.empty_like(S_)
        check(t2lib().svc_attn_softmax_fwd_rng_f32(ptr(S_), ptr(rel), ptr(mask), B, H, T, window, mask_mode, drop_u.ptr(), drop_u.site,
                                                   float(p_drop), ptr(Pd), stream_ptr()), "attn_softmax_fwd_rng")
        return Pd
    require_gpu(drop_u)
    Pd = torch.empty_like(S_) if drop_u is not None else None
    check(t2lib().svc_attn_softmax_fwd_f32(ptr(S_), ptr(rel), ptr(mask), B, H, T, window, mask_mode, ptr(drop_u),
                                           float(p_drop), ptr(Pd), stream_ptr()), "attn_softmax_fwd")
    return S_ if Pd is None else Pd


def attn_softmax_bwd(P, dP, B, H, T, drop_u=None, p_drop=0.0, mask=None, mask_mode=0):
    if isinstance(drop_u, HashDraw):
        check(t2lib().svc_attn_softmax_bwd_rng_f32(ptr(P), ptr(dP), B, H, T, drop_u.ptr(), drop_u.site, float(p_drop), ptr(mask),
                                                   mask_mode, stream_ptr()), "attn_softmax_bwd_rng")
        return dP
    check(t2lib().svc_attn_softmax_bwd_f32(ptr(P), ptr(dP), B, H, T, ptr(drop_u), float(p_drop), ptr(mask), mask_mode,
                                           stream_ptr()), "attn_softmax_bwd")
    return dP


def spectral_norm_fwd(W, u, v, power_iteration, eps=1e-12):
    """-> (w = W / sigma, sigma [1]); with power_iteration the buffers u, v are updated in place first (one iteration)."""
    require_gpu(W, u, v)
    if not (W.is_contiguous() and u.is_contiguous() and v.is_contiguous()):
        raise SvcError("spectral_norm: weight_orig / weight_u / weight_v must be contiguous")
    R = W.shape[0]
    K = W.numel() // R
    if u.numel() != R or v.numel() != K:
        raise SvcError(f"spectral_norm: u [{u.numel()}] / v [{v.numel()}] do not match the [{R}, {K}] weight matrix")
    w = torch.empty_like(W)
    sigma = torch.empty(1, device=W.device, dtype=torch.float32)
    tmp = torch.empty(R, device=W.device, dtype=torch.float32)
    check(t2lib().svc_spectral_norm_fwd_f32(ptr(W), ptr(u), ptr(v), ptr(w), ptr(sigma), ptr(tmp), R, K,
                                            1 if power_iteration else 0, float(eps), stream_ptr()), "spectral_norm_fwd")
    return w, sigma


def spectral_norm_bwd(W, u, v, sigma, g):
    require_gpu(W, u, v, sigma, g)
    R = W.shape[0]
    K = W.numel() // R
    g = g.contiguous()
    dW = torch.empty_like(W)
    ws = torch.empty(1, device=W.device, dtype=torch.float64)
    check(t2lib().svc_spectral_norm_bwd_f32(ptr(W), ptr(u), ptr(v), ptr(sigma), ptr(g), ptr(dW), C.c_void_p(ws.data_ptr()), R, K,
                                            stream_ptr()), "spectral_norm_bwd")
    return dW


def _tail_rows(x):
    if not x.is_contiguous() or x.shape[-1] % 4 or x.data_ptr() % 16:
        raise SvcError("lrelu_tail: needs a contiguous tensor whose last dimension is a multiple of 4 (16-byte rows)")
    return x.numel() // x.shape[-1], x.shape[-1]


def graph_capture(graph, pool=None):
    """torch.cuda.graph(...) for this engine's captures.  With a process group initialised, ProcessGroupNCCL's watchdog thread
    polls the events of earlier collectives (hipEventQuery) from ITS thread; under the default "global" capture mode that
    call is illegal while ANY stream captures and aborts the process ("operation not permitted when stream is capturing",
    seen on MI355X / ROCm 7 with RCCL 2.26).  "thread_local" restricts the check to the capturing thread."""
    kw = {} if pool is None else {"pool": pool}
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        kw["capture_error_mode"] = "thread_local"
    return torch.cuda.graph(graph, **kw)


def capture_error_mode():
    """hipStreamCaptureMode of this engine's captures (see graph_capture)."""
    import torch.distributed as dist
    return "thread_local" if dist.is_available() and dist.is_initialized() else "global"


class capture_stream:
    """What torch.cuda.graph does around capture_begin / capture_end, for captures that are begun and ended by hand (the
    data-parallel iteration is cut into several graphs inside one backward pass): device idle, cached blocks released,
    a side stream current for the duration."""
    _stream = None

    def __enter__(self):
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        if capture_stream._stream is None:
            capture_stream._stream = torch.cuda.Stream()
        self.ctx = torch.cuda.stream(capture_stream._stream)
        self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        return self.ctx.__exit__(*a)


def lrelu_tail_fwd(x, valid, slope):
    """y = leaky_relu(x, slope) on the first `valid` columns of every row, 0 on the rest (rows = all leading dims)."""
    require_gpu(x)
    rows, P = _tail_rows(x)
    y = torch.empty_like(x)
    check(t2lib().svc_lrelu_tail_fwd_f32(ptr(x), ptr(y), rows, P, int(valid), float(slope), stream_ptr()), "lrelu_tail_fwd")
    return y


def lrelu_tail_bwd(y, dy, valid, slope):
    require_gpu(y, dy)
    dy = dy.contiguous()
    rows, P = _tail_rows(y)
    dx = torch.empty_like(y)
    check(t2lib().svc_lrelu_tail_bwd_f32(ptr(y), ptr(dy), ptr(dx), rows, P, int(valid), float(slope), stream_ptr()),
          "lrelu_tail_bwd")
    return dx


def band_gather(M, n_rows, T, window):
    band = torch.empty((n_rows, 2 * window + 1), device=M.device, dtype=torch.float32)
    check(t2lib().svc_band_gather_f32(ptr(M), ptr(band), n_rows, T, window, stream_ptr()), "band_gather")
    return band


def band_scatter_add(M, band, n_rows, T, window):
    check(t2lib().svc_band_scatter_add_f32(ptr(M), ptr(band), n_rows, T, window, stream_ptr()), "band_scatter_add")
    return M


def embed_fwd(idx, W):
    """idx [B,T] int64, W [N,C] -> [B,C,T]."""
    require_gpu(W)
    idx = idx.contiguous()
    B, T = idx.shape
    Cc = W.shape[1]
    y = torch.empty((B, Cc, T), device=W.device, dtype=torch.float32)
    check(t2lib().svc_embed_fwd_f32(C.c_void_p(idx.data_ptr()), ptr(W.contiguous()), ptr(y), B, Cc, T, stream_ptr()),
          "embed_fwd")
    return y


def embed_bwd(idx, dy, n_rows):
    idx = idx.contiguous()
    dy = dy.contiguous()
    B, Cc, T = dy.shape
    dW = torch.zeros((n_rows, Cc), device=dy.device, dtype=torch.float32)
    check(t2lib().svc_embed_bwd_f32(C.c_void_p(idx.data_ptr()), ptr(dy), ptr(dW), B, Cc, T, n_rows, stream_ptr()), "embed_bwd")
    return dW


def reparam_bwd(stats, noise, mask, dz, scale):
    B, C2, T = stats.shape
    d = torch.empty_like(stats)
    check(t2lib().svc_reparam_bwd_f32(ptr(stats), ptr(noise), ptr(mask), ptr(dz.contiguous()), ptr(d), B, C2 // 2, T, scale,
                                      stream_ptr()), "reparam_bwd")
    return d


def nsf_source_train(f0, rand_ini, noise, lin_w, lin_b, upp, sampling_rate, sine_amp=0.1, noise_std=0.003):
    require_gpu(f0, rand_ini, noise, lin_w, lin_b)
    B, T = f0.shape
    H = rand_ini.shape[1]
    L = T * upp
    har = torch.empty((B, 1, L), device=f0.device, dtype=torch.float32)
    waves = torch.empty((B, L, H), device=f0.device, dtype=torch.float32)
    nbytes = lib().svc_nsf_source_scratch_bytes(B, T, H)
    scratch = torch.empty((nbytes + 7) // 8, device=f0.device, dtype=torch.float64)
    check(t2lib().svc_nsf_source_train_f32(ptr(f0.contiguous()), ptr(rand_ini.contiguous()), ptr(noise.contiguous()),
                                           ptr(lin_w.contiguous()), ptr(lin_b.contiguous()), ptr(har), ptr(waves),
                                           ptr(scratch), B, T, upp, H, float(sampling_rate), sine_amp, noise_std,
                                           stream_ptr()), "nsf_source_train")
    return har, waves


def nsf_linear_bwd(waves, har, dhar):
    H = waves.shape[-1]
    dw = torch.empty(H, device=waves.device, dtype=torch.float32)
    db = torch.empty(1, device=waves.device, dtype=torch.float32)
    check(t2lib().svc_nsf_linear_bwd_f32(ptr(waves), ptr(har), ptr(dhar.contiguous()), ptr(dw), ptr(db), har.numel(), H,
                                         stream_ptr()), "nsf_linear_bwd")
    return dw, db


def kl_fwd(z_p, logs_q, m_p, logs_p, mask):
    B, Cc, T = z_p.shape
    acc = torch.zeros(2, device=z_p.device, dtype=torch.float64)
    check(t2lib().svc_kl_fwd_f64(ptr(z_p.contiguous()), ptr(logs_q.contiguous()), ptr(m_p.contiguous()),
                                 ptr(logs_p.contiguous()), ptr(mask.contiguous()), C.c_void_p(acc.data_ptr()), B, Cc, T,
                                 stream_ptr()), "kl_fwd")
    return acc


def kl_bwd(z_p, m_p, logs_p, mask, g):
    B, Cc, T = z_p.shape
    outs = [torch.empty_like(z_p) for _ in range(4)]
    check(t2lib().svc_kl_bwd_f32(ptr(z_p), ptr(m_p), ptr(logs_p), ptr(mask), ptr(g), ptr(outs[0]), ptr(outs[1]),
                                 ptr(outs[2]), ptr(outs[3]), B, Cc, T, stream_ptr()), "kl_bwd")
    return outs   # dz_p, dlogs_q, dm_p, dlogs_p


def stft_frame(y, win, NF, nfft, hop, pad):
    y = y.contiguous()
    B, L = y.shape
    frames = torch.empty((B, NF, nfft), device=y.device, dtype=torch.float32)
    check(t2lib().svc_stft_frame_f32(ptr(y), ptr(win), ptr(frames), B, L, NF, nfft, hop, pad, stream_ptr()), "stft_frame")
    return frames


def stft_frame_bwd(dframes, win, L, hop, pad):
    dframes = dframes.contiguous()
    B, NF, nfft = dframes.shape
    dy = torch.empty((B, L), device=dframes.device, dtype=torch.float32)
    check(t2lib().svc_stft_frame_bwd_f32(ptr(dframes), ptr(win), ptr(dy), B, L, NF, nfft, hop, pad, stream_ptr()),
          "stft_frame_bwd")
    return dy


def dft_basis(N, NB, device):
    cs = torch.empty((N, NB), device=device, dtype=torch.float32)
    sn = torch.empty((N, NB), device=device, dtype=torch.float32)
    check(t2lib().svc_dft_basis_f32(ptr(cs), ptr(sn), N, NB, stream_ptr()), "dft_basis")
    return cs, sn


def cmag(re, im, eps):
    mag = torch.empty_like(re)
    check(t2lib().svc_cmag_f32(ptr(re), ptr(im), ptr(mag), re.numel(), eps, stream_ptr()), "cmag")
    return mag


def cmag_bwd(re, im, mag, dmag):
    dre, dim = torch.empty_like(re), torch.empty_like(re)
    check(t2lib().svc_cmag_bwd_f32(ptr(re), ptr(im), ptr(mag), ptr(dmag.contiguous()), ptr(dre), ptr(dim), re.numel(),
                                   stream_ptr()), "cmag_bwd")
    return dre, dim
