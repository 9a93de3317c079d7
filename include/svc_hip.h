/*
 * svc_hip.h — C-ABI of libsvc_hip.so, the MI355X (gfx950 / CDNA4) engine for the so-vits-svc
 * SynthesizerTrn hot path.
 *
 * The reference (svc-develop-team/so-vits-svc) has no FFI layer: its boundary is the Python class API
 * (models.SynthesizerTrn, vdecoder.hifigan.models.Generator, ...).  The Python mirror of that API in
 * so-vits-svc_amd/ binds these entry points with ctypes (so-vits-svc_amd/svc_hip.py); each entry point
 * below cites the reference code (path:line under /root/reference) whose arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data unless stated otherwise; the caller owns all memory;
 *   - activations are [B, C, T] with time contiguous; element (b,c,t) of tensor `x` lives at
 *     x + b*x_bs + c*x_cs + t (strides in ELEMENTS; channel strides may be negative, which is how the
 *     flow's channel Flip (modules/modules.py:232-239) is folded into its neighbours);
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - every function returns 0 on success or a negative svc_status; svc_last_error() returns a
 *     thread-local human readable message.  No exceptions cross this boundary.
 *   - nothing here allocates or frees device memory and nothing synchronises the device, so every entry
 *     point may be captured into a hipGraph.
 */
#ifndef SVC_HIP_H
#define SVC_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef enum svc_status {
  SVC_OK = 0,
  SVC_ERR_BAD_ARG = -1,
  SVC_ERR_UNSUPPORTED = -2,
  SVC_ERR_HIP = -3
} svc_status;

const char* svc_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int svc_abi_version(void);
/* Fills name[0..len) with the gcnArchName of the current device, returns number of CUs (or <0). */
int svc_device_info(char* name, int len);

/* ------------------------------------------------------------------------------------------------
 * Per-launch profiling (used by bench.py for the `roofline` object): when enabled every launcher
 * brackets its kernel with hipEvents on the launch stream and accumulates (calls, ms, flop, bytes)
 * per kernel family.  Must be disabled while capturing a hipGraph.
 * ---------------------------------------------------------------------------------------------- */
int svc_prof_enable(int on);
int svc_prof_reset(void);
/* Synchronises outstanding events and writes one line per kernel family:
 * "name calls total_ms flop bytes\n" into buf.  Returns bytes written (or <0). */
int svc_prof_report(char* buf, int len);

/* ------------------------------------------------------------------------------------------------
 * Weight packing.  torch.nn.utils.weight_norm (vdecoder/hifigan/models.py:41-56,335,340-342,355;
 * modules/modules.py:91-108 via modules/DSConv.py:65-70) stores (weight_g, weight_v) and recomputes
 * w = g * v / ||v|| on every forward; inference never removes it (inference/infer_tool.py:189-202).
 * We fold once into the layout the MFMA kernels read: dst[ci][k][co] (co fastest, CoutP = stride).
 * ---------------------------------------------------------------------------------------------- */
/* Conv1d weight v:[Cout][Cin][KS] (+ optional g:[Cout], norm over (Cin,KS)) -> dst:[Cin][KS][CoutP].
 * gate_half > 0 permutes output rows so that row 64*i+r (r<32) = channel 32*i+r and row 64*i+32+r =
 * channel gate_half+32*i+r (tanh/sigmoid halves adjacent: commons.fused_add_tanh_sigmoid_multiply,
 * modules/commons.py:129-136). */
int svc_pack_conv1d_weight(const float* v, const float* g, float* dst, int Cout, int Cin, int KS,
                           int CoutP, int gate_half, void* stream);
/* ConvTranspose1d weight v:[Cin][Cout][KS] (+ optional g:[Cin], norm over (Cout,KS): weight_norm dim=0
 * on a transposed conv, vdecoder/hifigan/models.py:340-342) -> dst:[Cin][KS][CoutP]. */
int svc_pack_convt1d_weight(const float* v, const float* g, float* dst, int Cin, int Cout, int KS,
                            int CoutP, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused Conv1d, fp32 in / fp32 accumulate on the matrix pipe (v_mfma_f32_32x32x2_f32 /
 * v_mfma_f32_16x16x4_f32), LDS-tiled along time.  Replaces every dense nn.Conv1d on the path:
 * models.py:400 (pre), modules/attentions.py:176-179,331-332 (q,k,v,o, FFN), models.py:139 (proj),
 * modules/modules.py:98-108,285-287 (WN in/res_skip, coupling pre/post),
 * vdecoder/hifigan/models.py:41-56,335,358 (ResBlock convs, conv_pre, cond).
 *
 *   v[b,co,t] = sum_{ci,k} w[co,ci,k] * pre(x[b,ci,t + k*dil - pad_left]) + bias[co] + cond[b,co,t]
 *   pre(u)   = leaky_relu(u * premask[b,t'], pre_slope)            (pre_slope = 1 -> identity)
 * epilogues (`epi`):
 *   SVC_EPI_PLAIN    v = act(v); v *= mask[b,t]; v = residual(v); v += beta*y_old; y = v / out_div
 *                    res_mode 0: none | 1: v + res | 2: (res - v) * mask | 3: v + res * mask
 *   SVC_EPI_GATE     Cout = 2H rows packed with gate_half=H; y[b,c,t] = tanh(v_c) * sigmoid(v_{H+c})
 *   SVC_EPI_RES_SKIP rows c < skip_from : y[b,c,t]  = (res[b,c,t] + v) * mask[b,t]      (y may alias res)
 *                    rows c >= skip_from: y2[b,c-skip_from,t] = v + (beta ? y2_old : 0)
 * ---------------------------------------------------------------------------------------------- */
enum { SVC_EPI_PLAIN = 0, SVC_EPI_GATE = 1, SVC_EPI_RES_SKIP = 2 };
enum { SVC_ACT_NONE = 0, SVC_ACT_RELU = 1, SVC_ACT_TANH = 2, SVC_ACT_LRELU = 3 };

typedef struct svc_conv1d_args {
  const float* x;
  const float* w;       /* packed [Cin][KS][CoutP] */
  const float* bias;    /* [Cout] or NULL */
  const float* cond;    /* NULL or element (b,co,t) at cond + b*cond_bs + co*cond_cs + t*cond_ts */
  const float* mask;    /* NULL or [B,Tout] at mask + b*mask_bs + t */
  const float* premask; /* NULL or [B,Tin]  at premask + b*premask_bs + t */
  const float* res;     /* NULL or residual, element (b,co,t) at res + b*res_bs + co*res_cs + t */
  float* y;
  float* y2;            /* SVC_EPI_RES_SKIP only */
  long long x_bs, x_cs, y_bs, y_cs, res_bs, res_cs, y2_bs, y2_cs;
  long long cond_bs, cond_cs, cond_ts, mask_bs, premask_bs;
  int B, Cin, Cout, Tin, Tout, KS, dil, pad_left, CoutP;
  int epi, post_act, res_mode, skip_from;
  float pre_slope, post_slope, beta, out_div;
} svc_conv1d_args;

int svc_conv1d_f32(const svc_conv1d_args* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SVC_HIP_H */
