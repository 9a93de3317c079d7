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
/* ABI version of this header; bumped on any signature OR argument-struct layout change (the structs are passed by pointer and
 * read in full: a caller built against an older header hands over a shorter struct).  A caller checks
 * svc_abi_version() == SVC_ABI_VERSION once after loading the library (svc_hip.py does) and refuses to run otherwise.
 *   3: svc_conv1d_args.mma / svc_wgrad_args.mma, svc_conv1d_multi_f32
 *   4: svc_attention_args grew {ws, ws_bytes} (key-split workspace), svc_gemm_args grew {split_k_atomic}; the 16-bit / split
 *      generator entry points (svc_conv1d_h*, svc_conv1d_hl*, svc_resblock_pair_h / _hl)
 *   5: SVC_MMA_BF16X6 removed (svc_conv1d_args.mma / svc_wgrad_args.mma accept fp32, bf16, fp16 only); split pipeline range guard:
 *      svc_conv1d_h_args.acc_scale, svc_pack_conv1d_hl(scale), svc_resblock_pair_hl(acc_scale1, acc_scale2), svc_hl_range_flag; svc_coupling_fused_h
 *   6: svc_conv1d_args / svc_convt1d_args grew {w_d4} (lane-linear weight pack of the short-sequence kernel), svc_pack_conv1d_d4,
 *      svc_conv1d_wants_d4 */
#define SVC_ABI_VERSION 6
int svc_abi_version(void);
/* Fills name[0..len) with the gcnArchName of the current device, returns number of CUs (or <0). */
int svc_device_info(char* name, int len);

/* ------------------------------------------------------------------------------------------------
 * Per-launch profiling (used by bench.py for the `roofline` object): when enabled every launcher
 * brackets its kernel with hipEvents on the launch stream and accumulates (calls, ms, flop, bytes)
 * per kernel family.  Must be disabled while capturing a hipGraph.
 * ---------------------------------------------------------------------------------------------- */
int svc_debug_empty_kernel(void* stream);   /* one empty one-wave kernel (bench.py's launch-latency probe) */
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
 * on a transposed conv, vdecoder/hifigan/models.py:340-342) -> stride*Cin*M*CoutP floats, M = ceil(KS/stride), private to
 * svc_conv_transpose1d_f32: polyphase blocks dst[p][ci][mr][co] = w[ci][co][p + (M-1-mr)*stride], or — strides 2, 4, 8, 16 —
 * the phases as rows of one convolution, dst[ci][mr][co*stride + p]. */
int svc_pack_convt1d_weight(const float* v, const float* g, float* dst, int Cin, int Cout, int KS,
                            int CoutP, int stride, void* stream);

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
 *                    rows c >= skip_from: y2[b,c-skip_from,t] = v + (beta ? y2_old : 0)   (* mask if res_mode==1:
 *                                         the final `output * x_mask` of WN.forward, modules/modules.py:138)
 * ---------------------------------------------------------------------------------------------- */
enum { SVC_EPI_PLAIN = 0, SVC_EPI_GATE = 1, SVC_EPI_RES_SKIP = 2 };
enum { SVC_ACT_NONE = 0, SVC_ACT_RELU = 1, SVC_ACT_TANH = 2, SVC_ACT_LRELU = 3, SVC_ACT_GELU = 4 /* exact erf GELU */ };

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
  /* polyphase output mapping (ordinary conv: n_phase=1, y_ts=1, y_t0=0, y_len=Tout): the kernel runs n_phase
   * dense sub-convolutions, phase p reading the packed weight block w + p*w_phase_stride and writing output
   * sample t_out = t*y_ts + y_t0 + p for t in [0,Tout), kept when 0 <= t_out < y_len.  mask/cond/res/y are all
   * indexed by t_out.  This is how svc_conv_transpose1d_f32 lowers ConvTranspose1d. */
  int n_phase, y_ts, y_t0, y_len;
  long long w_phase_stride;
  float pre_slope, post_slope, beta, out_div;
  /* Matrix-pipe operand format (the reference's `fp16_run` / `half_type: bf16` autocast mode, train.py:114,143,166,187,198:
   * convolutions take bf16 operands and accumulate in fp32).  SVC_MMA_F32 (0): fp32 operands, v_mfma_f32_32x32x2_f32 — the
   * default and the only format of inference.  SVC_MMA_BF16 (1): activations and weights are rounded to bf16 (round to nearest
   * even) as the operands are fetched and multiplied on v_mfma_f32_32x32x16_bf16; accumulation, bias, activation, residual and
   * the stored result stay fp32 (tensors in HBM are fp32 in both modes, the master weights too).  Honoured by the LDS-DMA tilings
   * of plain convolutions with 16-byte aligned rows and Cin a multiple of 16; every other shape runs in fp32 — never less
   * precise than asked. */
  int mma;
  /* Optional second pack of the same weights for the register-fed short-sequence kernel (svc_pack_conv1d_d4 below), or NULL.
   * That kernel feeds its MFMAs straight from L2; with the [Cin][KS][CoutP] pack every operand is a 4-byte-per-lane load and the
   * launch is bound by the vector-memory ADDRESS rate (one wave-wide load per 16 cycles per CU whatever its width:
   * profiles/r11e_front_conv_hot_cold.txt), with this pack a lane's four consecutive reduction steps are one 16-byte load. */
  const float* w_d4;
} svc_conv1d_args;
#define SVC_MMA_F32 0
#define SVC_MMA_BF16 1
#define SVC_MMA_F16 2  /* `half_type: fp16`: the same with fp16 operands (v_mfma_f32_32x32x16_f16); the caller scales the loss (GradScaler) */

int svc_conv1d_f32(const svc_conv1d_args* a, void* stream);

/* The lane-linear operand pack of svc_conv1d_args.w_d4, derived from the standard pack `wp` [Cin][KS][CoutP] (Cin even,
 * CoutP a multiple of 32): dst [CoutP/32][NG][64][4] floats, NG = ceil(Cin/2 * KS / 4);
 *   dst[rt][G][lk*32 + ln][e] = wp[((2*pr + lk)*KS + k)*CoutP + rt*32 + ln]   with  pr*KS + k = 4*G + e   (0 past the last pair).
 * svc_pack_conv1d_d4_floats returns the element count of dst. */
long long svc_pack_conv1d_d4_floats(int Cin, int KS, int CoutP);
/* 1 when svc_conv1d_f32 would read a lane-linear pack for these arguments (the launch takes the register-fed kernel and the shape
 * fits its banks), else 0; launches nothing, `w_d4` is ignored.  A caller asks before it makes the second pack. */
int svc_conv1d_wants_d4(const svc_conv1d_args* a);
int svc_pack_conv1d_d4(const float* wp, float* dst, int Cin, int KS, int CoutP, void* stream);

int svc_debug_bf16(int mode);            /* 0 / 1: ignore / honour SVC_MMA_BF16 requests (A/B); -1: bf16 conv launches so far */
int svc_debug_wgrad_bf16_launches(void);

/* ------------------------------------------------------------------------------------------------
 * ConvTranspose1d (upsampling `ups[i]`, vdecoder/hifigan/models.py:340-342,378), lowered to `stride`
 * dense polyphase sub-convolutions on the same MFMA kernel:
 *   y[b,co,t] = bias[co] + res[b,co,t] + sum_{ci,k} w[ci,co,k] * lrelu(x[b,ci,j], pre_slope),  t = j*stride - padding + k
 * Tout must equal (Tin-1)*stride - 2*padding + KS.  `res` (optional) is added in the epilogue: it carries
 * noise_convs[i](har_source) (:379-381).
 * ---------------------------------------------------------------------------------------------- */
typedef struct svc_convt1d_args {
  const float* x;
  const float* w;    /* from svc_pack_convt1d_weight */
  const float* bias; /* [Cout] or NULL */
  const float* res;  /* NULL or [B,Cout,Tout] */
  float* y;
  long long x_bs, x_cs, y_bs, y_cs, res_bs, res_cs;
  int B, Cin, Cout, Tin, Tout, KS, stride, padding, CoutP;
  float pre_slope;
  const float* w_d4; /* NULL, or svc_pack_conv1d_d4 of the phases-as-rows pack ([Cin][M][stride*CoutP]: power-of-two strides) */
} svc_convt1d_args;

int svc_conv_transpose1d_f32(const svc_convt1d_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Direct (VALU) Conv1d for the non-GEMM-shaped layers: noise_convs[i] Conv1d(1->C, k=2s, stride s)
 * (vdecoder/hifigan/models.py:343-348,379), conv_post Conv1d(C->1,k7) + tanh (:355,390-392),
 * F0Decoder.f0_prenet / proj (models.py:324-326).  Same packed weight layout as svc_conv1d_f32.
 *   y[b,co,t] = act(sum_{ci,k} w[co,ci,k]*lrelu(x[b,ci,t*stride + k*dil - pad_left], pre_slope) + bias[co])
 *               * mask[b,t] + res[b,co,t]
 * ---------------------------------------------------------------------------------------------- */
typedef struct svc_conv1d_direct_args {
  const float* x;
  const float* w;    /* packed [Cin][KS][CoutP] */
  const float* bias; /* [Cout] or NULL */
  const float* mask; /* NULL or [B,Tout] */
  const float* res;  /* NULL or [B,Cout,Tout] */
  float* y;
  long long x_bs, x_cs, y_bs, y_cs, res_bs, res_cs, mask_bs;
  int B, Cin, Cout, Tin, Tout, KS, dil, stride, pad_left, CoutP, post_act;
  float pre_slope, post_slope;
} svc_conv1d_direct_args;

int svc_conv1d_direct_f32(const svc_conv1d_direct_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * One ResBlock1 pair of the NARROW decoder stages in one launch (vdecoder/hifigan/models.py:60-67, one iteration of the
 * loop over dilations):   y = conv2( lrelu( conv1( lrelu(x) ) + b1 ) ) + b2 + x
 * conv1 = Conv1d(C, C, KS, dilation dil1, padding dil1*(KS-1)/2), conv2 = Conv1d(C, C, KS, dilation 1, padding (KS-1)/2),
 * lrelu slope `slope` (0.1).  Built for C = 16 (the last upsample stage, HBM-bound as separate launches), KS in {3, 7, 11} and
 * dilations whose tile (240 + 2 (dil1 + 1)(KS - 1)/2 columns) stays within 512 — SVC_ERR_UNSUPPORTED beyond; wider stages use svc_conv1d_f32.  x, y: [B, C, T] views (time contiguous, x != y); w1, w2: packed
 * [C][KS][CP] weights (svc_pack_conv1d_weight); b1, b2: [C] or NULL.  Epilogue options of the MRF sum: y = (pair(x) +
 * beta * y_old) / out_div.  Results are bit-identical to the two svc_conv1d_f32 launches it replaces.
 * ---------------------------------------------------------------------------------------------- */
typedef struct svc_resblock_pair_args {
  const float* x;
  const float* w1;
  const float* b1;
  const float* w2;
  const float* b2;
  float* y;
  long long x_bs, x_cs, y_bs, y_cs;
  int B, C, T, KS, dil1, CP;
  float slope, beta, out_div;
} svc_resblock_pair_args;

int svc_resblock_pair_f32(const svc_resblock_pair_args* a, void* stream);

/* The whole 16-channel ResBlock1 — all its dilation pairs (vdecoder/hifigan/models.py:60-67: `for c1, c2 in zip(convs1, convs2)`) — in ONE
 * launch:  for j < n_pairs:  x = conv2_j( lrelu( conv1_j( lrelu(x) ) + b1_j ) ) + b2_j + x ;  y = (x + beta * y_old) / out_div.
 * conv1_j has dilation dil[j], conv2_j dilation 1, both KS taps (3, 7 or 11) and 'same' padding; packed weights as for
 * svc_resblock_pair_f32.  Bit-identical to n_pairs svc_resblock_pair_f32 launches (same reduction order, same epilogue expressions);
 * x is read once and y written once.  SVC_ERR_UNSUPPORTED when the dilations' halo leaves fewer than 64 outputs per tile. */
typedef struct svc_resblock16_args {
  const float* x;
  float* y;
  const float* w1[3];
  const float* b1[3];
  const float* w2[3];
  const float* b2[3];
  long long x_bs, x_cs, y_bs, y_cs;
  int B, T, KS, n_pairs, dil[3], CP;
  float slope, beta, out_div;
} svc_resblock16_args;
int svc_resblock16_f32(const svc_resblock16_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * NSF harmonic source: nearest x`upp` f0 upsample (vdecoder/hifigan/models.py:369), SineGen (:138-166,
 * :250-271) and SourceModuleHnNSF (:307-320: Linear(H->1) + tanh), evaluated in closed form per frame
 * (see csrc/nsf_source.hip).  f0:[B,T]  rand_ini:[B,H] (column 0 ignored)  noise:[B,T*upp,H]
 * lin_w:[H] lin_b:[1]  ->  har:[B,T*upp].  `scratch` is caller-provided device memory of
 * svc_nsf_source_scratch_bytes(B,T,H) bytes, 8-byte aligned.
 * ---------------------------------------------------------------------------------------------- */
long long svc_nsf_source_scratch_bytes(int B, int T, int H);
int svc_nsf_source_f32(const float* f0, const float* rand_ini, const float* noise, const float* lin_w,
                       const float* lin_b, float* har, void* scratch, int B, int T, int upp, int H,
                       float sampling_rate, float sine_amp, float noise_std, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Small fused element-wise pieces.
 * ---------------------------------------------------------------------------------------------- */
/* y[b,c,t] = x[b,c,t] * mask[b,t] for arbitrary (possibly negative) channel strides: materialises a channel
 * Flip (modules/modules.py:232-239) or an `x * x_mask` (modules/attentions.py:97) when it cannot be folded. */
int svc_copy_bct_f32(const float* x, float* y, const float* mask, long long x_bs, long long x_cs,
                     long long y_bs, long long y_cs, long long mask_bs, int B, int C, int T, void* stream);
/* Anti-aliased Snake activation of the nsf-snake-hifigan decoder: SnakeAlias.forward
 * (vdecoder/hifiganwithsnake/alias/act.py:125-130) = UpSample1d (alias/resample.py:38-54) -> SnakeBeta log-scale
 * (alias/act.py:79-92) -> DownSample1d (alias/filter.py:93-110), fused: y = down2(snake(up2(x))), x,y [B,C,T] with
 * explicit batch/channel strides (time contiguous); alpha,beta [C] device; taps_host = the 12 Kaiser-sinc taps
 * (alias/filter.py:29-58) in HOST memory (copied into the launch by value: graph-capture safe). */
int svc_snake_alias_f32(const float* x, float* y, const float* alpha, const float* beta, const float* taps_host,
                        long long x_bs, long long x_cs, long long y_bs, long long y_cs, int B, int C, int T,
                        void* stream);
/* Backward of svc_snake_alias_f32: dx = U^T[(D^T dy) (1 + sin(2 e^alpha u) e^alpha / (e^beta + 1e-9))], u = U x recomputed;
 * dalpha, dbeta [C] are zeroed by the call and receive the parameter gradients.  T >= 6. */
int svc_snake_alias_bwd_f32(const float* x, const float* dy, const float* alpha, const float* beta, const float* taps_host,
                            float* dx, float* dalpha, float* dbeta, long long x_bs, long long x_cs, long long g_bs,
                            long long g_cs, long long d_bs, long long d_cs, int B, int C, int T, void* stream);
/* GroupNorm with one channel per group + GELU (vencoder/hubert/hubert_model.py:76,87: norm0 = GroupNorm(512, 512) after
 * conv0): y[b,c,t] = gelu((x - mean_t) / sqrt(var_t + eps) * gamma[c] + beta[c]), statistics over the T samples of each
 * (b, c) row (biased variance, like torch).  x, y:[B,C,T] contiguous. */
int svc_channel_norm_gelu_f32(const float* x, const float* gamma, const float* beta, float* y, int B, int C, int T,
                              float eps, int apply_gelu, void* stream);
/* The same for a batch of items of DIFFERENT lengths zero-padded to T (Svc.slice_inference's chunks through the unit encoder,
 * inference/infer_tool.py:446-495 + :220-224): item b's statistics run over its own lens[b] (int32, device) steps — what it gets
 * when processed alone — and its columns t >= lens[b] are written as 0. */
int svc_channel_norm_gelu_len_f32(const float* x, const float* gamma, const float* beta, const int* lens, float* y, int B, int C,
                                  int T, float eps, int apply_gelu, void* stream);
/* Windowed-sinc polyphase resampling, the conversion in front of the unit encoder (inference/infer_tool.py:219-222:
 * torchaudio.transforms.Resample(target_sample, 16000); :271-274 for inputs at another rate).  orig/nw are the two rates
 * divided by their gcd; kern:[K, nw] is the filter bank (tap-major: kern[k*nw + j] = torchaudio's kernels[j, 0, k],
 * K = 2*width + orig), built on the host by the caller; y[b, f*nw + j] = sum_k kern[k, j] * x[b, f*orig + k - width]
 * with x read as zero outside [0, Lin).  x:[B, Lin] (row stride x_bs), y:[B, Lout] (row stride y_bs),
 * Lout <= ceil(Lin*nw/orig). */
int svc_resample_sinc_f32(const float* x, const float* kern, float* y, long long x_bs, long long y_bs, int B, int Lin,
                          int Lout, int orig, int nw, int K, int width, void* stream);
/* SinusoidalPosEmb (diffusion/wavenet.py:16-28): out[b, i] = sin(t[b] f_i), out[b, dim/2 + i] = cos(t[b] f_i),
 * f_i = exp(-i ln(10000) / (dim/2 - 1)); t:[B] float, out:[B, dim]. */
int svc_sinusoidal_emb_f32(const float* t, float* out, int B, int dim, void* stream);
/* Automatic-f0 helpers (models.py:523-527 + utils.normalize_f0, utils.py:31-45).  f0,uv,mask,lf0,norm_lf0:[B,T];
 * factor:[B] or NULL (=1, inference).  lf0 = 2595*log10(1+f0/700)/500 (or f0 itself when input_is_lf0);
 * norm_lf0 = (lf0 - mean_voiced)*factor*mask. */
int svc_f0_norm_lf0_f32(const float* f0, const float* uv, const float* mask, const float* factor, float* lf0,
                        float* norm_lf0, int B, int T, int input_is_lf0, void* stream);
/* f0 = 700*(10^(lf0*500/2595) - 1) elementwise (models.py:527). */
int svc_lf0_to_f0_f32(const float* lf0, float* f0, long long n, void* stream);
/* utils.f0_to_coarse (utils.py:69-80): f0:[n] fp32 -> coarse:[n] int64 in [0,255]. */
int svc_f0_to_coarse(const float* f0, long long* coarse, long long n, void* stream);
/* models.py:520 and :156:  x = xin + emb_uv[uv] (+ vol_w*vol + vol_b);  x_enc = (x + f0_emb[coarse(f0)]) * mask.
 * xin,x,x_enc:[B,C,T] contiguous; uv,f0,mask,vol:[B,T]; emb_uv:[2,C]; f0_emb:[256,C]; vol_w,vol_b:[C]. */
int svc_prenet_embed_f32(const float* xin, const float* uv, const float* f0, const float* emb_uv,
                         const float* f0_emb, const float* mask, const float* vol, const float* vol_w,
                         const float* vol_b, float* x, float* x_enc, int B, int C, int T, void* stream);
/* y = LayerNorm_C(x + r) * gamma + beta, optionally * mask (modules/modules.py:23-35 applied as in
 * modules/attentions.py:98,102).  x,r,y:[B,C,T] contiguous (r may be NULL), mask:[B,T] or NULL. */
int svc_add_layernorm_f32(const float* x, const float* r, const float* gamma, const float* beta,
                          const float* mask, float* y, int B, int C, int T, float eps, void* stream);
/* z = (m + noise * exp(logs) * scale) * mask with stats = [m ; logs] : [B,2C,T] (models.py:158-160,122-124). */
int svc_reparam_f32(const float* stats, const float* noise, const float* mask, float* z, int B, int C, int T,
                    float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-head self-attention with window-`w` relative-position keys/values, flash style on the fp32
 * matrix pipe (MultiHeadAttention.attention, modules/attentions.py:207-239 and helpers :259-303).
 * q,k,v,out are [B, H*dk, T] views, element (b,c,t) at ptr + b*_bs + c*_cs + t.  q is divided by
 * sqrt(dk) inside.  mask_mode: 0 none | 1 padding: score(i,j) = -1e4 where mask[b,i]*mask[b,j]==0
 * (attentions.Encoder, :96) | 2 causal: -1e4 where j > i (attentions.FFT, :52).
 * emb_rel_k / emb_rel_v: [2*window+1, dk] (heads_share=True) or NULL when window == 0.
 * ---------------------------------------------------------------------------------------------- */
typedef struct svc_attention_args {
  const float* q;
  const float* k;
  const float* v;
  const float* emb_rel_k;
  const float* emb_rel_v;
  const float* mask; /* [B,T] or NULL */
  float* out;
  long long q_bs, q_cs, k_bs, k_cs, v_bs, v_cs, o_bs, o_cs, mask_bs;
  int B, H, dk, T, window, mask_mode;
  void* ws;           /* optional scratch (device memory, caller-owned, contents undefined before and after the call) ... */
  long long ws_bytes; /* ... of at least svc_attention_ws_bytes(a): with it a short sequence with few heads (one utterance:
                         27 query tiles x 2 heads at T = 862) also splits its KEYS over workgroups and merges them in a second
                         small launch; NULL / too small: one workgroup per (query tile, head) walks all keys */
} svc_attention_args;

int svc_attention_f32(const svc_attention_args* a, void* stream);
/* Scratch bytes the key-split form wants for this shape (0: the shape does not take it).  Reads B, H, dk, T only. */
long long svc_attention_ws_bytes(const svc_attention_args* a);
/* Tuning aid (A/B on one box): force the 8- or 16-wave workgroup variant of svc_attention_f32; 0 = automatic; 100 / 101: key-split
 * form off / automatic. */
int svc_debug_set_attention_waves(int nw);


/* ------------------------------------------------------------------------------------------------
 * 16-bit inference pipeline of the NSF-HiFiGAN generator (csrc/conv1d_h.hip): the engine's form of the reference's
 * half-precision inference — inference/infer_tool.py:196-198 (`"half" in net_g_path` -> `net_g_ms.half()`), checkpoints written
 * by compress_model.py:21-48.  Activations are fp16 in HBM and LDS in the BLOCKED layout [B][C/8][T][8] (8 channels of a time step
 * = 16 contiguous bytes), weights fp16 packed once, products on v_mfma_f32_32x32x16_f16 with fp32 accumulation; bias / activation
 * / residual / accumulate arithmetic in fp32 before the one rounding of the stored result.
 *
 * svc_pack_conv1d_h: dense fp32 weight (weight norm already folded) -> [Cin/16][taps][RP][16] fp16.  u == 1: Conv1d weight
 *   [Cout][Cin][K], taps = K.  u > 1: ConvTranspose1d weight [Cin][Cout][K], rows = u*Cout (row = phase*Cout + co),
 *   taps = ceil(K/u) (vdecoder/hifigan/models.py:340-342).  RP: row count rounded up to a multiple of 128.
 * svc_conv1d_h: y = epilogue(conv(lrelu(x, pre_slope))): + bias[co], post_act (NONE | LRELU), + res (blocked fp16, y-shaped),
 *   y = (beta*y_old + v) / out_div (the MRF mean of :382-389 accumulated in place).  u > 1: transposed form, Tq = number of input
 *   positions q that reach an output, y index = q*u + phase + y_t0 (y_t0 = -padding), KS = taps, pad_left = taps - 1.
 * svc_cvt_to_h / svc_cvt_from_h: fp32 [B,C,T] (strided; optional second addend) <-> blocked fp16.
 * svc_conv_post_h: leaky_relu(pre_slope) -> Conv1d(C -> 1, KS) -> act (SVC_ACT_TANH | NONE) with fp32 arithmetic and fp32 output
 *   [B,1,T] (:390-392); w = dense fp32 [C][KS] (weight norm folded).
 * ---------------------------------------------------------------------------------------------- */
typedef struct svc_conv1d_h_args {
  const void* x;     /* fp16 [B][Cin/8][Tin][8] */
  const void* w;     /* svc_pack_conv1d_h output */
  const float* bias; /* [Cout] fp32 or NULL */
  const void* res;   /* fp16 blocked [B][Cout/8][Ty][8] or NULL */
  void* y;           /* fp16 blocked [B][Cout/8][Ty][8] */
  int B, Cin, Cout, Tin, Tq, Ty;
  int KS, dil, pad_left;
  int u, y_t0, RP;
  int post_act;
  float pre_slope, post_slope, beta, out_div;
  float acc_scale;   /* svc_conv1d_hl only: the accumulators are multiplied by this before the bias — 1 / the power-of-two scale the weight
                      * pack was built with (svc_pack_conv1d_hl); 0 is read as 1.  svc_conv1d_h ignores it.  (ABI 5) */
} svc_conv1d_h_args;
int svc_pack_conv1d_h(const float* w, void* dst, int Cout, int Cin, int K, int u, int RP, void* stream);
int svc_conv1d_h(const svc_conv1d_h_args* a, void* stream);
/* One ResBlock1 pair of the 16-bit pipeline in ONE launch (vdecoder/hifigan/models.py:60-67):
 * y = (beta*y_old + conv2(lrelu(conv1(lrelu(x)) + b1)) + b2 + x) / out_div, conv1 with dilation dil1, conv2 with dilation 1, both
 * KS taps (3 | 7 | 11) and "same" padding; x / y blocked fp16 [B][C/8][T][8], w1 / w2 from svc_pack_conv1d_h (RP rows), C a multiple
 * of 16 in 16..128.  The intermediate lives in LDS (fp16); x and y may not alias. */
int svc_resblock_pair_h(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, int B, int C, int T,
                        int KS, int dil1, int RP, float slope, float beta, float out_div, void* stream);
/* SnakeAlias (vdecoder/hifiganwithsnake/alias/act.py:125-130) on blocked fp16 tensors: y = DownSample1d(SnakeBeta(UpSample1d(x))), the
 * 2x intermediate in LDS, fp32 arithmetic; alpha / beta [C] fp32 (log scale), taps12 = the 12 kaiser-sinc taps (HOST array).
 * x and y may alias only if identical. */
int svc_snake_alias_h(const void* x, void* y, const float* alpha, const float* beta, const float* taps12, int B, int C, int T, void* stream);
int svc_debug_set_conv_h(int cfg); /* tuning aid: 0 automatic tile choice, 1 128 x 128 tiles only, 2 four column tiles per wave where they fit, 3 64 x 128 (not 64 x 64) tiles for under-filled launches */
int svc_cvt_to_h(const float* x, const float* add, void* y, long long x_bs, long long x_cs, long long add_bs, long long add_cs,
                 int B, int C, int T, void* stream);
int svc_cvt_from_h(const void* x, float* y, int B, int C, int T, void* stream);
int svc_conv_post_h(const void* x, const float* w, const float* bias, float* y, int B, int C, int T, int KS, int pad,
                    float pre_slope, int act, void* stream);

/* ---- split pipeline (csrc/conv1d_hl.hip): the same decoder convolutions (vdecoder/hifigan/models.py:41-67,340-342,378,390-392) at
 * fp32-level precision on the fp16 matrix instruction.  A fp32 value v is two fp16 values, hi = rn16(v) and lo = rn16(v - hi)
 * (hi + lo = v to 22 mantissa bits); a product is a_hi b_hi + a_hi b_lo + a_lo b_hi, three fp16 instructions with fp32
 * accumulation in place of sixteen fp32 ones.  Every tensor of the 16-bit pipeline gains a second plane behind the first:
 * activations fp16 [2][B][C/8][T][8], weight packs fp16 [2][Cin/16][taps][RP][16] (plane 0 = hi, 1 = lo; contiguous).  Entry points
 * mirror the 16-bit ones one for one and take the same argument struct (x / res / y / w point at plane 0).
 *
 * RANGE (what fp32 has and two fp16 pieces do not, and how the boundary deals with it).  A stored value s is carried to
 * max(2^-22 |s|, 2^-25): below |s| ~ 0.125 the lo piece is a subnormal fp16 and the error is absolute; above 65 504 hi overflows.
 *   - Weights: svc_pack_conv1d_hl multiplies by `scale`, a power of two the caller derives from max |w| (svc_hip.pack_conv1d_h puts
 *     max |w| at 2^14: every weight within 2^-17 of the largest keeps 22 bits, whatever the tensor's magnitude — weight-norm gains of
 *     1e-6 or 1e5 alike); the convolution multiplies its accumulators by 1 / scale (acc_scale; both exact).
 *   - Activations: the planes hold 32 v (exact; internal to the kernels — svc_cvt_to_hl / svc_cvt_from_hl are the only way in and out):
 *     v ranges over +-2047 with an absolute floor of 2^-30 (9.3e-10), i.e. 22 bits down to |v| = 0.004.  Every value these kernels
 *     PRODUCE is checked as it is encoded; |v| > 2047 (or nan) ORs 1 into the int the calling host thread registered with
 *     svc_hl_range_flag (NULL = no reporting; the pointer is read at launch time, so it is part of a captured graph).  The flag is
 *     sticky; the owner reads and clears it after the clip and re-runs the fp32 path if set (SynthesizerTrn.split_range_exceeded,
 *     Svc.infer_units).  tests/test_split_gpu.py measures all three: weight magnitudes 1e-6..1e5, the flag, the small-value floor. */
int svc_hl_range_flag(int* flag);
int svc_pack_conv1d_hl(const float* w, void* dst, int Cout, int Cin, int K, int u, int RP, float scale, void* stream);
int svc_conv1d_hl(const svc_conv1d_h_args* a, void* stream);
/* svc_resblock_pair_h on the split planes: C a multiple of 16 in 16..128; acc_scale1 / 2 = 1 / the scales of the two packs. */
int svc_resblock_pair_hl(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, int B, int C, int T,
                         int KS, int dil1, int RP, float slope, float beta, float out_div, float acc_scale1, float acc_scale2,
                         void* stream);
/* svc_snake_alias_h on the split planes (x and y [2][B][C/8][T][8]; may not alias). */
int svc_snake_alias_hl(const void* x, void* y, const float* alpha, const float* beta, const float* taps12, int B, int C, int T, void* stream);
int svc_debug_set_conv_hl(int cfg); /* tuning aid: bit 0 = 64 x 128 (not 64 x 64) tiles for under-filled launches */
int svc_cvt_to_hl(const float* x, const float* add, void* y, long long x_bs, long long x_cs, long long add_bs, long long add_cs,
                  int B, int C, int T, void* stream);
int svc_cvt_from_hl(const void* x, float* y, int B, int C, int T, void* stream);
int svc_conv_post_hl(const void* x, const float* w, const float* bias, float* y, int B, int C, int T, int KS, int pad,
                     float pre_slope, int act, void* stream);

/* ---- fused coupling layer of the flow for the 16-bit and the split inference modes (csrc/flow_fused.hip): ONE launch for
 * modules/modules.py:288-307 ResidualCouplingLayer.forward with mean_only — pre 1x1 (:291), WN.forward :110-138 (per layer: k = 5 conv
 * hidden -> 2 hidden, + g_l, tanh * sigmoid gate, 1x1 res/skip, x = (x + res) * mask, output += skip), post 1x1 (:297), x1 <- m + x1 * mask /
 * (x1 - m) * mask (:300-306) — on the flow's fp32 working buffer IN PLACE, the pending channel Flip (:232-239, models.py:45-52) folded into
 * x_cs < 0.  The fp32 path runs these as 10 launches per coupling.  planes = 1: fp16 arithmetic of the reference's `.half()` mode
 * (fp16 operands, fp32 accumulation, h / activations stored as fp16 in LDS); planes = 2: the split pipeline's hi / lo planes (fp32-level,
 * range-checked into the svc_hl_range_flag word).  Weights: packs of svc_pack_conv1d_h (planes 1) / svc_pack_conv1d_hl (planes 2) of the
 * weight-norm-folded dense weights in their natural row order: w_pre [hidden, channels/2, 1], w_in[l] [2 hidden, hidden, 5],
 * w_rs[l] [2 hidden (last layer: hidden), hidden, 1], w_post [channels/2, hidden, 1]; s_* = their acc scales (planes 2; 0 reads as 1).
 * cond: cond_layer(g), [B][2 hidden n_layers][1 | T] through (cond_bs, cond_cs, cond_ts), or NULL.  Built for channels 192, hidden 192,
 * kernel size 5, dilation rate 1, 1..6 layers (both templates' flows); anything else: SVC_ERR_BAD_ARG, callers keep the unfused launches. */
#define SVC_COUPLING_MAX_LAYERS 8
typedef struct svc_coupling_args {
  float* x;                 /* channel 0 of the VIEW of the [B][channels][T] buffer; view channel c at x + c * x_cs */
  long long x_bs, x_cs;
  const float* mask;        /* [B][T] or NULL */
  const float* cond;
  long long cond_bs, cond_cs;
  int cond_ts;
  const void* w_pre;
  const float* b_pre;
  const void* w_in[SVC_COUPLING_MAX_LAYERS];
  const float* b_in[SVC_COUPLING_MAX_LAYERS];
  const void* w_rs[SVC_COUPLING_MAX_LAYERS];
  const float* b_rs[SVC_COUPLING_MAX_LAYERS];
  const void* w_post;
  const float* b_post;
  float s_pre, s_in[SVC_COUPLING_MAX_LAYERS], s_rs[SVC_COUPLING_MAX_LAYERS], s_post;
  int B, T, channels, hidden, kernel_size, n_layers, reverse, planes;
} svc_coupling_args;
int svc_coupling_fused_h(const svc_coupling_args* a, void* stream);
int svc_debug_set_coupling_fused(int prefetch); /* A/B: 0 = without the L2 prefetch workgroups */


/* ================================================================================================
 * TRAINING path.  Backward of the convolutions above plus the small ops of the
 * GAN step (train.py:150-213).  Same conventions; every gradient is fp32.
 * ============================================================================================== */

/* torch.nn.utils.weight_norm (dim=0) made explicit: w[r,:] = g[r] * v[r,:] / ||v[r,:]||;  norm[r] = ||v[r,:]||
 * (rows = Cout for Conv1d, Cin for ConvTranspose1d; cols = the remaining extent).  bwd: (dv, dg) from dw. */
int svc_weight_norm_fwd_f32(const float* v, const float* g, float* w, float* norm, int rows, int cols, void* stream);
int svc_weight_norm_bwd_f32(const float* v, const float* g, const float* norm, const float* dw, float* dv, float* dg,
                            int rows, int cols, void* stream);
/* dgrad weight of a Conv1d: w:[Cout][Cin][KS] -> dst:[Cout][KS][CinP], dst[co][KS-1-k][ci] = w[co][ci][k]; feeding it
 * to svc_conv1d_f32 with x = dy, pad_left = dil*(KS-1) - pad computes dx (autograd of F.conv1d). */
int svc_pack_conv1d_weight_T(const float* w, float* dst, int Cout, int Cin, int KS, int CinP, void* stream);
/* One-launch weight preparation of a training convolution, and its adjoint.  Every nn.Conv1d / Conv2d((k,1),(s,1)) /
 * ConvTranspose1d of the training graph (models.py:165-227, modules/*, vdecoder/hifigan/models.py:335-355) runs as a dense
 * stride-1 convolution whose weight wd[Od][Id][Kd] is an index map of the parameter v [R][C2][K], scaled per row by the
 * weight-norm factor g[r]/||v[r]|| when g != NULL (torch.nn.utils.weight_norm, dim 0):
 *   kind 0 dense        wd[r][c][k] = w[r][c][k]                                           (Od=R, Id=C2, Kd=K)
 *   kind 1 strided      wd[r][q*C2 + c][m] = w[r][c][k],  k + shift = s*m + q              (Od=R, Id=s*C2)
 *   kind 2 transposed   wd[ph*C2 + c][r][Kd-1-mm] = w[r][c][k],  k = ph + mm*s  (v = [Cin][Cout][K], s = stride, Kd = ceil(K/s))
 * prep writes wp[(i*Kd+m)*OdP + o] (the layout svc_conv1d_f32 reads) and, when wt != NULL, the dgrad operand
 * wt[(o*Kd + Kd-1-m)*IdP + i], and norm[r] = ||v[r]||.  Entries that no (r,c,k) maps to are NOT written: wp / wt must be
 * zero-filled once by the caller.  grad maps dwd [Od][Id][Kd] (svc_conv1d_wgrad_f32's output) back to dv [R][C2][K] and
 * dg [R] (dv = dw without g). */
typedef struct {
  const float* v;
  const float* g;
  float* wp;
  float* wt;
  float* norm;
  int kind, R, C2, K, Od, Id, Kd, OdP, IdP, s, shift;
} svc_conv_weight_args;
int svc_conv_weight_prep_f32(const svc_conv_weight_args* args, void* stream);
/* svc_conv_weight_prep_f32 for n_plans convolutions together: one launch for the row norms of the weight-normed plans, one for
 * all operand packs.  `host_args` (validated here) and `dev_args` hold the same n_plans structs, the latter in device memory;
 * `dev_row_start` / `dev_block_start` [n_plans] (device) = exclusive prefix sums of the plans' R resp. of
 * svc_conv_weight_prep_blocks(R, C2, K) (a scatter workgroup owns SVC_WEIGHT_PREP_ROWS rows x ~SVC_WEIGHT_PREP_COLS elements). */
#define SVC_WEIGHT_PREP_ROWS 32
#ifndef SVC_WEIGHT_PREP_COLS   /* (-D override: tuning builds only; the library reports its own value through svc_conv_weight_prep_blocks) */
#define SVC_WEIGHT_PREP_COLS 1024
#endif
int svc_conv_weight_prep_blocks(int R, int C2, int K);
int svc_conv_weight_prep_multi_f32(const svc_conv_weight_args* host_args, const svc_conv_weight_args* dev_args,
                                   const int* dev_row_start, const int* dev_block_start, int n_plans, void* stream);
int svc_conv_weight_grad_f32(const svc_conv_weight_args* args, const float* dwd, float* dv, float* dg, void* stream);

/* Weight gradient (and any "correlate two [B,C,T] signals over time" product):
 *   G[ca,cb,k] (+)= sum_{b,t} A[b,ca,t] * Bm[b,cb,t + k*dil - pad],  t in [0,TA), Bm index in [0,TB), KS <= 16.
 * G is [Ca][Cb][KS] contiguous (= nn.Conv1d.weight layout for A = dy, Bm = x). */
typedef struct svc_wgrad_args {
  const float* A;
  const float* Bm;
  float* G;
  long long a_bs, a_cs, b_bs, b_cs;
  int B, Ca, Cb, TA, TB, KS, dil, pad, accumulate;
  float* dbias; /* optional [Ca]: also produces the bias gradient sum_{b,t} A[b,ca,t] (zeroed by the call unless
                   `accumulate`), from the A tiles already staged in LDS — saves a separate reduction pass over dy */
  int mma;      /* SVC_MMA_F32 / SVC_MMA_BF16 (see svc_conv1d_args.mma): bf16 operands for the 128 x 64 tile kernel, fp32
                   accumulation and bias gradient; the small-channel kernel (Ca, Cb <= 32) always runs fp32 */
} svc_wgrad_args;
int svc_conv1d_wgrad_f32(const svc_wgrad_args* a, void* stream);

/* Batched strided fp32 GEMM on the matrix pipe: C[b,m,n] = alpha*sum_k A[b,m,k]*B[b,k,n] + beta*C[b,m,n]
 * (training-time attention products modules/attentions.py:207-239 and their gradients, mel filterbank
 * modules/mel_processing.py:67-76, Linear layers). */
typedef struct svc_gemm_args {
  const float* A;
  const float* B;
  float* C;
  long long a_bs, a_ms, a_ks, b_bs, b_ks, b_ns, c_bs, c_ms, c_ns;
  int batch, M, N, K;
  float alpha, beta;
  int split_k_atomic; /* 1: the caller accepts a reduction split over workgroups and combined with fp32 atomics (thin-M products,
                         M <= 16 and K >= 256: ~10x faster, sum order and thus the last bits vary run to run).  Set by the
                         training backward of the relative-position embeddings only; 0 (everything else, all of inference): every
                         element is one deterministic accumulation chain */
} svc_gemm_args;
int svc_gemm_f32(const svc_gemm_args* a, void* stream);

/* Reductions of [B,C,T]: mode 0 out[c] = sum_{b,t} (bias gradients), mode 1 out[b,c] = sum_t (gradient of a [B,C,1]
 * broadcast).  out = sum + beta*out.   svc_reduce_c: out[b,t] = sum_c x[b,c,t]*(w?w[c]:1). */
int svc_reduce_bct_f32(const float* x, float* out, long long x_bs, long long x_cs, int B, int C, int T, int mode,
                       float beta, void* stream);
int svc_reduce_c_f32(const float* x, const float* w, float* out, int B, int C, int T, void* stream);

/* Element-wise y = op(a, b) over n contiguous floats (b may be NULL for unary ops). */
enum {
  SVC_EW_ADD = 0,        /* alpha*a + beta*b */
  SVC_EW_MUL = 1,        /* alpha*a*b */
  SVC_EW_LRELU = 2,      /* leaky_relu(a, slope=alpha) */
  SVC_EW_LRELU_BWD = 3,  /* a=dy, b=x */
  SVC_EW_TANH = 4,
  SVC_EW_TANH_BWD = 5,   /* a=dy, b=y */
  SVC_EW_RELU = 6,
  SVC_EW_RELU_BWD = 7,   /* a=dy, b=x */
  SVC_EW_EXP = 8,        /* exp(alpha*a) */
  SVC_EW_LOG_CLAMP = 9,  /* log(max(a, alpha)) */
  SVC_EW_LOG_CLAMP_BWD = 10, /* a=dy, b=x */
  SVC_EW_SCALE = 11,     /* alpha*a + beta */
  SVC_EW_SIGMOID = 12,
  SVC_EW_SQUARE = 13,    /* alpha*a*a */
  SVC_EW_SIGN_MUL = 14,  /* alpha*sign(a) */
  SVC_EW_DIV = 15,       /* alpha*a/b */
  SVC_EW_GELU = 16,      /* 0.5 a (1 + erf(a / sqrt 2))  (vencoder/hubert/hubert_model.py:87-93,127) */
  SVC_EW_MISH = 17,      /* a tanh(softplus(a))  (diffusion/wavenet.py:76) */
  SVC_EW_CLAMP = 18,     /* min(max(a, alpha), beta)  (diffusion/diffusion.py:139) */
  SVC_EW_MISH_BWD = 19,  /* a = dy, b = x: d/dx [x tanh(softplus x)] (backward of diffusion/wavenet.py:76, train_diff.py) */
  SVC_EW_DROPOUT = 20    /* b = uniform [0,1) draws: a * (b >= alpha ? 1/(1-alpha) : 0) — nn.Dropout(alpha) forward, and its
                            backward with a = dy (modules/attentions.py:51,55,100,104,344) */
};
int svc_ew_f32(int op, const float* a, const float* b, float* y, long long n, float alpha, float beta, void* stream);
/* y = leaky_relu'(x; slope) * dy + r over n contiguous floats: the leaky-ReLU backward and the accumulation of the residual branch's
 * gradient in one launch (svc_autograd._LReluRes: the input of a ResBlock pair feeds lrelu -> conv AND the residual add). */
int svc_lrelu_bwd_add_f32(const float* dy, const float* x, const float* r, float* y, long long n, float slope, void* stream);
/* y[b,c,t] = op(x[b,c,t], side[b*s_bs + c*s_cs + t*s_ts]) — masks ([B,1,T]: s_cs = 0), speaker conditions ([B,C,1]:
 * s_ts = 0), channel-flipped operands (negative strides). */
int svc_ew_bct_f32(int op, const float* x, const float* side, float* y, long long x_bs, long long x_cs, long long s_bs,
                   long long s_cs, long long s_ts, long long y_bs, long long y_cs, int B, int C, int T, float alpha,
                   float beta, void* stream);
/* commons.fused_add_tanh_sigmoid_multiply (modules/commons.py:129-136) and its gradient: in:[B,2H,T] -> acts:[B,H,T]. */
int svc_gate_fwd_f32(const float* in, float* acts, int B, int H, int T, void* stream);
int svc_gate_bwd_f32(const float* in, const float* dacts, float* din, int B, int H, int T, void* stream);

/* Phase decimation over blocks of `w` samples: y[b, r*C + c, q*w + j] = xpad[b, c, (q*s + r + off)*w + j], r < s,
 * j < w (w = 1: plain samples; w = period: DiscriminatorP's [B,C,T/p,p] view, models.py:190, kept time-contiguous so
 * that its Conv2d((k,1),(s,1)) stack becomes dense dilation-p Conv1d's).  xpad is x reflect-padded on the right to `lp`
 * samples when lp > T (F.pad(..., "reflect"), models.py:185-189) and zero elsewhere.  A stride-s conv becomes a dense
 * conv over s*C channels; the adjoint is svc_decimate_bwd_f32.  y:[B, s*C, Q*w]. */
int svc_decimate_f32(const float* x, float* y, int B, int C, int T, int s, int w, int off, int Q, int lp, void* stream);
int svc_decimate_bwd_f32(const float* dy, float* dx, int B, int C, int T, int s, int w, int off, int Q, int lp,
                         void* stream);

/* HuBERT / ContentVec positional convolution (vencoder/hubert/hubert_model.py:116-129, fairseq `pos_conv`):
 *   y = x + gelu(conv1d(x, w, bias, kernel KS, padding pad, groups)[..., :T])      x, y: [B, C, T], 48 channels per group.
 * svc_posconv_pack_f32 folds weight_norm(dim=2) (g: [KS] or NULL) and packs v:[C][48][KS] to [groups][48][KS][64] (the two
 * 32-row MFMA tiles of a group read full rows; rows 48..63 are zero).  One MFMA kernel, x tile staged in LDS, weights streamed
 * from L2, bias + exact GELU + residual in the epilogue. */
int svc_posconv_pack_f32(const float* v, const float* g, float* dst, int C, int KS, int groups, void* stream);
int svc_posconv_f32(const float* x, const float* w, const float* bias, float* y, int B, int C, int T, int KS, int pad,
                    int groups, void* stream);

/* Grouped strided Conv1d (DiscriminatorS, models.py:206-211): w:[Cout][Cin/groups][KS]. */
int svc_gconv1d_fwd_f32(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int Tin,
                        int Tout, int KS, int stride, int pad, int groups, void* stream);
int svc_gconv1d_dgrad_f32(const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int Tin, int Tout, int KS,
                          int stride, int pad, int groups, void* stream);
int svc_gconv1d_wgrad_f32(const float* dy, const float* x, float* dw, int B, int Cin, int Cout, int Tin, int Tout, int KS,
                          int stride, int pad, int groups, void* stream);

/* Scalar loss reductions (modules/losses.py:4-58; train.py:202,206): *out += scale * sum_i f(...) in double. */
enum {
  SVC_RED_SUM = 0, SVC_RED_ABS_DIFF = 1, SVC_RED_SQ_DIFF = 2, SVC_RED_SQ_ONE_MINUS = 3, SVC_RED_SQ = 4, SVC_RED_KL = 5
};
int svc_reduce_scalar_f64(int op, const float* a, const float* b, const float* c, const float* d, long long n,
                          double* out, double scale, void* stream);
int svc_f64_to_f32(const double* in, float* out, int n, void* stream);
/* Guard of a hipGraph-replayed training iteration (no host sync per step): counter[2] += 1, counter[0] += number of non-finite
 * values among the n (<= 8) device scalars, counter[1] = the (1-based) launch number when any was seen.  `scalars`: HOST array of n
 * device pointers (read at call / capture time); counter: int[3] in device memory, zeroed by the caller.  The counters are sticky;
 * the caller reads them back every N steps (train.TrainStep). */
int svc_nonfinite_guard_f32(const float* const* scalars, int n, int* counter, void* stream);

/* Fused AdamW step over one flat parameter buffer (train.py:79-88; torch.optim.AdamW semantics).  The hyper-parameters
 * are read from DEVICE memory: hyper = float[7] {lr, beta1, beta2, eps, weight_decay, step (>= 1), grad_scale}
 * (grad_scale multiplies the gradient first: GradScaler's 1/scale), so a hipGraph that captured the step stays valid
 * across iterations; svc_adamw_advance does hyper[5] += 1 (call it before the step of each iteration). */
int svc_adamw_f32(float* p, const float* g, float* m, float* v, long long n, const float* hyper, void* stream);
int svc_adamw_advance(float* hyper, void* stream);


/* rocFFT-backed STFT magnitude (modules/mel_processing.py:40-64: torch.stft(..., onesided) -> sqrt(re^2+im^2+1e-6)).
 * A plan is an opaque handle for batched length-n real transforms (n even); *work_bytes is the size of the caller-owned
 * work buffer to pass to every execute (may be 0).  Forward: x:[batch][n] -> z:[batch][n/2+1][2] interleaved complex,
 * unnormalised.  Inverse: the unnormalised complex-to-real transform (it may overwrite gz); with the interior bins of
 * the spectrum gradient halved (svc_cmag_c_bwd_f32 does that) it is the exact adjoint of the forward.  Execution only
 * enqueues kernels on `stream`. */
int svc_rfft_plan_create(int n, int batch, void** plan_out, long long* work_bytes);
int svc_rfft_plan_destroy(void* plan);
int svc_rfft_forward_f32(void* plan, const float* x, float* z, void* work, void* stream);
int svc_rfft_inverse_f32(void* plan, float* gz, float* gx, void* work, void* stream);
/* mag = sqrt(re^2 + im^2 + eps) over n interleaved complex values; backward writes the C2R-ready gradient
 * gz = dmag/mag * z (interior bins x 1/2, imaginary parts of bins 0 and bins-1 zeroed). */
int svc_cmag_c_f32(const float* z, float* mag, long long n, float eps, void* stream);
int svc_cmag_c_bwd_f32(const float* z, const float* mag, const float* dmag, float* gz, long long n, int bins, void* stream);

/* Channel LayerNorm for training (modules/modules.py:23-35): forward also returns the per-column mean / rstd [B,T];
 * backward returns dx and the gamma / beta gradients. */
int svc_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int B,
                          int C, int T, float eps, void* stream);
int svc_layernorm_bwd_f32(const float* x, const float* gamma, const float* dy, const float* mean, const float* rstd,
                          float* dx, float* dgamma, float* dbeta, int B, int C, int T, void* stream);
/* Training-time attention pieces around svc_gemm_f32 (modules/attentions.py:207-303).  S:[B*H,T,T] scores (in place ->
 * probabilities): S[i,j] += rel[i, j-i+window] inside the band, masked to -1e4 (mask_mode 1: mask[b,i]*mask[b,j]==0,
 * 2: j>i), softmax over j.  Attention-probability dropout (modules/attentions.py:232) is fused in: with drop_u != NULL
 * (uniform [0,1) draws, [B*H,T,T]) S keeps the probabilities and Pd receives P * (u >= p_drop ? 1/(1-p_drop) : 0).
 * bwd: dP -> dS in place (with drop_u, dP is the gradient w.r.t. Pd and is masked first); dS is zero at the masked score
 * positions (masked_fill passes no gradient; same mask / mask_mode as the forward).  band gather/scatter move the (2*window+1)-wide diagonal band
 * between a [rows,T] matrix (row i of every T x T block) and a [rows, 2*window+1] array. */
int svc_attn_softmax_fwd_f32(float* S, const float* rel, const float* mask, int B, int H, int T, int window, int mask_mode,
                             const float* drop_u, float p_drop, float* Pd, void* stream);
int svc_attn_softmax_bwd_f32(const float* P, float* dP, int B, int H, int T, const float* drop_u, float p_drop,
                             const float* mask, int mask_mode, void* stream);
/* The same with the keep decisions made from a counter-based uniform draw u(seed[0], site, element) instead of a tensor of draws
 * (production path: no [B,H,T,T] torch.rand tensor per attention layer; the forward and the backward of a site pass the same seed
 * pointer and site number).  seed: int64[1] in device memory — a replayed hipGraph sees the value its own iteration left there.
 * svc_dropout_rng_f32: y = x * (u >= p ? 1/(1-p) : 0): nn.Dropout(p) on the activation sites (modules/attentions.py:51,100,344),
 * forward on x and backward on dy. */
int svc_attn_softmax_fwd_rng_f32(float* S, const float* rel, const float* mask, int B, int H, int T, int window, int mask_mode,
                                 const long long* seed, int site, float p_drop, float* Pd, void* stream);
int svc_attn_softmax_bwd_rng_f32(const float* P, float* dP, int B, int H, int T, const long long* seed, int site, float p_drop,
                                 const float* mask, int mask_mode, void* stream);
int svc_dropout_rng_f32(const float* x, float* y, long long n, const long long* seed, int site, float p, void* stream);
/* Leaky ReLU over rows with a padded tail (DiscriminatorP's feature maps, models.py:190-193: F.leaky_relu(l(x), 0.1) on the
 * [B,C,H,p] maps, kept here as [rows, P] with P % 4 == 0 and the first L columns meaningful): y = lrelu(x) for t < L, 0 for
 * L <= t < P; bwd: dx = dy * (y > 0 ? 1 : slope) for t < L, 0 on the tail.  slope = 1: tail mask only. */
int svc_lrelu_tail_fwd_f32(const float* x, float* y, long long rows, int P, int L, float slope, void* stream);
int svc_lrelu_tail_bwd_f32(const float* y, const float* dy, float* dx, long long rows, int P, int L, float slope,
                           void* stream);
/* torch.nn.utils.spectral_norm of a conv weight (models.py:170,205 with use_spectral_norm=True; one power iteration):
 * W [R][K] = weight_orig viewed as [Cout, rest], u [R], v [K] = the module's weight_u / weight_v buffers.  fwd with
 * power_iteration != 0 (training mode) updates v <- normalize(W^T u), u <- normalize(W v) IN PLACE, then (both modes)
 * sigma[0] = u . (W v) and w = W / sigma; tmp: R floats of workspace.  bwd (u, v constants, as in torch):
 * dW = g / sigma - (sum g*W) / sigma^2 * u v^T; dot_ws: one double of workspace. */
int svc_spectral_norm_fwd_f32(const float* W, float* u, float* v, float* w, float* sigma, float* tmp, int R, int K,
                              int power_iteration, float eps, void* stream);
int svc_spectral_norm_bwd_f32(const float* W, const float* u, const float* v, const float* sigma, const float* g, float* dW,
                              double* dot_ws, int R, int K, void* stream);
int svc_band_gather_f32(const float* M, float* band, long long n_rows, int T, int window, void* stream);
int svc_band_scatter_add_f32(float* M, const float* band, long long n_rows, int T, int window, void* stream);
/* Embedding lookups in channel-major form, y[b,c,t] = W[idx[b,t], c] (models.py:393,453,136) and the scatter-add of
 * their gradient into a zero-initialised dW. */
int svc_embed_fwd_f32(const long long* idx, const float* W, float* y, int B, int C, int T, void* stream);
int svc_embed_bwd_f32(const long long* idx, const float* dy, float* dW, int B, int C, int T, int n_rows, void* stream);
/* Gradient of svc_reparam_f32: dstats = [dm ; dlogs]. */
int svc_reparam_bwd_f32(const float* stats, const float* noise, const float* mask, const float* dz, float* dstats, int B,
                        int C, int T, float scale, void* stream);
/* The source module of vdecoder/nsf_hifigan (SineGen.forward, vdecoder/nsf_hifigan/models.py:136-181, + SourceModuleHnNSF
 * :216-218): same interface as svc_nsf_source_f32 but that generator integrates the phase in double at the sample rate
 * (rand_ini added to every sample of frame 0), so the sine is sin(2 pi frac(prefix)) evaluated in double.
 * scratch: B*H*T doubles, 8-byte aligned. */
int svc_nsf_source_exact_f32(const float* f0, const float* rand_ini, const float* noise, const float* lin_w,
                             const float* lin_b, float* har, void* scratch, int B, int T, int upp, int H,
                             float sampling_rate, float sine_amp, float noise_std, void* stream);
/* Training variant of svc_nsf_source_f32 that also stores the per-harmonic waves [B,T*upp,H]; l_linear + tanh on stored
 * waves and the (dw, db) gradients of l_linear (vdecoder/hifigan/models.py:318). */
int svc_nsf_source_train_f32(const float* f0, const float* rand_ini, const float* noise, const float* lin_w,
                             const float* lin_b, float* har, float* waves, void* scratch, int B, int T, int upp, int H,
                             float sampling_rate, float sine_amp, float noise_std, void* stream);
int svc_nsf_linear_fwd_f32(const float* waves, const float* w, const float* b0, float* har, long long n, int H, void* stream);
int svc_nsf_linear_bwd_f32(const float* waves, const float* har, const float* dhar, float* dw, float* db, long long n, int H,
                           void* stream);


/* Masked KL term (modules/losses.py:43-58): acc2[0] += sum(kl*mask), acc2[1] += sum(mask); backward of sum(kl*mask)
 * scaled by the device scalar *g. */
int svc_kl_fwd_f64(const float* z_p, const float* logs_q, const float* m_p, const float* logs_p, const float* mask,
                   double* acc2, int B, int C, int T, void* stream);
int svc_kl_bwd_f32(const float* z_p, const float* m_p, const float* logs_p, const float* mask, const float* g, float* dz_p,
                   float* dlogs_q, float* dm_p, float* dlogs_p, int B, int C, int T, void* stream);
/* STFT front end of the mel loss (modules/mel_processing.py:40-64): reflect-pad by `pad`, frame (hop), window ->
 * frames [B,NF,nfft]; adjoint (overlap-add with reflection folding) into dy [B,L].  The transform itself is a real DFT
 * as two svc_gemm_f32 products against the basis from svc_dft_basis_f32 (cs = cos, sn = -sin, [nfft][NB]);
 * |.| with the reference's 1e-6 floor and its gradient. */
int svc_stft_frame_f32(const float* y, const float* win, float* frames, int B, int L, int NF, int nfft, int hop, int pad,
                       void* stream);
int svc_stft_frame_bwd_f32(const float* dframes, const float* win, float* dy, int B, int L, int NF, int nfft, int hop, int pad,
                           void* stream);
int svc_dft_basis_f32(float* cs, float* sn, int N, int NB, void* stream);
int svc_cmag_f32(const float* re, const float* im, float* mag, long long n, float eps, void* stream);
int svc_cmag_bwd_f32(const float* re, const float* im, const float* mag, const float* dmag, float* dre, float* dim, long long n,
                     void* stream);

/* Tuning / debugging knob of svc_conv1d_f32 (tile-config override and ablation switches); 0 restores defaults. */
int svc_debug_set_conv_cfg(int cfg);
/* svc_conv1d_f32 routes long dense convolutions (B*T covered by one round of 224-column strips: the decoder's MRF ResBlock
 * convs, vdecoder/hifigan/models.py:41-67) to the one-workgroup-per-CU strip kernel (csrc/conv1d_strip.hip).
 * mode 0: never, 1: automatic (default), 2..5: force strip arrangement 0..3 (32x32 4x1 / 2x2 / 1x4, 16x16 4x1) for eligible shapes,
 * mode + 10: the same with one wave per strip instead of two (A/B of the first form of the kernel);
 * a negative mode changes nothing and returns the number of launches that have taken the strip kernel so far (tests). */
int svc_debug_set_conv_strip(int mode);
/* Tuning aid: 1 selects the first (one thread per output) grouped-conv kernels, 2 the LDS-tiled ones (default). */
int svc_debug_set_gconv_version(int version);
/* Tuning aid: number of workgroups svc_conv1d_wgrad_f32 splits a 3..5-tap launch into over time (default 256 = one per CU);
 * a negative value sets the small-channel kernel's target (default 256) to its magnitude. */
int svc_debug_set_wgrad_target(int workgroups);

#ifdef __cplusplus
}
#endif
#endif /* SVC_HIP_H */
