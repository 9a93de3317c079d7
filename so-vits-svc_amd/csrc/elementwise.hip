// elementwise.hip — the small HBM-bound pieces between the convolutions of SynthesizerTrn.infer/forward.
#include "common.h"

namespace {

// ---- f0_to_coarse (utils.py:69-80) ------------------------------------------------------------------------
// fp32 arithmetic in the reference's order.  torch.round is round-half-to-even -> rintf.  The one transcendental, log(1 + f0/700),
// is taken in fp64 and rounded ONCE to fp32 (= the correctly rounded fp32 logarithm): the integer result must equal the
// reference's bit for bit, and torch's CPU logf agrees with the correctly rounded value on every input tried (0 mismatching bins
// in 2 M random f0; ocml's 1-ulp logf flips a rounding boundary on ~2 per million).
__device__ __forceinline__ long long f0_to_coarse_dev(float f0) {
  // a and b are Python doubles in the reference, applied to a float tensor (=> cast to fp32 per op)
  const double mel_min_d = 1127.0 * log(1.0 + 50.0 / 700.0);
  const double mel_max_d = 1127.0 * log(1.0 + 1100.0 / 700.0);
  const double a_d = (256 - 2) / (mel_max_d - mel_min_d);
  const double b_d = mel_min_d * a_d - 1.0;
  const float x = 1.f + f0 / 700.f;
  float mel = 1127.f * (float)log((double)x);
  if (mel > 0.f) mel = mel * (float)a_d - (float)b_d;
  long long c = (long long)rintf(mel);
  if (c <= 0) c = 0;
  if (c < 1) c = 1;
  if (c >= 256) c = 0;  // reference quirk: zeroed before the ">= f0_bin -> 255" fix-up can see it (utils.py:77-79)
  return c;
}

__global__ void f0_to_coarse_kernel(const float* __restrict__ f0, long long* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = f0_to_coarse_dev(f0[i]);
}

// ---- pre-net embedding add (models.py:520 + models.py:156) -----------------------------------------------
//   x     = xin + emb_uv[uv] + (vol ? vol_w*vol + vol_b : 0)          (xin = pre(c) * mask, from the conv epilogue)
//   x_enc = (x + f0_emb[f0_to_coarse(f0)]) * mask
__global__ __launch_bounds__(256) void prenet_embed_kernel(const float* __restrict__ xin, const float* __restrict__ uv,
                                                           const float* __restrict__ f0, const float* __restrict__ emb_uv,
                                                           const float* __restrict__ f0_emb, const float* __restrict__ mask,
                                                           const float* __restrict__ vol, const float* __restrict__ vol_w,
                                                           const float* __restrict__ vol_b, float* __restrict__ x,
                                                           float* __restrict__ x_enc, int C, int T) {
  const int t = blockIdx.x * 64 + (threadIdx.x & 63);
  const int b = blockIdx.z;
  if (t >= T) return;
  const int uvi = uv[(long long)b * T + t] != 0.f ? 1 : 0;
  const long long coarse = f0_to_coarse_dev(f0[(long long)b * T + t]);
  const float mk = mask ? mask[(long long)b * T + t] : 1.f;
  const float vv = vol ? vol[(long long)b * T + t] : 0.f;
  for (int c = blockIdx.y * 4 + (threadIdx.x >> 6); c < C; c += gridDim.y * 4) {
    const long long o = ((long long)b * C + c) * T + t;
    float v = xin[o] + emb_uv[uvi * C + c];
    if (vol) v += vol_w[c] * vv + vol_b[c];
    x[o] = v;
    x_enc[o] = (v + f0_emb[coarse * C + c]) * mk;
  }
}

// ---- residual + LayerNorm over channels of [B,C,T] (modules/modules.py:23-35; attentions.py:98,102) --------
// LN_TT time steps x LN_CG channel groups per 256-thread block: 32 x 8 for long sequences (full 128 B segments per
// channel row), 4 x 64 when T / 32 would leave most CUs without a block (T = 500..862 frames).
template <int LN_TT, int LN_CG>
__global__ __launch_bounds__(LN_TT* LN_CG) void add_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta,
                                                                    const float* __restrict__ mask, float* __restrict__ y,
                                                                    int C, int T, float eps) {
  __shared__ float red[LN_CG][LN_TT];
  const int tl = threadIdx.x % LN_TT;
  const int cg = threadIdx.x / LN_TT;
  const int t = blockIdx.x * LN_TT + tl;
  const int b = blockIdx.y;
  const bool ok = t < T;
  const long long base = (long long)b * C * T + t;
  float s = 0.f;
  if (ok)
    for (int c = cg; c < C; c += LN_CG) s += x[base + (long long)c * T] + (r ? r[base + (long long)c * T] : 0.f);
  red[cg][tl] = s;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int i = 0; i < LN_CG; ++i) mean += red[i][tl];
  mean /= (float)C;
  __syncthreads();
  float q = 0.f;
  if (ok)
    for (int c = cg; c < C; c += LN_CG) {
      const float v = x[base + (long long)c * T] + (r ? r[base + (long long)c * T] : 0.f) - mean;
      q += v * v;
    }
  red[cg][tl] = q;
  __syncthreads();
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < LN_CG; ++i) var += red[i][tl];
  var /= (float)C;
  const float rstd = 1.f / sqrtf(var + eps);
  if (!ok) return;
  const float mk = mask ? mask[(long long)b * T + t] : 1.f;
  for (int c = cg; c < C; c += LN_CG) {
    const float v = x[base + (long long)c * T] + (r ? r[base + (long long)c * T] : 0.f);
    y[base + (long long)c * T] = ((v - mean) * rstd * gamma[c] + beta[c]) * mk;
  }
}

// Short sequences (T = 500..862 frames: the unit encoder's 26 and the prior encoder's 12 launches per clip): ONE pass over memory.
// 4 time steps x 64 channel groups per block (the block count of the <4, 64> form above: at T = 500 a coarser split leaves
// most CUs without a block — 8 x 32 measured 21 us per call against 16); a thread keeps its NV = C / 64 values of x + r in
// registers between the mean, the variance and the output pass instead of re-reading both tensors from L2 for each of them.
template <int NV>
__global__ __launch_bounds__(256) void add_layernorm_reg_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ mask, float* __restrict__ y, int C,
                                                                int T, float eps) {
  constexpr int TT = 4, CG = 64;
  __shared__ float red[CG][TT];
  const int tl = threadIdx.x % TT, cg = threadIdx.x / TT;
  const int t = blockIdx.x * TT + tl, b = blockIdx.y;
  const bool ok = t < T;
  const long long base = (long long)b * C * T + min(t, T - 1);
  float v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = cg + i * CG;
    v[i] = c < C ? x[base + (long long)c * T] + (r ? r[base + (long long)c * T] : 0.f) : 0.f;
    s += v[i];
  }
  red[cg][tl] = s;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int i = 0; i < CG; ++i) mean += red[i][tl];
  mean /= (float)C;
  __syncthreads();
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float d = cg + i * CG < C ? v[i] - mean : 0.f;
    q += d * d;
  }
  red[cg][tl] = q;
  __syncthreads();
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < CG; ++i) var += red[i][tl];
  var /= (float)C;
  const float rstd = 1.f / sqrtf(var + eps);
  if (!ok) return;
  const float mk = mask ? mask[(long long)b * T + t] : 1.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = cg + i * CG;
    if (c < C) y[base + (long long)c * T] = ((v[i] - mean) * rstd * gamma[c] + beta[c]) * mk;
  }
}

// ---- reparameterisation (models.py:158-160 / :122-124): z = (m + noise*exp(logs)*scale) * mask -------------
__global__ void reparam_kernel(const float* __restrict__ stats, const float* __restrict__ noise,
                               const float* __restrict__ mask, float* __restrict__ z, int C, int T, float scale,
                               long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int t = (int)(i % T);
  const long long bc = i / T;
  const int c = (int)(bc % C);
  const long long b = bc / C;
  const float m = stats[(b * 2 * C + c) * T + t];
  const float logs = stats[(b * 2 * C + C + c) * T + t];
  const float mk = mask ? mask[b * T + t] : 1.f;
  z[i] = (m + noise[i] * expf(logs) * scale) * mk;
}

// ---- strided [B,C,T] copy with optional mask multiply: Flip (modules/modules.py:232-239), x * x_mask ------
__global__ void copy_bct_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ mask,
                                long long x_bs, long long x_cs, long long y_bs, long long y_cs, long long mask_bs, int C,
                                int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  const int b = blockIdx.z;
  if (t >= T) return;
  float v = x[b * x_bs + c * x_cs + t];
  if (mask) v *= mask[b * mask_bs + t];
  y[b * y_bs + c * y_cs + t] = v;
}

// ---- log-f0 helpers of the automatic f0 predictor (models.py:523-527, utils.py:31-45) ----------------------
// lf0 = 2595*log10(1 + f0/700)/500 ; norm = (lf0 - sum(lf0*uv)/sum(uv)) * factor * mask.  One block per batch item.
__global__ __launch_bounds__(256) void f0_norm_lf0_kernel(const float* __restrict__ f0, const float* __restrict__ uv,
                                                          const float* __restrict__ mask, const float* __restrict__ factor,
                                                          float* __restrict__ lf0, float* __restrict__ norm, int T,
                                                          int input_is_lf0) {
  __shared__ float sh_a[256], sh_b[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  float sa = 0.f, sb = 0.f;
  for (int t = tid; t < T; t += 256) {
    const float fin = f0[(long long)b * T + t];
    const float l = input_is_lf0 ? fin : 2595.f * log10f(1.f + fin / 700.f) / 500.f;
    lf0[(long long)b * T + t] = l;
    const float u = uv[(long long)b * T + t];
    sa += l * u;
    sb += u;
  }
  sh_a[tid] = sa;
  sh_b[tid] = sb;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      sh_a[tid] += sh_a[tid + s];
      sh_b[tid] += sh_b[tid + s];
    }
    __syncthreads();
  }
  float cnt = sh_b[0];
  if (cnt == 0.f) cnt = 9999.f;  // utils.py:34
  const float mean = sh_a[0] / cnt;
  const float fac = factor ? factor[b] : 1.f;
  for (int t = tid; t < T; t += 256) {
    const float mk = mask ? mask[(long long)b * T + t] : 1.f;
    norm[(long long)b * T + t] = (lf0[(long long)b * T + t] - mean) * fac * mk;
  }
}

// f0 = 700 * (10^(lf0 * 500 / 2595) - 1)     (models.py:527)
__global__ void lf0_to_f0_kernel(const float* __restrict__ lf0, float* __restrict__ f0, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) f0[i] = 700.f * (powf(10.f, lf0[i] * 500.f / 2595.f) - 1.f);
}

// ---- GroupNorm(C, C) + GELU: per-(b,c) statistics over time (vencoder/hubert/hubert_model.py:76,87) ---------------
// One workgroup per row; the row (32000 samples for 10 s of 16 kHz audio) is read twice (statistics, then normalise);
// double accumulation keeps the variance stable for long rows.
__global__ __launch_bounds__(256) void channel_norm_gelu_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ y,
                                                                int C, int T, float eps, int apply_gelu) {
  __shared__ double sh[2][256];
  const long long row = blockIdx.x;
  const int c = (int)(row % C);
  const float* xr = x + row * T;
  float* yr = y + row * T;
  double s1 = 0.0, s2 = 0.0;
  for (int t = threadIdx.x; t < T; t += 256) {
    const double v = xr[t];
    s1 += v;
    s2 += v * v;
  }
  sh[0][threadIdx.x] = s1;
  sh[1][threadIdx.x] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  const double mean = sh[0][0] / T;
  const double var = fmax(sh[1][0] / T - mean * mean, 0.0);
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float mu = (float)mean, g = gamma[c], bt = beta[c];
  for (int t = threadIdx.x; t < T; t += 256) {
    const float v = (xr[t] - mu) * rstd * g + bt;
    yr[t] = apply_gelu ? svc_gelu(v) : v;
  }
}

// The same with per-item valid lengths (a batch of utterances of DIFFERENT lengths, zero-padded to T): item b's statistics run over
// its own lens[b] steps — exactly what the item gets when it is processed alone — and its columns beyond that are written as 0.
__global__ __launch_bounds__(256) void channel_norm_gelu_len_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, const int* __restrict__ lens,
                                                                    float* __restrict__ y, int C, int T, float eps, int apply_gelu) {
  __shared__ double sh[2][256];
  const long long row = blockIdx.x;
  const int c = (int)(row % C), b = (int)(row / C);
  const int Tv = min(max(lens[b], 1), T);
  const float* xr = x + row * T;
  float* yr = y + row * T;
  double s1 = 0.0, s2 = 0.0;
  for (int t = threadIdx.x; t < Tv; t += 256) {
    const double v = xr[t];
    s1 += v;
    s2 += v * v;
  }
  sh[0][threadIdx.x] = s1;
  sh[1][threadIdx.x] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  const double mean = sh[0][0] / Tv;
  const double var = fmax(sh[1][0] / Tv - mean * mean, 0.0);
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float mu = (float)mean, g = gamma[c], bt = beta[c];
  for (int t = threadIdx.x; t < T; t += 256) {
    const float v = (xr[t] - mu) * rstd * g + bt;
    yr[t] = t < Tv ? (apply_gelu ? svc_gelu(v) : v) : 0.f;
  }
}

// ---- SinusoidalPosEmb (diffusion/wavenet.py:16-28) ---------------------------------------------------------------
__global__ void sinusoidal_emb_kernel(const float* __restrict__ t, float* __restrict__ out, int B, int dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim / 2;
  if (i >= B * half) return;
  const int b = i / half, j = i - b * half;
  const float scale = logf(10000.f) / (float)(half - 1);
  const float f = expf((float)j * -scale);
  const float v = t[b] * f;
  out[(long long)b * dim + j] = sinf(v);
  out[(long long)b * dim + half + j] = cosf(v);
}

}  // namespace

extern "C" int svc_f0_norm_lf0_f32(const float* f0, const float* uv, const float* mask, const float* factor, float* lf0,
                                   float* norm_lf0, int B, int T, int input_is_lf0, void* stream) {
  SVC_REQUIRE(f0 && uv && lf0 && norm_lf0 && B > 0 && T > 0, "f0_norm_lf0: bad args");
  hipLaunchKernelGGL(f0_norm_lf0_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, f0, uv, mask, factor, lf0, norm_lf0,
                     T, input_is_lf0);
  return svc::check_launch("f0_norm_lf0");
}

extern "C" int svc_lf0_to_f0_f32(const float* lf0, float* f0, long long n, void* stream) {
  SVC_REQUIRE(lf0 && f0 && n > 0, "lf0_to_f0: bad args");
  hipLaunchKernelGGL(lf0_to_f0_kernel, dim3((unsigned)svc::cdivll(n, 256)), dim3(256), 0, (hipStream_t)stream, lf0, f0, n);
  return svc::check_launch("lf0_to_f0");
}

extern "C" int svc_copy_bct_f32(const float* x, float* y, const float* mask, long long x_bs, long long x_cs,
                                long long y_bs, long long y_cs, long long mask_bs, int B, int C, int T, void* stream) {
  SVC_REQUIRE(x && y, "copy_bct: null tensor");
  SVC_REQUIRE(B > 0 && C > 0 && T > 0 && C <= 65535 && B <= 65535, "copy_bct: bad shape");
  hipStream_t s = (hipStream_t)stream;
  svc::ProfScope prof(s, "copy_bct", 0.0, 8.0 * B * C * T);
  hipLaunchKernelGGL(copy_bct_kernel, dim3(svc::cdiv(T, 256), C, B), dim3(256), 0, s, x, y, mask, x_bs, x_cs, y_bs, y_cs,
                     mask_bs, C, T);
  return svc::check_launch("copy_bct");
}

extern "C" int svc_f0_to_coarse(const float* f0, long long* coarse, long long n, void* stream) {
  SVC_REQUIRE(f0 && coarse && n > 0, "f0_to_coarse: bad args");
  hipLaunchKernelGGL(f0_to_coarse_kernel, dim3((unsigned)svc::cdivll(n, 256)), dim3(256), 0, (hipStream_t)stream, f0,
                     coarse, n);
  return svc::check_launch("f0_to_coarse");
}

extern "C" int svc_prenet_embed_f32(const float* xin, const float* uv, const float* f0, const float* emb_uv,
                                    const float* f0_emb, const float* mask, const float* vol, const float* vol_w,
                                    const float* vol_b, float* x, float* x_enc, int B, int C, int T, void* stream) {
  SVC_REQUIRE(xin && uv && f0 && emb_uv && f0_emb && x && x_enc, "prenet_embed: null tensor");
  SVC_REQUIRE(B > 0 && C > 0 && T > 0, "prenet_embed: empty shape");
  SVC_REQUIRE(!vol || (vol_w && vol_b), "prenet_embed: vol given without emb_vol weights");
  hipStream_t s = (hipStream_t)stream;
  svc::ProfScope prof(s, "prenet_embed", 0.0, 12.0 * B * C * T);
  dim3 grid(svc::cdiv(T, 64), 8, B);
  hipLaunchKernelGGL(prenet_embed_kernel, grid, dim3(256), 0, s, xin, uv, f0, emb_uv, f0_emb, mask, vol, vol_w, vol_b, x,
                     x_enc, C, T);
  return svc::check_launch("prenet_embed");
}

static bool ln_reg_on() {      // A/B switch: SVC_LN_REG=0 keeps the three-pass kernel
  static const bool on = [] { const char* e = getenv("SVC_LN_REG"); return !(e && e[0] == '0'); }();
  return on;
}

extern "C" int svc_add_layernorm_f32(const float* x, const float* r, const float* gamma, const float* beta,
                                     const float* mask, float* y, int B, int C, int T, float eps, void* stream) {
  SVC_REQUIRE(x && gamma && beta && y, "add_layernorm: null tensor");
  SVC_REQUIRE(B > 0 && C > 0 && T > 0, "add_layernorm: empty shape");
  hipStream_t s = (hipStream_t)stream;
  svc::ProfScope prof(s, "add_layernorm", 0.0, 12.0 * B * C * T);
  if ((long long)svc::cdiv(T, 32) * B >= 256)
    hipLaunchKernelGGL((add_layernorm_kernel<32, 8>), dim3(svc::cdiv(T, 32), B), dim3(256), 0, s, x, r, gamma, beta, mask, y,
                       C, T, eps);
  else if (ln_reg_on() && C <= 3 * 64)
    hipLaunchKernelGGL((add_layernorm_reg_kernel<3>), dim3(svc::cdiv(T, 4), B), dim3(256), 0, s, x, r, gamma, beta, mask, y, C, T, eps);
  else if (ln_reg_on() && C <= 12 * 64)
    hipLaunchKernelGGL((add_layernorm_reg_kernel<12>), dim3(svc::cdiv(T, 4), B), dim3(256), 0, s, x, r, gamma, beta, mask, y, C, T, eps);
  else
    hipLaunchKernelGGL((add_layernorm_kernel<4, 64>), dim3(svc::cdiv(T, 4), B), dim3(256), 0, s, x, r, gamma, beta, mask, y,
                       C, T, eps);
  return svc::check_launch("add_layernorm");
}

extern "C" int svc_reparam_f32(const float* stats, const float* noise, const float* mask, float* z, int B, int C, int T,
                               float scale, void* stream) {
  SVC_REQUIRE(stats && noise && z, "reparam: null tensor");
  SVC_REQUIRE(B > 0 && C > 0 && T > 0, "reparam: empty shape");
  const long long n = (long long)B * C * T;
  hipStream_t s = (hipStream_t)stream;
  svc::ProfScope prof(s, "reparam", 0.0, 16.0 * n);
  hipLaunchKernelGGL(reparam_kernel, dim3((unsigned)svc::cdivll(n, 256)), dim3(256), 0, s, stats, noise, mask, z, C, T,
                     scale, n);
  return svc::check_launch("reparam");
}

extern "C" int svc_channel_norm_gelu_f32(const float* x, const float* gamma, const float* beta, float* y, int B, int C, int T,
                                         float eps, int apply_gelu, void* stream) {
  SVC_REQUIRE(x && gamma && beta && y && B > 0 && C > 0 && T > 0, "channel_norm_gelu: bad args");
  svc::ProfScope ps((hipStream_t)stream, "channel_norm_gelu", 0.0, 12.0 * B * C * (double)T);
  hipLaunchKernelGGL(channel_norm_gelu_kernel, dim3((unsigned)((long long)B * C)), dim3(256), 0, (hipStream_t)stream, x, gamma,
                     beta, y, C, T, eps, apply_gelu);
  return svc::check_launch("channel_norm_gelu");
}

extern "C" int svc_channel_norm_gelu_len_f32(const float* x, const float* gamma, const float* beta, const int* lens, float* y, int B,
                                             int C, int T, float eps, int apply_gelu, void* stream) {
  SVC_REQUIRE(x && gamma && beta && lens && y && B > 0 && C > 0 && T > 0, "channel_norm_gelu_len: bad args");
  svc::ProfScope ps((hipStream_t)stream, "channel_norm_gelu", 0.0, 12.0 * B * C * (double)T);
  hipLaunchKernelGGL(channel_norm_gelu_len_kernel, dim3((unsigned)((long long)B * C)), dim3(256), 0, (hipStream_t)stream, x, gamma,
                     beta, lens, y, C, T, eps, apply_gelu);
  return svc::check_launch("channel_norm_gelu_len");
}

extern "C" int svc_sinusoidal_emb_f32(const float* t, float* out, int B, int dim, void* stream) {
  SVC_REQUIRE(t && out && B > 0 && dim >= 4 && (dim % 2) == 0, "sinusoidal_emb: bad args");
  hipLaunchKernelGGL(sinusoidal_emb_kernel, dim3(svc::cdiv(B * (dim / 2), 256)), dim3(256), 0, (hipStream_t)stream, t, out, B,
                     dim);
  return svc::check_launch("sinusoidal_emb");
}
