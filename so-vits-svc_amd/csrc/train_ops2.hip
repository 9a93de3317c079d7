// train_ops2.hip — training-path kernels with a little more structure than train_ops.hip: channel LayerNorm
// forward/backward (modules/modules.py:23-35), masked banded softmax forward/backward and the band gather / scatter
// of the windowed relative-position attention (modules/attentions.py:207-303), embedding gather / scatter-add
// (models.py:393,453,136), reparameterisation backward (models.py:158-160,122-124).
#include "common.h"
#include <algorithm>

namespace {

__device__ __forceinline__ double block_sum_d2(double v, double* sh) {
  const int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (tid < s) sh[tid] += sh[tid + s];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

// ---- LayerNorm over C of [B,C,T]; one thread per (b,t) column, coalesced along t ---------------------------------
// Channel LayerNorm of [B,C,T] (modules/modules.py:23-35).  A workgroup owns 64 consecutive time steps of one batch row;
// its LN_TY waves split the C channels (wave w takes c = w, w + LN_TY, ...: every load is a coalesced 256-byte row segment)
// and combine their partial sums through LDS.  (The first version ran ONE thread per column through all C channels three
// times: 192 waves for B=16, T=768 — 87 us forward / 105 us backward per call, 4.6 ms of a training iteration, 20x the HBM
// time of the 9.4 MB tensor, profiles/r02_w_train_B16_kernel_stats_final_build.txt.)
constexpr int LN_TX = 64, LN_TY = 8;
__device__ __forceinline__ float ln_col_sum(float part, float (*sh)[LN_TX], int tx, int ty) {
  __syncthreads();                 // previous use of sh is over
  sh[ty][tx] = part;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < LN_TY; ++i) tot += sh[i][tx];      // same order in every wave: identical totals
  return tot;
}
__global__ __launch_bounds__(LN_TX * LN_TY) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ y,
                                                               float* __restrict__ mean, float* __restrict__ rstd, int C, int T,
                                                               float eps) {
  __shared__ float sh[LN_TY][LN_TX];
  const int tx = threadIdx.x & (LN_TX - 1), ty = threadIdx.x / LN_TX;
  const int t = blockIdx.x * LN_TX + tx, b = blockIdx.y;
  const bool ok = t < T;
  const float* xp = x + (long long)b * C * T + (ok ? t : T - 1);      // clamped: every thread takes part in the reductions
  float s = 0.f;
  for (int c = ty; c < C; c += LN_TY) s += xp[(long long)c * T];
  const float mu = ln_col_sum(s, sh, tx, ty) / C;
  float v = 0.f;
  for (int c = ty; c < C; c += LN_TY) {
    const float d = xp[(long long)c * T] - mu;
    v += d * d;
  }
  const float rs = rsqrtf(ln_col_sum(v, sh, tx, ty) / C + eps);
  if (!ok) return;
  if (ty == 0) {
    mean[(long long)b * T + t] = mu;
    rstd[(long long)b * T + t] = rs;
  }
  float* yp = y + (long long)b * C * T + t;
  for (int c = ty; c < C; c += LN_TY) yp[(long long)c * T] = (xp[(long long)c * T] - mu) * rs * gamma[c] + beta[c];
}

__global__ __launch_bounds__(LN_TX * LN_TY) void ln_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                  const float* __restrict__ dy, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, float* __restrict__ dx, int C,
                                                                  int T) {
  __shared__ float sh[LN_TY][LN_TX];
  const int tx = threadIdx.x & (LN_TX - 1), ty = threadIdx.x / LN_TX;
  const int t = blockIdx.x * LN_TX + tx, b = blockIdx.y;
  const bool ok = t < T;
  const int tc = ok ? t : T - 1;
  const long long o = (long long)b * C * T + tc;
  const float mu = mean[(long long)b * T + tc], rs = rstd[(long long)b * T + tc];
  float s1 = 0.f, s2 = 0.f;
  for (int c = ty; c < C; c += LN_TY) {
    const float g = dy[o + (long long)c * T] * gamma[c];
    const float xh = (x[o + (long long)c * T] - mu) * rs;
    s1 += g;
    s2 += g * xh;
  }
  s1 = ln_col_sum(s1, sh, tx, ty) / C;
  s2 = ln_col_sum(s2, sh, tx, ty) / C;
  if (!ok) return;
  for (int c = ty; c < C; c += LN_TY) {
    const float g = dy[o + (long long)c * T] * gamma[c];
    const float xh = (x[o + (long long)c * T] - mu) * rs;
    dx[o + (long long)c * T] = rs * (g - s1 - xh * s2);
  }
}
// one block per channel: dgamma[c] = sum dy*xhat, dbeta[c] = sum dy
__global__ void ln_bwd_param_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                    int B, int C, int T) {
  __shared__ double sh[256];
  const int c = blockIdx.x;
  double a1 = 0.0, a2 = 0.0;
  for (int b = 0; b < B; ++b)
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
      const long long o = ((long long)b * C + c) * T + t;
      const float xh = (x[o] - mean[(long long)b * T + t]) * rstd[(long long)b * T + t];
      a1 += (double)dy[o] * xh;
      a2 += dy[o];
    }
  const double s1 = block_sum_d2(a1, sh);
  const double s2 = block_sum_d2(a2, sh);
  if (threadIdx.x == 0) {
    dgamma[c] = (float)s1;
    dbeta[c] = (float)s2;
  }
}

// ---- counter-based uniform draw for the dropout sites (nn.Dropout: modules/attentions.py:232,51,100,344): u in [0, 1) as a
// function of (seed, site, element) — splitmix64's finaliser over seed + site * golden + element — so the forward and the
// backward kernel of a site make the SAME keep decision without a [B,H,T,T] tensor of draws between them (900 MB written and
// twice read per training iteration as torch.rand tensors).  `seed` lives in device memory: a replayed hipGraph sees the value
// the iteration's own increment left there.
__device__ __forceinline__ float svc_hash_uniform(unsigned long long seed, unsigned site, unsigned long long idx) {
  unsigned long long z = seed + (unsigned long long)(site + 1u) * 0x9E3779B97F4A7C15ull + idx * 0xD1B54A32D192ED03ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(unsigned)(z >> 40) * (1.0f / 16777216.0f);      // top 24 bits -> [0, 1)
}

__global__ __launch_bounds__(256) void dropout_rng_kernel(const float* __restrict__ x, float* __restrict__ y, long long n,
                                                          const long long* __restrict__ seed, unsigned site, float p) {
  const unsigned long long sd = (unsigned long long)seed[0];
  const float ks = 1.f / (1.f - p);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    y[i] = svc_hash_uniform(sd, site, (unsigned long long)i) >= p ? x[i] * ks : 0.f;
}

// ---- attention score post-processing: S[bh,i,:] += band(rel[bh,i,:]); mask; softmax over j --------------------------
// mask_mode 1: key/query padding mask m[b,i]*m[b,j] == 0 -> -1e4 (attentions.Encoder :96); 2: causal j > i -> -1e4
// (attentions.FFT :52 via commons.subsequent_mask).  One wave per row.
// Attention-probability dropout (modules/attentions.py:232, `p_attn = self.drop(p_attn)`) rides in the same pass: with
// drop_u (uniform [0,1) draws, one per probability) the kernel keeps P in S (the softmax backward needs it) and writes
// Pd[j] = P[j] * (u[j] >= p ? 1/(1-p) : 0) for the AV / relative-value products.
__global__ __launch_bounds__(256) void attn_softmax_fwd_kernel(float* __restrict__ S, const float* __restrict__ rel,
                                                               const float* __restrict__ mask, int H, int T, int window,
                                                               int mask_mode, long long n_rows,
                                                               const float* __restrict__ drop_u, float p_drop,
                                                               float* __restrict__ Pd, const long long* __restrict__ seed,
                                                               unsigned site) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  const int i = (int)(row % T);
  const long long bh = row / T;
  const int b = (int)(bh / H);
  float* sp = S + row * T;
  const int nrel = 2 * window + 1;
  const float* rp = rel ? rel + row * nrel : nullptr;
  const float* mp = mask ? mask + (long long)b * T : nullptr;
  const float mi = mp ? mp[i] : 1.f;
  float mx = -INFINITY;
  for (int j = lane; j < T; j += 64) {
    float v = sp[j];
    const int r = j - i + window;
    if (rp && r >= 0 && r < nrel) v += rp[r];
    if (mask_mode == 1 && mp && mi * mp[j] == 0.f) v = -1e4f;
    if (mask_mode == 2 && j > i) v = -1e4f;
    sp[j] = v;
    mx = fmaxf(mx, v);
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int j = lane; j < T; j += 64) {
    const float e = expf(sp[j] - mx);
    sp[j] = e;
    sum += e;
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 1.f / sum;
  if (drop_u) {
    const float* up = drop_u + row * T;
    float* dp = Pd + row * T;
    const float ks = 1.f / (1.f - p_drop);
    for (int j = lane; j < T; j += 64) {
      const float pv = sp[j] * inv;
      sp[j] = pv;
      dp[j] = up[j] >= p_drop ? pv * ks : 0.f;
    }
  } else if (seed) {
    float* dp = Pd + row * T;
    const float ks = 1.f / (1.f - p_drop);
    const unsigned long long sd = (unsigned long long)seed[0], base = (unsigned long long)row * T;
    for (int j = lane; j < T; j += 64) {
      const float pv = sp[j] * inv;
      sp[j] = pv;
      dp[j] = svc_hash_uniform(sd, site, base + j) >= p_drop ? pv * ks : 0.f;
    }
  } else {
    for (int j = lane; j < T; j += 64) sp[j] *= inv;
  }
}
// dS = P * (dP - sum_j dP*P), in place on dP; with drop_u the incoming gradient is w.r.t. the dropped probabilities and
// is first multiplied by the same keep mask
// Masked score positions were overwritten by the constant -1e4 in the forward (masked_fill, modules/attentions.py:231): no
// gradient reaches the scores there (matters only for fully masked query rows, whose probabilities are uniform, not 0).
__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, int T,
                                                               long long n_rows, const float* __restrict__ drop_u,
                                                               float p_drop, const float* __restrict__ mask, int H,
                                                               int mask_mode, const long long* __restrict__ seed, unsigned site) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  const int qi = (int)(row % T);
  const float* mp = (mask_mode == 1 && mask) ? mask + (row / T / H) * T : nullptr;
  const float mi = mp ? mp[qi] : 1.f;
  const float* pp = P + row * T;
  float* dp = dP + row * T;
  float dot = 0.f;
  if (drop_u) {
    const float* up = drop_u + row * T;
    const float ks = 1.f / (1.f - p_drop);
    for (int j = lane; j < T; j += 64) {
      const float g = up[j] >= p_drop ? dp[j] * ks : 0.f;
      dp[j] = g;
      dot += pp[j] * g;
    }
  } else if (seed) {
    const float ks = 1.f / (1.f - p_drop);
    const unsigned long long sd = (unsigned long long)seed[0], base = (unsigned long long)row * T;
    for (int j = lane; j < T; j += 64) {
      const float g = svc_hash_uniform(sd, site, base + j) >= p_drop ? dp[j] * ks : 0.f;
      dp[j] = g;
      dot += pp[j] * g;
    }
  } else {
    for (int j = lane; j < T; j += 64) dot += pp[j] * dp[j];
  }
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
  for (int j = lane; j < T; j += 64) {
    const bool masked = (mp && mi * mp[j] == 0.f) || (mask_mode == 2 && j > qi);
    dp[j] = masked ? 0.f : pp[j] * (dp[j] - dot);
  }
}
// band[row, r] = M[row, i + r - window] (0 outside);   scatter: M[row, i + r - window] += band[row, r]
__global__ void band_gather_kernel(const float* __restrict__ M, float* __restrict__ band, int T, int window,
                                   long long n_rows) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nrel = 2 * window + 1;
  if (idx >= n_rows * nrel) return;
  const long long row = idx / nrel;
  const int r = (int)(idx - row * nrel);
  const int i = (int)(row % T);
  const int j = i + r - window;
  band[idx] = (j >= 0 && j < T) ? M[row * T + j] : 0.f;
}
__global__ void band_scatter_add_kernel(float* __restrict__ M, const float* __restrict__ band, int T, int window,
                                        long long n_rows) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nrel = 2 * window + 1;
  if (idx >= n_rows * nrel) return;
  const long long row = idx / nrel;
  const int r = (int)(idx - row * nrel);
  const int i = (int)(row % T);
  const int j = i + r - window;
  if (j >= 0 && j < T) M[row * T + j] += band[idx];
}

// ---- embedding: y[b,c,t] = W[idx[b,t], c]  and its adjoint (atomic scatter-add into dW) ---------------------------
__global__ void embed_fwd_kernel(const long long* __restrict__ idx, const float* __restrict__ W, float* __restrict__ y,
                                 int C, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  y[((long long)b * C + c) * T + t] = W[idx[(long long)b * T + t] * C + c];
}
// One workgroup = one channel c and EB_POS consecutive (b,t) positions: the gradient rows are first summed in an LDS
// table (the path's tables have 2 (emb_uv), 200 (emb_g) or 256 (f0_emb) rows), so only n_rows global atomics leave the
// workgroup instead of one per position (2.4 M atomics onto 384 addresses for emb_uv at B=16, T=768).
constexpr int EB_POS = 2048;
__global__ __launch_bounds__(256) void embed_bwd_kernel(const long long* __restrict__ idx, const float* __restrict__ dy,
                                                        float* __restrict__ dW, int B, int C, int T, int n_rows) {
  extern __shared__ float hist[];
  const int c = blockIdx.y;
  const long long n = (long long)B * T;
  for (int r = threadIdx.x; r < n_rows; r += 256) hist[r] = 0.f;
  __syncthreads();
  const long long p0 = (long long)blockIdx.x * EB_POS;
  for (int i = threadIdx.x; i < EB_POS; i += 256) {
    const long long pos = p0 + i;
    if (pos >= n) break;
    const int b = (int)(pos / T), t = (int)(pos - (long long)b * T);
    const long long row = idx[pos];
    if (row >= 0 && row < n_rows) atomicAdd(&hist[row], dy[((long long)b * C + c) * T + t]);
  }
  __syncthreads();
  for (int r = threadIdx.x; r < n_rows; r += 256) {
    const float v = hist[r];
    if (v != 0.f) atomicAdd(dW + (long long)r * C + c, v);
  }
}

// ---- reparameterisation backward: z = (m + n*exp(logs)*scale)*mask  ->  dstats = [dm ; dlogs] -------------------------
__global__ void reparam_bwd_kernel(const float* __restrict__ stats, const float* __restrict__ noise,
                                   const float* __restrict__ mask, const float* __restrict__ dz, float* __restrict__ dstats,
                                   int C, int T, float scale) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const float mk = mask ? mask[(long long)b * T + t] : 1.f;
  const float g = dz[((long long)b * C + c) * T + t] * mk;
  const long long om = ((long long)b * 2 * C + c) * T + t, ol = ((long long)b * 2 * C + C + c) * T + t;
  dstats[om] = g;
  dstats[ol] = g * noise[((long long)b * C + c) * T + t] * expf(stats[ol]) * scale;
}

// ---- NSF source tail for training: har = tanh(sum_h waves[.,h]*w[h] + b0)  (vdecoder/hifigan/models.py:318) ---------
__global__ void nsf_linear_fwd_kernel(const float* __restrict__ waves, const float* __restrict__ w, const float* __restrict__ b0,
                                      float* __restrict__ har, long long n, int H) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc = b0[0];
  for (int h = 0; h < H; ++h) acc = fmaf(waves[i * H + h], w[h], acc);
  har[i] = tanhf(acc);
}
// dw[h] += sum_i dz_i * waves[i,h], db += sum_i dz_i, dz = dhar*(1-har^2)   (no gradient to waves: f0 is an input)
__global__ void nsf_linear_bwd_kernel(const float* __restrict__ waves, const float* __restrict__ har,
                                      const float* __restrict__ dhar, float* __restrict__ dw, float* __restrict__ db,
                                      long long n, int H) {
  __shared__ double sh[256];
  double acc[17];
  for (int h = 0; h <= H; ++h) acc[h] = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float y = har[i];
    const float dz = dhar[i] * (1.f - y * y);
    for (int h = 0; h < H; ++h) acc[h] += (double)dz * waves[i * H + h];
    acc[H] += dz;
  }
  for (int h = 0; h <= H; ++h) {
    const double s = block_sum_d2(acc[h], sh);
    if (threadIdx.x == 0) atomicAdd(h < H ? dw + h : db, (float)s);
  }
}

}  // namespace

extern "C" {

int svc_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int B,
                          int C, int T, float eps, void* stream) {
  SVC_REQUIRE(x && gamma && beta && y && mean && rstd && B > 0 && C > 0 && T > 0, "layernorm_fwd: bad args");
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(svc::cdiv(T, LN_TX), B), dim3(LN_TX * LN_TY), 0, (hipStream_t)stream, x, gamma, beta, y, mean,
                     rstd, C, T, eps);
  return svc::check_launch("layernorm_fwd");
}

int svc_layernorm_bwd_f32(const float* x, const float* gamma, const float* dy, const float* mean, const float* rstd,
                          float* dx, float* dgamma, float* dbeta, int B, int C, int T, void* stream) {
  SVC_REQUIRE(x && gamma && dy && mean && rstd && dx && dgamma && dbeta && B > 0 && C > 0 && T > 0, "layernorm_bwd: bad args");
  hipLaunchKernelGGL(ln_bwd_dx_kernel, dim3(svc::cdiv(T, LN_TX), B), dim3(LN_TX * LN_TY), 0, (hipStream_t)stream, x, gamma, dy, mean, rstd,
                     dx, C, T);
  hipLaunchKernelGGL(ln_bwd_param_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, dy, mean, rstd, dgamma, dbeta, B, C, T);
  return svc::check_launch("layernorm_bwd");
}

int svc_attn_softmax_fwd_f32(float* S, const float* rel, const float* mask, int B, int H, int T, int window, int mask_mode,
                             const float* drop_u, float p_drop, float* Pd, void* stream) {
  SVC_REQUIRE(S && B > 0 && H > 0 && T > 0 && window >= 0, "attn_softmax_fwd: bad args");
  SVC_REQUIRE(drop_u == nullptr || (Pd != nullptr && p_drop >= 0.f && p_drop < 1.f), "attn_softmax_fwd: dropout needs Pd and 0 <= p < 1");
  const long long rows = (long long)B * H * T;
  hipLaunchKernelGGL(attn_softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, rel, mask,
                     H, T, window, mask_mode, rows, drop_u, p_drop, Pd, (const long long*)nullptr, 0u);
  return svc::check_launch("attn_softmax_fwd");
}

int svc_attn_softmax_fwd_rng_f32(float* S, const float* rel, const float* mask, int B, int H, int T, int window, int mask_mode,
                                 const long long* seed, int site, float p_drop, float* Pd, void* stream) {
  SVC_REQUIRE(S && seed && Pd && B > 0 && H > 0 && T > 0 && window >= 0 && site >= 0, "attn_softmax_fwd_rng: bad args");
  SVC_REQUIRE(p_drop > 0.f && p_drop < 1.f, "attn_softmax_fwd_rng: 0 < p < 1");
  const long long rows = (long long)B * H * T;
  hipLaunchKernelGGL(attn_softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, rel, mask,
                     H, T, window, mask_mode, rows, (const float*)nullptr, p_drop, Pd, seed, (unsigned)site);
  return svc::check_launch("attn_softmax_fwd_rng");
}

int svc_attn_softmax_bwd_f32(const float* P, float* dP, int B, int H, int T, const float* drop_u, float p_drop,
                             const float* mask, int mask_mode, void* stream) {
  SVC_REQUIRE(P && dP && B > 0 && H > 0 && T > 0, "attn_softmax_bwd: bad args");
  const long long rows = (long long)B * H * T;
  hipLaunchKernelGGL(attn_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, P, dP, T, rows,
                     drop_u, p_drop, mask, H, mask_mode, (const long long*)nullptr, 0u);
  return svc::check_launch("attn_softmax_bwd");
}

int svc_attn_softmax_bwd_rng_f32(const float* P, float* dP, int B, int H, int T, const long long* seed, int site, float p_drop,
                                 const float* mask, int mask_mode, void* stream) {
  SVC_REQUIRE(P && dP && seed && B > 0 && H > 0 && T > 0 && site >= 0 && p_drop > 0.f && p_drop < 1.f, "attn_softmax_bwd_rng: bad args");
  const long long rows = (long long)B * H * T;
  hipLaunchKernelGGL(attn_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, P, dP, T, rows,
                     (const float*)nullptr, p_drop, mask, H, mask_mode, seed, (unsigned)site);
  return svc::check_launch("attn_softmax_bwd_rng");
}

/* y = x * (u >= p ? 1 / (1 - p) : 0) with u = the counter-based draw of (seed, site, element): nn.Dropout(p) forward on x, and its
 * backward on dy (same seed and site). */
int svc_dropout_rng_f32(const float* x, float* y, long long n, const long long* seed, int site, float p, void* stream) {
  SVC_REQUIRE(x && y && seed && n > 0 && site >= 0 && p > 0.f && p < 1.f, "dropout_rng: bad args");
  hipLaunchKernelGGL(dropout_rng_kernel, dim3((unsigned)std::min<long long>((n + 1023) / 1024, 65535)), dim3(256), 0, (hipStream_t)stream,
                     x, y, n, seed, (unsigned)site, p);
  return svc::check_launch("dropout_rng");
}

int svc_band_gather_f32(const float* M, float* band, long long n_rows, int T, int window, void* stream) {
  SVC_REQUIRE(M && band && n_rows > 0 && T > 0 && window >= 0, "band_gather: bad args");
  const long long n = n_rows * (2 * window + 1);
  hipLaunchKernelGGL(band_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, band, T, window, n_rows);
  return svc::check_launch("band_gather");
}

int svc_band_scatter_add_f32(float* M, const float* band, long long n_rows, int T, int window, void* stream) {
  SVC_REQUIRE(M && band && n_rows > 0 && T > 0 && window >= 0, "band_scatter_add: bad args");
  const long long n = n_rows * (2 * window + 1);
  hipLaunchKernelGGL(band_scatter_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, band, T, window, n_rows);
  return svc::check_launch("band_scatter_add");
}

int svc_embed_fwd_f32(const long long* idx, const float* W, float* y, int B, int C, int T, void* stream) {
  SVC_REQUIRE(idx && W && y && B > 0 && C > 0 && T > 0, "embed_fwd: bad args");
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(svc::cdiv(T, 64), C, B), dim3(64), 0, (hipStream_t)stream, idx, W, y, C, T);
  return svc::check_launch("embed_fwd");
}

int svc_embed_bwd_f32(const long long* idx, const float* dy, float* dW, int B, int C, int T, int n_rows, void* stream) {
  SVC_REQUIRE(idx && dy && dW && B > 0 && C > 0 && T > 0 && n_rows > 0, "embed_bwd: bad args");
  SVC_REQUIRE(n_rows <= 16384, "embed_bwd: table of %d rows does not fit the LDS accumulator", n_rows);
  const long long n = (long long)B * T;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)((n + EB_POS - 1) / EB_POS), C), dim3(256), sizeof(float) * n_rows,
                     (hipStream_t)stream, idx, dy, dW, B, C, T, n_rows);
  return svc::check_launch("embed_bwd");
}

int svc_reparam_bwd_f32(const float* stats, const float* noise, const float* mask, const float* dz, float* dstats, int B,
                        int C, int T, float scale, void* stream) {
  SVC_REQUIRE(stats && noise && dz && dstats && B > 0 && C > 0 && T > 0, "reparam_bwd: bad args");
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(svc::cdiv(T, 64), C, B), dim3(64), 0, (hipStream_t)stream, stats, noise, mask, dz,
                     dstats, C, T, scale);
  return svc::check_launch("reparam_bwd");
}

int svc_nsf_linear_fwd_f32(const float* waves, const float* w, const float* b0, float* har, long long n, int H, void* stream) {
  SVC_REQUIRE(waves && w && b0 && har && n > 0 && H > 0 && H <= 16, "nsf_linear_fwd: bad args");
  hipLaunchKernelGGL(nsf_linear_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, waves, w, b0, har, n, H);
  return svc::check_launch("nsf_linear_fwd");
}

int svc_nsf_linear_bwd_f32(const float* waves, const float* har, const float* dhar, float* dw, float* db, long long n, int H,
                           void* stream) {
  SVC_REQUIRE(waves && har && dhar && dw && db && n > 0 && H > 0 && H <= 16, "nsf_linear_bwd: bad args");
  if (hipMemsetAsync(dw, 0, sizeof(float) * H, (hipStream_t)stream) != hipSuccess ||
      hipMemsetAsync(db, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) {
    svc::set_error("nsf_linear_bwd: memset failed");
    return SVC_ERR_HIP;
  }
  const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, 512);
  hipLaunchKernelGGL(nsf_linear_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, waves, har, dhar, dw, db, n, H);
  return svc::check_launch("nsf_linear_bwd");
}

}  // extern "C"

// ---- KL term of the VITS loss (modules/losses.py:43-58) and STFT framing (modules/mel_processing.py:40-64) -----------
namespace {

// acc[0] += sum kl*mask, acc[1] += sum mask (each mask element once);  kl = lp - lq - 0.5 + 0.5*(zp-mp)^2*exp(-2 lp)
__global__ void kl_fwd_kernel(const float* __restrict__ zp, const float* __restrict__ lq, const float* __restrict__ mp,
                              const float* __restrict__ lp, const float* __restrict__ mask, double* __restrict__ acc, int C,
                              int T, long long n) {
  __shared__ double sh[256];
  double a0 = 0.0, a1 = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const long long bc = i / T;
    const int c = (int)(bc % C);
    const long long b = bc / C;
    const float mk = mask[b * T + t];
    const float d = zp[i] - mp[i];
    const float kl = lp[i] - lq[i] - 0.5f + 0.5f * d * d * expf(-2.f * lp[i]);
    a0 += (double)(kl * mk);
    if (c == 0) a1 += mk;
  }
  const double s0 = block_sum_d2(a0, sh);
  const double s1 = block_sum_d2(a1, sh);
  if (threadIdx.x == 0) {
    atomicAdd(acc, s0);
    atomicAdd(acc + 1, s1);
  }
}
// gradients of sum(kl*mask) scaled by the device scalar *g
__global__ void kl_bwd_kernel(const float* __restrict__ zp, const float* __restrict__ mp, const float* __restrict__ lp,
                              const float* __restrict__ mask, const float* __restrict__ g, float* __restrict__ dzp,
                              float* __restrict__ dlq, float* __restrict__ dmp, float* __restrict__ dlp, int C, int T,
                              long long n) {
  const float gs = g[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const long long b = i / T / C;
    const float mk = mask[b * T + t] * gs;
    const float d = zp[i] - mp[i];
    const float e = expf(-2.f * lp[i]);
    dzp[i] = mk * d * e;
    dmp[i] = -mk * d * e;
    dlq[i] = -mk;
    dlp[i] = mk * (1.f - d * d * e);
  }
}

// frames[b, f, n] = ypad[b, f*hop + n] * win[n], ypad = reflect pad of y by `pad` on both sides
__device__ __forceinline__ int reflect_idx(int p, int L) {   // p in [-pad, L+pad)
  if (p < 0) p = -p;
  if (p >= L) p = 2 * L - 2 - p;
  return p;
}
__global__ void stft_frame_kernel(const float* __restrict__ y, const float* __restrict__ win, float* __restrict__ frames,
                                  int L, int NF, int nfft, int hop, int pad) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int f = blockIdx.y, b = blockIdx.z;
  if (n >= nfft) return;
  const int p = reflect_idx(f * hop + n - pad, L);
  frames[((long long)b * NF + f) * nfft + n] = y[(long long)b * L + p] * win[n];
}
__global__ void stft_frame_bwd_kernel(const float* __restrict__ dframes, const float* __restrict__ win, float* __restrict__ dy,
                                      int L, int NF, int nfft, int hop, int pad) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int f = blockIdx.y, b = blockIdx.z;
  if (n >= nfft) return;
  const int p = reflect_idx(f * hop + n - pad, L);
  atomicAdd(dy + (long long)b * L + p, dframes[((long long)b * NF + f) * nfft + n] * win[n]);
}
// DFT basis: cs[n, k] = cos(2 pi k n / N), sn[n, k] = -sin(2 pi k n / N)   ([N][NB])
__global__ void dft_basis_kernel(float* __restrict__ cs, float* __restrict__ sn, int N, int NB) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (k >= NB) return;
  const int r = (int)(((long long)k * n) % N);
  const double th = 2.0 * 3.14159265358979323846 * (double)r / (double)N;
  cs[(long long)n * NB + k] = (float)cos(th);
  sn[(long long)n * NB + k] = (float)(-sin(th));
}
// mag = sqrt(re^2 + im^2 + eps);  bwd: dre = dmag*re/mag, dim = dmag*im/mag
__global__ void cmag_kernel(const float* __restrict__ re, const float* __restrict__ im, float* __restrict__ mag, long long n,
                            float eps) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    mag[i] = sqrtf(re[i] * re[i] + im[i] * im[i] + eps);
}
__global__ void cmag_bwd_kernel(const float* __restrict__ re, const float* __restrict__ im, const float* __restrict__ mag,
                                const float* __restrict__ dmag, float* __restrict__ dre, float* __restrict__ dim, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float s = dmag[i] / mag[i];
    dre[i] = s * re[i];
    dim[i] = s * im[i];
  }
}

// ---- leaky ReLU on rows with a padded tail -----------------------------------------------------------------------
// DiscriminatorP's feature maps are kept as [rows, P] with P = the row length rounded up so that every row starts on a
// 16-byte boundary (P % 4 == 0) and only the first L columns meaningful.  y = lrelu(x) on t < L and 0 on the tail;
// dx = dy * (y > 0 ? 1 : slope) on t < L and 0 on the tail.  The zero tail is what lets the neighbouring convolutions
// (forward, dgrad and wgrad) run over the PHYSICAL row length with the float4 / LDS-DMA staging paths: zeros beyond L are
// exactly the implicit zero padding of the logical signal.  slope = 1 gives the plain tail mask (after conv_post).
__global__ void lrelu_tail_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ y, long long n4, int P4, int L,
                                      float slope) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % P4) * 4;
    float4 v = x[i];
    v.x = t + 0 < L ? (v.x > 0.f ? v.x : slope * v.x) : 0.f;
    v.y = t + 1 < L ? (v.y > 0.f ? v.y : slope * v.y) : 0.f;
    v.z = t + 2 < L ? (v.z > 0.f ? v.z : slope * v.z) : 0.f;
    v.w = t + 3 < L ? (v.w > 0.f ? v.w : slope * v.w) : 0.f;
    y[i] = v;
  }
}
__global__ void lrelu_tail_bwd_kernel(const float4* __restrict__ y, const float4* __restrict__ dy, float4* __restrict__ dx,
                                      long long n4, int P4, int L, float slope) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % P4) * 4;
    const float4 o = y[i];
    float4 g = dy[i];
    g.x = t + 0 < L ? (o.x > 0.f ? g.x : slope * g.x) : 0.f;
    g.y = t + 1 < L ? (o.y > 0.f ? g.y : slope * g.y) : 0.f;
    g.z = t + 2 < L ? (o.z > 0.f ? g.z : slope * g.z) : 0.f;
    g.w = t + 3 < L ? (o.w > 0.f ? g.w : slope * g.w) : 0.f;
    dx[i] = g;
  }
}

// ---- spectral norm (torch.nn.utils.spectral_norm, one power iteration; models.py:170,205 with use_spectral_norm=True) --------
// W [R][K] (weight_orig viewed as [Cout, rest]), u [R], v [K].  Training: v <- normalize(W^T u), u <- normalize(W v) IN PLACE
// (the buffers of the module), then sigma = u . (W v) and w = W / sigma.  Eval: sigma from the stored u, v.
// Backward (u, v constants): dW = g / sigma - (sum g*W) / sigma^2 * u v^T.
__global__ void sn_matvec_t_kernel(const float* __restrict__ W, const float* __restrict__ u, float* __restrict__ out, int R,
                                   int K) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= K) return;
  double acc = 0.0;
  for (int r = 0; r < R; ++r) acc += (double)W[(long long)r * K + j] * (double)u[r];
  out[j] = (float)acc;
}
__global__ void sn_matvec_kernel(const float* __restrict__ W, const float* __restrict__ v, float* __restrict__ out, int K) {
  __shared__ double sh[256];
  const int r = blockIdx.x;
  double acc = 0.0;
  for (int j = threadIdx.x; j < K; j += blockDim.x) acc += (double)W[(long long)r * K + j] * (double)v[j];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[r] = (float)sh[0];
}
// single block: x <- x / max(||x||, eps)   (F.normalize(x, dim=0, eps))
__global__ void sn_normalize_kernel(float* __restrict__ x, int n, float eps) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += (double)x[i] * (double)x[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  const float nr = fmaxf((float)sqrt(sh[0]), eps);
  for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] = x[i] / nr;
}
// single block: u <- normalize(wv) when `update`, sigma = sum u[r] * wv[r]
__global__ void sn_sigma_kernel(float* __restrict__ u, const float* __restrict__ wv, float* __restrict__ sigma, int R, float eps,
                                int update) {
  __shared__ double sh[256];
  if (update) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < R; i += blockDim.x) acc += (double)wv[i] * (double)wv[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
      if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
      __syncthreads();
    }
    const float nr = fmaxf((float)sqrt(sh[0]), eps);
    __syncthreads();
    for (int i = threadIdx.x; i < R; i += blockDim.x) u[i] = wv[i] / nr;
    __syncthreads();
  }
  double acc = 0.0;
  for (int i = threadIdx.x; i < R; i += blockDim.x) acc += (double)u[i] * (double)wv[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) sigma[0] = (float)sh[0];
}
__global__ void sn_scale_kernel(const float* __restrict__ W, const float* __restrict__ sigma, float* __restrict__ w, long long n) {
  const float s = sigma[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) w[i] = W[i] / s;
}
__global__ void sn_dot_kernel(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ out, long long n) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    acc += (double)a[i] * (double)b[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(out, sh[0]);
}
__global__ void sn_bwd_kernel(const float* __restrict__ g, const float* __restrict__ u, const float* __restrict__ v,
                              const float* __restrict__ sigma, const double* __restrict__ dot, float* __restrict__ dW, int R,
                              int K) {
  const double s = sigma[0];
  const double c = dot[0] / (s * s);
  const long long n = (long long)R * K;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / K), j = (int)(i - (long long)r * K);
    dW[i] = (float)((double)g[i] / s - c * (double)u[r] * (double)v[j]);
  }
}

}  // namespace

struct GuardP {
  const float* p[8];
};
__global__ void nonfinite_guard_kernel(GuardP gp, int n, int* __restrict__ counter) {
  const int i = threadIdx.x;
  int bad = 0;
  if (i < n) {
    const float v = *gp.p[i];
    bad = !(v == v) || v == INFINITY || v == -INFINITY;
  }
  const unsigned long long m = __ballot(bad);
  if (i == 0) {
    const int it = counter[2] + 1;
    counter[2] = it;
    if (m) {
      counter[0] += __popcll(m);
      counter[1] = it;
    }
  }
}

extern "C" {

int svc_kl_fwd_f64(const float* z_p, const float* logs_q, const float* m_p, const float* logs_p, const float* mask,
                   double* acc2, int B, int C, int T, void* stream) {
  SVC_REQUIRE(z_p && logs_q && m_p && logs_p && mask && acc2 && B > 0 && C > 0 && T > 0, "kl_fwd: bad args");
  const long long n = (long long)B * C * T;
  hipLaunchKernelGGL(kl_fwd_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 1024)), dim3(256), 0,
                     (hipStream_t)stream, z_p, logs_q, m_p, logs_p, mask, acc2, C, T, n);
  return svc::check_launch("kl_fwd");
}

int svc_kl_bwd_f32(const float* z_p, const float* m_p, const float* logs_p, const float* mask, const float* g, float* dz_p,
                   float* dlogs_q, float* dm_p, float* dlogs_p, int B, int C, int T, void* stream) {
  SVC_REQUIRE(z_p && m_p && logs_p && mask && g && dz_p && dlogs_q && dm_p && dlogs_p, "kl_bwd: bad args");
  const long long n = (long long)B * C * T;
  hipLaunchKernelGGL(kl_bwd_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0,
                     (hipStream_t)stream, z_p, m_p, logs_p, mask, g, dz_p, dlogs_q, dm_p, dlogs_p, C, T, n);
  return svc::check_launch("kl_bwd");
}

int svc_stft_frame_f32(const float* y, const float* win, float* frames, int B, int L, int NF, int nfft, int hop, int pad,
                       void* stream) {
  SVC_REQUIRE(y && win && frames && B > 0 && L > pad && NF > 0, "stft_frame: bad args");
  hipLaunchKernelGGL(stft_frame_kernel, dim3(svc::cdiv(nfft, 256), NF, B), dim3(256), 0, (hipStream_t)stream, y, win, frames, L,
                     NF, nfft, hop, pad);
  return svc::check_launch("stft_frame");
}

int svc_stft_frame_bwd_f32(const float* dframes, const float* win, float* dy, int B, int L, int NF, int nfft, int hop, int pad,
                           void* stream) {
  SVC_REQUIRE(dframes && win && dy && B > 0 && L > pad && NF > 0, "stft_frame_bwd: bad args");
  if (hipMemsetAsync(dy, 0, sizeof(float) * (size_t)B * L, (hipStream_t)stream) != hipSuccess) {
    svc::set_error("stft_frame_bwd: memset failed");
    return SVC_ERR_HIP;
  }
  hipLaunchKernelGGL(stft_frame_bwd_kernel, dim3(svc::cdiv(nfft, 256), NF, B), dim3(256), 0, (hipStream_t)stream, dframes, win,
                     dy, L, NF, nfft, hop, pad);
  return svc::check_launch("stft_frame_bwd");
}

int svc_dft_basis_f32(float* cs, float* sn, int N, int NB, void* stream) {
  SVC_REQUIRE(cs && sn && N > 0 && NB > 0, "dft_basis: bad args");
  hipLaunchKernelGGL(dft_basis_kernel, dim3(svc::cdiv(NB, 256), N), dim3(256), 0, (hipStream_t)stream, cs, sn, N, NB);
  return svc::check_launch("dft_basis");
}

int svc_cmag_f32(const float* re, const float* im, float* mag, long long n, float eps, void* stream) {
  SVC_REQUIRE(re && im && mag && n > 0, "cmag: bad args");
  hipLaunchKernelGGL(cmag_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                     re, im, mag, n, eps);
  return svc::check_launch("cmag");
}

int svc_cmag_bwd_f32(const float* re, const float* im, const float* mag, const float* dmag, float* dre, float* dim, long long n,
                     void* stream) {
  SVC_REQUIRE(re && im && mag && dmag && dre && dim && n > 0, "cmag_bwd: bad args");
  hipLaunchKernelGGL(cmag_bwd_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0,
                     (hipStream_t)stream, re, im, mag, dmag, dre, dim, n);
  return svc::check_launch("cmag_bwd");
}

int svc_lrelu_tail_fwd_f32(const float* x, float* y, long long rows, int P, int L, float slope, void* stream) {
  SVC_REQUIRE(x && y && rows > 0 && P > 0 && (P % 4) == 0 && L >= 0 && L <= P, "lrelu_tail_fwd: bad args (P % 4 == 0, L <= P)");
  SVC_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, "lrelu_tail_fwd: 16 B alignment");
  const long long n4 = rows * (P / 4);
  hipLaunchKernelGGL(lrelu_tail_fwd_kernel, dim3((unsigned)std::min<long long>((n4 + 255) / 256, 8192)), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), n4, P / 4, L, slope);
  return svc::check_launch("lrelu_tail_fwd");
}

int svc_lrelu_tail_bwd_f32(const float* y, const float* dy, float* dx, long long rows, int P, int L, float slope,
                           void* stream) {
  SVC_REQUIRE(y && dy && dx && rows > 0 && P > 0 && (P % 4) == 0 && L >= 0 && L <= P, "lrelu_tail_bwd: bad args");
  SVC_REQUIRE(((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0,
              "lrelu_tail_bwd: 16 B alignment");
  const long long n4 = rows * (P / 4);
  hipLaunchKernelGGL(lrelu_tail_bwd_kernel, dim3((unsigned)std::min<long long>((n4 + 255) / 256, 8192)), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const float4*>(y), reinterpret_cast<const float4*>(dy),
                     reinterpret_cast<float4*>(dx), n4, P / 4, L, slope);
  return svc::check_launch("lrelu_tail_bwd");
}

int svc_spectral_norm_fwd_f32(const float* W, float* u, float* v, float* w, float* sigma, float* tmp, int R, int K,
                              int power_iteration, float eps, void* stream) {
  SVC_REQUIRE(W && u && v && w && sigma && tmp && R > 0 && K > 0, "spectral_norm_fwd: bad args");
  hipStream_t s = (hipStream_t)stream;
  if (power_iteration) {
    hipLaunchKernelGGL(sn_matvec_t_kernel, dim3(svc::cdiv(K, 256)), dim3(256), 0, s, W, u, v, R, K);
    hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(256), 0, s, v, K, eps);
  }
  hipLaunchKernelGGL(sn_matvec_kernel, dim3(R), dim3(256), 0, s, W, v, tmp, K);
  hipLaunchKernelGGL(sn_sigma_kernel, dim3(1), dim3(256), 0, s, u, tmp, sigma, R, eps, power_iteration ? 1 : 0);
  const long long n = (long long)R * K;
  hipLaunchKernelGGL(sn_scale_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0, s, W, sigma, w, n);
  return svc::check_launch("spectral_norm_fwd");
}

int svc_spectral_norm_bwd_f32(const float* W, const float* u, const float* v, const float* sigma, const float* g, float* dW,
                              double* dot_ws, int R, int K, void* stream) {
  SVC_REQUIRE(W && u && v && sigma && g && dW && dot_ws && R > 0 && K > 0, "spectral_norm_bwd: bad args");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(dot_ws, 0, sizeof(double), s) != hipSuccess) {
    svc::set_error("spectral_norm_bwd: memset failed");
    return SVC_ERR_HIP;
  }
  const long long n = (long long)R * K;
  const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(sn_dot_kernel, dim3(grid), dim3(256), 0, s, g, W, dot_ws, n);
  hipLaunchKernelGGL(sn_bwd_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0, s, g, u, v, sigma,
                     dot_ws, dW, R, K);
  return svc::check_launch("spectral_norm_bwd");
}


/* Device-side guard of a replayed training iteration: counter[2] += 1 (launches so far), counter[0] += number of non-finite values
 * among the n (<= 8) scalars, counter[1] = the launch number (1-based) of the LAST launch that saw one (0 = never).  The counters
 * are sticky; the host reads them every few hundred steps instead of synchronising on every loss. */
int svc_nonfinite_guard_f32(const float* const* scalars, int n, int* counter, void* stream) {
  SVC_REQUIRE(scalars && counter && n > 0 && n <= 8, "nonfinite_guard: 1..8 scalars");
  GuardP gp;
  for (int i = 0; i < 8; ++i) gp.p[i] = i < n ? scalars[i] : nullptr;
  for (int i = 0; i < n; ++i) SVC_REQUIRE(gp.p[i] != nullptr, "nonfinite_guard: null scalar %d", i);
  hipLaunchKernelGGL(nonfinite_guard_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, gp, n, counter);
  return svc::check_launch("nonfinite_guard");
}

}  // extern "C"
