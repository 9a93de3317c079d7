// snake_alias.hip — the anti-aliased Snake activation of the nsf-snake-hifigan decoder.
//
// Reference: SnakeAlias.forward (vdecoder/hifiganwithsnake/alias/act.py:125-130) =
//   UpSample1d   (alias/resample.py:38-54): replicate-pad 5, depthwise ConvTranspose1d(k=12, s=2) x2, crop 15/15
//   SnakeBeta    (alias/act.py:79-92, log-scale): u + sin^2(e^alpha u) / (e^beta + 1e-9)
//   DownSample1d (alias/filter.py:93-110): replicate-pad (5,6), depthwise Conv1d(k=12, s=2)
// i.e. three aten ops with a 2x-length intermediate in HBM per activation site (33 sites in the decoder).
//
// Here: one kernel, one read and one write of the [B,C,T] activation.  A workgroup owns TILE consecutive samples of one
// (b, c) row: the row segment (+5 halo each side, index-clamped = replicate padding) is staged in LDS, the 2x
// up-sampled + activated signal (2*TILE + 10 values, index-clamped to [0, 2T) = the down-sampler's replicate padding)
// is produced in LDS with the 6-tap polyphase branch that matches each sample's parity, and the 12-tap stride-2
// low-pass reads it back.  HBM-bound: 8 B per element.
#include "common.h"

namespace {

constexpr int SA_TILE = 1024;
constexpr int SA_THREADS = 256;

struct Taps {
  float f[12];
};

__global__ __launch_bounds__(SA_THREADS) void snake_alias_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 const float* __restrict__ alpha,
                                                                 const float* __restrict__ beta, Taps taps,
                                                                 long long x_bs, long long x_cs, long long y_bs,
                                                                 long long y_cs, int T) {
  __shared__ float xs[SA_TILE + 10];
  __shared__ float ua[2 * SA_TILE + 12];
  const int t0 = blockIdx.x * SA_TILE;
  const int c = blockIdx.y, b = blockIdx.z;
  const float* xr = x + b * x_bs + c * x_cs;
  float* yr = y + b * y_bs + c * y_cs;
  const int tid = threadIdx.x;
  for (int i = tid; i < SA_TILE + 10; i += SA_THREADS) {
    int t = t0 - 5 + i;
    t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
    xs[i] = xr[t];
  }
  const float ea = __expf(alpha[c]);
  const float inv_b = 1.f / (__expf(beta[c]) + 1e-9f);
  __syncthreads();
  const int n_lo = 2 * t0 - 5;
  for (int m = tid; m < 2 * SA_TILE + 10; m += SA_THREADS) {
    int n = n_lo + m;
    n = n < 0 ? 0 : (n > 2 * T - 1 ? 2 * T - 1 : n);
    // out[n] = 2 * sum_k f[k] * xpad[(n + 15 - k) / 2] over k with (n + 15 - k) even; xpad[j] = x[clamp(j - 5)]
    const int par = (n + 1) & 1;            // n odd -> even taps (par 0); n even -> odd taps (par 1)
    const int j0 = (n + 15 - par) >> 1;     // j for k = par
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      int xi = j0 - q - 5;                   // global x index before clamping
      xi = xi < 0 ? 0 : (xi > T - 1 ? T - 1 : xi);
      acc = fmaf(par ? taps.f[2 * q + 1] : taps.f[2 * q], xs[xi - (t0 - 5)], acc);
    }
    const float u = 2.f * acc;
    const float s = sinf(u * ea);
    ua[m] = u + inv_b * (s * s);
  }
  __syncthreads();
  for (int i = tid; i < SA_TILE; i += SA_THREADS) {
    const int t = t0 + i;
    if (t >= T) break;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 12; ++k) acc = fmaf(taps.f[k], ua[2 * i + k], acc);
    yr[t] = acc;
  }
}

// ---- backward -----------------------------------------------------------------------------------------------------
// y = D(a), a = u + ib sin^2(ea u), u = U(x)  with U / D the (linear) resamplers incl. their replicate padding.
//   da = D^T dy ;  du = da (1 + ib ea sin(2 ea u)) ;  dx = U^T du
//   dalpha[c] = sum da ib ea u sin(2 ea u) ;  dbeta[c] = - sum da sin^2(ea u) e^beta ib^2
// Replicate padding makes the edge samples absorb the out-of-range taps: with the UNCLAMPED gather
//   da_ext[m] = sum_t f[m + 5 - 2t] dy[t],  da[0] = sum_{m<=0} da_ext[m], da[2T-1] = sum_{m>=2T-1} da_ext[m], da[n] = da_ext[n]
// and  dxpad[j] = 2 sum_k f[k] du[2j - 15 + k],  dx[0] = sum_{j<=5} dxpad[j], dx[T-1] = sum_{j>=T+4} dxpad[j], dx[i] = dxpad[i+5].
// One workgroup = SA_TILE samples of one (b,c) row; u is recomputed from x (cheaper than storing the 2x intermediate).
__global__ __launch_bounds__(SA_THREADS) void snake_alias_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                     const float* __restrict__ alpha,
                                                                     const float* __restrict__ beta, Taps taps,
                                                                     float* __restrict__ dx, float* __restrict__ dalpha,
                                                                     float* __restrict__ dbeta, long long x_bs, long long x_cs,
                                                                     long long g_bs, long long g_cs, long long d_bs,
                                                                     long long d_cs, int T) {
  __shared__ float xs[SA_TILE + 10];
  __shared__ float gs[SA_TILE + 10];
  __shared__ float du[2 * SA_TILE + 12];
  __shared__ float red[2][SA_THREADS / 64];
  const int t0 = blockIdx.x * SA_TILE;
  const int c = blockIdx.y, b = blockIdx.z;
  const float* xr = x + b * x_bs + c * x_cs;
  const float* gr = dy + b * g_bs + c * g_cs;
  float* dr = dx + b * d_bs + c * d_cs;
  const int tid = threadIdx.x;
  for (int i = tid; i < SA_TILE + 10; i += SA_THREADS) {
    const int t = t0 - 5 + i;
    const int tc = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
    xs[i] = xr[tc];
    gs[i] = (t >= 0 && t < T) ? gr[t] : 0.f;       // dy outside the row does not exist (no padding on the output side)
  }
  const float ea = __expf(alpha[c]);
  const float eb = __expf(beta[c]);
  const float ib = 1.f / (eb + 1e-9f);
  __syncthreads();
  const int n_lo = 2 * t0 - 5;
  float sa = 0.f, sb = 0.f;
  auto da_ext = [&](int m) -> float {   // sum_t f[m + 5 - 2t] dy[t] over taps 0..11; t = (m + 5 - k) / 2
    float acc = 0.f;
    const int par = (m + 5) & 1;        // k must have the parity of m + 5
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int k = 2 * q + par;
      const int t = (m + 5 - k) >> 1;   // exact: m + 5 - k is even
      const int li = t - (t0 - 5);
      if (li >= 0 && li < SA_TILE + 10) acc = fmaf(taps.f[k], gs[li], acc);
    }
    return acc;
  };
  for (int mi = tid; mi < 2 * SA_TILE + 10; mi += SA_THREADS) {
    const int n = n_lo + mi;
    float dv = 0.f;
    if (n >= 0 && n <= 2 * T - 1) {
      // u[n] as in the forward kernel
      const int par = (n + 1) & 1;
      const int j0 = (n + 15 - par) >> 1;
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        int xi = j0 - q - 5;
        xi = xi < 0 ? 0 : (xi > T - 1 ? T - 1 : xi);
        acc = fmaf(par ? taps.f[2 * q + 1] : taps.f[2 * q], xs[xi - (t0 - 5)], acc);
      }
      const float u = 2.f * acc;
      float da;
      if (n == 0) {
        da = 0.f;
        for (int m = -5; m <= 0; ++m) da += da_ext(m);
        if (2 * T - 1 == 0) for (int m = 1; m <= 5; ++m) da += da_ext(m);
      } else if (n == 2 * T - 1) {
        da = 0.f;
        for (int m = 2 * T - 1; m <= 2 * T + 5; ++m) da += da_ext(m);
      } else {
        da = da_ext(n);
      }
      const float s1 = sinf(u * ea), s2 = sinf(2.f * u * ea);
      dv = da * (1.f + ib * ea * s2);
      if (n >= 2 * t0 && n < 2 * t0 + 2 * SA_TILE) {   // owned range: each n is counted by exactly one workgroup
        sa += da * ib * ea * u * s2;
        sb -= da * (s1 * s1) * eb * ib * ib;
      }
    }
    du[mi] = dv;
  }
  __syncthreads();
  auto dxpad = [&](int j) -> float {   // 2 sum_k f[k] du[2j - 15 + k]
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int mi = 2 * j - 15 + k - n_lo;
      if (mi >= 0 && mi < 2 * SA_TILE + 10) acc = fmaf(taps.f[k], du[mi], acc);
    }
    return 2.f * acc;
  };
  for (int i = tid; i < SA_TILE; i += SA_THREADS) {
    const int t = t0 + i;
    if (t >= T) break;
    float v = dxpad(t + 5);
    if (t == 0) for (int j = 0; j <= 4; ++j) v += dxpad(j);
    if (t == T - 1) for (int j = T + 5; j <= T + 9; ++j) v += dxpad(j);
    dr[t] = v;
  }
  // per-channel parameter gradients: wave shuffle, LDS across waves, one atomic pair per workgroup
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sa += __shfl_xor(sa, o);
    sb += __shfl_xor(sb, o);
  }
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = sa;
    red[1][tid >> 6] = sb;
  }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, bsum = 0.f;
    for (int w = 0; w < SA_THREADS / 64; ++w) {
      a += red[0][w];
      bsum += red[1][w];
    }
    atomicAdd(dalpha + c, a);
    atomicAdd(dbeta + c, bsum);
  }
}

}  // namespace

extern "C" int svc_snake_alias_f32(const float* x, float* y, const float* alpha, const float* beta,
                                   const float* taps_host, long long x_bs, long long x_cs, long long y_bs,
                                   long long y_cs, int B, int C, int T, void* stream) {
  SVC_REQUIRE(x && y && alpha && beta && taps_host && B > 0 && C > 0 && T > 0, "snake_alias: bad args");
  SVC_REQUIRE(C <= 65535 && B <= 65535, "snake_alias: B, C must be <= 65535");
  Taps tp;
  for (int k = 0; k < 12; ++k) tp.f[k] = taps_host[k];
  svc::ProfScope ps((hipStream_t)stream, "snake_alias", 0.0, 8.0 * B * C * (double)T);
  hipLaunchKernelGGL(snake_alias_kernel, dim3(svc::cdiv(T, SA_TILE), C, B), dim3(SA_THREADS), 0, (hipStream_t)stream, x,
                     y, alpha, beta, tp, x_bs, x_cs, y_bs, y_cs, T);
  return svc::check_launch("snake_alias");
}

extern "C" int svc_snake_alias_bwd_f32(const float* x, const float* dy, const float* alpha, const float* beta,
                                       const float* taps_host, float* dx, float* dalpha, float* dbeta, long long x_bs,
                                       long long x_cs, long long g_bs, long long g_cs, long long d_bs, long long d_cs, int B,
                                       int C, int T, void* stream) {
  SVC_REQUIRE(x && dy && alpha && beta && taps_host && dx && dalpha && dbeta && B > 0 && C > 0 && T > 0,
              "snake_alias_bwd: bad args");
  SVC_REQUIRE(C <= 65535 && B <= 65535, "snake_alias_bwd: B, C must be <= 65535");
  SVC_REQUIRE(T >= 6, "snake_alias_bwd: rows shorter than 6 samples are not supported (got %d)", T);
  Taps tp;
  for (int k = 0; k < 12; ++k) tp.f[k] = taps_host[k];
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(dalpha, 0, sizeof(float) * C, s) != hipSuccess || hipMemsetAsync(dbeta, 0, sizeof(float) * C, s) != hipSuccess) {
    svc::set_error("snake_alias_bwd: memset failed");
    return SVC_ERR_HIP;
  }
  svc::ProfScope ps(s, "snake_alias_bwd", 0.0, 12.0 * B * C * (double)T);
  hipLaunchKernelGGL(snake_alias_bwd_kernel, dim3(svc::cdiv(T, SA_TILE), C, B), dim3(SA_THREADS), 0, s, x, dy, alpha, beta, tp,
                     dx, dalpha, dbeta, x_bs, x_cs, g_bs, g_cs, d_bs, d_cs, T);
  return svc::check_launch("snake_alias_bwd");
}
