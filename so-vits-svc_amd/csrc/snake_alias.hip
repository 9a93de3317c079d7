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
