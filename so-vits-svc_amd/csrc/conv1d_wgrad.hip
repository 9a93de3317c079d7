// conv1d_wgrad.hip — weight gradient of a dense (stride-1, dilated) Conv1d on the fp32 matrix pipe:
//     G[ca, cb, k] = sum_{b,t} A[b, ca, t] * Bm[b, cb, t + k*dil - pad]          (t in [0,TA), Bm index in [0,TB))
// With A = dy [B,Cout,Tout] and Bm = x [B,Cin,Tin] this is dW of every nn.Conv1d on the training path
// (backward of the convs cited in conv1d_mfma.hip); with A = x and Bm = the phase-decimated dy it is dW of the
// polyphase ConvTranspose1d (vdecoder/hifigan/models.py:340-342); strided discriminator convs (models.py:171-177)
// go through the same kernel after svc_decimate_f32.
//
// GEMM view per tap: G_k = A (Ca x N) * Bm_k^T (N x Cb), N = (b,t) — the reduction runs over TIME, so the tiles
// staged in LDS are As[64][TT] and Bs[32][TT + halo] with an ODD row pitch: an MFMA operand fetch has its 32 lanes
// on 32 different channel rows at the same time step, which an odd pitch spreads over 32 banks.  One workgroup =
// 64 x 32 output channels, a contiguous range of time tiles; its 4 waves either split the time steps of a tile
// (KS <= 5: every wave accumulates all taps) or split the taps (KS <= 16).  Partial sums are combined with fp32
// atomics into a zero-initialised G (summation order is therefore not fixed run to run, error ~1e-7 relative).
#include "common.h"
#include <algorithm>

namespace {

constexpr int TT = 128;     // time steps per staged tile
constexpr int CA_T = 64;    // rows of A per workgroup (2 MFMA tiles)
constexpr int CB_T = 32;    // rows of Bm per workgroup (1 MFMA tile)

struct WgP {
  const float* A;
  const float* Bm;
  float* G;
  long long a_bs, a_cs, b_bs, b_cs;
  int B, Ca, Cb, TA, TB, KS, dil, pad;
  int tiles_per_b, n_tiles, tiles_per_wg, PB;
};

template <int MODE, int NK>  // MODE 0: waves split time, NK taps each;  MODE 1: waves split taps, NK taps each
__global__ __launch_bounds__(256) void conv1d_wgrad_kernel(WgP p) {
  constexpr int PA = TT + 1;
  extern __shared__ float lds[];
  float* As = lds;               // [CA_T][PA]
  float* Bs = lds + CA_T * PA;   // [CB_T][PB]
  const int PB = p.PB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 31, lk = lane >> 5;
  const int ca0 = blockIdx.y * CA_T, cb0 = blockIdx.z * CB_T;
  const int tile0 = blockIdx.x * p.tiles_per_wg;
  const int tile1 = min(tile0 + p.tiles_per_wg, p.n_tiles);
  const int halo = (p.KS - 1) * p.dil;
  const int XWB = TT + halo;

  f32x16 acc[NK][2];
#pragma unroll
  for (int q = 0; q < NK; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][i][r] = 0.f;

  for (int tile = tile0; tile < tile1; ++tile) {
    const int b = tile / p.tiles_per_b;
    const int t0 = (tile - b * p.tiles_per_b) * TT;
    __syncthreads();
    // stage A tile: CA_T rows x TT, coalesced along t
    for (int idx = tid; idx < CA_T * TT; idx += 256) {
      const int r = idx / TT, c = idx - r * TT;
      const int ca = ca0 + r, t = t0 + c;
      float v = 0.f;
      if (ca < p.Ca && t < p.TA) v = p.A[b * p.a_bs + ca * p.a_cs + t];
      As[r * PA + c] = v;
    }
    for (int idx = tid; idx < CB_T * XWB; idx += 256) {
      const int r = idx / XWB, c = idx - r * XWB;
      const int cb = cb0 + r, t = t0 + c - p.pad;
      float v = 0.f;
      if (cb < p.Cb && t >= 0 && t < p.TB) v = p.Bm[b * p.b_bs + cb * p.b_cs + t];
      Bs[r * PB + c] = v;
    }
    __syncthreads();
    const float* ap = As + ln * PA + lk;
    const float* bp = Bs + ln * PB + lk;
    if constexpr (MODE == 0) {
      const int s0 = wave * (TT / 4), s1 = s0 + TT / 4;
      for (int s = s0; s < s1; s += 2) {
        const float a0 = ap[s], a1 = ap[32 * PA + s];
#pragma unroll
        for (int q = 0; q < NK; ++q) {
          if (q < p.KS) {
            const float bv = bp[s + q * p.dil];
            acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[q][0], 0, 0, 0);
            acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[q][1], 0, 0, 0);
          }
        }
      }
    } else {
      for (int s = 0; s < TT; s += 2) {
        const float a0 = ap[s], a1 = ap[32 * PA + s];
#pragma unroll
        for (int q = 0; q < NK; ++q) {
          const int k = wave + 4 * q;
          if (k < p.KS) {
            const float bv = bp[s + k * p.dil];
            acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[q][0], 0, 0, 0);
            acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[q][1], 0, 0, 0);
          }
        }
      }
    }
  }
  // combine: G[ca][cb][k] += acc
  const int cb = cb0 + ln;
#pragma unroll
  for (int q = 0; q < NK; ++q) {
    const int k = MODE == 0 ? q : wave + 4 * q;
    if (k >= p.KS || cb >= p.Cb) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ca = ca0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (ca < p.Ca) atomicAdd(p.G + ((long long)ca * p.Cb + cb) * p.KS + k, acc[q][i][r]);
      }
  }
}

}  // namespace

extern "C" int svc_conv1d_wgrad_f32(const svc_wgrad_args* ap, void* stream) {
  SVC_REQUIRE(ap != nullptr, "wgrad: null args");
  const svc_wgrad_args& a = *ap;
  SVC_REQUIRE(a.A && a.Bm && a.G, "wgrad: null tensor");
  SVC_REQUIRE(a.B > 0 && a.Ca > 0 && a.Cb > 0 && a.TA > 0 && a.TB > 0, "wgrad: empty shape");
  SVC_REQUIRE(a.KS >= 1 && a.KS <= 16 && a.dil >= 1, "wgrad: KS must be in [1,16] (got %d)", a.KS);
  hipStream_t s = (hipStream_t)stream;
  const double flop = 2.0 * a.B * (double)a.Ca * a.Cb * a.KS * a.TA;
  svc::ProfScope prof(s, "conv1d_wgrad", flop, 4.0 * a.B * ((double)a.Ca * a.TA + (double)a.Cb * a.TB));
  if (!a.accumulate) {
    if (hipMemsetAsync(a.G, 0, sizeof(float) * (size_t)a.Ca * a.Cb * a.KS, s) != hipSuccess) {
      svc::set_error("wgrad: memset failed");
      return SVC_ERR_HIP;
    }
  }
  WgP p;
  p.A = a.A; p.Bm = a.Bm; p.G = a.G;
  p.a_bs = a.a_bs; p.a_cs = a.a_cs; p.b_bs = a.b_bs; p.b_cs = a.b_cs;
  p.B = a.B; p.Ca = a.Ca; p.Cb = a.Cb; p.TA = a.TA; p.TB = a.TB; p.KS = a.KS; p.dil = a.dil; p.pad = a.pad;
  p.tiles_per_b = svc::cdiv(a.TA, TT);
  p.n_tiles = p.tiles_per_b * a.B;
  const int n_ca = svc::cdiv(a.Ca, CA_T), n_cb = svc::cdiv(a.Cb, CB_T);
  // enough time-splits to fill the chip (~1024 workgroups), at least 1 tile each
  int splits = std::max(1, 1024 / (n_ca * n_cb));
  splits = std::min(splits, p.n_tiles);
  p.tiles_per_wg = svc::cdiv(p.n_tiles, splits);
  splits = svc::cdiv(p.n_tiles, p.tiles_per_wg);
  int pb = TT + (a.KS - 1) * a.dil;
  if ((pb & 1) == 0) ++pb;
  p.PB = pb;
  const size_t lds = sizeof(float) * ((size_t)CA_T * (TT + 1) + (size_t)CB_T * pb);
  SVC_REQUIRE(lds <= 64 * 1024, "wgrad: halo too large (KS=%d dil=%d)", a.KS, a.dil);
  dim3 grid(splits, n_ca, n_cb);
  if (a.KS <= 1) hipLaunchKernelGGL((conv1d_wgrad_kernel<0, 1>), grid, dim3(256), lds, s, p);
  else if (a.KS <= 3) hipLaunchKernelGGL((conv1d_wgrad_kernel<0, 3>), grid, dim3(256), lds, s, p);
  else if (a.KS <= 5) hipLaunchKernelGGL((conv1d_wgrad_kernel<0, 5>), grid, dim3(256), lds, s, p);
  else if (a.KS <= 12) hipLaunchKernelGGL((conv1d_wgrad_kernel<1, 3>), grid, dim3(256), lds, s, p);
  else hipLaunchKernelGGL((conv1d_wgrad_kernel<1, 4>), grid, dim3(256), lds, s, p);
  return svc::check_launch("conv1d_wgrad");
}
