// pack.hip — weight-norm fold + repack into the [Cin][KS][CoutP] layout read by the MFMA kernels.
// Reference: torch.nn.utils.weight_norm as applied at vdecoder/hifigan/models.py:41-56,335,340-342,355 and
// modules/modules.py:91-108 (via modules/DSConv.py:65-70): w = v * (g / ||v||_2), norm over all dims but 0.
#include "common.h"

namespace {

__device__ __forceinline__ double block_sum(double v, double* sh) {
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

// one block per packed output row p
__global__ void pack_conv1d_kernel(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ dst,
                                   int Cout, int Cin, int KS, int CoutP, int gate_half) {
  __shared__ double sh[256];
  const int p = blockIdx.x;
  int c;
  if (gate_half > 0) {
    const int i = p >> 6, r = p & 63;
    c = r < 32 ? 32 * i + r : gate_half + 32 * i + (r - 32);
    if ((r < 32 ? 32 * i + r : 32 * i + (r - 32)) >= gate_half) c = -1;
  } else {
    c = p;
  }
  if (c >= Cout) c = -1;
  const int n = Cin * KS;
  float scale = 0.f;
  if (c >= 0) {
    scale = 1.f;
    if (g) {
      double acc = 0.0;
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double x = v[(long long)c * n + i];
        acc += x * x;
      }
      const double ss = block_sum(acc, sh);
      scale = g[c] / (float)sqrt(ss);
    }
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    // i = ci*KS + k  -> dst[(ci*KS + k)*CoutP + p]
    dst[(long long)i * CoutP + p] = c >= 0 ? v[(long long)c * n + i] * scale : 0.f;
  }
}

// ConvTranspose1d: v [Cin][Cout][KS], g [Cin]; norm over (Cout,KS) per ci.  One block per ci.
// dst[ph][ci][mr][CoutP] = w[ci][co][ph + (M-1-mr)*u]  (taps time-reversed so each phase is a plain correlation), or — rows
// layout, svc::convt_rows_layout(u) — dst[ci][mr][co*u + ph] (the same element count): the u phases are rows of ONE convolution
__global__ void pack_convt1d_kernel(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ dst,
                                    int Cin, int Cout, int KS, int CoutP, int u, int M, int rows) {
  __shared__ double sh[256];
  const int ci = blockIdx.x;
  const int n = Cout * KS;
  float scale = 1.f;
  if (g) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const double x = v[(long long)ci * n + i];
      acc += x * x;
    }
    const double ss = block_sum(acc, sh);
    scale = g[ci] / (float)sqrt(ss);
  }
  const int per_ph = M * CoutP;
  for (int i = threadIdx.x; i < u * per_ph; i += blockDim.x) {
    const int ph = i / per_ph;
    const int r = i - ph * per_ph;
    const int mr = r / CoutP, co = r - mr * CoutP;
    const int k = ph + (M - 1 - mr) * u;
    float val = 0.f;
    if (co < Cout && k < KS) val = v[(long long)ci * n + co * KS + k] * scale;
    if (rows) dst[((long long)ci * M + mr) * ((long long)u * CoutP) + (long long)co * u + ph] = val;
    else dst[(((long long)ph * Cin + ci) * M + mr) * CoutP + co] = val;
  }
}

}  // namespace

extern "C" int svc_pack_conv1d_weight(const float* v, const float* g, float* dst, int Cout, int Cin, int KS,
                                      int CoutP, int gate_half, void* stream) {
  SVC_REQUIRE(v && dst, "pack_conv1d: null tensor");
  SVC_REQUIRE(Cout > 0 && Cin > 0 && KS > 0 && CoutP >= Cout, "pack_conv1d: bad shape");
  if (gate_half > 0)
    SVC_REQUIRE(Cout == 2 * gate_half && (gate_half % 32) == 0 && CoutP == Cout,
                "pack_conv1d: gate packing needs Cout == 2*gate_half, gate_half %% 32 == 0, CoutP == Cout");
  hipLaunchKernelGGL(pack_conv1d_kernel, dim3(CoutP), dim3(256), 0, (hipStream_t)stream, v, g, dst, Cout, Cin, KS,
                     CoutP, gate_half);
  return svc::check_launch("pack_conv1d");
}

extern "C" int svc_pack_convt1d_weight(const float* v, const float* g, float* dst, int Cin, int Cout, int KS,
                                       int CoutP, int stride, void* stream) {
  SVC_REQUIRE(v && dst, "pack_convt1d: null tensor");
  SVC_REQUIRE(Cout > 0 && Cin > 0 && KS > 0 && CoutP >= Cout && stride >= 1, "pack_convt1d: bad shape");
  const int M = (KS + stride - 1) / stride;
  hipLaunchKernelGGL(pack_convt1d_kernel, dim3(Cin), dim3(256), 0, (hipStream_t)stream, v, g, dst, Cin, Cout, KS,
                     CoutP, stride, M, svc::convt_rows_layout(stride) ? 1 : 0);
  return svc::check_launch("pack_convt1d");
}

// ---- lane-linear pack of the register-fed short-sequence kernel (conv1d_mfma.hip, conv1d_mfma_direct4_kernel) -----------------
// dst[rt][G][lk*32 + ln][e] = wp[((2*pr + lk)*KS + k)*CoutP + rt*32 + ln],  pr*KS + k = 4*G + e: the four A operands a lane feeds to
// four consecutive reduction steps (channel pair pr, tap k) of row tile rt are 16 contiguous bytes, a wave's 64 lanes 1 KiB.
namespace {
__global__ void pack_conv1d_d4_kernel(const float* __restrict__ wp, float* __restrict__ dst, int Cin, int KS, int CoutP, int NG,
                                      long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
  const long long gi = idx >> 8;
  const int G = (int)(gi % NG), rt = (int)(gi / NG);
  const int S = 4 * G + e, pr = S / KS, k = S - pr * KS;
  const int lk = lane >> 5, ln = lane & 31;
  const int c = 2 * pr + lk;
  dst[idx] = c < Cin ? wp[((long long)c * KS + k) * CoutP + rt * 32 + ln] : 0.f;
}
}  // namespace

extern "C" long long svc_pack_conv1d_d4_floats(int Cin, int KS, int CoutP) {
  if (Cin <= 0 || KS <= 0 || CoutP <= 0 || (Cin & 1) || (CoutP & 31)) return 0;
  const long long NG = ((long long)(Cin / 2) * KS + 3) / 4;
  return (long long)(CoutP / 32) * NG * 256;
}

extern "C" int svc_pack_conv1d_d4(const float* wp, float* dst, int Cin, int KS, int CoutP, void* stream) {
  SVC_REQUIRE(wp && dst, "pack_conv1d_d4: null tensor");
  SVC_REQUIRE(Cin > 0 && (Cin % 2) == 0 && KS > 0 && CoutP > 0 && (CoutP % 32) == 0, "pack_conv1d_d4: Cin must be even and CoutP a multiple of 32");
  const long long total = svc_pack_conv1d_d4_floats(Cin, KS, CoutP);
  const int NG = (int)(((long long)(Cin / 2) * KS + 3) / 4);
  hipLaunchKernelGGL(pack_conv1d_d4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wp, dst, Cin, KS,
                     CoutP, NG, total);
  return svc::check_launch("pack_conv1d_d4");
}
