// gemm.hip — batched, fully strided fp32 GEMM on the matrix pipe (v_mfma_f32_32x32x2_f32):
//     C[b,m,n] = alpha * sum_k A[b,m,k] * B[b,k,n] + beta * C[b,m,n]
// every operand addressed through element strides (so transposes are free).  Used by the TRAINING path where a
// T x T score matrix per head is affordable (T <= 790 frames, data_utils.py:112-118): attention QK^T / PV and all
// their gradients (modules/attentions.py:207-239), the banded relative-position projections (:259-303), the mel
// filterbank product and its gradient (modules/mel_processing.py:67-76), Linear layers.
// 64 x 64 tile per workgroup (2 x 2 waves, one 32x32 MFMA tile each), K staged 32 at a time into LDS k-major with an
// odd pitch so both the staging writes (lanes along whichever operand dimension is contiguous in memory) and the
// MFMA operand reads (lanes along m / n) are bank-conflict free.
#include "common.h"

namespace {

constexpr int GM = 64, GN = 64, GK = 32, GP = 65;

struct GemmP {
  svc_gemm_args a;
  int a_m_fast, b_n_fast;  // 1: lanes run along m (resp. n) when staging, else along k
};

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmP p) {
  const svc_gemm_args& a = p.a;
  __shared__ float As[GK * GP];
  __shared__ float Bs[GK * GP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ln = lane & 31, lk = lane >> 5;
  const int m0 = blockIdx.y * GM, n0 = blockIdx.x * GN, b = blockIdx.z;
  const float* Ab = a.A + (long long)b * a.a_bs;
  const float* Bb = a.B + (long long)b * a.b_bs;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int k0 = 0; k0 < a.K; k0 += GK) {
    __syncthreads();
    for (int idx = tid; idx < GM * GK; idx += 256) {
      int m, k;
      if (p.a_m_fast) { m = idx % GM; k = idx / GM; } else { k = idx % GK; m = idx / GK; }
      float v = 0.f;
      if (m0 + m < a.M && k0 + k < a.K) v = Ab[(long long)(m0 + m) * a.a_ms + (long long)(k0 + k) * a.a_ks];
      As[k * GP + m] = v;
    }
    for (int idx = tid; idx < GN * GK; idx += 256) {
      int n, k;
      if (p.b_n_fast) { n = idx % GN; k = idx / GN; } else { k = idx % GK; n = idx / GK; }
      float v = 0.f;
      if (n0 + n < a.N && k0 + k < a.K) v = Bb[(long long)(k0 + k) * a.b_ks + (long long)(n0 + n) * a.b_ns];
      Bs[k * GP + n] = v;
    }
    __syncthreads();
    const float* ap = As + lk * GP + wm * 32 + ln;
    const float* bp = Bs + lk * GP + wn * 32 + ln;
#pragma unroll
    for (int kk = 0; kk < GK; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk * GP], bp[kk * GP], acc, 0, 0, 0);
  }
  float* Cb = a.C + (long long)b * a.c_bs;
  const int n = n0 + wn * 32 + ln;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
    if (m < a.M && n < a.N) {
      float* cp = Cb + (long long)m * a.c_ms + (long long)n * a.c_ns;
      float v = a.alpha * acc[r];
      if (a.beta != 0.f) v += a.beta * (*cp);
      *cp = v;
    }
  }
}

// ---- 128 x 128 tile variant: 2 x 2 waves, each wave a 64 x 64 block (2 x 2 MFMA tiles: every A / B operand fetched
// from LDS feeds two MFMAs), K staged 16 at a time; the global loads of chunk i+1 are issued into registers before the
// MFMA loop over chunk i (same software pipeline as conv1d_mfma).  Used when both M and N are >= 96.
constexpr int HM = 128, HN = 128, HK = 16, HP = 129;

// (round 3: a software-pipelined inner loop — second operand register set, reads two k-steps ahead — measured 66.1 vs 64.6 us
// on the QK^T shape, profiles/r03a_gemmbench_*: slower, deleted.)
__global__ __launch_bounds__(256) void gemm_f32_big_kernel(GemmP p) {
  const svc_gemm_args& a = p.a;
  __shared__ float As[HK * HP];
  __shared__ float Bs[HK * HP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ln = lane & 31, lk = lane >> 5;
  const int m0 = blockIdx.y * HM, n0 = blockIdx.x * HN, b = blockIdx.z;
  const float* Ab = a.A + (long long)b * a.a_bs;
  const float* Bb = a.B + (long long)b * a.b_bs;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging slots: 8 per operand per thread.  fast-dim-along-lanes maps: (mn = tid & 127, k = 2*i + (tid >> 7)) when
  // the m / n dimension is contiguous in memory, else (k = tid & 15, mn = 16*i + (tid >> 4)).
  float ar[8], br[8];
  auto amap = [&](int i, int& m, int& k) {
    if (p.a_m_fast) { m = tid & 127; k = 2 * i + (tid >> 7); } else { k = tid & 15; m = 16 * i + (tid >> 4); }
  };
  auto bmap = [&](int i, int& n, int& k) {
    if (p.b_n_fast) { n = tid & 127; k = 2 * i + (tid >> 7); } else { k = tid & 15; n = 16 * i + (tid >> 4); }
  };
  auto load_chunk = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int m, k;
      amap(i, m, k);
      const int mm = min(m0 + m, a.M - 1), kk = min(k0 + k, a.K - 1);
      ar[i] = Ab[(long long)mm * a.a_ms + (long long)kk * a.a_ks];
      int n;
      bmap(i, n, k);
      const int nn = min(n0 + n, a.N - 1), kb = min(k0 + k, a.K - 1);
      br[i] = Bb[(long long)kb * a.b_ks + (long long)nn * a.b_ns];
    }
  };
  auto store_chunk = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int m, k;
      amap(i, m, k);
      As[k * HP + m] = (m0 + m < a.M && k0 + k < a.K) ? ar[i] : 0.f;
      int n;
      bmap(i, n, k);
      Bs[k * HP + n] = (n0 + n < a.N && k0 + k < a.K) ? br[i] : 0.f;
    }
  };

  const float* ap = As + lk * HP + wm * 64 + ln;
  const float* bp = Bs + lk * HP + wn * 64 + ln;
  load_chunk(0);
  for (int k0 = 0; k0 < a.K; k0 += HK) {
    __syncthreads();
    store_chunk(k0);
    __syncthreads();
    if (k0 + HK < a.K) load_chunk(k0 + HK);
#pragma unroll
    for (int kk = 0; kk < HK; kk += 2) {
      const float a0 = ap[kk * HP], a1 = ap[kk * HP + 32];
      const float b0 = bp[kk * HP], b1 = bp[kk * HP + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  
  }
  float* Cb = a.C + (long long)b * a.c_bs;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + ln;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m < a.M && n < a.N) {
          float* cp = Cb + (long long)m * a.c_ms + (long long)n * a.c_ns;
          float v = a.alpha * acc[i][j][r];
          if (a.beta != 0.f) v += a.beta * (*cp);
          *cp = v;
        }
      }
    }
}

}  // namespace

extern "C" int svc_gemm_f32(const svc_gemm_args* ap, void* stream) {
  SVC_REQUIRE(ap != nullptr, "gemm: null args");
  const svc_gemm_args& a = *ap;
  SVC_REQUIRE(a.A && a.B && a.C, "gemm: null tensor");
  SVC_REQUIRE(a.batch > 0 && a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty shape");
  hipStream_t s = (hipStream_t)stream;
  svc::ProfScope prof(s, "gemm_f32", 2.0 * a.batch * (double)a.M * a.N * a.K,
                      4.0 * a.batch * ((double)a.M * a.K + (double)a.K * a.N + (double)a.M * a.N));
  GemmP p;
  p.a = a;
  p.a_m_fast = (a.a_ms == 1 || (a.a_ks != 1 && llabs(a.a_ms) < llabs(a.a_ks))) ? 1 : 0;
  p.b_n_fast = (a.b_ns == 1 || (a.b_ks != 1 && llabs(a.b_ns) < llabs(a.b_ks))) ? 1 : 0;
  if (a.M >= 96 && a.N >= 96) {
    dim3 grid(svc::cdiv(a.N, HN), svc::cdiv(a.M, HM), a.batch);
    hipLaunchKernelGGL(gemm_f32_big_kernel, grid, dim3(256), 0, s, p);
  } else {
    dim3 grid(svc::cdiv(a.N, GN), svc::cdiv(a.M, GM), a.batch);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, s, p);
  }
  return svc::check_launch("gemm_f32");
}
