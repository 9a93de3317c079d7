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
#include <algorithm>
#include <cstdlib>
#include <cstdint>

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

// ---- thin-M variant: M <= 16 rows, a long reduction ------------------------------------------------------------------------
// The relative-position gradients of the training graph (d emb_rel_v: [2w+1 = 9] x [96] per head, reduced over T = 768 frames,
// modules/attentions.py:259-303) are 9 x 96 x 768 products: on the 64 x 64 MFMA tile above 64 workgroups walk the whole
// reduction behind 48 barriers with 86 % of every tile padding — 131 us for 42 MFLOP (0.3 TFLOP/s), 1.6 ms per training
// iteration (profiles/r06a_train_shapes_f32.txt).  Here the REDUCTION is split over workgroups (grid = k-ranges x n-tiles x batch,
// ~400 workgroups), a workgroup stages 64-step chunks of both operands in LDS and its threads own <= 8 outputs each (plain
// FMAs: the product is bound by reading B once, 9 MB); partial sums meet in C through fp32 atomics (C scaled by beta first).
constexpr int TK = 64, TN = 128, TM = 16, TOUT = TM * TN / 256;

__global__ __launch_bounds__(256) void gemm_thin_scale_kernel(svc_gemm_args a) {
  const long long n = (long long)a.batch * a.M * a.N;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int nn = (int)(i % a.N), m = (int)((i / a.N) % a.M), b = (int)(i / ((long long)a.N * a.M));
    float* cp = a.C + (long long)b * a.c_bs + (long long)m * a.c_ms + (long long)nn * a.c_ns;
    *cp = a.beta == 0.f ? 0.f : a.beta * (*cp);
  }
}

__global__ __launch_bounds__(256) void gemm_thin_kernel(GemmP p, int k_per_wg) {
  const svc_gemm_args& a = p.a;
  __shared__ float As[TK * (TM + 1)];
  __shared__ float Bs[TK * (TN + 1)];
  const int tid = threadIdx.x;
  const int n0 = blockIdx.y * TN, b = blockIdx.z;
  const int nt = min(TN, a.N - n0);
  const int k_lo = blockIdx.x * k_per_wg, k_hi = min(a.K, k_lo + k_per_wg);
  const float* Ab = a.A + (long long)b * a.a_bs;
  const float* Bb = a.B + (long long)b * a.b_bs;
  int om[TOUT], on[TOUT];
  float acc[TOUT];
#pragma unroll
  for (int j = 0; j < TOUT; ++j) {
    const int o = tid + 256 * j;
    om[j] = o / nt;
    on[j] = o - om[j] * nt;
    if (om[j] >= a.M) om[j] = -1;
    acc[j] = 0.f;
  }
  for (int k0 = k_lo; k0 < k_hi; k0 += TK) {
    __syncthreads();
    for (int idx = tid; idx < a.M * TK; idx += 256) {
      int m, k;
      if (p.a_m_fast) { m = idx % a.M; k = idx / a.M; } else { k = idx % TK; m = idx / TK; }
      As[k * (TM + 1) + m] = k0 + k < k_hi ? Ab[(long long)m * a.a_ms + (long long)(k0 + k) * a.a_ks] : 0.f;
    }
    for (int idx = tid; idx < nt * TK; idx += 256) {
      int n, k;
      if (p.b_n_fast) { n = idx % nt; k = idx / nt; } else { k = idx % TK; n = idx / TK; }
      Bs[k * (TN + 1) + n] = k0 + k < k_hi ? Bb[(long long)(k0 + k) * a.b_ks + (long long)(n0 + n) * a.b_ns] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TOUT; ++j) {
      if (om[j] < 0) continue;
      const float* ap = As + om[j];
      const float* bp = Bs + on[j];
      float s = 0.f;
#pragma unroll 8
      for (int kk = 0; kk < TK; ++kk) s = fmaf(ap[kk * (TM + 1)], bp[kk * (TN + 1)], s);
      acc[j] += s;
    }
  }
  float* Cb = a.C + (long long)b * a.c_bs;
#pragma unroll
  for (int j = 0; j < TOUT; ++j)
    if (om[j] >= 0) atomicAdd(Cb + (long long)om[j] * a.c_ms + (long long)(n0 + on[j]) * a.c_ns, a.alpha * acc[j]);
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


// ---- register-fed variant (round 3): no LDS, no barrier.  Every attention product of the training graph has each operand
// contiguous along either its reduction index k or its output index (m resp. n) — q / k / v / dO are [B, C, T] tensors, P / dS
// are [BH, T, T] — and both forms hand a lane its MFMA fragment straight from global memory:
//   k-contiguous: lane (m, lk) reads the float4 A[m][8g + 4 lk .. +3]; instruction i of group g consumes element i, i.e. the
//                 reduction index 8g + 4 lk + i (the pairing of k indices over the two lane halves is free as long as A and B agree);
//   m-contiguous: lane (m, lk) reads the scalar A[8g + 4 lk + i][m] for instruction i: 128 contiguous bytes per half wave.
// A wave owns an (MT x NT) block of 32x32 tiles and keeps RD groups of 8 reduction steps in flight in a register ring (L2 /
// HBM latency is hidden by the wave itself and by up to four waves per SIMD); waves of a workgroup are independent.  The
// LDS-staged kernels above restage both operands every 16 reduction steps behind two barriers with per-element 64-bit
// address arithmetic: 120 us (30 TFLOP/s) for out = v P^T (96 x 768 x 768, 32 heads), whose 96 rows also waste a quarter of
// the 128-row tile.
template <bool A_KC, bool B_KC, int MT, int NT>
__global__ __launch_bounds__(256) void gemm_f32_reg_kernel(svc_gemm_args a, int tiles_m, int tiles_n) {
  constexpr int RD = 3;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ln = lane & 31, lk = lane >> 5;
  long long wt = (long long)blockIdx.x * 4 + wave;          // wave tile: n fastest, then m, then batch
  const long long total = (long long)tiles_m * tiles_n * a.batch;
  if (wt >= total) return;
  const int tn = (int)(wt % tiles_n);
  wt /= tiles_n;
  const int tm = (int)(wt % tiles_m);
  const int b = (int)(wt / tiles_m);
  const int m0 = tm * (32 * MT), n0 = tn * (32 * NT);
  const float* Ab = a.A + (long long)b * a.a_bs;
  const float* Bb = a.B + (long long)b * a.b_bs;
  const float* pa[MT];
  const float* pb[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
    pa[i] = A_KC ? Ab + (long long)(m0 + i * 32 + ln) * a.a_ms + 4 * lk : Ab + (m0 + i * 32 + ln) + (long long)(4 * lk) * a.a_ks;
#pragma unroll
  for (int j = 0; j < NT; ++j)
    pb[j] = B_KC ? Bb + (long long)(n0 + j * 32 + ln) * a.b_ns + 4 * lk : Bb + (n0 + j * 32 + ln) + (long long)(4 * lk) * a.b_ks;
  const long long a_gs = A_KC ? 8 : 8 * a.a_ks, b_gs = B_KC ? 8 : 8 * a.b_ks;    // pointer step per group of 8 reduction steps

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[RD][MT], rb[RD][NT];
  auto fetch = [&](int slot, long long g) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const float* q = pa[i] + g * a_gs;
      if constexpr (A_KC) ra[slot][i] = *reinterpret_cast<const float4*>(q);
      else ra[slot][i] = make_float4(q[0], q[a.a_ks], q[2 * a.a_ks], q[3 * a.a_ks]);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float* q = pb[j] + g * b_gs;
      if constexpr (B_KC) rb[slot][j] = *reinterpret_cast<const float4*>(q);
      else rb[slot][j] = make_float4(q[0], q[a.b_ks], q[2 * a.b_ks], q[3 * a.b_ks]);
    }
  };
  auto f4 = [](const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; };
  const int G = a.K >> 3;                                   // launcher: K % 8 == 0
  auto mfmas = [&](int u) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4(ra[u][i], e), f4(rb[u][j], e), acc[i][j], 0, 0, 0);
  };
  static_assert(RD == 3, "the remainder below assumes a ring of three");
  fetch(0, 0);
  fetch(1, min(1, G - 1));
  const int Gb = (G / RD) * RD;
  for (int g0 = 0; g0 < Gb; g0 += RD) {                     // whole trips of the ring: straight-line, no conditionals
#pragma unroll
    for (int u = 0; u < RD; ++u) {
      fetch((u + RD - 1) % RD, min(g0 + u + RD - 1, G - 1));  // past the end: re-reads the last group (unused)
      mfmas(u);
    }
  }
  // the last trip left groups Gb and Gb + 1 (clamped) in slots 0 and 1
  if (Gb < G) mfmas(0);
  if (Gb + 1 < G) mfmas(1);
  float* Cb = a.C + (long long)b * a.c_bs;                  // launcher: c_ns == 1
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float* cp = Cb + (n0 + j * 32 + ln);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        float* q = cp + (long long)m * a.c_ms;
        float v = a.alpha * acc[i][j][r];
        if (a.beta != 0.f) v += a.beta * (*q);
        *q = v;
      }
    }
}

template <bool A_KC, bool B_KC>
int launch_reg(const svc_gemm_args& a, hipStream_t s) {
  auto go = [&](auto k, int mt, int nt) {
    const int tiles_m = a.M / (32 * mt), tiles_n = a.N / (32 * nt);
    const long long total = (long long)tiles_m * tiles_n * a.batch;
    hipLaunchKernelGGL(k, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, a, tiles_m, tiles_n);
    return svc::check_launch("gemm_f32_reg");
  };
  if ((a.M % 64) == 0 && (a.N % 64) == 0) return go(gemm_f32_reg_kernel<A_KC, B_KC, 2, 2>, 2, 2);
  if ((a.M % 96) == 0) return go(gemm_f32_reg_kernel<A_KC, B_KC, 3, 1>, 3, 1);
  return go(gemm_f32_reg_kernel<A_KC, B_KC, 1, 3>, 1, 3);
}

int g_gemm_thin = -1;  // A/B switch: SVC_GEMM_THIN=0 keeps thin-M products on the MFMA tiles
int g_gemm_reg = -1;   // A/B switch: SVC_GEMM_REG=0 keeps every product on the LDS-staged kernels

// operands as the register-fed kernel needs them; everything else stays on the LDS-staged kernels
bool reg_ok(const svc_gemm_args& a) {
  if (g_gemm_reg < 0) {
    const char* e = getenv("SVC_GEMM_REG");
    g_gemm_reg = (e && e[0] == '0') ? 0 : 1;
  }
  if (!g_gemm_reg) return false;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool a_kc = a.a_ks == 1, a_mc = a.a_ms == 1, b_kc = a.b_ks == 1, b_nc = a.b_ns == 1;
  if (!(a_kc || a_mc) || !(b_kc || b_nc) || a.c_ns != 1) return false;
  // long reductions only: measured on the attention products of a B = 16, T = 768 batch (profiles/r04p_gemm_reg{0,1}.txt) the
  // 96 x T x T products (K = 768) go 95 -> 60 us, the T x T x 96 ones (K = 96: one 64 x 64 tile per wave, twice the operand
  // traffic per FLOP of the 128 x 128 LDS tile, and a store-heavy epilogue) 64 -> 76 us
  if ((a.K % 8) != 0 || a.K < 256 || (a.M % 32) != 0 || (a.N % 32) != 0) return false;
  if (!(((a.M % 64) == 0 && (a.N % 64) == 0) || (a.M % 96) == 0 || (a.N % 96) == 0)) return false;
  if (a_kc && !(al16(a.A) && (a.a_ms % 4) == 0 && (a.a_bs % 4) == 0)) return false;
  if (b_kc && !(al16(a.B) && (a.b_ns % 4) == 0 && (a.b_bs % 4) == 0)) return false;
  if (a.a_ms < 0 || a.a_ks < 0 || a.b_ks < 0 || a.b_ns < 0 || a.c_ms < 0) return false;
  return true;
}

}  // namespace

extern "C" int svc_gemm_f32(const svc_gemm_args* ap, void* stream) {
  SVC_REQUIRE(ap != nullptr, "gemm: null args");
  const svc_gemm_args& a = *ap;
  SVC_REQUIRE(a.A && a.B && a.C, "gemm: null tensor");
  SVC_REQUIRE(a.batch > 0 && a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty shape");
  hipStream_t s = (hipStream_t)stream;
  char pname[128];
  if (svc::prof_on() && svc::prof_shapes())   // SVC_PROF_SHAPES=1: one profile row per shape and operand layout (k-contiguous A / B)
    snprintf(pname, sizeof(pname), "gemm_f32[b%d,M%d,N%d,K%d,a%c,b%c]", a.batch, a.M, a.N, a.K, a.a_ks == 1 ? 'k' : (a.a_ms == 1 ? 'm' : 's'),
             a.b_ks == 1 ? 'k' : (a.b_ns == 1 ? 'n' : 's'));
  else
    snprintf(pname, sizeof(pname), "gemm_f32");
  svc::ProfScope prof(s, pname, 2.0 * a.batch * (double)a.M * a.N * a.K,
                      4.0 * a.batch * ((double)a.M * a.K + (double)a.K * a.N + (double)a.M * a.N));
  GemmP p;
  p.a = a;
  p.a_m_fast = (a.a_ms == 1 || (a.a_ks != 1 && llabs(a.a_ms) < llabs(a.a_ks))) ? 1 : 0;
  p.b_n_fast = (a.b_ns == 1 || (a.b_ks != 1 && llabs(a.b_ns) < llabs(a.b_ks))) ? 1 : 0;
  if (g_gemm_thin < 0) {
    const char* e = getenv("SVC_GEMM_THIN");
    g_gemm_thin = (e && e[0] == '0') ? 0 : 1;
  }
  if (g_gemm_thin && a.split_k_atomic && a.M <= TM && a.K >= 4 * TK) {
    const int n_tiles = svc::cdiv(a.N, TN);
    int ksplits = std::max(1, std::min(svc::cdiv(a.K, TK), 512 / std::max(1, n_tiles * a.batch)));
    const int k_per_wg = svc::cdiv(svc::cdiv(a.K, ksplits), TK) * TK;
    ksplits = svc::cdiv(a.K, k_per_wg);
    hipLaunchKernelGGL(gemm_thin_scale_kernel, dim3(std::min(1024, svc::cdiv(a.batch * a.M * a.N, 256))), dim3(256), 0, s, a);
    hipLaunchKernelGGL(gemm_thin_kernel, dim3(ksplits, n_tiles, a.batch), dim3(256), 0, s, p, k_per_wg);
    return svc::check_launch("gemm_thin");
  }
  if (reg_ok(a)) {
    const bool a_kc = a.a_ks == 1, b_kc = a.b_ks == 1;
    if (a_kc && b_kc) return launch_reg<true, true>(a, s);
    if (a_kc) return launch_reg<true, false>(a, s);
    if (b_kc) return launch_reg<false, true>(a, s);
    return launch_reg<false, false>(a, s);
  }
  if (a.M >= 96 && a.N >= 96) {
    dim3 grid(svc::cdiv(a.N, HN), svc::cdiv(a.M, HM), a.batch);
    hipLaunchKernelGGL(gemm_f32_big_kernel, grid, dim3(256), 0, s, p);
  } else {
    dim3 grid(svc::cdiv(a.N, GN), svc::cdiv(a.M, GM), a.batch);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, s, p);
  }
  return svc::check_launch("gemm_f32");
}
