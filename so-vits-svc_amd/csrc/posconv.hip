// posconv.hip — HuBERT / ContentVec positional convolution on the fp32 matrix pipe:
//     y = x + gelu(conv1d(x, w, bias, kernel 128, padding 64, groups 16)[..., :-1])
// (vencoder/hubert/hubert_model.py:116-129 `PositionalConvEmbedding`, fairseq `pos_conv`; 768 channels, 48 per group).
//
// Round 2 ran it on the generic grouped-conv kernel written for DiscriminatorS (one thread per output, k = 41): 1.88 ms per
// 10 s clip (T = 500 frames), 2.5 TFLOP/s — a third of the whole unit encoder (profiles/r03l_*).  Per group it is a GEMM
// M = 48 output channels, N = T, K = 48 x 128 = 6144, with a Toeplitz B operand:
//   * one workgroup per (32-column tile, group, batch row): T/32 x 16 x B workgroups (256 at T = 500: one per CU);
//   * the group's x tile [48][32 + 127] is staged ONCE in LDS (zero padded), every tap is an address offset into it;
//   * A = the group's weights, packed [ci][k][64] (48 output channels padded to 64 so that two 32-row MFMA tiles read full
//     128-byte rows), streamed from L2 straight into registers (the tile is visited once: no reuse to stage for);
//   * the 4 waves split the reduction by input channel (12 each), v_mfma_f32_32x32x2_f32 consuming two consecutive taps per
//     instruction; 8 tap pairs (16 weight loads + 8 LDS reads) are in flight per wave ahead of their 16 MFMAs;
//   * partial tiles meet in LDS; bias + exact GELU + residual in the epilogue.
#include "common.h"
#include <algorithm>

namespace {

constexpr int PC_CG = 48;     // channels per group (in = out)
constexpr int PC_MP = 64;     // output channels padded to two MFMA row tiles
constexpr int PC_BN = 32;     // output columns per workgroup

struct PosConvP {
  const float* x;      // [B][C][T]
  const float* w;      // packed [G][CG][KS][64]
  const float* bias;   // [C]
  float* y;            // [B][C][T]
  int B, C, T, KS, pad, G;
};

template <int UNR>
__global__ __launch_bounds__(256) void posconv_kernel(PosConvP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 31, lk = lane >> 5;
  const int t0 = blockIdx.x * PC_BN, g = blockIdx.y, b = blockIdx.z;
  const int XW = PC_BN + p.KS;                       // LDS row: columns t0 - pad .. t0 - pad + XW - 1 (KS - 1 halo + 1 spare)
  float* xs = smem;                                  // [CG][XW]
  const float* xg = p.x + ((long long)b * p.C + (long long)g * PC_CG) * p.T;
  for (int i = tid; i < PC_CG * XW; i += 256) {
    const int c = i / XW, col = i - c * XW;
    const int t = t0 - p.pad + col;
    xs[i] = (t >= 0 && t < p.T) ? xg[(long long)c * p.T + t] : 0.f;
  }
  __syncthreads();

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // wave w reduces input channels [12 w, 12 w + 12); a step = (ci, tap pair kp): lane half lk feeds tap 2 kp + lk
  const int cpw = PC_CG / 4;
  const float* wg = p.w + (long long)g * PC_CG * p.KS * PC_MP;
  const int npair = p.KS >> 1;                        // KS is even (launcher checks)
  for (int ci = wave * cpw; ci < (wave + 1) * cpw; ++ci) {
    const float* wc = wg + ((long long)ci * p.KS + lk) * PC_MP + ln;      // + 2 kp * 64 (+ 32 for the second row tile)
    const float* xc = xs + ci * XW + ln + lk;                             // + 2 kp
    for (int kp0 = 0; kp0 < npair; kp0 += UNR) {
      float a0[UNR], a1[UNR], bx[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int kp = min(kp0 + u, npair - 1);
        a0[u] = wc[(2 * kp) * PC_MP];
        a1[u] = wc[(2 * kp) * PC_MP + 32];
        bx[u] = xc[2 * kp];
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (kp0 + u < npair) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], bx[u], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], bx[u], acc[1], 0, 0, 0);
        }
      }
    }
  }

  // ---- partial tiles -> LDS [4][64][33], summed in a fixed order; bias + exact GELU + residual
  __syncthreads();
  constexpr int CP = PC_BN + 1;
  float* red = smem;                                 // reuses the x tile (everyone is past the MFMA loop)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * PC_MP + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * CP + ln] = acc[i][r];
  __syncthreads();
  for (int i = tid; i < PC_CG * PC_BN; i += 256) {
    const int row = i / PC_BN, col = i - row * PC_BN;
    const int t = t0 + col;
    if (t >= p.T) continue;
    float v = red[row * CP + col];
#pragma unroll
    for (int w = 1; w < 4; ++w) v += red[(w * PC_MP + row) * CP + col];
    const long long o = ((long long)b * p.C + (long long)g * PC_CG + row) * p.T + t;
    v = svc_gelu(v + p.bias[g * PC_CG + row]);
    p.y[o] = p.x[o] + v;
  }
}

// v [C][CG][KS] (+ weight_norm over dim 2: w[:, :, k] = g[k] v[:, :, k] / ||v[:, :, k]||) -> packed [G][CG][KS][64], zero rows 48..63
__global__ void posconv_pack_kernel(const float* __restrict__ v, const float* __restrict__ gk, float* __restrict__ dst, int C,
                                    int KS, int G) {
  // one block per tap: the norm runs over all (co, ci) of that tap
  const int k = blockIdx.x;
  __shared__ float part[256];
  float s = 0.f;
  const int n = C * PC_CG;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float q = v[(long long)i * KS + k];
    s += q * q;
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
    __syncthreads();
  }
  const float scale = gk ? gk[k] / sqrtf(part[0]) : 1.f;
  for (int i = threadIdx.x; i < G * PC_CG * PC_MP; i += 256) {
    const int m = i % PC_MP, ci = (i / PC_MP) % PC_CG, g = i / (PC_MP * PC_CG);
    float q = 0.f;
    if (m < PC_CG) q = v[((long long)(g * PC_CG + m) * PC_CG + ci) * KS + k] * scale;
    dst[(((long long)g * PC_CG + ci) * KS + k) * PC_MP + m] = q;
  }
}

}  // namespace

extern "C" int svc_posconv_pack_f32(const float* v, const float* g, float* dst, int C, int KS, int groups, void* stream) {
  SVC_REQUIRE(v && dst && C > 0 && KS > 0 && groups > 0, "posconv_pack: bad args");
  SVC_REQUIRE(C == groups * PC_CG, "posconv_pack: %d channels in %d groups: this kernel is built for 48 channels per group", C, groups);
  hipLaunchKernelGGL(posconv_pack_kernel, dim3(KS), dim3(256), 0, (hipStream_t)stream, v, g, dst, C, KS, groups);
  return svc::check_launch("posconv_pack");
}

extern "C" int svc_posconv_f32(const float* x, const float* w, const float* bias, float* y, int B, int C, int T, int KS, int pad,
                               int groups, void* stream) {
  SVC_REQUIRE(x && w && bias && y && B > 0 && C > 0 && T > 0, "posconv: bad args");
  SVC_REQUIRE(C == groups * PC_CG && (KS % 2) == 0 && KS >= 2, "posconv: needs 48 channels per group and an even kernel size (got C=%d groups=%d KS=%d)", C, groups, KS);
  SVC_REQUIRE(x != y, "posconv: in-place is not supported (neighbouring tiles read the input)");
  hipStream_t s = (hipStream_t)stream;
  svc::ProfScope prof(s, "posconv", 2.0 * B * (double)C * PC_CG * KS * T, 4.0 * B * 2.0 * C * T + 4.0 * C * PC_CG * KS);
  PosConvP p{x, w, bias, y, B, C, T, KS, pad, groups};
  const size_t lds = std::max((size_t)PC_CG * (PC_BN + KS) * 4, (size_t)4 * PC_MP * (PC_BN + 1) * 4);
  auto k = posconv_kernel<8>;
  static bool done = false;
  if (!done && lds > 64 * 1024) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done = true;
  }
  hipLaunchKernelGGL(k, dim3(svc::cdiv(T, PC_BN), groups, B), dim3(256), lds, s, p);
  return svc::check_launch("posconv");
}
