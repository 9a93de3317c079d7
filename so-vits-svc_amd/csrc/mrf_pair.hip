// mrf_pair.hip — the 16-channel decoder stage's ResBlock1 (vdecoder/hifigan/models.py:60-67) as fused kernels:
//   svc_resblock_pair_f32   one dilation pair   out = conv2( lrelu( conv1( lrelu(x) ) + b1 ) ) + b2 + x        (mrf_pair16_kernel)
//   svc_resblock16_f32      all three pairs of the block in one launch                                         (mrf_block16_kernel)
// As separate launches the six convs of a block at C = 16 (T = 441,344 samples per 10 s clip) are HBM-bound: each reads its input
// AND the residual and writes its output, 84 MB per launch (2.0-2.5 TB/s measured, 20-50 TFLOP/s).  Fused, x is read once and the
// result written once; intermediates and residuals stay in LDS / registers.  C = 16 -> v_mfma_f32_16x16x4_f32 (M = 16 output
// channels, 4 input channels per instruction) with the weights of a pair in registers.  Reduction order = the unfused kernels'
// (channel groups outer, taps inner; fp32 fmaf chains), epilogue arithmetic in the same order: bit-identical to the conv launches.
// (Rounds 2-5 also carried the first form of the pair — persistent workgroups, leaky-ReLU per operand read, C = 16 and a C = 32
// variant that never beat the two strip launches; removed in round 6 with the merged-launch path: DESIGN.md §3.)
#include "common.h"
#include <algorithm>
#include <type_traits>

namespace {

struct PairP {
  svc_resblock_pair_args a;
  int n_tiles;   // time tiles per batch item
  int XW, MW;    // LDS row pitches of the x and mid tiles (floats)
};

__device__ __forceinline__ float lrelu01(float v, float s) { return fmaxf(v, v * s); }   // 0 <= s <= 1

// ---- round 4: the 16-channel pair, second form --------------------------------------------------------------------------------
// What the first form lost (ISA + counters): per MFMA one ds_read -> s_waitcnt lgkmcnt(0) -> 3 VALU of leaky-ReLU -> MFMA
// on ONE accumulator, i.e. LDS latency, VALU and the matrix pipe never overlapped (43 TFLOP/s = 0.27 of peak); persistent
// workgroups over 1724 tiles on 512 slots (3.37 tiles each, the slowest takes 4); three un-overlapped phases per tile.  Here:
//   * a wave owns FOUR column tiles at a time (four independent accumulators sharing the weight operand): per (channel group,
//     tap) four independent LDS reads feed four MFMAs, and the reads of the next tap are in flight under them;
//   * the leaky-ReLU is applied ONCE, when the x tile is staged (the first form applied it to every operand read: K times per
//     element); the residual is re-read from global memory in the epilogue (the tile is L2-resident: the workgroup loaded it a few
//     microseconds earlier) and prefetched in front of the second conv's MFMA loop together with the accumulate operand;
//   * BN = 240 outputs per workgroup: conv1 needs 240 + 2*H2 <= 250 columns = 16 tiles (4 per wave, none left over), conv2 15;
//   * ONE tile per workgroup, 1839 workgroups for a 10 s clip, three resident per CU: the hardware dispatcher balances them and
//     one workgroup's staging / store phases run under its neighbours' MFMA loops.
// Same reduction order and epilogue expression as the first form: bit-identical results.
template <int KS>
__global__ __launch_bounds__(256, 2) void mrf_pair16_kernel(PairP p) {
  constexpr int C = 16, TS = 16, KPI = 4, NG = 4, BN = 240, NTW = 4;
  constexpr int H2 = (KS - 1) / 2;
  const svc_resblock_pair_args& a = p.a;
  const int d = a.dil1, H1 = d * H2, HX = H1 + H2;
  const int XW = p.XW, MW = p.MW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;                 // [C][XW]  lrelu(x), columns t0 - HX ...
  float* ms = xs + C * XW;         // [C][MW]  lrelu(conv1 + b1), columns t0 - H2 ...

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 15, lk = lane >> 4;
  const float slope = a.slope;
  const int tile = blockIdx.x;
  const int b = tile / p.n_tiles;
  const int t0 = (tile - b * p.n_tiles) * BN;
  const float* xb = a.x + (long long)b * a.x_bs;
  float* yb = a.y + (long long)b * a.y_bs;

  // ---- 1. stage lrelu(x): [C][BN + 2*HX] (zero outside the sequence), coalesced along time; all loads of a thread in flight ----
  const int xcols = BN + 2 * HX;
  {
    float v[C][2];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int col = tid + 256 * h;
        const int t = t0 - HX + col;
        const int tc = min(max(t, 0), a.T - 1);
        const float x = xb[(long long)c * a.x_cs + tc];
        v[c][h] = (t >= 0 && t < a.T) ? x : 0.f;
      }
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int col = tid + 256 * h;
        if (col < xcols) xs[c * XW + col] = lrelu01(v[c][h], slope);
      }
  }
  // weights and biases in registers while the tile lands
  float w1r[NG][KS], w2r[NG][KS];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      w1r[g][k] = a.w1[((g * KPI + lk) * KS + k) * a.CP + ln];
      w2r[g][k] = a.w2[((g * KPI + lk) * KS + k) * a.CP + ln];
    }
  float b1r[4], b2r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    b1r[r] = a.b1 ? a.b1[4 * lk + r] : 0.f;
    b2r[r] = a.b2 ? a.b2[4 * lk + r] : 0.f;
  }
  __syncthreads();

  // ---- 2. conv1 -> mid: this wave's four column tiles [64*wave, 64*wave + 64) of the 256 staged mid columns ----
  {
    f32x4 acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xr = xs + lk * XW + 64 * wave + ln;
    // operand reads one (channel group, tap) step ahead of their MFMAs, pinned with sched_barriers: left alone the compiler
    // emits ds_read2 -> s_waitcnt lgkmcnt(0) -> 2 MFMAs on ONE reused register pair (read off the ISA)
    float bv[2][NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) bv[0][j] = xr[j * TS];
#pragma unroll
    for (int st = 0; st < NG * KS; ++st) {
      __builtin_amdgcn_sched_barrier(0);
      if (st + 1 < NG * KS) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) bv[(st + 1) & 1][j] = xr[((st + 1) / KS) * KPI * XW + ((st + 1) % KS) * d + j * TS];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1r[st / KS][st % KS], bv[st & 1][j], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int col = 64 * wave + j * TS + ln;
      const int t = t0 - H2 + col;
      const bool ok = t >= 0 && t < a.T;     // conv2's zero padding pads lrelu(conv1(..)), not conv1 evaluated past the ends
#pragma unroll
      for (int r = 0; r < 4; ++r) ms[(4 * lk + r) * MW + col] = ok ? lrelu01(acc[j][r] + b1r[r], slope) : 0.f;
    }
  }
  // ---- 3. conv2 + residual -> out: column tiles 4*wave .. 4*wave + 3 of the 15 (the 16th is computed and dropped) ----
  float xres[NTW][4], yold[NTW][4];
  const bool accum = a.beta != 0.f;
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int t = min(t0 + 64 * wave + j * TS + ln, a.T - 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xres[j][r] = xb[(long long)(4 * lk + r) * a.x_cs + t];
      yold[j][r] = accum ? yb[(long long)(4 * lk + r) * a.y_cs + t] : 0.f;
    }
  }
  __syncthreads();
  {
    f32x4 acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* mr = ms + lk * MW + 64 * wave + ln;
    float bv[2][NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) bv[0][j] = mr[j * TS];
#pragma unroll
    for (int st = 0; st < NG * KS; ++st) {
      __builtin_amdgcn_sched_barrier(0);
      if (st + 1 < NG * KS) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) bv[(st + 1) & 1][j] = mr[((st + 1) / KS) * KPI * MW + ((st + 1) % KS) + j * TS];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2r[st / KS][st % KS], bv[st & 1][j], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int col = 64 * wave + j * TS + ln;
      const int t = t0 + col;
      if (col < BN && t < a.T) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[j][r] + b2r[r];
          v = v + xres[j][r];
          if (accum) v = v + a.beta * yold[j][r];
          if (a.out_div != 1.f) v = v / a.out_div;
          yb[(long long)(4 * lk + r) * a.y_cs + t] = v;
        }
      }
    }
  }
}


// ---- round 6: the whole 16-channel ResBlock1 (all three dilation pairs) in ONE launch ---------------------------------------------
//     for d in dilations:  x = conv2_d( lrelu( conv1_d( lrelu(x) ) + b1_d ) ) + b2_d + x          (vdecoder/hifigan/models.py:60-67)
// The pair kernel above leaves a 16-channel MRF stage at 0.33 of the fp32 matrix peak: a workgroup lives ~29 us for ~5 us of matrix
// work (stage the tile, load 88 weight words per lane, two convs, store) and the stage moves 106 MB per launch where a whole ResBlock
// needs ~56 (profiles/r05d_pmc_mrf_pair16.txt).  Here a workgroup carries its tile through all three pairs: x is read once and the
// result written once (a third of the pair path's HBM bytes and launches), and the fixed costs of a workgroup are paid once per three
// pairs.  Price: a halo of sum_d (d + 1)(K - 1)/2 columns per side, recomputed (K = 11: 60 of 512 columns per side).
//   * Column c of EVERY tensor of the chain is the same sample t = t0 - HT + c (tiles carry margins instead of shifting their origin):
//     a lane's accumulator registers of one pair are the residual operand of the next — the raw x_j never goes through LDS, only
//     lrelu(x_j) (conv1's operand) and lrelu(mid) (conv2's) do;
//   * what lies outside [0, T) is forced to zero after every conv, exactly the zero padding of the unfused convs; columns whose
//     inputs reach beyond the tile compute garbage that no valid output reads (the valid range shrinks by (d + 1)(K - 1)/2 per pair);
//   * weights of a pair in registers as in the pair kernel; the next pair's are fetched under the current pair's matrix loops;
//   * same reduction order and epilogue expressions as the pair kernel: bit-identical to the three launches it replaces.
struct BlkP {
  svc_resblock16_args a;
  int n_tiles;   // tiles per batch item
};

// NW waves of NTW column tiles each: 4 x 4 / 4 x 6 for 3 / 7 taps (two or three workgroups per CU); 11 taps need 88 weight registers per
// lane, so their 512 columns are EIGHT waves of four tiles (one workgroup per CU, two waves per SIMD).  The dilations are template
// parameters (the decoder's ResBlock1 is always (1, 3, 5)): every LDS operand address is then `one base register + immediate`, where
// run-time dilations and pitches cost ~130 registers of precomputed addresses (the first build spilled).
template <int KS, int NTW, int NW, int D0, int D1, int D2>
struct Blk16 {
  static constexpr int C = 16, TS = 16, KPI = 4, NG = 4, W = 16 * NTW * NW, H2 = (KS - 1) / 2, NTHR = 64 * NW;
  static constexpr int DMAX = D0 > D1 ? (D0 > D2 ? D0 : D2) : (D1 > D2 ? D1 : D2);
  static constexpr int HT = (D0 + D1 + D2 + 3) * H2;       // halo per side: sum over pairs of (d + 1)(K - 1)/2
  static constexpr int MG = DMAX * H2;                     // margin of the x tile
  static constexpr int BN = W - 2 * HT;                    // outputs per workgroup
  static constexpr int pitch(int w) {                      // 4 channel rows x 16 lanes of one operand fetch on disjoint bank groups
    w = (w + 3) & ~3;
    while ((w & 63) != 16 && (w & 63) != 48) w += 4;
    return w;
  }
  static constexpr int AW = pitch(W + 2 * MG), MW = pitch(W + 2 * H2);
  static constexpr size_t LDS = (size_t)16 * (AW + MW) * 4;
  static_assert(BN >= 64, "tile too narrow for these dilations");
};

template <int KS, int NTW, int NW, int D0, int D1, int D2>
__global__ __launch_bounds__(64 * NW, 2) void mrf_block16_kernel(BlkP p) {
  using Cf = Blk16<KS, NTW, NW, D0, D1, D2>;
  constexpr int C = 16, TS = 16, KPI = 4, NG = 4, W = Cf::W, H2 = Cf::H2, NTHR = Cf::NTHR;
  constexpr int AW = Cf::AW, MW = Cf::MW, MG = Cf::MG, HT = Cf::HT, BN = Cf::BN;
  const svc_resblock16_args& a = p.a;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;                 // [C][AW]  lrelu(x_j): column c at MG + c
  float* ms = xs + C * AW;         // [C][MW]  lrelu(conv1 + b1): column c at H2 + c
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 15, lk = lane >> 4;
  const float slope = a.slope;
  const int tile = blockIdx.x;
  const int b = tile / p.n_tiles;
  const int t0 = (tile - b * p.n_tiles) * BN;
  const float* xb = a.x + (long long)b * a.x_bs;
  float* yb = a.y + (long long)b * a.y_bs;
  const int cbase = wave * (TS * NTW) + ln;      // this lane's column in its wave's tile j: cbase + 16 j

  // ---- stage lrelu(x): columns [0, W) (zero outside the sequence), zero margins; raw x of this lane's C-layout elements in registers
  {
    constexpr int CPT = (W + NTHR - 1) / NTHR;   // columns per thread and channel (the last round may be partial)
    float v[C][CPT];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int h = 0; h < CPT; ++h) {
        const int t = t0 - HT + tid + NTHR * h;
        const float x = xb[(long long)c * a.x_cs + min(max(t, 0), a.T - 1)];
        v[c][h] = (t >= 0 && t < a.T) ? x : 0.f;
      }
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int h = 0; h < CPT; ++h)
        if (tid + NTHR * h < W) xs[c * AW + MG + tid + NTHR * h] = lrelu01(v[c][h], slope);
    for (int i = tid; i < C * 2 * MG; i += NTHR) {
      const int c = i / (2 * MG), m = i - c * 2 * MG;
      xs[c * AW + (m < MG ? m : W + m)] = 0.f;
    }
    for (int i = tid; i < C * 2 * H2; i += NTHR) {
      const int c = i / (2 * H2), m = i - c * 2 * H2;
      ms[c * MW + (m < H2 ? m : W + m)] = 0.f;
    }
  }
  float xreg[NTW][4];
  bool ok[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int t = t0 - HT + cbase + TS * j;
    ok[j] = t >= 0 && t < a.T;
    const int tc = min(max(t, 0), a.T - 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float x = xb[(long long)(4 * lk + r) * a.x_cs + tc];
      xreg[j][r] = ok[j] ? x : 0.f;
    }
  }
  float w1r[NG][KS], w2r[NG][KS], b1r[4], b2r[4];
  auto load_w = [&](float (&wr)[NG][KS], float (&br)[4], const float* w, const float* bias) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int k = 0; k < KS; ++k) wr[g][k] = w[((g * KPI + lk) * KS + k) * a.CP + ln];
#pragma unroll
    for (int r = 0; r < 4; ++r) br[r] = bias ? bias[4 * lk + r] : 0.f;
  };
  load_w(w1r, b1r, a.w1[0], a.b1[0]);
  load_w(w2r, b2r, a.w2[0], a.b2[0]);
  __syncthreads();

  const float* xr0 = xs + lk * AW + MG + cbase;
  const float* mr = ms + lk * MW + cbase;
  float* msw = ms + (4 * lk) * MW + H2 + cbase;
  float* xsw = xs + (4 * lk) * AW + MG + cbase;

  auto pair = [&](auto PJC) {
    constexpr int PJ = decltype(PJC)::value;
    constexpr int d = PJ == 0 ? D0 : (PJ == 1 ? D1 : D2), H1 = d * H2;
    constexpr bool lastp = PJ == 2;
    // ---- conv1 (dilation d) -> lrelu(. + b1), zero outside the sequence -> mid tile
    {
      f32x4 acc[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* xr = xr0 - H1;
      float bv[2][NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) bv[0][j] = xr[j * TS];
#pragma unroll
      for (int st = 0; st < NG * KS; ++st) {
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < NG * KS) {
#pragma unroll
          for (int j = 0; j < NTW; ++j) bv[(st + 1) & 1][j] = xr[((st + 1) / KS) * KPI * AW + ((st + 1) % KS) * d + j * TS];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1r[st / KS][st % KS], bv[st & 1][j], acc[j], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) msw[r * MW + TS * j] = ok[j] ? lrelu01(acc[j][r] + b1r[r], slope) : 0.f;
    }
    if constexpr (!lastp) load_w(w1r, b1r, a.w1[PJ + 1], a.b1[PJ + 1]);      // in flight under conv2
    __syncthreads();
    // ---- conv2 (dilation 1) + b2 + x_j -> x_{j+1} (registers; lrelu of it back into the x tile for the next pair)
    {
      f32x4 acc[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      float bv[2][NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) bv[0][j] = mr[j * TS];
#pragma unroll
      for (int st = 0; st < NG * KS; ++st) {
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < NG * KS) {
#pragma unroll
          for (int j = 0; j < NTW; ++j) bv[(st + 1) & 1][j] = mr[((st + 1) / KS) * KPI * MW + ((st + 1) % KS) + j * TS];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2r[st / KS][st % KS], bv[st & 1][j], acc[j], 0, 0, 0);
      }
      if constexpr (!lastp) {
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[j][r] + b2r[r];
            v = v + xreg[j][r];
            v = ok[j] ? v : 0.f;
            xreg[j][r] = v;
            xsw[r * AW + TS * j] = lrelu01(v, slope);
          }
      } else {
        // (the accumulate operand is read here, once per workgroup, into registers the weights no longer need)
        const bool accum = a.beta != 0.f;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          const int c = cbase + TS * j;
          const int t = t0 - HT + c;
          if (c >= HT && c < HT + BN && t < a.T) {
            float yold[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) yold[r] = accum ? yb[(long long)(4 * lk + r) * a.y_cs + t] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = acc[j][r] + b2r[r];
              v = v + xreg[j][r];
              if (accum) v = v + a.beta * yold[r];
              if (a.out_div != 1.f) v = v / a.out_div;
              yb[(long long)(4 * lk + r) * a.y_cs + t] = v;
            }
          }
        }
      }
    }
    if constexpr (!lastp) {
      load_w(w2r, b2r, a.w2[PJ + 1], a.b2[PJ + 1]);                       // in flight under the next conv1
      __syncthreads();
    }
  };
  pair(std::integral_constant<int, 0>{});
  pair(std::integral_constant<int, 1>{});
  pair(std::integral_constant<int, 2>{});
}

template <int KS, int NTW, int NW, int D0, int D1, int D2>
int launch_block16(const svc_resblock16_args& a, hipStream_t s) {
  using Cf = Blk16<KS, NTW, NW, D0, D1, D2>;
  BlkP p;
  p.a = a;
  p.n_tiles = svc::cdiv(a.T, Cf::BN);
  auto kern = mrf_block16_kernel<KS, NTW, NW, D0, D1, D2>;
  if (Cf::LDS > 64 * 1024) {
    static bool done = false;
    if (!done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      done = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)((long long)p.n_tiles * a.B)), dim3(64 * NW), Cf::LDS, s, p);
  return svc::check_launch("resblock16");
}

template <int KS>
int launch_pair16(const svc_resblock_pair_args& a, hipStream_t s) {
  constexpr int H2 = (KS - 1) / 2, BN = 240;
  const int H1 = a.dil1 * H2;
  PairP p;
  p.a = a;
  p.n_tiles = svc::cdiv(a.T, BN);
  auto pitch = [](int w) {
    w = (w + 3) & ~3;
    while ((w & 63) != 16 && (w & 63) != 48) w += 4;   // 4 channel rows x 16 lanes of one operand fetch on disjoint bank groups
    return w;
  };
  p.XW = pitch(256 + 2 * H1);                          // the 256 mid columns of the four waves read up to 2*H1 columns further
  p.MW = pitch(256 + 2 * H2 + 16);
  const size_t lds = (size_t)16 * (p.XW + p.MW) * 4;
  auto kern = mrf_pair16_kernel<KS>;
  if (lds > 64 * 1024) {
    static bool done = false;
    if (!done) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      done = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)((long long)p.n_tiles * a.B)), dim3(256), lds, s, p);
  return svc::check_launch("resblock_pair16");
}

}  // namespace

extern "C" int svc_resblock_pair_f32(const svc_resblock_pair_args* ap, void* stream) {
  SVC_REQUIRE(ap != nullptr, "resblock_pair: null args");
  const svc_resblock_pair_args& a = *ap;
  SVC_REQUIRE(a.x && a.w1 && a.w2 && a.y, "resblock_pair: null tensor");
  SVC_REQUIRE(a.B > 0 && a.T > 0 && a.dil1 >= 1, "resblock_pair: bad shape");
  SVC_REQUIRE(a.C == 16, "resblock_pair: C = %d (the 16-channel stage is built; wider stages use svc_conv1d_f32)", a.C);
  SVC_REQUIRE(a.KS == 3 || a.KS == 7 || a.KS == 11, "resblock_pair: KS = %d not in {3,7,11}", a.KS);
  SVC_REQUIRE(a.CP >= a.C, "resblock_pair: packed row pitch < C");
  SVC_REQUIRE(a.slope >= 0.f && a.slope <= 1.f, "resblock_pair: leaky slope outside [0,1]");
  SVC_REQUIRE(a.x != a.y, "resblock_pair: in-place call (tiles read their neighbours' halo)");
  if ((long long)(240 + 2 * (a.dil1 + 1) * ((a.KS - 1) / 2)) > 512) {
    svc::set_error("resblock_pair: dilation %d too wide for the tile (KS %d): use two svc_conv1d_f32 launches", a.dil1, a.KS);
    return SVC_ERR_UNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  const double flop = 2.0 * 2.0 * a.B * (double)a.C * a.C * a.KS * a.T;
  const double bytes = 4.0 * a.B * (double)a.C * a.T * (a.beta != 0.f ? 3 : 2);
  svc::ProfScope prof(s, "resblock_pair", flop, bytes);
  switch (a.KS) {
    case 3: return launch_pair16<3>(a, s);
    case 7: return launch_pair16<7>(a, s);
    default: return launch_pair16<11>(a, s);
  }
}

extern "C" int svc_resblock16_f32(const svc_resblock16_args* ap, void* stream) {
  SVC_REQUIRE(ap != nullptr, "resblock16: null args");
  const svc_resblock16_args& a = *ap;
  SVC_REQUIRE(a.x && a.y && a.x != a.y, "resblock16: null tensor or in-place call (tiles read their neighbours' halo)");
  SVC_REQUIRE(a.B > 0 && a.T > 0, "resblock16: empty shape");
  if (!(a.n_pairs == 3 && a.dil[0] == 1 && a.dil[1] == 3 && a.dil[2] == 5)) {
    svc::set_error("resblock16: built for the decoder's three pairs with dilations (1, 3, 5)");
    return SVC_ERR_UNSUPPORTED;
  }
  SVC_REQUIRE(a.KS == 3 || a.KS == 7 || a.KS == 11, "resblock16: KS = %d not in {3,7,11}", a.KS);
  SVC_REQUIRE(a.CP >= 16, "resblock16: packed row pitch < 16");
  SVC_REQUIRE(a.slope >= 0.f && a.slope <= 1.f, "resblock16: leaky slope outside [0,1]");
  for (int j = 0; j < a.n_pairs; ++j)
    SVC_REQUIRE(a.w1[j] && a.w2[j] && a.dil[j] >= 1, "resblock16: null weight / bad dilation of pair %d", j);
  hipStream_t s = (hipStream_t)stream;
  const double flop = 2.0 * 2.0 * a.n_pairs * a.B * 16.0 * 16.0 * a.KS * a.T;
  const double bytes = 4.0 * a.B * 16.0 * a.T * (a.beta != 0.f ? 3 : 2);
  svc::ProfScope prof(s, "resblock16", flop, bytes);
  int rc;
  switch (a.KS) {
    case 3: rc = launch_block16<3, 4, 4, 1, 3, 5>(a, s); break;
    case 7: rc = launch_block16<7, 6, 4, 1, 3, 5>(a, s); break;
    default: rc = launch_block16<11, 4, 8, 1, 3, 5>(a, s); break;
  }
  return rc;
}
