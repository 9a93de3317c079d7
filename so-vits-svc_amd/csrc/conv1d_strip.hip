// conv1d_strip.hip — dense Conv1d over LONG sequences (the MRF ResBlock convs of the NSF-HiFiGAN decoder,
// vdecoder/hifigan/models.py:41-67: 32..256 channels, k = 3/7/11, dilation 1/3/5, 7k..441k samples per row).
//
// Why a second kernel next to conv1d_mfma.hip: the round-2 decomposition of the 128-channel k=11 launch (192 us) was
// main loop 156 us (96 % MFMA-busy on the CUs that own two 128x128 tiles) + 27 us epilogue + 9 us prologue, with 431 tiles on
// 512 slots (16 % of the chip idle in the second half of the launch).  The matrix loop was never the problem; the PHASES were:
// every workgroup of a launch is in its epilogue at the same time, so nothing covers it.  This kernel removes the phases
// instead of overlapping them:
//   * ONE workgroup per CU, ONE wave per SIMD, and a wave owns a strip of NT = 7 MFMA tiles along time (32 x 224 outputs, or
//     16 x 112 with the 16x16x4 instruction).  The decoder's lengths are 862 * 2^k: 224-column strips cover them in 247 (248)
//     workgroups — one round of the 256 CUs at 96.5 % — for EVERY channel count (the waves of a workgroup are arranged
//     4x1 / 2x2 / 1x4 over rows x strips for 128 / 64 / 32 channels, 4x1 of the 16-row form for 256 channels).
//   * no LDS epilogue: a lane's accumulator register is 32 (16) consecutive time steps of one output row, so bias /
//     activation / residual / accumulate / store happen straight from the accumulators with 128-byte (64-byte) row segments
//     per half wave.  The residual values are PREFETCHED into the accumulation-register half of the file while the last
//     chunk's MFMAs run — a wave alone on its SIMD has 512 registers — so the epilogue is `add, store` with nothing to wait for.
//   * operands double-buffered in LDS by LDS-DMA, one 1 KiB piece at a time, issued BETWEEN the MFMAs of the running chunk (a
//     piece's scalar + address code fits in the 64-cycle shadow of one fp32 MFMA) instead of as a burst in front of it; a
//     piece's source addresses are computed on the fly (no per-piece register arrays, no setup phase);
//   * the operand reads of the next channel group's first two taps are issued before the current group's last two taps (no
//     exposed LDS round trip at the loop back-edge: 13 % of a k=3 group in round 2), and the leaky-ReLU pre-activation is
//     applied to each B operand as it is read (v_mul + v_med3 in the MFMA shadow) instead of in an LDS pass per chunk.
// Arithmetic: the same fp32 MFMA chain in the same order as conv1d_mfma_kernel (ci-major, tap-minor) and the same epilogue
// expression, so results are bit-identical to that kernel's.
#include "common.h"
#include <algorithm>
#include <cmath>
#include <type_traits>

namespace {

struct StripP {
  svc_conv1d_args a;
  int XW;          // LDS row width of the X tile (floats, multiple of 4)
  int BC;          // input channels per chunk
  int n_t_tiles, n_m_tiles;
  int npw, np;     // LDS-DMA pieces (1 KiB) per chunk: weights, total
  unsigned xw4_magic;   // ceil(2^32 / (XW/4)): slot -> row by multiply-high
};

// One LDS-DMA piece: lane l's 16 B at base + off[l] land at LDS byte address lds_byte + l*16 (wave-uniform LDS base in M0).
// Not tracked by hipcc's s_waitcnt bookkeeping: the kernel counts these itself (strip_vmcnt0 before the barrier).
__device__ __forceinline__ void strip_glds16(unsigned off, const void* base, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(off), "s"(base), "s"(lds_byte)
               : "memory");
}
__device__ __forceinline__ void strip_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int TS> struct AccT;
template <> struct AccT<32> { typedef f32x16 type; };
template <> struct AccT<16> { typedef f32x4 type; };

// TS: MFMA tile edge (32: v_mfma_f32_32x32x2_f32, 16: v_mfma_f32_16x16x4_f32); WM x WN waves (rows x strips), NT tiles per strip.
template <int TS, int WM, int WN, int NT, int KSC, bool PREACT, bool HAS_RES>
__global__ __launch_bounds__(256, 1) void conv1d_strip_kernel(StripP p) {
  static_assert(WM * WN == 4, "one wave per SIMD");
  static_assert(KSC >= 3, "the operand pipeline runs two taps ahead");
  constexpr bool M16 = TS == 16;
  constexpr int KPI = M16 ? 4 : 2;    // input channels consumed per MFMA
  constexpr int NACC = M16 ? 4 : 16;  // accumulator registers per tile
  constexpr int BM = WM * TS, BN = WN * NT * TS, BM4 = BM / 4;
  constexpr int RPP = 64 / BM4;       // weight rows per piece
  typedef typename AccT<TS>::type acc_t;
  const svc_conv1d_args& a = p.a;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int ln = lane & (TS - 1), lk = lane / TS;

  int bid = blockIdx.x;
  const int tt = bid % p.n_t_tiles;
  bid /= p.n_t_tiles;
  const int mtile = bid % p.n_m_tiles;
  const int b = bid / p.n_m_tiles;
  const int t0 = tt * BN, co0 = mtile * BM;
  const int XW = p.XW, XW4 = XW >> 2, BC = p.BC, NPW = p.npw, NP = p.np;

  const float* xb = a.x + (long long)b * a.x_bs;
  const int tin0 = t0 - a.pad_left;
  const int sh = ((tin0 % 4) + 4) % 4;   // tile start rounded down to a 16 B boundary
  const int tin_base = tin0 - sh;

  // ---- LDS-DMA: a chunk is NP pieces of 1 KiB, linear in LDS: pieces 0..NPW-1 the W block [BC*KSC][BM] (RPP whole rows per
  // piece), pieces NPW.. the X block [BC][XW].  Piece pc is fetched by wave pc % 4.  Its source is (chunk base of the tensor, in
  // SGPRs) + (per-lane 32-bit byte offset, computed when the piece is issued).  Every lane reads a VALID address: weight
  // columns >= CoutP, rows / slots past the block are clamped (they feed output rows / LDS words nobody uses); X slots outside
  // [0, Tin) — only the first and last tile of a row have them — are clamped too and zeroed in LDS afterwards by the wave that
  // fetched them (edge_fix, wave-uniform branch).
  const int XF4 = BC * XW4, WROWS = BC * KSC;
  const int buf_f = NP * 256;   // floats per buffer
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  const unsigned wcol = 4u * (unsigned)min(co0 + (lane % BM4) * 4, a.CoutP - 4);
  const char* wsrc = reinterpret_cast<const char*>(a.w);
  const char* xsrc = reinterpret_cast<const char*>(xb);
  const long long wstep = (long long)BC * KSC * a.CoutP * 4, xstep = (long long)BC * a.x_cs * 4;
  auto issue_piece = [&](int pc, int buf) {   // pc, buf wave-uniform
    const unsigned dst = lds_base + ((unsigned)buf * (unsigned)NP + (unsigned)pc) * 1024u;
    if (pc < NPW) {
      const int row = min(pc * RPP + lane / BM4, WROWS - 1);
      strip_glds16(4u * (unsigned)row * (unsigned)a.CoutP + wcol, wsrc, dst);
    } else {
      const int sx = min((pc - NPW) * 64 + lane, XF4 - 1);
      const int r = (int)__umulhi((unsigned)sx, p.xw4_magic), c4 = sx - r * XW4;
      const int tin = min(max(tin_base + c4 * 4, 0), a.Tin - 4);
      strip_glds16(4u * ((unsigned)r * (unsigned)a.x_cs + (unsigned)tin), xsrc, dst);
    }
  };
  const bool edge = tin_base < 0 || tin_base + XW > a.Tin;   // this tile's X block reaches past an end of the sequence
  auto edge_fix = [&](int buf) {
    for (int pc = NPW + ((wave - NPW) & 3); pc < NP; pc += 4) {   // this wave's X pieces
      const int sx = (pc - NPW) * 64 + lane;
      const int r = (int)__umulhi((unsigned)sx, p.xw4_magic), c4 = sx - r * XW4;
      const int tin = tin_base + c4 * 4;
      if (sx < XF4 && (tin < 0 || tin >= a.Tin))
        *reinterpret_cast<float4*>(smem + (buf * NP + pc) * 256 + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  acc_t acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < NACC; ++r) acc[j][r] = 0.f;

  const float ps = a.pre_slope;
  const int dil = a.dil;
  const int n_cc = BC / KPI;

  // One chunk of MFMAs over buffer `buf`.  ISSUE: the pieces of the next chunk (this wave's: pc = wave, wave+4, ...) are issued
  // one per tap, after the tap's first MFMA — under the matrix pipe's 64 busy cycles.
  auto chunk = [&](int buf, auto issue_tag) {
    constexpr bool ISSUE = decltype(issue_tag)::value;
    const float* wl = smem + buf * buf_f + wm * TS + ln + lk * (KSC * BM);
    const float* xl = smem + buf * buf_f + NPW * 256 + wn * (NT * TS) + ln + sh + lk * XW;
    int pc = wave;
    float av[KSC], bv[KSC][NT];
#define SVC_STRIP_LD(k_, wa_, xa_)                                           \
  {                                                                          \
    av[k_] = (wa_)[(k_) * BM];                                               \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) bv[k_][j] = (xa_)[(k_) * dil + j * TS]; \
  }
    SVC_STRIP_LD(0, wl, xl)
    SVC_STRIP_LD(1, wl, xl)
    for (int q = 0; q < n_cc; ++q) {
      const float* wa = wl + q * (KPI * KSC * BM);
      const float* xa = xl + q * (KPI * XW);
      const int qn = min(q + 1, n_cc - 1);
      const float* wnx = wl + qn * (KPI * KSC * BM);
      const float* xnx = xl + qn * (KPI * XW);
#pragma unroll
      for (int k = 0; k < KSC; ++k) {
        // operand reads run two taps ahead of the MFMAs that use them — across the loop back-edge too
        __builtin_amdgcn_sched_barrier(0);
        if (k + 2 < KSC) SVC_STRIP_LD(k + 2, wa, xa)
        else SVC_STRIP_LD(k + 2 - KSC, wnx, xnx)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          float bj = bv[k][j];
          if constexpr (PREACT) bj = __builtin_amdgcn_fmed3f(bj, bj * ps, __builtin_inff());   // max(v, slope*v), 0 <= slope <= 1
          if constexpr (M16) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k], bj, acc[j], 0, 0, 0);
          else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k], bj, acc[j], 0, 0, 0);
          if constexpr (ISSUE) {
            if (j == 0) {
              __builtin_amdgcn_sched_barrier(0);
              if (pc < NP) {
                issue_piece(pc, buf ^ 1);
                pc += 4;
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
    }
#undef SVC_STRIP_LD
    if constexpr (ISSUE) {
      for (; pc < NP; pc += 4) issue_piece(pc, buf ^ 1);   // (not reached for the shapes the launcher admits: taps >= pieces per wave)
    }
  };

  // ---- first chunk in, then [barrier, MFMAs of chunk i with the DMA of chunk i+1 riding along, wait for own pieces]
  for (int pc = wave; pc < NP; pc += 4) issue_piece(pc, 0);
  wsrc += wstep;
  xsrc += xstep;
  strip_vmcnt0();
  if (edge) edge_fix(0);
  int it = 0;
  for (int c0 = BC; c0 < a.Cin; c0 += BC, ++it) {
    __syncthreads();   // chunk `it` has landed for every wave; everyone is done reading the other buffer
    chunk(it & 1, std::true_type{});
    wsrc += wstep;
    xsrc += xstep;
    strip_vmcnt0();    // this wave's pieces of chunk it+1 have landed (they had the whole MFMA loop to do so)
    if (edge) edge_fix((it + 1) & 1);
  }
  __syncthreads();

  // ---- this lane's outputs: row(r) = rowu + rowc(r) + 4*lk, column(j) = colb + j*TS.  Addresses are
  //   (uniform row base in SGPRs: tensor + (rowu + rowc(r)) * channel stride)  +  (per-lane 32-bit byte offset of (4*lk, column j))
  // so the prefetch / epilogue need 7 offset registers per tensor instead of 112 pointers.  Columns past Tout are clamped for
  // loads and masked for stores; a wave whose TS rows lie past Cout (Cout is a multiple of TS) reads row block 0 and stores nothing.
  const int rowu = co0 + wm * TS;
  const bool rows_ok = rowu < a.Cout;
  const int rowl = rows_ok ? rowu : 0;
  const int colb = t0 + wn * (NT * TS) + ln;
  auto rowc = [](int r) { return M16 ? r : (r & 3) + 8 * (r >> 2); };
  float rr[NT][NACC], bc_[NACC];
  float* yb = a.y + (long long)b * a.y_bs;
  const float* resb = a.res ? a.res + (long long)b * a.res_bs : a.x;
  const float* condb = a.cond ? a.cond + (long long)b * a.cond_bs : nullptr;
  unsigned roff[NT], yoff[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const unsigned tc = (unsigned)min(colb + j * TS, a.Tout - 1);
    roff[j] = 4u * ((unsigned)(4 * lk) * (unsigned)a.res_cs + tc);
    yoff[j] = 4u * ((unsigned)(4 * lk) * (unsigned)a.y_cs + tc);
  }
  // Residual prefetch, issued in front of the LAST chunk's MFMAs: global_load_dword <accumulation register>, <lane offset>,
  // <uniform row base>.  Written as asm so that the loads (a) use the SGPR-base form — left to itself the compiler materialises
  // 112 64-bit addresses, spills them, and guards every load with a branch — and (b) land in the accumulation-register half of
  // the file next to the accumulators.  Invisible to hipcc's s_waitcnt bookkeeping: the epilogue waits vmcnt(0) itself.
#pragma unroll
  for (int r = 0; r < NACC; ++r) bc_[r] = 0.f;
  if (a.bias) {
#pragma unroll
    for (int r = 0; r < NACC; ++r) bc_[r] = a.bias[rowl + rowc(r) + 4 * lk];
  }
  if (condb) {
#pragma unroll
    for (int r = 0; r < NACC; ++r) bc_[r] = bc_[r] + condb[(rowl + rowc(r) + 4 * lk) * a.cond_cs];
  }
  if constexpr (HAS_RES) {
#pragma unroll
    for (int r = 0; r < NACC; ++r) {
      const float* rp = resb + (long long)(rowl + rowc(r)) * a.res_cs;   // wave-uniform
#pragma unroll
      for (int j = 0; j < NT; ++j) asm volatile("global_load_dword %0, %1, %2" : "=a"(rr[j][r]) : "v"(roff[j]), "s"(rp));
    }
  }
  chunk(it & 1, std::false_type{});

  // ---- epilogue straight from the accumulators (same expression and order as conv_epilogue's plain path).  The accumulate
  // operand y_old (beta != 0: the last conv of an MRF chain adds into the stage sum) is not prefetched — accumulators plus one
  // prefetched tensor fill the accumulation registers — but fetched here tile by tile: 2 of a stage's 18 launches pay for it.
  const float oslope = a.post_act == SVC_ACT_LRELU ? a.post_slope : 1.f;   // the launcher admits none / leaky-ReLU with 0 <= slope <= 1
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the residual prefetch (landed long ago)
  // Column predicates are per MFMA tile (7 exec-mask regions, not 112); the accumulate / divide form (last conv of an MRF
  // chain: y = (v + beta*y_old) / out_div, IEEE division as in conv_epilogue) is a wave-uniform second copy.
  auto finish = [&](auto accdiv_tag) {
    constexpr bool ACCDIV = decltype(accdiv_tag)::value;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (rows_ok && colb + j * TS < a.Tout) {
        float yo[NACC];
        if constexpr (ACCDIV) {
#pragma unroll
          for (int r = 0; r < NACC; ++r) {
            const float* yp = yb + (long long)(rowl + rowc(r)) * a.y_cs;
            asm volatile("global_load_dword %0, %1, %2" : "=v"(yo[r]) : "v"(yoff[j]), "s"(yp));
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
          float* yp = yb + (long long)(rowl + rowc(r)) * a.y_cs;
          float v = acc[j][r] + bc_[r];
          v = __builtin_amdgcn_fmed3f(v, v * oslope, __builtin_inff());   // == svc_lrelu for 0 <= slope <= 1; slope 1: identity
          if constexpr (HAS_RES) v = v + rr[j][r];
          if constexpr (ACCDIV) {
            v = v + a.beta * yo[r];
            v = v / a.out_div;
          }
          asm volatile("global_store_dword %0, %1, %2" : : "v"(yoff[j]), "v"(v), "s"(yp) : "memory");
        }
      }
    }
  };
  if (a.beta != 0.f || a.out_div != 1.f) finish(std::true_type{});
  else finish(std::false_type{});
}

int g_strip_mode = 1;   // 0: off, 1: auto (svc_debug_set_conv_strip)
int g_strip_launches = 0;   // launches that took this kernel (tests ask through svc_debug_set_conv_strip(-1))

struct StripCfg { int TS, WM, WN; };

template <int TS, int WM, int WN, int KSC, bool PREACT, bool HAS_RES>
int strip_launch(const svc_conv1d_args& a, hipStream_t s) {
  constexpr int NT = 7, KPI = TS == 16 ? 4 : 2, BM = WM * TS, BN = WN * NT * TS;
  StripP p;
  memset(&p, 0, sizeof(p));
  p.a = a;
  int xw = BN + (a.KS - 1) * a.dil + 3;
  xw = (xw + 3) & ~3;
  if (TS == 16) {   // consecutive channel rows on disjoint bank halves for the 16-lane groups of a B read
    while ((xw & 31) != 16) xw += 4;
  }
  p.XW = xw;
  // largest chunk (power-of-two multiple of KPI dividing Cin) whose two buffers fit 160 KiB and whose pieces per wave do not
  // outnumber the chunk's taps (one piece rides on each tap of the previous chunk)
  int bc = 0, npw = 0, np = 0;
  for (int c = 64; c >= KPI; c >>= 1) {
    if (c > a.Cin || a.Cin % c) continue;
    const int w_pieces = svc::cdiv(c * a.KS * (BM / 4), 64), x_pieces = svc::cdiv(c * (xw / 4), 64);
    const int n = w_pieces + x_pieces;
    if ((size_t)2 * n * 1024 <= 160 * 1024 && svc::cdiv(n, 4) <= (c / KPI) * a.KS) { bc = c; npw = w_pieces; np = n; break; }
  }
  if (bc == 0) return 1;
  p.BC = bc;
  p.npw = npw;
  p.np = np;
  const unsigned xw4 = (unsigned)(xw / 4);
  p.xw4_magic = (unsigned)((0x100000000ull + xw4 - 1) / xw4);
  p.n_t_tiles = svc::cdiv(a.Tout, BN);
  p.n_m_tiles = svc::cdiv(a.Cout, BM);
  const long long nblk = (long long)p.n_t_tiles * p.n_m_tiles * a.B;
  const size_t lds = (size_t)2 * np * 1024;
  auto kd = conv1d_strip_kernel<TS, WM, WN, NT, KSC, PREACT, HAS_RES>;
  static bool done = false;
  if (!done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done = true;
  }
  hipLaunchKernelGGL(kd, dim3((unsigned)nblk), dim3(256), lds, s, p);
  ++g_strip_launches;
  return svc::check_launch("conv1d_strip");
}

template <int TS, int WM, int WN, int KSC>
int strip_launch_mode(const svc_conv1d_args& a, hipStream_t s) {
  const bool pre = a.pre_slope != 1.f, res = a.res_mode != 0;
  if (pre && !res) return strip_launch<TS, WM, WN, KSC, true, false>(a, s);    // first conv of a ResBlock1 pair
  if (!pre && res) return strip_launch<TS, WM, WN, KSC, false, true>(a, s);    // second conv (its input was activated by the first's epilogue)
  if (pre && res) return strip_launch<TS, WM, WN, KSC, true, true>(a, s);      // ResBlock2 / un-fused second activation
  return 1;
}

template <int TS, int WM, int WN>
int strip_launch_ks(const svc_conv1d_args& a, hipStream_t s) {
  switch (a.KS) {
    case 3: return strip_launch_mode<TS, WM, WN, 3>(a, s);
    case 7: return strip_launch_mode<TS, WM, WN, 7>(a, s);
    case 11: return strip_launch_mode<TS, WM, WN, 11>(a, s);
    default: return 1;
  }
}

}  // namespace

extern "C" int svc_debug_set_conv_strip(int mode) {
  if (mode < 0) return g_strip_launches;
  g_strip_mode = mode;
  return SVC_OK;
}

namespace svc {

// Returns 1 when the shape is not one for this kernel (the caller then runs conv1d_mfma_kernel), else the launch status.
// mode 1 (default): take the strip kernel when one of its four wave arrangements covers the launch in whole rounds of the
// chip at >= 85 % (useful tile area / (rounds * 256 CUs * tile area)); modes 2..5 force arrangement 0..3 (tests / tuning).
int conv1d_strip_try(const svc_conv1d_args& a, hipStream_t s) {
  if (g_strip_mode == 0) return 1;
  if (a.epi != SVC_EPI_PLAIN || a.n_phase != 1 || a.y_ts != 1 || a.y_t0 != 0 || a.mask || a.premask) return 1;
  if (a.cond && a.cond_ts != 0) return 1;
  if (!(a.res_mode == 0 || a.res_mode == 1)) return 1;
  if (!(a.post_act == SVC_ACT_NONE || (a.post_act == SVC_ACT_LRELU && a.post_slope >= 0.f && a.post_slope <= 1.f))) return 1;
  if (!(a.KS == 3 || a.KS == 7 || a.KS == 11) || (a.KS - 1) * a.dil > 50) return 1;
  if (!(a.pre_slope >= 0.f && a.pre_slope <= 1.f)) return 1;
  const bool xvec = (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (a.x_bs % 4) == 0 && (a.x_cs % 4) == 0 && (a.Tin % 4) == 0;
  if (!xvec || (a.Cin % 4) != 0 || a.x_cs < 0 || (a.Cout % 32) != 0 || a.Tin < 4) return 1;
  if (a.x_cs >= (1ll << 23) || (long long)a.CoutP * a.KS * 4 * 64 >= (1ll << 31)) return 1;    // 32-bit byte offsets inside a chunk
  if (a.y_cs < 0 || a.y_cs >= (1ll << 24) || a.res_cs < 0 || a.res_cs >= (1ll << 24)) return 1;   // ... and of 16 output rows
  static const StripCfg cfgs[4] = {{32, 4, 1}, {32, 2, 2}, {32, 1, 4}, {16, 4, 1}};
  int best = -1;
  double best_eff = 0.0;
  for (int i = 0; i < 4; ++i) {
    const int BM = cfgs[i].WM * cfgs[i].TS, BN = cfgs[i].WN * 7 * cfgs[i].TS;
    const double n = (double)svc::cdiv(a.Cout, BM) * svc::cdiv(a.Tout, BN) * a.B;
    const double eff = ((double)a.Cout * a.Tout * a.B) / (std::ceil(n / 256.0) * 256.0 * BM * BN);
    if (g_strip_mode >= 2) {
      if (g_strip_mode - 2 == i) { best = i; best_eff = 1.0; }
    } else if (n >= 200 && eff > best_eff) {
      best = i;
      best_eff = eff;
    }
  }
  if (best < 0 || best_eff < 0.85) return 1;
  switch (best) {
    case 0: return strip_launch_ks<32, 4, 1>(a, s);
    case 1: return strip_launch_ks<32, 2, 2>(a, s);
    case 2: return strip_launch_ks<32, 1, 4>(a, s);
    default: return strip_launch_ks<16, 4, 1>(a, s);
  }
}

}  // namespace svc
