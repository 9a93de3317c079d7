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
#include <vector>

namespace {

struct StripP {
  svc_conv1d_args a;
  int XW;          // LDS row width of the X tile (floats, multiple of 4)
  int BC;          // input channels per chunk
  int n_t_tiles, n_m_tiles;
  int npw;         // 1 KiB LDS-DMA pieces of a chunk's weight block
  int buf_f;       // floats per LDS buffer (W block + X block)
  int dbg;         // timing experiments (results are then garbage): 1 no output stores, 2 no MFMA loop, 4 no residual prefetch
};
#ifdef SVC_TIMING_DEBUG       // the shipped library has no way to produce garbage: the flags are compiled out
#define STRIP_DBG(p) ((p).dbg)
#else
#define STRIP_DBG(p) 0
#endif

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
// WPS waves per SIMD share a strip: 1 = one wave owns all NT tiles (a wave alone on its SIMD pays for every non-MFMA instruction
// of its stream: ds_reads, DMA issue, addressing — measured 1.2..1.6x the MFMA time, profiles/r03a_*, r03b_*); 2 = the strip's
// tiles are split 4 + 3 between two waves of the same SIMD (waves w and w + 4), so one wave's operand reads / DMA issue /
// epilogue stores run under the other's MFMAs.
// SPLITK (WM = WN = 1, WPS = 1): ONE strip per workgroup, every wave computes all NT tiles over its own quarter of the input
// channels and the four partial strips meet in LDS — for stages with few columns and many channels (256 channels x 6896
// samples: 32-row strips give 8 x 31 = 248 workgroups where the row-parallel arrangements give 62).
template <int TS, int WM, int WN, int NT, int KSC, bool PREACT, bool HAS_RES, int WPS, bool SPLITK = false>
__device__ __forceinline__ void conv1d_strip_body(const StripP& p, int bid) {
  static_assert(SPLITK ? (WM == 1 && WN == 1 && WPS == 1 && TS == 32) : WM * WN == 4, "one strip per SIMD");
  static_assert(WPS == 1 || (WPS == 2 && NT == 7), "two waves per SIMD split a 7-tile strip 4 + 3");
  constexpr int NWV = 4 * WPS;
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
  const int strip = wave & 3, half = wave >> 2;   // waves w and w + 4 run on the same SIMD (cyclic wave -> SIMD placement)
  const int wm = SPLITK ? 0 : strip / WN, wn = SPLITK ? 0 : strip % WN;
  const int ln = lane & (TS - 1), lk = lane / TS;

  const int tt = bid % p.n_t_tiles;
  bid /= p.n_t_tiles;
  const int mtile = bid % p.n_m_tiles;
  const int b = bid / p.n_m_tiles;
  const int t0 = tt * BN, co0 = mtile * BM;
  const int XW = p.XW, XW4 = XW >> 2, BC = p.BC, NPW = p.npw;

  const float* xb = a.x + (long long)b * a.x_bs;
  const int tin0 = t0 - a.pad_left;
  const int sh = ((tin0 % 4) + 4) % 4;   // tile start rounded down to a 16 B boundary
  const int tin_base = tin0 - sh;

  // ---- LDS-DMA.  A chunk in LDS = W block [BC*KSC][BM] (NPW pieces of 1 KiB = RPP whole rows each) then X block [BC][XW].
  // Fetching wave w (of four) takes the W pieces pc = w, w+4, ... and the X rows r = w, w+4, ...; an X row is PPR pieces, piece pp covering the
  // float4 columns [min(64*pp, XW4-64), +64) — the last piece overlaps its neighbour instead of running past the row (XW4 >= 64).
  // A piece's source is (uniform base: tensor chunk base + piece / row offset, all scalar arithmetic) + (a per-lane byte offset:
  // constant for W pieces, 4 VALU for X pieces), so issuing a piece costs ~10 SALU + 1 VMEM: it hides under one fp32 MFMA.
  // Every lane reads a VALID address: weight columns >= CoutP are clamped (they feed output rows nobody stores), X columns
  // outside [0, Tin) — only the first and last tile of a row have them — are clamped and zeroed in LDS afterwards by the wave
  // that fetched them, in the same in-place pass that applies the leaky-ReLU pre-activation (fix_rows).
  const int PPR = (XW4 + 63) >> 6;
  const int wfl = NPW * 256;                     // floats of the W block
  const int buf_f = p.buf_f;                     // floats per buffer
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  const unsigned woff = 4u * ((unsigned)(lane / BM4) * (unsigned)a.CoutP + (unsigned)min(co0 + (lane % BM4) * 4, a.CoutP - 4));
  const char* wsrc = reinterpret_cast<const char*>(a.w);
  const char* xsrc = reinterpret_cast<const char*>(xb);
  const long long wstep = (long long)BC * KSC * a.CoutP * 4, xstep = (long long)BC * a.x_cs * 4;
  const long long wpiece = (long long)RPP * a.CoutP * 4, xrow = a.x_cs * 4;
  // Who fetches: with one wave per SIMD, all four; with two, ONLY the second wave of each SIMD (waves 4..7, the 3-tile halves).
  // Their partner (4 tiles) goes from the barrier straight to its MFMAs and keeps the matrix pipe busy while the burst is issued;
  // the 3-tile wave has one MFMA slot per tap of slack (44 x 64 cycles per chunk at k = 11), more than the burst costs.
  const bool is_dma = WPS == 1 || half == 1;
  const int dw = strip;   // index among the four fetching waves
  const int nw_mine = is_dma ? max(0, (NPW - dw + 3) / 4) : 0;
  const int steps_mine = is_dma ? nw_mine + max(0, (BC - dw + 3) / 4) * PPR : 0;
  // Issue state of the chunk being fetched, all wave-uniform and advanced incrementally (no multiplies per piece):
  int si = 0, xpp = 0;
  const char *wptr = nullptr, *xptr = nullptr;
  unsigned wdst = 0, xdst = 0;
  auto issue_begin = [&](int buf) {
    const unsigned bufb = lds_base + (unsigned)buf * (unsigned)buf_f * 4u;
    si = 0; xpp = 0;
    wptr = wsrc + dw * wpiece;
    xptr = xsrc + dw * xrow;
    wdst = bufb + (unsigned)dw * 1024u;
    xdst = bufb + 4u * (unsigned)(wfl + dw * XW);
  };
  // One piece in three parts, so that each part fits under ONE fp32 MFMA (64 cycles) when they are placed between the MFMAs of
  // a tap: A = addresses (branch-free selects between the W and the X form), B = the DMA instruction, C = state update.
  bool isw = false;
  const char* ibase = nullptr;
  unsigned idst = 0, ioff = 0;
  auto issue_a = [&]() {
    isw = si < nw_mine;
    const int c4s = min(64 * xpp, XW4 - 64);
    const int tin = min(max(tin_base + 4 * (c4s + lane), 0), a.Tin - 4);
    ibase = isw ? wptr : xptr;
    idst = isw ? wdst : xdst + 16u * (unsigned)c4s;
    ioff = isw ? woff : 4u * (unsigned)tin;
    asm volatile("" ::"s"(ibase), "s"(idst), "v"(ioff));   // materialise here (the compiler would sink all of it next to the DMA)
  };
  auto issue_b = [&]() {
    if (si < steps_mine) strip_glds16(ioff, ibase, idst);
  };
  bool wrap = false;
  auto issue_c = [&]() {
    wrap = !isw && xpp + 1 == PPR;
    wptr += isw ? 4 * wpiece : 0;
    wdst += isw ? 4096u : 0u;
    xpp = isw ? xpp : (wrap ? 0 : xpp + 1);
    asm volatile("" ::"s"(wptr), "s"(wdst), "s"(xpp));
  };
  auto issue_d = [&]() {
    xptr += wrap ? 4 * xrow : 0;
    xdst += wrap ? 16u * (unsigned)XW : 0u;
    ++si;
    asm volatile("" ::"s"(xptr), "s"(xdst), "s"(si));
  };
  const bool edge = tin_base < 0 || tin_base + XW > a.Tin;   // this tile's X block reaches past an end of the sequence
  const float ps = a.pre_slope;
  auto fix_rows = [&](int buf) {   // own rows, in place: zero padding (edge tiles) and the pre-activation max(v, slope*v)
    if (!(PREACT || edge) || !is_dma) return;
    for (int r = dw; r < BC; r += 4) {
      float* row = smem + buf * buf_f + wfl + r * XW;
      for (int c4 = lane; c4 < XW4; c4 += 64) {
        const int tin = tin_base + c4 * 4;
        float4 v = *reinterpret_cast<float4*>(row + c4 * 4);
        if (tin < 0 || tin >= a.Tin) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (PREACT) {
          v.x = __builtin_amdgcn_fmed3f(v.x, v.x * ps, __builtin_inff());
          v.y = __builtin_amdgcn_fmed3f(v.y, v.y * ps, __builtin_inff());
          v.z = __builtin_amdgcn_fmed3f(v.z, v.z * ps, __builtin_inff());
          v.w = __builtin_amdgcn_fmed3f(v.w, v.w * ps, __builtin_inff());
        }
        *reinterpret_cast<float4*>(row + c4 * 4) = v;
      }
    }
  };

  // ---- everything from here on is per wave: NTW tiles starting at tile J0 of the strip
  auto body = [&](auto ntw_tag, auto j0_tag) {
    constexpr int NTW = decltype(ntw_tag)::value, J0 = decltype(j0_tag)::value;
    acc_t acc[NTW];
  #pragma unroll
    for (int j = 0; j < NTW; ++j)
  #pragma unroll
      for (int r = 0; r < NACC; ++r) acc[j][r] = 0.f;

    const int dil = a.dil;
    const int n_cc = BC / KPI;
    // split-K: wave `strip` reduces the channel groups [qa, qb) of every chunk (n_cc is a multiple of 4: launcher)
    const int qa = SPLITK ? strip * (n_cc >> 2) : 0, qb = SPLITK ? qa + (n_cc >> 2) : n_cc;

    // MFMAs over channel groups [q0, q1) of buffer `buf`.  Operand reads run two taps ahead of their MFMAs, across the loop
    // back-edge too (the chunk's last group looks ahead at itself: re-read, unused).
    float av[KSC], bv[KSC][NTW];
  #define SVC_STRIP_LD(k_, wa_, xa_)                                           \
    {                                                                          \
      av[k_] = (wa_)[(k_) * BM];                                               \
      _Pragma("unroll") for (int j = 0; j < NTW; ++j) bv[k_][j] = (xa_)[(k_) * dil + j * TS]; \
    }
    auto groups = [&](int buf, int q0, int q1) {
      if (STRIP_DBG(p) & 2) return;
      const float* wl = smem + buf * buf_f + wm * TS + ln + lk * (KSC * BM);
      const float* xl = smem + buf * buf_f + wfl + wn * (NT * TS) + J0 * TS + ln + sh + lk * XW;
      for (int q = q0; q < q1; ++q) {
        const float* wa = wl + q * (KPI * KSC * BM);
        const float* xa = xl + q * (KPI * XW);
        const int qn = min(q + 1, qb - 1);
        const float* wnx = wl + qn * (KPI * KSC * BM);
        const float* xnx = xl + qn * (KPI * XW);
  #pragma unroll
        for (int k = 0; k < KSC; ++k) {
          __builtin_amdgcn_sched_barrier(0);
          if (k + 2 < KSC) SVC_STRIP_LD(k + 2, wa, xa)
          else SVC_STRIP_LD(k + 2 - KSC, wnx, xnx)
          __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
          for (int j = 0; j < NTW; ++j) {
            if constexpr (M16) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k], bv[k][j], acc[j], 0, 0, 0);
            else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k], bv[k][j], acc[j], 0, 0, 0);
          }
        }
      }
    };
    auto first_reads = [&](int buf) {
      const float* wl = smem + buf * buf_f + wm * TS + ln + lk * (KSC * BM) + qa * (KPI * KSC * BM);
      const float* xl = smem + buf * buf_f + wfl + wn * (NT * TS) + J0 * TS + ln + sh + lk * XW + qa * (KPI * XW);
      SVC_STRIP_LD(0, wl, xl)
      SVC_STRIP_LD(1, wl, xl)
    };

    // ---- first chunk in, then [barrier, DMA burst of chunk i+1, MFMAs of chunk i, wait for own pieces, fix own rows]
    issue_begin(0);
    while (si < steps_mine) { issue_a(); issue_b(); issue_c(); issue_d(); }
    wsrc += wstep;
    xsrc += xstep;
    strip_vmcnt0();
    fix_rows(0);
    int it = 0;
    for (int c0 = BC; c0 < a.Cin; c0 += BC, ++it) {
      __syncthreads();   // chunk `it` has landed (and is activated) for every wave; everyone is done reading the other buffer
      // the next chunk's pieces go out as ONE burst, before this wave has any operand read in flight: an LDS-DMA instruction
      // issued behind pending ds_reads waits for them (measured: one piece per tap inside the MFMA stream cost ~75 cycles of
      // every SIMD per piece, 1.3x the MFMA time at 128 channels, 1.8x at 256 — profiles/r03b_*, r03c_*); back to back after the
      // barrier a piece costs ~60 cycles of the issuing wave only
      issue_begin((it & 1) ^ 1);
      while (si < steps_mine) { issue_a(); issue_b(); issue_c(); issue_d(); }
      first_reads(it & 1);
      groups(it & 1, qa, qb);
      wsrc += wstep;
      xsrc += xstep;
      strip_vmcnt0();    // this wave's pieces of chunk it+1 have landed (they had the whole MFMA loop to do so)
      fix_rows((it + 1) & 1);
    }
    __syncthreads();

    // ---- this lane's outputs: row(r) = rowu + rowc(r) + 4*lk, column(j) = colb + j*TS.  Addresses are
    //   (uniform row base in SGPRs: tensor + (rowu + rowc(r)) * channel stride)  +  (per-lane 32-bit byte offset of (4*lk, column j))
    // so the prefetch / epilogue need 7 offset registers per tensor instead of 112 pointers.  Columns past Tout are clamped for
    // loads and masked for stores; a wave whose TS rows lie past Cout (Cout is a multiple of TS) reads row block 0 and stores nothing.
    const int rowu = co0 + wm * TS;
    const bool rows_ok = rowu < a.Cout;
    const int rowl = rows_ok ? rowu : 0;
    const int colb = t0 + wn * (NT * TS) + J0 * TS + ln;
    auto rowc = [](int r) { return M16 ? r : (r & 3) + 8 * (r >> 2); };
    // tiles this wave finishes: all of its own (NTW), or — split-K — tiles strip, strip + 4 of the reduced strip
    constexpr int NE = SPLITK ? (NT + 3) / 4 : NTW;
    auto jcol = [&](int j) { return SPLITK ? strip + 4 * j : j; };
    float rr[NE][NACC], bc_[NACC];
    float* yb = a.y + (long long)b * a.y_bs;
    const float* resb = a.res ? a.res + (long long)b * a.res_bs : a.x;
    const float* condb = a.cond ? a.cond + (long long)b * a.cond_bs : nullptr;
    unsigned roff[NE], yoff[NE];
  #pragma unroll
    for (int j = 0; j < NE; ++j) {
      const unsigned tc = (unsigned)min(colb + jcol(j) * TS, a.Tout - 1);
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
      if (!(STRIP_DBG(p) & 4)) {
  #pragma unroll
      for (int r = 0; r < NACC; ++r) {
        const float* rp = resb + (long long)(rowl + rowc(r)) * a.res_cs;   // wave-uniform
  #pragma unroll
        for (int j = 0; j < NE; ++j) asm volatile("global_load_dword %0, %1, %2" : "=a"(rr[j][r]) : "v"(roff[j]), "s"(rp));
      }
      }
    }
    first_reads(it & 1);
    groups(it & 1, qa, qb);
  #undef SVC_STRIP_LD

    // ---- epilogue straight from the accumulators (same expression and order as conv_epilogue's plain path).  The accumulate
    // operand y_old (beta != 0: the last conv of an MRF chain adds into the stage sum) is not prefetched — accumulators plus one
    // prefetched tensor fill the accumulation registers — but fetched here tile by tile: 2 of a stage's 18 launches pay for it.
    const float oslope = a.post_act == SVC_ACT_LRELU ? a.post_slope : 1.f;   // the launcher admits none / leaky-ReLU with 0 <= slope <= 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the residual prefetch (landed long ago)
    if constexpr (HAS_RES) {   // ... and tie every later use of rr to this point (the asm loads are invisible to the scheduler)
#pragma unroll
      for (int j = 0; j < NE; ++j)
#pragma unroll
        for (int r = 0; r < NACC; ++r) asm volatile("" : "+a"(rr[j][r]));
    }
    if constexpr (SPLITK) {
      // the four partial strips meet in LDS ([wave][tile][reg][lane]: every access is 64 consecutive floats), summed in the
      // fixed order wave 0..3 by the wave that finishes the tile; the sums replace acc[0 .. NE)
      __syncthreads();                                   // everyone is done reading operands: the buffers are free
      float* red = smem;
  #pragma unroll
      for (int j = 0; j < NTW; ++j)
  #pragma unroll
        for (int r = 0; r < NACC; ++r) red[((strip * NT + j) * NACC + r) * 64 + lane] = acc[j][r];
      __syncthreads();
  #pragma unroll
      for (int j = 0; j < NE; ++j) {
        const int jt = min(jcol(j), NT - 1);
  #pragma unroll
        for (int r = 0; r < NACC; ++r) {
          float v = red[((0 * NT + jt) * NACC + r) * 64 + lane];
  #pragma unroll
          for (int w = 1; w < 4; ++w) v += red[((w * NT + jt) * NACC + r) * 64 + lane];
          acc[j][r] = v;
        }
      }
    }
    // Column predicates are per MFMA tile (7 exec-mask regions, not 112); the accumulate / divide form (last conv of an MRF
    // chain: y = (v + beta*y_old) / out_div, IEEE division as in conv_epilogue) is a wave-uniform second copy.
    auto finish = [&](auto accdiv_tag) {
      constexpr bool ACCDIV = decltype(accdiv_tag)::value;
  #pragma unroll
      for (int j = 0; j < NE; ++j) {
        if (rows_ok && jcol(j) < NT && colb + jcol(j) * TS < a.Tout) {
          float yo[NACC];
          if constexpr (ACCDIV) {
  #pragma unroll
            for (int r = 0; r < NACC; ++r) {
              const float* yp = yb + (long long)(rowl + rowc(r)) * a.y_cs;
              asm volatile("global_load_dword %0, %1, %2" : "=v"(yo[r]) : "v"(yoff[j]), "s"(yp));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < NACC; ++r) asm volatile("" : "+v"(yo[r]));   // uses of yo stay behind the wait
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
            if (!(STRIP_DBG(p) & 1) || r == 0) asm volatile("global_store_dword %0, %1, %2" : : "v"(yoff[j]), "v"(v), "s"(yp) : "memory");
          }
        }
      }
    };
    if (a.beta != 0.f || a.out_div != 1.f) finish(std::true_type{});
    else finish(std::false_type{});

  };
  if constexpr (WPS == 1) {
    body(std::integral_constant<int, NT>{}, std::integral_constant<int, 0>{});
  } else {
    if (half == 0) body(std::integral_constant<int, 4>{}, std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 3>{}, std::integral_constant<int, 4>{});
  }
}

template <int TS, int WM, int WN, int NT, int KSC, bool PREACT, bool HAS_RES, int WPS, bool SPLITK = false>
__global__ __launch_bounds__(256 * WPS, WPS) void conv1d_strip_kernel(StripP p) {
  conv1d_strip_body<TS, WM, WN, NT, KSC, PREACT, HAS_RES, WPS, SPLITK>(p, blockIdx.x);
}

int g_strip_mode = 1;   // 0: off, 1: auto (svc_debug_set_conv_strip)
int g_strip_dbg = 0;        // StripP.dbg (svc_debug_set_conv_strip(1000 * dbg + mode))
int g_strip_launches = 0;   // launches that took this kernel (tests ask through svc_debug_set_conv_strip(-1))

struct StripCfg { int TS, WM, WN; };

template <int TS, int WM, int WN, int KSC, bool PREACT, bool HAS_RES, int WPS, bool SPLITK = false>
int strip_launch(const svc_conv1d_args& a, hipStream_t s) {
  constexpr int NT = 7, KPI = TS == 16 ? 4 : 2, BM = WM * TS, BN = WN * NT * TS;
  constexpr int CMULT = SPLITK ? 4 * KPI : KPI;   // split-K: whole channel groups per wave
  StripP p;
  memset(&p, 0, sizeof(p));
  p.a = a;
  int xw = BN + (a.KS - 1) * a.dil + 3;
  xw = std::max((xw + 3) & ~3, 256);   // >= 64 float4 per row: the last DMA piece of a row overlaps instead of overrunning
  if (TS == 16) {   // consecutive channel rows on disjoint bank halves for the 16-lane groups of a B read
    while ((xw & 31) != 16) xw += 4;
  }
  p.XW = xw;
  // largest chunk (power-of-two multiple of KPI dividing Cin) whose weight block is whole pieces and whose two buffers fit 160 KiB
  constexpr int RPP = 64 / (BM / 4);
  int bc = 0, npw = 0, buf_f = 0;
  for (int c = 64; c >= CMULT; c >>= 1) {
    if (c > a.Cin || a.Cin % c || (c * a.KS) % RPP) continue;
    const int w_pieces = c * a.KS / RPP;
    const int f = w_pieces * 256 + c * xw;
    if ((size_t)2 * f * 4 <= 160 * 1024) { bc = c; npw = w_pieces; buf_f = f; break; }
  }
  if (bc == 0) return 1;
  p.BC = bc;
  p.npw = npw;
  p.buf_f = buf_f;
  p.dbg = g_strip_dbg;
  p.n_t_tiles = svc::cdiv(a.Tout, BN);
  p.n_m_tiles = svc::cdiv(a.Cout, BM);
  const long long nblk = (long long)p.n_t_tiles * p.n_m_tiles * a.B;
  const size_t lds = std::max((size_t)2 * buf_f * 4, SPLITK ? (size_t)4 * NT * 16 * 64 * 4 : (size_t)0);
  auto kd = conv1d_strip_kernel<TS, WM, WN, NT, KSC, PREACT, HAS_RES, WPS, SPLITK>;
  static bool done = false;
  if (!done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done = true;
  }
  hipLaunchKernelGGL(kd, dim3((unsigned)nblk), dim3(256 * WPS), lds, s, p);
  ++g_strip_launches;
  return svc::check_launch("conv1d_strip");
}

template <int TS, int WM, int WN, int KSC, int WPS>
int strip_launch_mode(const svc_conv1d_args& a, hipStream_t s) {
  const bool pre = a.pre_slope != 1.f, res = a.res_mode != 0;
  if (pre && !res) return strip_launch<TS, WM, WN, KSC, true, false, WPS>(a, s);    // first conv of a ResBlock1 pair
  if (!pre && res) return strip_launch<TS, WM, WN, KSC, false, true, WPS>(a, s);    // second conv (its input was activated by the first's epilogue)
  if (pre && res) return strip_launch<TS, WM, WN, KSC, true, true, WPS>(a, s);      // ResBlock2 / un-fused second activation
  return 1;
}

template <int KSC>
int strip_launch_splitk_mode(const svc_conv1d_args& a, hipStream_t s) {
  const bool pre = a.pre_slope != 1.f, res = a.res_mode != 0;
  if (pre && !res) return strip_launch<32, 1, 1, KSC, true, false, 1, true>(a, s);
  if (!pre && res) return strip_launch<32, 1, 1, KSC, false, true, 1, true>(a, s);
  if (pre && res) return strip_launch<32, 1, 1, KSC, true, true, 1, true>(a, s);
  return 1;
}

int strip_launch_splitk(const svc_conv1d_args& a, hipStream_t s) {
  switch (a.KS) {
    case 3: return strip_launch_splitk_mode<3>(a, s);
    case 7: return strip_launch_splitk_mode<7>(a, s);
    case 11: return strip_launch_splitk_mode<11>(a, s);
    default: return 1;
  }
}

template <int TS, int WM, int WN>
int strip_launch_ks(const svc_conv1d_args& a, hipStream_t s, int wps) {
  if (wps == 1) {
    switch (a.KS) {
      case 3: return strip_launch_mode<TS, WM, WN, 3, 1>(a, s);
      case 7: return strip_launch_mode<TS, WM, WN, 7, 1>(a, s);
      case 11: return strip_launch_mode<TS, WM, WN, 11, 1>(a, s);
      default: return 1;
    }
  }
  switch (a.KS) {
    case 3: return strip_launch_mode<TS, WM, WN, 3, 2>(a, s);
    case 7: return strip_launch_mode<TS, WM, WN, 7, 2>(a, s);
    case 11: return strip_launch_mode<TS, WM, WN, 11, 2>(a, s);
    default: return 1;
  }
}

}  // namespace

extern "C" int svc_debug_set_conv_strip(int mode) {
  if (mode < 0) return g_strip_launches;
#ifdef SVC_TIMING_DEBUG
  g_strip_dbg = mode / 1000;      // timing experiments only (garbage results): see StripP.dbg
#else
  SVC_REQUIRE(mode < 1000, "svc_debug_set_conv_strip: the timing-decomposition modes (>= 1000: results are garbage) exist only in builds with -DSVC_TIMING_DEBUG");
#endif
  g_strip_mode = mode % 1000;
  return SVC_OK;
}

namespace svc {

// Returns 1 when the shape is not one for this kernel (the caller then runs conv1d_mfma_kernel), else the launch status
// (`dry`: 0 without launching where it would launch).
// mode 1 (default): take the strip kernel when one of its four wave arrangements covers the launch in whole rounds of the
// chip at >= 85 % (useful tile area / (rounds * 256 CUs * tile area)); modes 2..6 force arrangement 0..4 (tests / tuning; 4 = split-K).
// mode + 10: the same with ONE wave per SIMD (the first form of this kernel, kept for A/B).
int conv1d_strip_try(const svc_conv1d_args& a, hipStream_t s, bool dry) {
  if (g_strip_mode == 0) return 1;
  const int wps = g_strip_mode >= 10 ? 1 : 2, mode = g_strip_mode % 10;
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
  static const StripCfg cfgs[5] = {{32, 4, 1}, {32, 2, 2}, {32, 1, 4}, {16, 4, 1}, {32, 1, 1}};   // the last one: split-K
  int best = -1;
  double best_eff = 0.0;
  for (int i = 0; i < 5; ++i) {
    const int BM = cfgs[i].WM * cfgs[i].TS, BN = cfgs[i].WN * 7 * cfgs[i].TS;
    const double n = (double)svc::cdiv(a.Cout, BM) * svc::cdiv(a.Tout, BN) * a.B;
    const double eff = ((double)a.Cout * a.Tout * a.B) / (std::ceil(n / 256.0) * 256.0 * BM * BN);
    if (mode >= 2) {
      if (mode - 2 == i) { best = i; best_eff = 1.0; }
    } else if (i < 3 && n >= 200 && eff > best_eff) {   // (the 16x16x4 and split-K forms lose to the tiled kernel at 256 channels — 127 vs
                                                         //  112..123 us at k = 11, profiles/r03v_* — and stay forced-mode only)
      best = i;
      best_eff = eff;
    }
  }
  if (best < 0 || best_eff < 0.85) return 1;
  if (dry) return 0;       // (svc_conv1d_wants_d4: this kernel would take the launch)
  switch (best) {
    case 0: return strip_launch_ks<32, 4, 1>(a, s, wps);
    case 1: return strip_launch_ks<32, 2, 2>(a, s, wps);
    case 2: return strip_launch_ks<32, 1, 4>(a, s, wps);
    case 4: return strip_launch_splitk(a, s);
    default: return strip_launch_ks<16, 4, 1>(a, s, wps);
  }
}

}  // namespace svc
