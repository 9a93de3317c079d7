// conv1d_wgrad.hip — weight gradient of a dense (stride-1, dilated) Conv1d on the fp32 matrix pipe:
//     G[ca, cb, k] = sum_{b,t} A[b, ca, t] * Bm[b, cb, t + k*dil - pad]          (t in [0,TA), Bm index in [0,TB))
// With A = dy [B,Cout,Tout] and Bm = x [B,Cin,Tin] this is dW of every nn.Conv1d on the training path
// (backward of the convs cited in conv1d_mfma.hip); with A = x and Bm = the phase-decimated dy it is dW of the
// polyphase ConvTranspose1d (vdecoder/hifigan/models.py:340-342); strided / period discriminator convs
// (models.py:171-177) go through the same kernel after svc_decimate_f32.
//
// GEMM view per tap: G_k = A (Ca x N) * Bm_k^T (N x Cb), N = (b,t) — the reduction runs over TIME.  A workgroup owns a
// 128 (ca) x 64 (cb) x NK (taps) block of G and a contiguous range of time tiles; its 4 waves (2 x 2) each hold a
// 64 x 32 x NK accumulator block (2*NK MFMA tiles), so one staged time tile feeds 8*NK MFMA tiles.  Tiles staged in
// LDS: As[128][TT] and Bs[64][TT + (NK-1)*dil] with an ODD row pitch — an MFMA operand fetch has its 32 lanes on 32
// different channel rows at the same time step, which an odd pitch spreads over 32 banks.  The global loads of tile
// i+1 are issued into registers before the MFMA loop over tile i and written to LDS after it (same software pipeline
// as conv1d_mfma).  KS > 5 is split into tap groups that run as separate workgroups (blockIdx.x carries the group).
// Partial sums over the time splits are combined with fp32 atomics into a zero-initialised G (summation order is
// therefore not fixed run to run, error ~1e-7 relative).
#include "common.h"
#include <algorithm>
#include <type_traits>

namespace {

// Workgroups a 3..5-tap launch is split into.  256 = one workgroup per CU in ONE round: measured on the B=16 training step
// (profiles/r02_u_wgrad_split_target_sweep.txt) 128 / 192 / 224 / 256 / 384 / 512 -> wgrad family 40.4 / 35.4 / 33.8 / 30.3 /
// 37.2 / 34.9 ms per iteration (every split pays a prologue, an exposed first load and an atomic pass over G).
int g_wgrad_target = 256;
// 1- and 2-tap launches of the tile kernel (svc_debug_set_wgrad_target(100000 + n)).  Swept on the attention / WN 1 x 1 layers of the
// B = 16, T = 768 batch (profiles/r05k_wgrad_k1_target_sweep.txt): 512 / 384 / 256 / 192 / 128 / 64 workgroups -> 36.7 / 37.1 /
// 33.7 / 37.8 / 54.7 / 97.8 us at 192 x 192, 47.4 / 53.1 / 44.7 / 56.3 / 73.2 / 133 us at 384 x 192 — and the same with bf16
// operands: these launches are bound by how the tiles are staged (0.5 TB/s of operand traffic), neither by the matrix pipe nor by
// the atomics of the combine pass.
int g_wgrad_k12_target = 256;
int g_wgrad_small_target = 256;   // same for the small-channel kernel (negative argument of svc_debug_set_wgrad_target): 128 / 256 / 512 -> 128.3 / 127.1 / 126.9 ms per iteration

constexpr int TT = 64;      // time steps per staged tile
constexpr int CA_T = 128;   // rows of A per workgroup (2 x 2 MFMA tiles)
constexpr int CB_T = 64;    // rows of Bm per workgroup (2 MFMA tiles)
constexpr int PA = TT + 1;
constexpr int MAXHALO_W = 44;            // (NK-1)*dil supported (k5 at dilation 11: DiscriminatorP period 11)
constexpr int BCOLS = (TT + MAXHALO_W + 63) / 64;   // column iterations per B row

struct WgP {
  const float* A;
  const float* Bm;
  float* G;
  float* dbias;
  long long a_bs, a_cs, b_bs, b_cs;
  int B, Ca, Cb, TA, TB, KS, dil, pad;
  int tiles_per_b, n_tiles, tiles_per_wg, PB, n_kgroups, splits;
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8v __attribute__((ext_vector_type(8)));
template <int MMA> struct WOp16;
template <> struct WOp16<SVC_MMA_BF16> {
  typedef bf16x8 frag;
  static constexpr int TERMS = 1;
  static __device__ __forceinline__ frag cvt(const f32x8v& t) { return __builtin_convertvector(t, bf16x8); }
  template <int T>
  static __device__ __forceinline__ f32x16 term(const frag& a, const frag& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct WOp16<SVC_MMA_F16> {
  typedef f16x8 frag;
  static constexpr int TERMS = 1;
  static __device__ __forceinline__ frag cvt(const f32x8v& t) { return __builtin_convertvector(t, f16x8); }
  template <int T>
  static __device__ __forceinline__ f32x16 term(const frag& a, const frag& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
// ---- LDS-DMA staging (DMA = true) -------------------------------------------------------------------------------------
// The register-staged tile hand-over above costs a launch more than its MFMAs when a tap group is short: per 64-step tile a
// thread issues 64 global loads, waits for them behind the MFMA loop, writes 64 LDS words and passes two barriers — 4.3 us per
// tile against 1.9 us of matrix work at one tap (profiles/r05k_wgrad_k1_target_sweep.txt: 97.8 / 54.7 / 33.7 us at 20 / 10 / 5
// tiles per workgroup), the same with 16-bit operands.  With DMA = true the tiles go from global memory / L2 straight into LDS
// (global_load_lds_dword: one instruction = one 64-step row piece, 256 B, written at an M0-given LDS address — so the ODD row
// pitch of the operand fetch survives, which the 16 B form's lane-linear image would not allow), double-buffered: the pieces of
// tile i + 1 are issued BETWEEN the MFMAs of tile i (one piece behind each of the first MFMAs of a macro-step, pinned with
// sched_barrier: the ~13 scalar / vector instructions of a piece run in the shadow of the 64-cycle matrix instruction), no
// staging registers, no LDS writes, ONE barrier per tile.  Zero padding (time steps outside [0, TB), the tail of the last
// tile) = the lane reads a zero word instead; rows past Ca / Cb are clamped (their accumulators are never written out).
__device__ float g_wgrad_zero[4];

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N) — the macro-step schedules below index registers and DMA
// pieces by it (a `#pragma unroll` loop whose body differs per iteration was peeled instead: the remaining iterations became a
// run-time loop over the accumulator array, 80 v_accvgpr copies per tile)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// tile buffers of the LDS-DMA form (taps per workgroup, 32-row MFMA tiles per wave along ca): host and device agree on this
constexpr __host__ __device__ int wgrad_nbuf(int nk, int mt) { return mt == 1 ? 3 : 2; }

__device__ __forceinline__ void glds4(const void* g, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(g), "s"(lds_byte));
}
__device__ __forceinline__ void wg_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// MMA (svc_wgrad_args.mma = SVC_MMA_BF16 / SVC_MMA_F16): the staged fp32 tiles are multiplied on v_mfma_f32_32x32x16_{bf16,f16} — an
// instruction reduces 16 time steps, lane half lk supplying steps 8*lk .. 8*lk + 7 of its channel row, rounded to the 16-bit format
// (round to nearest even) as they are read; fp32 accumulation, the bias gradient is summed from the fp32 tile as before.
template <int NK, int MMA = SVC_MMA_F32, bool DMA = false, int MT = 2>
__global__ __launch_bounds__(256) void conv1d_wgrad_kernel(WgP p) {
  // MT: 32-row MFMA tiles per wave along ca (workgroup block 64*MT x 64).  MT = 1 halves the block a workgroup adds into G —
  // the launch's atomic volume is (workgroups x block), and the combine pass is bound by L2's one-dword-per-clock-per-channel
  // atomic rate: ~7.5 us per tap plane of a 128 x 64 block at 256 workgroups (profiles/r05q_wgrad_dma_sweep.txt: a launch costs
  // ~8 us + 7.5 us x NK on top of its tiles) — at the price of twice as many, half as long tiles per workgroup.
  static_assert(MT == 2 || (MT == 1 && DMA), "wgrad: 64-row blocks exist in the LDS-DMA form only");
  constexpr int CAT = 64 * MT;
  extern __shared__ float lds[];
  float* As = lds;               // [CAT][PA]
  float* Bs = lds + CAT * PA;   // [CB_T][PB]
  const int PB = p.PB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR): row pointers below stay scalar
  const int ln = lane & 31, lk = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int ca0 = blockIdx.y * CAT, cb0 = blockIdx.z * CB_T;
  const int kg = blockIdx.x % p.n_kgroups;
  const int split = blockIdx.x / p.n_kgroups;
  const int k0 = kg * NK;
  const int nk = min(NK, p.KS - k0);               // taps of this group
  const int tile0 = split * p.tiles_per_wg;
  const int tile1 = min(tile0 + p.tiles_per_wg, p.n_tiles);
  const int halo = (nk - 1) * p.dil;
  const int XWB = TT + halo;
  const int boff = k0 * p.dil - p.pad;             // Bm index = t + boff + q*dil

  f32x16 acc[NK][MT];
#pragma unroll
  for (int q = 0; q < NK; ++q)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][i][r] = 0.f;

  // staging maps: A slot i -> row i*4 + wave, column lane;  B slot (j,cj) -> row j*4 + wave, column cj*64 + lane
  float areg[CAT / 4];
  float breg[(CB_T / 4) * BCOLS];

  auto load_tile = [&](int tile) {
    const int b = tile / p.tiles_per_b;
    const int t0 = (tile - b * p.tiles_per_b) * TT;
    const float* ab = p.A + (long long)b * p.a_bs;
    const float* bb = p.Bm + (long long)b * p.b_bs;
    // addresses = wave-uniform row pointer (SGPRs) + one 32-bit lane offset: no per-slot 64-bit address registers
    const int ta = min(t0 + lane, p.TA - 1);
#pragma unroll
    for (int i = 0; i < CAT / 4; ++i) {
      const int ca = min(ca0 + i * 4 + wave, p.Ca - 1);
      areg[i] = (ab + (long long)ca * p.a_cs)[ta];
    }
    int tb[BCOLS];
#pragma unroll
    for (int cj = 0; cj < BCOLS; ++cj) tb[cj] = min(max(t0 + boff + cj * 64 + lane, 0), p.TB - 1);
#pragma unroll
    for (int j = 0; j < CB_T / 4; ++j) {
      const int cb = min(cb0 + j * 4 + wave, p.Cb - 1);
      const float* brow = bb + (long long)cb * p.b_cs;
#pragma unroll
      for (int cj = 0; cj < BCOLS; ++cj) breg[j * BCOLS + cj] = brow[tb[cj]];
    }
  };
  auto store_tile = [&](int tile) {
    const int b = tile / p.tiles_per_b;
    const int t0 = (tile - b * p.tiles_per_b) * TT;
    const bool ta_ok = t0 + lane < p.TA;
#pragma unroll
    for (int i = 0; i < CAT / 4; ++i) {
      const int r = i * 4 + wave;
      As[r * PA + lane] = (ta_ok && ca0 + r < p.Ca) ? areg[i] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < CB_T / 4; ++j) {
      const int r = j * 4 + wave;
      const bool rok = cb0 + r < p.Cb;
#pragma unroll
      for (int cj = 0; cj < BCOLS; ++cj) {
        const int c = cj * 64 + lane;
        const int tb = t0 + boff + c;
        if (c < XWB) Bs[r * PB + c] = (rok && tb >= 0 && tb < p.TB) ? breg[j * BCOLS + cj] : 0.f;
      }
    }
  };

  // bias gradient (optional): row sums of the A tiles, taken by the workgroups of the first cb tile / tap group
  const bool do_bias = p.dbias != nullptr && blockIdx.z == 0 && kg == 0;
  constexpr int TPR = 256 / CAT, BSC = TT / TPR;   // threads per A row (2 / 4), columns each sums
  float bsum = 0.f;                       // thread tid: row tid / TPR, columns (tid % TPR) * BSC .. + BSC
  const float* brow_sum = As + (tid / TPR) * PA + (tid % TPR) * BSC;

  const float* ap = As + (wm * (32 * MT) + ln) * PA + lk;
  const float* bp = Bs + (wn * 32 + ln) * PB + lk;
  const int dil = p.dil;
  if constexpr (DMA) {
    // ---- LDS-DMA multi-buffered tiles (see glds4 above) ----
    // NBUF tile buffers, pieces issued NBUF - 1 tiles ahead, one behind each of the first MFMAs of every macro-step.  Three for the
    // 64-row blocks (50 KiB each): a piece then has a whole tile's MFMA loop to land whichever MFMA it was issued behind — with two
    // the last pieces of a tile had no time to land before the next tile's wait (one memory round trip exposed per tile: 5.8 us
    // per 2.8 us tile at 3 taps, profiles/r05r_*; three buffers: 768 x 192 x 3 taps 163 -> 149 us, 128 x 128 x 11 122 -> 116 us,
    // profiles/r05s_*).  Two for the 128-row blocks (66 KiB each at 2..5 taps; a third buffer for their one-tap form measured
    // slower, 29.7 -> 32.8 us: two more tiles of pieces per workgroup of five tiles), whose tiles are long enough to cover most of
    // it (10.4 us per 9.3 us tile at 5 taps); issuing their pieces behind the FIRST MFMAs of the tile instead was slower still
    // (384 x 192 x 5: 129 -> 138 us — a piece's ~12 instructions do not fit under every consecutive MFMA).
    constexpr int NBUF = wgrad_nbuf(NK, MT), AHEAD = NBUF - 1;
    constexpr int BC = NK == 1 ? 1 : 2;            // 64-column pieces per Bm row (PB = 65 / 129: set by the launcher)
    constexpr int NP = CAT / 4 + (CB_T / 4) * BC; // pieces per wave and tile: A rows i*4 + wave, then Bm rows j*4 + wave x BC
    const int BUF_F = CAT * PA + CB_T * PB;       // floats per buffer
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
    const char* zsrc = reinterpret_cast<const char*>(g_wgrad_zero);
    const long long a_csb = p.a_cs * 4, b_csb = p.b_cs * 4;
    // lane state of the tile whose pieces are being issued
    const char* n_ab = nullptr;
    const char* n_bb = nullptr;
    unsigned n_lds = 0, n_aoff = 0, n_boff[BC];
    bool n_aok = false, n_bok[BC];
    auto set_next = [&](int tile, int nbuf) {
      const bool live = tile < tile1;              // past the range: every lane reads the zero word (no branches in the loop)
      const int tl = live ? tile : tile0;
      const int b = tl / p.tiles_per_b;
      const int t0 = (tl - b * p.tiles_per_b) * TT;
      n_ab = reinterpret_cast<const char*>(p.A + (long long)b * p.a_bs);
      n_bb = reinterpret_cast<const char*>(p.Bm + (long long)b * p.b_bs);
      const int ta = t0 + lane;
      n_aok = live && ta < p.TA;
      n_aoff = (unsigned)ta * 4u;
#pragma unroll
      for (int cj = 0; cj < BC; ++cj) {
        const int tb = t0 + boff + cj * 64 + lane;
        n_bok[cj] = live && tb >= 0 && tb < p.TB;
        n_boff[cj] = (unsigned)tb * 4u;
      }
      n_lds = lds0 + (unsigned)(nbuf * BUF_F) * 4u;
    };
    auto dma_piece = [&](int pi) {                 // pi: compile-time after unrolling
      if (pi < CAT / 4) {
        const int r = pi * 4 + wave;
        const char* row = n_ab + (long long)min(ca0 + r, p.Ca - 1) * a_csb;
        const char* src = n_aok ? row + n_aoff : zsrc;
        glds4(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(n_lds + (unsigned)(r * PA) * 4u)));
      } else {
        const int j = (pi - CAT / 4) / BC, cj = (pi - CAT / 4) % BC;
        const int r = j * 4 + wave;
        const char* row = n_bb + (long long)min(cb0 + r, p.Cb - 1) * b_csb;
        const char* src = n_bok[cj] ? row + n_boff[cj] : zsrc;
        glds4(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(n_lds + (unsigned)(CAT * PA + r * PB + cj * 64) * 4u)));
      }
    };
    static_assert(NBUF == 2 || NP <= 63, "wgrad dma: s_waitcnt vmcnt holds 6 bits");
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) {      // (past the range: zero-source pieces — every tile slot issues exactly NP pieces)
      set_next(tile0 + a, a);
#pragma unroll
      for (int pi = 0; pi < NP; ++pi) dma_piece(pi);
    }
    int buf = 0;
    for (int tile = tile0; tile < tile1; ++tile, buf = buf + 1 == NBUF ? 0 : buf + 1) {
      // this wave's pieces of `tile` have landed (LDS-DMA loads complete in order: the NP pieces of tile + 1 may still fly)
      if constexpr (NBUF == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
      else wg_vmcnt0();
      __syncthreads();   // ... everyone's have; everyone is done reading the buffer of tile - 1
      set_next(tile + AHEAD, buf + AHEAD >= NBUF ? buf + AHEAD - NBUF : buf + AHEAD);
      const float* apb = ap + buf * BUF_F;
      const float* bpb = bp + buf * BUF_F;
      if (do_bias) {
        const float* bs = brow_sum + buf * BUF_F;
#pragma unroll
        for (int c = 0; c < BSC; ++c) bsum += bs[c];
      }
      if constexpr (MMA != SVC_MMA_F32) {
        // 4 macro-steps of 16 time steps; operands of macro-step g + 1 are read (and rounded) in front of the MFMAs of g
        typedef WOp16<MMA == SVC_MMA_F32 ? SVC_MMA_BF16 : MMA> OP;
        constexpr int G = TT / 16, PPG = NP / G, PPM = (PPG + MT * NK - 1) / (MT * NK);   // pieces per group / behind each MFMA
        static_assert(NP % G == 0, "wgrad dma: pieces per macro-step");
        const float* ap8 = apb + 7 * lk;
        const float* bp8 = bpb + 7 * lk;
        constexpr int SL = OP::TERMS > 1 ? 1 : 2;     // several instructions per operand pair: the LDS round trip is small beside them, one operand set
        typename OP::frag fa[SL][MT], fb[SL][NK];
        auto load16 = [&](int slot, int s) {
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            f32x8v t0;
#pragma unroll
            for (int j = 0; j < 8; ++j) t0[j] = ap8[i * 32 * PA + s + j];
            fa[slot][i] = OP::cvt(t0);
          }
#pragma unroll
          for (int q = 0; q < NK; ++q) {
            f32x8v tb;
#pragma unroll
            for (int j = 0; j < 8; ++j) tb[j] = bp8[s + q * dil + j];
            fb[slot][q] = OP::cvt(tb);
          }
        };
        if constexpr (SL == 2) load16(0, 0);
        static_for<0, G>([&](auto gc) {
          constexpr int g = decltype(gc)::value;
          if constexpr (SL == 1) load16(0, 16 * g);
          else if constexpr (g + 1 < G) load16((g + 1) & 1, 16 * (g + 1));
          __builtin_amdgcn_sched_barrier(0);
          static_for<0, OP::TERMS>([&](auto tc) {       // term-major: NK MT independent accumulators between two on the same one
            constexpr int T = decltype(tc)::value;
            static_for<0, NK * MT>([&](auto ec) {
              constexpr int e = decltype(ec)::value, q = e / MT, i = e % MT;
              acc[q][i] = OP::template term<T>(fa[g & (SL - 1)][i], fb[g & (SL - 1)][q], acc[q][i]);
              if constexpr (T == 0) {
                static_for<0, PPM>([&](auto xc) {
                  constexpr int d = e * PPM + decltype(xc)::value;
                  if constexpr (d < PPG && g * PPG + d < NP) dma_piece(g * PPG + d);
                });
              }
              __builtin_amdgcn_sched_barrier(0);
            });
          });
        });
      } else {
        // 16 macro-steps of 4 time steps (two MFMA steps s, s + 2 per operand pair: ds_read2_b32); operands of macro-step
        // m + 1 are read in front of the MFMAs of m
        constexpr int G = TT / 4, PPG = NP / G;
        static_assert(NP % G == 0 && PPG <= 2 * MT * NK, "wgrad dma: pieces per macro-step");
        float oa[2][MT][2], ob[2][NK][2];         // [slot][row tile | tap][step]
        auto load32 = [&](int slot, int s) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < MT; ++i) oa[slot][i][h] = apb[i * 32 * PA + s + 2 * h];
#pragma unroll
            for (int q = 0; q < NK; ++q) ob[slot][q][h] = bpb[s + q * dil + 2 * h];
          }
        };
        load32(0, 0);
        static_for<0, G>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          if constexpr (m + 1 < G) load32((m + 1) & 1, 4 * (m + 1));
          __builtin_amdgcn_sched_barrier(0);
          static_for<0, 2 * NK * MT>([&](auto ec) {
            constexpr int e = decltype(ec)::value, h = e / (NK * MT), q = (e / MT) % NK, i = e % MT;
            acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[m & 1][i][h], ob[m & 1][q][h], acc[q][i], 0, 0, 0);
            if constexpr (e < PPG && m * PPG + e < NP) dma_piece(m * PPG + e);
            __builtin_amdgcn_sched_barrier(0);
          });
        });
      }
    }
    wg_vmcnt0();   // the zero-source pieces issued during the last tile: landed before the epilogue reuses the buffers
  } else {
  if (tile0 < tile1) load_tile(tile0);
  for (int tile = tile0; tile < tile1; ++tile) {
    __syncthreads();   // previous tile fully consumed
    store_tile(tile);
    __syncthreads();
    if (tile + 1 < tile1) load_tile(tile + 1);   // in flight during the MFMA loop
    if (do_bias) {
#pragma unroll
      for (int c = 0; c < 32; ++c) bsum += brow_sum[c];
    }
    // Straight-line MFMA loop, NO per-tap `q < nk` guards: a partial tap group (the 3-tap tail of KS = 7 = 4+3 or
    // 11 = 4+4+3) runs its NK - nk surplus taps on whatever the Bs rows hold past the staged width — finite or not, those
    // accumulators are never written out (the epilogue keeps its guard) — at the price of 14 % / 9 % surplus MFMAs in that
    // group only.  With the guards hipcc turned every tap into its own basic block: a branch, a ds_read and an
    // `s_waitcnt lgkmcnt(0)` in front of EVERY MFMA pair (LDS latency fully exposed at one wave per SIMD), and the
    // branch-merged accumulators were copied between AGPRs and VGPRs inside the loop (16 v_accvgpr moves per MFMA pair in
    // the NK = 5 ISA).  Unguarded, the loop body is 2 + NK ds_reads feeding 2*NK back-to-back MFMAs, software-pipelined
    // over the 4x unroll.
    if constexpr (MMA != SVC_MMA_F32) {
      typedef WOp16<MMA == SVC_MMA_F32 ? SVC_MMA_BF16 : MMA> OP;
      // ap / bp already carry the fp32 instruction's lane offset lk; the 16-bit instruction's is 8 * lk
      const float* ap8 = ap + 7 * lk;
      const float* bp8 = bp + 7 * lk;
#pragma unroll 2
      for (int s = 0; s < TT; s += 16) {
        f32x8v t0, t1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          t0[j] = ap8[s + j];
          t1[j] = ap8[32 * PA + s + j];
        }
        const typename OP::frag a0 = OP::cvt(t0), a1 = OP::cvt(t1);
#pragma unroll
        for (int q = 0; q < NK; ++q) {
          f32x8v tb;
#pragma unroll
          for (int j = 0; j < 8; ++j) tb[j] = bp8[s + q * dil + j];
          const typename OP::frag bq = OP::cvt(tb);
          static_for<0, OP::TERMS>([&](auto tc) {
            constexpr int T = decltype(tc)::value;
            acc[q][0] = OP::template term<T>(a0, bq, acc[q][0]);
            acc[q][1] = OP::template term<T>(a1, bq, acc[q][1]);
          });
        }
      }
    } else {
#pragma unroll 4
    for (int s = 0; s < TT; s += 2) {
      const float a0 = ap[s], a1 = ap[32 * PA + s];
      float bv[NK];
#pragma unroll
      for (int q = 0; q < NK; ++q) bv[q] = bp[s + q * dil];
#pragma unroll
      for (int q = 0; q < NK; ++q) {
        acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[q], acc[q][0], 0, 0, 0);
        acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[q], acc[q][1], 0, 0, 0);
      }
    }
    }
  }
  }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 1);
    if (TPR == 4) bsum += __shfl_xor(bsum, 2);
    const int ca = ca0 + tid / TPR;
    if (tid % TPR == 0 && ca < p.Ca) atomicAdd(p.dbias + ca, bsum);
  }
  // combine: G[ca][cb][k0..k0+nk) += acc.  The MFMA C layout has one cb column per lane (addresses KS floats apart), so
  // each wave first transposes 16 ca rows at a time through LDS into G's own [cb][k] order: consecutive lanes then
  // add to consecutive addresses (coalesced atomics: one request per cache line instead of one per lane).
  __syncthreads();   // all waves are done with As / Bs
  // (index arithmetic with the tap count as a compile-time constant for full tap groups: with runtime divisors the two
  //  divisions per element made this pass ~100 VALU instructions per atomic — 12 k VALU per wave against 2.2 k MFMAs on the
  //  384 x 192 k = 5 layer, SQ_INSTS_VALU in profiles/r04u_pmc_wgrad.txt)
  auto combine = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    const int nkc = FULL ? NK : nk;
    const int RW = 32 * nkc;                // floats per staged row
    const int RP = RW + 1;
    float* Wt = lds + wave * 16 * (32 * NK + 1);
    const int cbw = cb0 + wn * 32;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int q = 0; q < NK; ++q) {
          if (q < nkc) {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
              const int r = hf * 8 + rr;
              const int rl = (rr & 3) + 8 * (rr >> 2) + 4 * lk;      // 0..15 inside this half
              Wt[rl * RP + ln * nkc + q] = acc[q][i][r];
            }
          }
        }
        __syncthreads();   // slab written (uniform trip counts: every wave reaches the barriers)
        // rows of this half: MFMA row (r&3) + 8*(r>>2) + 4*lk with r in [8hf, 8hf+8) -> 16*hf + rl
        const int ca_base = ca0 + wm * (32 * MT) + i * 32 + 16 * hf;
        for (int rl = 0; rl < 16; ++rl) {
          const int ca = ca_base + rl;
          if (ca >= p.Ca) break;              // wave-uniform
          float* grow = p.G + ((long long)ca * p.Cb + cbw) * p.KS + k0;
          for (int col = lane; col < RW; col += 64) {
            const int cbl = FULL ? col / NK : col / nkc, q = col - cbl * nkc;
            if (cbw + cbl < p.Cb) {
              float* g = grow + (long long)cbl * p.KS + q;
              const float v = Wt[rl * RP + col];
              if (p.splits > 1) atomicAdd(g, v);
              else *g += v;
            }
          }
        }
        __syncthreads();   // slab consumed before the next half overwrites it
      }
    }
  };
  if (nk == NK) combine(std::true_type{});
  else combine(std::false_type{});
}

// ---- small-channel variant (Ca, Cb <= 32: the 16/32-channel MRF stages of the decoder, first / last layers) ----------
// The 128 x 64 tile above would be >= 94 % padding there.  Here a workgroup owns the WHOLE [Ca][Cb][NKS taps] gradient and
// a range of 256-step time tiles; each of its 4 waves reduces a quarter of every tile with v_mfma_f32_16x16x4_f32
// (M = ca, N = cb, K = 4 time steps), one accumulator quad per (tap, 16x16 block).
constexpr int STT = 192;          // time steps per staged tile (48 per wave); As + Bs stay under 64 KiB of LDS
constexpr int SPA = STT + 1;

struct WgSP {
  const float* A;
  const float* Bm;
  float* G;
  float* dbias;
  long long a_bs, a_cs, b_bs, b_cs;
  int B, Ca, Cb, TA, TB, KS, dil, pad;
  int tiles_per_b, n_tiles, tiles_per_wg, PB, n_kgroups, splits;
};

template <int NKS, int MA, int NB>   // taps per workgroup, 16-row blocks of ca / cb
__global__ __launch_bounds__(256) void conv1d_wgrad_small_kernel(WgSP p) {
  extern __shared__ float lds[];
  float* As = lds;                       // [16*MA][SPA]
  float* Bs = lds + 16 * MA * SPA;       // [16*NB][PB]
  const int PB = p.PB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, lk = lane >> 4;
  const int kg = blockIdx.x % p.n_kgroups, split = blockIdx.x / p.n_kgroups;
  const int k0 = kg * NKS;
  const int nk = min(NKS, p.KS - k0);
  const int tile0 = split * p.tiles_per_wg, tile1 = min(tile0 + p.tiles_per_wg, p.n_tiles);
  const int XWB = STT + (nk - 1) * p.dil;
  const int boff = k0 * p.dil - p.pad;
  const int dil = p.dil;

  f32x4 acc[NKS][MA][NB];
#pragma unroll
  for (int q = 0; q < NKS; ++q)
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[q][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool do_bias = p.dbias != nullptr && kg == 0;

  for (int tile = tile0; tile < tile1; ++tile) {
    const int b = tile / p.tiles_per_b;
    const int t0 = (tile - b * p.tiles_per_b) * STT;
    const float* ab = p.A + (long long)b * p.a_bs;
    const float* bb = p.Bm + (long long)b * p.b_bs;
    __syncthreads();
    for (int r = wave; r < 16 * MA; r += 4) {            // rows wave-uniform, lanes along time (coalesced)
      const float* arow = ab + (long long)min(r, p.Ca - 1) * p.a_cs;
      for (int c = lane; c < STT; c += 64) {
        const int t = t0 + c;
        As[r * SPA + c] = (r < p.Ca && t < p.TA) ? arow[t] : 0.f;
      }
    }
    for (int r = wave; r < 16 * NB; r += 4) {
      const float* brow = bb + (long long)min(r, p.Cb - 1) * p.b_cs;
      for (int c = lane; c < XWB; c += 64) {
        const int t = t0 + boff + c;
        Bs[r * PB + c] = (r < p.Cb && t >= 0 && t < p.TB) ? brow[t] : 0.f;
      }
    }
    __syncthreads();
    if (do_bias && tid < 16 * MA * 8) {                   // 8 threads per row, STT/8 columns each
      const float* rp = As + (tid >> 3) * SPA + (tid & 7) * (STT / 8);
#pragma unroll
      for (int c = 0; c < STT / 8; ++c) bsum += rp[c];
    }
    const float* ap = As + ln * SPA + wave * (STT / 4) + lk;
    const float* bp = Bs + ln * PB + wave * (STT / 4) + lk;
#pragma unroll 2
    for (int s = 0; s < STT / 4; s += 4) {
      float av[MA];
#pragma unroll
      for (int i = 0; i < MA; ++i) av[i] = ap[i * 16 * SPA + s];
      // no `q < nk` guard (see conv1d_wgrad_kernel): the surplus taps of a partial group read past the staged width inside
      // the row pitch (PB covers NKS taps) into accumulators that are never written out
#pragma unroll
      for (int q = 0; q < NKS; ++q) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const float bv = bp[j * 16 * PB + s + q * dil];
#pragma unroll
          for (int i = 0; i < MA; ++i) acc[q][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv, acc[q][i][j], 0, 0, 0);
        }
      }
    }
  }
  if (do_bias && tid < 16 * MA * 8) {
    bsum += __shfl_xor(bsum, 1);
    bsum += __shfl_xor(bsum, 2);
    bsum += __shfl_xor(bsum, 4);
    const int ca = tid >> 3;
    if ((tid & 7) == 0 && ca < p.Ca) atomicAdd(p.dbias + ca, bsum);
  }
  // Every wave holds a partial sum of the SAME [taps][ca][cb] block (the waves split time): fold waves 2,3 into 0,1 and
  // then 1 into 0 through LDS so that the workgroup issues one set of atomics, not four (the gradient is a few KB that
  // every workgroup of the launch adds into: same-address atomics serialise in L2).
  constexpr int NQ = NKS * MA * NB;       // accumulator quads per wave
  float* red = lds;                        // [2][NQ*4][64]
  for (int round = 0; round < 2; ++round) {
    const int src_lo = round == 0 ? 2 : 1;            // waves >= src_lo write, waves < src_lo add
    __syncthreads();
    if (wave >= src_lo && wave < 2 * src_lo) {
      float* dst = red + (wave - src_lo) * NQ * 256 + lane;
#pragma unroll
      for (int q = 0; q < NKS; ++q)
#pragma unroll
        for (int i = 0; i < MA; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(((q * MA + i) * NB + j) * 4 + r) * 64] = acc[q][i][j][r];
    }
    __syncthreads();
    if (wave < src_lo) {
      const float* src = red + wave * NQ * 256 + lane;
#pragma unroll
      for (int q = 0; q < NKS; ++q)
#pragma unroll
        for (int i = 0; i < MA; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[q][i][j][r] += src[(((q * MA + i) * NB + j) * 4 + r) * 64];
    }
  }
  if (wave != 0) return;
  // MFMA 16x16 C layout: column n = ln, rows 4*lk + r
#pragma unroll
  for (int q = 0; q < NKS; ++q) {
    if (q >= nk) continue;
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ca = i * 16 + 4 * lk + r, cb = j * 16 + ln;
          if (ca < p.Ca && cb < p.Cb) {
            float* g = p.G + ((long long)ca * p.Cb + cb) * p.KS + k0 + q;
            if (p.splits > 1) atomicAdd(g, acc[q][i][j][r]);
            else *g += acc[q][i][j][r];
          }
        }
  }
}

template <int NKS, int MA, int NB>
void launch_small(const WgSP& p, dim3 grid, size_t lds, hipStream_t s) {
  hipLaunchKernelGGL((conv1d_wgrad_small_kernel<NKS, MA, NB>), grid, dim3(256), lds, s, p);
}

int g_wgrad_bf16_launches = 0;

// staging form of the tile kernel: 0 register-staged, 1 LDS-DMA double buffer, 2 (default) by rule
// (svc_debug_set_wgrad_target(200000 + v), SVC_WGRAD_DMA)
int g_wgrad_dma = -1;

// block rows of the LDS-DMA form: 0 = by rule (see svc_conv1d_wgrad_f32), 1 = 64, 2 = 128 (svc_debug_set_wgrad_target(300000 + v))
int g_wgrad_mt = 0;

template <int NK, int MMA, bool DMA, int MT>
void launch_one(const WgP& p, dim3 grid, size_t lds, hipStream_t s) {
  auto k = conv1d_wgrad_kernel<NK, MMA, DMA, MT>;
  if constexpr (DMA) {
    static bool done = false;
    if (!done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      done = true;
    }
  }
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, p);
}

template <int NK, bool DMA, int MT>
void launch_fmt(const WgP& p, dim3 grid, size_t lds, hipStream_t s, int mma) {
  if (mma == SVC_MMA_BF16) launch_one<NK, SVC_MMA_BF16, DMA, MT>(p, grid, lds, s);
  else if (mma == SVC_MMA_F16) launch_one<NK, SVC_MMA_F16, DMA, MT>(p, grid, lds, s);
  else launch_one<NK, SVC_MMA_F32, DMA, MT>(p, grid, lds, s);
}

template <int NK>
void launch(const WgP& p, dim3 grid, size_t lds, hipStream_t s, int mma, bool dma, int mt) {
  if (mma == SVC_MMA_BF16 || mma == SVC_MMA_F16) ++g_wgrad_bf16_launches;
  if (dma && mt == 1) launch_fmt<NK, true, 1>(p, grid, lds, s, mma);
  else if (dma) launch_fmt<NK, true, 2>(p, grid, lds, s, mma);
  else launch_fmt<NK, false, 2>(p, grid, lds, s, mma);
}

}  // namespace

extern "C" int svc_debug_wgrad_bf16_launches(void) { return g_wgrad_bf16_launches; }

extern "C" int svc_debug_set_wgrad_target(int workgroups) {
  if (workgroups == 0) return SVC_ERR_BAD_ARG;
  if (workgroups >= 300000) g_wgrad_mt = workgroups - 300000;            // block rows of the LDS-DMA form (0: rule)
  else if (workgroups >= 200000) g_wgrad_dma = workgroups - 200000;      // staging form of the tile kernel: 0 registers, 1 LDS-DMA, 2 rule
  else if (workgroups > 100000) g_wgrad_k12_target = workgroups - 100000;   // the 1- / 2-tap launches of the tile kernel
  else if (workgroups > 0) g_wgrad_target = workgroups;
  else g_wgrad_small_target = -workgroups;      // negative: the small-channel kernel's target
  return SVC_OK;
}

extern "C" int svc_conv1d_wgrad_f32(const svc_wgrad_args* ap, void* stream) {
  SVC_REQUIRE(ap != nullptr, "wgrad: null args");
  const svc_wgrad_args& a = *ap;
  SVC_REQUIRE(a.A && a.Bm && a.G, "wgrad: null tensor");
  SVC_REQUIRE(a.B > 0 && a.Ca > 0 && a.Cb > 0 && a.TA > 0 && a.TB > 0, "wgrad: empty shape");
  SVC_REQUIRE(a.KS >= 1 && a.KS <= 256 && a.dil >= 1, "wgrad: KS must be in [1,256] (got %d)", a.KS);
  hipStream_t s = (hipStream_t)stream;
  const double flop = 2.0 * a.B * (double)a.Ca * a.Cb * a.KS * a.TA;
  char pname[160];
  if (svc::prof_on() && svc::prof_shapes())
    snprintf(pname, sizeof(pname), "conv1d_wgrad[B%d,Ca%d,Cb%d,K%d,d%d,T%d]", a.B, a.Ca, a.Cb, a.KS, a.dil, a.TA);
  else
    snprintf(pname, sizeof(pname), "conv1d_wgrad");
  svc::ProfScope prof(s, pname, flop, 4.0 * a.B * ((double)a.Ca * a.TA + (double)a.Cb * a.TB));
  if (!a.accumulate) {
    if (hipMemsetAsync(a.G, 0, sizeof(float) * (size_t)a.Ca * a.Cb * a.KS, s) != hipSuccess) {
      svc::set_error("wgrad: memset failed");
      return SVC_ERR_HIP;
    }
  }
  if (a.dbias && !a.accumulate) {
    if (hipMemsetAsync(a.dbias, 0, sizeof(float) * (size_t)a.Ca, s) != hipSuccess) {
      svc::set_error("wgrad: memset failed");
      return SVC_ERR_HIP;
    }
  }
  if (a.Ca <= 32 && a.Cb <= 32) {
    WgSP q;
    q.A = a.A; q.Bm = a.Bm; q.G = a.G; q.dbias = a.dbias;
    q.a_bs = a.a_bs; q.a_cs = a.a_cs; q.b_bs = a.b_bs; q.b_cs = a.b_cs;
    q.B = a.B; q.Ca = a.Ca; q.Cb = a.Cb; q.TA = a.TA; q.TB = a.TB; q.KS = a.KS; q.dil = a.dil; q.pad = a.pad;
    const int MA = a.Ca > 16 ? 2 : 1, NB = a.Cb > 16 ? 2 : 1;
    // taps per workgroup: 4 accumulator registers per (tap, block); keep the LDS halo (nks-1)*dil <= 64
    int nks = (MA * NB == 1) ? 12 : (MA * NB == 2 ? 8 : 4);
    nks = std::min(nks, a.KS);
    while (nks > 1 && (nks - 1) * a.dil > 64) --nks;
    nks = nks > 8 ? 12 : (nks > 4 ? 8 : (nks > 2 ? 4 : nks));     // instantiated: 1, 2, 4, 8, 12
    if ((nks - 1) * a.dil > 64 && nks > 1) nks = nks == 12 ? 8 : nks / 2;
    q.n_kgroups = svc::cdiv(a.KS, nks);
    q.tiles_per_b = svc::cdiv(a.TA, STT);
    q.n_tiles = q.tiles_per_b * a.B;
    // few time splits: every split adds the whole (tiny) gradient with same-address atomics
    int splits = std::max(1, g_wgrad_small_target / q.n_kgroups);
    splits = std::min(splits, q.n_tiles);
    q.tiles_per_wg = svc::cdiv(q.n_tiles, splits);
    q.splits = svc::cdiv(q.n_tiles, q.tiles_per_wg);
    int pb = STT + (nks - 1) * a.dil;      // row pitch covers all NKS taps of the instantiation (unguarded MFMA loop)
    if ((pb & 1) == 0) ++pb;
    q.PB = pb;
    size_t lds = sizeof(float) * ((size_t)16 * MA * SPA + (size_t)16 * NB * pb);
    lds = std::max(lds, sizeof(float) * (size_t)2 * nks * MA * NB * 256);      // cross-wave reduction slabs
    dim3 grid(q.splits * q.n_kgroups);
#define SVC_WGS(NKS_)                                                     \
  if (MA == 1 && NB == 1) launch_small<NKS_, 1, 1>(q, grid, lds, s);      \
  else if (MA == 2 && NB == 1) launch_small<NKS_, 2, 1>(q, grid, lds, s); \
  else if (MA == 1 && NB == 2) launch_small<NKS_, 1, 2>(q, grid, lds, s); \
  else launch_small<NKS_, 2, 2>(q, grid, lds, s);
    switch (nks) {
      case 1: SVC_WGS(1) break;
      case 2: SVC_WGS(2) break;
      case 4: SVC_WGS(4) break;
      case 8: if (MA * NB <= 2) { SVC_WGS(8) } else { SVC_WGS(4) } break;
      default: if (MA * NB == 1) { launch_small<12, 1, 1>(q, grid, lds, s); } else { SVC_WGS(4) } break;
    }
#undef SVC_WGS
    return svc::check_launch("conv1d_wgrad_small");
  }
  // taps per workgroup: as many as the halo budget and 5 accumulator sets allow
  int nk = std::min(a.KS, 5);
  while (nk > 1 && (nk - 1) * a.dil > MAXHALO_W) --nk;
  if (a.KS > 5 && nk > 4) nk = 4;          // balanced groups for 7 (4+3) and 11 (4+4+3)
  WgP p;
  p.A = a.A; p.Bm = a.Bm; p.G = a.G; p.dbias = a.dbias;
  p.a_bs = a.a_bs; p.a_cs = a.a_cs; p.b_bs = a.b_bs; p.b_cs = a.b_cs;
  p.B = a.B; p.Ca = a.Ca; p.Cb = a.Cb; p.TA = a.TA; p.TB = a.TB; p.KS = a.KS; p.dil = a.dil; p.pad = a.pad;
  p.n_kgroups = svc::cdiv(a.KS, nk);
  p.tiles_per_b = svc::cdiv(a.TA, TT);
  p.n_tiles = p.tiles_per_b * a.B;
  if (g_wgrad_dma < 0) {
    const char* e = getenv("SVC_WGRAD_DMA");
    g_wgrad_dma = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2;
  }
  const int n_cb = svc::cdiv(a.Cb, CB_T);
  // enough time-splits to fill the chip (every split costs one prologue and one atomic pass over G), at least 2 tiles each so
  // the prefetch has something to hide: the target is tunable (svc_debug_set_wgrad_target)
  // (1- and 2-tap register-staged workgroups fit two per CU: a launch of >= 128 blocks — DiscriminatorP's 1024 x 1536 x 2 taps,
  //  192 blocks — runs as 384 co-resident workgroups of half the tiles: 658 -> 526 us, 16-bit operands 357 -> 214 us,
  //  profiles/r05r_*; the few-block 1 x 1 layers stay at one round of 256, profiles/r05k_*)
  int target = nk >= 3 ? g_wgrad_target : g_wgrad_k12_target;
  if ((nk == 2 || (nk == 1 && g_wgrad_dma == 0)) && g_wgrad_k12_target == 256 && svc::cdiv(a.Ca, CA_T) * n_cb * p.n_kgroups >= 128) target = 512;
  auto plan = [&](int rows, int& n_ca, int& splits, int& tiles_per_wg) {
    n_ca = svc::cdiv(a.Ca, rows);
    splits = std::max(1, target / (n_ca * n_cb * p.n_kgroups));
    splits = std::min(splits, std::max(1, p.n_tiles / 2));
    tiles_per_wg = svc::cdiv(p.n_tiles, splits);
    splits = svc::cdiv(p.n_tiles, tiles_per_wg);
  };
  // Staging form and block rows (kernel header).  Measured on the training step's shapes (profiles/r05s_wgrad_dma_triple_buffer_sweep.txt,
  // 256 workgroups, register-staged 128 x 64 -> LDS-DMA 64 x 64): 64-row blocks win wherever the launch is split over time many
  // times — 6 blocks x 42 splits (192 x 192): 5 taps 91.8 -> 73.7 us, 3 taps 66.7 -> 51.2, 1 tap 32.9 -> 24.7; 128 x 128 x 11 taps
  // 147 -> 116; 9 blocks x 28 splits (384 x 192 x 5) 130 -> 121 (16-bit operands 72 -> 58) — or 128 rows would be padding (Ca = 192:
  // 192 x 768 x 3 taps 197 -> 148); they change nothing at 18 blocks x 14 splits (768 x 192 x 3: 145 / 149) and lose where the
  // tile loop is all there is (128 blocks x 2 splits, 1024 x 1024 x 5 taps: 611 -> 705).  128-row blocks keep the register-staged
  // tiles, except one tap with 16-bit operands, where the LDS-DMA form is ahead (384 x 192: 43.7 -> 38.7 us; fp32 44.4 / 49.7,
  // profiles/r05v_wgrad_sweep_targets.txt).
  int n_ca, splits, tiles_per_wg;
  plan(CA_T, n_ca, splits, tiles_per_wg);
  const int ca_rem = a.Ca % CA_T;
  int mt = 2;
  if (g_wgrad_mt == 1 || (g_wgrad_mt == 0 && g_wgrad_dma != 0 && (splits >= 16 || (ca_rem > 0 && ca_rem <= CA_T / 2 && splits >= 4)))) mt = 1;
  const bool dma = g_wgrad_dma == 1 || mt == 1 || (g_wgrad_dma == 2 && nk == 1 && a.mma != SVC_MMA_F32);
  if (mt == 1) plan(CA_T / 2, n_ca, splits, tiles_per_wg);
  p.tiles_per_wg = tiles_per_wg;
  p.splits = splits;
  // the DMA pieces are 4-byte loads at (row base + 4 * t): any row / batch stride, any time offset
  int pb = TT + (nk - 1) * a.dil;
  if ((pb & 1) == 0) ++pb;
  if (dma) pb = nk == 1 ? TT + 1 : 2 * TT + 1;     // whole 64-column pieces per row (one / two), odd pitch
  p.PB = pb;
  size_t lds = sizeof(float) * ((size_t)(32 * mt * 2) * PA + (size_t)CB_T * pb) * (dma ? wgrad_nbuf(nk, mt) : 1);
  lds = std::max(lds, sizeof(float) * (size_t)4 * 16 * (32 * 5 + 1));     // epilogue transpose slabs
  dim3 grid(splits * p.n_kgroups, n_ca, n_cb);
  switch (nk) {
    case 1: launch<1>(p, grid, lds, s, a.mma, dma, mt); break;
    case 2: launch<2>(p, grid, lds, s, a.mma, dma, mt); break;
    case 3: launch<3>(p, grid, lds, s, a.mma, dma, mt); break;
    case 4: launch<4>(p, grid, lds, s, a.mma, dma, mt); break;
    default: launch<5>(p, grid, lds, s, a.mma, dma, mt); break;
  }
  return svc::check_launch("conv1d_wgrad");
}
