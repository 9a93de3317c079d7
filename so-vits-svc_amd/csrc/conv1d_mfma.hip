// conv1d_mfma.hip — fused dense Conv1d on the gfx950 fp32 matrix pipe.
//
// Implicit GEMM, no im2col in memory:  out[co,t] = sum_{ci,k} W[co,ci,k] * X[ci, t + k*dil - pad]
//   M = co (MFMA rows), N = t (MFMA cols), K = (ci,k).
// A workgroup owns a BM(co) x BN(t) tile.  Per chunk of BC input channels it stages
//   Xs[BC][XW]      (time contiguous: coalesced HBM reads along t — float4 when the rows are 16 B aligned —
//                    pre-activation applied once per element, not once per tap)
//   Ws[BC][KS][BM]  (co contiguous, from the packed [Cin][KS][CoutP] weight)
// in LDS; every wave then issues v_mfma_f32_32x32x2_f32 (or 16x16x4 for Cout<=16) where the A operand is one Ws
// dword per lane and the B operand one Xs dword per lane (lanes contiguous in m resp. t, so both ds_read_b32 are
// bank-conflict free and a dilated tap is just an address offset k*dil).
//
// Software pipeline: the global loads of chunk i+1 are issued into registers BEFORE the MFMA loop over chunk i
// and written to LDS after it, so HBM/L2 latency hides under the matrix work of the same workgroup (no reliance
// on a co-resident workgroup); inside the MFMA loop the operands of step i+1 are read from LDS while step i's
// MFMAs issue.  WK > 1 splits the K (input-channel) range of a chunk across waves that share one output tile and
// reduces through LDS at the end: that is what fills 256 CUs when T is a few hundred frames (encoder / flow).
// Polyphase ConvTranspose1d runs its `stride` sub-convolutions as separate workgroups.
//
// fp32 in, fp32 accumulate: bitwise an fmaf chain (MI355X_MICROARCH.md §Matrix cores), 157.3 TF peak.
//
// Replaces (reference path:line): vdecoder/hifigan/models.py:41-67 (ResBlock1 convs + leaky_relu + residual),
// :335,:358,:373-374 (conv_pre + cond), :340-342 (ups), modules/modules.py:110-138 (WN in_layers, gate, res_skip),
// modules/modules.py:288-307 (coupling pre/post), modules/attentions.py:198-205,337-345 (q,k,v,o,FFN),
// models.py:400,139 (pre, proj).
#include "common.h"
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>

namespace {

constexpr int WLD = 16;   // float4 weight loads in flight per thread per chunk
constexpr int XLD4 = 12;  // float4 (or 4x scalar) activation loads in flight per thread per chunk
constexpr int MAXHALO = 50;  // (KS-1)*dil supported by the staging maps (k=11, d=5)
// DB kernels: steps per trip of the operand ring for a compile-time tap count (0 = per-pair pipeline) and its look-ahead.
// Measured on one box (profiles/r04d..f_*): k = 3 gains 8..14 % with two channel pairs per trip (147 -> 135 us on the training
// graph's 768 -> 192 FFN conv, 43 -> 40.5 us on the decoder's 256-channel stage); k = 1 (8 steps) no change; k = 5 (10 steps)
// and k = 5 / 7 / 11 (one pair per trip) 2..5 % slower than the per-pair form with its sched groups; look-ahead 3 = look-ahead 2.
constexpr int RING_D = 2;
constexpr int ring_steps(int ksc) { return ksc == 3 ? 6 : 0; }

struct ConvP {
  svc_conv1d_args a;
  int XW;         // LDS row width of the X tile (floats, multiple of 4)
  int BC;         // input channels staged per chunk (multiple of KPI*WK)
  int n_t_tiles;  // number of BN tiles along t
  int n_m_tiles;
  int dump_off;   // float offset of the per-thread dump slots (staging writes of out-of-chunk slots)
  int row_phases; // u > 1: polyphase ConvTranspose1d with the u phases as output ROWS (row = co*u + phase): the epilogue writes
                  // row r, column q to y[r / u][q*u + r % u + y_t0] — u*BN consecutive samples per channel and tile
  int xvec;       // 1: X rows are 16 B aligned -> float4 staging (selects the XVEC kernel instantiation)
  int yvec;       // 1: y / res / y2 rows are 16 B aligned and unit-stride in time -> float4 epilogue
  int dbg;        // tuning experiments: 1 no staging, 2 no MFMA, 4 no epilogue (results are then garbage)
  int db_ni;      // DB kernels: LDS-DMA pieces (1 KiB each) per wave per chunk
  int fast_epi;   // DB kernels: 1 = plain epilogue with all residual / accumulate loads issued up front (conv_epilogue<BATCH>)
  int direct_epi; // DB kernels: 1 = epilogue straight from the accumulators (conv_epilogue_direct): no LDS transpose, no barrier
  const float* zero;  // DB kernels: 16 B of zeros in global memory (source of padding / out-of-tile pieces)
};

__device__ float4 g_zero_block[2];

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8v __attribute__((ext_vector_type(8)));

// 16-bit operand formats of the matrix pipe (svc_conv1d_args.mma): eight fp32 values -> one operand fragment (round to nearest
// even: v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32), and the 32x32x16 instruction of that format.  fp32 accumulation either way.
template <int MMA> struct Op16;
template <> struct Op16<SVC_MMA_BF16> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ frag cvt(const f32x8v& t) { return __builtin_convertvector(t, bf16x8); }
  static __device__ __forceinline__ f32x16 mfma(const frag& a, const frag& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Op16<SVC_MMA_F16> {
  typedef f16x8 frag;
  static __device__ __forceinline__ frag cvt(const f32x8v& t) { return __builtin_convertvector(t, f16x8); }
  static __device__ __forceinline__ f32x16 mfma(const frag& a, const frag& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
// One LDS-DMA piece: every lane's 16 B at `g` land at LDS byte address lds_byte + lane*16 (wave-uniform base in M0).
// Not tracked by hipcc's s_waitcnt bookkeeping: the kernel counts these itself (svc_vmcnt0 before the barrier).
__device__ __forceinline__ void glds16(const void* g, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(g), "s"(lds_byte)
               : "memory");
}
__device__ __forceinline__ void svc_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ float acc_probe(const f32x16& a, const f32x4& b) { return a[0] + b[0]; }

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case SVC_ACT_RELU: return v > 0.f ? v : 0.f;
    case SVC_ACT_TANH: return tanhf(v);
    case SVC_ACT_LRELU: return svc_lrelu(v, slope);
    case SVC_ACT_GELU: return svc_gelu(v);
    default: return v;
  }
}

// ---- shared epilogue: the WK partial tiles of a workgroup sit in LDS as [WK][BM][BN + 4]; every thread owns 4 consecutive
// time steps of one output row: 16 B residual loads / stores, full cache lines per row.  Split-K partial tiles are summed
// here (fixed order wk = 0..WK-1: deterministic).
template <int BM, int BN, int NTHR, int WK, int EPI, bool BATCH = false>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, const float* smem, int tid, int b, int ph, int t0, int co0) {
  const svc_conv1d_args& a = p.a;
  constexpr int CP = BN + 4;
  const float* maskb = a.mask ? a.mask + (long long)b * a.mask_bs : nullptr;
  const float* condb = a.cond ? a.cond + (long long)b * a.cond_bs : nullptr;
  const float* resb = a.res ? a.res + (long long)b * a.res_bs : nullptr;
  float* yb = a.y + (long long)b * a.y_bs;
  const float* biasp = a.bias;
  const long long cond_cs = a.cond_cs, cond_ts = a.cond_ts;
  constexpr int BN4 = BN / 4;
  constexpr int OUT_ROWS = EPI == SVC_EPI_GATE ? BM / 2 : BM;
  const bool yvec = p.yvec != 0;
  const int H = a.Cout >> 1;

  if constexpr (EPI == SVC_EPI_PLAIN) {
    if (p.row_phases > 1) {
      // phases-as-rows ConvTranspose1d: consecutive threads write consecutive output samples (q*u + phase) of one channel:
      // full cache lines, where the phase-per-workgroup form writes every u-th float from u different workgroups
      const int u = p.row_phases, lu = 31 - __builtin_clz(u);      // u is a power of two (svc::convt_rows_layout)
      const int cout = a.Cout >> lu, cobase = co0 >> lu;
      const bool has_res = a.res_mode == 1;
      for (int idx = tid; idx < BM * BN; idx += NTHR) {
        const int phs = idx & (u - 1), e = idx >> lu;               // e = (channel, column) pair, phase fastest
        const int co_l = e / BN, tl = e - co_l * BN;                // BN is a compile-time constant
        const int co = cobase + co_l, tq = t0 + tl;
        const int tt = (tq << lu) + phs + a.y_t0;
        if (co >= cout || tq >= a.Tout || tt < 0 || tt >= a.y_len) continue;
        const int prow = (co_l << lu) + phs;
        float v = smem[prow * CP + tl];
#pragma unroll
        for (int w = 1; w < WK; ++w) v += smem[(w * BM + prow) * CP + tl];
        v += biasp ? biasp[co] : 0.f;
        v = apply_act(v, a.post_act, a.post_slope);
        if (has_res) v += resb[(long long)co * a.res_cs + tt];
        if (a.out_div != 1.f) v /= a.out_div;
        yb[(long long)co * a.y_cs + tt] = v;
      }
      return;
    }
  }
  auto tile4 = [&](int prow, int c4) -> float4 {
    float4 q = *reinterpret_cast<const float4*>(smem + prow * CP + c4 * 4);
#pragma unroll
    for (int w = 1; w < WK; ++w) {
      const float4 q2 = *reinterpret_cast<const float4*>(smem + (w * BM + prow) * CP + c4 * 4);
      q.x += q2.x; q.y += q2.y; q.z += q2.z; q.w += q2.w;
    }
    return q;
  };
  auto tout = [&](int tq, int e) { return (tq + e) * a.y_ts + a.y_t0 + ph; };
  auto tvalid = [&](int tq, int e) {
    const int t = tout(tq, e);
    return tq + e < a.Tout && t >= 0 && t < a.y_len;
  };
  auto tclamp = [&](int tq, int e) { return min(max(tout(tq, e), 0), a.y_len - 1); };
  // 4 values of an output-shaped row at time steps tq..tq+3 (vector when aligned, else element-wise with bounds)
  auto load4 = [&](const float* rowp, int tq) -> float4 {
    if (yvec && tq + 3 < a.Tout) return *reinterpret_cast<const float4*>(rowp + tq);
    float4 r;
    r.x = tvalid(tq, 0) ? rowp[tclamp(tq, 0)] : 0.f;
    r.y = tvalid(tq, 1) ? rowp[tclamp(tq, 1)] : 0.f;
    r.z = tvalid(tq, 2) ? rowp[tclamp(tq, 2)] : 0.f;
    r.w = tvalid(tq, 3) ? rowp[tclamp(tq, 3)] : 0.f;
    return r;
  };
  auto store4 = [&](float* rowp, int tq, float4 v) {
    if (yvec && tq + 3 < a.Tout) {
      *reinterpret_cast<float4*>(rowp + tq) = v;
    } else {
      if (tvalid(tq, 0)) rowp[tout(tq, 0)] = v.x;
      if (tvalid(tq, 1)) rowp[tout(tq, 1)] = v.y;
      if (tvalid(tq, 2)) rowp[tout(tq, 2)] = v.z;
      if (tvalid(tq, 3)) rowp[tout(tq, 3)] = v.w;
    }
  };
  auto side4 = [&](const float* base, long long ts, int tq) -> float4 {   // base[t*ts] for the 4 (clamped) steps
    return make_float4(base[(long long)tclamp(tq, 0) * ts], base[(long long)tclamp(tq, 1) * ts],
                       base[(long long)tclamp(tq, 2) * ts], base[(long long)tclamp(tq, 3) * ts]);
  };
#define SVC_F4_MAP(dst, expr)            \
  {                                      \
    { const int e = 0; (dst).x = (expr); } \
    { const int e = 1; (dst).y = (expr); } \
    { const int e = 2; (dst).z = (expr); } \
    { const int e = 3; (dst).w = (expr); } \
  }
#define F4C(q) (e == 0 ? (q).x : e == 1 ? (q).y : e == 2 ? (q).z : (q).w)

  if constexpr (BATCH && EPI == SVC_EPI_PLAIN && (BM * BN4) % NTHR == 0) {
    // Big tiles of the MRF ResBlock convs (p.fast_epi: aligned rows, residual add or none, no mask / time-varying cond):
    // ALL residual (and accumulate) loads of the thread are issued first, then the trips compute and store.  Measured on the
    // 128-channel k=11 stage (128 x 224 tile, 28 trips per thread): the epilogue is 27 of 192 us when every trip waits for
    // its own 16-byte load (one L2 / HBM round trip per trip); nothing else can hide that latency in a kernel that runs
    // one tile per CU with the whole chip in the same phase.  Same arithmetic, same order: bit-identical results.
    if (p.fast_epi) {
      constexpr int NIT = BM * BN4 / NTHR;
      float4 rr[NIT], yy[NIT];
      const bool has_res = a.res_mode != 0, has_acc = a.beta != 0.f;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * NTHR;
        const int row = idx / BN4, c4 = idx - row * BN4;
        const int co = min(co0 + row, a.Cout - 1);
        const int tq = min(t0 + c4 * 4, a.Tout - 4);
        rr[it] = has_res ? *reinterpret_cast<const float4*>(resb + (long long)co * a.res_cs + tq) : make_float4(0.f, 0.f, 0.f, 0.f);
        yy[it] = has_acc ? *reinterpret_cast<const float4*>(yb + (long long)co * a.y_cs + tq) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * NTHR;
        const int row = idx / BN4, c4 = idx - row * BN4;
        const int co = co0 + row, tq = t0 + c4 * 4;
        if (co >= a.Cout || tq >= a.Tout) continue;
        float4 v = tile4(row, c4);
        const float bb = biasp ? biasp[co] : 0.f;
        const float cc = condb ? condb[co * cond_cs] : 0.f;
        SVC_F4_MAP(v, F4C(v) + bb + cc);
        SVC_F4_MAP(v, apply_act(F4C(v), a.post_act, a.post_slope) * 1.f);
        if (has_res) { SVC_F4_MAP(v, F4C(v) + F4C(rr[it])); }
        if (has_acc) { SVC_F4_MAP(v, F4C(v) + a.beta * F4C(yy[it])); }
        if (a.out_div != 1.f) { SVC_F4_MAP(v, F4C(v) / a.out_div); }
        *reinterpret_cast<float4*>(yb + (long long)co * a.y_cs + tq) = v;
      }
      return;
    }
  }
  for (int idx = tid; idx < OUT_ROWS * BN4; idx += NTHR) {
    const int row = idx / BN4, c4 = idx - row * BN4;
    const int tq = t0 + c4 * 4;
    if (tq >= a.Tout) continue;
    const float4 mk = maskb ? side4(maskb, 1, tq) : make_float4(1.f, 1.f, 1.f, 1.f);
    if constexpr (EPI == SVC_EPI_GATE) {
      static_assert(EPI != SVC_EPI_GATE || (BM % 64) == 0, "gate epilogue needs tile pairs");
      const int prow = (row >> 5) * 64 + (row & 31);  // packed tanh row inside the tile; +32 = its sigmoid row
      const int c = (co0 >> 1) + row;
      if (c >= H) continue;
      const float4 vt = tile4(prow, c4), vs = tile4(prow + 32, c4);
      const float bt = biasp ? biasp[c] : 0.f, bs = biasp ? biasp[H + c] : 0.f;
      float4 ct = make_float4(0.f, 0.f, 0.f, 0.f), cs = ct;
      if (condb) {
        ct = side4(condb + c * cond_cs, cond_ts, tq);
        cs = side4(condb + (H + c) * cond_cs, cond_ts, tq);
      }
      float4 o;
      SVC_F4_MAP(o, tanhf(F4C(vt) + bt + F4C(ct)) * svc_sigmoid(F4C(vs) + bs + F4C(cs)));
      store4(yb + (long long)c * a.y_cs, tq, o);
    } else {
      const int co = co0 + row;
      if (co >= a.Cout) continue;
      float4 v = tile4(row, c4);
      const float bb = biasp ? biasp[co] : 0.f;
      float4 cc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (condb) cc = side4(condb + co * cond_cs, cond_ts, tq);
      SVC_F4_MAP(v, F4C(v) + bb + F4C(cc));
      if constexpr (EPI == SVC_EPI_RES_SKIP) {
        if (co < a.skip_from) {
          const float4 r = load4(resb + (long long)co * a.res_cs, tq);
          SVC_F4_MAP(v, (F4C(r) + F4C(v)) * F4C(mk));
          store4(yb + (long long)co * a.y_cs, tq, v);
        } else {
          float* y2p = a.y2 + (long long)b * a.y2_bs + (long long)(co - a.skip_from) * a.y2_cs;
          if (a.beta != 0.f) {
            const float4 r = load4(y2p, tq);
            SVC_F4_MAP(v, F4C(v) + a.beta * F4C(r));
          }
          if (a.res_mode == 1) {  // last WN layer: `output * x_mask` (modules/modules.py:138)
            SVC_F4_MAP(v, F4C(v) * F4C(mk));
          }
          store4(y2p, tq, v);
        }
      } else {
        float* yp = yb + (long long)co * a.y_cs;
        SVC_F4_MAP(v, apply_act(F4C(v), a.post_act, a.post_slope) * F4C(mk));
        if (a.res_mode != 0) {
          const float4 r = load4(resb + (long long)co * a.res_cs, tq);
          if (a.res_mode == 1) { SVC_F4_MAP(v, F4C(v) + F4C(r)); }
          else if (a.res_mode == 2) { SVC_F4_MAP(v, (F4C(r) - F4C(v)) * F4C(mk)); }
          else { SVC_F4_MAP(v, F4C(v) + F4C(r) * F4C(mk)); }
        }
        if (a.beta != 0.f) {
          const float4 yo = load4(yp, tq);
          SVC_F4_MAP(v, F4C(v) + a.beta * F4C(yo));
        }
        if (a.out_div != 1.f) { SVC_F4_MAP(v, F4C(v) / a.out_div); }
        store4(yp, tq, v);
      }
    }
  }
#undef SVC_F4_MAP
#undef F4C
}

// ---- direct epilogue (DB kernels, 32x32 tiles, plain epilogue without mask / time-varying cond): a lane's accumulator register
// is 32 consecutive time steps of one output row (MFMA C layout), so bias / activation / residual / accumulate / store run
// straight from the accumulators with 128-byte row segments per half wave: no LDS transpose, no barriers, and the residual loads
// of ALL of the wave's tiles are in flight together.  Round 2 measured the LDS epilogue at 27 us of a 192 us launch (14 %), with
// every workgroup of a launch in it at the same time; round 3's conv1d_strip.hip showed the direct form costs ~2 us.  Same
// expression and order as conv_epilogue: bit-identical results.  Addresses = (wave-uniform row base in SGPRs) + (per-lane 32-bit
// byte offset), written as asm so that hipcc neither materialises one 64-bit address per element nor branches per load.
template <int MT, int NT>
__device__ __forceinline__ void conv_epilogue_direct(const ConvP& p, f32x16 (&acc)[MT][NT], int b_, int t0, int rowu0_, int col0, int lane) {
  const svc_conv1d_args& a = p.a;
  const int b = __builtin_amdgcn_readfirstlane(b_), rowu0 = __builtin_amdgcn_readfirstlane(rowu0_);   // wave-uniform by construction
  const int ln = lane & 31, lk = lane >> 5;
  float* yb = a.y + (long long)b * a.y_bs;
  const float* resb = a.res ? a.res + (long long)b * a.res_bs : a.x;
  const float* condb = a.cond ? a.cond + (long long)b * a.cond_bs : nullptr;
  const bool has_res = a.res_mode != 0;
  const float oslope = a.post_act == SVC_ACT_LRELU ? a.post_slope : 1.f;   // launcher: none, or leaky-ReLU with 0 <= slope <= 1
  auto rowc = [](int r) { return (r & 3) + 8 * (r >> 2); };
  unsigned roff[NT], yoff[NT];
  bool colok[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int t = t0 + col0 + j * 32 + ln;
    const unsigned tc = (unsigned)min(t, a.Tout - 1);
    colok[j] = t < a.Tout;
    roff[j] = 4u * ((unsigned)(4 * lk) * (unsigned)a.res_cs + tc);
    yoff[j] = 4u * ((unsigned)(4 * lk) * (unsigned)a.y_cs + tc);
  }
  float rr[MT][NT][16];
  if (has_res) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int rowl = rowu0 + i * 32 < a.Cout ? rowu0 + i * 32 : 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* rp = resb + (long long)(rowl + rowc(r)) * a.res_cs;   // wave-uniform
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("global_load_dword %0, %1, %2" : "=v"(rr[i][j][r]) : "v"(roff[j]), "s"(rp));
      }
    }
  }
  float bc_[MT][16];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int rowl = rowu0 + i * 32 < a.Cout ? rowu0 + i * 32 : 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = rowl + rowc(r) + 4 * lk;
      bc_[i][r] = (a.bias ? a.bias[co] : 0.f) + (condb ? condb[co * a.cond_cs] : 0.f);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (has_res) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(rr[i][j][r]));   // uses of rr stay behind the wait
  }
  auto finish = [&](auto accdiv_tag) {
    constexpr bool ACCDIV = decltype(accdiv_tag)::value;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const bool rows_ok = rowu0 + i * 32 < a.Cout;
      const int rowl = rows_ok ? rowu0 + i * 32 : 0;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (rows_ok && colok[j]) {
          float yo[16];
          if constexpr (ACCDIV) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float* yp = yb + (long long)(rowl + rowc(r)) * a.y_cs;
              asm volatile("global_load_dword %0, %1, %2" : "=v"(yo[r]) : "v"(yoff[j]), "s"(yp));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(yo[r]));
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* yp = yb + (long long)(rowl + rowc(r)) * a.y_cs;
            float v = acc[i][j][r] + bc_[i][r];
            v = __builtin_amdgcn_fmed3f(v, v * oslope, __builtin_inff());   // == svc_lrelu for 0 <= slope <= 1; slope 1: identity
            if (has_res) v = v + rr[i][j][r];
            if constexpr (ACCDIV) {
              v = v + a.beta * yo[r];
              v = v / a.out_div;
            }
            asm volatile("global_store_dword %0, %1, %2" : : "v"(yoff[j]), "v"(v), "s"(yp) : "memory");
          }
        }
      }
    }
  };
  if (a.beta != 0.f || a.out_div != 1.f) finish(std::true_type{});
  else finish(std::false_type{});
}

// MT x NT MFMA tiles per wave; WM x WN x WK waves per workgroup (WK waves split the reduction).
template <int MT, int NT, int WM, int WN, int WK, bool M16, int EPI, int KSC, bool XVEC, bool DB = false, int MMA = SVC_MMA_F32>
// (scalar-staging instantiations — unaligned rows, e.g. DiscriminatorP's period layout — keep 48 X + 64 W staging registers
// in flight next to the accumulators: at two workgroups per CU they spilled 92..296 B/lane of scratch into the chunk loop
// and ran 3.7x slower than their aligned twins; they get the whole register file of a SIMD instead)
__device__ __forceinline__ void conv1d_mfma_body(const ConvP& p, int bid) {
  constexpr int TS = M16 ? 16 : 32;   // MFMA tile edge
  constexpr int KPI = M16 ? 4 : 2;    // K indices consumed per MFMA
  constexpr int NACC = M16 ? 4 : 16;  // accumulator regs per tile
  constexpr int BM = WM * MT * TS;
  constexpr int BN = WN * NT * TS;
  constexpr int NW = WM * WN * WK;
  constexpr int NTHR = NW * 64;
  constexpr int BM4 = BM / 4;
  const svc_conv1d_args& a = p.a;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int KS = a.KS, BC = p.BC, XW = p.XW;
  float* Ws = smem;                 // [BC][KS][BM]
  float* Xs = smem + BC * KS * BM;  // [BC][XW]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wk = wave / (WM * WN);
  const int wmn = wave % (WM * WN);
  const int wm = wmn / WN, wn = wmn % WN;
  const int ln = lane & (TS - 1);  // position inside tile row/col
  const int lk = lane / TS;        // which K index of the instruction this lane feeds

  // block -> (phase, t tile, m tile, batch); phase fastest so the polyphase siblings share the X tile in L2
  // (round 3: an XCD-aware map — all row tiles of a column tile on the XCD whose L2 holds its X rows — changed nothing on the
  //  training shapes (3..12 row tiles) or the decoder's, profiles/r04a_trainconv_swz*.txt: X is not what these launches wait for)
  const int ph = bid % a.n_phase;
  bid /= a.n_phase;
  const int tt = bid % p.n_t_tiles;
  bid /= p.n_t_tiles;
  const int mtile = bid % p.n_m_tiles;
  const int b = bid / p.n_m_tiles;
  const int t0 = tt * BN;
  const int co0 = mtile * BM;

  const float* xb = a.x + (long long)b * a.x_bs;
  const float* pm = a.premask ? a.premask + (long long)b * a.premask_bs : nullptr;
  const int tin0 = t0 - a.pad_left;
  const int sh = XVEC ? (((tin0 % 4) + 4) % 4) : 0;  // tile start rounded down to a 16 B boundary
  const int tin_base = tin0 - sh;
  const int w_rows_total = a.Cin * KS;
  const float* wph = a.w + (long long)ph * a.w_phase_stride;

  f32x16 acc32[MT][NT];
  f32x4 acc16[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if constexpr (M16) {
        acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;
      }
    }

  // ---- register staging of one chunk -------------------------------------------------------------------
  // Thread -> element maps are affine in the slot index (per-thread base + wave-uniform slot offset) and every
  // slot is branch-free: the address is clamped into the tensor and the value selected to zero when the slot is
  // outside the chunk / tensor (no exec-mask branches in the staging code):
  //   W slot i : row r0 + i*RSTEP of the chunk's [BC*KS][BM] block, float4 column c4      (r0 = tid / BM4)
  //   X slot (ri,cj): channel row ri*NW + wave, column (cj*64 + lane) [float4 or float]
  constexpr int RSTEP = NTHR / BM4;
  constexpr int XCIV = (BN + MAXHALO + 3 + 255) / 256;  // float4 column iterations per row (vector path)
  constexpr int XCIS = (BN + MAXHALO + 63) / 64;        // column iterations per row (scalar path)
  constexpr int XSLV = (XLD4 / XCIV) * XCIV;            // slots used, vector path
  constexpr int XSLS = ((4 * XLD4) / XCIS) * XCIS;      // slots used, scalar path
  float4 wreg[WLD];
  float xr[4 * XLD4];
  const int w_rows_chunk = BC * KS;
  const int wr0 = tid / BM4, wc4 = tid - wr0 * BM4;
  const int wcol = min(co0 + wc4 * 4, a.CoutP - 4);
  const bool wcol_ok = co0 + wc4 * 4 < a.CoutP;
  float* wlds = Ws + wr0 * BM + wc4 * 4;
  const int XW4 = XW >> 2;
  float* xlds = Xs + wave * XW + (XVEC ? lane * 4 : lane);
  // slots that fall outside the chunk write to a per-thread dump slot behind the tiles: the staging code is
  // straight-line (no exec-mask branches), which matters because every wave runs it between two barriers
  float* dump = smem + p.dump_off + tid * 4;

  // premask depends only on the column: fetched once (scalar path only; the vector path is taken only without one)
  float pmv[XCIS];
#pragma unroll
  for (int cj = 0; cj < XCIS; ++cj) {
    const int tc = min(max(tin_base + cj * 64 + lane, 0), a.Tin - 1);
    pmv[cj] = pm ? pm[tc] : 1.f;
  }

  // loads are raw (addresses clamped into the tensor); out-of-chunk / out-of-tensor slots are zeroed when the
  // registers are written to LDS, so nothing consumes a load before the MFMA loop it overlaps with
  auto load_chunk = [&](int c0) {
    const int row0 = c0 * KS;
#pragma unroll
    for (int i = 0; i < WLD; ++i) {
      const int gr = min(row0 + wr0 + i * RSTEP, w_rows_total - 1);
      wreg[i] = *reinterpret_cast<const float4*>(wph + (long long)gr * a.CoutP + wcol);
    }
    if constexpr (XVEC) {
#pragma unroll
      for (int s = 0; s < XSLV; ++s) {
        const int ri = s / XCIV, cj = s % XCIV;
        const int ci = min(c0 + ri * NW + wave, a.Cin - 1);
        const int tc = min(max(tin_base + (cj * 64 + lane) * 4, 0), a.Tin - 4);
        const float4 v = *reinterpret_cast<const float4*>(xb + (long long)ci * a.x_cs + tc);
        xr[4 * s + 0] = v.x; xr[4 * s + 1] = v.y; xr[4 * s + 2] = v.z; xr[4 * s + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int s = 0; s < XSLS; ++s) {
        const int ri = s / XCIS, cj = s % XCIS;
        const int ci = min(c0 + ri * NW + wave, a.Cin - 1);
        const int tc = min(max(tin_base + cj * 64 + lane, 0), a.Tin - 1);
        xr[s] = xb[(long long)ci * a.x_cs + tc];
      }
    }
  };

  auto store_chunk = [&](int c0) {
    const int row0 = c0 * KS;
#pragma unroll
    for (int i = 0; i < WLD; ++i) {
      const int r = wr0 + i * RSTEP;
      const bool ok = row0 + r < w_rows_total && wcol_ok;
      const float4 v = wreg[i];
      float* dst = r < w_rows_chunk ? wlds + i * RSTEP * BM : dump;
      *reinterpret_cast<float4*>(dst) = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
    const float ps = a.pre_slope;
    if constexpr (XVEC) {
#pragma unroll
      for (int s = 0; s < XSLV; ++s) {
        const int ri = s / XCIV, cj = s % XCIV;
        const int r = ri * NW + wave, c4 = cj * 64 + lane;
        const int tin = tin_base + c4 * 4;
        const bool ok = c0 + r < a.Cin && tin >= 0 && tin < a.Tin;
        float* dst = (r < BC && c4 < XW4) ? xlds + ri * NW * XW + cj * 256 : dump;
        *reinterpret_cast<float4*>(dst) =
            make_float4(ok ? svc_lrelu(xr[4 * s], ps) : 0.f, ok ? svc_lrelu(xr[4 * s + 1], ps) : 0.f,
                        ok ? svc_lrelu(xr[4 * s + 2], ps) : 0.f, ok ? svc_lrelu(xr[4 * s + 3], ps) : 0.f);
      }
    } else {
#pragma unroll
      for (int s = 0; s < XSLS; ++s) {
        const int ri = s / XCIS, cj = s % XCIS;
        const int r = ri * NW + wave, c = cj * 64 + lane;
        const int tin = tin_base + c;
        const bool ok = c0 + r < a.Cin && tin >= 0 && tin < a.Tin;
        float* dst = (r < BC && c < XW) ? xlds + ri * NW * XW + cj * 64 : dump;
        *dst = ok ? svc_lrelu(xr[s] * pmv[cj], ps) : 0.f;
      }
    }
  };

  // ---- main loop over input-channel chunks ---------------------------------------------------------------
  const float* wbase = Ws + wm * (MT * TS) + ln;
  const float* xbase = Xs + wn * (NT * TS) + ln + sh;
  const int dbg = p.dbg;
  // (tried: offsetting the chunk phase of co-resident workgroups by half a chunk — no gain, −3 %: the two workgroups
  // of a CU are not phase-locked)
  if constexpr (DB) {
    // ---- LDS-DMA, double-buffered chunks ---------------------------------------------------------------------
    // A chunk's W block [BC*KS][BM] and X block [BC][XW] are one linear range of float4 slots; wave w's piece i covers
    // slots (i*NW + w)*64 .. +63 (1 KiB, lane-linear as the DMA writes it).  Each lane keeps the global source of its
    // slot and advances it by the chunk stride; padding (t outside the sequence, weight columns >= CoutP) and slots past
    // the tile read a 16 B zero block with stride 0, so the issue code has no branches and the LDS image needs no
    // fix-up.  While the MFMA loop runs on buffer i&1 the pieces of chunk i+1 are in flight into the other buffer:
    // one vmcnt(0) + one barrier per chunk, neither on the critical path (the loads had a whole MFMA loop to land).
    // The leaky-ReLU pre-activation moves from the staging pass to the operand read: max(x, slope*x), 0 <= slope <= 1.
    constexpr int NI = 16;
    static_assert(!M16 && WK == 1 && XVEC, "DB kernels: 32x32 tiles, no split-K, aligned rows");
    const int WF4 = BC * KS * BM4, TOT = WF4 + BC * XW4;
    const int ni = p.db_ni;
    const int buf_f = ni * NW * 256;  // floats per buffer
    const char* src[NI];
    int stp[NI];  // signed: channel-flipped views have negative channel strides
    unsigned xmask = 0;  // bit i: this lane's slot of piece i is an X slot (pre-activation applies)
    {
      const int wstep = BC * KS * a.CoutP * 4, xstep = (int)(BC * a.x_cs * 4);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int sl = (i * NW + wave) * 64 + lane;
        const float* q = p.zero;
        int st = 0;
        if (sl < WF4) {
          const int row = sl / BM4, c4 = sl - row * BM4;
          if (co0 + c4 * 4 < a.CoutP) {
            q = wph + (long long)row * a.CoutP + co0 + c4 * 4;
            st = wstep;
          }
        } else if (sl < TOT) {
          const int sx = sl - WF4;
          const int r = sx / XW4, c4 = sx - r * XW4;
          const int tin = tin_base + c4 * 4;
          if (tin >= 0 && tin < a.Tin) {
            q = xb + (long long)r * a.x_cs + tin;
            st = xstep;
            xmask |= 1u << i;
          }
        }
        src[i] = reinterpret_cast<const char*>(q);
        stp[i] = st;
      }
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);
    auto issue = [&](int buf) {
      const unsigned d0 = lds_base + (unsigned)buf * (unsigned)buf_f * 4u + wave_u * 1024u;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (i < ni) {
          glds16(src[i], (unsigned)__builtin_amdgcn_readfirstlane((int)(d0 + (unsigned)(i * NW) * 1024u)));
          src[i] += stp[i];
        }
      }
    };
    // Pre-activation: each wave applies it IN PLACE to the X slots it fetched itself (its own vmcnt covers them, no
    // barrier needed), right after the MFMA loop of the previous chunk; the barrier at the top of the loop publishes it.
    // (For 1..3 taps the chunks are short and that pass — a dependent LDS round trip per chunk — costs more than applying
    // the activation to each operand as it is read: 2 VALU per B operand, hidden under the MFMAs.)
    // (round 3, profiles/r04l_convbench_*: the pre-activation costs a 256-channel launch 8..12 us whatever the tap count and
    //  whichever way it is applied — on read for every k, or the in-place pass for every k >= 3: 48.3 / 89 / 124 us either way
    //  against 40 / 77.5 / 113 us for plain input)
    //  (on read WITH sched groups that place the VALU: 83 us at k = 7, no change at k = 3 / 11, profiles/r04m_*: the register
    //  allocator reuses the B registers tap after tap, which rules the look-ahead out)
    constexpr bool ACT_ON_READ = KSC >= 1 && KSC <= 3;
    const float ps = a.pre_slope;
    const bool act = !ACT_ON_READ && ps != 1.f && !(dbg & 8);
    auto activate = [&](int buf) {
      float* d0 = smem + buf * buf_f + wave * 256 + lane * 4;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (i < ni && ((xmask >> i) & 1u)) {
          float4* q = reinterpret_cast<float4*>(d0 + i * NW * 256);
          float4 v = *q;
          v.x = fmaxf(v.x, v.x * ps); v.y = fmaxf(v.y, v.y * ps); v.z = fmaxf(v.z, v.z * ps); v.w = fmaxf(v.w, v.w * ps);
          *q = v;
        }
      }
    };
    issue(0);
    svc_vmcnt0();
    if (act) activate(0);
    // The chunk loop is instantiated per activation form (ACT = leaky-ReLU applied to each B operand as it is read, 1..3
    // taps with a slope != 1) and the form chosen ONCE, outside it: chosen per chunk, the two MFMA loops kept the 64
    // accumulator registers in different places and every chunk paid 64 v_mov_b64 to move them over and back.
    auto chunks = [&](auto act_tag) {
    constexpr bool ACT = decltype(act_tag)::value;
    int it = 0;
    for (int c0 = 0; c0 < a.Cin; c0 += BC, ++it) {
      __syncthreads();  // chunk `it` is in LDS (landed + activated by its owners); everyone is done reading the other buffer
      if (c0 + BC < a.Cin && !(dbg & 1)) issue((it + 1) & 1);
      const float* wbuf = wbase + (it & 1) * buf_f;
      const float* xbuf = xbase + (it & 1) * buf_f;
      const int n_cc = BC / KPI;
      // (round 3: ONE continuous operand pipeline over the chunk — reads of step s + 2 in front of the MFMAs of step s across
      //  channel pairs, operand ring, sched_barrier between groups — instead of restarting at every channel pair: slower on
      //  every shape, 1 x 1 convs included (profiles/r03z_trainconv.txt vs r03y_*): the per-pair restart is not what holds
      //  the short-reduction shapes back)
      if constexpr (MMA != SVC_MMA_F32) {
        // ---- 16-bit operands (svc_conv1d_args.mma = SVC_MMA_BF16 / SVC_MMA_F16): v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulate.  The chunk sits in
        // LDS exactly as for the fp32 loop (fp32, [ci][k][BM] weights and [ci][XW] activations, fetched by LDS-DMA); an
        // instruction reduces 16 input channels of one tap, lane half lk supplying channels 8*lk .. 8*lk + 7 of the group: eight
        // ds_read_b32 per fragment, rounded in pairs (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, round to nearest even) on the way
        // into the operand registers.  Same C layout as the fp32 instruction: the epilogues are shared.
        static_assert(MMA == SVC_MMA_F32 || (DB && KSC == 0 && !M16), "16-bit operands: LDS-DMA tilings, run-time tap count");
        typedef Op16<MMA == SVC_MMA_F32 ? SVC_MMA_BF16 : MMA> OP;
        const int dil = a.dil, n_g = BC >> 4, wrow = KS * BM;
        for (int g = 0; g < n_g; ++g) {
          const float* wa = wbuf + (g * 16 + 8 * lk) * wrow;
          const float* xa = xbuf + (g * 16 + 8 * lk) * XW;
          for (int k = 0; k < KS; ++k) {
            typename OP::frag af[MT], bq[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
              f32x8v t;
#pragma unroll
              for (int j = 0; j < 8; ++j) t[j] = wa[j * wrow + k * BM + i * TS];
              af[i] = OP::cvt(t);
            }
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) {
              f32x8v t;
#pragma unroll
              for (int j = 0; j < 8; ++j) t[j] = xa[j * XW + k * dil + jn * TS];
              bq[jn] = OP::cvt(t);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int jn = 0; jn < NT; ++jn)
                acc32[i][jn] = OP::mfma(af[i], bq[jn], acc32[i][jn]);
          }
        }
      } else if constexpr (KSC > 0) {
        const int dil = a.dil;
        // ACT: leaky-ReLU applied to each B operand as it is read (1..3 taps, see above).  Instantiated twice: plain inputs
        // (slope 1: every training conv, the encoder / flow convs) must not pay its 3 VALU per operand — nor their effect on
        // the schedule: the VALU sat outside the sched groups below and the k = 3 loop ran at ~100 cycles per MFMA against
        // ~70 for k = 5, which has no such pass (profiles/r04b_trainconv_dbg.txt).
        if constexpr (!ACT && ring_steps(KSC) > 0) {
          // Operand ring: one pipeline over the chunk's (channel pair, tap) steps — the reads of step s + RING_D are issued in
          // front of the MFMAs of step s, across channel pairs; R steps (compile-time slots) = R / KSC channel pairs per loop
          // trip.  The per-pair form below restarts its pipeline at every pair (one exposed LDS latency per KSC * MT * NT
          // MFMAs); which form wins is per tap count and measured (ring_steps()).  Same MFMA order either way: bit-equal.
          constexpr int R = ring_steps(KSC) > 0 ? ring_steps(KSC) : KSC, D = RING_D;   // (never 0: the discarded branch is still parsed)
          constexpr int QPB = R / KSC;                 // channel pairs per trip (launcher: BC % (KPI * QPB) == 0)
          const int n_blk = n_cc / QPB;
          constexpr int wblk = QPB * KPI * KSC * BM;
          const int xblk = QPB * KPI * XW;
          const float* wq = wbuf + lk * (KSC * BM);
          const float* xq = xbuf + lk * XW;
          float av[R][MT], bv[R][NT];
          auto fetch = [&](int r, const float* wb, const float* xb_) {
            const int ql = r / KSC, k = r % KSC;
#pragma unroll
            for (int i = 0; i < MT; ++i) av[r][i] = wb[ql * (KPI * KSC * BM) + k * BM + i * TS];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[r][j] = xb_[ql * KPI * XW + k * dil + j * TS];
          };
#pragma unroll
          for (int r = 0; r < D; ++r) fetch(r, wq, xq);
          for (int blk = 0; blk < n_blk; ++blk) {
            const bool more = blk + 1 < n_blk;         // past the chunk's last trip the look-ahead re-reads this trip (unused)
            const float* wn = more ? wq + wblk : wq;
            const float* xn = more ? xq + xblk : xq;
#pragma unroll
            for (int r = 0; r < R; ++r) {
              if (r + D < R) fetch(r + D, wq, xq);
              else fetch(r + D - R, wn, xn);
#pragma unroll
              for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                  acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r][i], bv[r][j], acc32[i][j], 0, 0, 0);
            }
            constexpr int DSPT = (MT + 1) / 2 + (NT + 1) / 2;
#pragma unroll
            for (int r = 0; r < R; ++r) {
              __builtin_amdgcn_sched_group_barrier(0x100, DSPT, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
            }
            wq = wn;
            xq = xn;
          }
        } else
        {
          for (int q = 0; q < n_cc; ++q) {
            const int cl = q * KPI + lk;
            const float* wa = wbuf + cl * (KSC * BM);
            const float* xa = xbuf + cl * XW;
            float av[KSC][MT], bv[KSC][NT];
            auto fetch = [&](int k) {
#pragma unroll
              for (int i = 0; i < MT; ++i) av[k][i] = wa[k * BM + i * TS];
#pragma unroll
              for (int j = 0; j < NT; ++j) bv[k][j] = xa[k * dil + j * TS];
            };
            if constexpr (ACT) {
              // explicit order (fences): the sched groups below do not place the activation's VALU — left to itself the
              // scheduler waited for every read and chained the three MFMAs of one accumulator back to back
              fetch(0);
              if constexpr (KSC > 1) fetch(1);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int k = 0; k < KSC; ++k) {
#pragma unroll
                for (int j = 0; j < NT; ++j) bv[k][j] = fmaxf(bv[k][j], bv[k][j] * ps);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                  for (int j = 0; j < NT; ++j)
                    acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k][i], bv[k][j], acc32[i][j], 0, 0, 0);
                if (k + 2 < KSC) fetch(k + 2);
                __builtin_amdgcn_sched_barrier(0);
              }
            } else {
#pragma unroll
              for (int k = 0; k < KSC; ++k) fetch(k);
#pragma unroll
              for (int k = 0; k < KSC; ++k)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                  for (int j = 0; j < NT; ++j)
                    acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k][i], bv[k][j], acc32[i][j], 0, 0, 0);
              if constexpr (KSC >= 3) {
                constexpr int DSPT = (MT + 1) / 2 + (NT + 1) / 2;
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * DSPT, 0);
#pragma unroll
                for (int k = 0; k < KSC; ++k) {
                  __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
                  if (k + 2 < KSC) __builtin_amdgcn_sched_group_barrier(0x100, DSPT, 0);
                }
              }
            }
          }
        }
      } else {
        const int n_it = n_cc * KS;
        const int a_step_cc = KPI * KS * BM - KS * BM;
        const int b_step_cc = KPI * XW - KS * a.dil;
        int a_off = (lk * KS) * BM;
        int b_off = lk * XW;
        int k = 0;
        for (int itk = 0; itk < n_it; ++itk) {
          float av[MT], bv[NT];
#pragma unroll
          for (int i = 0; i < MT; ++i) av[i] = wbuf[a_off + i * TS];
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            bv[j] = xbuf[b_off + j * TS];
          }
          a_off += BM;
          b_off += a.dil;
          if (++k == KS) {
            k = 0;
            a_off += a_step_cc;
            b_off += b_step_cc;
          }
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc32[i][j], 0, 0, 0);
        }
      }
      if (c0 + BC < a.Cin) {
        svc_vmcnt0();  // this wave's pieces of chunk it+1 have landed (they had the whole MFMA loop to do so)
        if (act) activate((it + 1) & 1);
      }
    }
    };
    if constexpr (ACT_ON_READ) {
      if (ps != 1.f) chunks(std::true_type{});
      else chunks(std::false_type{});
    } else {
      chunks(std::false_type{});
    }
  } else {
  const int cur = BC;
  load_chunk(0);
  for (int c0 = 0; c0 < a.Cin; c0 += cur) {
    const int n_cc = cur / (KPI * WK);  // k-groups per wave per tap
    __syncthreads();  // everyone finished reading the previous chunk from LDS
    if (!(dbg & 1) || c0 == 0) store_chunk(c0);
    __syncthreads();
    if (c0 + cur < a.Cin && !(dbg & 1)) load_chunk(c0 + cur);  // in flight during the MFMA loop below
    if (dbg & 2) continue;

    if constexpr (KSC > 0) {
      // compile-time tap count: all KS operand pairs of a channel group are fetched up front, then KS x MT x NT
      // MFMAs issue back to back (the LDS latency is paid once per group, not once per tap)
      const int dil = a.dil;
      for (int q = 0; q < n_cc; ++q) {
        const int cl = q * (KPI * WK) + wk * KPI + lk;
        const float* wa = wbase + cl * (KSC * BM);
        const float* xa = xbase + cl * XW;
        float av[KSC][MT], bv[KSC][NT];
#pragma unroll
        for (int k = 0; k < KSC; ++k) {
#pragma unroll
          for (int i = 0; i < MT; ++i) av[k][i] = wa[k * BM + i * TS];
#pragma unroll
          for (int j = 0; j < NT; ++j) bv[k][j] = xa[k * dil + j * TS];
        }
#pragma unroll
        for (int k = 0; k < KSC; ++k)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              if constexpr (M16)
                acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k][i], bv[k][j], acc16[i][j], 0, 0, 0);
              else
                acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k][i], bv[k][j], acc32[i][j], 0, 0, 0);
            }
        // Instruction order inside this block: the LDS operand reads of tap k+2 are issued BEFORE the MFMAs of tap k
        // (left alone, the scheduler sinks each tap's reads next to its MFMAs to save registers, and the matrix pipe
        // idles for an LDS round trip per tap whenever no second wave is resident on the SIMD).
        if constexpr (KSC >= 3) {
          constexpr int DSPT = (MT + 1) / 2 + (NT + 1) / 2;   // ds_read(2)_b32 instructions per tap
          __builtin_amdgcn_sched_group_barrier(0x100, 2 * DSPT, 0);
#pragma unroll
          for (int k = 0; k < KSC; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
            if (k + 2 < KSC) __builtin_amdgcn_sched_group_barrier(0x100, DSPT, 0);
          }
        }
      }
    } else {
      const int n_it = n_cc * KS;
      const int a_step_cc = (KPI * WK) * KS * BM - KS * BM;  // extra A offset when k wraps
      const int b_step_cc = (KPI * WK) * XW - KS * a.dil;
      int a_off = ((wk * KPI + lk) * KS) * BM;
      int b_off = (wk * KPI + lk) * XW;
      int k = 0;
      for (int it = 0; it < n_it; ++it) {
        float av[MT], bv[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) av[i] = wbase[a_off + i * TS];
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[j] = xbase[b_off + j * TS];
        a_off += BM;
        b_off += a.dil;
        if (++k == KS) {
          k = 0;
          a_off += a_step_cc;
          b_off += b_step_cc;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            if constexpr (M16) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc16[i][j], 0, 0, 0);
            else acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc32[i][j], 0, 0, 0);
          }
      }
    }
  }

  }

  // ---- epilogue: accumulators -> LDS ([WK][BM][CP], the MFMA C layout has one time column per lane), then every
  // thread owns 4 consecutive time steps of one output row: 16 B residual loads / stores, full cache lines per row.
  // Split-K partial tiles are summed here (fixed order wk = 0..WK-1: deterministic).
  if ((dbg & 4) && acc_probe(acc32[0][0], acc16[0][0]) != 12345.678f) return;
  if constexpr (DB) {
    if (p.direct_epi) {
      conv_epilogue_direct<MT, NT>(p, acc32, b, t0, co0 + wm * (MT * TS), wn * (NT * TS), lane);
      return;
    }
  }
  constexpr int CP = BN + 4;
  __syncthreads();
  {
    float* cw = smem + (wk * BM + wm * (MT * TS)) * CP + wn * (NT * TS) + ln;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
          const int row = i * TS + (M16 ? 4 * lk + r : (r & 3) + 8 * (r >> 2) + 4 * lk);
          if constexpr (M16) cw[row * CP + j * TS] = acc16[i][j][r];
          else cw[row * CP + j * TS] = acc32[i][j][r];
        }
  }
  __syncthreads();

  // (batched epilogue loads: -4..5 us per launch on the 64x128 / 64x256 / 32x512 / 128x128 tiles, +3..5 us on 128x224 where the
  // extra live registers push the kernel to the 256-VGPR line and the main loop's allocation suffers: not used there)
  conv_epilogue<BM, BN, NTHR, WK, EPI, DB && NT != 7>(p, smem, tid, b, ph, t0, co0);
}

thread_local int t_row_phases = 1;  // set by svc_conv_transpose1d_f32 around its dispatch (see ConvP::row_phases)
template <int MT, int NT, int WM, int WN, int WK, bool M16, int EPI, int KSC, bool XVEC, bool DB = false, int MMA = SVC_MMA_F32>
__global__ __launch_bounds__(WM* WN* WK * 64, (WM * WN * WK <= 4 && NT != 7 && (XVEC || WK > 1) ? 2 : 1)) void conv1d_mfma_kernel(ConvP p) {
  conv1d_mfma_body<MT, NT, WM, WN, WK, M16, EPI, KSC, XVEC, DB, MMA>(p, blockIdx.x);
}

int g_bf16_enabled = 1;    // svc_debug_bf16(0) forces fp32 operands whatever the calls ask for (A/B)
int g_bf16_launches = 0;   // launches that ran with bf16 operands (tests ask through svc_debug_bf16(-1))
int g_force_cfg = -1;  // debug/tuning override (svc_debug_set_conv_cfg)
int g_no_ksc = 0;      // debug: 1 disables the compile-time-KS kernels
int g_dbg = 0;         // debug: ConvP.dbg
int g_db_budget_kb = 64;
int g_db_mode = 1;     // 1: use the LDS-DMA double-buffered kernels where eligible (svc_debug_set_conv_cfg: +10000 disables)
int g_no224 = 1;       // 1: the 128x224 one-workgroup-per-CU tile stays out of the selection (svc_debug_set_conv_cfg: +100000000
                       // re-admits it).  Launched alone it wins 8..13 % for KS >= 7 at 128 channels (one round of 247 tiles), but it
                       // owns a whole CU (223 VGPR + 112 AGPR, 116 KB LDS): with the decoder's three MRF chains on concurrent
                       // streams the 128x128 tile (two workgroups per CU, from different launches) lets one launch's epilogue
                       // overlap another's MFMA loop: clip 8.29 / 8.44 ms with 128x224 vs 8.17 / 8.19 ms without (same box)
int g_no192 = -1;     // 1: never pick the 64x192 tiling (A/B switch: environment SVC_CONV_NO192=1, read at the first launch)
int g_direct_epi = 1;  // 1: DB kernels store straight from the accumulators where the epilogue form allows (svc_debug_set_conv_cfg: +1000000000 disables)
int g_fast_epi = 1;    // 1: DB kernels batch the epilogue's residual loads (svc_debug_set_conv_cfg: +10000000 disables)
int g_direct_mode = 1; // 1: short-sequence split-K shapes run the register-fed direct kernel (svc_debug_set_conv_cfg: +1000000 disables)

template <int MT, int NT, int WM, int WN, int WK, bool M16, int EPI = SVC_EPI_PLAIN, int KSC = 0>
int launch_cfg(const svc_conv1d_args& a, hipStream_t s) {
  constexpr int TS = M16 ? 16 : 32;
  constexpr int KPI = M16 ? 4 : 2;
  constexpr int NACC = M16 ? 4 : 16;
  constexpr int BM = WM * MT * TS;
  constexpr int BN = WN * NT * TS;
  constexpr int NTHR = WM * WN * WK * 64;
  constexpr int KG = KPI * WK;
  ConvP p;
  p.a = a;
  const bool xvec = a.premask == nullptr && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (a.x_bs % 4) == 0 &&
                    (a.x_cs % 4) == 0 && (a.Tin % 4) == 0;
  p.xvec = xvec ? 1 : 0;
  p.row_phases = t_row_phases;
  if (p.row_phases > 1 && (BM % p.row_phases) != 0) {
    svc::set_error("conv1d: %d row phases do not divide the %d-row tile", p.row_phases, BM);
    return SVC_ERR_UNSUPPORTED;
  }
  p.dbg = g_dbg;
  auto al4 = [](const void* ptr, long long bs, long long cs) {
    return ptr == nullptr || ((reinterpret_cast<uintptr_t>(ptr) & 15) == 0 && (bs % 4) == 0 && (cs % 4) == 0);
  };
  p.yvec = (a.n_phase == 1 && a.y_ts == 1 && a.y_t0 == 0 && al4(a.y, a.y_bs, a.y_cs) && al4(a.res, a.res_bs, a.res_cs) &&
            al4(a.y2, a.y2_bs, a.y2_cs)) ? 1 : 0;
  int xw = BN + (a.KS - 1) * a.dil + (xvec ? 3 : 0);
  xw = (xw + 3) & ~3;
  if (M16) {  // keep consecutive channel rows on disjoint bank halves for the 16-lane groups
    while ((xw & 31) != 16) xw += 4;
  }
  p.XW = xw;
  // BC: as many channels per chunk as the staging registers (WLD / XLD4 slots per thread) and ~64 KiB of LDS allow
  constexpr int NWV = WM * WN * WK;
  constexpr int RSTEP = NTHR / (BM / 4);
  const int per_c = (a.KS * BM + xw) * 4;
  constexpr int XCIV = (BN + MAXHALO + 3 + 255) / 256, XCIS = (BN + MAXHALO + 63) / 64;
  const int x_rows = xvec ? (XLD4 / XCIV) : ((4 * XLD4) / XCIS);   // channel-row iterations the X slots cover
  if ((a.KS - 1) * a.dil > MAXHALO) {
    svc::set_error("conv1d: (KS-1)*dil = %d exceeds the supported halo %d", (a.KS - 1) * a.dil, MAXHALO);
    return SVC_ERR_UNSUPPORTED;
  }
  // LDS budget per workgroup: 64 KiB, or what the epilogue transpose buffer needs anyway when that is larger
  const int lds_budget = std::max(64 * 1024, WK * BM * (BN + 4) * 4);
  int bc = lds_budget / per_c;
  bc = std::min(bc, (WLD * RSTEP) / a.KS);
  bc = std::min(bc, x_rows * NWV);
  bc = (bc / KG) * KG;
  if (bc < KG) bc = KG;
  const int cin_r = ((a.Cin + KG - 1) / KG) * KG;
  if (bc > cin_r) bc = cin_r;
  // balance the chunks (e.g. Cin=128, bc=48 -> 3 chunks of 44 instead of 48+48+32)
  {
    const int nch = (cin_r + bc - 1) / bc;
    int bal = (cin_r + nch - 1) / nch;
    bal = ((bal + KG - 1) / KG) * KG;
    if (bal < bc) bc = bal;
  }
  p.BC = bc;
  if (x_rows < 1 || bc * a.KS > WLD * RSTEP || (bc + NWV - 1) / NWV > x_rows) {
    svc::set_error("conv1d: tile does not fit the staging registers (KS=%d dil=%d Cin=%d)", a.KS, a.dil, a.Cin);
    return SVC_ERR_UNSUPPORTED;
  }
  p.n_t_tiles = svc::cdiv(a.Tout, BN);
  p.n_m_tiles = svc::cdiv(a.Cout, BM);
  const long long nblk = (long long)p.n_t_tiles * p.n_m_tiles * a.B * a.n_phase;

  // ---- LDS-DMA double-buffered variant (DB): aligned rows, whole chunks, pre-activation expressible as max(x, s*x) ----
  if constexpr (!M16 && WK == 1) {
    const size_t epi_bytes = (size_t)BM * (BN + 4) * 4;
    // (tried: 80 KiB for the 128x128 tiles, two per CU, half as many chunks / barriers: no measurable change)
    // (round 3: a 128 KiB budget — twice the chunk, half the hand-overs — for launches of at most one workgroup per CU: no
    //  change on any shape, profiles/r03y_trainconv_solo{64,128}.txt)
    size_t budget = std::max((size_t)(BM * BN >= 128 * 128 ? g_db_budget_kb : 64) * 1024, epi_bytes);
    // bf16 operands are built for the tilings the batched (training) convolutions take; anything else stays fp32
    constexpr bool BF16_TILING = EPI == SVC_EPI_PLAIN && ((MT == 2 && NT == 2 && WM == 2 && WN == 2) || (MT == 2 && NT == 1 && WM == 1 && WN == 4) ||
                                                          (MT == 1 && NT == 3 && WM == 2 && WN == 2) || (MT == 1 && NT == 5 && WM == 4 && WN == 1) ||
                                                          (MT == 2 && NT == 2 && WM == 1 && WN == 4));
    bool want_bf16 = BF16_TILING && (a.mma == SVC_MMA_BF16 || a.mma == SVC_MMA_F16) && (a.Cin % 16) == 0 &&
                     g_bf16_enabled;
    auto set_epi = [&]() {
      p.fast_epi = (g_fast_epi && p.row_phases == 1 && a.epi == SVC_EPI_PLAIN && a.mask == nullptr && (a.res_mode == 0 || a.res_mode == 1) &&
                    (a.cond == nullptr || a.cond_ts == 0) && p.yvec && (a.Tout % 4) == 0 && a.Tout >= 4) ? 1 : 0;
      p.direct_epi = (g_direct_epi && p.row_phases == 1 && a.epi == SVC_EPI_PLAIN && a.mask == nullptr && (a.res_mode == 0 || a.res_mode == 1) &&
                      (a.cond == nullptr || a.cond_ts == 0) && a.n_phase == 1 && a.y_ts == 1 && a.y_t0 == 0 && (a.Cout % 32) == 0 &&
                      (a.post_act == SVC_ACT_NONE || (a.post_act == SVC_ACT_LRELU && a.post_slope >= 0.f && a.post_slope <= 1.f)) &&
                      a.y_cs >= 0 && a.y_cs < (1ll << 24) && a.res_cs >= 0 && a.res_cs < (1ll << 24)) ? 1 : 0;
    };
    // a bf16 chunk is at least 16 channels (one instruction's reduction): 5..11 taps of them do not fit the 64 KiB the fp32
    // chunks are sized for — they take up to 144 KiB (one workgroup per CU; the instruction stream is 16x shorter per chunk)
    const size_t base_budget = budget;
    if (want_bf16) budget = std::max(budget, (size_t)144 * 1024);
    bool use_db = g_db_mode != 0 && xvec && a.pre_slope >= 0.f && a.pre_slope <= 1.f && (a.Cin % KG) == 0 &&
                  std::llabs((long long)a.x_cs) * 4 * 64 < (1ll << 31) && (long long)a.CoutP * a.KS * 4 * 64 < (1ll << 31);
    int bcd = 0, ni = 0;
    if (use_db) {
      // largest chunk (multiple of KG dividing Cin) whose two buffers fit the budget and 16 pieces per wave
      constexpr int CQF = (KSC > 0 && ring_steps(KSC) > 0) ? KG * (ring_steps(KSC) / KSC) : KG;   // whole ring trips per chunk
      auto find_chunk = [&](int CQ) {                                                             // bf16: 16 channels per MFMA
        for (int c = std::min(a.Cin, 64) / CQ * CQ; c >= CQ; c -= CQ) {
          if (a.Cin % c) continue;
          const int tot4 = c * (a.KS * BM + xw) / 4;
          const int n = svc::cdiv(tot4, NWV * 64);
          if (n <= 16 && (size_t)2 * n * NWV * 1024 <= budget) { bcd = c; ni = n; break; }
        }
      };
      find_chunk(want_bf16 ? 16 : CQF);
      if (!bcd && want_bf16) {
        // no 16-channel chunk fits the piece count (7 / 11 taps on a 128-row tile): the fp32 form of THIS kernel, with its own chunk
        // rule — falling through to the register-staged kernel cost 164 against 129 us (profiles/r09h_conv_mma_modes.txt)
        want_bf16 = false;
        budget = base_budget;
        find_chunk(CQF);
      }
      use_db = bcd > 0;
    }
    if (use_db) {
      static const float* zero = nullptr;
      if (!zero) {
        void* zp = nullptr;
        if (hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero_block)) != hipSuccess) return svc::check_launch("conv1d zero block");
        zero = static_cast<const float*>(zp);
      }
      p.BC = bcd;
      p.db_ni = ni;
      set_epi();
      p.zero = zero;
      p.dump_off = 0;
      const size_t lds = std::max((size_t)2 * ni * NWV * 1024, epi_bytes);
      if constexpr (BF16_TILING) {
        if (want_bf16) {
          auto kb = conv1d_mfma_kernel<MT, NT, WM, WN, WK, M16, EPI, 0, true, true, SVC_MMA_BF16>;
          auto kh = conv1d_mfma_kernel<MT, NT, WM, WN, WK, M16, EPI, 0, true, true, SVC_MMA_F16>;
          static bool doneb = false;
          if (!doneb) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(kh), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            doneb = true;
          }
          ++g_bf16_launches;
          if (a.mma == SVC_MMA_F16) hipLaunchKernelGGL(kh, dim3((unsigned)nblk), dim3(NTHR), lds, s, p);
          else hipLaunchKernelGGL(kb, dim3((unsigned)nblk), dim3(NTHR), lds, s, p);
          return svc::check_launch("conv1d_mfma_16bit");
        }
      }
      auto kd = conv1d_mfma_kernel<MT, NT, WM, WN, WK, M16, EPI, KSC, true, true>;
      if (lds > 64 * 1024) {
        static bool done = false;
        if (!done) {
          hipFuncSetAttribute(reinterpret_cast<const void*>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          done = true;
        }
      }
      hipLaunchKernelGGL(kd, dim3((unsigned)nblk), dim3(NTHR), lds, s, p);
      return svc::check_launch("conv1d_mfma_db");
    }
  }

  size_t lds = (size_t)bc * per_c;
  lds = std::max(lds, (size_t)WK * BM * (BN + 4) * 4);   // epilogue transpose buffer
  p.dump_off = (int)(lds / 4);
  lds += (size_t)NTHR * 16;                              // per-thread dump slots
  if (lds > 160 * 1024) {
    svc::set_error("conv1d: LDS tile too large (KS=%d dil=%d)", a.KS, a.dil);
    return SVC_ERR_UNSUPPORTED;
  }
  auto kv = conv1d_mfma_kernel<MT, NT, WM, WN, WK, M16, EPI, KSC, true>;
  auto ks = conv1d_mfma_kernel<MT, NT, WM, WN, WK, M16, EPI, KSC, false>;
  if (lds > 64 * 1024) {
    static bool done = false;
    if (!done) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(kv), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipFuncSetAttribute(reinterpret_cast<const void*>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      done = true;
    }
  }
  if (xvec) hipLaunchKernelGGL(kv, dim3((unsigned)nblk), dim3(NTHR), lds, s, p);
  else hipLaunchKernelGGL(ks, dim3((unsigned)nblk), dim3(NTHR), lds, s, p);
  return svc::check_launch("conv1d_mfma");
}


// ---- short-sequence kernel: operands straight from global memory / L2 into registers --------------------------------------
// The encoder / flow convs of a single 10 s utterance have only B*T = 862 columns and 192..768 rows: at most a few hundred
// 32x32 tiles, each with a long reduction (Cin*KS up to 2304).  The LDS-staged kernels above are built for thousands of
// columns; here their per-chunk staging passes (sized for 128-wide tiles) and two barriers per chunk cost more than the
// MFMAs of the chunk (8 chunks x ~3.5 us for a 768 -> 192 k3 conv = 30 us for 2 us of matrix work).  This kernel has NO
// staging and NO barrier in its main loop: a workgroup owns a (32*MT) x 32 output tile, its 4 waves split the reduction
// (input channels), and every wave feeds its MFMAs from its own registers:
//     A (weights): packed [Cin][KS][CoutP] -> lane ln reads row (c, k), column co0 + ln: 128 contiguous bytes per half wave
//     B (input)  : lane ln reads x[c][t0 + ln + k*dil - pad]: 128 contiguous bytes per half wave, any alignment
// (c = the lane's channel of the pair the instruction consumes).  Loads run DEPTH reduction steps ahead of the MFMAs that
// use them (two register banks, straight-line code), so L2 latency is hidden by the wave itself; zero padding, the channel
// tail and the leaky-ReLU / pre-mask prologue are applied to the B value in registers.  The partial tiles meet in LDS once,
// in the shared epilogue.
template <int MT, int EPI, int KSC>
__global__ __launch_bounds__(256) void conv1d_mfma_direct_kernel(ConvP p) {
  static_assert(KSC >= 1 && KSC <= 7, "direct kernel: compile-time tap count");
  // PR channel pairs x KSC taps = one register bank (12..16 reduction steps, 36..48 loads in flight per wave)
  constexpr int PR = KSC == 1 ? 8 : (KSC <= 3 ? 5 : (KSC <= 5 ? 3 : 2));
  constexpr int NS = PR * KSC;
  constexpr int WK = 4, BM = MT * 32, BN = 32, NTHR = 256, CP = BN + 4;
  const svc_conv1d_args& a = p.a;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wk = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: everything derived from it stays scalar
  const int ln = lane & 31, lk = lane >> 5;
  int bid = blockIdx.x;
  const int tt = bid % p.n_t_tiles;
  bid /= p.n_t_tiles;
  const int mtile = bid % p.n_m_tiles;
  const int b = bid / p.n_m_tiles;
  const int t0 = tt * BN, co0 = mtile * BM;
  const float ps = a.pre_slope;

  // this wave's slice of the channel pairs (Cin is even: checked by the launcher)
  const int npairs = a.Cin >> 1;
  const int ppw = (npairs + WK - 1) / WK;
  const int pr0 = wk * ppw;
  const int my_pairs = max(0, min(ppw, npairs - pr0));
  const int n_it = (my_pairs + PR - 1) / PR;

  f32x16 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // Addressing: ALL per-step address arithmetic is scalar.  A load is  (wave-uniform base in SGPRs) + (per-lane 32-bit byte
  // offset in a VGPR that never changes): the lane offsets below are computed once, the bases advance with s_add per step.
  //   weights  w[((2*pr + lk)*KS + k)*CoutP + col]  = base_w(pr, k) + [lk*KS*CoutP + col]
  //   input    x[chan(2*pr + lk)*xcs + clamp(t)]    = base_x(pr)    + [lk'*xcs + clamp(t0 + ln + k*dil - pad)]
  // (a channel-flipped view — negative channel stride — is re-based on its lowest address, channels mirrored: lk' = 1 - lk).
  const bool xflip = a.x_cs < 0;
  const unsigned xcs = (unsigned)(xflip ? -a.x_cs : a.x_cs);
  const char* xlo = reinterpret_cast<const char*>(a.x + (long long)b * a.x_bs + (xflip ? (long long)(a.Cin - 1) * a.x_cs : 0ll));
  const char* wlo = reinterpret_cast<const char*>(a.w);
  unsigned vw[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) vw[i] = 4u * ((unsigned)(lk * KSC * a.CoutP) + (unsigned)min(co0 + i * 32 + ln, a.CoutP - 1));
  unsigned vx[KSC];
  float okt[KSC];   // zero padding as a multiply by 0 / 1 (a select on the loaded value would let the compiler sink the load
                    // under a branch, and a load under a branch costs an s_waitcnt vmcnt(0) at every step)
#pragma unroll
  for (int k = 0; k < KSC; ++k) {
    const int tin = t0 + ln - a.pad_left + k * a.dil;
    vx[k] = 4u * ((unsigned)(xflip ? 1 - lk : lk) * xcs + (unsigned)min(max(tin, 0), a.Tin - 1));
    okt[k] = (tin >= 0 && tin < a.Tin) ? 1.f : 0.f;
  }

  // register banks: raw loads only — nothing in load_bank may USE a loaded value (that would put an s_waitcnt inside the
  // load block); zero padding and the leaky-ReLU prologue are applied in mfma_bank, right before the MFMA
  float av[2][NS][MT], bx[2][NS];
  auto load_bank = [&](int it, float (&A_)[NS][MT], float (&X_)[NS]) {
#pragma unroll
    for (int j = 0; j < PR; ++j) {
      const int pr = pr0 + min(it * PR + j, my_pairs - 1);            // scalar; pairs past the slice re-read the last one
      const char* wb = wlo + 4ull * (unsigned)(2 * pr * KSC * a.CoutP);
      const char* xbp = xlo + 4ull * ((unsigned)(xflip ? a.Cin - 2 - 2 * pr : 2 * pr) * xcs);
#pragma unroll
      for (int k = 0; k < KSC; ++k) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          A_[j * KSC + k][i] = *reinterpret_cast<const float*>(wb + 4ull * (unsigned)(k * a.CoutP) + vw[i]);
        X_[j * KSC + k] = *reinterpret_cast<const float*>(xbp + vx[k]);
      }
    }
  };
  auto mfma_bank = [&](int it, const float (&A_)[NS][MT], const float (&X_)[NS]) {
#pragma unroll
    for (int j = 0; j < PR; ++j) {
      const float live = it * PR + j < my_pairs ? 1.f : 0.f;           // scalar
#pragma unroll
      for (int k = 0; k < KSC; ++k) {
        const float xv = X_[j * KSC + k];
        const float bvu = fmaxf(xv, xv * ps) * (okt[k] * live);        // leaky-ReLU for 0 <= slope <= 1 (launcher checks)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[j * KSC + k][i], bvu, acc[i], 0, 0, 0);
      }
    }
  };
  // Every load_bank is UNCONDITIONAL (banks past the slice re-read the last pair and are not consumed): a load issued under
  // a branch would make the compiler's vmcnt bookkeeping wait for ALL outstanding loads before the next MFMAs, i.e.
  // serialise L2 latency with the matrix work.  The sched_barriers pin the order [loads of bank i+1][MFMAs of bank i]: left
  // alone, the scheduler sinks every load next to its MFMA to save registers (vmcnt(2) in front of each MFMA).
  if (n_it > 0) {
    load_bank(0, av[0], bx[0]);
    for (int it = 0; it < n_it; it += 2) {
      __builtin_amdgcn_sched_barrier(0);
      load_bank(it + 1, av[1], bx[1]);
      __builtin_amdgcn_sched_barrier(0);
      mfma_bank(it, av[0], bx[0]);
      __builtin_amdgcn_sched_barrier(0);
      load_bank(it + 2, av[0], bx[0]);
      __builtin_amdgcn_sched_barrier(0);
      if (it + 1 < n_it) mfma_bank(it + 1, av[1], bx[1]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  {
    float* cw = smem + wk * BM * CP + ln;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) cw[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * CP] = acc[i][r];
  }
  __syncthreads();
  conv_epilogue<BM, BN, NTHR, WK, EPI>(p, smem, tid, b, 0, t0, co0);
}

// ---- the same with 16-byte operand loads (round 6) ---------------------------------------------------------------------------
// What bounds the kernel above (profiles/r11e_front_conv_hot_cold.txt, r11d_*): not the matrix pipe and not L2 — hot and cold
// operands time the same, and 16 waves per workgroup instead of 4 change nothing.  Every operand is a 4-byte-per-lane load and a
// CU's address unit takes one wave-wide vector-memory instruction per ~16 cycles whatever its width: 2 (3) loads per step x the
// workgroup's steps x 16 cycles IS the measured time (768 -> 192 x 3 taps: 2304 loads = 15 us of the launch's 23; 35-54 TFLOP/s
// on every shape).  Here a lane's A operands of FOUR consecutive steps are one 16-byte load out of a lane-linear second pack of the
// weights (svc_pack_conv1d_d4) and its B operands of all KSC taps of a channel are one KSC-float load (taps at dilation 1 are
// consecutive samples): 1.75 instead of 6 loads per channel pair at 3 taps and a 32-row tile, 2.5 instead of 9 at 64 rows.
// Zero padding and the edge columns: the B vector is fetched from a start clamped into the row, b = clamp(ts, 0, Tin - KSC), and a
// lane whose window was moved (delta = ts - b != 0: only in the first / last column tile) picks element k + delta per tap — a
// compare / select chain that waves without such lanes skip (wave-uniform branch chosen once, outside the loop).
template <int N>
__device__ __forceinline__ void load_taps(const char* q, float (&d)[N]) {   // N consecutive floats from a 4-byte aligned address
  struct __attribute__((packed, aligned(4))) V { float v[N]; };
  const V t = *reinterpret_cast<const V*>(q);
#pragma unroll
  for (int i = 0; i < N; ++i) d[i] = t.v[i];
}

__host__ __device__ constexpr int direct4_pairs(int mt, int ksc) { return (ksc <= 3 && mt == 1) ? 8 : 4; }
// sched groups "one vector-memory read, then its share of NMF MFMAs", NLD times
template <int G, int NLD, int NMF>
__device__ __forceinline__ void interleave_groups() {
  if constexpr (G < NLD) {
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    constexpr int q = (G + 1) * NMF / NLD - G * NMF / NLD;
    if constexpr (q > 0) __builtin_amdgcn_sched_group_barrier(0x008, q, 0);
    interleave_groups<G + 1, NLD, NMF>();
  }
}

template <int MT, int EPI, int KSC>
__global__ __launch_bounds__(256) void conv1d_mfma_direct4_kernel(ConvP p) {
  static_assert(KSC >= 1 && KSC <= 7, "direct kernel: compile-time tap count");
  constexpr int PR = direct4_pairs(MT, KSC);  // channel pairs per register bank; PR * KSC steps = NL 16-byte A loads per row tile
  constexpr int NS = PR * KSC, NL = NS / 4;
  static_assert(NS % 4 == 0, "a bank is a whole number of 4-step groups");
  constexpr int WK = 4, BM = MT * 32, BN = 32, NTHR = 256, CP = BN + 4;
  const svc_conv1d_args& a = p.a;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wk = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 31, lk = lane >> 5;
  int bid = blockIdx.x;
  const int tt = bid % p.n_t_tiles;
  bid /= p.n_t_tiles;
  const int mtile = bid % p.n_m_tiles;
  const int b = bid / p.n_m_tiles;
  const int t0 = tt * BN, co0 = mtile * BM;
  const float ps = a.pre_slope;

  const int npairs = a.Cin >> 1;
  const int ppw = npairs / WK;                 // launcher: npairs % (WK * PR) == 0
  const int pr0 = wk * ppw;
  const int n_it = ppw / PR;

  f32x16 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const bool xflip = a.x_cs < 0;
  const unsigned xcs = (unsigned)(xflip ? -a.x_cs : a.x_cs);
  const char* xlo = reinterpret_cast<const char*>(a.x + (long long)b * a.x_bs + (xflip ? (long long)(a.Cin - 1) * a.x_cs : 0ll));
  // A: group G of row tile rt is 1 KiB at ((rt * NG + G) * 64 + lane) * 16
  const int NG = (npairs * KSC) >> 2;
  const int n_rt = a.CoutP >> 5;
  const char* wrow[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int rt = __builtin_amdgcn_readfirstlane(min((co0 >> 5) + i, n_rt - 1));
    wrow[i] = reinterpret_cast<const char*>(a.w_d4) + (size_t)rt * NG * 1024;
  }
  const unsigned va = (unsigned)lane * 16u;
  // B: the lane's KSC-sample window starts at ts; fetched from b0 (inside the row), element k + delta is tap k
  const int ts = t0 + ln - a.pad_left;
  const int b0 = min(max(ts, 0), a.Tin - KSC);
  const int delta = ts - b0;
  const unsigned vxb = 4u * ((unsigned)(xflip ? 1 - lk : lk) * xcs + (unsigned)b0);
  const bool edge = __any(delta != 0);

  f32x4 av[3][NL][MT];
  float bx[3][PR][KSC];
  auto load_bank = [&](int it, f32x4 (&A_)[NL][MT], float (&X_)[PR][KSC]) {
    const int pr = pr0 + min(it, n_it - 1) * PR;                       // scalar; banks past the slice re-read the last one
    const unsigned g0 = (unsigned)(pr * KSC) >> 2;
    // program order = order of first use (A group l feeds steps 4l .., pair j's taps steps j*KSC ..): with the loads spread over the
    // previous bank's MFMAs every operand is then issued one bank ahead of its MFMA
    auto load_a = [&](int l) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
        A_[l][i] = *reinterpret_cast<const f32x4*>(wrow[i] + (size_t)(g0 + l) * 1024 + va);
    };
#pragma unroll
    for (int j = 0; j < PR; ++j) {
#pragma unroll
      for (int l = 0; l < NL; ++l)
        if (4 * l <= j * KSC && (j == 0 || 4 * l > (j - 1) * KSC)) load_a(l);
      const char* xbp = xlo + 4ull * ((unsigned)(xflip ? a.Cin - 2 - 2 * (pr + j) : 2 * (pr + j)) * xcs);
      load_taps<KSC>(xbp + vxb, X_[j]);
    }
#pragma unroll
    for (int l = 0; l < NL; ++l)
      if (4 * l > (PR - 1) * KSC) load_a(l);
  };
  // NA accumulators per row tile, step s on accumulator s % NA: a chain of MFMAs on ONE accumulator with the operand VALU between
  // them runs at ~110 cycles per instruction instead of 64 (profiles/r11h_mfma_pattern_few_accumulators.txt, r11i_*: the
  // 768 -> 192 x 3 launch spent 13.4 us in its 288 chained MFMAs per wave); the partial sums are added once, after the loop.
  constexpr int NA = MT == 1 ? 4 : 2;
  f32x16 accs[MT][NA];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int u = 0; u < NA; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) accs[i][u][r] = 0.f;
  auto run = [&](auto edge_tag, auto act_tag) {
    constexpr bool EDGE = decltype(edge_tag)::value, ACT = decltype(act_tag)::value;
    auto mfma_bank = [&](const f32x4 (&A_)[NL][MT], const float (&X_)[PR][KSC]) {
#pragma unroll
      for (int j = 0; j < PR; ++j) {
#pragma unroll
        for (int k = 0; k < KSC; ++k) {
          float xv;
          if constexpr (EDGE) {
            xv = 0.f;
#pragma unroll
            for (int e = 0; e < KSC; ++e) xv = (delta == e - k) ? X_[j][e] : xv;
          } else {
            xv = X_[j][k];
          }
          const float bvu = ACT ? fmaxf(xv, xv * ps) : xv;               // leaky-ReLU for 0 <= slope <= 1 (launcher checks)
#pragma unroll
          for (int i = 0; i < MT; ++i)
            accs[i][(j * KSC + k) % NA] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[(j * KSC + k) >> 2][i][(j * KSC + k) & 3], bvu,
                                                                              accs[i][(j * KSC + k) % NA], 0, 0, 0);
        }
      }
    };
    // The loads of the next bank are spread EVENLY between the MFMAs of this one (sched groups: one load, then its share of the
    // MFMAs).  Issued as one burst in front of them, the 14..40 loads of a bank block the wave at the address unit's queue (four
    // waves share it) for as long as the unit needs to take them, and a wave that is blocked issues no MFMAs: loads and matrix
    // work of the launch ran one after the other, 6.5 + 13.4 us (profiles/r11i_front_conv_decomp.txt).
    constexpr int NLD = NL * MT + PR * (KSC > 4 ? 2 : 1), NMF = NS * MT;
    // THREE banks: every operand is issued two banks (~3000 cycles at 32 rows) ahead of its MFMA — a clip uses each weight once,
    // they come from HBM, and one bank of look-ahead left the launch as slow cold as the 4-byte form (r11j_front_conv_d4_*)
    load_bank(0, av[0], bx[0]);
    load_bank(1, av[1], bx[1]);
    int it = 0;
    for (; it + 2 < n_it; it += 3) {
      __builtin_amdgcn_sched_barrier(0);
      load_bank(it + 2, av[2], bx[2]);
      mfma_bank(av[0], bx[0]);
      interleave_groups<0, NLD, NMF>();
      __builtin_amdgcn_sched_barrier(0);
      load_bank(it + 3, av[0], bx[0]);
      mfma_bank(av[1], bx[1]);
      interleave_groups<0, NLD, NMF>();
      __builtin_amdgcn_sched_barrier(0);
      load_bank(it + 4, av[1], bx[1]);
      mfma_bank(av[2], bx[2]);
      interleave_groups<0, NLD, NMF>();
    }
    __builtin_amdgcn_sched_barrier(0);
    if (it < n_it) mfma_bank(av[0], bx[0]);          // one or two banks left: fetched by the last trip (or the two loads above)
    if (it + 1 < n_it) mfma_bank(av[1], bx[1]);
  };
  if (n_it > 0) {
    // (plain inputs — every encoder / flow convolution — must not pay the activation's VALU between their MFMAs)
    if (ps != 1.f) {
      if (edge) run(std::true_type{}, std::true_type{});
      else run(std::false_type{}, std::true_type{});
    } else {
      if (edge) run(std::true_type{}, std::false_type{});
      else run(std::false_type{}, std::false_type{});
    }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    acc[i] = accs[i][0];
#pragma unroll
    for (int u = 1; u < NA; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] += accs[i][u][r];
  }

  {
    float* cw = smem + wk * BM * CP + ln;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) cw[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * CP] = acc[i][r];
  }
  __syncthreads();
  conv_epilogue<BM, BN, NTHR, WK, EPI>(p, smem, tid, b, 0, t0, co0);
}

// shapes the register-fed kernel takes: dense (one phase), no pre-mask, 1 / 3 / 5 / 7 taps, an even channel count, a
// pre-activation expressible as max(x, slope*x), and 32-bit byte offsets inside one batch row of x and inside the weights
static bool direct_ok(const svc_conv1d_args& a) {
  return a.n_phase == 1 && a.premask == nullptr && (a.KS == 1 || a.KS == 3 || a.KS == 5 || a.KS == 7 || (a.KS == 2 && a.epi == SVC_EPI_PLAIN)) && (a.Cin % 2) == 0 &&
         a.pre_slope >= 0.f && a.pre_slope <= 1.f && (long long)a.Cin * a.KS * a.CoutP < (1ll << 29) &&
         std::llabs((long long)a.x_cs) * a.Cin + a.Tin < (1ll << 29);
}

template <int MT, int EPI, int KSC>
int launch_direct(const svc_conv1d_args& a, hipStream_t s) {
  constexpr int BM = MT * 32, BN = 32, WK = 4;
  ConvP p;
  memset(&p, 0, sizeof(p));
  p.a = a;
  p.row_phases = t_row_phases;
  if (p.row_phases > 1 && (BM % p.row_phases) != 0) {
    svc::set_error("conv1d: %d row phases do not divide the %d-row tile", p.row_phases, BM);
    return SVC_ERR_UNSUPPORTED;
  }
  auto al4 = [](const void* ptr, long long bs, long long cs) {
    return ptr == nullptr || ((reinterpret_cast<uintptr_t>(ptr) & 15) == 0 && (bs % 4) == 0 && (cs % 4) == 0);
  };
  p.yvec = (a.n_phase == 1 && a.y_ts == 1 && a.y_t0 == 0 && al4(a.y, a.y_bs, a.y_cs) && al4(a.res, a.res_bs, a.res_cs) &&
            al4(a.y2, a.y2_bs, a.y2_cs)) ? 1 : 0;
  p.n_t_tiles = svc::cdiv(a.Tout, BN);
  p.n_m_tiles = svc::cdiv(a.Cout, BM);
  const long long nblk = (long long)p.n_t_tiles * p.n_m_tiles * a.B;
  const size_t lds = (size_t)WK * BM * (BN + 4) * 4;
  hipLaunchKernelGGL((conv1d_mfma_direct_kernel<MT, EPI, KSC>), dim3((unsigned)nblk), dim3(256), lds, s, p);
  return svc::check_launch("conv1d_mfma_direct");
}

// the 16-byte-load form: the second pack is there, taps are consecutive samples, and every wave's slice of the channel pairs is a
// whole number of register banks (Cin a multiple of 32; of 64 for 32-row tiles up to three taps)
int g_direct4 = -1;   // 0: never (A/B switch: environment SVC_CONV_DIRECT4=0, read at the first launch)
template <int MT>
static bool direct4_ok(const svc_conv1d_args& a) {
  if (g_direct4 < 0) {
    const char* e = getenv("SVC_CONV_DIRECT4");
    g_direct4 = (e && e[0] == '0') ? 0 : 1;
  }
  const int pr = direct4_pairs(MT, a.KS);
  return g_direct4 && a.w_d4 != nullptr && a.dil == 1 && a.Tin >= a.KS && ((a.Cin / 2) % (4 * pr)) == 0 && (a.CoutP % 32) == 0 &&
         (reinterpret_cast<uintptr_t>(a.w_d4) & 15) == 0;
}

template <int MT, int EPI, int KSC>
int launch_direct4(const svc_conv1d_args& a, hipStream_t s) {
  constexpr int BM = MT * 32, BN = 32, WK = 4;
  ConvP p;
  memset(&p, 0, sizeof(p));
  p.a = a;
  p.row_phases = t_row_phases;
  if (p.row_phases > 1 && (BM % p.row_phases) != 0) {
    svc::set_error("conv1d: %d row phases do not divide the %d-row tile", p.row_phases, BM);
    return SVC_ERR_UNSUPPORTED;
  }
  auto al4 = [](const void* ptr, long long bs, long long cs) {
    return ptr == nullptr || ((reinterpret_cast<uintptr_t>(ptr) & 15) == 0 && (bs % 4) == 0 && (cs % 4) == 0);
  };
  p.yvec = (a.n_phase == 1 && a.y_ts == 1 && a.y_t0 == 0 && al4(a.y, a.y_bs, a.y_cs) && al4(a.res, a.res_bs, a.res_cs) &&
            al4(a.y2, a.y2_bs, a.y2_cs)) ? 1 : 0;
  p.n_t_tiles = svc::cdiv(a.Tout, BN);
  p.n_m_tiles = svc::cdiv(a.Cout, BM);
  const long long nblk = (long long)p.n_t_tiles * p.n_m_tiles * a.B;
  const size_t lds = (size_t)WK * BM * (BN + 4) * 4;
  hipLaunchKernelGGL((conv1d_mfma_direct4_kernel<MT, EPI, KSC>), dim3((unsigned)nblk), dim3(256), lds, s, p);
  return svc::check_launch("conv1d_mfma_direct4");
}

thread_local int* t_probe = nullptr;   // svc_conv1d_wants_d4: the dispatch runs up to its decision and reports it here, launching nothing
template <int MT, int EPI>
int launch_direct_ks(const svc_conv1d_args& a, hipStream_t s) {
  if (t_probe) {
    *t_probe = direct4_ok<MT>(a) ? 1 : 0;
    return SVC_OK;
  }
  if (direct4_ok<MT>(a)) {
    switch (a.KS) {
      case 1: return launch_direct4<MT, EPI, 1>(a, s);
      case 2:
        if constexpr (EPI == SVC_EPI_PLAIN) return launch_direct4<MT, EPI, 2>(a, s);
        else return SVC_ERR_UNSUPPORTED;
      case 3: return launch_direct4<MT, EPI, 3>(a, s);
      case 5: return launch_direct4<MT, EPI, 5>(a, s);
      default: return launch_direct4<MT, EPI, 7>(a, s);
    }
  }
  switch (a.KS) {
    case 1: return launch_direct<MT, EPI, 1>(a, s);
    case 2:   // (plain epilogue only, direct_ok(): the two taps per phase of the decoder's x8 ConvTranspose1d stages)
      if constexpr (EPI == SVC_EPI_PLAIN) return launch_direct<MT, EPI, 2>(a, s);
      else return SVC_ERR_UNSUPPORTED;
    case 3: return launch_direct<MT, EPI, 3>(a, s);
    case 5: return launch_direct<MT, EPI, 5>(a, s);
    default: return launch_direct<MT, EPI, 7>(a, s);
  }
}

}  // namespace

extern "C" int svc_debug_set_conv_cfg(int cfg) {
  // cfg = nodb*10000 + dbg*1000 + noksc*100 + (forced tile config + 1), 0 / negative = defaults
  if (cfg <= 0) { g_force_cfg = -1; g_no_ksc = 0; g_dbg = 0; g_db_mode = 1; g_db_budget_kb = 64; g_direct_mode = 1; g_fast_epi = 1; g_no224 = 1; g_direct_epi = 1; return SVC_OK; }
  g_direct_epi = ((cfg / 1000000000) % 10) ? 0 : 1;
  g_fast_epi = ((cfg / 10000000) % 10) ? 0 : 1;
  g_no224 = ((cfg / 100000000) % 10) ? 0 : 1;
  g_direct_mode = ((cfg / 1000000) % 10) ? 0 : 1;
  g_force_cfg = cfg % 100 - 1;
  g_no_ksc = (cfg / 100) % 10;
  g_dbg = (cfg / 1000) % 10;
  g_db_mode = ((cfg / 10000) % 10) ? 0 : 1;
  g_db_budget_kb = ((cfg / 100000) % 10) ? 80 : 64;
  return SVC_OK;
}

static int conv1d_dispatch(const svc_conv1d_args& a, void* stream) {
  SVC_REQUIRE(a.x && a.w && a.y, "conv1d: null tensor");
  SVC_REQUIRE(a.B > 0 && a.Cin > 0 && a.Cout > 0 && a.Tin > 0 && a.Tout > 0, "conv1d: empty shape");
  SVC_REQUIRE(a.KS >= 1 && a.dil >= 1, "conv1d: bad KS/dil");
  SVC_REQUIRE(a.n_phase >= 1 && a.y_ts >= 1 && a.y_len >= 1, "conv1d: bad n_phase/y_ts/y_len");
  SVC_REQUIRE(a.CoutP >= a.Cout && (a.CoutP % 4) == 0, "conv1d: CoutP must be >= Cout and a multiple of 4");
  SVC_REQUIRE((reinterpret_cast<uintptr_t>(a.w) & 15) == 0, "conv1d: packed weight must be 16B aligned");
  SVC_REQUIRE(a.res_mode == 0 || a.res != nullptr, "conv1d: res_mode set but res is null");
  hipStream_t s = (hipStream_t)stream;
  const double flop = 2.0 * a.B * (double)a.Cout * a.Cin * a.KS * a.Tout * a.n_phase;
  const double bytes = 4.0 * a.B * ((double)a.Cin * a.Tin + (double)a.Cout * a.Tout) + 4.0 * a.Cin * a.KS * a.Cout;
  char pname[160];
  if (svc::prof_on() && svc::prof_shapes())   // SVC_PROF_SHAPES=1: one profile row per shape (tuning aid)
    snprintf(pname, sizeof(pname), "%s[B%d,Ci%d,Co%d,K%d,d%d,T%d,e%d]", (a.n_phase > 1 || t_row_phases > 1) ? "convt1d_mfma" : "conv1d_mfma", a.B,
             a.Cin, a.Cout, a.KS, a.dil, a.Tout, a.epi);
  else
    snprintf(pname, sizeof(pname), "%s", (a.n_phase > 1 || t_row_phases > 1) ? "convt1d_mfma" : "conv1d_mfma");
  svc::ProfScope prof(s, pname, flop, bytes, t_probe == nullptr);

  if (g_no192 < 0) {
    const char* e = getenv("SVC_CONV_NO192");
    g_no192 = (e && e[0] == '1') ? 1 : 0;
  }
  // long sequences that one round of 224-column strips covers (the decoder's MRF convs): conv1d_strip.hip
  if (g_force_cfg < 0 && t_row_phases == 1) {      // (the strip kernel's epilogue knows nothing of phases-as-rows outputs)
    const int rs = svc::conv1d_strip_try(a, s, t_probe != nullptr);
    if (rs <= 0) {
      if (t_probe) *t_probe = 0;
      return rs;
    }
  }
  const long long cols = (long long)a.B * a.Tout;
  // workgroup counts of the candidate tilings for short sequences
  const long long wg64x128 = (long long)svc::cdiv(a.Cout, 64) * svc::cdiv(a.Tout, 128) * a.B * a.n_phase;
  const long long wg64x32 = (long long)svc::cdiv(a.Cout, 64) * svc::cdiv(a.Tout, 32) * a.B * a.n_phase;
  const bool m64 = (a.Cout % 64) == 0;
  int cfg;  // 0: 16x512  1: 32x512  2: 64x256  3: 128x128  4: 64x128  5: 64x32 split-K  6: 32x32 split-K
            // 7: 128x224 (one workgroup per CU, 7 MFMA column tiles per wave)
  if (a.epi == SVC_EPI_GATE) {
    SVC_REQUIRE((a.Cout % 64) == 0, "conv1d: gate epilogue needs Cout %% 64 == 0 (got %d)", a.Cout);
    SVC_REQUIRE(a.res_mode == 0, "conv1d: gate epilogue takes no residual");
    cfg = cols >= 16384 ? 3 : (wg64x128 >= 200 ? 4 : 5);
  } else if (a.epi == SVC_EPI_RES_SKIP) {
    SVC_REQUIRE(a.res && a.y2, "conv1d: res_skip epilogue needs res and y2");
    cfg = cols >= 16384 ? 3 : (wg64x128 >= 200 && m64 ? 4 : (m64 && wg64x32 >= 200 ? 5 : 6));
  } else {
    SVC_REQUIRE(a.epi == SVC_EPI_PLAIN, "conv1d: unknown epilogue %d", a.epi);
    if (a.Cout <= 16) cfg = cols >= 16384 ? 0 : 6;   // few columns (DiscriminatorP's 1024 -> 1 conv_post): split the reduction instead
    else if (a.Cout <= 32) cfg = cols >= 16384 ? 1 : 6;
    else if (cols >= 16384) cfg = a.Cout <= 64 ? 2 : 3;
    // rows of at most 64 columns (the last layers of the discriminators: 1024 -> 1024 x 5 taps on T = 32, 2048 -> 512 x 2 taps on
    // T = 16, B = 16 / 32 rows): a 128-column tile is >= 50 % padding there however many workgroups the launch has — 64 x 32 /
    // 32 x 32 split-K instead: 192 -> 66 us, 340 -> 137 us (B = 32), 168 -> 38 us (profiles/r06c_train_small_conv_sweep.txt)
    else if (m64 && a.Tout <= 64) cfg = wg64x32 >= 200 ? 5 : 6;
    else if (m64 && wg64x128 >= 128) cfg = 4;   // measured (768 -> 2304/3072, T=500): 64x128 at 144..192 workgroups beats
                                                // the 64x32 split-K tiling by 1.5x (4x the operand reuse per LDS byte)
    // (more than 7 taps run on the LDS-staged split-K tiles, where 32 x 32 beats 64 x 32 until the launch is large:
    //  256 -> 256 x 11 taps, T = 128, B = 16: 79 -> 65 us)
    else if (m64 && wg64x32 >= (a.KS > 7 ? 512 : 200)) cfg = 5;
    else cfg = 6;
    // Wide outputs (Cout > 64) with enough columns to fill the chip: pick among 128x128, 64x128 and 128x224 by modelled
    // time = (rounds of 256 CUs) * tile area.  Two effects it captures, both measured: (1) the decoder's lengths are
    // 862 * 2^k samples, so power-of-two tiles leave the last round ~2/3 full (431 tiles of 128) while 128x224 covers the
    // sequence in ONE round of one workgroup per CU (247 tiles; +8..13 % for KS >= 7, -6 % for KS = 3 where its
    // un-overlapped staging phases weigh more); (2) for batched medium sequences (training: B=16, T~1000) the smaller
    // 64x128 tile fills all CUs where 128x128 / 128x224 would occupy a third of them (76 vs 48 vs 30 TFLOP/s).
    if (a.Cout > 64 && a.n_phase == 1 && (cfg == 3 || cfg == 4)) {
      const double n128 = (double)svc::cdiv(a.Cout, 128) * svc::cdiv(a.Tout, 128) * a.B;
      const double n64 = (double)svc::cdiv(a.Cout, 64) * svc::cdiv(a.Tout, 128) * a.B;
      const double n224 = (double)svc::cdiv(a.Cout, 128) * svc::cdiv(a.Tout, 224) * a.B;
      double best = std::ceil(n128 / 256.0) * 128 * 128;
      cfg = 3;
      if (m64) {
        const double t64 = std::ceil(n64 / 256.0) * 64 * 128 * 1.05;
        if (t64 < best) { best = t64; cfg = 4; }
      }
      if (a.KS >= 7 && n224 >= 200 && !g_no224) {
        const double t224 = std::ceil(n224 / 256.0) * 128 * 224 * 1.04;
        if (t224 < best) { best = t224; cfg = 7; }
      }
      // 64x192 (three column tiles per wave): the training graph's 192- and 384-row convolutions at B x T = 16 x 768 —
      // 6 (12) row tiles x 384 column tiles = 2.25 (4.5) tiles per SIMD, which 2x2-tile waves round up to 4 (8) tile times
      // and 3-tile waves to 3 (6).  Batched launches only: a single utterance's stages are tuned separately (§4).
      // 128x160 (five column tiles per wave): DiscriminatorP's period-11 maps are 132 columns long (B = 16 / 32 rows of them per
      // launch, 1024 channels) — two 128-column tiles per row waste 48 % of the second one, one 160-column tile 17 %
      if (a.B > 1 && !g_no192) {
        const double n160 = (double)svc::cdiv(a.Cout, 128) * svc::cdiv(a.Tout, 160) * a.B;
        const double t160 = std::ceil(n160 / 256.0) * 128 * 160 * 1.03;
        if (t160 < best) { best = t160; cfg = 10; }
      }
      if (m64 && a.B > 1 && !g_no192) {
        const double n192 = (double)svc::cdiv(a.Cout, 64) * svc::cdiv(a.Tout, 192) * a.B;
        const double t192 = std::ceil(n192 / 256.0) * 64 * 192 * 1.03;
        if (t192 < best) { best = t192; cfg = 9; }
      }
    }
  }
  {
    // the wide-tile configs stage X with float4 rows; unaligned activations fall back to narrower tiles
    const bool xvec = a.premask == nullptr && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (a.x_bs % 4) == 0 &&
                      (a.x_cs % 4) == 0 && (a.Tin % 4) == 0;
    if (!xvec && a.epi == SVC_EPI_PLAIN) {
      if (cfg == 0 || cfg == 1) cfg = 6;
      else if (cfg == 2) cfg = 4;
      else if (cfg == 7) cfg = 3;
    }
  }
  {
    // tuning aid: SVC_CONV_C256_CFG=<cfg> forces a tiling for the decoder's 256-channel stage only (in-situ A/B of that stage)
    static int c256 = -2;
    if (c256 == -2) {
      const char* e = getenv("SVC_CONV_C256_CFG");
      c256 = e ? atoi(e) : -1;
    }
    if (c256 >= 0 && a.Cout == 256 && a.Cin == 256 && a.epi == SVC_EPI_PLAIN && cols < 16384 && cols > 2048 && t_row_phases == 1) cfg = c256;
  }
  if (g_force_cfg >= 0 && a.Cout > 16) {
    const bool ok = (g_force_cfg == 3 || g_force_cfg == 4 || g_force_cfg == 5) ||
                    (a.epi != SVC_EPI_GATE && ((g_force_cfg <= 7 && g_force_cfg >= 1) || g_force_cfg == 9 || g_force_cfg == 10));
    if (ok) cfg = g_force_cfg;
  }
  // phases-as-rows ConvTranspose1d (two taps per phase): the register-fed kernel for the short first stage and for the x2
  // stages, whose <= 128 rows make them streaming work (profiles/r04h_ups_*: 75 vs 98 us at 512 -> 256 x8 on 862 frames,
  // 52 vs 72 us at 64 -> 32 x2, 58 vs 65 us at 32 -> 16 x2); the LDS-staged tiles for the 1024-row x8 stage (127 us)
  if (t_row_phases > 1 && g_force_cfg < 0 && a.epi == SVC_EPI_PLAIN && direct_ok(a) && g_direct_mode && (cols < 16384 || a.Cout <= 128))
    return (a.Cout % 64) == 0 ? launch_direct_ks<2, SVC_EPI_PLAIN>(a, s) : launch_direct_ks<1, SVC_EPI_PLAIN>(a, s);
  if ((cfg == 5 || cfg == 6 || (a.epi == SVC_EPI_GATE && cfg != 3 && cfg != 4)) && direct_ok(a) && g_direct_mode) {
    const bool two = cfg == 5 || a.epi == SVC_EPI_GATE;
    if (a.epi == SVC_EPI_GATE) return launch_direct_ks<2, SVC_EPI_GATE>(a, s);
    if (a.epi == SVC_EPI_RES_SKIP) return two ? launch_direct_ks<2, SVC_EPI_RES_SKIP>(a, s) : launch_direct_ks<1, SVC_EPI_RES_SKIP>(a, s);
    return two ? launch_direct_ks<2, SVC_EPI_PLAIN>(a, s) : launch_direct_ks<1, SVC_EPI_PLAIN>(a, s);
  }
  if (t_probe) {       // an LDS-staged tiling: no use for the lane-linear pack
    *t_probe = 0;
    return SVC_OK;
  }
  if (a.epi == SVC_EPI_GATE) {
    switch (cfg) {
      case 3: return launch_cfg<2, 2, 2, 2, 1, false, SVC_EPI_GATE>(a, s);
      case 4: return launch_cfg<2, 1, 1, 4, 1, false, SVC_EPI_GATE>(a, s);
      default: return launch_cfg<2, 1, 1, 1, 4, false, SVC_EPI_GATE>(a, s);
    }
  }
  if (a.epi == SVC_EPI_RES_SKIP) {
    switch (cfg) {
      case 3: return launch_cfg<2, 2, 2, 2, 1, false, SVC_EPI_RES_SKIP>(a, s);
      case 4: return launch_cfg<2, 1, 1, 4, 1, false, SVC_EPI_RES_SKIP>(a, s);
      case 5: return launch_cfg<2, 1, 1, 1, 4, false, SVC_EPI_RES_SKIP>(a, s);
      default: return launch_cfg<1, 1, 1, 1, 4, false, SVC_EPI_RES_SKIP>(a, s);
    }
  }
#define SVC_KS_CASES(MT_, NT_, WM_, WN_, M16_)                                                      \
  switch (g_no_ksc ? 0 : a.KS) {                                                                     \
    case 1: return launch_cfg<MT_, NT_, WM_, WN_, 1, M16_, SVC_EPI_PLAIN, 1>(a, s);                  \
    case 2: return launch_cfg<MT_, NT_, WM_, WN_, 1, M16_, SVC_EPI_PLAIN, 2>(a, s);                  \
    case 5: return launch_cfg<MT_, NT_, WM_, WN_, 1, M16_, SVC_EPI_PLAIN, 5>(a, s);                  \
    case 3: return launch_cfg<MT_, NT_, WM_, WN_, 1, M16_, SVC_EPI_PLAIN, 3>(a, s);                  \
    case 7: return launch_cfg<MT_, NT_, WM_, WN_, 1, M16_, SVC_EPI_PLAIN, 7>(a, s);                  \
    case 11: return launch_cfg<MT_, NT_, WM_, WN_, 1, M16_, SVC_EPI_PLAIN, 11>(a, s);                \
    default: return launch_cfg<MT_, NT_, WM_, WN_, 1, M16_, SVC_EPI_PLAIN, 0>(a, s);                 \
  }
  switch (cfg) {
    case 0: SVC_KS_CASES(1, 8, 1, 4, true)    // 16 x 512
    case 1: SVC_KS_CASES(1, 4, 1, 4, false)   // 32 x 512
    case 2: SVC_KS_CASES(2, 2, 1, 4, false)   // 64 x 256
    case 3: SVC_KS_CASES(2, 2, 2, 2, false)   // 128 x 128
    case 4: SVC_KS_CASES(2, 1, 1, 4, false)   // 64 x 128
    case 5: return launch_cfg<2, 1, 1, 1, 4, false>(a, s);   // 64 x 32, 4-way split-K
    case 7: SVC_KS_CASES(1, 7, 4, 1, false)   // 128 x 224
    case 9: SVC_KS_CASES(1, 3, 2, 2, false)   // 64 x 192
    case 10: SVC_KS_CASES(1, 5, 4, 1, false)  // 128 x 160
    // (round 3: a 128 x 128 tile with 8 waves, two per 64 x 64 wave tile splitting the reduction, for the 256-channel stage
    //  whose 128 x 128 tiling covers only 108 CUs: 224 vs 112 us at k = 11 on the register-staged path — not kept,
    //  profiles/r03t_cfg8_*)
    default: return launch_cfg<1, 1, 1, 1, 4, false>(a, s);  // 32 x 32, 4-way split-K
  }
}

extern "C" int svc_conv1d_wants_d4(const svc_conv1d_args* ap) {
  if (ap == nullptr) return 0;
  svc_conv1d_args a = *ap;
  a.w_d4 = reinterpret_cast<const float*>(static_cast<uintptr_t>(16));   // "a pack is there": the question is whether the launch would read one
  int res = 0;
  t_probe = &res;
  const int rc = conv1d_dispatch(a, nullptr);
  t_probe = nullptr;
  return rc == SVC_OK ? res : 0;
}

extern "C" int svc_conv1d_f32(const svc_conv1d_args* ap, void* stream) {
  SVC_REQUIRE(ap != nullptr, "conv1d: null args");
  return conv1d_dispatch(*ap, stream);
}

extern "C" int svc_debug_bf16(int mode) {
  if (mode < 0) return g_bf16_launches;
  g_bf16_enabled = mode ? 1 : 0;
  return SVC_OK;
}

// ConvTranspose1d as `stride` dense polyphase sub-convolutions (vdecoder/hifigan/models.py:340-342,378):
//   y[co, q*u + p - pad] = sum_ci sum_m x[ci, q - m] * W[ci, co, p + m*u],   m < M = ceil(KS/u)
// i.e. phase p is a conv with M taps (time-reversed) and pad_left = M-1, writing every u-th output sample.
extern "C" int svc_conv_transpose1d_f32(const svc_convt1d_args* ap, void* stream) {
  SVC_REQUIRE(ap != nullptr, "convt1d: null args");
  const svc_convt1d_args& t = *ap;
  SVC_REQUIRE(t.x && t.w && t.y, "convt1d: null tensor");
  SVC_REQUIRE(t.stride >= 1 && t.KS >= 1 && t.padding >= 0, "convt1d: bad stride/KS/padding");
  const int u = t.stride;
  const int M = (t.KS + u - 1) / u;
  const int Lout = (t.Tin - 1) * u - 2 * t.padding + t.KS;
  SVC_REQUIRE(Lout == t.Tout, "convt1d: Tout=%d but (Tin-1)*stride-2*padding+KS=%d", t.Tout, Lout);
  svc_conv1d_args a;
  memset(&a, 0, sizeof(a));
  a.x = t.x; a.w = t.w; a.bias = t.bias; a.res = t.res; a.y = t.y;
  // (the pack follows the phases-as-rows layout only; measured per stage of a 10 s clip, profiles/r11l_*: 512 -> 256 x8 70 -> 55 us,
  //  256 -> 128 x8 126 -> 107 us, but the x2 stages — 28 MB in, 28 MB out, 128 / 64 input channels — 76 -> 82 and 51 -> 68 us:
  //  they stream activations, and three register banks cost them their second wave per SIMD)
  a.w_d4 = (svc::convt_rows_layout(t.stride) && t.Cin >= 256) ? t.w_d4 : nullptr;
  a.x_bs = t.x_bs; a.x_cs = t.x_cs; a.y_bs = t.y_bs; a.y_cs = t.y_cs; a.res_bs = t.res_bs; a.res_cs = t.res_cs;
  a.B = t.B; a.Cin = t.Cin; a.Cout = t.Cout; a.Tin = t.Tin;
  a.Tout = (Lout - 1 + t.padding) / u + 1;   // number of q positions
  a.KS = M; a.dil = 1; a.pad_left = M - 1; a.CoutP = t.CoutP;
  a.epi = SVC_EPI_PLAIN; a.post_act = SVC_ACT_NONE; a.res_mode = t.res ? 1 : 0;
  a.pre_slope = t.pre_slope; a.post_slope = 0.f; a.beta = 0.f; a.out_div = 1.f;
  a.y_t0 = -t.padding; a.y_len = Lout;
  if (svc::convt_rows_layout(u)) {
    // phases as output ROWS of ONE dense convolution (weight packed [Cin][M][u*CoutP], row = co*u + phase): every workgroup
    // holds all u phases of its channels and writes u*BN consecutive samples per channel.  The phase-per-workgroup form
    // below stored every u-th float from u different workgroups: 150 us for the 512 -> 256 x8 stage of a 10 s clip (24 TFLOP/s)
    // and 62..76 us for the three x2 stages, which move 28 MB each (profiles/r03u_infer_*).
    a.Cout = u * t.Cout; a.CoutP = u * t.CoutP;
    a.n_phase = 1; a.y_ts = 1; a.w_phase_stride = 0;
    t_row_phases = u;
    const int rc = conv1d_dispatch(a, stream);
    t_row_phases = 1;
    return rc;
  }
  a.n_phase = u; a.y_ts = u;
  a.w_phase_stride = (long long)t.Cin * M * t.CoutP;
  return conv1d_dispatch(a, stream);
}
