// flow_fused.hip — one launch per coupling layer of the normalizing flow, for the 16-bit and the split inference modes.
//
// What it replaces (reference): modules/modules.py:288-307 ResidualCouplingLayer.forward (mean_only), i.e. `pre` 1x1 (:291),
// WN.forward :110-138 — per layer the weight-normed k = 5 conv 192 -> 384, `+ g_l`, fused_add_tanh_sigmoid_multiply, the 1x1 res/skip
// conv, `x = (x + res) * mask`, `output += skip` — then `post` 1x1 (:297) and the affine update (:300-306), with the channel Flip
// (:232-239) of models.py:45-52 folded into the caller's channel stride.  The fp32 path runs this as 10 launches per coupling (41 per
// flow), each a 7-17 us latency chain on 862 columns: 0.5 ms per clip for 2 % of its FLOPs.
//
// Why only in the 16-bit / split modes.  Fusing a coupling means one workgroup owns ALL rows of a column tile (the gate needs a tanh row
// and its sigmoid row, the 1x1 needs all 192 gated rows, the next layer all 192 residual rows): 862 columns are 18 such workgroups.
// On the fp32 matrix instruction 18 CUs deliver 11 TFLOP/s — the coupling's 3.05 GFLOP would take 280 us, twice the unfused launches
// that spread rows over 160-320 workgroups.  v_mfma_f32_32x32x16_f16 is 16x faster per CU (the split form, 3 instructions per
// product, 5.3x): there the same 18 workgroups finish a coupling in the time its weights (3.5 MB fp16, 7 MB split) stream from L2.
//
// Structure.  A workgroup (4 waves) owns NC = 64 computed columns: NOUT = NC - 2 HALO outputs plus a halo of 2 columns per WN layer
// each side (the k = 5 convs eat two columns per layer; what lies outside [0, T) is exactly zero, as the unfused convs' zero padding).
// LDS holds the residual stream h [192 x (NC + 4)] and one work tile [192 x NC] (x0, the gated activations, the skip sum in turn), in
// the blocked fp16 layout of conv1d_h.hip ([C/8][col][8]: a B fragment is one ds_read_b128) — one plane in fp16 mode, a hi and a lo
// plane of 32 v in split mode (conv1d_hl.hip).  Weights are the packs of svc_pack_conv1d_h / _hl in natural row order, read from L2
// straight into registers through a two-chunk ring.  Wave w owns row tiles 3w..3w+2 of every 384-row product: waves 0, 1 hold the tanh
// half / the residual rows, waves 2, 3 the sigmoid half / the skip rows — the gate's two halves meet through an fp32 LDS scratch
// (two barriers), the skip sum lives in the registers of waves 2, 3 across all layers.
#include "common.h"
#include <algorithm>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int MAXL = SVC_COUPLING_MAX_LAYERS;
constexpr int H = 192, HB = H / 8, HALF = 96, KSIN = 5;
constexpr float ASC = 32.f, IASC = 1.f / 32.f;      // split planes hold 32 v (conv1d_hl.hip); fp16 planes hold v

struct CPL {
  float* x;
  long long x_bs, x_cs;
  const float* mask;
  const float* cond;
  long long cond_bs, cond_cs;
  int cond_ts;
  const h8* w_pre;
  const float* b_pre;
  const h8* w_in[MAXL];
  const float* b_in[MAXL];
  const h8* w_rs[MAXL];
  const float* b_rs[MAXL];
  const h8* w_post;
  const float* b_post;
  float s_pre, s_in[MAXL], s_rs[MAXL], s_post;
  int B, T, L, reverse;
  int* flag;
  int n_tiles;      // column tiles per batch item; blocks with blockIdx.x >= n_tiles (row blockIdx.y == 0 only) are L2 prefetchers
  int* sink;        // never written (keeps the prefetchers' loads alive)
};

// ---- plane policies: P = 1 one fp16 plane of v; P = 2 hi / lo planes of 32 v, range-checked (include/svc_hip.h, RANGE)
template <int P>
__device__ __forceinline__ void enc(float v, _Float16& hi, _Float16& lo, bool& bad) {
  if constexpr (P == 1) {
    hi = (_Float16)v;
    lo = (_Float16)0.f;
  } else {
    const float s = v * ASC;
    hi = (_Float16)s;
    lo = (_Float16)(s - (float)hi);
    bad |= !(fabsf(s) <= 65504.f);
  }
}
template <int P>
__device__ __forceinline__ float dec(_Float16 hi, _Float16 lo) {
  if constexpr (P == 1) return (float)hi;
  else return ((float)hi + (float)lo) * IASC;
}

// Branch-free activations (the libm tanhf is two divergent branches and an IEEE division per element: 96 elements per lane and layer).
// |error| <= ~1e-7 absolute: below the fp32 kernels' own distance from float64 on this block (tests/test_flow_fused_gpu.py).
__device__ __forceinline__ float fast_tanh(float v) {
  const float t = __expf(-2.f * fabsf(v));
  const float r = (1.f - t) * __builtin_amdgcn_rcpf(1.f + t);
  return copysignf(r, v);
}
__device__ __forceinline__ float fast_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

// acc[mt][nt] += W[rows of tile mt] x X[:, columns of tile nt] over S = (channel group, tap) steps.
//   wp: this lane's A fragment of its first row tile at step 0 (+ step * sstride + mt * 64; lo plane at + wplane8)
//   xw: this lane's B fragment base (tile + kh * XW + li; step (g, tap) at + 2 g XW + tap, column tile nt at + 32 nt, lo plane at + lplane8)
template <int P, int KS, int S, int MT, int NT, int CH>
__device__ __forceinline__ void mma_run(f32x16 (&acc)[MT][NT], const h8* __restrict__ wp, long long wplane8, long long sstride,
                                        const h8* __restrict__ xw, int lplane8, int XW) {
  static_assert(S % (3 * CH) == 0, "a multiple of three whole chunks: every load of the three-buffer ring is unconditional");
  auto wload = [&](h8 (&af)[CH][MT][P], int s0) {
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const h8* q = wp + (long long)(s0 + j) * sstride + mt * 64;
        af[j][mt][0] = q[0];
        if constexpr (P == 2) af[j][mt][1] = q[wplane8];
      }
  };
  auto chunk = [&](const h8 (&af)[CH][MT][P], int s0) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int sidx = s0 + j;
      const int g = sidx / KS, tap = sidx - g * KS;
      const h8* xr = xw + 2 * g * XW + tap;
      h8 bh[NT], bl[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        bh[nt] = xr[nt * 32];
        if constexpr (P == 2) bl[nt] = xr[lplane8 + nt * 32];
      }
      if constexpr (P == 2) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][mt][1], bh[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][mt][0], bl[nt], acc[mt][nt], 0, 0, 0);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][mt][0], bh[nt], acc[mt][nt], 0, 0, 0);
    }
  };
  // A ring of THREE chunk buffers: the loads of chunks c + 1 and c + 2 are in flight while chunk c multiplies.  The weights of a
  // coupling are cold (the decoder has streamed hundreds of MB through L2 since the last clip), a fragment takes 1-2 us to arrive, and
  // 18 workgroups cannot hide that behind each other: bytes in flight per wave are the kernel's bandwidth (two buffers: 31 GB/s per CU
  // measured, profiles/r10d_*).  Every load is issued IN FRONT of a chunk's instructions and pinned there (sched_barrier): left alone,
  // the scheduler of this 500-register kernel sinks each load to its use — `global_load -> s_waitcnt vmcnt(0) -> 2 MFMAs`, an exposed
  // round trip per fragment (the first build: 630 us per coupling).  Loads are unconditional (the last trips re-read the last chunk):
  // a conditional load makes the compiler's vmcnt bookkeeping wait for the NEWEST chunk at the join.
  h8 a0[CH][MT][P], a1[CH][MT][P], a2[CH][MT][P];
  constexpr int LASTC = S - CH;
  wload(a0, 0);
  wload(a1, CH);
#pragma unroll 1
  for (int s0 = 0; s0 < S; s0 += 3 * CH) {
    wload(a2, min(s0 + 2 * CH, LASTC));
    __builtin_amdgcn_sched_barrier(0);
    chunk(a0, s0);
    __builtin_amdgcn_sched_barrier(0);
    wload(a0, min(s0 + 3 * CH, LASTC));
    __builtin_amdgcn_sched_barrier(0);
    chunk(a1, s0 + CH);
    __builtin_amdgcn_sched_barrier(0);
    wload(a1, min(s0 + 4 * CH, LASTC));
    __builtin_amdgcn_sched_barrier(0);
    chunk(a2, s0 + 2 * CH);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int MT, int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[MT][NT]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
}

// pack geometry (svc_pack_conv1d_h: [Cin/16][taps][RP][16] halves, RP = rows rounded up to 128), in h8 units
constexpr long long sstride_of(int RP) { return (long long)RP * 2; }
constexpr long long wplane_of(int G, int KS, int RP) { return (long long)G * KS * RP * 2; }

// ---- L2 prefetchers.  One clip is 18 compute workgroups on a 256-CU chip, each streaming ALL of the coupling's weights (3.5 MB fp16, 7 MB
// split) — cold: they come from HBM / the infinity cache at 1-2 us per fragment, and a workgroup's bandwidth is its bytes in flight.
// The launch therefore carries NPF extra workgroups (idle CUs otherwise) that do nothing but read the packs, in the order the compute
// workgroups consume them, eight workgroups per XCD (workgroup -> XCD is round-robin over the linear block index on this chip — used
// for speed only: a wrong guess costs the prefetch, not the result), each touching every eighth 4 KB piece, 8 loads in flight per lane.
// The compute workgroups then find their fragments in their XCD's L2.
constexpr int NPF = 64;
template <int P>
__device__ __forceinline__ void prefetch_packs(const CPL& p, int j) {
  const int slot = j >> 3, tid = threadIdx.x;
  unsigned acc = 0;
  auto sweep = [&](const h8* base, long long n8) {      // n8: h8 words of the whole pack (all planes)
    const long long stride = 8ll * 256;
    for (long long k = (long long)slot * 256 + tid; k < n8; k += 8 * stride) {
      h8 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long idx = k + u * stride;
        v[u] = base[idx < n8 ? idx : k];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= __builtin_bit_cast(unsigned, __builtin_shufflevector(v[u], v[u], 0, 1));
    }
  };
  const int L = p.L;
  sweep(p.w_pre, P * wplane_of(HALF / 16, 1, 256));
  for (int l = 0; l < L; ++l) {
    sweep(p.w_in[l], P * wplane_of(HB / 2, KSIN, 384));
    sweep(p.w_rs[l], P * wplane_of(HB / 2, 1, l == L - 1 ? 256 : 384));
  }
  sweep(p.w_post, P * wplane_of(HB / 2, 1, 128));
  if (acc == 0x9e3779b9u && p.sink) atomicOr(p.sink, 1);      // (never both: the loads must not be optimised away)
}

template <int P, int NT>
__global__ __launch_bounds__(256, 1) void coupling_fused_kernel(CPL p) {
  if (blockIdx.x >= (unsigned)p.n_tiles) {
    if (blockIdx.y == 0) prefetch_packs<P>(p, blockIdx.x - p.n_tiles);
    return;
  }
  constexpr int NC = 32 * NT, XW = NC + 4;
  constexpr int CHI = P == 1 ? 4 : 2, CHR = P == 1 ? 4 : 2, CHP = 2;      // steps per ring chunk (60 / 12 / 6 steps = 12 or 15 / 3 or 6 / 3 chunks)
  constexpr int HPL = HB * XW;      // h8 words per plane of the residual-stream tile
  constexpr int APL = HB * NC;      // ... of the work tile
  constexpr int WORK8 = (P * HB > 48 ? P * HB : 48) * NC;      // h8 words of the work tile's region (>= the fp32 gate scratch [H][NC])
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_cf[];
  h8* hs = reinterpret_cast<h8*>(smem_cf);                 // [P][HB][XW]   column jj <-> computed column jj - 2
  h8* as = hs + P * HPL;                                   // [P][HB][NC]   (x0 [P][12][NC] / gated activations / skip sum; fp32 gate scratch [H][NC])
  float* sc = reinterpret_cast<float*>(as);
  float* outs = reinterpret_cast<float*>(as + WORK8);      // [H][NC] fp32: the skip sum `output` (waves 2, 3)
  float* lb = outs + H * NC;                               // [2][768] per-layer biases (in_layer + conditioning row, res_skip), double-buffered;
  float* pb = lb + 2 * 768;                                // [192 + 96] b_pre, b_post
  _Float16* hsh = reinterpret_cast<_Float16*>(hs);
  _Float16* ash = reinterpret_cast<_Float16*>(as);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int L = p.L, HALO = 2 * L, NOUT = NC - 2 * HALO;
  const int t0 = blockIdx.x * NOUT, b = blockIdx.y, T = p.T;
  bool bad = false;
  const bool cond_row = p.cond != nullptr && p.cond_ts == 0;      // one conditioning row per (batch item, layer): folded into the staged bias
  const bool cond_frame = p.cond != nullptr && p.cond_ts != 0;    // per-frame conditioning (speaker mix): read per element

  float insv[NT];
  int tcol[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int t = t0 - HALO + nt * 32 + li;
    tcol[nt] = t;
    insv[nt] = (t >= 0 && t < T) ? (p.mask ? p.mask[(long long)b * T + t] : 1.f) : 0.f;
  }

  // per-layer bias vectors into LDS: [0, 384) in_layer bias (+ the layer's conditioning row), [384, 768) res_skip bias
  auto stage_bias = [&](int l) {
    float* dst = lb + (l & 1) * 768;
    const float* cn = cond_row ? p.cond + (long long)b * p.cond_bs + (long long)(l * 2 * H) * p.cond_cs : nullptr;
    const int nrs = l == L - 1 ? H : 2 * H;
    for (int i = tid; i < 768; i += 256) {
      float v = 0.f;
      if (i < 384) {
        v = p.b_in[l][i];
        if (cn) v += cn[(long long)i * p.cond_cs];
      } else if (i - 384 < nrs) {
        v = p.b_rs[l][i - 384];
      }
      dst[i] = v;
    }
  };

  // ---- x0 (channels 0..95 of the view) -> work tile; zero pads of the residual-stream tile; biases
  {
    const float* xb = p.x + (long long)b * p.x_bs;
    for (int idx = tid; idx < (HALF / 8) * NC; idx += 256) {
      const int cb = idx / NC, j = idx - cb * NC;
      const int t = t0 - HALO + j;
      h8 vh = {0, 0, 0, 0, 0, 0, 0, 0}, vl = {0, 0, 0, 0, 0, 0, 0, 0};
      if (t >= 0 && t < T) {
        float xv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = xb[(long long)(cb * 8 + e) * p.x_cs + t];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          _Float16 hi, lo;
          enc<P>(xv[e], hi, lo, bad);
          vh[e] = hi;
          vl[e] = lo;
        }
      }
      as[idx] = vh;
      if constexpr (P == 2) as[APL + idx] = vl;
    }
    for (int idx = tid; idx < P * HB * 4; idx += 256) {
      const int pl = idx / (HB * 4), r = idx - pl * HB * 4;
      const int cb = r >> 2, e = r & 3;
      const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      hs[pl * HPL + cb * XW + (e < 2 ? e : XW - 4 + e)] = z;
    }
    for (int i = tid; i < H + HALF; i += 256) pb[i] = i < H ? p.b_pre[i] : p.b_post[i - H];
    stage_bias(0);
  }
  __syncthreads();

  // ---- pre: h = (W_pre x0 + b) * mask   (modules/modules.py:291); 6 row tiles: waves 0..2 take two each
  if (w < 3) {
    f32x16 acc[2][NT];
    zero_acc(acc);
    mma_run<P, 1, HALF / 16, 2, NT, CHP>(acc, p.w_pre + ((long long)(64 * w + li)) * 2 + kh, wplane_of(HALF / 16, 1, 256), sstride_of(256),
                                       as + kh * NC + li, APL, NC);
    const float s = p.s_pre * (P == 2 ? IASC : 1.f);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row8 = 64 * w + mt * 32 + 8 * i;
          const f32x4 bq = *reinterpret_cast<const f32x4*>(pb + row8 + 4 * kh);
          h4 oh, ol;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = (acc[mt][nt][4 * i + e] * s + bq[e]) * insv[nt];
            _Float16 hi, lo;
            enc<P>(v, hi, lo, bad);
            oh[e] = hi;
            ol[e] = lo;
          }
          const int o = ((row8 >> 3) * XW + nt * 32 + li + 2) * 8 + 4 * kh;
          *reinterpret_cast<h4*>(hsh + o) = oh;
          if constexpr (P == 2) *reinterpret_cast<h4*>(hsh + HPL * 8 + o) = ol;
        }
  }
  __syncthreads();

  // ---- the WN layers (modules/modules.py:110-138)
  for (int l = 0; l < L; ++l) {
    const float* lbl = lb + (l & 1) * 768;
    // in_layer: k = 5 conv 192 -> 384, + bias + g_l, tanh (rows 0..191: waves 0, 1) / sigmoid (rows 192..383: waves 2, 3)
    f32x16 acc[3][NT];
    zero_acc(acc);
    mma_run<P, KSIN, (HB / 2) * KSIN, 3, NT, CHI>(acc, p.w_in[l] + ((long long)(96 * w + li)) * 2 + kh, wplane_of(HB / 2, KSIN, 384),
                                                 sstride_of(384), hs + kh * XW + li, HPL, XW);
    {
      const float s = p.s_in[l] * (P == 2 ? IASC : 1.f);
      const float* cn = cond_frame ? p.cond + (long long)b * p.cond_bs + (long long)(l * 2 * H) * p.cond_cs : nullptr;
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int tc = min(max(tcol[nt], 0), T - 1);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row4 = 96 * w + mt * 32 + 8 * i + 4 * kh;
            const f32x4 bq = *reinterpret_cast<const f32x4*>(lbl + row4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = acc[mt][nt][4 * i + e] * s + bq[e];
              if (cn) v += cn[(long long)(row4 + e) * p.cond_cs + tc];
              acc[mt][nt][4 * i + e] = w >= 2 ? fast_sigmoid(v) : fast_tanh(v);
            }
          }
        }
      if (w >= 2) {
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              sc[(96 * (w - 2) + mt * 32 + 8 * (r >> 2) + 4 * kh + (r & 3)) * NC + nt * 32 + li] = acc[mt][nt][r];
      }
      __syncthreads();
      if (w < 2) {
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              acc[mt][nt][r] *= sc[(96 * w + mt * 32 + 8 * (r >> 2) + 4 * kh + (r & 3)) * NC + nt * 32 + li];
      }
      __syncthreads();      // the scratch is read: the gated activations may overwrite it
      if (w < 2) {
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              h4 oh, ol;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                _Float16 hi, lo;
                enc<P>(acc[mt][nt][4 * i + e], hi, lo, bad);
                oh[e] = hi;
                ol[e] = lo;
              }
              const int o = ((12 * w + 4 * mt + i) * NC + nt * 32 + li) * 8 + 4 * kh;
              *reinterpret_cast<h4*>(ash + o) = oh;
              if constexpr (P == 2) *reinterpret_cast<h4*>(ash + APL * 8 + o) = ol;
            }
      }
      __syncthreads();
    }
    // res_skip_layer: 1x1 192 -> 384 (last layer: 192 -> 192, all skip).  Waves 0, 1: residual rows -> h = (h + res) * mask;
    // waves 2, 3: skip rows -> output += skip
    const bool last = l == L - 1;
    if (!last || w >= 2) {
      const int RP = last ? 256 : 384;
      const int rt0 = last ? 3 * (w - 2) : 3 * w;
      const float s = p.s_rs[l] * (P == 2 ? IASC : 1.f);
      f32x16 acc2[3][NT];
      zero_acc(acc2);
      mma_run<P, 1, HB / 2, 3, NT, CHR>(acc2, p.w_rs[l] + ((long long)(32 * rt0 + li)) * 2 + kh, wplane_of(HB / 2, 1, RP), sstride_of(RP),
                                        as + kh * NC + li, APL, NC);
      const float* br = lbl + 384 + 32 * rt0 + 4 * kh;      // this wave's rows of the res_skip bias
      if (w < 2) {
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int o = ((12 * w + 4 * mt + i) * XW + nt * 32 + li + 2) * 8 + 4 * kh;
              const h4 qh = *reinterpret_cast<const h4*>(hsh + o);
              h4 ql = qh;
              if constexpr (P == 2) ql = *reinterpret_cast<const h4*>(hsh + HPL * 8 + o);
              const f32x4 bq = *reinterpret_cast<const f32x4*>(br + mt * 32 + 8 * i);
              h4 oh, ol;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float v = (dec<P>(qh[e], ql[e]) + acc2[mt][nt][4 * i + e] * s + bq[e]) * insv[nt];
                _Float16 hi, lo;
                enc<P>(v, hi, lo, bad);
                oh[e] = hi;
                ol[e] = lo;
              }
              *reinterpret_cast<h4*>(hsh + o) = oh;
              if constexpr (P == 2) *reinterpret_cast<h4*>(hsh + HPL * 8 + o) = ol;
            }
      } else {
        // the skip sum `output` accumulates in an fp32 LDS tile [192][NC] (each element owned by one lane: no race)
        float* oq = outs + (96 * (w - 2) + 4 * kh) * NC + li;
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const f32x4 bq = *reinterpret_cast<const f32x4*>(br + mt * 32 + 8 * i);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float* q = oq + (mt * 32 + 8 * i + e) * NC + nt * 32;
                const float v = acc2[mt][nt][4 * i + e] * s + bq[e];
                *q = l == 0 ? v : *q + v;
              }
            }
      }
    }
    if (!last) stage_bias(l + 1);      // into the other buffer: nobody reads it before the barrier below
    __syncthreads();
  }

  // ---- output * mask -> work tile; post 1x1 192 -> 96; x1 update in place (modules/modules.py:297-306, mean_only: logs = 0)
  if (w >= 2) {
    const float* oq = outs + (96 * (w - 2) + 4 * kh) * NC + li;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          h4 oh, ol;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            _Float16 hi, lo;
            enc<P>(oq[(mt * 32 + 8 * i + e) * NC + nt * 32] * insv[nt], hi, lo, bad);
            oh[e] = hi;
            ol[e] = lo;
          }
          const int o = ((12 * (w - 2) + 4 * mt + i) * NC + nt * 32 + li) * 8 + 4 * kh;
          *reinterpret_cast<h4*>(ash + o) = oh;
          if constexpr (P == 2) *reinterpret_cast<h4*>(ash + APL * 8 + o) = ol;
        }
  }
  __syncthreads();
  if (w < 3) {
    f32x16 acc3[1][NT];
    zero_acc(acc3);
    mma_run<P, 1, HB / 2, 1, NT, CHR>(acc3, p.w_post + ((long long)(32 * w + li)) * 2 + kh, wplane_of(HB / 2, 1, 128), sstride_of(128),
                                     as + kh * NC + li, APL, NC);
    const float s = p.s_post * (P == 2 ? IASC : 1.f);
    float* xb = p.x + (long long)b * p.x_bs + (long long)(HALF + 32 * w + 4 * kh) * p.x_cs;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int j = nt * 32 + li, t = tcol[nt];
      if (j >= HALO && j < NC - HALO && t < T) {
        const float mv = insv[nt];
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = xb[(long long)(8 * (r >> 2) + (r & 3)) * p.x_cs + t];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float m = (acc3[0][nt][r] * s + pb[H + 32 * w + 8 * (r >> 2) + 4 * kh + (r & 3)]) * mv;
          xb[(long long)(8 * (r >> 2) + (r & 3)) * p.x_cs + t] = p.reverse ? (old[r] - m) * mv : m + old[r] * mv;
        }
      }
    }
  }
  if (bad && p.flag) atomicOr(p.flag, 1);
}

int g_cf_prefetch = 1;      // svc_debug_set_coupling_fused bit 0: 0 = no L2 prefetch workgroups (A/B)
int g_cf_nt = 0;            // ... bits 4-5: 0 = by rule, 1 / 2 = 32 / 64 computed columns per workgroup

template <int P, int NT>
int launch_coupling(const CPL& p, hipStream_t s) {
  constexpr int NC = 32 * NT, XW = NC + 4;
  const int NOUT = NC - 4 * p.L;
  const size_t lds = ((size_t)P * HB * XW + (size_t)std::max(P * HB, 48) * NC) * 16 + ((size_t)H * NC + 2 * 768 + H + HALF) * 4;
  auto kern = coupling_fused_kernel<P, NT>;
  static bool done = false;
  if (!done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done = true;
  }
  CPL q = p;
  q.n_tiles = svc::cdiv(p.T, NOUT);
  q.sink = nullptr;
  // prefetchers only where the compute workgroups are too few to hide the weight latency behind each other
  const int npf = (g_cf_prefetch && (long long)q.n_tiles * p.B < 128) ? NPF : 0;
  hipLaunchKernelGGL(kern, dim3(q.n_tiles + npf, p.B), dim3(256), lds, s, q);
  return svc::check_launch("coupling_fused");
}

}  // namespace

namespace svc {
int* hl_range_flag_ptr();      // conv1d_hl.hip: the calling thread's registered flag word (svc_hl_range_flag)
}

extern "C" int svc_debug_set_coupling_fused(int cfg) {
  g_cf_prefetch = cfg & 1;
  g_cf_nt = (cfg >> 4) & 3;
  return SVC_OK;
}

extern "C" int svc_coupling_fused_h(const svc_coupling_args* ap, void* stream) {
  SVC_REQUIRE(ap != nullptr, "coupling_fused_h: null args");
  const svc_coupling_args& a = *ap;
  SVC_REQUIRE(a.x && a.w_pre && a.b_pre && a.w_post && a.b_post, "coupling_fused_h: null tensor");
  SVC_REQUIRE(a.channels == 2 * HALF && a.hidden == H && a.kernel_size == KSIN,
              "coupling_fused_h: built for channels 192, hidden 192, kernel size 5 (got %d, %d, %d)", a.channels, a.hidden, a.kernel_size);
  SVC_REQUIRE(a.n_layers >= 1 && a.n_layers <= 6 && a.n_layers <= MAXL, "coupling_fused_h: 1..6 WN layers (got %d)", a.n_layers);
  SVC_REQUIRE(a.planes == 1 || a.planes == 2, "coupling_fused_h: planes must be 1 (fp16) or 2 (split)");
  SVC_REQUIRE(a.B > 0 && a.T > 0, "coupling_fused_h: empty shape");
  CPL p;
  p.x = a.x; p.x_bs = a.x_bs; p.x_cs = a.x_cs;
  p.mask = a.mask;
  p.cond = a.cond; p.cond_bs = a.cond_bs; p.cond_cs = a.cond_cs; p.cond_ts = a.cond_ts;
  p.w_pre = reinterpret_cast<const h8*>(a.w_pre); p.b_pre = a.b_pre;
  p.w_post = reinterpret_cast<const h8*>(a.w_post); p.b_post = a.b_post;
  p.s_pre = a.s_pre != 0.f ? a.s_pre : 1.f;
  p.s_post = a.s_post != 0.f ? a.s_post : 1.f;
  for (int l = 0; l < a.n_layers; ++l) {
    SVC_REQUIRE(a.w_in[l] && a.b_in[l] && a.w_rs[l] && a.b_rs[l], "coupling_fused_h: null weight of layer %d", l);
    p.w_in[l] = reinterpret_cast<const h8*>(a.w_in[l]); p.b_in[l] = a.b_in[l];
    p.w_rs[l] = reinterpret_cast<const h8*>(a.w_rs[l]); p.b_rs[l] = a.b_rs[l];
    p.s_in[l] = a.s_in[l] != 0.f ? a.s_in[l] : 1.f;
    p.s_rs[l] = a.s_rs[l] != 0.f ? a.s_rs[l] : 1.f;
  }
  p.B = a.B; p.T = a.T; p.L = a.n_layers; p.reverse = a.reverse;
  p.flag = a.planes == 2 ? svc::hl_range_flag_ptr() : nullptr;
  hipStream_t s = (hipStream_t)stream;
  // FLOPs of the coupling as the reference computes it (pre + L x (k5 conv + res/skip) + post), columns once (the halo's recompute is ours)
  const double per_col = 2.0 * (HALF * H + a.n_layers * (double)(2 * H) * H * KSIN + (a.n_layers - 1) * (double)(2 * H) * H + (double)H * H + H * HALF);
  svc::ProfScope prof(s, a.planes == 2 ? "coupling_fused_hl" : "coupling_fused_h", per_col * a.B * a.T, 8.0 * a.B * (2.0 * HALF) * a.T);
  // 32 computed columns (16 outputs) per workgroup while that still leaves CUs idle: one clip is 54 workgroups instead of 18, each with
  // half the matrix and activation work behind the same weight stream (311 against 452 us per flow in fp16, 478 against 735 split:
  // profiles/r10g_flow_bench.txt); 64 columns (48 outputs: a third of the halo recompute and of the weight traffic) for batches
  const long long wg64 = (long long)svc::cdiv(a.T, 64 - 4 * a.n_layers) * a.B;
  const int nt = g_cf_nt ? g_cf_nt : (wg64 < 160 ? 1 : 2);
  if (a.planes == 2) return nt == 1 ? launch_coupling<2, 1>(p, s) : launch_coupling<2, 2>(p, s);
  return nt == 1 ? launch_coupling<1, 1>(p, s) : launch_coupling<1, 2>(p, s);
}
