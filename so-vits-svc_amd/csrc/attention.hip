// attention.hip — multi-head self-attention with windowed relative-position keys/values on the fp32 matrix pipe.
// Reference: MultiHeadAttention.attention, modules/attentions.py:207-239 (+ the relative-position helpers
// :259-303 which pad emb_rel_k/v to 2T-1 rows and skew; restated here in the banded form
//   scores[i,j] += (q_i/sqrt(d)) . E_k[j-i+w]  and  out_i += sum_{|r|<=w} p[i,i+r] * E_v[r+w],
// verified against the reference module in tests/golden).  Masked scores are set to -1e4 (:231), not -inf.
//
// Layout: q,k,v,out are [B, heads*dk, T] channel-major views (time contiguous) — exactly what the 1x1 conv
// projections produce, so no transposes exist anywhere.  One workgroup = (32 queries, one head, one batch
// item), 8 waves splitting the key tiles.  We compute S^T = K Q^T (M = keys, N = queries) so that in the MFMA
// C layout every lane owns ONE query column: softmax statistics are a 16-register reduction plus one
// lane^32 exchange, and P^T is already the B operand of the second MFMA  O^T = V^T P^T  (the k index
// order inside the tile is permuted to match the C layout; a sum does not care).  T x T scores are never
// materialised.  One pass over the keys with an online softmax: every wave keeps a running (max, sum) per query
// column and rescales its O accumulators when the max grows; the 8 waves of a workgroup split the key tiles and are
// combined at the end (M = max m_w, L = sum l_w e^{m_w-M}); the 2w+1 band probabilities the relative-VALUE term needs
// are recomputed from q.k once M and L are known (9 dot products per query).
#include "common.h"
#include <algorithm>
#include <cstdlib>

namespace {

constexpr int MAXREL = 17;   // 2*window+1 <= 17

struct AttnP {
  svc_attention_args a;
  int nJ;
  float inv_unused;
};

__device__ __forceinline__ int crow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// NDT: dk = 32 * NDT.  NW: waves per workgroup — they split the key tiles.  A 10 s utterance has 27 key tiles and only
// 27 x heads query tiles: with 8 waves a workgroup walks 4 key tiles per wave on 54 of 256 CUs (88 us per layer); 16 waves
// halve the walk (2 key tiles per wave, 4 waves per SIMD hiding each other's load latency).  To fit 16 waves in LDS the V
// operand is staged 32 channel rows at a time (4.2 KB per wave instead of 12.7) and the per-wave output tiles are folded
// 16 -> 8 before the final combine.
template <int NDT, int NW>
__global__ __launch_bounds__(NW * 64) void attention_kernel(AttnP p) {
  constexpr int DK = 32 * NDT;
  constexpr int VP = 33;  // V tile pitch
  constexpr int VROWS = NW > 8 ? 32 : DK;      // channel rows of V staged per wave at a time
  constexpr int NWC = NW > 8 ? NW / 2 : NW;    // output tiles that meet in LDS for the final combine
  const svc_attention_args& a = p.a;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // carve
  constexpr int SLAB = (NW * VROWS * VP > NWC * DK * 32) ? NW * VROWS * VP : NWC * DK * 32;
  float* Vl = lds;                                  // [NW][VROWS][VP]   (aliased by Ol [NWC][DK][32] after the loop)
  float* Rk = Vl + SLAB;                            // [MAXREL][32]  relative-key logits of this query tile
  float* Pl = Rk + MAXREL * 32;                     // [MAXREL][32]  normalised band probabilities
  float* Ml = Pl + MAXREL * 32;                     // [NW][32]
  float* Ll = Ml + NW * 32;                         // [NW][32]
  float* Ql = Ll + NW * 32;                         // NW > 8: [DK/2][64] the Q fragment, shared by all waves (lane-linear)

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int half = lane >> 5, li = lane & 31;
  const int i0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
  const int T = a.T;
  const int i = i0 + li;
  const int nrel = a.window > 0 ? 2 * a.window + 1 : 0;
  const float sqrtk = sqrtf((float)DK);

  const float* qb = a.q + (long long)b * a.q_bs + (long long)h * DK * a.q_cs;
  const float* kb = a.k + (long long)b * a.k_bs + (long long)h * DK * a.k_cs;
  const float* vb = a.v + (long long)b * a.v_bs + (long long)h * DK * a.v_cs;
  const float* mq = a.mask ? a.mask + (long long)b * a.mask_bs : nullptr;

  // Q fragment (B operand of S^T = K Q^T): lane (half, li) holds q[d = 2s + half][i] / sqrt(dk).  It is the same for every
  // wave of the workgroup: the 16-wave variant (128 VGPRs per wave) keeps it in LDS instead of DK/2 registers per lane.
  float qreg[NW > 8 ? 1 : DK / 2];
  if constexpr (NW > 8) {
    for (int idx = tid; idx < (DK / 2) * 64; idx += NW * 64) {
      const int s = idx >> 6, l = idx & 63;
      const int d = 2 * s + (l >> 5), iq = i0 + (l & 31);
      Ql[idx] = iq < T ? qb[(long long)d * a.q_cs + iq] / sqrtk : 0.f;
    }
  } else {
#pragma unroll
    for (int s = 0; s < DK / 2; ++s) {
      const int d = 2 * s + half;
      qreg[s] = i < T ? qb[(long long)d * a.q_cs + i] / sqrtk : 0.f;
    }
  }
  // relative-key logits for this query tile: Rk[m][ii] = sum_d q[d][i0+ii]/sqrt(dk) * E_k[m][d]
  for (int idx = tid; idx < nrel * 32; idx += NW * 64) {
    const int m = idx >> 5, ii = idx & 31;
    float acc = 0.f;
    if (i0 + ii < T)
      for (int d = 0; d < DK; ++d) acc = fmaf(qb[(long long)d * a.q_cs + i0 + ii] / sqrtk, a.emb_rel_k[m * DK + d], acc);
    Rk[idx] = acc;
  }
  __syncthreads();

  const float mi = mq ? (i < T ? mq[i] : 0.f) : 1.f;
  const int n_iter = (p.nJ + NW - 1) / NW;

  // ---------------- one pass over this wave's key tiles: online softmax (running max / sum per query column, which
  // is one LANE in the MFMA C layout), O^T += V^T P^T with P = exp(s - m_run) ----------------
  float mrun = -INFINITY, lrun = 0.f;
  f32x16 oacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
  float* Vw = Vl + w * VROWS * VP;
  for (int it = 0; it < n_iter; ++it) {
    const int jt = it * NW + w;
    const int j0 = jt * 32;
    if (jt >= p.nJ) continue;
    if (a.mask_mode == 2 && j0 > i0 + 31) continue;  // fully-masked causal tile contributes exp(-1e4 - max) == 0
    // ---- scores S^T tile (keys x queries) ----
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int jl = j0 + li;
    const bool jok = jl < T;
    const int jc = jok ? jl : T - 1;
    const int koff = half * (int)a.k_cs + jc, voff = half * (int)a.v_cs + jc;
    // 16 K rows at a time (a scheduling barrier between groups keeps at most 16 loads in flight: the fully hoisted
    // form needs DK/2 more live registers than the 256 available at two waves per SIMD)
#pragma unroll
    for (int c = 0; c < DK / 32; ++c) {
      float kv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) kv[u] = (kb + (long long)(2 * (c * 16 + u)) * a.k_cs)[koff];   // uniform row + lane offset
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float qv = NW > 8 ? Ql[(c * 16 + u) * 64 + lane] : qreg[NW > 8 ? 0 : c * 16 + u];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(jok ? kv[u] : 0.f, qv, acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float sc[16];
    float tm = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + crow(r, half);
      float s = acc[r];
      const int rel = j - i + a.window;
      if (nrel && rel >= 0 && rel < nrel) s += Rk[rel * 32 + li];
      bool masked = false;
      if (a.mask_mode == 1) masked = (mi * (j < T ? mq[j] : 0.f)) == 0.f;
      else if (a.mask_mode == 2) masked = j > i;
      if (masked) s = -1e4f;
      if (j >= T) s = -INFINITY;
      sc[r] = s;
      tm = fmaxf(tm, s);
    }
    tm = fmaxf(tm, __shfl_xor(tm, 32));
    const float mnew = fmaxf(mrun, tm);          // finite: every tile has at least one key < T
    const float resc = __expf(mrun - mnew);      // exp(-inf) = 0 on the first tile
    float pr[16];
    float ts = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pr[r] = __expf(sc[r] - mnew);
      ts += pr[r];
    }
    ts += __shfl_xor(ts, 32);
    lrun = lrun * resc + ts;
    mrun = mnew;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[dt][r] *= resc;
    // ---- stage V (pitch 33: the A-operand fetch below has its 32 lanes on 32 different d rows) and O^T += V^T P^T ----
    if constexpr (NW > 8) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        float vv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) vv[u] = (vb + (long long)(dt * 32 + 2 * u) * a.v_cs)[voff];
#pragma unroll
        for (int u = 0; u < 16; ++u) Vw[(2 * u + half) * VP + li] = jok ? vv[u] : 0.f;
        // the slab is private to this wave: LDS ops of one wave complete in order, no workgroup barrier needed
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float av = Vw[li * VP + crow(r, half)];
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, pr[r], oacc[dt], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    } else {
#pragma unroll
      for (int c = 0; c < DK / 32; ++c) {
        float vv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) vv[u] = (vb + (long long)(2 * (c * 16 + u)) * a.v_cs)[voff];
#pragma unroll
        for (int u = 0; u < 16; ++u) Vw[(2 * (c * 16 + u) + half) * VP + li] = jok ? vv[u] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float av = Vw[(dt * 32 + li) * VP + crow(r, half)];
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, pr[r], oacc[dt], 0, 0, 0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }

  // ---------------- combine the waves: M = max m_w, L = sum l_w e^{m_w - M}, O = sum O_w e^{m_w - M} / L ----------------
  if (half == 0) {
    Ml[w * 32 + li] = mrun;
    Ll[w * 32 + li] = lrun;
  }
  __syncthreads();
  float M = -INFINITY, Lsum = 0.f;
#pragma unroll
  for (int ww = 0; ww < NW; ++ww) M = fmaxf(M, Ml[ww * 32 + li]);
#pragma unroll
  for (int ww = 0; ww < NW; ++ww) {
    const float mw = Ml[ww * 32 + li];
    if (mw > -INFINITY) Lsum += Ll[ww * 32 + li] * __expf(mw - M);
  }
  const float wsc = mrun > -INFINITY ? __expf(mrun - M) / Lsum : 0.f;
  // relative-value band: p[i, i+r] for |r| <= window recomputed from q.k (9 dot products per query; needs M and L)
  for (int idx = tid; idx < nrel * 32; idx += NW * 64) {
    const int m = idx >> 5, ii = idx & 31;
    const int iq = i0 + ii, j = iq + m - a.window;
    float pv = 0.f;
    if (iq < T && j >= 0 && j < T) {
      float s = 0.f;
      for (int d = 0; d < DK; ++d)
        s = fmaf(kb[(long long)d * a.k_cs + j], qb[(long long)d * a.q_cs + iq] / sqrtk, s);
      s += Rk[idx];
      bool masked = false;
      if (a.mask_mode == 1) masked = (mq[iq] * mq[j]) == 0.f;
      else if (a.mask_mode == 2) masked = j > iq;
      if (masked) s = -1e4f;
      // M, L of query column ii live in lanes li == ii: read them back through LDS-free recomputation
      float Mq = -INFINITY, Lq = 0.f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) Mq = fmaxf(Mq, Ml[ww * 32 + ii]);
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) {
        const float mw = Ml[ww * 32 + ii];
        if (mw > -INFINITY) Lq += Ll[ww * 32 + ii] * __expf(mw - Mq);
      }
      pv = __expf(s - Mq) / Lq;
    }
    Pl[idx] = pv;
  }
  __syncthreads();   // every wave is past its V slab; Pl complete
  float* Ol = Vl;  // [NWC][DK][32]
  if constexpr (NW > 8) {
    // fold 16 -> 8: the upper half of the waves parks its (already weighted) tile, its partner adds it in registers
    if (w >= NWC) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) Ol[((w - NWC) * DK + dt * 32 + crow(r, half)) * 32 + li] = oacc[dt][r] * wsc;
    }
    __syncthreads();
    if (w < NWC) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* q = Ol + (w * DK + dt * 32 + crow(r, half)) * 32 + li;
          *q = oacc[dt][r] * wsc + *q;
        }
    }
  } else {
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) Ol[(w * DK + dt * 32 + crow(r, half)) * 32 + li] = oacc[dt][r] * wsc;
  }
  __syncthreads();
  float* ob = a.out + (long long)b * a.o_bs + (long long)h * DK * a.o_cs;
  for (int idx = tid; idx < DK * 32; idx += NW * 64) {
    const int d = idx >> 5, ii = idx & 31;
    if (i0 + ii >= T) continue;
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < NWC; ++ww) v += Ol[(ww * DK + d) * 32 + ii];
    for (int m = 0; m < nrel; ++m) v = fmaf(Pl[m * 32 + ii], a.emb_rel_v[m * DK + d], v);
    ob[(long long)d * a.o_cs + i0 + ii] = v;
  }
}

// ---- short sequences, few heads (one utterance: T = 862 -> 27 query tiles x 2 heads): the KEYS are split over workgroups too.
// The one-workgroup-per-query-tile form above puts all the matrix work of a layer on 54 CUs x 4 SIMDs (4 waves each walk two key
// tiles on ONE SIMD's pipe: >= 22 us of MFMA issue per SIMD while 800 SIMDs idle; measured 55-70 us per layer).  Here a workgroup
// is (query tile, head, key split) with 4 waves = one per SIMD, the split count chosen so that every wave owns ONE key tile
// (ceil(nJ / 4) splits: 7 x 54 = 378 workgroups at T = 862): per wave 48 + 48 MFMAs, K and V fragments loaded straight into
// registers at kernel entry (one wave per SIMD: the 512-register budget holds both), before the Q staging they do not depend on.
// Each workgroup leaves (O^T weighted and summed over its waves, running max, running sum, raw band scores) in the workspace;
// attention_combine_kernel merges the splits, adds the relative-value term and writes the output.  The band scores (2w+1 per
// query) are the main loop's own score-tile entries, kept as they pass — no recomputed dot products.
constexpr int SPLIT_NW = 4;
constexpr int MAX_KSPLIT = 8;
__host__ __device__ constexpr int part_floats(int dk) { return dk * 32 + 64 + MAXREL * 32; }

template <int NDT>
__global__ __launch_bounds__(SPLIT_NW * 64, 2) void attention_split_kernel(AttnP p, float* __restrict__ ws, int ksplit) {
  constexpr int DK = 32 * NDT, NW = SPLIT_NW, VP = 33;
  const svc_attention_args& a = p.a;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Vl = lds;                                  // [NW][DK][VP], aliased by Ol [NW][DK][32] after the loop
  float* Rk = Vl + NW * DK * VP;                    // [MAXREL][32] relative-key logits of this query tile
  float* Sb = Rk + MAXREL * 32;                     // [MAXREL][32] raw (masked) band scores met by this workgroup, else -inf
  float* Ml = Sb + MAXREL * 32;                     // [NW][32]
  float* Ll = Ml + NW * 32;                         // [NW][32]
  float* Ql = Ll + NW * 32;                         // [DK/2][64] Q fragment (lane-linear), scaled by 1/sqrt(dk)
  float* Ek = Ql + (DK / 2) * 64;                   // [MAXREL][DK]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int half = lane >> 5, li = lane & 31;
  const int i0 = blockIdx.x * 32, h = blockIdx.y;
  const int b = blockIdx.z / ksplit, sp = blockIdx.z - b * ksplit;
  const int T = a.T;
  const int i = i0 + li;
  const int nrel = a.window > 0 ? 2 * a.window + 1 : 0;
  const float sqrtk = sqrtf((float)DK);

  const float* qb = a.q + (long long)b * a.q_bs + (long long)h * DK * a.q_cs;
  const float* kb = a.k + (long long)b * a.k_bs + (long long)h * DK * a.k_cs;
  const float* vb = a.v + (long long)b * a.v_bs + (long long)h * DK * a.v_cs;
  const float* mq = a.mask ? a.mask + (long long)b * a.mask_bs : nullptr;

  // this wave's key tile(s): global wave gw of ksplit * NW walks tiles gw, gw + ksplit*NW, ...
  const int gw = sp * NW + w, gstride = ksplit * NW;
  // ---- K and V fragments of the FIRST tile go into registers now: they do not depend on anything staged below
  float kv[DK / 2], vv[DK / 2];
  {
    const int j0 = gw * 32, jl = j0 + li;
    const int jc = jl < T ? jl : T - 1;
    const int koff = half * (int)a.k_cs + jc, voff = half * (int)a.v_cs + jc;
    const bool live = gw < p.nJ;
#pragma unroll
    for (int u = 0; u < DK / 2; ++u) kv[u] = live ? (kb + (long long)(2 * u) * a.k_cs)[koff] : 0.f;
#pragma unroll
    for (int u = 0; u < DK / 2; ++u) vv[u] = live ? (vb + (long long)(2 * u) * a.v_cs)[voff] : 0.f;
  }
  for (int idx = tid; idx < (DK / 2) * 64; idx += NW * 64) {
    const int s = idx >> 6, l = idx & 63;
    const int d = 2 * s + (l >> 5), iq = i0 + (l & 31);
    Ql[idx] = iq < T ? qb[(long long)d * a.q_cs + iq] / sqrtk : 0.f;
  }
  for (int idx = tid; idx < nrel * DK; idx += NW * 64) Ek[idx] = a.emb_rel_k[idx];
  for (int idx = tid; idx < MAXREL * 32; idx += NW * 64) Sb[idx] = -INFINITY;
  __syncthreads();
  // relative-key logits: Rk[m][ii] = sum_d q[d][i0+ii]/sqrt(dk) * E_k[m][d], operands from LDS
  for (int idx = tid; idx < nrel * 32; idx += NW * 64) {
    const int m = idx >> 5, ii = idx & 31;
    float acc = 0.f;
#pragma unroll 8
    for (int d = 0; d < DK; ++d) acc = fmaf(Ql[(d >> 1) * 64 + (d & 1) * 32 + ii], Ek[m * DK + d], acc);
    Rk[idx] = acc;
  }
  float qreg[DK / 2];
#pragma unroll
  for (int u = 0; u < DK / 2; ++u) qreg[u] = Ql[u * 64 + lane];
  __syncthreads();

  const float mi = mq ? (i < T ? mq[i] : 0.f) : 1.f;
  float mrun = -INFINITY, lrun = 0.f;
  f32x16 oacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
  float* Vw = Vl + w * DK * VP;
  for (int jt = gw; jt < p.nJ; jt += gstride) {
    const int j0 = jt * 32;
    const int jl = j0 + li;
    const bool jok = jl < T;
    if (jt != gw) {          // (only when ksplit was capped: more than one tile per wave)
      const int jc = jok ? jl : T - 1;
      const int koff = half * (int)a.k_cs + jc, voff = half * (int)a.v_cs + jc;
#pragma unroll
      for (int u = 0; u < DK / 2; ++u) kv[u] = (kb + (long long)(2 * u) * a.k_cs)[koff];
#pragma unroll
      for (int u = 0; u < DK / 2; ++u) vv[u] = (vb + (long long)(2 * u) * a.v_cs)[voff];
    }
    if (a.mask_mode == 2 && j0 > i0 + 31) continue;  // fully-masked causal tile contributes exp(-1e4 - max) == 0
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < DK / 2; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(jok ? kv[u] : 0.f, qreg[u], acc, 0, 0, 0);
    // V to this wave's slab while the score MFMAs drain (pitch 33: the A-operand fetch has its 32 lanes on 32 different d rows)
#pragma unroll
    for (int u = 0; u < DK / 2; ++u) Vw[(2 * u + half) * VP + li] = jok ? vv[u] : 0.f;
    float sc[16];
    float tm = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + crow(r, half);
      float s = acc[r];
      const int rel = j - i + a.window;
      const bool inband = nrel && rel >= 0 && rel < nrel;
      if (inband) s += Rk[rel * 32 + li];
      bool masked = false;
      if (a.mask_mode == 1) masked = (mi * (j < T ? mq[j] : 0.f)) == 0.f;
      else if (a.mask_mode == 2) masked = j > i;
      if (masked) s = -1e4f;
      if (j >= T) s = -INFINITY;
      if (inband && i < T) Sb[rel * 32 + li] = s;     // (i, j) is met by exactly one lane of one workgroup
      sc[r] = s;
      tm = fmaxf(tm, s);
    }
    tm = fmaxf(tm, __shfl_xor(tm, 32));
    const float mnew = fmaxf(mrun, tm);          // finite: every tile has at least one key < T
    const float resc = __expf(mrun - mnew);      // exp(-inf) = 0 on the first tile
    float pr[16];
    float ts = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pr[r] = __expf(sc[r] - mnew);
      ts += pr[r];
    }
    ts += __shfl_xor(ts, 32);
    lrun = lrun * resc + ts;
    mrun = mnew;
    if (jt != gw) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] *= resc;
    }
    // the slab is private to this wave: LDS ops of one wave complete in order, no workgroup barrier needed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float av = Vw[(dt * 32 + li) * VP + crow(r, half)];
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, pr[r], oacc[dt], 0, 0, 0);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  // ---- combine the 4 waves: M = max m_w, L = sum l_w e^{m_w - M}, O = sum O_w e^{m_w - M} (NOT divided by L: the merge does) ----
  if (half == 0) {
    Ml[w * 32 + li] = mrun;
    Ll[w * 32 + li] = lrun;
  }
  __syncthreads();    // every wave is past its V slab; statistics and band scores visible
  float M = -INFINITY;
#pragma unroll
  for (int ww = 0; ww < NW; ++ww) M = fmaxf(M, Ml[ww * 32 + li]);
  const float wsc = mrun > -INFINITY ? __expf(mrun - M) : 0.f;
  float* Ol = Vl;     // [NW][DK][32]
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) Ol[(w * DK + dt * 32 + crow(r, half)) * 32 + li] = oacc[dt][r] * wsc;
  __syncthreads();
  float* part = ws + ((((long long)b * a.H + h) * gridDim.x + blockIdx.x) * ksplit + sp) * part_floats(DK);
  for (int idx = tid; idx < DK * 32; idx += NW * 64) {
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) v += Ol[ww * DK * 32 + idx];
    part[idx] = v;
  }
  if (tid < 32) {
    float Mq = -INFINITY, Lq = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) Mq = fmaxf(Mq, Ml[ww * 32 + tid]);
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) {
      const float mw = Ml[ww * 32 + tid];
      if (mw > -INFINITY) Lq += Ll[ww * 32 + tid] * __expf(mw - Mq);
    }
    part[DK * 32 + tid] = Mq;
    part[DK * 32 + 32 + tid] = Lq;
  }
  for (int idx = tid; idx < MAXREL * 32; idx += NW * 64) part[DK * 32 + 64 + idx] = Sb[idx];
}

// Merge of the key splits of one (query tile, head, batch item): out[d][i] = sum_s O_s[d][i] e^{m_s - M} / L
//   + sum_{|r| <= w} p[i, i + r] E_v[r + w][d],  p = exp(band score - M) / L  (modules/attentions.py:236-239).
template <int NDT>
__global__ __launch_bounds__(256) void attention_combine_kernel(AttnP p, const float* __restrict__ ws, int ksplit) {
  constexpr int DK = 32 * NDT;
  constexpr int TAIL = 64 + MAXREL * 32;  // floats behind a partial's O tile: m[32], l[32], band scores
  const svc_attention_args& a = p.a;
  __shared__ float St[MAX_KSPLIT][TAIL];   // the splits' statistics, staged by all threads (coalesced) before 32 of them merge
  __shared__ float Sc[MAX_KSPLIT][32];     // e^{m_s - M} / L
  __shared__ float Pl[MAXREL][32];
  __shared__ float Ev[MAXREL * DK];
  const int tid = threadIdx.x;
  const int i0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
  const int nrel = a.window > 0 ? 2 * a.window + 1 : 0;
  const float* base = ws + ((((long long)b * a.H + h) * gridDim.x + blockIdx.x) * ksplit) * part_floats(DK);
  const int tail = 64 + nrel * 32;
  for (int idx = tid; idx < ksplit * tail; idx += 256) {
    const int s = idx / tail, e = idx - s * tail;
    St[s][e] = base[(long long)s * part_floats(DK) + DK * 32 + e];
  }
  for (int idx = tid; idx < nrel * DK; idx += 256) Ev[idx] = a.emb_rel_v[idx];
  // the O tiles of this thread's output elements: issued before the statistics are merged (DK * 32 / 256 elements x ksplit loads)
  constexpr int PER = DK * 32 / 256;
  float ov[PER][MAX_KSPLIT];
#pragma unroll
  for (int e = 0; e < PER; ++e)
#pragma unroll
    for (int s = 0; s < MAX_KSPLIT; ++s) ov[e][s] = s < ksplit ? base[(long long)s * part_floats(DK) + tid + e * 256] : 0.f;
  __syncthreads();
  if (tid < 32) {
    float M = -INFINITY, L = 0.f;
    for (int s = 0; s < ksplit; ++s) M = fmaxf(M, St[s][tid]);
    for (int s = 0; s < ksplit; ++s) {
      const float ms = St[s][tid];
      const float e = ms > -INFINITY ? __expf(ms - M) : 0.f;
      Sc[s][tid] = e;
      L += St[s][32 + tid] * e;
    }
    const float rl = L > 0.f ? 1.f / L : 0.f;      // (columns beyond T carry no statistics and are never written)
    for (int s = 0; s < ksplit; ++s) Sc[s][tid] *= rl;
    for (int m = 0; m < nrel; ++m) {
      float sb = -INFINITY;
      for (int s = 0; s < ksplit; ++s) sb = fmaxf(sb, St[s][64 + m * 32 + tid]);
      Pl[m][tid] = sb > -INFINITY ? __expf(sb - M) * rl : 0.f;
    }
  }
  __syncthreads();
  float* ob = a.out + (long long)b * a.o_bs + (long long)h * DK * a.o_cs;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int idx = tid + e * 256;
    const int d = idx >> 5, ii = idx & 31;
    if (i0 + ii >= a.T) continue;
    float v = 0.f;
#pragma unroll
    for (int s = 0; s < MAX_KSPLIT; ++s) v = fmaf(ov[e][s], s < ksplit ? Sc[s][ii] : 0.f, v);
    for (int m = 0; m < nrel; ++m) v = fmaf(Pl[m][ii], Ev[m * DK + d], v);
    ob[(long long)d * a.o_cs + i0 + ii] = v;
  }
}

// key splits of the short-sequence form: 0 = not taken (enough query-tile workgroups to fill the chip, no workspace, or disabled)
int g_attn_split = -1;   // svc_debug_set_attention_waves(100 + v) / environment SVC_ATTN_SPLIT: 0 never, 1 automatic (default)
static int split_count(const svc_attention_args& a) {
  if (g_attn_split < 0) {
    const char* e = getenv("SVC_ATTN_SPLIT");
    g_attn_split = (e && e[0] == '0') ? 0 : 1;
  }
  if (!g_attn_split || a.dk > 128) return 0;
  const int nJ = svc::cdiv(a.T, 32);
  const long long base = (long long)nJ * a.H * a.B;
  // (the unit encoder's 16 query tiles x 12 heads = 192 workgroups of 16 waves already run at 36 TFLOP/s: measured 21 us against
  //  32 us split — the split form pays where the query tiles alone leave most CUs idle)
  if (base >= 128 || nJ < 4) return 0;
  int ks = svc::cdiv(nJ, SPLIT_NW);                       // one key tile per wave ...
  ks = std::min(ks, MAX_KSPLIT);
  while (ks > 1 && base * ks > 1024) --ks;                // ... unless that is more than four workgroups per CU
  return ks;
}

template <int NDT>
int launch_split(const svc_attention_args& a, int ks, hipStream_t s) {
  constexpr int DK = 32 * NDT;
  AttnP p;
  p.a = a;
  p.nJ = svc::cdiv(a.T, 32);
  p.inv_unused = 0.f;
  const size_t lds = (size_t)(SPLIT_NW * DK * 33 + 2 * MAXREL * 32 + 2 * SPLIT_NW * 32 + (DK / 2) * 64 + MAXREL * DK) * 4;
  auto kern = attention_split_kernel<NDT>;
  if (lds > 64 * 1024) {
    static bool done = false;
    if (!done) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      done = true;
    }
  }
  float* ws = reinterpret_cast<float*>(a.ws);
  hipLaunchKernelGGL(kern, dim3(p.nJ, a.H, a.B * ks), dim3(SPLIT_NW * 64), lds, s, p, ws, ks);
  hipLaunchKernelGGL(attention_combine_kernel<NDT>, dim3(p.nJ, a.H, a.B), dim3(256), 0, s, p, (const float*)ws, ks);
  return svc::check_launch("attention_split");
}

int g_attn_nw = 0;   // debug: force 8 or 16 waves (svc_debug_set_attention_waves)

template <int NDT, int NW>
int launch(const svc_attention_args& a, hipStream_t s) {
  constexpr int DK = 32 * NDT;
  constexpr int VROWS = NW > 8 ? 32 : DK, NWC = NW > 8 ? NW / 2 : NW;
  constexpr int SLAB = (NW * VROWS * 33 > NWC * DK * 32) ? NW * VROWS * 33 : NWC * DK * 32;
  AttnP p;
  p.a = a;
  p.nJ = svc::cdiv(a.T, 32);
  p.inv_unused = 0.f;
  const size_t lds = (size_t)(SLAB + 2 * MAXREL * 32 + 2 * NW * 32 + (NW > 8 ? (DK / 2) * 64 : 0)) * 4;
  auto kern = attention_kernel<NDT, NW>;
  if (lds > 64 * 1024) {
    static bool done = false;
    if (!done) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      done = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3(svc::cdiv(a.T, 32), a.H, a.B), dim3(NW * 64), lds, s, p);
  return svc::check_launch("attention");
}

}  // namespace

extern "C" int svc_attention_f32(const svc_attention_args* ap, void* stream) {
  SVC_REQUIRE(ap != nullptr, "attention: null args");
  const svc_attention_args& a = *ap;
  SVC_REQUIRE(a.q && a.k && a.v && a.out, "attention: null tensor");
  SVC_REQUIRE(a.B > 0 && a.H > 0 && a.T > 0, "attention: empty shape");
  SVC_REQUIRE(a.dk % 32 == 0 && a.dk >= 32 && a.dk <= 128, "attention: head dim %d not in {32,64,96,128}", a.dk);
  SVC_REQUIRE(a.window >= 0 && 2 * a.window + 1 <= MAXREL, "attention: window %d too large", a.window);
  SVC_REQUIRE(a.window == 0 || (a.emb_rel_k && a.emb_rel_v), "attention: window set without relative embeddings");
  SVC_REQUIRE(a.mask_mode >= 0 && a.mask_mode <= 2, "attention: bad mask_mode");
  SVC_REQUIRE(a.mask_mode != 1 || a.mask, "attention: mask_mode 1 needs mask");
  hipStream_t s = (hipStream_t)stream;
  const double flop = 4.0 * a.B * a.H * (double)a.T * a.T * a.dk;
  svc::ProfScope prof(s, "attention", flop, 16.0 * a.B * a.H * a.dk * a.T);
  // one utterance (few query tiles, few heads): keys split over workgroups as well, merged by a second small launch
  if (const int ks = split_count(a); ks > 0 && a.ws != nullptr &&
      a.ws_bytes >= (long long)a.B * a.H * svc::cdiv(a.T, 32) * ks * part_floats(a.dk) * 4) {
    switch (a.dk / 32) {
      case 1: return launch_split<1>(a, ks, s);
      case 2: return launch_split<2>(a, ks, s);
      case 3: return launch_split<3>(a, ks, s);
      default: return launch_split<4>(a, ks, s);
    }
  }
  // 16 waves per workgroup when the key walk is long enough to give every wave work and the query tiles alone cannot fill
  // the chip (one utterance: 27 x heads workgroups); many short rows (batched training-size inputs) keep 8
  const int nJ = svc::cdiv(a.T, 32);
  const long long wgs = (long long)nJ * a.H * a.B;
  const bool wide = (g_attn_nw ? g_attn_nw == 16 : (nJ >= 16 && wgs < 512)) && a.dk <= 96;   // dk = 128 would spill at 128 VGPRs
  if (wide) {
    switch (a.dk / 32) {
      case 1: return launch<1, 16>(a, s);
      case 2: return launch<2, 16>(a, s);
      case 3: return launch<3, 16>(a, s);
      default: return launch<4, 16>(a, s);
    }
  }
  switch (a.dk / 32) {
    case 1: return launch<1, 8>(a, s);
    case 2: return launch<2, 8>(a, s);
    case 3: return launch<3, 8>(a, s);
    default: return launch<4, 8>(a, s);
  }
}

extern "C" long long svc_attention_ws_bytes(const svc_attention_args* ap) {
  if (ap == nullptr || ap->dk % 32 != 0 || ap->dk < 32 || ap->dk > 128 || ap->T <= 0) return 0;
  const int ks = split_count(*ap);
  return ks > 0 ? (long long)ap->B * ap->H * svc::cdiv(ap->T, 32) * ks * part_floats(ap->dk) * 4 : 0;
}

extern "C" int svc_debug_set_attention_waves(int nw) {
  if (nw >= 100) {
    g_attn_split = nw - 100 ? 1 : 0;
    return SVC_OK;
  }
  g_attn_nw = (nw == 8 || nw == 16) ? nw : 0;
  return SVC_OK;
}
