// conv1d_hl.hip — the decoder's convolutions at fp32-level precision on the fp16 matrix pipe ("split" pipeline).
//
// gfx950 multiplies fp16 sixteen times faster than fp32 (v_mfma_f32_32x32x16_f16: 2.5 PFLOP/s; v_mfma_f32_32x32x2_f32: 157 TFLOP/s),
// both accumulating in fp32.  A fp32 number v is carried as TWO fp16 numbers, hi = rn16(v), lo = rn16(v - hi): hi + lo reproduces v
// to 22 mantissa bits (|v - hi - lo| <= 2^-22 |v| while lo is a normal or subnormal fp16, i.e. down to 2^-24 absolute), and a product
// of two such pairs is  a_hi b_hi + a_hi b_lo + a_lo b_hi  (+ a_lo b_lo, 2^-22 relative, dropped): THREE fp16 instructions — each
// fp16 x fp16 product is exact in the fp32 accumulator — replace SIXTEEN fp32 ones, and the sums round in fp32 exactly as the fp32
// instruction's do.  The pipeline is the 16-bit one of conv1d_h.hip (blocked [B][C/8][T][8] activations, [Cin/16][tap][rows][16] weight
// packs, B operand = one ds_read_b128, A operand from L2 through a ring of chunks, phases-as-rows ConvTranspose1d) with a second PLANE
// behind every tensor: activations [2][B][C/8][T][8], weights [2][Cin/16][tap][RP][16] (plane 0 = hi, plane 1 = lo) — the same bytes
// as the fp32 tensors.  What the reference computes here: ResBlock1 convs vdecoder/hifigan/models.py:41-67, ups :340-342,378,
// conv_post :390-392 — in fp32 (`net_g_ms` without .half()); measured against that the split pipeline is as close as the fp32 MFMA
// path is (tests/test_split_gpu.py), which is why it is offered as a precision mode of fp32 inference and not of the half mode.
//
// Differences from conv1d_h_kernel: two planes double the LDS tile, so the input channels are staged in chunks of GC groups (one
// barrier pair per chunk; the accumulators live across chunks); the pre-activation is applied to hi + lo in fp32 and re-split;
// the ring holds 2 steps x 2 planes per chunk (a step is 3 MT NT instructions: the same matrix time per chunk as four 16-bit steps).
#include "common.h"
#include <algorithm>
#include <cmath>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

struct HLP {
  svc_conv1d_h_args a;
  int XW;   // columns of the staged tile: BN + (KS - 1) * dil
  int G;    // Cin / 16
  int GC;   // channel groups staged per chunk
  int R;    // valid rows
  long long xplane, yplane, wplane;   // elements (halves) between the hi and the lo plane
  int* flag;                          // range flag word (may be null)
  float acc_scale;                    // 1 / the weight pack's power-of-two scale
};

__device__ __forceinline__ void split1(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)(v - (float)hi);
}
// Range of the representation: hi is an fp16, so a stored value must stay below 65 520 (above it hi = inf and every later product is
// inf / nan), and it is carried to max(2^-22 |s|, 2^-25): below |s| ~ 0.125 the lo piece is a subnormal fp16 and the error is ABSOLUTE.
//   * Weights are packed times a per-tensor power of two that puts max |w| at 2^14 (exact; the accumulators are multiplied back by its
//     inverse in the epilogue: svc_conv1d_h_args.acc_scale).
//   * Activation planes hold ASC * v (ASC = 32, exact): the range of v is +-2047 and the absolute floor 2^-30 (9.3e-10) — 22 bits down
//     to |v| = 0.004, where the unscaled planes of round 5 started losing bits at 0.125 (floor 3e-8).  Nothing else changes: the
//     pre-activation is positively homogeneous (lrelu(32 v) = 32 lrelu(v)), the products carry the factor into the accumulators and the
//     epilogue's acc_scale takes it out; decoding a plane pair is (hi + lo) / 32.
//   * Every value a kernel of this file PRODUCES is range-checked as it is encoded: enc_act ORs "not (|32 v| <= 65504)" (overflow or
//     nan) into `bad`, and a lane with bad set raises the caller's sticky flag word (svc_hl_range_flag) — the caller reads it after
//     the clip and re-runs the fp32 path (SynthesizerTrn.split_range_exceeded / Svc do).
constexpr float ASC = 32.f, IASC = 1.f / 32.f;
__device__ __forceinline__ void enc_act(float v, _Float16& hi, _Float16& lo, bool& bad) {
  const float s = v * ASC;
  hi = (_Float16)s;
  lo = (_Float16)(s - (float)hi);
  bad |= !(fabsf(s) <= 65504.f);
}
__device__ __forceinline__ float dec_act(_Float16 hi, _Float16 lo) { return ((float)hi + (float)lo) * IASC; }
__device__ __forceinline__ void raise_range(int* flag, bool bad) {
  if (bad && flag) atomicOr(flag, 1);
}
thread_local int* t_hl_flag = nullptr;      // svc_hl_range_flag: the flag word the launches of this host thread report into

template <int KS, int MT, int NT, int CH>
__device__ __forceinline__ void mma_steps_hl(f32x16 (&acc)[MT][NT], const h8* __restrict__ wp, long long wplane8, long long sstride,
                                             int S, const h8* __restrict__ xw, int lplane8, int XW, int dil) {
  // wp: this lane's first hi A fragment (+ step * sstride + mt * 64; lo at + wplane8); xw: this lane's hi B base (lo at + lplane8)
  auto wload = [&](h8 (&af)[CH][MT][2], int s0) {
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const h8* q = wp + min(s0 + j, S - 1) * sstride + mt * 64;
        af[j][mt][0] = q[0];
        af[j][mt][1] = q[wplane8];
      }
  };
  // (Reading the B fragments of step s + 1 in front of the instructions of step s — two register sets — was measured: nothing at
  // 64..256 channels, 10-15 % slower at 16 / 32; profiles/r09d_conv_hl_cfg.txt.  The second wave of the SIMD covers the LDS round trip.)
  auto chunk = [&](const h8 (&af)[CH][MT][2], int s0) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int sidx = s0 + j;
      if (sidx < S) {
        const int g = sidx / KS, tap = sidx - g * KS;
        const h8* xr = xw + 2 * g * XW + tap * dil;
        h8 bh[NT], bl[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          bh[nt] = xr[nt * 32];
          bl[nt] = xr[lplane8 + nt * 32];
        }
        // term-major: MT NT independent accumulators between two instructions on the same one
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
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][mt][0], bh[nt], acc[mt][nt], 0, 0, 0);
      }
    }
  };
  h8 a0[CH][MT][2], a1[CH][MT][2];
  wload(a0, 0);
  for (int s0 = 0; s0 < S; s0 += 2 * CH) {
    if (s0 + CH < S) wload(a1, s0 + CH);
    chunk(a0, s0);
    if (s0 + CH < S) {
      if (s0 + 2 * CH < S) wload(a0, s0 + 2 * CH);
      chunk(a1, s0 + CH);
    }
  }
}

template <int KS, int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv1d_hl_kernel(HLP p) {
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  static_assert(WM * WN == 4, "four waves per workgroup");
  const svc_conv1d_h_args& a = p.a;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_hl[];
  const int XW = p.XW, GC = p.GC;
  const int lplane8 = 2 * GC * XW;              // h8 words per LDS plane
  h8* xs = reinterpret_cast<h8*>(smem_hl);      // [2][2 GC][XW]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int wm = w / WN, wn = w - wm * WN;
  const int t0 = blockIdx.x * BN, r0 = blockIdx.y * BM, b = blockIdx.z;
  const int CB = a.Cin >> 3;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const h8* xg = reinterpret_cast<const h8*>(a.x) + (long long)b * CB * a.Tin;
  const long long xplane8 = p.xplane >> 3, wplane8 = p.wplane >> 3;
  const h8* wp = reinterpret_cast<const h8*>(a.w) + ((long long)(r0 + wm * MT * 32 + li)) * 2 + kh;
  const long long sstride = (long long)a.RP * 2;
  const float ps = a.pre_slope;
  const bool act = a.pre_slope != 1.f;

  for (int g0 = 0; g0 < p.G; g0 += GC) {
    const int gc = min(GC, p.G - g0);
    if (g0) __syncthreads();                    // every wave is done with the previous chunk's tile
    // ---- stage channel blocks [2 g0, 2 (g0 + gc)) of both planes; the pre-activation acts on hi + lo and is split again
    {
      const int total = 2 * gc * XW;
      constexpr int LD = 4;
      for (int base = tid; base < total; base += 256 * LD) {
        h8 vh[LD], vl[LD];
#pragma unroll
        for (int j = 0; j < LD; ++j) {
          const int idx = base + j * 256;
          h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
          vh[j] = z;
          vl[j] = z;
          if (idx < total) {
            const int cbl = idx / XW, tl = idx - cbl * XW;
            const int tin = t0 - a.pad_left + tl;
            if (tin >= 0 && tin < a.Tin) {
              const h8* q = xg + (long long)(2 * g0 + cbl) * a.Tin + tin;
              vh[j] = q[0];
              vl[j] = q[xplane8];
            }
          }
        }
#pragma unroll
        for (int j = 0; j < LD; ++j) {
          const int idx = base + j * 256;
          if (idx < total) {
            if (act) {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float v = svc_lrelu((float)vh[j][e] + (float)vl[j][e], ps);
                _Float16 hi, lo;
                split1(v, hi, lo);
                vh[j][e] = hi;
                vl[j][e] = lo;
              }
            }
            xs[idx] = vh[j];
            xs[lplane8 + idx] = vl[j];
          }
        }
      }
    }
    __syncthreads();
    mma_steps_hl<KS, MT, NT, 2>(acc, wp + (long long)g0 * KS * sstride, wplane8, sstride, gc * KS, xs + kh * XW + wn * NT * 32 + li,
                                lplane8, XW, a.dil);
  }

  // ---- epilogue: bias, activation, residual, accumulate / divide in fp32, split, 8-byte stores into both planes
  _Float16* yb = reinterpret_cast<_Float16*>(a.y) + (long long)b * a.Cout * a.Ty;
  const _Float16* rb = a.res ? reinterpret_cast<const _Float16*>(a.res) + (long long)b * a.Cout * a.Ty : nullptr;
  const bool lre = a.post_act == SVC_ACT_LRELU;
  const float asc = p.acc_scale;
  bool bad = false;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int q = t0 + wn * NT * 32 + nt * 32 + li;
      if (q >= a.Tq) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row8 = r0 + (wm * MT + mt) * 32 + 8 * i;      // this lane holds rows row8 + 4 kh + (0..3) of column q
        if (row8 >= p.R) continue;
        int ph = 0, co8 = row8;
        if (a.u > 1) {
          ph = row8 / a.Cout;
          co8 = row8 - ph * a.Cout;
        }
        const int t = a.u > 1 ? q * a.u + ph + a.y_t0 : q;
        if (t < 0 || t >= a.Ty) continue;
        const long long off = ((long long)(co8 >> 3) * a.Ty + t) * 8 + 4 * kh;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[mt][nt][4 * i + e] * asc + (a.bias ? a.bias[co8 + 4 * kh + e] : 0.f);
          if (lre) v[e] = svc_lrelu(v[e], a.post_slope);
        }
        if (rb) {
          const h4 rh = *reinterpret_cast<const h4*>(rb + off);
          const h4 rl = *reinterpret_cast<const h4*>(rb + p.yplane + off);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += dec_act(rh[e], rl[e]);
        }
        if (a.beta != 0.f) {
          const h4 oh = *reinterpret_cast<const h4*>(yb + off);
          const h4 ol = *reinterpret_cast<const h4*>(yb + p.yplane + off);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaf(a.beta, dec_act(oh[e], ol[e]), v[e]);
        }
        if (a.out_div != 1.f) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] /= a.out_div;
        }
        h4 oh, ol;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          _Float16 hi, lo;
          enc_act(v[e], hi, lo, bad);
          oh[e] = hi;
          ol[e] = lo;
        }
        *reinterpret_cast<h4*>(yb + off) = oh;
        *reinterpret_cast<h4*>(yb + p.yplane + off) = ol;
      }
    }
  }
  raise_range(p.flag, bad);
}

constexpr size_t HL_LDS_TARGET = 72 * 1024;   // two workgroups per CU
int g_hl_cfg = 0;   // svc_debug_set_conv_hl: bit 0 = 64 x 128 instead of 64 x 64 tiles for under-filled launches

template <int KS, int MT, int NT, int WM, int WN>
int launch_hl(const svc_conv1d_h_args& a, int R, hipStream_t s) {
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  HLP p;
  p.a = a;
  p.XW = BN + (a.KS - 1) * a.dil;
  p.G = a.Cin / 16;
  p.R = R;
  const size_t per_group = (size_t)64 * p.XW;                 // 2 planes x 2 channel blocks x XW x 16 bytes
  const int gmax = (int)std::max<size_t>(1, HL_LDS_TARGET / per_group);
  const int nch = svc::cdiv(p.G, gmax);
  p.GC = svc::cdiv(p.G, nch);
  p.xplane = (long long)a.B * a.Cin * a.Tin;
  p.yplane = (long long)a.B * a.Cout * a.Ty;
  p.wplane = (long long)p.G * a.KS * a.RP * 16;
  p.flag = t_hl_flag;
  p.acc_scale = (a.acc_scale != 0.f ? a.acc_scale : 1.f) * IASC;      // the B operand planes hold ASC * x
  const size_t lds = per_group * p.GC;
  SVC_REQUIRE(lds <= 160 * 1024, "conv1d_hl: tile of %zu bytes does not fit LDS (KS %d, dil %d)", lds, a.KS, a.dil);
  auto kern = conv1d_hl_kernel<KS, MT, NT, WM, WN>;
  if (lds > 64 * 1024) {
    static bool done = false;
    if (!done) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      done = true;
    }
  }
  dim3 grid(svc::cdiv(a.Tq, BN), svc::cdiv(R, BM), a.B);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
  return svc::check_launch("conv1d_hl");
}

template <int KS>
int launch_hl_ks(const svc_conv1d_h_args& a, int R, hipStream_t s) {
  // tile choice as launch_h_ks (conv1d_h.hip): narrow wave tiles; 64 x 128 where 128 x 128 would leave CUs without a workgroup
  // Tried and dropped (profiles/r09c_conv_hl_cfg.txt): s_setprio around the matrix bursts (-1 %), three workgroups per CU
  // (168 registers: spills, no gain).
  if (R >= 128) {
    const long long wgs128 = (long long)svc::cdiv(R, 128) * svc::cdiv(a.Tq, 128) * a.B;
    if (wgs128 < 200) {
      if (g_hl_cfg & 1) return launch_hl<KS, 2, 1, 1, 4>(a, R, s);    //  64 rows x 128 columns
      // the 256-channel stage of one clip (6 896 columns): 216 workgroups of 64 x 128 are one per CU (30 / 46 / 62 us for 3 / 7 / 11
      // taps), 432 of 64 x 64 two (20 / 36 / 52 us; profiles/r09c_conv_hl_cfg.txt)
      return launch_hl<KS, 1, 1, 2, 2>(a, R, s);                      //  64 rows x  64 columns
    }
    return launch_hl<KS, 2, 2, 2, 2>(a, R, s);                        // 128 rows x 128 columns
  }
  if (R > 32) return launch_hl<KS, 2, 2, 1, 4>(a, R, s);              //  64 rows x 256 columns
  return launch_hl<KS, 1, 2, 1, 4>(a, R, s);                          //  32 rows x 256 columns
}

// ---- fused ResBlock1 pair on the split planes (vdecoder/hifigan/models.py:60-67): y = c2(lrelu(c1(lrelu(x)) + b1)) + b2 + x, ONE launch.
// respair_h_kernel (conv1d_h.hip) with a lo plane behind every tile: the intermediate tile ts (all channels, both planes) stays in
// LDS for the whole launch, the input tile xs is staged in channel chunks while conv1 accumulates (the two planes of both tiles at
// once would not leave room for two workgroups per CU).  From 64 channels down the single launches move their algorithmic bytes at
// 2.5-4 TB/s (profiles/r09a_conv_hl_shapes.txt): the intermediate's round trip is a third of the pair's traffic.  Built for C <= 128
// (four waves and two workgroups per CU up to 64 channels; eight waves and one workgroup at 65..128: launch_respair_hl_ks).
struct PPL {
  const void* x;
  const void* w1;
  const void* w2;
  const float* b1;
  const float* b2;
  void* y;
  int B, C, T, d1, RP, GC;
  float slope, beta, out_div;
  float s1, s2;     // 1 / the power-of-two scales of the two weight packs
  int* flag;
};

template <int KS, int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, 2) void respair_hl_kernel(PPL p) {
  constexpr int NTHR = 64 * WM * WN;          // four waves (two workgroups per CU) or eight (one: the 128-channel form)
  constexpr int N1P = 32 * NT * WN;            // intermediate columns computed per workgroup
  constexpr int N2 = N1P - (KS - 1);           // output columns per workgroup
  constexpr int H2 = (KS - 1) / 2;
  static_assert(WM * WN == 4 || WM * WN == 8, "four or eight waves per workgroup");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_hl[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int wm = w / WN, wn = w - wm * WN;
  const int CB = p.C >> 3, G = p.C >> 4, GC = p.GC;
  const int h1 = H2 * p.d1;
  const int XW1 = N1P + 2 * h1, XW2 = N1P + (KS - 1);
  const int tplane8 = CB * XW2, xplane8l = 2 * GC * XW1;
  h8* ts = reinterpret_cast<h8*>(smem_hl);     // [2][CB][XW2]    lrelu(c1 + b1), tile column j <-> global t0 - H2 + j
  h8* xs = ts + 2 * tplane8;                   // [2][2 GC][XW1]  lrelu(x) of one channel chunk, column tl <-> global t0 - H2 - h1 + tl
  const int t0 = blockIdx.x * N2, b = blockIdx.y;
  const long long gplane8 = (long long)p.B * CB * p.T;        // h8 words between the planes of x / y
  const h8* xg = reinterpret_cast<const h8*>(p.x) + (long long)b * CB * p.T;
  const long long wplane8 = (long long)G * KS * p.RP * 2;
  const long long rowoff = ((long long)(wm * MT * 32 + li)) * 2 + kh;
  const long long sstride = (long long)p.RP * 2;
  const int col = wn * NT * 32 + li;

  for (int idx = tid; idx < 2 * CB * (KS - 1); idx += NTHR) {   // the halo tail of the intermediate tile (never computed)
    const int pl = idx / (CB * (KS - 1)), r = idx - pl * CB * (KS - 1);
    const int cb = r / (KS - 1), e = r - cb * (KS - 1);
    h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    ts[pl * tplane8 + cb * XW2 + N1P + e] = z;
  }

  bool bad = false;
  // ---- conv1 over the N1P intermediate columns, the input staged chunk by chunk
  {
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    for (int g0 = 0; g0 < G; g0 += GC) {
      const int gc = min(GC, G - g0);
      if (g0) __syncthreads();
      const int total = 2 * gc * XW1;
      constexpr int LD = 4;
      for (int base = tid; base < total; base += NTHR * LD) {
        h8 vh[LD], vl[LD];
#pragma unroll
        for (int j = 0; j < LD; ++j) {
          const int idx = base + j * NTHR;
          h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
          vh[j] = z;
          vl[j] = z;
          if (idx < total) {
            const int cbl = idx / XW1, tl = idx - cbl * XW1;
            const int tin = t0 - H2 - h1 + tl;
            if (tin >= 0 && tin < p.T) {
              const h8* q = xg + (long long)(2 * g0 + cbl) * p.T + tin;
              vh[j] = q[0];
              vl[j] = q[gplane8];
            }
          }
        }
#pragma unroll
        for (int j = 0; j < LD; ++j) {
          const int idx = base + j * NTHR;
          if (idx < total) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float v = svc_lrelu((float)vh[j][e] + (float)vl[j][e], p.slope);
              _Float16 hi, lo;
              split1(v, hi, lo);
              vh[j][e] = hi;
              vl[j][e] = lo;
            }
            xs[idx] = vh[j];
            xs[xplane8l + idx] = vl[j];
          }
        }
      }
      __syncthreads();
      mma_steps_hl<KS, MT, NT, 2>(acc, reinterpret_cast<const h8*>(p.w1) + rowoff + (long long)g0 * KS * sstride, wplane8, sstride,
                                  gc * KS, xs + kh * XW1 + col, xplane8l, XW1, p.d1);
    }
    _Float16* th = reinterpret_cast<_Float16*>(ts);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int j = col + nt * 32;
        const int tg = t0 - H2 + j;
        const bool inside = tg >= 0 && tg < p.T;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row8 = (wm * MT + mt) * 32 + 8 * i;
          if (row8 >= p.C) continue;
          h4 oh, ol;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = inside ? svc_lrelu(acc[mt][nt][4 * i + e] * p.s1 + p.b1[row8 + 4 * kh + e], p.slope) : 0.f;
            _Float16 hi, lo;
            enc_act(v, hi, lo, bad);
            oh[e] = hi;
            ol[e] = lo;
          }
          const long long o = ((long long)((row8 >> 3) * XW2 + j)) * 8 + 4 * kh;
          *reinterpret_cast<h4*>(th + o) = oh;
          *reinterpret_cast<h4*>(th + (long long)tplane8 * 8 + o) = ol;
        }
      }
  }
  __syncthreads();

  // ---- conv2 from the intermediate tile, residual, accumulate, split, store
  {
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    mma_steps_hl<KS, MT, NT, 2>(acc, reinterpret_cast<const h8*>(p.w2) + rowoff, wplane8, sstride, G * KS, ts + kh * XW2 + col, tplane8,
                                XW2, 1);
    const long long gplane = gplane8 * 8;
    _Float16* yb = reinterpret_cast<_Float16*>(p.y) + (long long)b * p.C * p.T;
    const _Float16* xb = reinterpret_cast<const _Float16*>(p.x) + (long long)b * p.C * p.T;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int j = col + nt * 32;
        const int t = t0 + j;
        if (j >= N2 || t >= p.T) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row8 = (wm * MT + mt) * 32 + 8 * i;
          if (row8 >= p.C) continue;
          const long long off = ((long long)(row8 >> 3) * p.T + t) * 8 + 4 * kh;
          const h4 rh = *reinterpret_cast<const h4*>(xb + off);
          const h4 rl = *reinterpret_cast<const h4*>(xb + gplane + off);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * i + e] * p.s2 + p.b2[row8 + 4 * kh + e] + dec_act(rh[e], rl[e]);
          if (p.beta != 0.f) {
            const h4 oh = *reinterpret_cast<const h4*>(yb + off);
            const h4 ol = *reinterpret_cast<const h4*>(yb + gplane + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(p.beta, dec_act(oh[e], ol[e]), v[e]);
          }
          if (p.out_div != 1.f) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] /= p.out_div;
          }
          h4 oh, ol;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            _Float16 hi, lo;
            enc_act(v[e], hi, lo, bad);
            oh[e] = hi;
            ol[e] = lo;
          }
          *reinterpret_cast<h4*>(yb + off) = oh;
          *reinterpret_cast<h4*>(yb + gplane + off) = ol;
        }
      }
  }
  raise_range(p.flag, bad);
}

template <int KS, int MT, int NT, int WM, int WN>
int launch_respair_hl(PPL p, hipStream_t s) {
  constexpr int N1P = 32 * NT * WN, N2 = N1P - (KS - 1);
  const int h1 = (KS - 1) / 2 * p.d1;
  const int XW1 = N1P + 2 * h1, XW2 = N1P + KS - 1;
  const int G = p.C / 16;
  const size_t ts_bytes = (size_t)2 * (p.C / 8) * XW2 * 16;
  const size_t per_group = (size_t)64 * XW1;
  const size_t budget = (WM * WN == 8 ? 150 : 78) * 1024;   // two four-wave workgroups per CU, or one of eight waves
  const int gmax = ts_bytes + per_group <= budget ? (int)((budget - ts_bytes) / per_group) : 1;
  const int nch = svc::cdiv(G, gmax);
  p.GC = svc::cdiv(G, nch);
  const size_t lds = ts_bytes + per_group * p.GC;
  SVC_REQUIRE(lds <= 160 * 1024, "resblock_pair_hl: tiles of %zu bytes do not fit LDS", lds);
  auto kern = respair_hl_kernel<KS, MT, NT, WM, WN>;
  if (lds > 64 * 1024) {
    static bool done = false;
    if (!done) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      done = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3(svc::cdiv(p.T, N2), p.B), dim3(64 * WM * WN), lds, s, p);
  return svc::check_launch("resblock_pair_hl");
}

template <int KS>
int launch_respair_hl_ks(const PPL& p, hipStream_t s) {
  // 128 channels: both tiles' two planes are 116-139 KB — ONE workgroup per CU, so it has eight waves (two per SIMD, as two
  // four-wave workgroups would), 32 x 64 wave tiles: the weight bytes per instruction of the 128 x 128 single launches
  if (p.C > 64) return launch_respair_hl<KS, 1, 2, 4, 2>(p, s);    // 128 rows, 128 intermediate columns, 512 threads
  if (p.C > 32) return launch_respair_hl<KS, 2, 1, 1, 4>(p, s);    // 64 rows, 128 intermediate columns
  return launch_respair_hl<KS, 1, 2, 1, 4>(p, s);                  // 32 rows, 256
}

// ---- weight pack: dense fp32 -> [2][Cin/16][tap][RP][16] fp16 (hi plane, lo plane); index rules of pack_h_kernel (conv1d_h.hip)
__global__ void pack_hl_kernel(const float* __restrict__ w, _Float16* __restrict__ dst, int Cout, int Cin, int K, int taps, int RP,
                               int u, long long n, float scale) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int j = (int)(idx & 15);
  long long r = idx >> 4;
  const int row = (int)(r % RP);
  r /= RP;
  const int tap = (int)(r % taps), g = (int)(r / taps);
  const int ci = g * 16 + j;
  float v = 0.f;
  if (u <= 1) {
    if (row < Cout) v = w[((long long)row * Cin + ci) * K + tap];
  } else if (row < u * Cout) {
    const int ph = row / Cout, co = row - ph * Cout;
    const int k = ph + (taps - 1 - tap) * u;
    if (k < K) v = w[((long long)ci * Cout + co) * K + k];
  }
  _Float16 hi, lo;
  split1(v * scale, hi, lo);      // scale: a power of two chosen by the caller from max |w| (exact; 1 / scale goes to acc_scale)
  dst[idx] = hi;
  dst[n + idx] = lo;
}

// ---- fp32 [B,C,T] (strided) (+ a second fp32 tensor) -> split blocked planes, and back
__global__ void cvt_to_hl_kernel(const float* __restrict__ x, const float* __restrict__ add, h8* __restrict__ y, long long x_bs,
                                 long long x_cs, long long a_bs, long long a_cs, int B, int C, int T, int* flag) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int CB = C >> 3;
  const long long n = (long long)B * CB * T;
  if (idx >= n) return;
  const int t = (int)(idx % T);
  const long long r = idx / T;
  const int cb = (int)(r % CB), b = (int)(r / CB);
  h8 oh, ol;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = x[(long long)b * x_bs + (long long)(cb * 8 + j) * x_cs + t];
    if (add) v += add[(long long)b * a_bs + (long long)(cb * 8 + j) * a_cs + t];
    _Float16 hi, lo;
    enc_act(v, hi, lo, bad);
    oh[j] = hi;
    ol[j] = lo;
  }
  y[idx] = oh;
  y[n + idx] = ol;
  raise_range(flag, bad);
}

__global__ void cvt_from_hl_kernel(const h8* __restrict__ x, float* __restrict__ y, int B, int C, int T) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int CB = C >> 3;
  const long long n = (long long)B * CB * T;
  if (idx >= n) return;
  const int t = (int)(idx % T);
  const long long r = idx / T;
  const int cb = (int)(r % CB), b = (int)(r / CB);
  const h8 vh = x[idx], vl = x[n + idx];
#pragma unroll
  for (int j = 0; j < 8; ++j) y[((long long)b * C + cb * 8 + j) * T + t] = dec_act(vh[j], vl[j]);
}

// ---- conv_post (vdecoder/hifigan/models.py:390-392) from the split planes: leaky_relu(0.01) -> Conv1d(C, 1, KS) -> tanh, fp32 out
__global__ __launch_bounds__(256) void conv_post_hl_kernel(const h8* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int B, int C, int T,
                                                           int KS, int pad, float pre_slope, int act) {
  extern __shared__ float wsh_hl[];       // [C][KS]
  for (int i = threadIdx.x; i < C * KS; i += blockDim.x) wsh_hl[i] = w[i];
  __syncthreads();
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * T) return;
  const int t = (int)(idx % T), b = (int)(idx / T);
  const int CB = C >> 3;
  const long long plane = (long long)B * CB * T;
  float acc = bias ? bias[0] : 0.f;
  for (int cb = 0; cb < CB; ++cb) {
    const h8* xr = x + ((long long)b * CB + cb) * T;
    for (int k = 0; k < KS; ++k) {
      const int tin = t - pad + k;
      if (tin < 0 || tin >= T) continue;
      const h8 vh = xr[tin], vl = xr[plane + tin];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = fmaf(svc_lrelu(dec_act(vh[j], vl[j]), pre_slope), wsh_hl[(cb * 8 + j) * KS + k], acc);
    }
  }
  y[idx] = act == SVC_ACT_TANH ? tanhf(acc) : acc;
}

// ---- SnakeAlias on the split planes (vdecoder/hifiganwithsnake/alias/act.py:125-130; snake_alias_h_kernel of conv1d_h.hip with a lo
// plane behind the input and the output): UpSample1d x2 -> SnakeBeta -> DownSample1d x2, the 2x intermediate in LDS, fp32 arithmetic
// between reading hi + lo and storing the split result.
constexpr int SHL_TILE = 256;
struct TapsHL {
  float f[12];
};
__global__ __launch_bounds__(256) void snake_alias_hl_kernel(const h8* __restrict__ x, h8* __restrict__ y, const float* __restrict__ alpha,
                                                             const float* __restrict__ beta, TapsHL taps, int CB, int T, long long plane8,
                                                             int* flag) {
  __shared__ float xs[(SHL_TILE + 10) * 8];
  __shared__ float ua[(2 * SHL_TILE + 12) * 8];
  const int t0 = blockIdx.x * SHL_TILE, cb = blockIdx.y, b = blockIdx.z;
  const h8* xr = x + ((long long)b * CB + cb) * T;
  h8* yr = y + ((long long)b * CB + cb) * T;
  const int tid = threadIdx.x;
  for (int i = tid; i < SHL_TILE + 10; i += 256) {
    int t = t0 - 5 + i;
    t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
    const h8 vh = xr[t], vl = xr[plane8 + t];
#pragma unroll
    for (int j = 0; j < 8; ++j) xs[i * 8 + j] = dec_act(vh[j], vl[j]);
  }
  __syncthreads();
  const int n_lo = 2 * t0 - 5;
  for (int e = tid; e < (2 * SHL_TILE + 10) * 8; e += 256) {
    const int m = e >> 3, j = e & 7;
    int n = n_lo + m;
    n = n < 0 ? 0 : (n > 2 * T - 1 ? 2 * T - 1 : n);
    const int par = (n + 1) & 1;
    const int j0 = (n + 15 - par) >> 1;
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      int xi = j0 - q - 5;
      xi = xi < 0 ? 0 : (xi > T - 1 ? T - 1 : xi);
      acc = fmaf(par ? taps.f[2 * q + 1] : taps.f[2 * q], xs[(xi - (t0 - 5)) * 8 + j], acc);
    }
    const float u = 2.f * acc;
    const int c = cb * 8 + j;
    const float sn = sinf(u * __expf(alpha[c]));
    ua[m * 8 + j] = u + (sn * sn) / (__expf(beta[c]) + 1e-9f);
  }
  __syncthreads();
  bool bad = false;
  for (int i = tid; i < SHL_TILE; i += 256) {
    const int t = t0 + i;
    if (t >= T) break;
    h8 oh, ol;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 12; ++k) acc = fmaf(taps.f[k], ua[(2 * i + k) * 8 + j], acc);
      _Float16 hi, lo;
      enc_act(acc, hi, lo, bad);
      oh[j] = hi;
      ol[j] = lo;
    }
    yr[t] = oh;
    yr[plane8 + t] = ol;
  }
  raise_range(flag, bad);
}

}  // namespace

namespace svc {
int* hl_range_flag_ptr() { return t_hl_flag; }      // for the other split kernels of the library (flow_fused.hip)
}

extern "C" int svc_hl_range_flag(int* flag) {
  t_hl_flag = flag;
  return SVC_OK;
}

extern "C" int svc_pack_conv1d_hl(const float* w, void* dst, int Cout, int Cin, int K, int u, int RP, float scale, void* stream) {
  SVC_REQUIRE(w && dst, "pack_conv1d_hl: null tensor");
  {
    int ex = 0;
    SVC_REQUIRE(scale > 0.f && std::isfinite(scale) && std::frexp(scale, &ex) == 0.5f, "pack_conv1d_hl: scale %g is not a power of two", (double)scale);
  }
  SVC_REQUIRE(Cout > 0 && Cin > 0 && (Cin % 16) == 0 && K >= 1 && u >= 1, "pack_conv1d_hl: bad shape (Cin must be a multiple of 16)");
  const int taps = u > 1 ? (K + u - 1) / u : K;
  const int R = u > 1 ? u * Cout : Cout;
  SVC_REQUIRE(RP >= R && (RP % 128) == 0, "pack_conv1d_hl: RP must be a multiple of 128 >= the row count %d", R);
  const long long n = (long long)taps * (Cin / 16) * RP * 16;
  hipLaunchKernelGGL(pack_hl_kernel, dim3((unsigned)svc::cdivll(n, 256)), dim3(256), 0, (hipStream_t)stream, w,
                     reinterpret_cast<_Float16*>(dst), Cout, Cin, K, taps, RP, u, n, scale);
  return svc::check_launch("pack_conv1d_hl");
}

extern "C" int svc_conv1d_hl(const svc_conv1d_h_args* ap, void* stream) {
  SVC_REQUIRE(ap != nullptr, "conv1d_hl: null args");
  const svc_conv1d_h_args& a = *ap;
  SVC_REQUIRE(a.x && a.w && a.y, "conv1d_hl: null tensor");
  SVC_REQUIRE(a.B > 0 && a.Cin > 0 && a.Cout > 0 && a.Tin > 0 && a.Tq > 0 && a.Ty > 0, "conv1d_hl: empty shape");
  SVC_REQUIRE((a.Cin % 16) == 0 && (a.Cout % 8) == 0, "conv1d_hl: Cin must be a multiple of 16 and Cout of 8 (got %d, %d)", a.Cin, a.Cout);
  SVC_REQUIRE(a.dil >= 1 && a.u >= 1 && (a.RP % 128) == 0, "conv1d_hl: bad dil / u / RP");
  SVC_REQUIRE(a.post_act == SVC_ACT_NONE || a.post_act == SVC_ACT_LRELU, "conv1d_hl: post_act must be none or leaky_relu");
  SVC_REQUIRE(((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.w) | reinterpret_cast<uintptr_t>(a.y) |
                reinterpret_cast<uintptr_t>(a.res)) & 15) == 0, "conv1d_hl: tensors must be 16-byte aligned");
  SVC_REQUIRE((((long long)a.B * a.Cin * a.Tin) & 7) == 0 && (((long long)a.B * a.Cout * a.Ty) & 7) == 0, "conv1d_hl: plane size");
  const int R = a.u > 1 ? a.u * a.Cout : a.Cout;
  SVC_REQUIRE(a.RP >= R, "conv1d_hl: RP %d below the row count %d", a.RP, R);
  hipStream_t s = (hipStream_t)stream;
  // profile rows carry the fp32 convolution's FLOPs (what the launch delivers), not the three instructions issued per product
  const double flop = 2.0 * a.B * (double)R * a.Cin * a.KS * a.Tq;
  const double bytes = 4.0 * a.B * ((double)a.Cin * a.Tin + (double)a.Cout * a.Ty * (a.res ? 2 : 1));
  char pname[96];
  if (svc::prof_on() && svc::prof_shapes())
    snprintf(pname, sizeof(pname), "%s[B%d,Ci%d,Co%d,K%d,d%d,T%d]", a.u > 1 ? "convt1d_hl" : "conv1d_hl", a.B, a.Cin, a.Cout, a.KS, a.dil, a.Tq);
  else
    snprintf(pname, sizeof(pname), "%s", a.u > 1 ? "convt1d_hl" : "conv1d_hl");
  svc::ProfScope prof(s, pname, flop, bytes);
  switch (a.KS) {
    case 1: return launch_hl_ks<1>(a, R, s);
    case 2: return launch_hl_ks<2>(a, R, s);
    case 3: return launch_hl_ks<3>(a, R, s);
    case 5: return launch_hl_ks<5>(a, R, s);
    case 7: return launch_hl_ks<7>(a, R, s);
    case 11: return launch_hl_ks<11>(a, R, s);
    default: SVC_REQUIRE(false, "conv1d_hl: tap count %d not built (1, 2, 3, 5, 7, 11)", a.KS);
  }
  return SVC_OK;
}

extern "C" int svc_cvt_to_hl(const float* x, const float* add, void* y, long long x_bs, long long x_cs, long long add_bs,
                             long long add_cs, int B, int C, int T, void* stream) {
  SVC_REQUIRE(x && y && B > 0 && C > 0 && (C % 8) == 0 && T > 0, "cvt_to_hl: bad args (C must be a multiple of 8)");
  const long long n = (long long)B * (C / 8) * T;
  hipLaunchKernelGGL(cvt_to_hl_kernel, dim3((unsigned)svc::cdivll(n, 256)), dim3(256), 0, (hipStream_t)stream, x, add,
                     reinterpret_cast<h8*>(y), x_bs, x_cs, add_bs, add_cs, B, C, T, t_hl_flag);
  return svc::check_launch("cvt_to_hl");
}

extern "C" int svc_cvt_from_hl(const void* x, float* y, int B, int C, int T, void* stream) {
  SVC_REQUIRE(x && y && B > 0 && C > 0 && (C % 8) == 0 && T > 0, "cvt_from_hl: bad args");
  const long long n = (long long)B * (C / 8) * T;
  hipLaunchKernelGGL(cvt_from_hl_kernel, dim3((unsigned)svc::cdivll(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const h8*>(x), y, B, C, T);
  return svc::check_launch("cvt_from_hl");
}

extern "C" int svc_conv_post_hl(const void* x, const float* w, const float* bias, float* y, int B, int C, int T, int KS, int pad,
                                float pre_slope, int act, void* stream) {
  SVC_REQUIRE(x && w && y && B > 0 && C > 0 && (C % 8) == 0 && T > 0 && KS >= 1, "conv_post_hl: bad args");
  SVC_REQUIRE((size_t)C * KS * 4 <= 48 * 1024, "conv_post_hl: weight does not fit LDS");
  const long long n = (long long)B * T;
  svc::ProfScope prof((hipStream_t)stream, "conv_post_hl", 2.0 * n * C * KS, 4.0 * n * C + 4.0 * n);
  hipLaunchKernelGGL(conv_post_hl_kernel, dim3((unsigned)svc::cdivll(n, 256)), dim3(256), (size_t)C * KS * 4, (hipStream_t)stream,
                     reinterpret_cast<const h8*>(x), w, bias, y, B, C, T, KS, pad, pre_slope, act);
  return svc::check_launch("conv_post_hl");
}

extern "C" int svc_resblock_pair_hl(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, int B,
                                   int C, int T, int KS, int dil1, int RP, float slope, float beta, float out_div, float acc_scale1,
                                   float acc_scale2, void* stream) {
  SVC_REQUIRE(x && w1 && w2 && b1 && b2 && y, "resblock_pair_hl: null tensor");
  SVC_REQUIRE(B > 0 && T > 0 && C >= 16 && C <= 128 && (C % 16) == 0, "resblock_pair_hl: C must be a multiple of 16 in 16..128 (got %d)", C);
  SVC_REQUIRE(dil1 >= 1 && (RP % 128) == 0 && RP >= C, "resblock_pair_hl: bad dil1 / RP");
  SVC_REQUIRE(slope > 0.f && slope <= 1.f, "resblock_pair_hl: leaky-ReLU slope in (0, 1]");
  SVC_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2) |
                reinterpret_cast<uintptr_t>(y)) & 15) == 0, "resblock_pair_hl: tensors must be 16-byte aligned");
  PPL p;
  p.x = x; p.w1 = w1; p.w2 = w2; p.b1 = b1; p.b2 = b2; p.y = y;
  p.B = B; p.C = C; p.T = T; p.d1 = dil1; p.RP = RP; p.GC = 1;
  p.slope = slope; p.beta = beta; p.out_div = out_div;
  p.s1 = (acc_scale1 != 0.f ? acc_scale1 : 1.f) * IASC;      // both convs read planes that hold ASC * value
  p.s2 = (acc_scale2 != 0.f ? acc_scale2 : 1.f) * IASC;
  p.flag = t_hl_flag;
  hipStream_t s = (hipStream_t)stream;
  char pname[96];
  if (svc::prof_on() && svc::prof_shapes()) snprintf(pname, sizeof(pname), "resblock_pair_hl[B%d,C%d,K%d,d%d,T%d]", B, C, KS, dil1, T);
  else snprintf(pname, sizeof(pname), "resblock_pair_hl");
  svc::ProfScope prof(s, pname, 4.0 * B * (double)C * C * KS * T, 4.0 * B * (double)C * T * 3);
  switch (KS) {
    case 3: return launch_respair_hl_ks<3>(p, s);
    case 7: return launch_respair_hl_ks<7>(p, s);
    case 11: return launch_respair_hl_ks<11>(p, s);
    default: SVC_REQUIRE(false, "resblock_pair_hl: tap count %d not built (3, 7, 11)", KS);
  }
  return SVC_OK;
}

extern "C" int svc_debug_set_conv_hl(int cfg) {
  g_hl_cfg = cfg;
  return SVC_OK;
}

extern "C" int svc_snake_alias_hl(const void* x, void* y, const float* alpha, const float* beta, const float* taps12, int B, int C, int T,
                                  void* stream) {
  SVC_REQUIRE(x && y && alpha && beta && taps12 && B > 0 && C > 0 && (C % 8) == 0 && T > 0, "snake_alias_hl: bad args");
  TapsHL tp;
  for (int i = 0; i < 12; ++i) tp.f[i] = taps12[i];
  svc::ProfScope prof((hipStream_t)stream, "snake_alias_hl", 0.0, 8.0 * B * (double)C * T);
  hipLaunchKernelGGL(snake_alias_hl_kernel, dim3(svc::cdiv(T, SHL_TILE), C / 8, B), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const h8*>(x), reinterpret_cast<h8*>(y), alpha, beta, tp, C / 8, T, (long long)B * (C / 8) * T,
                     t_hl_flag);
  return svc::check_launch("snake_alias_hl");
}
