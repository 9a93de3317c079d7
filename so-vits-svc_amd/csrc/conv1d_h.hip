// conv1d_h.hip — the decoder's convolutions as a 16-bit pipeline: fp16 activations in HBM AND in LDS, fp16 weights packed once,
// v_mfma_f32_32x32x16_f16 with fp32 accumulation.  This is the engine's form of the reference's half-precision inference
// (inference/infer_tool.py:196-198: a checkpoint whose name contains "half" — compress_model.py:21-48 — runs `net_g_ms.half()`):
// there every tensor of the model is fp16; here the NSF-HiFiGAN generator (94 % of the FLOPs: ResBlock1 convs
// vdecoder/hifigan/models.py:41-67, ups :340-342,378, conv_post :390-392) stores and multiplies in fp16 while the encoder / flow /
// harmonic source keep the fp32 kernels (they are latency-bound at 6 % of the FLOPs; the source's phase integration is not
// representable in fp16 at all — the reference's own half mode loses 8e-4 MSE there).
//
// Layout.  An activation tensor is [B][C/8][T][8] fp16 ("blocked": 8 channels of one time step are 16 contiguous bytes).  Why: the
// 32x32x16 instruction reduces 16 input channels of ONE tap per issue, lane (n = lane & 31, kh = lane >> 5) supplying channels
// 8*kh .. 8*kh+7 of column n — in this layout that B operand is ONE ds_read_b128 at [(2g + kh)][t + tap*dil] (a dilated tap is an
// address offset, consecutive lanes read consecutive 16-byte words: conflict-free), a tile row is staged by 16-byte loads that
// are contiguous along time, and the C layout (lane holds rows 8i + 4kh .. +3 of column n) stores 8-byte groups of 4 channels.
// A whole input-channel extent fits LDS at once (256 channels x 178 columns = 91 KB): no chunk loop, one barrier per launch.
// The A operand (weights, [Cin/16][tap][rows][16] fp16: one linear index per (channel group, tap) step) is read straight from
// L2 into registers, one 16-byte load per lane per (step, row tile), through a ring of two 4-step chunks: the loads of chunk c + 1
// are issued in front of the instructions of chunk c.  A wave owns a (32 MT) x (32 NT) tile: every B fragment feeds MT
// instructions (the LDS port delivers one 1 KB fragment per 8 clocks, the matrix pipe eats one per 32 clocks per SIMD: with
// MT = 1 four SIMDs would run the port at 100 %) and every A fragment NT of them — first measurements (profiles/r08a_*: 181 TFLOP/s
// with 32 x 64 / 64 x 64 wave tiles) were bound by the weight stream (every workgroup pulls its rows' whole weight slab through
// L1: bytes per FLOP = 1 / tile width), hence NT = 4 wherever the sequence is long enough to still fill the chip.
//
// ConvTranspose1d (ups) = the same kernel on "phases as rows": row = phase * Cout + co, M = ceil(K / u) taps, the epilogue writes
// row (phase, co), column q to y[co][q * u + phase - padding].
#include "common.h"
#include <algorithm>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

struct HP {
  svc_conv1d_h_args a;
  int XW;  // columns of the staged tile: BN + (KS - 1) * dil
  int G;   // Cin / 16
  int R;   // valid rows: Cout (conv) or u * Cout (transposed)
};

__device__ __forceinline__ h8 lrelu8(h8 v, _Float16 s) {
  const h8 sv = v * s;
  return __builtin_elementwise_max(v, sv);
}

template <int KS, int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv1d_h_kernel(HP p) {
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  constexpr int CH = 4;                         // steps per chunk of the weight ring
  static_assert(WM * WN == 4, "four waves per workgroup");
  const svc_conv1d_h_args& a = p.a;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
  h8* xs = reinterpret_cast<h8*>(smem_h);       // [Cin/8][XW]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int wm = w / WN, wn = w - wm * WN;
  const int t0 = blockIdx.x * BN, r0 = blockIdx.y * BM, b = blockIdx.z;
  const int CB = a.Cin >> 3, XW = p.XW;
  const int S = p.G * KS;                       // (channel group, tap) steps, group-major

  // ---- the first weight chunk now: it depends on nothing (L2 latency runs under the staging below)
  const h8* wp = reinterpret_cast<const h8*>(a.w) + ((long long)(r0 + wm * MT * 32 + li)) * 2 + kh;   // + (step * RP + mt * 32) * 2
  const long long sstride = (long long)a.RP * 2;
  auto wload = [&](h8 (&af)[CH][MT], int s0) {
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int sidx = min(s0 + j, S - 1);    // (a chunk may reach past the last step: re-read it, never use it)
        af[j][mt] = wp[sidx * sstride + mt * 64];
      }
  };
  h8 a0[CH][MT], a1[CH][MT];
  wload(a0, 0);

  // ---- stage the activation tile: 16-byte words, contiguous along time within a channel block; pre-activation applied once;
  // 8 loads in flight per thread
  {
    const h8* xg = reinterpret_cast<const h8*>(a.x) + (long long)b * CB * a.Tin;
    const _Float16 ps = (_Float16)a.pre_slope;
    const bool act = a.pre_slope != 1.f;
    const int total = CB * XW;
    constexpr int LD = 8;
    for (int base = tid; base < total; base += 256 * LD) {
      h8 v[LD];
#pragma unroll
      for (int j = 0; j < LD; ++j) {
        const int idx = base + j * 256;
        h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        v[j] = z;
        if (idx < total) {
          const int cb = idx / XW, tl = idx - cb * XW;
          const int tin = t0 - a.pad_left + tl;
          if (tin >= 0 && tin < a.Tin) v[j] = xg[(long long)cb * a.Tin + tin];
        }
      }
#pragma unroll
      for (int j = 0; j < LD; ++j) {
        const int idx = base + j * 256;
        if (idx < total) xs[idx] = act ? lrelu8(v[j], ps) : v[j];
      }
    }
  }
  __syncthreads();

  // ---- matrix loop over the steps, four at a time; the weights of the next four are in flight meanwhile
  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  const h8* xw = xs + kh * XW + wn * NT * 32 + li;
  const int dil = a.dil;
  auto chunk = [&](const h8 (&af)[CH][MT], int s0) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int sidx = s0 + j;
      if (sidx < S) {
        const int g = sidx / KS, tap = sidx - g * KS;
        const h8* xr = xw + 2 * g * XW + tap * dil;
        h8 bq[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bq[nt] = xr[nt * 32];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][mt], bq[nt], acc[mt][nt], 0, 0, 0);
      }
    }
  };
  for (int s0 = 0; s0 < S; s0 += 2 * CH) {
    if (s0 + CH < S) wload(a1, s0 + CH);
    chunk(a0, s0);
    if (s0 + CH < S) {
      if (s0 + 2 * CH < S) wload(a0, s0 + 2 * CH);
      chunk(a1, s0 + CH);
    }
  }

  // ---- epilogue straight from the accumulators: bias, activation, residual, accumulate / divide, fp16, 8-byte stores
  _Float16* yb = reinterpret_cast<_Float16*>(a.y) + (long long)b * a.Cout * a.Ty;
  const _Float16* rb = a.res ? reinterpret_cast<const _Float16*>(a.res) + (long long)b * a.Cout * a.Ty : nullptr;
  const bool lre = a.post_act == SVC_ACT_LRELU;
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
          v[e] = acc[mt][nt][4 * i + e] + (a.bias ? a.bias[co8 + 4 * kh + e] : 0.f);
          if (lre) v[e] = svc_lrelu(v[e], a.post_slope);
        }
        if (rb) {
          const h4 rv = *reinterpret_cast<const h4*>(rb + off);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
        }
        if (a.beta != 0.f) {
          const h4 ov = *reinterpret_cast<const h4*>(yb + off);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaf(a.beta, (float)ov[e], v[e]);
        }
        if (a.out_div != 1.f) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] /= a.out_div;
        }
        h4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
        *reinterpret_cast<h4*>(yb + off) = o;
      }
    }
  }
}

template <int KS, int MT, int NT, int WM, int WN>
int launch_h(const svc_conv1d_h_args& a, int R, hipStream_t s) {
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  HP p;
  p.a = a;
  p.XW = BN + (a.KS - 1) * a.dil;
  p.G = a.Cin / 16;
  p.R = R;
  const size_t lds = (size_t)(a.Cin / 8) * p.XW * 16;
  SVC_REQUIRE(lds <= 160 * 1024, "conv1d_h: tile of %zu bytes does not fit LDS (Cin %d, KS %d, dil %d)", lds, a.Cin, a.KS, a.dil);
  auto kern = conv1d_h_kernel<KS, MT, NT, WM, WN>;
  if (lds > 64 * 1024) {
    static bool done = false;
    if (!done) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      done = true;
    }
  }
  dim3 grid(svc::cdiv(a.Tq, BN), svc::cdiv(R, BM), a.B);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
  return svc::check_launch("conv1d_h");
}

int g_h_cfg = 0;   // svc_debug_set_conv_h(cfg): 0 automatic; 1 128 x 128 only; 2 the wide wave tiles (NT = 4) where they fit; 3 64 x 128 (not 64 x 64) for under-filled launches

template <int KS>
int launch_h_ks(const svc_conv1d_h_args& a, int R, hipStream_t s) {
  // Measured per shape (profiles/r08b_conv_h_shapes.txt, one clip's MRF convs): the wave tiles with FOUR column tiles (128 x 256,
  // 64 x 512, 32 x 512: half the weight stream per FLOP, one workgroup per CU) lose to the ones with two on every shape — 17..41
  // against 17..34 us at 128 channels, 14..30 against 14..24 us at 64 — because a workgroup's phases (stage the tile, matrix loop,
  // store) only overlap with those of a co-resident workgroup, and the narrow forms keep two or three per CU.  They stay behind
  // the debug switch.  The 256-channel stage has 6 896 columns: 128 x 128 tiles are 108 workgroups for 256 CUs, 64 x 128 tiles 216.
  const long long cols = (long long)a.Tq * a.B;
  const bool wide = g_h_cfg == 2;
  if (R >= 128) {
    const long long wgs128 = (long long)svc::cdiv(R, 128) * svc::cdiv(a.Tq, 128) * a.B;
    const size_t lds_wide = (size_t)(a.Cin / 8) * (256 + (a.KS - 1) * a.dil) * 16;
    if (wide && wgs128 >= 320 && lds_wide <= 160 * 1024) return launch_h<KS, 2, 4, 2, 2>(a, R, s);   // 128 rows x 256 columns
    if (g_h_cfg == 3 && wgs128 < 200) return launch_h<KS, 2, 1, 1, 4>(a, R, s);                       //  64 rows x 128 columns
    // 64 x 64 (two workgroups per CU on the 256-channel stage) against 64 x 128: 11.3 / 17.3 / 22.5 against 13.5 / 18.9 / 23.9 us for
    // 3 / 7 / 11 taps, 23.4 against 30.1 at dilation 5 (profiles/r09d_conv_h_cfg.txt)
    if (g_h_cfg != 1 && wgs128 < 200) return launch_h<KS, 1, 1, 2, 2>(a, R, s);                       //  64 rows x  64 columns
    return launch_h<KS, 2, 2, 2, 2>(a, R, s);                                                         // 128 rows x 128 columns
  }
  if (R > 32) {
    if (wide && cols >= 160 * 512) return launch_h<KS, 2, 4, 1, 4>(a, R, s);                          //  64 rows x 512 columns
    return launch_h<KS, 2, 2, 1, 4>(a, R, s);                                                         //  64 rows x 256 columns
  }
  if (wide && cols >= 160 * 512) return launch_h<KS, 1, 4, 1, 4>(a, R, s);                            //  32 rows x 512 columns
  return launch_h<KS, 1, 2, 1, 4>(a, R, s);                                                           //  32 rows x 256 columns
}

// ---- fused ResBlock1 pair (vdecoder/hifigan/models.py:60-67): y = c2(lrelu(c1(lrelu(x)) + b1)) + b2 + x in ONE launch ----------
// From 64 channels down the single convolutions above move their algorithmic bytes at 2-3.6 TB/s: the pair's intermediate
// (written, then read back with its halo) is half of those bytes and half of the launches.  In fp16 both tiles fit LDS: a
// workgroup owns all C rows of N2 = N1P - (KS - 1) output columns; it stages lrelu(x) for N1P + (KS - 1) d1 columns, runs conv1
// over N1P intermediate columns (N1P = 128 or 256: whole MFMA column tiles; the KS - 1 extra ones are the second conv's halo,
// 4-8 % recompute), writes lrelu(. + b1) as fp16 into the second LDS tile in the blocked layout (zero outside the sequence:
// conv2's zero padding), runs conv2 from that tile, and finishes with the residual (raw x from global memory — its lines were
// read by this workgroup microseconds ago), the MRF accumulate / divide and one 8-byte store per 4 channels.  Same MFMA loop,
// weight ring and operand layouts as conv1d_h_kernel.  Built for C <= 128 (at 256 channels the 6 896-column stage would be 59
// workgroups).
struct PP {
  const void* x;
  const void* w1;
  const void* w2;
  const float* b1;
  const float* b2;
  void* y;
  int B, C, T, d1, RP;
  float slope, beta, out_div;
};

template <int KS, int MT, int NT, int CH>
__device__ __forceinline__ void mma_steps_h(f32x16 (&acc)[MT][NT], const h8* __restrict__ wp, long long sstride, int S,
                                            const h8* __restrict__ xw, int XW, int dil) {
  // wp: this lane's first A fragment (+ step * sstride + mt * 64); xw: this lane's B base (+ 2 g XW + tap dil + nt * 32)
  auto wload = [&](h8 (&af)[CH][MT], int s0) {
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[j][mt] = wp[min(s0 + j, S - 1) * sstride + mt * 64];
  };
  auto chunk = [&](const h8 (&af)[CH][MT], int s0) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int sidx = s0 + j;
      if (sidx < S) {
        const int g = sidx / KS, tap = sidx - g * KS;
        const h8* xr = xw + 2 * g * XW + tap * dil;
        h8 bq[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bq[nt] = xr[nt * 32];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][mt], bq[nt], acc[mt][nt], 0, 0, 0);
      }
    }
  };
  h8 a0[CH][MT], a1[CH][MT];
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
__global__ __launch_bounds__(256, 2) void respair_h_kernel(PP p) {
  constexpr int N1P = 32 * NT * WN;            // intermediate columns computed per workgroup
  constexpr int N2 = N1P - (KS - 1);           // output columns per workgroup
  constexpr int H2 = (KS - 1) / 2;
  static_assert(WM * WN == 4, "four waves per workgroup");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int wm = w / WN, wn = w - wm * WN;
  const int CB = p.C >> 3, G = p.C >> 4;
  const int h1 = H2 * p.d1;
  const int XW1 = N1P + 2 * h1, XW2 = N1P + (KS - 1);
  h8* xs = reinterpret_cast<h8*>(smem_h);      // [CB][XW1]  lrelu(x), tile column tl <-> global t0 - H2 - h1 + tl
  h8* ts = xs + CB * XW1;                      // [CB][XW2]  lrelu(c1 + b1), tile column j <-> global t0 - H2 + j
  const int t0 = blockIdx.x * N2, b = blockIdx.y;
  const h8* xg = reinterpret_cast<const h8*>(p.x) + (long long)b * CB * p.T;

  // ---- stage lrelu(x); zero the halo tail of the intermediate tile
  {
    const _Float16 ps = (_Float16)p.slope;
    const int total = CB * XW1;
    constexpr int LD = 8;
    for (int base = tid; base < total; base += 256 * LD) {
      h8 v[LD];
#pragma unroll
      for (int j = 0; j < LD; ++j) {
        const int idx = base + j * 256;
        h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        v[j] = z;
        if (idx < total) {
          const int cb = idx / XW1, tl = idx - cb * XW1;
          const int tin = t0 - H2 - h1 + tl;
          if (tin >= 0 && tin < p.T) v[j] = xg[(long long)cb * p.T + tin];
        }
      }
#pragma unroll
      for (int j = 0; j < LD; ++j) {
        const int idx = base + j * 256;
        if (idx < total) xs[idx] = lrelu8(v[j], ps);
      }
    }
    for (int idx = tid; idx < CB * (KS - 1); idx += 256) {
      const int cb = idx / (KS - 1), e = idx - cb * (KS - 1);
      h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      ts[cb * XW2 + N1P + e] = z;
    }
  }
  __syncthreads();

  const long long rowoff = ((long long)(wm * MT * 32 + li)) * 2 + kh;
  const long long sstride = (long long)p.RP * 2;
  const int col = wn * NT * 32 + li;
  // ---- conv1 over the N1P intermediate columns -> ts (fp16, blocked), zero outside [0, T)
  {
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    mma_steps_h<KS, MT, NT, 4>(acc, reinterpret_cast<const h8*>(p.w1) + rowoff, sstride, G * KS, xs + kh * XW1 + col, XW1, p.d1);
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
          h4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = svc_lrelu(acc[mt][nt][4 * i + e] + p.b1[row8 + 4 * kh + e], p.slope);
            o[e] = (_Float16)(inside ? v : 0.f);
          }
          *reinterpret_cast<h4*>(th + ((long long)((row8 >> 3) * XW2 + j)) * 8 + 4 * kh) = o;
        }
      }
  }
  __syncthreads();

  // ---- conv2 from the intermediate tile, residual, accumulate, store
  {
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    mma_steps_h<KS, MT, NT, 4>(acc, reinterpret_cast<const h8*>(p.w2) + rowoff, sstride, G * KS, ts + kh * XW2 + col, XW2, 1);
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
          const h4 rv = *reinterpret_cast<const h4*>(xb + off);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * i + e] + p.b2[row8 + 4 * kh + e] + (float)rv[e];
          if (p.beta != 0.f) {
            const h4 ov = *reinterpret_cast<const h4*>(yb + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(p.beta, (float)ov[e], v[e]);
          }
          if (p.out_div != 1.f) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] /= p.out_div;
          }
          h4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
          *reinterpret_cast<h4*>(yb + off) = o;
        }
      }
  }
}

template <int KS, int MT, int NT, int WM, int WN>
int launch_respair(const PP& p, hipStream_t s) {
  constexpr int N1P = 32 * NT * WN, N2 = N1P - (KS - 1);
  const int h1 = (KS - 1) / 2 * p.d1;
  const size_t lds = (size_t)(p.C / 8) * ((N1P + 2 * h1) + (N1P + KS - 1)) * 16;
  SVC_REQUIRE(lds <= 160 * 1024, "resblock_pair_h: tiles of %zu bytes do not fit LDS", lds);
  auto kern = respair_h_kernel<KS, MT, NT, WM, WN>;
  if (lds > 64 * 1024) {
    static bool done = false;
    if (!done) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      done = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3(svc::cdiv(p.T, N2), p.B), dim3(256), lds, s, p);
  return svc::check_launch("resblock_pair_h");
}

template <int KS>
int launch_respair_ks(const PP& p, hipStream_t s) {
  if (p.C > 64) return launch_respair<KS, 2, 2, 2, 2>(p, s);    // 128 rows, 128 intermediate columns
  if (p.C > 32) return launch_respair<KS, 2, 1, 1, 4>(p, s);    //  64 rows, 128
  return launch_respair<KS, 1, 2, 1, 4>(p, s);                  //  32 rows, 256
}

// ---- SnakeAlias on the blocked fp16 layout (vdecoder/hifiganwithsnake/alias/act.py:125-130; the fp32 form and its derivation:
// snake_alias.hip): UpSample1d x2 (replicate pad 5, 12-tap polyphase) -> SnakeBeta (log-scale alpha / beta) -> DownSample1d x2
// (replicate pad (5, 6), 12 taps), the 2x intermediate in LDS, fp32 arithmetic between the fp16 load and the fp16 store.  A
// workgroup owns SH_TILE time steps of ONE channel block (8 channels): a 16-byte word per time step in, one out.
constexpr int SH_TILE = 256;
struct TapsH {
  float f[12];
};
__global__ __launch_bounds__(256) void snake_alias_h_kernel(const h8* __restrict__ x, h8* __restrict__ y, const float* __restrict__ alpha,
                                                            const float* __restrict__ beta, TapsH taps, int CB, int T) {
  __shared__ float xs[(SH_TILE + 10) * 8];
  __shared__ float ua[(2 * SH_TILE + 12) * 8];
  const int t0 = blockIdx.x * SH_TILE, cb = blockIdx.y, b = blockIdx.z;
  const h8* xr = x + ((long long)b * CB + cb) * T;
  h8* yr = y + ((long long)b * CB + cb) * T;
  const int tid = threadIdx.x;
  for (int i = tid; i < SH_TILE + 10; i += 256) {
    int t = t0 - 5 + i;
    t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
    const h8 v = xr[t];
#pragma unroll
    for (int j = 0; j < 8; ++j) xs[i * 8 + j] = (float)v[j];
  }
  __syncthreads();
  const int n_lo = 2 * t0 - 5;
  // (position m, channel j) pairs: j fastest, so a wave's 64 lanes touch 8 consecutive positions x 8 channels (LDS rows of 32 B)
  for (int e = tid; e < (2 * SH_TILE + 10) * 8; e += 256) {
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
  for (int i = tid; i < SH_TILE; i += 256) {
    const int t = t0 + i;
    if (t >= T) break;
    h8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 12; ++k) acc = fmaf(taps.f[k], ua[(2 * i + k) * 8 + j], acc);
      o[j] = (_Float16)acc;
    }
    yr[t] = o;
  }
}

// ---- weight pack: dense fp32 (weight norm already folded) -> [Cin/16][tap][RP][16] fp16.
// conv (u == 1): w [Cout][Cin][KS], row = co, tap = k.   transposed (u > 1): w [Cin][Cout][K], row = ph * Cout + co, tap mr of
// M = ceil(K / u): k = ph + (M - 1 - mr) * u (taps time-reversed: each phase is a plain correlation, as pack_convt1d_kernel).
__global__ void pack_h_kernel(const float* __restrict__ w, _Float16* __restrict__ dst, int Cout, int Cin, int K, int taps, int RP,
                              int u, long long n) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int j = (int)(idx & 15);
  long long r = idx >> 4;
  const int row = (int)(r % RP);
  r /= RP;
  const int tap = (int)(r % taps), g = (int)(r / taps);        // [Cin/16][tap][RP][16]: (group, tap) steps linear, group-major
  const int ci = g * 16 + j;
  float v = 0.f;
  if (u <= 1) {
    if (row < Cout) v = w[((long long)row * Cin + ci) * K + tap];
  } else if (row < u * Cout) {
    const int ph = row / Cout, co = row - ph * Cout;
    const int k = ph + (taps - 1 - tap) * u;
    if (k < K) v = w[((long long)ci * Cout + co) * K + k];
  }
  dst[idx] = (_Float16)v;
}

// ---- fp32 [B,C,T] (strided) (+ a second fp32 tensor) -> blocked fp16, and back
__global__ void cvt_to_h_kernel(const float* __restrict__ x, const float* __restrict__ add, h8* __restrict__ y, long long x_bs,
                                long long x_cs, long long a_bs, long long a_cs, int B, int C, int T) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int CB = C >> 3;
  if (idx >= (long long)B * CB * T) return;
  const int t = (int)(idx % T);
  const long long r = idx / T;
  const int cb = (int)(r % CB), b = (int)(r / CB);
  h8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = x[(long long)b * x_bs + (long long)(cb * 8 + j) * x_cs + t];
    if (add) v += add[(long long)b * a_bs + (long long)(cb * 8 + j) * a_cs + t];
    o[j] = (_Float16)v;
  }
  y[idx] = o;
}

__global__ void cvt_from_h_kernel(const h8* __restrict__ x, float* __restrict__ y, int B, int C, int T) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int CB = C >> 3;
  if (idx >= (long long)B * CB * T) return;
  const int t = (int)(idx % T);
  const long long r = idx / T;
  const int cb = (int)(r % CB), b = (int)(r / CB);
  const h8 v = x[idx];
#pragma unroll
  for (int j = 0; j < 8; ++j) y[((long long)b * C + cb * 8 + j) * T + t] = (float)v[j];
}

// ---- conv_post (vdecoder/hifigan/models.py:390-392): leaky_relu(0.01) -> Conv1d(C, 1, KS) -> tanh, fp16 blocked in, fp32 out.
// One thread per output sample; fp32 arithmetic (the waveform itself is never rounded to 16 bits).
__global__ __launch_bounds__(256) void conv_post_h_kernel(const h8* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ y, int B, int C, int T, int KS, int pad, float pre_slope,
                                                          int act) {
  extern __shared__ float wsh[];       // [C][KS]
  for (int i = threadIdx.x; i < C * KS; i += blockDim.x) wsh[i] = w[i];
  __syncthreads();
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * T) return;
  const int t = (int)(idx % T), b = (int)(idx / T);
  const int CB = C >> 3;
  float acc = bias ? bias[0] : 0.f;
  for (int cb = 0; cb < CB; ++cb) {
    const h8* xr = x + ((long long)b * CB + cb) * T;
    for (int k = 0; k < KS; ++k) {
      const int tin = t - pad + k;
      if (tin < 0 || tin >= T) continue;
      const h8 v = xr[tin];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = fmaf(svc_lrelu((float)v[j], pre_slope), wsh[(cb * 8 + j) * KS + k], acc);
    }
  }
  y[idx] = act == SVC_ACT_TANH ? tanhf(acc) : acc;
}

}  // namespace

extern "C" int svc_pack_conv1d_h(const float* w, void* dst, int Cout, int Cin, int K, int u, int RP, void* stream) {
  SVC_REQUIRE(w && dst, "pack_conv1d_h: null tensor");
  SVC_REQUIRE(Cout > 0 && Cin > 0 && (Cin % 16) == 0 && K >= 1 && u >= 1, "pack_conv1d_h: bad shape (Cin must be a multiple of 16)");
  const int taps = u > 1 ? (K + u - 1) / u : K;
  const int R = u > 1 ? u * Cout : Cout;
  SVC_REQUIRE(RP >= R && (RP % 128) == 0, "pack_conv1d_h: RP must be a multiple of 128 >= the row count %d", R);
  const long long n = (long long)taps * (Cin / 16) * RP * 16;
  hipLaunchKernelGGL(pack_h_kernel, dim3((unsigned)svc::cdivll(n, 256)), dim3(256), 0, (hipStream_t)stream, w,
                     reinterpret_cast<_Float16*>(dst), Cout, Cin, K, taps, RP, u, n);
  return svc::check_launch("pack_conv1d_h");
}

extern "C" int svc_conv1d_h(const svc_conv1d_h_args* ap, void* stream) {
  SVC_REQUIRE(ap != nullptr, "conv1d_h: null args");
  const svc_conv1d_h_args& a = *ap;
  SVC_REQUIRE(a.x && a.w && a.y, "conv1d_h: null tensor");
  SVC_REQUIRE(a.B > 0 && a.Cin > 0 && a.Cout > 0 && a.Tin > 0 && a.Tq > 0 && a.Ty > 0, "conv1d_h: empty shape");
  SVC_REQUIRE((a.Cin % 16) == 0 && (a.Cout % 8) == 0, "conv1d_h: Cin must be a multiple of 16 and Cout of 8 (got %d, %d)", a.Cin, a.Cout);
  SVC_REQUIRE(a.dil >= 1 && a.u >= 1 && (a.RP % 128) == 0, "conv1d_h: bad dil / u / RP");
  SVC_REQUIRE(a.post_act == SVC_ACT_NONE || a.post_act == SVC_ACT_LRELU, "conv1d_h: post_act must be none or leaky_relu");
  SVC_REQUIRE(((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.w) | reinterpret_cast<uintptr_t>(a.y) |
                reinterpret_cast<uintptr_t>(a.res)) & 15) == 0, "conv1d_h: tensors must be 16-byte aligned");
  const int R = a.u > 1 ? a.u * a.Cout : a.Cout;
  SVC_REQUIRE(a.RP >= R, "conv1d_h: RP %d below the row count %d", a.RP, R);
  SVC_REQUIRE(a.u == 1 || (a.Cout % 8) == 0, "conv1d_h: transposed form needs Cout %% 8 == 0");
  hipStream_t s = (hipStream_t)stream;
  const double flop = 2.0 * a.B * (double)R * a.Cin * a.KS * a.Tq;
  const double bytes = 2.0 * a.B * ((double)a.Cin * a.Tin + (double)a.Cout * a.Ty * (a.res ? 2 : 1));
  char pname[96];
  if (svc::prof_on() && svc::prof_shapes())   // SVC_PROF_SHAPES=1: one profile row per shape (tuning aid)
    snprintf(pname, sizeof(pname), "%s[B%d,Ci%d,Co%d,K%d,d%d,T%d]", a.u > 1 ? "convt1d_h" : "conv1d_h", a.B, a.Cin, a.Cout, a.KS, a.dil, a.Tq);
  else
    snprintf(pname, sizeof(pname), "%s", a.u > 1 ? "convt1d_h" : "conv1d_h");
  svc::ProfScope prof(s, pname, flop, bytes);
  switch (a.KS) {
    case 1: return launch_h_ks<1>(a, R, s);
    case 2: return launch_h_ks<2>(a, R, s);
    case 3: return launch_h_ks<3>(a, R, s);
    case 5: return launch_h_ks<5>(a, R, s);
    case 7: return launch_h_ks<7>(a, R, s);
    case 11: return launch_h_ks<11>(a, R, s);
    default: SVC_REQUIRE(false, "conv1d_h: tap count %d not built (1, 2, 3, 5, 7, 11)", a.KS);
  }
  return SVC_OK;
}

extern "C" int svc_resblock_pair_h(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, int B,
                                  int C, int T, int KS, int dil1, int RP, float slope, float beta, float out_div, void* stream) {
  SVC_REQUIRE(x && w1 && w2 && b1 && b2 && y, "resblock_pair_h: null tensor");
  SVC_REQUIRE(B > 0 && T > 0 && C >= 16 && C <= 128 && (C % 16) == 0, "resblock_pair_h: C must be a multiple of 16 in 16..128 (got %d)", C);
  SVC_REQUIRE(dil1 >= 1 && (RP % 128) == 0 && RP >= C, "resblock_pair_h: bad dil1 / RP");
  SVC_REQUIRE(slope > 0.f && slope <= 1.f, "resblock_pair_h: leaky-ReLU slope in (0, 1]");
  SVC_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2) |
                reinterpret_cast<uintptr_t>(y)) & 15) == 0, "resblock_pair_h: tensors must be 16-byte aligned");
  PP p;
  p.x = x; p.w1 = w1; p.w2 = w2; p.b1 = b1; p.b2 = b2; p.y = y;
  p.B = B; p.C = C; p.T = T; p.d1 = dil1; p.RP = RP;
  p.slope = slope; p.beta = beta; p.out_div = out_div;
  hipStream_t s = (hipStream_t)stream;
  char pname[96];
  if (svc::prof_on() && svc::prof_shapes()) snprintf(pname, sizeof(pname), "resblock_pair_h[B%d,C%d,K%d,d%d,T%d]", B, C, KS, dil1, T);
  else snprintf(pname, sizeof(pname), "resblock_pair_h");
  svc::ProfScope prof(s, pname, 4.0 * B * (double)C * C * KS * T, 2.0 * B * (double)C * T * 3);
  switch (KS) {
    case 3: return launch_respair_ks<3>(p, s);
    case 7: return launch_respair_ks<7>(p, s);
    case 11: return launch_respair_ks<11>(p, s);
    default: SVC_REQUIRE(false, "resblock_pair_h: tap count %d not built (3, 7, 11)", KS);
  }
  return SVC_OK;
}

extern "C" int svc_snake_alias_h(const void* x, void* y, const float* alpha, const float* beta, const float* taps12, int B, int C, int T,
                                 void* stream) {
  SVC_REQUIRE(x && y && alpha && beta && taps12 && B > 0 && C > 0 && (C % 8) == 0 && T > 0, "snake_alias_h: bad args");
  TapsH tp;
  for (int i = 0; i < 12; ++i) tp.f[i] = taps12[i];
  svc::ProfScope prof((hipStream_t)stream, "snake_alias_h", 0.0, 4.0 * B * (double)C * T);
  hipLaunchKernelGGL(snake_alias_h_kernel, dim3(svc::cdiv(T, SH_TILE), C / 8, B), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const h8*>(x), reinterpret_cast<h8*>(y), alpha, beta, tp, C / 8, T);
  return svc::check_launch("snake_alias_h");
}

extern "C" int svc_debug_set_conv_h(int cfg) {
  g_h_cfg = cfg;
  return SVC_OK;
}

extern "C" int svc_cvt_to_h(const float* x, const float* add, void* y, long long x_bs, long long x_cs, long long add_bs,
                            long long add_cs, int B, int C, int T, void* stream) {
  SVC_REQUIRE(x && y && B > 0 && C > 0 && (C % 8) == 0 && T > 0, "cvt_to_h: bad args (C must be a multiple of 8)");
  const long long n = (long long)B * (C / 8) * T;
  hipLaunchKernelGGL(cvt_to_h_kernel, dim3((unsigned)svc::cdivll(n, 256)), dim3(256), 0, (hipStream_t)stream, x, add,
                     reinterpret_cast<h8*>(y), x_bs, x_cs, add_bs, add_cs, B, C, T);
  return svc::check_launch("cvt_to_h");
}

extern "C" int svc_cvt_from_h(const void* x, float* y, int B, int C, int T, void* stream) {
  SVC_REQUIRE(x && y && B > 0 && C > 0 && (C % 8) == 0 && T > 0, "cvt_from_h: bad args");
  const long long n = (long long)B * (C / 8) * T;
  hipLaunchKernelGGL(cvt_from_h_kernel, dim3((unsigned)svc::cdivll(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const h8*>(x), y, B, C, T);
  return svc::check_launch("cvt_from_h");
}

extern "C" int svc_conv_post_h(const void* x, const float* w, const float* bias, float* y, int B, int C, int T, int KS, int pad,
                               float pre_slope, int act, void* stream) {
  SVC_REQUIRE(x && w && y && B > 0 && C > 0 && (C % 8) == 0 && T > 0 && KS >= 1, "conv_post_h: bad args");
  SVC_REQUIRE((size_t)C * KS * 4 <= 48 * 1024, "conv_post_h: weight does not fit LDS");
  const long long n = (long long)B * T;
  svc::ProfScope prof((hipStream_t)stream, "conv_post_h", 2.0 * n * C * KS, 2.0 * n * C + 4.0 * n);
  hipLaunchKernelGGL(conv_post_h_kernel, dim3((unsigned)svc::cdivll(n, 256)), dim3(256), (size_t)C * KS * 4, (hipStream_t)stream,
                     reinterpret_cast<const h8*>(x), w, bias, y, B, C, T, KS, pad, pre_slope, act);
  return svc::check_launch("conv_post_h");
}
