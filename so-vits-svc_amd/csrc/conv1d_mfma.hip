// conv1d_mfma.hip — fused dense Conv1d on the gfx950 fp32 matrix pipe.
//
// Implicit GEMM, no im2col in memory:  out[co,t] = sum_{ci,k} W[co,ci,k] * X[ci, t + k*dil - pad]
//   M = co (MFMA rows), N = t (MFMA cols), K = (ci,k).
// A workgroup owns a BM(co) x BN(t) tile.  Per chunk of BC input channels it stages
//   Xs[BC][BN + (KS-1)*dil]  (time contiguous: coalesced HBM reads along t, pre-activation applied once)
//   Ws[BC][KS][BM]           (co contiguous, from the packed [Cin][KS][CoutP] weight)
// in LDS; every wave then issues v_mfma_f32_32x32x2_f32 (or 16x16x4 for Cout<=16) where the A operand
// is one Ws dword per lane and the B operand one Xs dword per lane (lanes contiguous in m resp. t, so
// both ds_read_b32 are bank-conflict free and a dilated tap is just an address offset k*dil).
// fp32 in, fp32 accumulate: bitwise an fmaf chain (MI355X_MICROARCH.md §Matrix cores), same 157.3 TF
// peak as the packed-fp32 VALU but ~32x fewer operand reads per FMA.
//
// Replaces (reference path:line): vdecoder/hifigan/models.py:41-67 (ResBlock1 convs + leaky_relu + residual),
// :335,:358,:373-374 (conv_pre + cond), modules/modules.py:110-138 (WN in_layers, gate, res_skip),
// modules/modules.py:288-307 (coupling pre/post), modules/attentions.py:198-205,337-345 (q,k,v,o,FFN),
// models.py:400,139 (pre, proj).
#include "common.h"

namespace {

struct ConvP {
  svc_conv1d_args a;
  int XW;        // LDS row width of the X tile (floats)
  int BC;        // input channels staged per chunk
  int n_t_tiles; // number of BN tiles along t
  int n_m_tiles;
};

template <bool M16>
__device__ __forceinline__ void mfma_step(float a, float b, f32x16& acc32, f32x4& acc16) {
  if constexpr (M16) {
    acc16 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc16, 0, 0, 0);
  } else {
    acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc32, 0, 0, 0);
  }
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case SVC_ACT_RELU: return v > 0.f ? v : 0.f;
    case SVC_ACT_TANH: return tanhf(v);
    case SVC_ACT_LRELU: return svc_lrelu(v, slope);
    default: return v;
  }
}

// MT x NT MFMA tiles per wave, WM x WN waves per workgroup.
template <int MT, int NT, int WM, int WN, bool M16, int EPI>
__global__ __launch_bounds__(WM* WN * 64, 2) void conv1d_mfma_kernel(ConvP p) {
  constexpr int TS = M16 ? 16 : 32;      // MFMA tile edge
  constexpr int KPI = M16 ? 4 : 2;       // K indices consumed per MFMA
  constexpr int NACC = M16 ? 4 : 16;     // accumulator regs per tile
  constexpr int BM = WM * MT * TS;
  constexpr int BN = WN * NT * TS;
  constexpr int NTHR = WM * WN * 64;
  const svc_conv1d_args& a = p.a;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int KS = a.KS, BC = p.BC, XW = p.XW;
  float* Ws = smem;                     // [BC][KS][BM]
  float* Xs = smem + BC * KS * BM;      // [BC][XW]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int ln = lane & (TS - 1);       // position inside tile row/col
  const int lk = lane / TS;             // which K index of the instruction this lane feeds

  int bid = blockIdx.x;
  const int tt = bid % p.n_t_tiles;
  bid /= p.n_t_tiles;
  const int mtile = bid % p.n_m_tiles;
  const int b = bid / p.n_m_tiles;
  const int t0 = tt * BN;
  const int co0 = mtile * BM;

  const float* xb = a.x + (long long)b * a.x_bs;
  const float* pm = a.premask ? a.premask + (long long)b * a.premask_bs : nullptr;
  const int tin0 = t0 - a.pad_left;
  const int w_rows_total = a.Cin * KS;

  // Polyphase transposed conv runs n_phase dense sub-convolutions over the same input tile (n_phase = 1 and
  // y_ts = 1 for an ordinary conv); each phase has its own packed weight block and writes outputs
  // t_out = t * y_ts + y_t0 + phase.
  for (int ph = 0; ph < a.n_phase; ++ph) {
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

  for (int c0 = 0; c0 < a.Cin; c0 += BC) {
    // ---- stage W chunk: rows (ci_l,k) of BM floats, float4 granularity ----
    {
      constexpr int BM4 = BM / 4;
      const int total4 = BC * KS * BM4;
      const int row0 = c0 * KS;
      for (int idx = tid; idx < total4; idx += NTHR) {
        const int r = idx / BM4;
        const int c4 = idx - r * BM4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int grow = row0 + r;
        const int gco = co0 + c4 * 4;
        if (grow < w_rows_total && gco < a.CoutP)
          v = *reinterpret_cast<const float4*>(wph + (long long)grow * a.CoutP + gco);
        *reinterpret_cast<float4*>(Ws + r * BM + c4 * 4) = v;
      }
    }
    // ---- stage X chunk: BC rows of XW floats, coalesced along t, pre-activation applied once ----
    {
      constexpr int NW = WM * WN;
      for (int r = wave; r < BC; r += NW) {
        const int ci = c0 + r;
        const float* xr = xb + (long long)ci * a.x_cs;
        float* dst = Xs + r * XW;
        const bool cok = ci < a.Cin;
        for (int c = lane; c < XW; c += 64) {
          const int tin = tin0 + c;
          float v = 0.f;
          if (cok && tin >= 0 && tin < a.Tin) {
            v = xr[tin];
            if (pm) v *= pm[tin];
            v = svc_lrelu(v, a.pre_slope);
          }
          dst[c] = v;
        }
      }
    }
    __syncthreads();

    // ---- MFMA over the chunk ----
    const float* wbase = Ws + wm * (MT * TS) + ln;
    const float* xbase = Xs + wn * (NT * TS) + ln;
    for (int k = 0; k < KS; ++k) {
      const int xoff = k * a.dil;
      for (int cc = 0; cc < BC; cc += KPI) {
        const int cl = cc + lk;
        float av[MT], bv[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) av[i] = wbase[(cl * KS + k) * BM + i * TS];
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[j] = xbase[cl * XW + xoff + j * TS];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) mfma_step<M16>(av[i], bv[j], acc32[i][j], acc16[i][j]);
      }
    }
    __syncthreads();
  }

  // ---- epilogue (one MFMA tile at a time; sched barriers keep the live set to one tile) ----
  const float* maskb = a.mask ? a.mask + (long long)b * a.mask_bs : nullptr;
  const float* condb = a.cond ? a.cond + (long long)b * a.cond_bs : nullptr;
  const float* resb = a.res ? a.res + (long long)b * a.res_bs : nullptr;
  float* yb = a.y + (long long)b * a.y_bs;
  const float* biasp = a.bias;
  const long long cond_cs = a.cond_cs, cond_ts = a.cond_ts;

#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int tq = t0 + wn * (NT * TS) + j * TS + ln;
    const int t = tq * a.y_ts + a.y_t0 + ph;
    const bool tok = tq < a.Tout && t >= 0 && t < a.y_len;
    const float mk = (maskb && tok) ? maskb[t] : 1.f;
    if constexpr (EPI == SVC_EPI_GATE) {
      static_assert(EPI != SVC_EPI_GATE || ((MT % 2) == 0 && !M16), "gate epilogue needs tile pairs");
      const int H = a.Cout >> 1;
#pragma unroll
      for (int i = 0; i < MT; i += 2) {
        const int prow = co0 + wm * (MT * TS) + i * TS;  // packed row of the tanh tile (multiple of 64)
        const int cbase = (prow >> 6) * 32 + 4 * lk;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = cbase + (r & 3) + 8 * (r >> 2);
          if (tok && c < H) {
            float vt = acc32[i][j][r];
            float vs = acc32[i + 1][j][r];
            if (biasp) {
              vt += biasp[c];
              vs += biasp[H + c];
            }
            if (condb) {
              vt += condb[c * cond_cs + t * cond_ts];
              vs += condb[(H + c) * cond_cs + t * cond_ts];
            }
            yb[(long long)c * a.y_cs + t] = tanhf(vt) * svc_sigmoid(vs);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int cobase = co0 + wm * (MT * TS) + i * TS + (M16 ? 4 : 4) * lk;
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
          const int co = cobase + (M16 ? r : (r & 3) + 8 * (r >> 2));
          if (tok && co < a.Cout) {
            float v;
            if constexpr (M16) v = acc16[i][j][r];
            else v = acc32[i][j][r];
            if (biasp) v += biasp[co];
            if (condb) v += condb[co * cond_cs + t * cond_ts];
            if constexpr (EPI == SVC_EPI_RES_SKIP) {
              if (co < a.skip_from) {
                float* yp = yb + (long long)co * a.y_cs + t;
                const float rv = resb[(long long)co * a.res_cs + t];
                *yp = (rv + v) * mk;
              } else {
                float* yp = a.y2 + (long long)b * a.y2_bs + (long long)(co - a.skip_from) * a.y2_cs + t;
                if (a.beta != 0.f) v += a.beta * (*yp);
                if (a.res_mode == 1) v *= mk;  // last WN layer: `output * x_mask` (modules/modules.py:138)
                *yp = v;
              }
            } else {
              v = apply_act(v, a.post_act, a.post_slope);
              v *= mk;
              if (a.res_mode == 1) v = v + resb[(long long)co * a.res_cs + t];
              else if (a.res_mode == 2) v = (resb[(long long)co * a.res_cs + t] - v) * mk;
              else if (a.res_mode == 3) v = v + resb[(long long)co * a.res_cs + t] * mk;
              float* yp = yb + (long long)co * a.y_cs + t;
              if (a.beta != 0.f) v += a.beta * (*yp);
              if (a.out_div != 1.f) v = v / a.out_div;
              *yp = v;
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  }  // phase loop
}

template <int MT, int NT, int WM, int WN, bool M16, int EPI = SVC_EPI_PLAIN>
int launch_cfg(const svc_conv1d_args& a, hipStream_t s) {
  constexpr int TS = M16 ? 16 : 32;
  constexpr int KPI = M16 ? 4 : 2;
  constexpr int BM = WM * MT * TS;
  constexpr int BN = WN * NT * TS;
  ConvP p;
  p.a = a;
  int xw = BN + (a.KS - 1) * a.dil;
  if (M16) {  // keep consecutive channel rows on disjoint bank halves for the 16-lane groups
    while ((xw & 31) != 16) ++xw;
  }
  p.XW = xw;
  // choose BC: as many channels per chunk as fit in ~48 KiB (bounded by Cin), multiple of KPI
  const int per_c = (a.KS * BM + xw) * 4;
  int bc = (48 * 1024) / per_c;
  bc = (bc / KPI) * KPI;
  if (bc < KPI) bc = KPI;
  if (bc > 32) bc = 32;
  const int cin_r = ((a.Cin + KPI - 1) / KPI) * KPI;
  if (bc > cin_r) bc = cin_r;
  p.BC = bc;
  const size_t lds = (size_t)bc * per_c;
  if (lds > 160 * 1024) {
    svc::set_error("conv1d: LDS tile too large (KS=%d dil=%d)", a.KS, a.dil);
    return SVC_ERR_UNSUPPORTED;
  }
  p.n_t_tiles = svc::cdiv(a.Tout, BN);
  p.n_m_tiles = svc::cdiv(a.Cout, BM);
  const long long nblk = (long long)p.n_t_tiles * p.n_m_tiles * a.B;
  auto kern = conv1d_mfma_kernel<MT, NT, WM, WN, M16, EPI>;
  if (lds > 64 * 1024) {
    static bool done = false;
    if (!done) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                          160 * 1024);
      done = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(WM * WN * 64), lds, s, p);
  return svc::check_launch("conv1d_mfma");
}

}  // namespace

static int conv1d_dispatch(const svc_conv1d_args& a, void* stream) {
  SVC_REQUIRE(a.x && a.w && a.y, "conv1d: null tensor");
  SVC_REQUIRE(a.B > 0 && a.Cin > 0 && a.Cout > 0 && a.Tin > 0 && a.Tout > 0, "conv1d: empty shape");
  SVC_REQUIRE(a.KS >= 1 && a.dil >= 1, "conv1d: bad KS/dil");
  SVC_REQUIRE(a.n_phase >= 1 && a.y_ts >= 1 && a.y_len >= 1, "conv1d: bad n_phase/y_ts/y_len");
  SVC_REQUIRE(a.CoutP >= a.Cout && (a.CoutP % 4) == 0, "conv1d: CoutP must be >= Cout and a multiple of 4");
  SVC_REQUIRE((reinterpret_cast<uintptr_t>(a.w) & 15) == 0, "conv1d: packed weight must be 16B aligned");
  SVC_REQUIRE(a.res_mode == 0 || a.res != nullptr, "conv1d: res_mode set but res is null");
  SVC_REQUIRE(a.res_mode != 2 || a.mask != nullptr || true, "conv1d");
  hipStream_t s = (hipStream_t)stream;
  const double flop = 2.0 * a.B * (double)a.Cout * a.Cin * a.KS * a.Tout * a.n_phase;
  const double bytes = 4.0 * a.B * ((double)a.Cin * a.Tin + (double)a.Cout * a.Tout) + 4.0 * a.Cin * a.KS * a.Cout;
  svc::ProfScope prof(s, a.n_phase > 1 ? "convt1d_mfma" : "conv1d_mfma", flop, bytes);

  const long long cols = (long long)a.B * a.Tout;
  if (a.epi == SVC_EPI_GATE) {
    SVC_REQUIRE((a.Cout % 64) == 0, "conv1d: gate epilogue needs Cout %% 64 == 0 (got %d)", a.Cout);
    SVC_REQUIRE(a.res_mode == 0, "conv1d: gate epilogue takes no residual");
    if (cols >= 16384) return launch_cfg<2, 2, 2, 2, false, SVC_EPI_GATE>(a, s);
    return launch_cfg<2, 1, 1, 4, false, SVC_EPI_GATE>(a, s);
  }
  if (a.epi == SVC_EPI_RES_SKIP) {
    SVC_REQUIRE(a.res && a.y2, "conv1d: res_skip epilogue needs res and y2");
    if (cols >= 16384) return launch_cfg<2, 2, 2, 2, false, SVC_EPI_RES_SKIP>(a, s);
    return launch_cfg<2, 1, 1, 4, false, SVC_EPI_RES_SKIP>(a, s);
  }
  SVC_REQUIRE(a.epi == SVC_EPI_PLAIN, "conv1d: unknown epilogue %d", a.epi);
  if (a.Cout <= 16) return launch_cfg<1, 8, 1, 4, true>(a, s);             // 16 x 512
  if (a.Cout <= 32) return launch_cfg<1, 4, 1, 4, false>(a, s);            // 32 x 512
  if (cols < 16384) {                                                      // short sequences: small tiles
    if (a.Cout <= 64 || (a.Cout % 64) != 0) return launch_cfg<1, 1, 1, 4, false>(a, s);  // 32 x 128
    return launch_cfg<2, 1, 1, 4, false>(a, s);                            // 64 x 128
  }
  if (a.Cout <= 64) return launch_cfg<2, 2, 1, 4, false>(a, s);            // 64 x 256
  return launch_cfg<2, 2, 2, 2, false>(a, s);                              // 128 x 128
}

extern "C" int svc_conv1d_f32(const svc_conv1d_args* ap, void* stream) {
  SVC_REQUIRE(ap != nullptr, "conv1d: null args");
  return conv1d_dispatch(*ap, stream);
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
  a.x_bs = t.x_bs; a.x_cs = t.x_cs; a.y_bs = t.y_bs; a.y_cs = t.y_cs; a.res_bs = t.res_bs; a.res_cs = t.res_cs;
  a.B = t.B; a.Cin = t.Cin; a.Cout = t.Cout; a.Tin = t.Tin;
  a.Tout = (Lout - 1 + t.padding) / u + 1;   // number of q positions
  a.KS = M; a.dil = 1; a.pad_left = M - 1; a.CoutP = t.CoutP;
  a.epi = SVC_EPI_PLAIN; a.post_act = SVC_ACT_NONE; a.res_mode = t.res ? 1 : 0;
  a.n_phase = u; a.y_ts = u; a.y_t0 = -t.padding; a.y_len = Lout;
  a.w_phase_stride = (long long)t.Cin * M * t.CoutP;
  a.pre_slope = t.pre_slope; a.post_slope = 0.f; a.beta = 0.f; a.out_div = 1.f;
  return conv1d_dispatch(a, stream);
}
