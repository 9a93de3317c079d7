// conv1d_direct.hip — VALU direct convolution for the layers that are NOT GEMM-shaped:
//   * noise_convs[i]: Conv1d(1 -> C_i, k = 2s, stride s) of har_source (vdecoder/hifigan/models.py:343-348,379)
//   * conv_post:      Conv1d(16 -> 1, k7) with leaky_relu(0.01) in front and tanh behind (:355,390-392)
//   * f0_prenet (1 -> 192, k3), F0Decoder.proj (192 -> 1) (models.py:324-326)
// One thread per output sample, COT output channels per thread; the input window is staged in LDS in
// polyphase order (index i -> row i % stride) so a strided read is bank-conflict free; weights are wave-uniform
// and come through the scalar cache.  HBM-bound by design (<= 0.2 % of the path's FLOPs).
#include "common.h"
#include <type_traits>

namespace {

struct DirectP {
  svc_conv1d_direct_args a;
  int win;   // input samples staged per channel
  int Qp;    // polyphase row pitch (odd)
  int bci;   // channels per LDS chunk
  int wlds;  // 1: the chunk's weights [bci][KS][COT] are staged in LDS behind the input window (else read through the scalar cache)
};

constexpr int DT = 256;  // outputs (threads) per block

template <int COT>
__global__ __launch_bounds__(DT) void conv1d_direct_kernel(DirectP p) {
  const svc_conv1d_direct_args& a = p.a;
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [bci][stride][Qp]
  const int tid = threadIdx.x;
  const int t0 = blockIdx.x * DT;
  const int co0 = blockIdx.y * COT;
  const int b = blockIdx.z;
  const int s = a.stride;
  const int rowsz = s * p.Qp;
  const int in0 = t0 * s - a.pad_left;
  const float* xb = a.x + (long long)b * a.x_bs;

  float acc[COT];
#pragma unroll
  for (int i = 0; i < COT; ++i) acc[i] = 0.f;

  // polyphase position (i % s, i / s) of staged sample i and (k*dil % s, k*dil / s) of tap k, advanced incrementally: the
  // integer divisions by the run-time stride cost two per staged sample and two per tap and channel (~30 VALU instructions each
  // against ONE fma of useful work per tap).  conv_post (16 -> 1, 7 taps, a 24 MB read) step by step: 59 us -> 53 us without
  // the divisions -> 46 us with the loads of a staging group issued together -> 38 us with its weights in LDS
  // (profiles/r05m_*, r06f_*, r07a_*, r07b_infer_T862_kernel_stats_serialised.txt)
  const int dq = DT / s, dr = DT - dq * s;
  const int q0 = tid / s, r0 = tid - q0 * s;
  for (int c0 = 0; c0 < a.Cin; c0 += p.bci) {
    const int nc = min(p.bci, a.Cin - c0);
    // staging with the global loads of a group issued together (RU rows x SU column slots per pass): one row and one slot at a
    // time every load waited for the previous one's LDS write — 16 exposed round trips per block for conv_post's 16 channels
    auto stage = [&](auto ru_tag, auto su_tag) {
      constexpr int RU = decltype(ru_tag)::value, SU = decltype(su_tag)::value;
      for (int rb = 0; rb < nc; rb += RU) {
        int iq = q0, ir = r0;
        for (int i = tid; i < p.win; i += SU * DT) {
          float v[RU][SU];
          int pos[SU];
#pragma unroll
          for (int j = 0; j < SU; ++j) {
            const int ii = i + j * DT;
            const int ti = in0 + ii;
            const bool ok = ii < p.win && ti >= 0 && ti < a.Tin;
            pos[j] = ii < p.win ? ir * p.Qp + iq : -1;
#pragma unroll
            for (int u = 0; u < RU; ++u) {
              const float* xr = xb + (long long)(c0 + min(rb + u, nc - 1)) * a.x_cs;
              v[u][j] = ok ? xr[ti] : 0.f;
            }
            iq += dq;
            ir += dr;
            if (ir >= s) {
              ir -= s;
              ++iq;
            }
          }
#pragma unroll
          for (int j = 0; j < SU; ++j)
#pragma unroll
            for (int u = 0; u < RU; ++u)
              if (pos[j] >= 0 && rb + u < nc) xs[(rb + u) * rowsz + pos[j]] = svc_lrelu(v[u][j], a.pre_slope);
        }
      }
    };
    if (nc == 1) stage(std::integral_constant<int, 1>{}, std::integral_constant<int, 8>{});
    else stage(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{});
    // the chunk's weights into LDS: read wave-uniformly through the scalar cache inside the tap loop, every tap waited for its own
    // s_load (a run-time trip count: nothing to batch) — conv_post's 16 x 7 taps = 112 exposed scalar-load latencies per block
    float* ws = xs + p.bci * rowsz;
    if (p.wlds) {
      const int nw = nc * a.KS * COT;
      for (int i = tid; i < nw; i += DT) {
        const int rk = i / COT, c = i - rk * COT;
        ws[i] = a.w[(long long)c0 * a.KS * a.CoutP + (long long)rk * a.CoutP + co0 + c];
      }
    }
    __syncthreads();
    for (int r = 0; r < nc; ++r) {
      const float* wr = a.w + (long long)(c0 + r) * a.KS * a.CoutP + co0;
      const float* xr = xs + r * rowsz + tid;
      int kr = 0, kq = 0;       // (k*dil) % s, (k*dil) / s
      const float* wl = ws + r * a.KS * COT;
      for (int k = 0; k < a.KS; ++k) {
        const float xv = xr[kr * p.Qp + kq];
        const float* wk = p.wlds ? wl + k * COT : wr + (long long)k * a.CoutP;
#pragma unroll
        for (int i = 0; i < COT; ++i) acc[i] = fmaf(wk[i], xv, acc[i]);
        kr += a.dil;
        while (kr >= s) {
          kr -= s;
          ++kq;
        }
      }
    }
    __syncthreads();
  }

  const int t = t0 + tid;
  if (t >= a.Tout) return;
  const float mk = a.mask ? a.mask[(long long)b * a.mask_bs + t] : 1.f;
#pragma unroll
  for (int i = 0; i < COT; ++i) {
    const int co = co0 + i;
    if (co < a.Cout) {
      float v = acc[i];
      if (a.bias) v += a.bias[co];
      if (a.post_act == SVC_ACT_RELU) v = v > 0.f ? v : 0.f;
      else if (a.post_act == SVC_ACT_TANH) v = tanhf(v);
      else if (a.post_act == SVC_ACT_LRELU) v = svc_lrelu(v, a.post_slope);
      else if (a.post_act == SVC_ACT_GELU) v = svc_gelu(v);
      v *= mk;
      if (a.res) v += a.res[(long long)b * a.res_bs + (long long)co * a.res_cs + t];
      a.y[(long long)b * a.y_bs + (long long)co * a.y_cs + t] = v;
    }
  }
}

template <int COT>
int launch(DirectP p, hipStream_t s) {
  // (16 output channels per thread read their 16 weights of a tap with ONE s_load_dwordx16: the scalar path is the better one
  //  there — noise_convs 23.6 us against 26.1 us with the slab, profiles/r07a_* / r07b_*)
  p.wlds = (COT < 16 && (size_t)p.bci * p.a.KS * COT * 4 <= 48 * 1024) ? 1 : 0;
  const svc_conv1d_direct_args& a = p.a;
  size_t lds = (size_t)p.bci * a.stride * p.Qp * 4;
  if (p.wlds) lds += (size_t)p.bci * a.KS * COT * 4;
  auto kern = conv1d_direct_kernel<COT>;
  if (lds > 64 * 1024) {
    static bool done = false;
    if (!done) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      done = true;
    }
  }
  dim3 grid(svc::cdiv(a.Tout, DT), svc::cdiv(a.Cout, COT), a.B);
  hipLaunchKernelGGL(kern, grid, dim3(DT), lds, s, p);
  return svc::check_launch("conv1d_direct");
}

}  // namespace

extern "C" int svc_conv1d_direct_f32(const svc_conv1d_direct_args* ap, void* stream) {
  SVC_REQUIRE(ap != nullptr, "conv1d_direct: null args");
  const svc_conv1d_direct_args& a = *ap;
  SVC_REQUIRE(a.x && a.w && a.y, "conv1d_direct: null tensor");
  SVC_REQUIRE(a.B > 0 && a.Cin > 0 && a.Cout > 0 && a.Tin > 0 && a.Tout > 0, "conv1d_direct: empty shape");
  SVC_REQUIRE(a.KS >= 1 && a.dil >= 1 && a.stride >= 1, "conv1d_direct: bad KS/dil/stride");
  SVC_REQUIRE(a.CoutP >= a.Cout, "conv1d_direct: CoutP < Cout");
  DirectP p;
  p.a = a;
  p.win = (DT - 1) * a.stride + (a.KS - 1) * a.dil + 1;
  int q = svc::cdiv(p.win, a.stride);
  if ((q & 1) == 0) ++q;
  p.Qp = q;
  const size_t per_c = (size_t)a.stride * q * 4;
  int bci = (int)((96 * 1024) / per_c);
  if (bci < 1) {
    svc::set_error("conv1d_direct: window too large for LDS (KS=%d stride=%d)", a.KS, a.stride);
    return SVC_ERR_UNSUPPORTED;
  }
  if (bci > a.Cin) bci = a.Cin;
  p.bci = bci;
  hipStream_t s = (hipStream_t)stream;
  const double flop = 2.0 * a.B * (double)a.Cout * a.Cin * a.KS * a.Tout;
  const double bytes = 4.0 * a.B * ((double)a.Cin * a.Tin + (double)a.Cout * a.Tout * (a.res ? 2 : 1));
  svc::ProfScope prof(s, "conv1d_direct", flop, bytes);
  // packed weights are padded to a multiple of 32 output channels, so COT-wide reads never run off the row
  if (a.Cout >= 16 && a.CoutP % 16 == 0) return launch<16>(p, s);
  if (a.Cout >= 4 && a.CoutP % 4 == 0) return launch<4>(p, s);
  return launch<1>(p, s);
}
