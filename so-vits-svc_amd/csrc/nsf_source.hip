// nsf_source.hip — harmonic-plus-noise excitation of the NSF-HiFiGAN generator.
// Reference: SineGen._f02sine / SineGen.forward / SourceModuleHnNSF.forward,
// vdecoder/hifigan/models.py:138-166, 250-271, 307-320, fed by the nearest x`upp` upsample of f0 (:369).
//
// The reference computes, per harmonic h and sample t (all fp32 tensors; torch's CPU cumsum accumulates in
// double and stores float):
//     rad(t)   = (f0(t)*h / sr) % 1 ;  rad(0) += rand_ini[h]
//     wrap(t)  = frac(float(S1(t))) < frac(float(S1(t-1))),   S1 = cumsum(rad)
//     phase(t) = float( cumsum( float(rad(t) - wrap(t)) ) )          (range reduction, :160-166)
//     sine     = sin(phase * 2 * pi) * 0.1
// f0 is piecewise constant over `upp` samples, so instead of a 441k-step scan we evaluate this in CLOSED FORM:
//     S1(t)    = A_f + (n+1) * rad_f                  (frame f, sample n; A_f = prefix over frames, double)
//     N_w(t)   = floor(float(S1(t))) - floor(float(S1(0)))        (float rounding is monotone, so a wrap is
//                                                                  detected exactly when the integer part steps)
//     phase(t) = S1(t) - N_w(t) + E_f + (N_w(t) - W_f) * eps_f,   eps_f = float(rad_f - 1) - (rad_f - 1)
// where eps_f is the rounding error the reference commits each time it subtracts 1 in fp32 — carrying it (E) is
// what makes this agree with the reference to fp32 round-off instead of the 1e-4..1e-2 drift of an exact phase
// (SURVEY.md §8a a18).  A tiny sequential per-(batch,harmonic) frame scan produces (A_f, W_f, E_f); the
// sample kernel is then embarrassingly parallel and HBM-bound on the 9-wide noise tensor.
#include "common.h"

namespace {

constexpr int MAXH = 16;

struct ScanRec {
  double A;  // S1 before the first sample of the frame
  double E;  // accumulated fp32 "-1" rounding error before the frame
  double W;  // number of wraps detected before the frame
};

// one workgroup per batch item.  The recurrence over frames is sequential per harmonic, but only through ONE double-precision
// fma (A += upp * rad): everything else of a step — the fp32 `(f0 * h / sr) % 1` (a multiply, a divide and an fmodf), the wrap
// count and the carried rounding error — hangs off it.  So the 256 threads first fill LDS with rad for a chunk of frames and
// all harmonics (the expensive, fully parallel part), then thread h walks its row four frames at a time (the LDS reads of a
// group are issued together).  (With rad computed inside the loop a step cost ~400 cycles: 159 us for the 862 frames of a 10 s
// clip, 2 % of the synthesis — profiles/r05m_infer_T862_kernel_stats_serialised.txt.)  Same arithmetic per step as before.
constexpr int SCAN_CHUNK = 512;   // frames staged per pass
constexpr int SCAN_PITCH = SCAN_CHUNK + 1;
__global__ __launch_bounds__(256) void nsf_frame_scan_kernel(const float* __restrict__ f0, const float* __restrict__ rand_ini,
                                                             ScanRec* __restrict__ rec, int B, int T, int upp, int H,
                                                             float sr) {
  __shared__ float rads[MAXH * SCAN_PITCH];
  const int b = blockIdx.x, h = threadIdx.x;
  const float ri = (h == 0 || h >= H) ? 0.f : rand_ini[b * H + h];
  double A = 0.0, E = 0.0, W = 0.0;
  double base = 0.0;  // floor(float(S1(0)))
  const double dupp = (double)upp;
  for (int c0 = 0; c0 < T; c0 += SCAN_CHUNK) {
    const int nc = min(SCAN_CHUNK, T - c0);
    __syncthreads();
    for (int idx = threadIdx.x; idx < H * nc; idx += 256) {
      const int hh = idx / nc, i = idx - hh * nc;
      const float fn = f0[(long long)b * T + c0 + i] * (float)(hh + 1);
      rads[hh * SCAN_PITCH + i] = fmodf(fn / sr, 1.0f);
    }
    __syncthreads();
    if (h < H) {
      const float* rp = rads + h * SCAN_PITCH;
      ScanRec* rr = rec + ((long long)b * H + h) * T + c0;
      auto step = [&](int i, float rad) {
        ScanRec r;
        r.A = A;
        r.E = E;
        r.W = W;
        rr[i] = r;
        double s_end;
        if (c0 + i == 0) {
          const float rad0 = rad + ri;  // fp32 add, models.py:149
          base = floor((double)(float)(double)rad0);
          s_end = (double)rad0 + (double)(upp - 1) * (double)rad;
        } else {
          s_end = A + dupp * (double)rad;
        }
        const double w_end = floor((double)(float)s_end) - base;
        const double eps = (double)(rad - 1.0f) - ((double)rad - 1.0);
        E += (w_end - W) * eps;
        W = w_end;
        A = s_end;
      };
      int i = 0;
      for (; i + 4 <= nc; i += 4) {
        const float r0 = rp[i], r1 = rp[i + 1], r2 = rp[i + 2], r3 = rp[i + 3];
        step(i, r0);
        step(i + 1, r1);
        step(i + 2, r2);
        step(i + 3, r3);
      }
      for (; i < nc; ++i) step(i, rp[i]);
    }
  }
}

__global__ __launch_bounds__(256) void nsf_sample_kernel(const float* __restrict__ f0, const float* __restrict__ rand_ini,
                                                         const float* __restrict__ noise, const float* __restrict__ lin_w,
                                                         const float* __restrict__ lin_b, const ScanRec* __restrict__ rec,
                                                         float* __restrict__ har, float* __restrict__ waves, int B, int T,
                                                         int upp, int H, float sr, float sine_amp, float noise_std) {
  const long long L = (long long)T * upp;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= L) return;
  const int f = (int)(t / upp);
  const int n = (int)(t - (long long)f * upp);
  const float f0v = f0[(long long)b * T + f];
  const float uv = f0v > 0.f ? 1.f : 0.f;
  const float noise_amp = uv * noise_std + (1.f - uv) * sine_amp / 3.f;
  const float* nz = noise + ((long long)b * L + t) * H;
  const float pi_f = 3.14159265358979323846f;
  float acc = 0.f;
  for (int h = 0; h < H; ++h) {
    const float hm = (float)(h + 1);
    const float fn = f0v * hm;
    const float rad = fmodf(fn / sr, 1.0f);
    const ScanRec r = rec[((long long)b * H + h) * T + f];
    // S1(0) for the wrap base
    const float f00 = f0[(long long)b * T] * hm;
    const float rad00 = fmodf(f00 / sr, 1.0f) + (h == 0 ? 0.f : rand_ini[b * H + h]);
    const double base = floor((double)rad00);
    double s1;
    if (f == 0) s1 = (double)rad00 + (double)n * (double)rad;
    else s1 = r.A + (double)(n + 1) * (double)rad;
    const double nw = floor((double)(float)s1) - base;
    const double eps = (double)(rad - 1.0f) - ((double)rad - 1.0);
    const double ph = s1 - nw + r.E + (nw - r.W) * eps;
    const float ph32 = (float)ph;
    const float sine = sinf(ph32 * 2.f * pi_f) * sine_amp;
    const float sw = sine * uv + noise_amp * nz[h];
    if (waves) waves[((long long)b * L + t) * H + h] = sw;
    acc += lin_w[h] * sw;
  }
  har[(long long)b * L + t] = tanhf(acc + lin_b[0]);
}

// ---- vdecoder/nsf_hifigan variant (SineGen.forward, vdecoder/nsf_hifigan/models.py:136-181) ----------------------------
// That generator integrates the phase in DOUBLE at the sample rate: sin(2 pi cumsum(rad_up + shift)), rad_up = the
// frame-rate rad (fp32 `(f0*h/sr) % 1`, rand_ini added to FRAME 0, i.e. to each of its upp samples) nearest-upsampled,
// shift = -1 at detected wraps (integers: they only keep the argument small).  So sine(t) = sin(2 pi frac(A_f + (n+1) rad_f))
// with A_f the double prefix over frames — no fp32 error to carry.  Scratch: double A[B][H][T].
__global__ __launch_bounds__(64) void nsf_frame_scan_exact_kernel(const float* __restrict__ f0, const float* __restrict__ rand_ini,
                                                                  double* __restrict__ A, int T, int upp, int H, float sr) {
  __shared__ float f0s[SCAN_CHUNK];
  const int b = blockIdx.x, h = threadIdx.x;
  const float hm = (float)(h + 1);
  const float ri = (h == 0 || h >= H) ? 0.f : rand_ini[b * H + h];
  double acc = 0.0;
  for (int c0 = 0; c0 < T; c0 += SCAN_CHUNK) {
    const int nc = min(SCAN_CHUNK, T - c0);
    __syncthreads();
    for (int i = threadIdx.x; i < nc; i += 64) f0s[i] = f0[(long long)b * T + c0 + i];
    __syncthreads();
    if (h < H) {
      for (int i = 0; i < nc; ++i) {
        const int f = c0 + i;
        float rad = fmodf(f0s[i] * hm / sr, 1.0f);
        if (f == 0) rad = rad + ri;
        A[((long long)b * H + h) * T + f] = acc;
        acc += (double)upp * (double)rad;
      }
    }
  }
}

__global__ __launch_bounds__(256) void nsf_sample_exact_kernel(const float* __restrict__ f0, const float* __restrict__ rand_ini,
                                                               const float* __restrict__ noise, const float* __restrict__ lin_w,
                                                               const float* __restrict__ lin_b, const double* __restrict__ A,
                                                               float* __restrict__ har, int T, int upp, int H, float sr,
                                                               float sine_amp, float noise_std) {
  const long long L = (long long)T * upp;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= L) return;
  const int f = (int)(t / upp);
  const int n = (int)(t - (long long)f * upp);
  const float f0v = f0[(long long)b * T + f];
  const float uv = f0v > 0.f ? 1.f : 0.f;
  const float noise_amp = uv * noise_std + (1.f - uv) * sine_amp / 3.f;
  const float* nz = noise + ((long long)b * L + t) * H;
  float acc = 0.f;
  for (int h = 0; h < H; ++h) {
    float rad = fmodf(f0v * (float)(h + 1) / sr, 1.0f);
    if (f == 0 && h > 0) rad = rad + rand_ini[b * H + h];
    const double ph = A[((long long)b * H + h) * T + f] + (double)(n + 1) * (double)rad;
    const double fr = ph - floor(ph);
    const float sine = (float)sin(fr * 6.283185307179586476925287) * sine_amp;
    acc += lin_w[h] * (sine * uv + noise_amp * nz[h]);
  }
  har[(long long)b * L + t] = tanhf(acc + lin_b[0]);
}

}  // namespace

extern "C" long long svc_nsf_source_scratch_bytes(int B, int T, int H) {
  return (long long)B * H * T * (long long)sizeof(ScanRec);
}

static int nsf_source_impl(const float* f0, const float* rand_ini, const float* noise, const float* lin_w,
                           const float* lin_b, float* har, float* waves, void* scratch, int B, int T, int upp, int H,
                           float sampling_rate, float sine_amp, float noise_std, void* stream) {
  SVC_REQUIRE(f0 && rand_ini && noise && lin_w && lin_b && har && scratch, "nsf_source: null tensor");
  SVC_REQUIRE(B > 0 && T > 0 && upp > 0 && H > 0 && H <= MAXH, "nsf_source: bad shape (H <= %d)", MAXH);
  SVC_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 7) == 0, "nsf_source: scratch must be 8-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  ScanRec* rec = reinterpret_cast<ScanRec*>(scratch);
  const long long L = (long long)T * upp;
  svc::ProfScope prof(s, "nsf_source", 0.0, 4.0 * B * L * (H + 1));
  hipLaunchKernelGGL(nsf_frame_scan_kernel, dim3(B), dim3(256), 0, s, f0, rand_ini, rec, B, T, upp, H,
                     sampling_rate);
  hipLaunchKernelGGL(nsf_sample_kernel, dim3((unsigned)svc::cdivll(L, 256), B), dim3(256), 0, s, f0, rand_ini, noise,
                     lin_w, lin_b, rec, har, waves, B, T, upp, H, sampling_rate, sine_amp, noise_std);
  return svc::check_launch("nsf_source");
}

extern "C" int svc_nsf_source_f32(const float* f0, const float* rand_ini, const float* noise, const float* lin_w,
                                  const float* lin_b, float* har, void* scratch, int B, int T, int upp, int H,
                                  float sampling_rate, float sine_amp, float noise_std, void* stream) {
  return nsf_source_impl(f0, rand_ini, noise, lin_w, lin_b, har, nullptr, scratch, B, T, upp, H, sampling_rate, sine_amp,
                         noise_std, stream);
}

// Training variant: additionally writes the per-harmonic excitation waves [B, T*upp, H] that feed l_linear
// (vdecoder/hifigan/models.py:318), so that its weight gradient can be formed (svc_nsf_linear_bwd_f32).
extern "C" int svc_nsf_source_train_f32(const float* f0, const float* rand_ini, const float* noise, const float* lin_w,
                                        const float* lin_b, float* har, float* waves, void* scratch, int B, int T, int upp,
                                        int H, float sampling_rate, float sine_amp, float noise_std, void* stream) {
  SVC_REQUIRE(waves != nullptr, "nsf_source_train: null waves");
  return nsf_source_impl(f0, rand_ini, noise, lin_w, lin_b, har, waves, scratch, B, T, upp, H, sampling_rate, sine_amp,
                         noise_std, stream);
}

// vdecoder/nsf_hifigan SineGen + SourceModuleHnNSF (vdecoder/nsf_hifigan/models.py:136-218): double-precision phase.
// scratch: B*H*T doubles (8-byte aligned).
extern "C" int svc_nsf_source_exact_f32(const float* f0, const float* rand_ini, const float* noise, const float* lin_w,
                                        const float* lin_b, float* har, void* scratch, int B, int T, int upp, int H,
                                        float sampling_rate, float sine_amp, float noise_std, void* stream) {
  SVC_REQUIRE(f0 && rand_ini && noise && lin_w && lin_b && har && scratch, "nsf_source_exact: null tensor");
  SVC_REQUIRE(B > 0 && T > 0 && upp > 0 && H > 0 && H <= MAXH, "nsf_source_exact: bad shape (H <= %d)", MAXH);
  SVC_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 7) == 0, "nsf_source_exact: scratch must be 8-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  double* A = reinterpret_cast<double*>(scratch);
  const long long L = (long long)T * upp;
  svc::ProfScope prof(s, "nsf_source", 0.0, 4.0 * B * L * (H + 1));
  hipLaunchKernelGGL(nsf_frame_scan_exact_kernel, dim3(B), dim3(64), 0, s, f0, rand_ini, A, T, upp, H, sampling_rate);
  hipLaunchKernelGGL(nsf_sample_exact_kernel, dim3((unsigned)svc::cdivll(L, 256), B), dim3(256), 0, s, f0, rand_ini, noise,
                     lin_w, lin_b, A, har, T, upp, H, sampling_rate, sine_amp, noise_std);
  return svc::check_launch("nsf_source_exact");
}
