// resample.hip — polyphase windowed-sinc sample-rate conversion (the step in front of the unit encoder:
// inference/infer_tool.py:219-222 `torchaudio.transforms.Resample(target_sample, 16000)`, and :271-274 for inputs whose
// rate differs from the model's).  torchaudio realises it as a strided conv1d with `new` output channels followed by a
// transpose/reshape; here one thread owns one OUTPUT sample n = f*new + j and accumulates
//     y[n] = sum_k kern[k][j] * x[f*orig + k - width]          (zero outside the signal)
// so neighbouring lanes (j, j+1, ...) read neighbouring kernel columns (coalesced, L1/L2 resident: 4*K*new bytes) and
// the same handful of x cache lines.  HBM-bound: 4 B in per orig/new outputs + 4 B out per sample; K FMAs per output
// (K = 475 for 44.1 kHz -> 16 kHz) is ~0.08 GFLOP per 10 s clip.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void resample_sinc_kernel(const float* __restrict__ x, const float* __restrict__ kern,
                                                            float* __restrict__ y, long long x_bs, long long y_bs, int Lin,
                                                            int Lout, int orig, int nw, int K, int width) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (n >= Lout) return;
  const int f = n / nw, j = n - f * nw;
  const float* xb = x + (long long)b * x_bs;
  const int base = f * orig - width;
  const int k0 = max(0, -base), k1 = min(K, Lin - base);
  float acc = 0.f;
  for (int k = k0; k < k1; ++k) acc = fmaf(kern[(long long)k * nw + j], xb[base + k], acc);
  y[(long long)b * y_bs + n] = acc;
}

}  // namespace

extern "C" int svc_resample_sinc_f32(const float* x, const float* kern, float* y, long long x_bs, long long y_bs, int B, int Lin,
                                     int Lout, int orig, int nw, int K, int width, void* stream) {
  SVC_REQUIRE(x && kern && y, "resample_sinc: null tensor");
  SVC_REQUIRE(B > 0 && Lin > 0 && Lout > 0 && orig > 0 && nw > 0 && K > 0 && width >= 0, "resample_sinc: bad shape");
  SVC_REQUIRE((long long)Lout <= ((long long)Lin * nw + orig - 1) / orig, "resample_sinc: Lout exceeds ceil(Lin*new/orig)");
  hipStream_t s = (hipStream_t)stream;
  svc::ProfScope prof(s, "resample_sinc", 2.0 * B * (double)Lout * K, 4.0 * B * ((double)Lin + Lout));
  hipLaunchKernelGGL(resample_sinc_kernel, dim3(svc::cdiv(Lout, 256), B), dim3(256), 0, s, x, kern, y, x_bs, y_bs, Lin, Lout, orig,
                     nw, K, width);
  return svc::check_launch("resample_sinc");
}
