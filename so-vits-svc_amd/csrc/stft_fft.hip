// stft_fft.hip — rocFFT-backed STFT magnitude for the mel loss.
//
// Reference: spectrogram_torch (modules/mel_processing.py:40-64) = reflect pad, torch.stft(n_fft=win=2048, hop 512,
// hann, center=False, onesided) -> sqrt(re^2 + im^2 + 1e-6); called on y_hat every training step (train.py:172-181).
// Here: svc_stft_frame_f32 cuts the windowed frames [B*NF, n_fft] (train_ops2.hip), this file runs ONE batched
// real-to-complex rocFFT over all frames and a fused magnitude kernel on the interleaved half spectrum.  Backward:
// d|X| -> (dRe, dIm), interior bins halved, then the complex-to-real rocFFT — the exact adjoint of the forward R2C:
//   dL/dx_n = sum_{k=0}^{N/2} [gRe_k cos(2 pi k n / N) - gIm_k sin(2 pi k n / N)]
//           = C2R(Y)_n  with Y_0 = gRe_0, Y_{N/2} = gRe_{N/2}, Y_k = (gRe_k + i gIm_k) / 2 otherwise.
// Plans are opaque handles created once per (n_fft, batch); the work buffer is caller-owned (hipGraph-capturable:
// rocfft_execute only enqueues kernels on the given stream).
#include "common.h"
#include <rocfft/rocfft.h>
#include <mutex>

namespace {

struct FftPlan {
  rocfft_plan fwd = nullptr, inv = nullptr;
  rocfft_execution_info info_f = nullptr, info_i = nullptr;
  int n = 0, batch = 0;
  size_t work_f = 0, work_i = 0;
};

std::once_flag g_setup;

#define RF_CHECK(expr, what)                                        \
  do {                                                              \
    const rocfft_status st__ = (expr);                              \
    if (st__ != rocfft_status_success) {                            \
      svc::set_error("rocfft: %s failed (status %d)", what, (int)st__); \
      return SVC_ERR_HIP;                                           \
    }                                                               \
  } while (0)

// mag[r][k] = sqrt(re^2 + im^2 + eps) from interleaved z[r][k][2]
__global__ void cmag_c_kernel(const float2* __restrict__ z, float* __restrict__ mag, long long n, float eps) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float2 v = z[i];
    mag[i] = sqrtf(v.x * v.x + v.y * v.y + eps);
  }
}
// gz[r][k] = dmag/mag * z * (k interior ? 1/2 : 1)   — the input of the C2R adjoint (see header)
__global__ void cmag_c_bwd_kernel(const float2* __restrict__ z, const float* __restrict__ mag, const float* __restrict__ dmag,
                                  float2* __restrict__ gz, long long n, int bins) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % bins);
    const float s = dmag[i] / mag[i] * ((k == 0 || k == bins - 1) ? 1.f : 0.5f);
    const float2 v = z[i];
    gz[i] = make_float2(s * v.x, (k == 0 || k == bins - 1) ? 0.f : s * v.y);
  }
}

}  // namespace

extern "C" {

int svc_rfft_plan_create(int n, int batch, void** plan_out, long long* work_bytes) {
  SVC_REQUIRE(n >= 2 && (n % 2) == 0 && batch > 0 && plan_out && work_bytes, "rfft_plan_create: bad args");
  std::call_once(g_setup, [] { rocfft_setup(); });
  FftPlan* p = new FftPlan();
  p->n = n;
  p->batch = batch;
  const size_t len[1] = {(size_t)n};
  RF_CHECK(rocfft_plan_create(&p->fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                              rocfft_precision_single, 1, len, (size_t)batch, nullptr), "plan_create(real_forward)");
  RF_CHECK(rocfft_plan_create(&p->inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                              rocfft_precision_single, 1, len, (size_t)batch, nullptr), "plan_create(real_inverse)");
  RF_CHECK(rocfft_plan_get_work_buffer_size(p->fwd, &p->work_f), "get_work_buffer_size");
  RF_CHECK(rocfft_plan_get_work_buffer_size(p->inv, &p->work_i), "get_work_buffer_size");
  RF_CHECK(rocfft_execution_info_create(&p->info_f), "execution_info_create");
  RF_CHECK(rocfft_execution_info_create(&p->info_i), "execution_info_create");
  *plan_out = p;
  *work_bytes = (long long)std::max(p->work_f, p->work_i);
  return SVC_OK;
}

int svc_rfft_plan_destroy(void* plan) {
  FftPlan* p = static_cast<FftPlan*>(plan);
  if (!p) return SVC_OK;
  if (p->info_f) rocfft_execution_info_destroy(p->info_f);
  if (p->info_i) rocfft_execution_info_destroy(p->info_i);
  if (p->fwd) rocfft_plan_destroy(p->fwd);
  if (p->inv) rocfft_plan_destroy(p->inv);
  delete p;
  return SVC_OK;
}

/* x:[batch][n] real -> z:[batch][n/2+1][2] (interleaved complex), unnormalised forward DFT. */
int svc_rfft_forward_f32(void* plan, const float* x, float* z, void* work, void* stream) {
  FftPlan* p = static_cast<FftPlan*>(plan);
  SVC_REQUIRE(p && x && z && (p->work_f == 0 || work), "rfft_forward: bad args");
  RF_CHECK(rocfft_execution_info_set_stream(p->info_f, stream), "set_stream");
  if (p->work_f) RF_CHECK(rocfft_execution_info_set_work_buffer(p->info_f, work, p->work_f), "set_work_buffer");
  void* in[1] = {const_cast<float*>(x)};
  void* out[1] = {z};
  RF_CHECK(rocfft_execute(p->fwd, in, out, p->info_f), "execute(real_forward)");
  return SVC_OK;
}

/* gz:[batch][n/2+1][2] (OVERWRITTEN: rocFFT's real inverse may destroy its input) -> gx:[batch][n], unnormalised. */
int svc_rfft_inverse_f32(void* plan, float* gz, float* gx, void* work, void* stream) {
  FftPlan* p = static_cast<FftPlan*>(plan);
  SVC_REQUIRE(p && gz && gx && (p->work_i == 0 || work), "rfft_inverse: bad args");
  RF_CHECK(rocfft_execution_info_set_stream(p->info_i, stream), "set_stream");
  if (p->work_i) RF_CHECK(rocfft_execution_info_set_work_buffer(p->info_i, work, p->work_i), "set_work_buffer");
  void* in[1] = {gz};
  void* out[1] = {gx};
  RF_CHECK(rocfft_execute(p->inv, in, out, p->info_i), "execute(real_inverse)");
  return SVC_OK;
}

int svc_cmag_c_f32(const float* z, float* mag, long long n, float eps, void* stream) {
  SVC_REQUIRE(z && mag && n > 0, "cmag_c: bad args");
  hipLaunchKernelGGL(cmag_c_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const float2*>(z), mag, n, eps);
  return svc::check_launch("cmag_c");
}

int svc_cmag_c_bwd_f32(const float* z, const float* mag, const float* dmag, float* gz, long long n, int bins, void* stream) {
  SVC_REQUIRE(z && mag && dmag && gz && n > 0 && bins > 1, "cmag_c_bwd: bad args");
  hipLaunchKernelGGL(cmag_c_bwd_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const float2*>(z), mag, dmag, reinterpret_cast<float2*>(gz), n, bins);
  return svc::check_launch("cmag_c_bwd");
}

}  // extern "C"
