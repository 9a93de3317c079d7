// common.h — shared host-side helpers for libsvc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "../../include/svc_hip.h"

namespace svc {

void set_error(const char* fmt, ...);

// Per-launch profiling hooks (profile.hip).
bool prof_on();
bool prof_shapes();   // SVC_PROF_SHAPES=1: profile rows keyed by kernel shape
void prof_begin(hipStream_t s, const char* name, double flop, double bytes);
void prof_end(hipStream_t s);

struct ProfScope {
  hipStream_t s;
  bool on;
  ProfScope(hipStream_t s_, const char* name, double flop, double bytes, bool enable = true) : s(s_), on(enable && prof_on()) {
    if (on) prof_begin(s, name, flop, bytes);
  }
  ~ProfScope() {
    if (on) prof_end(s);
  }
};

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return SVC_ERR_HIP;
  }
  return SVC_OK;
}

#define SVC_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      ::svc::set_error(__VA_ARGS__);    \
      return SVC_ERR_BAD_ARG;           \
    }                                   \
  } while (0)

// conv1d_strip.hip: long-sequence dense conv, one workgroup per CU; returns 1 when the shape is not one of its shapes
int conv1d_strip_try(const svc_conv1d_args& a, hipStream_t s, bool dry = false);
// ConvTranspose1d weight layout: true = phases as rows [Cin][M][u*CoutP] (row = co*u + phase), one dense convolution;
// false = one block per phase [u][Cin][M][CoutP].  Powers of two divide every row tile of the conv kernels (16..128).
inline bool convt_rows_layout(int stride) {
  static const bool on = [] { const char* e = getenv("SVC_CONVT_ROWS"); return !(e && e[0] == '0'); }();   // A/B switch
  return on && stride >= 2 && stride <= 16 && (stride & (stride - 1)) == 0;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }

}  // namespace svc

// ---- device helpers -------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float svc_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float svc_sigmoid(float v) { return 1.f / (1.f + __expf(-v)); }
__device__ __forceinline__ float svc_gelu(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }
