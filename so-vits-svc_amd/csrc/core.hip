// core.hip — error reporting, device info and per-launch hipEvent profiling for libsvc_hip.so.
#include "common.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace svc {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- profiling -----------------------------------------------------------------------------------
struct ProfRec {
  std::string name;
  hipEvent_t e0, e1;
  double flop, bytes;
};
struct ProfAgg {
  long calls = 0;
  double ms = 0, flop = 0, bytes = 0;
};
static bool g_prof = false;
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_pending;
static std::vector<hipEvent_t> g_pool;
static std::map<std::string, ProfAgg> g_agg;
static std::vector<std::string> g_order;
static thread_local ProfRec g_cur;

bool prof_on() { return g_prof; }
bool prof_shapes() {
  static const bool on = getenv("SVC_PROF_SHAPES") != nullptr;
  return on;
}

static hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

void prof_begin(hipStream_t s, const char* name, double flop, double bytes) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_cur.name = name;
  g_cur.flop = flop;
  g_cur.bytes = bytes;
  g_cur.e0 = get_event();
  g_cur.e1 = get_event();
  hipEventRecord(g_cur.e0, s);
}

void prof_end(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  hipEventRecord(g_cur.e1, s);
  g_pending.push_back(g_cur);
}

static void drain() {
  for (auto& r : g_pending) {
    hipEventSynchronize(r.e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.e0, r.e1);
    if (!g_agg.count(r.name)) g_order.push_back(r.name);
    ProfAgg& a = g_agg[r.name];
    a.calls++;
    a.ms += ms;
    a.flop += r.flop;
    a.bytes += r.bytes;
    g_pool.push_back(r.e0);
    g_pool.push_back(r.e1);
  }
  g_pending.clear();
}

}  // namespace svc

extern "C" {

const char* svc_last_error(void) { return svc::g_err; }

int svc_abi_version(void) { return SVC_ABI_VERSION; }   // history: include/svc_hip.h (SVC_ABI_VERSION)

int svc_device_info(char* name, int len) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    svc::set_error("hipGetDevice failed");
    return SVC_ERR_HIP;
  }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
    svc::set_error("hipGetDeviceProperties failed");
    return SVC_ERR_HIP;
  }
  if (name && len > 0) {
    strncpy(name, p.gcnArchName, len - 1);
    name[len - 1] = 0;
  }
  return p.multiProcessorCount;
}

namespace {
__global__ void svc_empty_kernel() {}
}  // namespace
/* bench.py's launch-latency probe: one empty one-wave kernel on `stream` (capturable). */
int svc_debug_empty_kernel(void* stream) {
  hipLaunchKernelGGL(svc_empty_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream);
  return svc::check_launch("svc_debug_empty_kernel");
}

int svc_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(svc::g_prof_mu);
  svc::g_prof = on != 0;
  return SVC_OK;
}

int svc_prof_reset(void) {
  std::lock_guard<std::mutex> lk(svc::g_prof_mu);
  svc::drain();
  svc::g_agg.clear();
  svc::g_order.clear();
  return SVC_OK;
}

int svc_prof_report(char* buf, int len) {
  std::lock_guard<std::mutex> lk(svc::g_prof_mu);
  svc::drain();
  int off = 0;
  for (auto& n : svc::g_order) {
    auto& a = svc::g_agg[n];
    int w = snprintf(buf + off, len > off ? len - off : 0, "%s %ld %.6f %.0f %.0f\n", n.c_str(), a.calls,
                     a.ms, a.flop, a.bytes);
    if (w < 0 || off + w >= len) break;
    off += w;
  }
  return off;
}

}  // extern "C"
