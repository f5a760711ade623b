// Library identification and the thread-local error text behind omnihd_last_error().
#include "common.h"

namespace omnihd {

char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
}

}  // namespace omnihd

extern "C" const char* omnihd_version(void) { return "omnihd_hip 0.4 (gfx950)"; }

extern "C" const char* omnihd_last_error(void) { return omnihd::error_buffer(); }

extern "C" int omnihd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// ---------------------------------------------------------------------------------------------
// Read-ahead of static tables into the L2 / Infinity Cache.
// The pooling forward is a chain of dependent reads per tile (descriptor -> rank table -> depth gather -> feature gathers);
// its rank tables are a pure function of the calibration, are read once per step and are long evicted when the kernel
// runs again 30-100 ms later: measured 45 us per launch with the tables cache-resident vs 73 us with everything cold
// (scripts/lab/pool_context.py).  Launched on a side stream while the elementwise kernels in front of the pooling run, this
// streaming read (10.5 MB at R1, 2-3 us) makes the tables resident just in time.  Pure hint: no effect on results.
// ---------------------------------------------------------------------------------------------
namespace omnihd {
namespace {
struct PrefetchArgs {
  const uint4* p[4];
  unsigned long long n16[4];     // 16-byte units
};
__global__ __launch_bounds__(256) void k_prefetch(PrefetchArgs a, unsigned* sink) {
  unsigned acc = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b)
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < a.n16[b]; i += (unsigned long long)gridDim.x * 256) {
      const uint4 v = a.p[b][i];
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
  if (acc == 0x9e3779b9u && sink) *sink = acc;       // never true in practice; keeps the loads alive
}
}  // namespace
}  // namespace omnihd

extern "C" int omnihd_prefetch(const void* const* ptrs, const size_t* bytes, int n, void* stream) {
  using namespace omnihd;
  OMNIHD_REQUIRE(n >= 0 && n <= 4 && (n == 0 || (ptrs && bytes)), "at most 4 buffers");
  if (n == 0) return OMNIHD_OK;
  PrefetchArgs a{};
  unsigned long long total = 0;
  for (int i = 0; i < n; ++i) {
    OMNIHD_REQUIRE(ptrs[i] && (reinterpret_cast<uintptr_t>(ptrs[i]) & 15u) == 0, "16-byte aligned buffers");
    a.p[i] = static_cast<const uint4*>(ptrs[i]);
    a.n16[i] = bytes[i] / 16;
    total += a.n16[i];
  }
  if (total == 0) return OMNIHD_OK;
  hipLaunchKernelGGL(k_prefetch, dim3(grid_for((int64_t)total, 256 * 4)), dim3(256), 0, (hipStream_t)stream, a,
                     static_cast<unsigned*>(nullptr));
  return check_launch("prefetch");
}
