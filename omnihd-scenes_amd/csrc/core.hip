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

extern "C" const char* omnihd_version(void) { return "omnihd_hip 0.1 (gfx950)"; }

extern "C" const char* omnihd_last_error(void) { return omnihd::error_buffer(); }

extern "C" int omnihd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
