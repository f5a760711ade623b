// Shared helpers for the libomnihd_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "omnihd_hip.h"

namespace omnihd {

// Thread-local last-error text behind omnihd_last_error().
char* error_buffer();
void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return OMNIHD_ERR_LAUNCH;
  }
  return OMNIHD_OK;
}

#define OMNIHD_HIP_TRY(expr)                                                   \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) {                                                    \
      ::omnihd::set_error("%s failed: %s", #expr, hipGetErrorString(_e));      \
      return OMNIHD_ERR_RUNTIME;                                               \
    }                                                                          \
  } while (0)

#define OMNIHD_REQUIRE(cond, msg)                                              \
  do {                                                                         \
    if (!(cond)) {                                                             \
      ::omnihd::set_error("%s: requirement failed: %s", __func__, msg);        \
      return OMNIHD_ERR_ARG;                                                   \
    }                                                                          \
  } while (0)

// MI355X: 256 CUs in 8 XCDs.  Memory-bound grid-stride kernels are capped at 8 workgroups of
// 256 threads per CU (cdna_hip_programming.md, Guideline 11).
constexpr int kCUs = 256;
constexpr int kMaxBlocks = kCUs * 8;

inline int grid_for(int64_t work_items, int items_per_block) {
  int64_t blocks = (work_items + items_per_block - 1) / items_per_block;
  if (blocks < 1) blocks = 1;
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  return (int)blocks;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace omnihd
