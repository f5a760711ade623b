// bev_pool (v1) forward / backward for gfx950.
//
// The reference (ops/bev_pool/src/bev_pool_cuda.cu:20-42, :61-84) sums pre-multiplied point
// features x[n,c] that are already sorted by voxel rank.  Rows of one interval are therefore
// CONSECUTIVE in memory: a group of C/4 lanes streams them with float4 loads (forward) or
// broadcasts one out_grad row over them with float4 stores (backward).
#include "common.h"

namespace omnihd {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ size_t v1_out_row(const int* g, int d, int h, int w) {
  // geom_feats = (h_idx, w_idx, d_idx, b_idx); out is [b, d, h, w, c]
  // (bev_pool_cuda.cu:34-36: g[3]*d*h*w + g[2]*h*w + g[0]*w + g[1]).
  return ((size_t)g[3] * d + g[2]) * h * (size_t)w + (size_t)g[0] * w + g[1];
}

template <int C4>
__global__ __launch_bounds__(kBlock) void k_v1_fwd(const float4* __restrict__ x4,
                                                   const int* __restrict__ geom,
                                                   const int* __restrict__ starts,
                                                   const int* __restrict__ lengths,
                                                   float4* __restrict__ out4, int d, int h, int w,
                                                   int n_intervals) {
  constexpr int G = kBlock / C4;
  const int sub = threadIdx.x % C4, grp = threadIdx.x / C4;
  for (int base = blockIdx.x * G; base < n_intervals; base += gridDim.x * G) {
    const int iv = base + grp;
    if (iv >= n_intervals) continue;
    const int s = starts[iv], len = lengths[iv];
    const float4* p = x4 + (size_t)s * C4 + sub;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < len; ++i) {
      const float4 v = p[(size_t)i * C4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    out4[v1_out_row(geom + (size_t)s * 4, d, h, w) * C4 + sub] = acc;
  }
}

template <int C4>
__global__ __launch_bounds__(kBlock) void k_v1_bwd(const float4* __restrict__ og4,
                                                   const int* __restrict__ geom,
                                                   const int* __restrict__ starts,
                                                   const int* __restrict__ lengths,
                                                   float4* __restrict__ xg4, int d, int h, int w,
                                                   int n_intervals) {
  constexpr int G = kBlock / C4;
  const int sub = threadIdx.x % C4, grp = threadIdx.x / C4;
  for (int base = blockIdx.x * G; base < n_intervals; base += gridDim.x * G) {
    const int iv = base + grp;
    if (iv >= n_intervals) continue;
    const int s = starts[iv], len = lengths[iv];
    const float4 g = og4[v1_out_row(geom + (size_t)s * 4, d, h, w) * C4 + sub];
    float4* p = xg4 + (size_t)s * C4 + sub;
    for (int i = 0; i < len; ++i) p[(size_t)i * C4] = g;
  }
}

__global__ __launch_bounds__(kBlock) void k_v1_fwd_generic(const float* __restrict__ x,
                                                           const int* __restrict__ geom,
                                                           const int* __restrict__ starts,
                                                           const int* __restrict__ lengths,
                                                           float* __restrict__ out, int d, int h,
                                                           int w, int c, int n_intervals) {
  const int64_t total = (int64_t)n_intervals * c;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * kBlock) {
    const int iv = (int)(t / c), ch = (int)(t % c);
    const int s = starts[iv], len = lengths[iv];
    float acc = 0.f;
    for (int i = 0; i < len; ++i) acc += x[(size_t)(s + i) * c + ch];
    out[v1_out_row(geom + (size_t)s * 4, d, h, w) * c + ch] = acc;
  }
}

__global__ __launch_bounds__(kBlock) void k_v1_bwd_generic(const float* __restrict__ og,
                                                           const int* __restrict__ geom,
                                                           const int* __restrict__ starts,
                                                           const int* __restrict__ lengths,
                                                           float* __restrict__ xg, int d, int h,
                                                           int w, int c, int n_intervals) {
  const int64_t total = (int64_t)n_intervals * c;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * kBlock) {
    const int iv = (int)(t / c), ch = (int)(t % c);
    const int s = starts[iv], len = lengths[iv];
    const float g = og[v1_out_row(geom + (size_t)s * 4, d, h, w) * c + ch];
    for (int i = 0; i < len; ++i) xg[(size_t)(s + i) * c + ch] = g;
  }
}

inline bool vec_ok(int c, const void* a, const void* b) {
  if (c % 4) return false;
  const int c4 = c / 4;
  if (c4 > 64 || 64 % c4) return false;
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15u) == 0;
}

}  // namespace
}  // namespace omnihd

using namespace omnihd;

#define OMNIHD_V1_SWITCH(KERNEL, A, B)                                                        \
  switch (c / 4) {                                                                            \
    case 1: hipLaunchKernelGGL((KERNEL<1>), dim3(grid), dim3(kBlock), 0, st, A, geom_feats, interval_starts, interval_lengths, B, d, h, w, n_intervals); break;   \
    case 2: hipLaunchKernelGGL((KERNEL<2>), dim3(grid), dim3(kBlock), 0, st, A, geom_feats, interval_starts, interval_lengths, B, d, h, w, n_intervals); break;   \
    case 4: hipLaunchKernelGGL((KERNEL<4>), dim3(grid), dim3(kBlock), 0, st, A, geom_feats, interval_starts, interval_lengths, B, d, h, w, n_intervals); break;   \
    case 8: hipLaunchKernelGGL((KERNEL<8>), dim3(grid), dim3(kBlock), 0, st, A, geom_feats, interval_starts, interval_lengths, B, d, h, w, n_intervals); break;   \
    case 16: hipLaunchKernelGGL((KERNEL<16>), dim3(grid), dim3(kBlock), 0, st, A, geom_feats, interval_starts, interval_lengths, B, d, h, w, n_intervals); break; \
    case 32: hipLaunchKernelGGL((KERNEL<32>), dim3(grid), dim3(kBlock), 0, st, A, geom_feats, interval_starts, interval_lengths, B, d, h, w, n_intervals); break; \
    default: hipLaunchKernelGGL((KERNEL<64>), dim3(grid), dim3(kBlock), 0, st, A, geom_feats, interval_starts, interval_lengths, B, d, h, w, n_intervals); break; \
  }

extern "C" int omnihd_bev_pool_v1_fwd(const float* x, const int* geom_feats,
                                      const int* interval_starts, const int* interval_lengths,
                                      float* out, int b, int d, int h, int w, int n, int c,
                                      int n_intervals, void* stream) {
  OMNIHD_REQUIRE(b > 0 && d > 0 && h > 0 && w > 0 && c > 0 && n >= 0 && n_intervals >= 0,
                 "positive shape");
  if (n_intervals == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(x && geom_feats && interval_starts && interval_lengths && out, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (vec_ok(c, x, out)) {
    const int grid = grid_for(n_intervals, (kBlock / (c / 4)) * 4);
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float4* o4 = reinterpret_cast<float4*>(out);
    OMNIHD_V1_SWITCH(k_v1_fwd, x4, o4)
  } else {
    const int grid = grid_for((int64_t)n_intervals * c, kBlock);
    hipLaunchKernelGGL(k_v1_fwd_generic, dim3(grid), dim3(kBlock), 0, st, x, geom_feats,
                       interval_starts, interval_lengths, out, d, h, w, c, n_intervals);
  }
  return check_launch("bev_pool_v1_fwd");
}

extern "C" int omnihd_bev_pool_v1_bwd(const float* out_grad, const int* geom_feats,
                                      const int* interval_starts, const int* interval_lengths,
                                      float* x_grad, int b, int d, int h, int w, int n, int c,
                                      int n_intervals, void* stream) {
  OMNIHD_REQUIRE(b > 0 && d > 0 && h > 0 && w > 0 && c > 0 && n >= 0 && n_intervals >= 0,
                 "positive shape");
  if (n_intervals == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(out_grad && geom_feats && interval_starts && interval_lengths && x_grad,
                 "null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (vec_ok(c, out_grad, x_grad)) {
    const int grid = grid_for(n_intervals, (kBlock / (c / 4)) * 4);
    const float4* g4 = reinterpret_cast<const float4*>(out_grad);
    float4* x4 = reinterpret_cast<float4*>(x_grad);
    OMNIHD_V1_SWITCH(k_v1_bwd, g4, x4)
  } else {
    const int grid = grid_for((int64_t)n_intervals * c, kBlock);
    hipLaunchKernelGGL(k_v1_bwd_generic, dim3(grid), dim3(kBlock), 0, st, out_grad, geom_feats,
                       interval_starts, interval_lengths, x_grad, d, h, w, c, n_intervals);
  }
  return check_launch("bev_pool_v1_bwd");
}
