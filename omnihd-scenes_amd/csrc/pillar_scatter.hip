// Pillar scatter (BEV canvas) and its backward gather for gfx950.
//
// Reference: mmdet3d PointPillarsScatter.forward_batch (un-vendored; call site
// bevfusion/detectors/bevf_faster_rcnn_bevdepth.py:101, config bevfusion.py:60-61): per sample a
// zero canvas (C, ny*nx), `canvas[:, y*nx + x] = feats.t()` — a 39.3 MB zero-fill followed by
// M*C scattered 4-byte writes at a 614 KB stride.
//
// Here the canvas is written exactly once, densely and coalesced: a small cell->pillar map is
// filled first (atomicMax = "last pillar wins", the sequential semantics), then every canvas
// element is produced by a streaming kernel that looks its pillar up.
#include "common.h"

namespace omnihd {
namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void k_cell_map(const int* __restrict__ coors, int m,
                                                     int batch, int ny, int nx,
                                                     int* __restrict__ cell_map) {
  for (int v = blockIdx.x * kBlock + threadIdx.x; v < m; v += gridDim.x * kBlock) {
    const int b = coors[v * 4 + 0], y = coors[v * 4 + 2], x = coors[v * 4 + 3];
    if (b < 0 || b >= batch || y < 0 || y >= ny || x < 0 || x >= nx) continue;
    atomicMax(&cell_map[((size_t)b * ny + y) * nx + x], v);
  }
}

// canvas [batch, c, ny, nx]; one thread per 4 consecutive x.
__global__ __launch_bounds__(kBlock) void k_canvas_nchw4(const float* __restrict__ feats,
                                                         const int* __restrict__ cell_map, int c,
                                                         int batch, int plane4,
                                                         float4* __restrict__ canvas4) {
  const int64_t total = (int64_t)batch * c * plane4;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * kBlock) {
    const int q = (int)(t % plane4);
    const int64_t bc = t / plane4;
    const int ch = (int)(bc % c);
    const int b = (int)(bc / c);
    const int4 id = reinterpret_cast<const int4*>(cell_map)[(size_t)b * plane4 + q];
    float4 o;
    o.x = id.x >= 0 ? feats[(size_t)id.x * c + ch] : 0.f;
    o.y = id.y >= 0 ? feats[(size_t)id.y * c + ch] : 0.f;
    o.z = id.z >= 0 ? feats[(size_t)id.z * c + ch] : 0.f;
    o.w = id.w >= 0 ? feats[(size_t)id.w * c + ch] : 0.f;
    __builtin_nontemporal_store(o.x, &canvas4[t].x);
    __builtin_nontemporal_store(o.y, &canvas4[t].y);
    __builtin_nontemporal_store(o.z, &canvas4[t].z);
    __builtin_nontemporal_store(o.w, &canvas4[t].w);
  }
}

__global__ __launch_bounds__(kBlock) void k_canvas_nchw1(const float* __restrict__ feats,
                                                         const int* __restrict__ cell_map, int c,
                                                         int batch, int plane,
                                                         float* __restrict__ canvas) {
  const int64_t total = (int64_t)batch * c * plane;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * kBlock) {
    const int q = (int)(t % plane);
    const int64_t bc = t / plane;
    const int ch = (int)(bc % c);
    const int b = (int)(bc / c);
    const int id = cell_map[(size_t)b * plane + q];
    canvas[t] = id >= 0 ? feats[(size_t)id * c + ch] : 0.f;
  }
}

// canvas [batch, ny, nx, c]; one thread per element (rows of c floats are contiguous).
__global__ __launch_bounds__(kBlock) void k_canvas_nhwc(const float* __restrict__ feats,
                                                        const int* __restrict__ cell_map, int c,
                                                        int64_t cells, float* __restrict__ canvas) {
  const int64_t total = cells * c;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * kBlock) {
    const int id = cell_map[t / c];
    canvas[t] = id >= 0 ? feats[(size_t)id * c + (int)(t % c)] : 0.f;
  }
}

// canvas [batch, ny, nx, c] with c % 4 == 0: one lane per float4 (a pillar row of 64 channels = 16 lanes = one 256-byte
// coalesced, non-temporal store).  RESET: the map entry goes back to -1 once its row is out, so a caller-owned map that was
// clean before the cell-map kernel is clean again after this one (no memset launch per call).
template <bool RESET>
__global__ __launch_bounds__(kBlock) void k_canvas_nhwc4(const float4* __restrict__ feats4, int* cell_map, int c4,
                                                         unsigned total, float4* __restrict__ canvas4) {
  for (unsigned t = blockIdx.x * kBlock + threadIdx.x; t < total; t += gridDim.x * kBlock) {
    const unsigned cell = t / (unsigned)c4, sub = t - cell * (unsigned)c4;
    const int id = cell_map[cell];
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (id >= 0) o = feats4[(size_t)id * c4 + sub];
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v v = {o.x, o.y, o.z, o.w};
    __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(canvas4 + t));
    if (RESET && id >= 0 && sub == 0) cell_map[cell] = -1;
  }
}

__global__ __launch_bounds__(kBlock) void k_gather(const float* __restrict__ cg,
                                                   const int* __restrict__ coors, int m, int c,
                                                   int batch, int ny, int nx, int channels_last,
                                                   float* __restrict__ fg) {
  const int64_t total = (int64_t)m * c;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * kBlock) {
    const int v = (int)(t / c), ch = (int)(t % c);
    const int b = coors[v * 4 + 0], y = coors[v * 4 + 2], x = coors[v * 4 + 3];
    float g = 0.f;
    if (b >= 0 && b < batch && y >= 0 && y < ny && x >= 0 && x < nx) {
      g = channels_last ? cg[(((size_t)b * ny + y) * nx + x) * c + ch]
                        : cg[(((size_t)b * c + ch) * ny + y) * nx + x];
    }
    fg[t] = g;
  }
}

}  // namespace
}  // namespace omnihd

using namespace omnihd;

extern "C" size_t omnihd_pillar_scatter_workspace_bytes(int batch, int ny, int nx) {
  if (batch <= 0 || ny <= 0 || nx <= 0) return 256;
  return align_up((size_t)batch * ny * nx * sizeof(int), 256);
}

extern "C" int omnihd_pillar_scatter(const float* feats, const int* coors, int m, int c,
                                     int batch, int ny, int nx, int channels_last, float* canvas,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  OMNIHD_REQUIRE(m >= 0 && c > 0 && batch > 0 && ny > 0 && nx > 0, "sizes");
  OMNIHD_REQUIRE(canvas && workspace && (m == 0 || (feats && coors)), "null pointer");
  const size_t cells = (size_t)batch * ny * nx;
  if (workspace_bytes < cells * sizeof(int)) {
    set_error("pillar_scatter: workspace %zu < required %zu", workspace_bytes, cells * sizeof(int));
    return OMNIHD_ERR_WORKSPACE;
  }
  int* cell_map = static_cast<int*>(workspace);
  OMNIHD_HIP_TRY(hipMemsetAsync(cell_map, 0xFF, cells * sizeof(int), st));
  if (m > 0)
    hipLaunchKernelGGL(k_cell_map, dim3(grid_for(m, kBlock)), dim3(kBlock), 0, st, coors, m, batch,
                       ny, nx, cell_map);
  const int plane = ny * nx;
  if (channels_last) {
    hipLaunchKernelGGL(k_canvas_nhwc, dim3(grid_for((int64_t)cells * c, kBlock * 4)), dim3(kBlock),
                       0, st, feats, cell_map, c, (int64_t)cells, canvas);
  } else if (plane % 4 == 0 && (reinterpret_cast<uintptr_t>(canvas) & 15u) == 0 &&
             (reinterpret_cast<uintptr_t>(cell_map) & 15u) == 0) {
    hipLaunchKernelGGL(k_canvas_nchw4, dim3(grid_for((int64_t)batch * c * (plane / 4), kBlock * 2)),
                       dim3(kBlock), 0, st, feats, cell_map, c, batch, plane / 4,
                       reinterpret_cast<float4*>(canvas));
  } else {
    hipLaunchKernelGGL(k_canvas_nchw1, dim3(grid_for((int64_t)batch * c * plane, kBlock * 4)),
                       dim3(kBlock), 0, st, feats, cell_map, c, batch, plane, canvas);
  }
  return check_launch("pillar_scatter");
}

extern "C" int omnihd_pillar_cell_map(const int* coors, int m, int batch, int ny, int nx, int* cell_map, void* stream) {
  OMNIHD_REQUIRE(m >= 0 && batch > 0 && ny > 0 && nx > 0 && cell_map && (m == 0 || coors), "arguments");
  if (m == 0) return OMNIHD_OK;
  hipLaunchKernelGGL(k_cell_map, dim3(grid_for(m, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, coors, m, batch, ny, nx, cell_map);
  return check_launch("pillar_cell_map");
}

extern "C" int omnihd_pillar_canvas(const float* feats, int* cell_map, int c, int batch, int ny, int nx, int channels_last,
                                    int reset_map, float* canvas, void* stream) {
  OMNIHD_REQUIRE(c > 0 && batch > 0 && ny > 0 && nx > 0 && cell_map && canvas, "arguments");
  hipStream_t st = (hipStream_t)stream;
  const size_t cells = (size_t)batch * ny * nx;
  const int plane = ny * nx;
  if (channels_last && c % 4 == 0 && cells * (c / 4) < (1ull << 32) && feats != nullptr &&
      ((reinterpret_cast<uintptr_t>(feats) | reinterpret_cast<uintptr_t>(canvas)) & 15u) == 0) {
    const unsigned total = (unsigned)(cells * (c / 4));
    const int grid = grid_for(total, kBlock * 2);
    // The in-kernel reset is ordered only inside ONE wavefront: lane sub == 0 clears the entry its cell's other lanes read in
    // the same load instruction.  That holds when the c/4 lanes of a cell are consecutive lanes of one 64-aligned group, i.e.
    // c/4 a power of two <= 64 (c = 4 ... 256); with c = 48, 96, 320 ... a cell straddles wavefronts or workgroups and a later
    // one would read the cleared entry and store zeros for part of a pillar row — those take the memset.
    const int c4 = c / 4;
    const bool reset_in_kernel = reset_map && (c4 & (c4 - 1)) == 0 && c4 <= 64;
    if (reset_in_kernel)
      hipLaunchKernelGGL(k_canvas_nhwc4<true>, dim3(grid), dim3(kBlock), 0, st, reinterpret_cast<const float4*>(feats), cell_map,
                         c4, total, reinterpret_cast<float4*>(canvas));
    else
      hipLaunchKernelGGL(k_canvas_nhwc4<false>, dim3(grid), dim3(kBlock), 0, st, reinterpret_cast<const float4*>(feats), cell_map,
                         c4, total, reinterpret_cast<float4*>(canvas));
    if (reset_map && !reset_in_kernel) OMNIHD_HIP_TRY(hipMemsetAsync(cell_map, 0xFF, cells * sizeof(int), st));
    return check_launch("pillar_canvas(nhwc4)");
  }
  if (channels_last) {
    hipLaunchKernelGGL(k_canvas_nhwc, dim3(grid_for((int64_t)cells * c, kBlock * 4)), dim3(kBlock), 0, st, feats, cell_map, c,
                       (int64_t)cells, canvas);
  } else if (plane % 4 == 0 && (reinterpret_cast<uintptr_t>(canvas) & 15u) == 0 && (reinterpret_cast<uintptr_t>(cell_map) & 15u) == 0) {
    hipLaunchKernelGGL(k_canvas_nchw4, dim3(grid_for((int64_t)batch * c * (plane / 4), kBlock * 2)), dim3(kBlock), 0, st, feats,
                       cell_map, c, batch, plane / 4, reinterpret_cast<float4*>(canvas));
  } else {
    hipLaunchKernelGGL(k_canvas_nchw1, dim3(grid_for((int64_t)batch * c * plane, kBlock * 4)), dim3(kBlock), 0, st, feats, cell_map,
                       c, batch, plane, canvas);
  }
  if (reset_map) OMNIHD_HIP_TRY(hipMemsetAsync(cell_map, 0xFF, cells * sizeof(int), st));
  return check_launch("pillar_canvas");
}

extern "C" int omnihd_pillar_gather(const float* canvas_grad, const int* coors, int m, int c,
                                    int batch, int ny, int nx, int channels_last,
                                    float* feats_grad, void* stream) {
  OMNIHD_REQUIRE(m >= 0 && c > 0 && batch > 0 && ny > 0 && nx > 0, "sizes");
  if (m == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(canvas_grad && coors && feats_grad, "null pointer");
  hipLaunchKernelGGL(k_gather, dim3(grid_for((int64_t)m * c, kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, canvas_grad, coors, m, c, batch, ny, nx, channels_last,
                     feats_grad);
  return check_launch("pillar_gather");
}
