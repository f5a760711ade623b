// Rank-table preparation on the device.
//
// What the reference does (bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:302-362):
// materialise int64 (Ntot,4) coordinates, boolean-compress three arrays, argsort int64 keys,
// gather three arrays, build intervals with torch.where — every forward; and a second argsort
// + gathers in every backward (ops/bev_pool_v2/bev_pool.py:47-57).
//
// Here: one fused pass turns each frustum point into a 32-bit voxel key (or a sentinel), a
// rocPRIM LSD radix sort restricted to the bits the grid needs orders (key, point index)
// stably — stable order IS the canonical order of SURVEY D6 — and the interval tables come
// from a head-flag scan.  The host keeps the result cached per calibration.
#include "common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include <rocprim/functional.hpp>

namespace omnihd {
namespace {

constexpr int kBlock = 256;

struct GridSpec {
  float off[3];
  float dx[3];
  int nx[3];
};

// -ffp-contract is irrelevant here (no a*b+c), and hipcc's default fp32 division is correctly
// rounded, so t below is the IEEE result torch computes on any backend.
__global__ __launch_bounds__(kBlock) void k_rank_keys(const float* __restrict__ geom,
                                                      int64_t n_total, int64_t pts_per_batch,
                                                      GridSpec g, uint32_t* __restrict__ keys,
                                                      int* __restrict__ idx, uint32_t sentinel) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_total;
       i += (int64_t)gridDim.x * kBlock) {
    const float* p = geom + i * 3;
    const float tx = (p[0] - g.off[0]) / g.dx[0];
    const float ty = (p[1] - g.off[1]) / g.dx[1];
    const float tz = (p[2] - g.off[2]) / g.dx[2];
    // trunc(t) in [0, n)  <=>  -1 < t < n   (false for NaN, like the reference's int64 compare
    // after an x86 float->int64 conversion, which yields INT64_MIN for NaN/overflow).
    const bool kept = tx > -1.f && tx < (float)g.nx[0] && ty > -1.f && ty < (float)g.nx[1] &&
                      tz > -1.f && tz < (float)g.nx[2];
    uint32_t key = sentinel;
    if (kept) {
      const int x = (int)tx, y = (int)ty, z = (int)tz;  // toward zero: (-1,0) -> 0 (D3)
      const int64_t b = i / pts_per_batch;
      key = (uint32_t)(((b * g.nx[2] + z) * g.nx[1] + y) * g.nx[0] + x);
    }
    keys[i] = key;
    idx[i] = (int)i;
  }
}

__global__ __launch_bounds__(kBlock) void k_iota(int* p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock)
    p[i] = (int)i;
}

__global__ __launch_bounds__(kBlock) void k_gather3(const int* __restrict__ perm, int64_t n,
                                                    const int* __restrict__ a,
                                                    const int* __restrict__ b,
                                                    const int* __restrict__ c,
                                                    int* __restrict__ ao, int* __restrict__ bo,
                                                    int* __restrict__ co) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock) {
    const int p = perm[i];
    if (a) ao[i] = a[p];
    if (b) bo[i] = b[p];
    if (c) co[i] = c[p];
  }
}

struct HeadFlag {
  const uint32_t* keys;
  uint32_t sentinel;
  __host__ __device__ int operator()(int64_t i) const {
    const uint32_t k = keys[i];
    if (k == sentinel) return 0;
    return (i == 0 || keys[i - 1] != k) ? 1 : 0;
  }
};

// starts[pos[i]] = i for every head; counts[0] = n_points, counts[1] = n_intervals.
__global__ __launch_bounds__(kBlock) void k_scatter_heads(const uint32_t* __restrict__ keys,
                                                          const int* __restrict__ pos, int64_t n,
                                                          uint32_t sentinel,
                                                          int* __restrict__ starts,
                                                          int* __restrict__ counts) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock) {
    const uint32_t k = keys[i];
    const bool valid = k != sentinel;
    const bool head = valid && (i == 0 || keys[i - 1] != k);
    if (head) starts[pos[i]] = (int)i;
    if (valid && (i == n - 1 || keys[i + 1] == sentinel)) counts[0] = (int)(i + 1);
    if (i == n - 1) counts[1] = pos[i] + (head ? 1 : 0);
  }
}

__global__ __launch_bounds__(kBlock) void k_lengths(const int* __restrict__ starts,
                                                    const int* __restrict__ counts,
                                                    int* __restrict__ lengths, int64_t cap) {
  const int n_points = counts[0];
  const int n_int = counts[1];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < cap;
       i += (int64_t)gridDim.x * kBlock) {
    if (i < n_int) lengths[i] = (i + 1 < n_int ? starts[i + 1] : n_points) - starts[i];
  }
}

__global__ __launch_bounds__(kBlock) void k_ranks_feat(const int* __restrict__ rd, int64_t n,
                                                       int dhw, int hw, int* __restrict__ rf) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock) {
    const int r = rd[i];
    rf[i] = (r / dhw) * hw + r % hw;
  }
}

__global__ __launch_bounds__(kBlock) void k_csr(const uint32_t* __restrict__ keys, int n_points,
                                                int n_rows, int* __restrict__ row_ptr) {
  for (int r = blockIdx.x * kBlock + threadIdx.x; r <= n_rows; r += gridDim.x * kBlock) {
    int lo = 0, hi = n_points;  // first i with keys[i] >= r
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (keys[mid] < (uint32_t)r) lo = mid + 1; else hi = mid;
    }
    row_ptr[r] = lo;
  }
}

__global__ __launch_bounds__(kBlock) void k_perm_zyx_yxz(const int* __restrict__ in, int64_t n,
                                                         int nz, int ny, int nx,
                                                         int* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock) {
    int r = in[i];
    const int x = r % nx; r /= nx;
    const int y = r % ny; r /= ny;
    const int z = r % nz; const int b = r / nz;
    out[i] = ((b * ny + y) * nx + x) * nz + z;
  }
}

// A new tile starts at row r when the running item count (r + row_ptr[r] = rows + points before
// r) crosses a multiple of tile_items, and around every LONG row (more than long_len points),
// which always forms a tile of its own.
struct TileStart {
  const int* row_ptr;
  int tile_items;
  int long_len;
  __host__ __device__ bool is_long(int r) const { return row_ptr[r + 1] - row_ptr[r] > long_len; }
  __host__ __device__ int operator()(int r) const {
    if (r == 0) return 1;
    if (is_long(r) || is_long(r - 1)) return 1;
    return ((long long)r + row_ptr[r]) / tile_items != ((long long)r - 1 + row_ptr[r - 1]) / tile_items;
  }
};

__global__ void k_close_tiles(int* tile_row, const int* count, int n_rows) { tile_row[*count] = n_rows; }

struct SortWs {
  size_t tmp_bytes;   // rocprim temporary storage (max over the calls we make)
  size_t off_perm_in, off_perm_out, off_pos, total;
};

int sort_ws_layout(int64_t n, SortWs* ws) {
  size_t a = 0, b = 0, c = 0;
  uint32_t* k = nullptr;
  int* v = nullptr;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, a, k, k, v, v, (size_t)n, 0, 32, 0, false);
  if (e != hipSuccess) { set_error("rocprim radix_sort_pairs size query: %s", hipGetErrorString(e)); return OMNIHD_ERR_RUNTIME; }
  HeadFlag hf{nullptr, 0};
  auto flags = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), hf);
  e = rocprim::exclusive_scan(nullptr, b, flags, v, 0, (size_t)n, rocprim::plus<int>(), 0, false);
  if (e != hipSuccess) { set_error("rocprim exclusive_scan size query: %s", hipGetErrorString(e)); return OMNIHD_ERR_RUNTIME; }
  (void)c;
  ws->tmp_bytes = align_up(a > b ? a : b, 256) + 256;
  ws->off_perm_in = ws->tmp_bytes;
  ws->off_perm_out = ws->off_perm_in + align_up((size_t)n * 4, 256);
  ws->off_pos = ws->off_perm_out + align_up((size_t)n * 4, 256);
  ws->total = ws->off_pos + align_up((size_t)n * 4, 256);
  return OMNIHD_OK;
}

}  // namespace
}  // namespace omnihd

using namespace omnihd;

extern "C" int omnihd_bev_rank_keys(const float* geom, int64_t n_total, int64_t pts_per_batch,
                                    const float* h_off3, const float* h_dx3, const int* h_nx3,
                                    uint32_t* keys, int* idx, uint32_t sentinel, void* stream) {
  OMNIHD_REQUIRE(n_total >= 0 && pts_per_batch > 0, "sizes");
  OMNIHD_REQUIRE(n_total < (int64_t)1 << 31, "n_total must fit int32 (reference tables are int32)");
  if (n_total == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(geom && h_off3 && h_dx3 && h_nx3 && keys && idx, "null pointer");
  GridSpec g;
  for (int a = 0; a < 3; ++a) { g.off[a] = h_off3[a]; g.dx[a] = h_dx3[a]; g.nx[a] = h_nx3[a]; }
  hipLaunchKernelGGL(k_rank_keys, dim3(grid_for(n_total, kBlock * 4)), dim3(kBlock), 0,
                     (hipStream_t)stream, geom, n_total, pts_per_batch, g, keys, idx, sentinel);
  return check_launch("bev_rank_keys");
}

extern "C" size_t omnihd_sort_ranks_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  SortWs ws;
  if (sort_ws_layout(n, &ws) != OMNIHD_OK) return 0;
  return ws.total;
}

extern "C" int omnihd_sort_ranks(const uint32_t* keys_in, const int* p0_in, const int* p1_in,
                                 const int* p2_in, int64_t n, int key_bits, uint32_t sentinel,
                                 uint32_t* keys_out, int* p0_out, int* p1_out, int* p2_out,
                                 int* interval_starts, int* interval_lengths, int* counts,
                                 int* h_counts, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  hipStream_t st = (hipStream_t)stream;
  OMNIHD_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "0 <= n < 2^31");
  OMNIHD_REQUIRE(key_bits >= 1 && key_bits <= 32, "1 <= key_bits <= 32");
  OMNIHD_REQUIRE(counts, "counts is null");
  OMNIHD_HIP_TRY(hipMemsetAsync(counts, 0, 2 * sizeof(int), st));
  if (n > 0) {
    OMNIHD_REQUIRE(keys_in && keys_out && interval_starts && interval_lengths && workspace,
                   "null pointer");
    OMNIHD_REQUIRE((!p0_in) == (!p0_out) && (!p1_in) == (!p1_out) && (!p2_in) == (!p2_out),
                   "payload in/out must be given together");
    SortWs ws;
    int rc = sort_ws_layout(n, &ws);
    if (rc != OMNIHD_OK) return rc;
    if (workspace_bytes < ws.total) {
      set_error("sort_ranks: workspace %zu < required %zu", workspace_bytes, ws.total);
      return OMNIHD_ERR_WORKSPACE;
    }
    char* base = static_cast<char*>(workspace);
    int* perm_in = reinterpret_cast<int*>(base + ws.off_perm_in);
    int* perm_out = reinterpret_cast<int*>(base + ws.off_perm_out);
    int* pos = reinterpret_cast<int*>(base + ws.off_pos);
    size_t tmp = ws.tmp_bytes;
    const int grid = grid_for(n, kBlock * 4);
    const bool single = p0_in && !p1_in && !p2_in;
    if (single) {
      OMNIHD_HIP_TRY(rocprim::radix_sort_pairs(base, tmp, keys_in, keys_out, p0_in, p0_out,
                                               (size_t)n, 0, (unsigned)key_bits, st, false));
    } else {
      hipLaunchKernelGGL(k_iota, dim3(grid), dim3(kBlock), 0, st, perm_in, n);
      OMNIHD_HIP_TRY(rocprim::radix_sort_pairs(base, tmp, keys_in, keys_out, perm_in, perm_out,
                                               (size_t)n, 0, (unsigned)key_bits, st, false));
      if (p0_in || p1_in || p2_in)
        hipLaunchKernelGGL(k_gather3, dim3(grid), dim3(kBlock), 0, st, perm_out, n, p0_in, p1_in,
                           p2_in, p0_out, p1_out, p2_out);
    }
    HeadFlag hf{keys_out, sentinel};
    auto flags = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), hf);
    tmp = ws.tmp_bytes;
    OMNIHD_HIP_TRY(rocprim::exclusive_scan(base, tmp, flags, pos, 0, (size_t)n,
                                           rocprim::plus<int>(), st, false));
    hipLaunchKernelGGL(k_scatter_heads, dim3(grid), dim3(kBlock), 0, st, keys_out, pos, n,
                       sentinel, interval_starts, counts);
    hipLaunchKernelGGL(k_lengths, dim3(grid), dim3(kBlock), 0, st, interval_starts, counts,
                       interval_lengths, n);
    rc = check_launch("sort_ranks");
    if (rc != OMNIHD_OK) return rc;
  }
  if (h_counts) {
    OMNIHD_HIP_TRY(hipMemcpyAsync(h_counts, counts, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    OMNIHD_HIP_TRY(hipStreamSynchronize(st));
  }
  return OMNIHD_OK;
}

extern "C" int omnihd_ranks_feat_from_depth(const int* ranks_depth, int64_t n, int d, int hw,
                                            int* ranks_feat, void* stream) {
  OMNIHD_REQUIRE(n >= 0 && d > 0 && hw > 0, "sizes");
  if (n == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(ranks_depth && ranks_feat, "null pointer");
  hipLaunchKernelGGL(k_ranks_feat, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0,
                     (hipStream_t)stream, ranks_depth, n, d * hw, hw, ranks_feat);
  return check_launch("ranks_feat_from_depth");
}

extern "C" int omnihd_csr_from_sorted_keys(const uint32_t* sorted_keys, int n_points, int n_rows,
                                           int* row_ptr, void* stream) {
  OMNIHD_REQUIRE(n_points >= 0 && n_rows >= 0, "sizes");
  OMNIHD_REQUIRE(row_ptr && (sorted_keys || n_points == 0), "null pointer");
  hipLaunchKernelGGL(k_csr, dim3(grid_for((int64_t)n_rows + 1, kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, sorted_keys, n_points, n_rows, row_ptr);
  return check_launch("csr_from_sorted_keys");
}

extern "C" int omnihd_permute_rows_zyx_to_yxz(const int* rows_in, int64_t n, int nz, int ny,
                                              int nx, int* rows_out, void* stream) {
  OMNIHD_REQUIRE(n >= 0 && nz > 0 && ny > 0 && nx > 0, "sizes");
  if (n == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(rows_in && rows_out, "null pointer");
  hipLaunchKernelGGL(k_perm_zyx_yxz, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0,
                     (hipStream_t)stream, rows_in, n, nz, ny, nx, rows_out);
  return check_launch("permute_rows_zyx_to_yxz");
}

extern "C" size_t omnihd_csr_tiles_workspace_bytes(int n_rows) {
  if (n_rows <= 0) return 256;
  size_t b = 0;
  TileStart ts{nullptr, 1, 1};
  auto flags = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int>(0), ts);
  int* o = nullptr;
  hipError_t e = rocprim::select(nullptr, b, rocprim::make_counting_iterator<int>(0), flags, o, o,
                                 (size_t)n_rows, 0, false);
  if (e != hipSuccess) { set_error("csr_tiles: size query: %s", hipGetErrorString(e)); return 0; }
  return align_up(b, 256) + 256;
}

extern "C" int omnihd_csr_tiles(const int* row_ptr, int n_rows, int tile_items, int long_len,
                                int* tile_row, int* count, int* h_count, void* workspace,
                                size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  OMNIHD_REQUIRE(n_rows > 0 && tile_items >= 16 && long_len >= 1, "sizes");
  OMNIHD_REQUIRE(row_ptr && tile_row && count && workspace, "null pointer");
  TileStart ts{row_ptr, tile_items, long_len};
  auto flags = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int>(0), ts);
  size_t need = 0;
  int* o = nullptr;
  OMNIHD_HIP_TRY(rocprim::select(nullptr, need, rocprim::make_counting_iterator<int>(0), flags, o, o,
                                 (size_t)n_rows, st, false));
  if (workspace_bytes < need) {
    set_error("csr_tiles: workspace %zu < required %zu", workspace_bytes, need);
    return OMNIHD_ERR_WORKSPACE;
  }
  OMNIHD_HIP_TRY(rocprim::select(workspace, need, rocprim::make_counting_iterator<int>(0), flags,
                                 tile_row, count, (size_t)n_rows, st, false));
  hipLaunchKernelGGL(k_close_tiles, dim3(1), dim3(1), 0, st, tile_row, count, n_rows);
  int rc = check_launch("csr_tiles");
  if (rc != OMNIHD_OK) return rc;
  if (h_count) {
    OMNIHD_HIP_TRY(hipMemcpyAsync(h_count, count, sizeof(int), hipMemcpyDeviceToHost, st));
    OMNIHD_HIP_TRY(hipStreamSynchronize(st));
  }
  return OMNIHD_OK;
}
