// Pooling plan built ON THE DEVICE, with no host round trip (round 6).
//
// The reference rebuilds its rank tables in every forward (bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:283-300 ->
// voxel_pooling_prepare_v2 :302-362) because its camera calibration differs from sample to sample (lidar2img is composed per
// frame from the ego poses: datasets/newscenes_dataset.py:203-216), and re-sorts all points in every backward
// (ops/bev_pool_v2/bev_pool.py:47-57).  This file produces everything the dense forward (k_pool_fwd_direct) and the patch
// backward (k_pool_bwd_patch) read — for a NEW calibration — as one chain of launches on one stream:
//
//   keys          one pass: frustum point -> output row (or a sentinel), the geometry optionally fused in (no 48 MB tensor)
//   sort          rocPRIM LSD radix sort of (row, point index): stable = the canonical table order of SURVEY D6
//   row CSR       binary search per row
//   row scan      ONE 64-bit scan over the rows: (non-empty rows before | tile starts before << 32)
//   tiles         tile table, per-tile azimuth key + work, small radix sort, one-workgroup cut into 8 XCD runs
//   forward       pt / ivl_rel / desc32 as csrc/bev_pool_v2.hip documents them
//   backward      NO second sort: the points of an image pixel are its D depth bins, so a workgroup per 16-pixel patch orders
//                 the <= D surviving (row, bin) pairs of every pixel in LDS (= the stable re-sort by pixel of the row-sorted
//                 tables), after a count + scan pass gave the CSR over pixels
//   patches       one-workgroup cost-balanced schedule of the patch backward
//
// Every count the host used to read back (points, non-empty rows, tiles) stays in `hdr` on the device: buffers are sized by
// host-known upper bounds, grids likewise, and the consuming kernels read the counts themselves.
#include "common.h"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include <rocprim/functional.hpp>

// hipcc contracts a*b+c into one fused multiply-add by default (and __fmul_rn / __fadd_rn are plain operators in HIP): the fused
// geometry below must round after every step like the torch formulation it restates, so contraction is off for this file.
#pragma clang fp contract(off)

namespace omnihd {
namespace {

constexpr int kBlock = 256;
constexpr int kWide = 1024;          // the one-workgroup scheduling kernels
constexpr int kPatchPx = 16;         // pixels per patch of the patch backward (csrc/bev_pool_v2.hip: kPatch)
constexpr int kGroups = 16;          // lane groups per workgroup of k_pool_fwd_direct
constexpr int kPatchFixedCost = 400; // omnihd_amd/plan.py: PATCH_FIXED_COST

struct PlanGrid {
  float off[3];
  float dx[3];
  int nx[3];
};

struct FusedGeom {        // geometry of cam_stream_lss_bevpoolv2_depthnet.py:235-264 without post_* / extra_* transforms
  const float* rots;      // (B*N, 3, 3)
  const float* trans;     // (B*N, 3)
  const float* xs;        // (fW)  frustum u
  const float* ys;        // (fH)  frustum v
  const float* ds;        // (D)   frustum depth
  int D, fH, fW;
};

// ---- keys -------------------------------------------------------------------------------------------------------------
// Every fp32 step is its own rounding (no contraction), in the order of the torch formulation in
// projects/mmdet3d_plugin/bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py::get_geometry / _rotate:
// p = (u*d, v*d, d); g_a = ((R_a0*p0 + R_a1*p1) + R_a2*p2) + t_a.
template <bool FUSED>
__global__ __launch_bounds__(kBlock) void k_plan_keys(const float* __restrict__ geom, FusedGeom fg, int64_t n_total,
                                                      int64_t pts_per_batch, PlanGrid g, int yxz, uint32_t sentinel,
                                                      uint32_t* __restrict__ keys, int* __restrict__ idx) {
  // n_total < 2^31 (checked by the caller): 32-bit index arithmetic (64-bit divisions are software routines)
  const unsigned n = (unsigned)n_total, ppb = (unsigned)pts_per_batch;
  for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    float px, py, pz;
    if (FUSED) {
      const unsigned fhw = (unsigned)(fg.fH * fg.fW);
      const unsigned q = i / (unsigned)fg.fW;
      const int w = (int)(i - q * (unsigned)fg.fW);
      const int h = (int)(q % (unsigned)fg.fH);
      const unsigned qd = i / fhw;
      const int d = (int)(qd % (unsigned)fg.D);
      const int cam = (int)(qd / (unsigned)fg.D);
      const float dd = fg.ds[d];
      // plain operators: the contract(off) pragma governs THIS file's operations (the __fmul_rn / __fadd_rn of the HIP headers
      // are operators compiled under the default contraction mode and fuse all the same)
      const float p0 = fg.xs[w] * dd, p1 = fg.ys[h] * dd, p2 = dd;
      const float* R = fg.rots + (size_t)cam * 9;
      const float* T = fg.trans + (size_t)cam * 3;
      const float m00 = R[0] * p0, m01 = R[1] * p1, m02 = R[2] * p2;
      const float m10 = R[3] * p0, m11 = R[4] * p1, m12 = R[5] * p2;
      const float m20 = R[6] * p0, m21 = R[7] * p1, m22 = R[8] * p2;
      px = ((m00 + m01) + m02) + T[0];
      py = ((m10 + m11) + m12) + T[1];
      pz = ((m20 + m21) + m22) + T[2];
    } else {
      const float* p = geom + (size_t)i * 3;
      px = p[0]; py = p[1]; pz = p[2];
    }
    // the arithmetic of csrc/rank_prep.hip::k_rank_keys (reference :328-337; truncation toward zero = defect D3, NaN dropped)
    const float tx = (px - g.off[0]) / g.dx[0];
    const float ty = (py - g.off[1]) / g.dx[1];
    const float tz = (pz - g.off[2]) / g.dx[2];
    const bool kept = tx > -1.f && tx < (float)g.nx[0] && ty > -1.f && ty < (float)g.nx[1] && tz > -1.f && tz < (float)g.nx[2];
    uint32_t key = sentinel;
    if (kept) {
      const int x = (int)tx, y = (int)ty, z = (int)tz;
      const int64_t b = i / ppb;
      key = yxz ? (uint32_t)(((b * g.nx[1] + y) * g.nx[0] + x) * g.nx[2] + z)
                : (uint32_t)(((b * g.nx[2] + z) * g.nx[1] + y) * g.nx[0] + x);
    }
    keys[i] = key;
    idx[i] = (int)i;
  }
}

// first index with sorted[i] >= r, for r = 0 .. n_rows (the sentinel n_rows sorts behind every row: row_ptr[n_rows] = n_points)
__global__ __launch_bounds__(kBlock) void k_plan_csr(const uint32_t* __restrict__ sorted, int n_total, int n_rows,
                                                     int* __restrict__ row_ptr) {
  for (int r = blockIdx.x * kBlock + threadIdx.x; r <= n_rows; r += gridDim.x * kBlock) {
    int lo = 0, hi = n_total;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (sorted[mid] < (uint32_t)r) lo = mid + 1; else hi = mid;
    }
    row_ptr[r] = lo;
  }
}

// (non-empty | tile start << 32) per row; the tile rule is csrc/rank_prep.hip::TileStart
struct RowFlags {
  const int* row_ptr;
  int n_rows, tile_items, long_len;
  __host__ __device__ bool is_long(int r) const { return row_ptr[r + 1] - row_ptr[r] > long_len; }
  __host__ __device__ long long operator()(int r) const {
    if (r >= n_rows) return 0;
    const long long nonempty = row_ptr[r + 1] > row_ptr[r] ? 1 : 0;
    long long start;
    if (r == 0) start = 1;
    else if (is_long(r) || is_long(r - 1)) start = 1;
    else start = ((long long)r + row_ptr[r]) / tile_items != ((long long)r - 1 + row_ptr[r - 1]) / tile_items;
    return nonempty | (start << 32);
  }
};

// hdr: [0] points, [1] non-empty rows, [2] tiles, [3] tiles per XCD, [4] longest patch run, [5] status bits
__global__ __launch_bounds__(kBlock) void k_plan_tiles(const long long* __restrict__ S, const int* __restrict__ row_ptr,
                                                       int n_rows, int tiles_cap, int* __restrict__ tile_row,
                                                       int* __restrict__ hdr) {
  for (int r = blockIdx.x * kBlock + threadIdx.x; r <= n_rows; r += gridDim.x * kBlock) {
    const int tb = (int)(S[r] >> 32);
    if (r == n_rows) {
      const int n_tiles = tb < tiles_cap ? tb : tiles_cap;
      tile_row[n_tiles] = n_rows;
      hdr[0] = row_ptr[n_rows];
      hdr[1] = (int)(S[r] & 0xffffffffll);
      hdr[2] = n_tiles;
      hdr[3] = (n_tiles + 7) / 8;
      hdr[5] = tb > tiles_cap ? 1 : 0;      // cannot happen with the capacity of omnihd_pool_plan_sizes; reported, not hidden
    } else if ((int)(S[r + 1] >> 32) != tb && tb < tiles_cap) {
      tile_row[tb] = r;
    }
  }
}

// per tile: sortable azimuth of its middle BEV cell around the rig centre + its work (rows + points); beyond n_tiles: sentinel
__global__ __launch_bounds__(kBlock) void k_plan_tile_keys(const int* __restrict__ tile_row, const int* __restrict__ row_ptr,
                                                           const int* __restrict__ hdr, int tiles_cap, int n_rows, PlanGrid g,
                                                           int yxz, const float* __restrict__ trans, int n_cam,
                                                           uint32_t* __restrict__ tkey, int* __restrict__ tidx,
                                                           int* __restrict__ work) {
  const int n_tiles = hdr[2];
  float ox = 0.5f * (float)(g.nx[0] - 1), oy = 0.5f * (float)(g.nx[1] - 1);
  if (trans && n_cam > 0) {                              // centroid of the camera positions, in cells
    float sx = 0.f, sy = 0.f;
    for (int c = 0; c < n_cam; ++c) { sx += trans[c * 3]; sy += trans[c * 3 + 1]; }
    ox = (sx / (float)n_cam - g.off[0]) / g.dx[0] - 0.5f;
    oy = (sy / (float)n_cam - g.off[1]) / g.dx[1] - 0.5f;
  }
  for (int t = blockIdx.x * kBlock + threadIdx.x; t < tiles_cap; t += gridDim.x * kBlock) {
    uint32_t key = 0xffffffffu;
    int wk = 0;
    if (t < n_tiles) {
      const int ra = tile_row[t], rb = tile_row[t + 1];
      wk = (row_ptr[rb] - row_ptr[ra]) + (rb - ra);
      int mid = (int)(((long long)ra + rb) / 2);
      if (mid > n_rows - 1) mid = n_rows - 1;
      int xx, yy;
      if (yxz) { const int cell = mid / g.nx[2]; yy = (cell / g.nx[0]) % g.nx[1]; xx = cell % g.nx[0]; }
      else { yy = (mid / g.nx[0]) % g.nx[1]; xx = mid % g.nx[0]; }
      const float a = atan2f((float)yy - oy, (float)xx - ox);
      const uint32_t bits = __float_as_uint(a);
      key = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
      if (key == 0xffffffffu) key = 0xfffffffeu;
    }
    tkey[t] = key;
    tidx[t] = t;
    work[t] = wk;
  }
}

// exclusive scan of one value per thread over a 1024-thread workgroup; *total = the sum (same for every thread)
__device__ __forceinline__ long long block_scan_1024(long long v, long long* s_wave, long long* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const long long up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  __syncthreads();                       // s_wave may still be read by the previous call
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  long long base = 0, tot = 0;
  for (int w = 0; w < kWide / 64; ++w) {
    const long long x = s_wave[w];
    if (w < wave) base += x;
    tot += x;
  }
  *total = tot;
  return base + inc - v;
}

// cuts of an ordered list with inclusive cumulative cost `cum` into 8 runs of (nearly) equal cost, no run longer than
// `per`: cut[k] = first i with 8*cum[i] >= k*scale, clamped so that every run fits and the remainder still fits behind it
__device__ void cut_runs(const long long* cum, int n, long long scale, int per, int* cut) {
  cut[0] = 0;
  for (int k = 1; k < 8; ++k) {
    int lo = 0, hi = n;
    const long long target = (long long)k * scale;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (8 * cum[mid] < target) lo = mid + 1; else hi = mid;
    }
    int c = lo;
    const int least = max(cut[k - 1], n - (8 - k) * per);
    const int most = min(n, cut[k - 1] + per);
    c = max(c, least);
    c = min(c, most);
    cut[k] = c;
  }
  cut[8] = n;
}

// ONE workgroup: tiles in azimuth order -> 8 XCD runs of equal work; order[k * per + j] = tile (-1 idle), per = hdr[3]
__global__ __launch_bounds__(kWide) void k_plan_tile_sched(const int* __restrict__ stile, const int* __restrict__ work,
                                                           const int* __restrict__ hdr, long long* __restrict__ cw,
                                                           int* __restrict__ order) {
  __shared__ long long s_wave[kWide / 64];
  __shared__ int s_cut[9];
  const int n = hdr[2], per = hdr[3];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8 * per; i += kWide) order[i] = -1;
  const int chunk = (n + kWide - 1) / kWide;
  const int i0 = min(n, tid * chunk), i1 = min(n, i0 + chunk);
  long long mine = 0;
  for (int i = i0; i < i1; ++i) mine += work[stile[i]];
  long long total;
  long long run = block_scan_1024(mine, s_wave, &total);
  for (int i = i0; i < i1; ++i) { run += work[stile[i]]; cw[i] = run; }
  __syncthreads();
  if (tid == 0) cut_runs(cw, n, total + 1, per, s_cut);
  __syncthreads();
  for (int i = tid; i < n; i += kWide) {
    int k = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) k += (i >= s_cut[q]) ? 1 : 0;
    order[k * per + (i - s_cut[k])] = stile[i];
  }
}

// desc32 of k_pool_fwd_direct (csrc/bev_pool_v2.hip): one thread per (slot, lane group)
__global__ __launch_bounds__(kBlock) void k_plan_desc(const int* __restrict__ order, const int* __restrict__ tile_row,
                                                      const int* __restrict__ row_ptr, const uint32_t* __restrict__ sorted,
                                                      const long long* __restrict__ S, const int* __restrict__ hdr,
                                                      int slots_cap, int* __restrict__ desc32) {
  const int n_slots = 8 * hdr[3];
  for (int e = blockIdx.x * kBlock + threadIdx.x; e < slots_cap * kGroups; e += gridDim.x * kBlock) {
    const int s = e / kGroups, j = e % kGroups;
    if (s >= n_slots) continue;
    const int t = order[s];
    int* d = desc32 + (size_t)s * 32;
    if (t < 0) {
      d[j] = 0; d[kGroups + j] = 0;
      continue;
    }
    const int ra = tile_row[t], rb = tile_row[t + 1];
    const int pa = row_ptr[ra], npts = row_ptr[rb] - pa;
    const int w = (npts + kGroups - 1) / kGroups;
    const int off = min(j * w, npts);
    int gi = 0;
    if (off < npts) {
      const int q = pa + off;
      const uint32_t rq = sorted[q];
      gi = (int)(S[rq] & 0xffffffffll);                            // non-empty rows in front of q's row = rows closed before q
      if (off > 0 && sorted[q - 1] == rq) gi |= (int)0x80000000;   // q continues the row of the point in front of it
    }
    d[8 + j] = gi;
    if (j < 8) d[j] = j == 0 ? ra : j == 1 ? rb - ra : j == 2 ? pa : j == 3 ? npts : 0;
    else d[16 + j] = 0;
  }
}

__global__ __launch_bounds__(kBlock) void k_plan_pt(const uint32_t* __restrict__ sorted, const int* __restrict__ rd,
                                                    int64_t n_total, uint32_t sentinel, int* __restrict__ pt) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_total; i += (int64_t)gridDim.x * kBlock) {
    const uint32_t k = sorted[i];
    if (k == sentinel) continue;
    const bool closing = (i + 1 == n_total) || sorted[i + 1] != k;
    pt[i] = rd[i] | (closing ? (int)0x80000000 : 0);
  }
}

__global__ __launch_bounds__(kBlock) void k_plan_ivl(const long long* __restrict__ S, const int* __restrict__ tile_row,
                                                     int n_rows, int tiles_cap, int* __restrict__ ivl_rel) {
  for (int r = blockIdx.x * kBlock + threadIdx.x; r < n_rows; r += gridDim.x * kBlock) {
    const long long a = S[r], b = S[r + 1];
    if ((int)(b & 0xffffffffll) == (int)(a & 0xffffffffll)) continue;      // empty row
    const int t = min((int)(b >> 32), tiles_cap) - 1;
    ivl_rel[(int)(a & 0xffffffffll)] = r - tile_row[t];
  }
}

// ---- backward tables ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_plan_pix_count(const uint32_t* __restrict__ keys, int n_pix, int fhw, int d_bins,
                                                           uint32_t sentinel, int* __restrict__ cnt) {
  for (int f = blockIdx.x * kBlock + threadIdx.x; f <= n_pix; f += gridDim.x * kBlock) {
    int c = 0;
    if (f < n_pix) {
      const int img = f / fhw, hw = f - img * fhw;
      const uint32_t* p = keys + (size_t)img * d_bins * fhw + hw;
      for (int d = 0; d < d_bins; ++d) c += p[(size_t)d * fhw] != sentinel ? 1 : 0;
    }
    cnt[f] = c;
  }
}

// one workgroup per patch of 16 pixels: row_bin[pix_ptr[f] + rank] = row | bin << 24, rank by (row, bin)
__global__ __launch_bounds__(kBlock) void k_plan_pix_fill(const uint32_t* __restrict__ keys, const int* __restrict__ pix_ptr,
                                                          int fhw, int d_bins, int patches_per_img, uint32_t sentinel,
                                                          int* __restrict__ row_bin) {
  extern __shared__ int s_key[];                     // [d_bins][16]
  const int patch = blockIdx.x;
  const int img = patch / patches_per_img;
  const int hw0 = (patch - img * patches_per_img) * kPatchPx;
  const int npx = min(kPatchPx, fhw - hw0);
  const int tid = threadIdx.x;
  const uint32_t* base = keys + (size_t)img * d_bins * fhw + hw0;
  for (int i = tid; i < d_bins * kPatchPx; i += kBlock) {
    const int d = i / kPatchPx, px = i % kPatchPx;
    uint32_t k = sentinel;
    if (px < npx) k = base[(size_t)d * fhw + px];
    s_key[i] = k == sentinel ? 0x7fffffff : (int)((k << 7) | (uint32_t)d);
  }
  __syncthreads();
  const int grp = tid / kPatchPx, sub = tid % kPatchPx;
  if (grp >= npx) return;
  const int s = pix_ptr[img * fhw + hw0 + grp];
  for (int d = sub; d < d_bins; d += kPatchPx) {
    const int v = s_key[d * kPatchPx + grp];
    if (v == 0x7fffffff) continue;
    int rank = 0;
    for (int dd = 0; dd < d_bins; ++dd) rank += s_key[dd * kPatchPx + grp] < v ? 1 : 0;
    row_bin[s + rank] = (v >> 7) | ((v & 127) << 24);
  }
}

// ONE workgroup: the patch backward's schedule (omnihd_amd/plan.py::patch_schedule on the device).  `walk` lists the patches
// image by image in bands of 4 image rows, `wblock` the band of every walk position (both static per frustum shape).  The walk
// is cut into 8 runs of equal cost (points + a fixed cost per patch; no run longer than `per`), and inside a run the pieces
// of bands are issued heaviest mean cost first.  patch_order[k * per + j], -1 = idle.
__global__ __launch_bounds__(kWide) void k_plan_patch_sched(const int* __restrict__ walk, const int* __restrict__ wblock,
                                                            const int* __restrict__ pix_ptr, int n_patch, int patches_per_img,
                                                            int fhw, int per, long long* __restrict__ cum,
                                                            int* __restrict__ sid, int* __restrict__ seg_start,
                                                            int* __restrict__ seg_off, int* __restrict__ patch_order,
                                                            int* __restrict__ hdr) {
  __shared__ long long s_wave[kWide / 64];
  __shared__ int s_cut[9];
  __shared__ int s_first_seg[9];
  const int tid = threadIdx.x, n = n_patch;
  for (int i = tid; i < 8 * per; i += kWide) patch_order[i] = -1;
  auto cost_of = [&](int i) -> long long {
    const int p = walk[i];
    const int img = p / patches_per_img, hw0 = (p - img * patches_per_img) * kPatchPx;
    const int f0 = img * fhw + hw0, f1 = img * fhw + min(fhw, hw0 + kPatchPx);
    return (long long)(pix_ptr[f1] - pix_ptr[f0]) + kPatchFixedCost;
  };
  const int chunk = (n + kWide - 1) / kWide;
  const int i0 = min(n, tid * chunk), i1 = min(n, i0 + chunk);
  long long mine = 0;
  for (int i = i0; i < i1; ++i) mine += cost_of(i);
  long long total;
  long long run = block_scan_1024(mine, s_wave, &total);
  for (int i = i0; i < i1; ++i) { run += cost_of(i); cum[i] = run; }
  __syncthreads();
  if (tid == 0) {
    cut_runs(cum, n, total, per, s_cut);
    int longest = 0;
    for (int k = 0; k < 8; ++k) longest = max(longest, s_cut[k + 1] - s_cut[k]);
    hdr[4] = longest;
  }
  __syncthreads();
  // pieces: maximal stretches of one band inside one run
  auto run_of = [&](int i) { int k = 0; for (int q = 1; q < 8; ++q) k += (i >= s_cut[q]) ? 1 : 0; return k; };
  auto starts = [&](int i) { return i == 0 || wblock[i] != wblock[i - 1] || run_of(i) != run_of(i - 1); };
  long long flags = 0;
  for (int i = i0; i < i1; ++i) flags += starts(i) ? 1 : 0;
  long long n_seg_ll;
  long long before = block_scan_1024(flags, s_wave, &n_seg_ll);
  const int n_seg = (int)n_seg_ll;
  {
    int s = (int)before;
    for (int i = i0; i < i1; ++i) {
      if (starts(i)) { seg_start[s] = i; ++s; }
      sid[i] = s - 1;
    }
  }
  if (tid == 0) seg_start[n_seg] = n;
  __syncthreads();
  if (tid < 9) {                                   // first piece of every run (pieces are in walk order, runs are contiguous)
    int lo = 0, hi = n_seg;
    const int target = tid < 8 ? s_cut[tid] : n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (seg_start[mid] < target) lo = mid + 1; else hi = mid; }
    s_first_seg[tid] = lo;
  }
  __syncthreads();
  for (int s = tid; s < n_seg; s += kWide) {
    const int a = seg_start[s], b = seg_start[s + 1];
    const int k = run_of(a);
    const long long tot = cum[b - 1] - (a ? cum[a - 1] : 0), cnt = b - a;
    int off = 0;
    for (int o = s_first_seg[k]; o < s_first_seg[k + 1]; ++o) {
      if (o == s) continue;
      const int oa = seg_start[o], ob = seg_start[o + 1];
      const long long otot = cum[ob - 1] - (oa ? cum[oa - 1] : 0), ocnt = ob - oa;
      const long long lhs = otot * cnt, rhs = tot * ocnt;          // mean cost of o vs mean cost of s
      if (lhs > rhs || (lhs == rhs && o < s)) off += (int)ocnt;
    }
    seg_off[s] = off;
  }
  __syncthreads();
  for (int i = tid; i < n; i += kWide) {
    const int s = sid[i];
    const int k = run_of(i);
    patch_order[k * per + seg_off[s] + (i - seg_start[s])] = walk[i];
  }
}

struct PlanWs {
  size_t tmp_bytes;
  size_t off_keys, off_idx, off_S, off_tile_row, off_tkey, off_tidx, off_skey, off_stile, off_work, off_cw, off_order, off_cnt,
      off_cum, off_sid, off_seg_start, off_seg_off, total;
};

int plan_ws_layout(int64_t n_total, int n_rows, int n_pix, int tiles_cap, int n_patch, PlanWs* ws) {
  size_t a = 0, b = 0, c = 0, d = 0;
  uint32_t* k = nullptr;
  int* v = nullptr;
  long long* s = nullptr;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, a, k, k, v, v, (size_t)n_total, 0, 32, 0, false);
  if (e != hipSuccess) { set_error("pool_plan: radix_sort_pairs size query: %s", hipGetErrorString(e)); return OMNIHD_ERR_RUNTIME; }
  e = rocprim::radix_sort_pairs(nullptr, b, k, k, v, v, (size_t)tiles_cap, 0, 32, 0, false);
  if (e != hipSuccess) { set_error("pool_plan: radix_sort_pairs size query: %s", hipGetErrorString(e)); return OMNIHD_ERR_RUNTIME; }
  RowFlags rf{nullptr, 0, 1, 1};
  auto flags = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int>(0), rf);
  e = rocprim::exclusive_scan(nullptr, c, flags, s, 0ll, (size_t)n_rows + 1, rocprim::plus<long long>(), 0, false);
  if (e != hipSuccess) { set_error("pool_plan: exclusive_scan size query: %s", hipGetErrorString(e)); return OMNIHD_ERR_RUNTIME; }
  e = rocprim::exclusive_scan(nullptr, d, v, v, 0, (size_t)n_pix + 1, rocprim::plus<int>(), 0, false);
  if (e != hipSuccess) { set_error("pool_plan: exclusive_scan size query: %s", hipGetErrorString(e)); return OMNIHD_ERR_RUNTIME; }
  size_t m = a > b ? a : b;
  m = m > c ? m : c;
  m = m > d ? m : d;
  size_t o = align_up(m, 256) + 256;
  ws->tmp_bytes = o;
  auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes, 256); return at; };
  ws->off_keys = take((size_t)n_total * 4);
  ws->off_idx = take((size_t)n_total * 4);
  ws->off_S = take(((size_t)n_rows + 1) * 8);
  ws->off_tile_row = take(((size_t)tiles_cap + 1) * 4);
  ws->off_tkey = take((size_t)tiles_cap * 4);
  ws->off_tidx = take((size_t)tiles_cap * 4);
  ws->off_skey = take((size_t)tiles_cap * 4);
  ws->off_stile = take((size_t)tiles_cap * 4);
  ws->off_work = take((size_t)tiles_cap * 4);
  ws->off_cw = take((size_t)tiles_cap * 8);
  ws->off_order = take((size_t)tiles_cap * 4 + 64);
  ws->off_cnt = take(((size_t)n_pix + 1) * 4);
  ws->off_cum = take((size_t)n_patch * 8);
  ws->off_sid = take((size_t)n_patch * 4);
  ws->off_seg_start = take(((size_t)n_patch + 1) * 4);
  ws->off_seg_off = take((size_t)n_patch * 4);
  ws->total = o;
  return OMNIHD_OK;
}

long long tiles_capacity(long long n_total, long long n_rows, int tile_items, int long_len) {
  // regular cuts: one per `tile_items` of (rows + points); every long row opens at most two more tiles
  long long cap = (n_rows + n_total) / tile_items + 2 * (n_total / (long_len + 1)) + 2;
  return (cap + 7) / 8 * 8;
}

}  // namespace
}  // namespace omnihd

using namespace omnihd;

extern "C" int omnihd_pool_plan_sizes(long long n_total, int n_rows, int n_pix, int fhw, int tile_items, int long_len,
                                      long long* out4) {
  OMNIHD_REQUIRE(out4, "null pointer");
  OMNIHD_REQUIRE(n_total > 0 && n_total < (1ll << 31) && n_rows > 0 && n_pix > 0 && fhw > 0 && n_pix % fhw == 0 && tile_items >= 16 &&
                     long_len >= 1, "sizes");
  const long long tiles_cap = tiles_capacity(n_total, n_rows, tile_items, long_len);
  const int ppi = (fhw + kPatchPx - 1) / kPatchPx;
  const long long n_patch = (long long)(n_pix / fhw) * ppi;
  // no run of the patch schedule is longer than 1.5 x the even share (the cuts are clamped to it)
  const long long patch_per = ((n_patch + 7) / 8 * 3 + 1) / 2;
  PlanWs ws;
  const int rc = plan_ws_layout(n_total, n_rows, n_pix, (int)tiles_cap, (int)n_patch, &ws);
  if (rc != OMNIHD_OK) return rc;
  out4[0] = (long long)ws.total;
  out4[1] = tiles_cap;
  out4[2] = n_patch;
  out4[3] = patch_per;
  return OMNIHD_OK;
}

extern "C" int omnihd_pool_plan_build(const float* geom, const float* rots, const float* trans, const float* xs, const float* ys,
                                      const float* ds, int B, int N, int D, int fH, int fW, const float* h_off3,
                                      const float* h_dx3, const int* h_nx3, int layout_yxz, const int* walk, const int* wblock,
                                      int tile_items, int long_len, int* pt, int* ivl_rel, int* desc32, int* row_ptr,
                                      int* row_bin, int* pix_ptr, int* patch_order, int* hdr, uint32_t* rows_sorted,
                                      int* ranks_depth_sorted, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  OMNIHD_REQUIRE(B > 0 && N > 0 && D > 0 && fH > 0 && fW > 0 && tile_items >= 16 && long_len >= 1, "sizes");
  OMNIHD_REQUIRE(D <= 127, "the packed backward table holds at most 127 depth bins");
  OMNIHD_REQUIRE(geom || (rots && trans && xs && ys && ds), "either the geometry tensor or rots / trans / frustum axes");
  OMNIHD_REQUIRE(h_off3 && h_dx3 && h_nx3 && walk && wblock && pt && ivl_rel && desc32 && row_ptr && row_bin && pix_ptr &&
                     patch_order && hdr && rows_sorted && ranks_depth_sorted && workspace, "null pointer");
  const long long n_total = (long long)B * N * D * fH * fW;
  const long long n_rows_ll = (long long)B * h_nx3[0] * h_nx3[1] * h_nx3[2];
  OMNIHD_REQUIRE(n_total < (1ll << 31) && n_rows_ll < 0x00ffffff, "point indices must fit int32 and output rows 24 bits");
  const int n_rows = (int)n_rows_ll;
  const int fhw = fH * fW, n_pix = B * N * fhw;
  const int ppi = (fhw + kPatchPx - 1) / kPatchPx;
  const int n_patch = B * N * ppi;
  const int tiles_cap = (int)tiles_capacity(n_total, n_rows, tile_items, long_len);
  const int patch_per = (int)((((long long)n_patch + 7) / 8 * 3 + 1) / 2);
  PlanWs ws;
  int rc = plan_ws_layout(n_total, n_rows, n_pix, tiles_cap, n_patch, &ws);
  if (rc != OMNIHD_OK) return rc;
  if (workspace_bytes < ws.total) {
    set_error("pool_plan_build: workspace %zu < required %zu", workspace_bytes, ws.total);
    return OMNIHD_ERR_WORKSPACE;
  }
  char* base = static_cast<char*>(workspace);
  uint32_t* keys = reinterpret_cast<uint32_t*>(base + ws.off_keys);
  int* idx = reinterpret_cast<int*>(base + ws.off_idx);
  long long* S = reinterpret_cast<long long*>(base + ws.off_S);
  int* tile_row = reinterpret_cast<int*>(base + ws.off_tile_row);
  uint32_t* tkey = reinterpret_cast<uint32_t*>(base + ws.off_tkey);
  int* tidx = reinterpret_cast<int*>(base + ws.off_tidx);
  uint32_t* skey = reinterpret_cast<uint32_t*>(base + ws.off_skey);
  int* stile = reinterpret_cast<int*>(base + ws.off_stile);
  int* work = reinterpret_cast<int*>(base + ws.off_work);
  long long* cw = reinterpret_cast<long long*>(base + ws.off_cw);
  int* order = reinterpret_cast<int*>(base + ws.off_order);
  int* cnt = reinterpret_cast<int*>(base + ws.off_cnt);
  long long* cum = reinterpret_cast<long long*>(base + ws.off_cum);
  int* sid = reinterpret_cast<int*>(base + ws.off_sid);
  int* seg_start = reinterpret_cast<int*>(base + ws.off_seg_start);
  int* seg_off = reinterpret_cast<int*>(base + ws.off_seg_off);

  PlanGrid g;
  for (int a = 0; a < 3; ++a) { g.off[a] = h_off3[a]; g.dx[a] = h_dx3[a]; g.nx[a] = h_nx3[a]; }
  const uint32_t sentinel = (uint32_t)n_rows;
  int key_bits = 1;
  while (key_bits < 32 && (sentinel >> key_bits) != 0) ++key_bits;
  const int grid_pts = grid_for(n_total, kBlock * 4);
  const FusedGeom fg{rots, trans, xs, ys, ds, D, fH, fW};

  // 1. keys in point order (kept for the backward tables), 2. the stable sort = the forward tables
  if (geom)
    hipLaunchKernelGGL(k_plan_keys<false>, dim3(grid_pts), dim3(kBlock), 0, st, geom, fg, (int64_t)n_total,
                       (int64_t)(n_total / B), g, layout_yxz, sentinel, keys, idx);
  else
    hipLaunchKernelGGL(k_plan_keys<true>, dim3(grid_pts), dim3(kBlock), 0, st, geom, fg, (int64_t)n_total,
                       (int64_t)(n_total / B), g, layout_yxz, sentinel, keys, idx);
  size_t tmp = ws.tmp_bytes;
  OMNIHD_HIP_TRY(rocprim::radix_sort_pairs(base, tmp, keys, rows_sorted, idx, ranks_depth_sorted, (size_t)n_total, 0,
                                           (unsigned)key_bits, st, false));
  // 3. CSR over the rows, 4. the row scan, 5. tiles + counts
  hipLaunchKernelGGL(k_plan_csr, dim3(grid_for((int64_t)n_rows + 1, kBlock)), dim3(kBlock), 0, st, rows_sorted, (int)n_total,
                     n_rows, row_ptr);
  {
    RowFlags rf{row_ptr, n_rows, tile_items, long_len};
    auto flags = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int>(0), rf);
    tmp = ws.tmp_bytes;
    OMNIHD_HIP_TRY(rocprim::exclusive_scan(base, tmp, flags, S, 0ll, (size_t)n_rows + 1, rocprim::plus<long long>(), st, false));
  }
  hipLaunchKernelGGL(k_plan_tiles, dim3(grid_for((int64_t)n_rows + 1, kBlock)), dim3(kBlock), 0, st, S, row_ptr, n_rows,
                     tiles_cap, tile_row, hdr);
  // 6. tile schedule
  hipLaunchKernelGGL(k_plan_tile_keys, dim3(grid_for(tiles_cap, kBlock)), dim3(kBlock), 0, st, tile_row, row_ptr, hdr, tiles_cap,
                     n_rows, g, layout_yxz, geom ? nullptr : trans, B * N, tkey, tidx, work);
  tmp = ws.tmp_bytes;
  OMNIHD_HIP_TRY(rocprim::radix_sort_pairs(base, tmp, tkey, skey, tidx, stile, (size_t)tiles_cap, 0, 32, st, false));
  hipLaunchKernelGGL(k_plan_tile_sched, dim3(1), dim3(kWide), 0, st, stile, work, hdr, cw, order);
  // 7. forward tables
  hipLaunchKernelGGL(k_plan_desc, dim3(grid_for((int64_t)tiles_cap * kGroups, kBlock)), dim3(kBlock), 0, st, order, tile_row,
                     row_ptr, rows_sorted, S, hdr, tiles_cap, desc32);
  hipLaunchKernelGGL(k_plan_pt, dim3(grid_pts), dim3(kBlock), 0, st, rows_sorted, ranks_depth_sorted, (int64_t)n_total, sentinel,
                     pt);
  hipLaunchKernelGGL(k_plan_ivl, dim3(grid_for(n_rows, kBlock)), dim3(kBlock), 0, st, S, tile_row, n_rows, tiles_cap, ivl_rel);
  // 8. backward tables: count, scan, fill
  hipLaunchKernelGGL(k_plan_pix_count, dim3(grid_for((int64_t)n_pix + 1, kBlock)), dim3(kBlock), 0, st, keys, n_pix, fhw, D,
                     sentinel, cnt);
  tmp = ws.tmp_bytes;
  OMNIHD_HIP_TRY(rocprim::exclusive_scan(base, tmp, cnt, pix_ptr, 0, (size_t)n_pix + 1, rocprim::plus<int>(), st, false));
  hipLaunchKernelGGL(k_plan_pix_fill, dim3(n_patch), dim3(kBlock), (size_t)D * kPatchPx * sizeof(int), st, keys, pix_ptr, fhw, D,
                     ppi, sentinel, row_bin);
  // 9. patch schedule
  hipLaunchKernelGGL(k_plan_patch_sched, dim3(1), dim3(kWide), 0, st, walk, wblock, pix_ptr, n_patch, ppi, fhw, patch_per, cum,
                     sid, seg_start, seg_off, patch_order, hdr);
  return check_launch("pool_plan_build");
}
