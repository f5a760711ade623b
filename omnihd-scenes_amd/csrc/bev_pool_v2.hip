// bev_pool_v2 forward / backward for gfx950 (MI355X).
//
// What the reference computes: ops/bev_pool_v2/src/bev_pool_cuda.cu:21-48 (forward, one CUDA
// thread per (interval, channel) with a serial loop) and :67-121 (backward, one thread per
// interval doing len*C serial work twice).  This file is a different program for the same
// arithmetic:
//
//   * a ROW (C floats of one image-feature pixel, or of one BEV voxel) is owned by a group of
//     C/4 lanes, each lane holding a float4 -> every feature gather / voxel write is one
//     fully coalesced 16 B-per-lane access (256 B for C=64), 4 rows per 64-wide wavefront;
//   * the three rank tables are read coalesced, C/4 points at a time, one point per lane, the
//     per-point depth value is gathered by that lane in parallel, and rank / depth are
//     handed to the row's other lanes through a sub-wave shuffle (no LDS round trip);
//   * intervals longer than kLongLen (near-ego voxels collect thousands of frustum points)
//     are parked in an LDS list and afterwards split over all groups of the workgroup, with
//     a fixed-order LDS combine -> no atomics anywhere, results are run-to-run identical;
//   * the dense (CSR) forward writes every output row, so the caller's zero-fill pass, the
//     permute copy and the s2c concat copy of the reference disappear.
//
// Arithmetic: one fmaf per (point, channel) in table order, i.e. the same rounding chain as
// the reference's `psum += feat * depth` under nvcc's default contraction.
#include "common.h"
#include <stdlib.h>

// (The traffic-attribution, ablation and phase-timeline builds are PATCHES against this file, scripts/lab/patches/*.patch, applied
// by scripts/build_abl.sh / scripts/lab/build_patched.sh; the product source carries no instrumentation.)

namespace omnihd {
namespace {

constexpr int kBlock = 256;
constexpr int kLongLen = 512;  // intervals longer than this are split over the workgroup
constexpr int kMaxLong = 48;   // capacity of the per-workgroup deferred list

__device__ __forceinline__ float4 fma4(float s, float4 v, float4 a) {
  a.x = fmaf(v.x, s, a.x);
  a.y = fmaf(v.y, s, a.y);
  a.z = fmaf(v.z, s, a.z);
  a.w = fmaf(v.w, s, a.w);
  return a;
}

// Accumulate points [start, start+len) of the tables into one float4 (4 channels of one row).
// All C4 lanes of the group call this together with the same (start, len).
template <int C4>
__device__ __forceinline__ float4 pool_range(const float* __restrict__ depth,
                                             const float4* __restrict__ feat4,
                                             const int* __restrict__ ranks_depth,
                                             const int* __restrict__ ranks_feat, int start,
                                             int len, int sub) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int base = 0; base < len; base += C4) {
    const int mine = base + sub;
    int my_rf = 0;
    float my_d = 0.f;
    if (mine < len) {
      my_rf = ranks_feat[start + mine];
      my_d = depth[ranks_depth[start + mine]];
    }
    const int n = min(C4, len - base);
    int j = 0;
    // 4 independent feature-row gathers in flight per group.
    for (; j + 4 <= n; j += 4) {
      const int f0 = __shfl(my_rf, j + 0, C4), f1 = __shfl(my_rf, j + 1, C4);
      const int f2 = __shfl(my_rf, j + 2, C4), f3 = __shfl(my_rf, j + 3, C4);
      const float d0 = __shfl(my_d, j + 0, C4), d1 = __shfl(my_d, j + 1, C4);
      const float d2 = __shfl(my_d, j + 2, C4), d3 = __shfl(my_d, j + 3, C4);
      const float4 v0 = feat4[(size_t)f0 * C4 + sub];
      const float4 v1 = feat4[(size_t)f1 * C4 + sub];
      const float4 v2 = feat4[(size_t)f2 * C4 + sub];
      const float4 v3 = feat4[(size_t)f3 * C4 + sub];
      acc = fma4(d0, v0, acc);
      acc = fma4(d1, v1, acc);
      acc = fma4(d2, v2, acc);
      acc = fma4(d3, v3, acc);
    }
    for (; j < n; ++j) {
      const int f0 = __shfl(my_rf, j, C4);
      const float d0 = __shfl(my_d, j, C4);
      acc = fma4(d0, feat4[(size_t)f0 * C4 + sub], acc);
    }
  }
  return acc;
}

typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_row(float4* p, float4 v, bool streaming) {
  if (streaming) {
    // one global_store_dwordx4 ... nt: the BEV tensor is written once and never re-read here
    f4v t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(p));
  } else {
    *p = v;
  }
}

// DENSE == false: unit = interval (tables from the reference API), only named rows written.
// DENSE == true : unit = output row r with points [row_ptr[r], row_ptr[r+1]), all rows written.
template <int C4, bool DENSE>
__global__ __launch_bounds__(kBlock) void k_pool_fwd(
    const float* __restrict__ depth, const float4* __restrict__ feat4,
    const int* __restrict__ ranks_depth, const int* __restrict__ ranks_feat,
    const int* __restrict__ ranks_bev, const int* __restrict__ starts,
    const int* __restrict__ lengths_or_rowptr, float4* __restrict__ out4, int n_units) {
  constexpr int G = kBlock / C4;  // rows in flight per workgroup
  __shared__ int s_long[kMaxLong];
  __shared__ int s_nlong;
  __shared__ float4 s_part[kBlock];
  const int tid = threadIdx.x;
  const int sub = tid % C4;
  const int grp = tid / C4;
  if (tid == 0) s_nlong = 0;
  __syncthreads();

  for (int base = blockIdx.x * G; base < n_units; base += gridDim.x * G) {
    const int u = base + grp;
    if (u >= n_units) continue;
    int s, len, row;
    if (DENSE) {
      s = lengths_or_rowptr[u];
      len = lengths_or_rowptr[u + 1] - s;
      row = u;
    } else {
      s = starts[u];
      len = lengths_or_rowptr[u];
      if (len <= 0) continue;
      row = ranks_bev[s];
    }
    if (len > kLongLen) {
      int slot = 0;
      if (sub == 0) slot = atomicAdd(&s_nlong, 1);
      slot = __shfl(slot, 0, C4);
      if (slot < kMaxLong) {
        if (sub == 0) s_long[slot] = u;
        continue;
      }
    }
    const float4 acc = pool_range<C4>(depth, feat4, ranks_depth, ranks_feat, s, len, sub);
    store_row(out4 + (size_t)row * C4 + sub, acc, DENSE);
  }
  __syncthreads();

  const int n_long = min(s_nlong, kMaxLong);
  for (int k = 0; k < n_long; ++k) {
    const int u = s_long[k];
    int s, len, row;
    if (DENSE) {
      s = lengths_or_rowptr[u];
      len = lengths_or_rowptr[u + 1] - s;
      row = u;
    } else {
      s = starts[u];
      len = lengths_or_rowptr[u];
      row = ranks_bev[s];
    }
    const int chunk = (len + G - 1) / G;
    const int a = min(grp * chunk, len);
    const int b = min(a + chunk, len);
    s_part[tid] = pool_range<C4>(depth, feat4, ranks_depth, ranks_feat, s + a, b - a, sub);
    __syncthreads();
    if (grp == 0) {
      float4 t = s_part[sub];
      for (int g = 1; g < G; ++g) {
        const float4 p = s_part[g * C4 + sub];
        t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
      }
      store_row(out4 + (size_t)row * C4 + sub, t, DENSE);
    }
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------------------------
// Tiled dense forward.
//
// A TILE is a run of whole output rows that holds at most kCap points, or one single row of any
// length; the plan builds tiles of ~tile_items points+rows so every workgroup gets the same
// amount of work although 40 % of the BEV rows are empty and a few near-ego rows hold thousands
// of points.  The launch schedule is an array of 16-byte descriptors {first row, #rows, first
// point, #points}, one per (XCD, slot): workgroup b runs on XCD b % 8 (observed dispatch rule,
// used for locality only), and the plan orders the schedule so that one XCD works on tiles that
// gather from the same image columns — its 4 MiB L2 then holds the feature rows it needs
// (measured: 24 TB/s of 256-byte row gathers from an L2-resident table vs 9 TB/s from the
// Infinity Cache).  One workgroup per tile, phases:
//   L  ALL 256 lanes read the tile's rank tables coalesced and gather the depth values 256-wide;
//      they land in LDS as { pixel-row index | last-point-of-row flag, depth, output row };
//   Z  meanwhile the empty rows of the tile are zero-filled (row_ptr read coalesced);
//   P  the points are cut into G equal pieces, one per group of C4 lanes.  A group reads one
//      record (broadcast LDS read), gathers the 16 B x C4 feature row (U gathers in flight),
//      accumulates, and on a last-point flag stores the finished row (256 B for C=64).
//      A row cut by a piece boundary leaves partials in LDS which are added in piece order after
//      one barrier.
// 19 KiB of LDS and <= 64 VGPRs: 8 workgroups (32 waves) per CU hide the dependent latencies.
// No atomics, no dependence on dispatch order: results are run-to-run identical.
// ---------------------------------------------------------------------------------------------
constexpr int kCap = 1280;   // LDS point records per workgroup

__device__ __forceinline__ float4 add4(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}


template <int C4, int U>
__global__ __launch_bounds__(kBlock) void k_pool_fwd_tiles(
    const float* __restrict__ depth, const float4* __restrict__ feat4,
    const int* __restrict__ ranks_depth, const int* __restrict__ ranks_feat,
    const int* __restrict__ ranks_row, const int* __restrict__ row_ptr,
    const int4* __restrict__ tile_desc, float4* __restrict__ out4, int tiles_per_xcd,
    int n_points_total) {
  constexpr int G = kBlock / C4;
  constexpr int GPW = 64 / C4;   // groups per wavefront
  constexpr int kRecInts = kCap * 3;
  // s_mem: [0, 2*kCap) int2 {rf|last, depth}; [2*kCap, 3*kCap) output row; the tail partials
  // (kBlock float4 = 1024 ints) reuse the front of the record area after a barrier.
  __shared__ int s_mem[kRecInts > kBlock * 4 ? kRecInts : kBlock * 4];
  __shared__ float4 s_head[kBlock];
  __shared__ int s_head_row[G];
  __shared__ int s_tail_row[G];
  int2* s_rfd = reinterpret_cast<int2*>(s_mem);
  int* s_row = s_mem + 2 * kCap;
  float4* s_tail = reinterpret_cast<float4*>(s_mem);

  if ((int)(blockIdx.x >> 3) >= tiles_per_xcd) return;
  const int4 desc = tile_desc[(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3)];
  const int Ra = desc.x, nrows = desc.y, Pa = desc.z, npts = desc.w;
  if (nrows <= 0) return;

  const int tid = threadIdx.x;
  const int sub = tid % C4;
  const int grp = tid / C4;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool long_row = (nrows == 1 && npts > kCap);
  const bool staged = (npts > 0 && npts <= kCap);

  // ---- phase L (issue): rank tables -> registers ------------------------------------------------
  constexpr int kPer = (kCap + kBlock - 1) / kBlock;   // records per lane
  int l_rf[kPer], l_rd[kPer], l_row[kPer], l_nxt[kPer];
  if (staged) {
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int i = tid + k * kBlock;
      if (i < npts) {
        const int q = Pa + i;
        l_rf[k] = ranks_feat[q];
        l_rd[k] = ranks_depth[q];
        l_row[k] = ranks_row[q];
        l_nxt[k] = (q + 1 < n_points_total) ? ranks_row[q + 1] : -1;
      }
    }
  }

  // ---- phase Z: zero-fill the empty rows (stores only; overlaps the loads above) ---------------
  if (!(nrows == 1 && npts > 0)) {
    const int lane = tid & 63;
    const int gw = lane / C4;
    for (int base = 0; base < nrows; base += kBlock) {
      const int i = base + tid;
      bool empty = false;
      if (i < nrows) empty = row_ptr[Ra + i + 1] == row_ptr[Ra + i];
      const unsigned long long m = __ballot(empty);
      if (m == 0ull) continue;
      const int wave_row0 = Ra + base + (tid & ~63);
      for (int k = 0; k < 64; k += GPW) {
        const unsigned long long window = (GPW >= 64) ? m : ((m >> k) & ((1ull << GPW) - 1ull));
        if (window == 0ull) continue;
        if ((m >> (k + gw)) & 1ull)
          store_row(out4 + (size_t)(wave_row0 + k + gw) * C4 + sub, zero4, true);
      }
    }
  }
  if (npts == 0) return;

  float4 acc = zero4;

  // ---- a single long row: windows of kCap points, every group accumulates, one combine --------
  if (long_row) {
    for (int base = 0; base < npts; base += kCap) {
      const int n = min(kCap, npts - base);
      for (int i = tid; i < n; i += kBlock) {
        const int q = Pa + base + i;
        s_rfd[i] = make_int2(ranks_feat[q], __float_as_int(depth[ranks_depth[q]]));
      }
      __syncthreads();
      const int cw = (n + G - 1) / G;
      const int j0 = min(grp * cw, n), j1 = min(j0 + cw, n);
      for (int j = j0; j < j1; j += U) {
        float4 v[U];
        float d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int2 rc = s_rfd[min(j + u, j1 - 1)];
          d[u] = (j + u < j1) ? __int_as_float(rc.y) : 0.f;
          v[u] = feat4[(size_t)rc.x * C4 + sub];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = fma4(d[u], v[u], acc);
      }
      __syncthreads();
    }
    s_tail[tid] = acc;
    __syncthreads();
    if (grp == 0) {
      float4 tsum = s_tail[sub];
      for (int g = 1; g < G; ++g) tsum = add4(tsum, s_tail[g * C4 + sub]);
      store_row(out4 + (size_t)Ra * C4 + sub, tsum, true);
    }
    return;
  }

  // ---- a tile that does not fit the LDS window (only for foreign tile tables): row by row -----
  if (!staged) {
    for (int r = Ra + grp; r < Ra + nrows; r += G) {
      const int s0 = row_ptr[r], len = row_ptr[r + 1] - s0;
      if (len > 0)
        store_row(out4 + (size_t)r * C4 + sub,
                  pool_range<C4>(depth, feat4, ranks_depth, ranks_feat, s0, len, sub), true);
    }
    return;
  }

  // ---- phase L (finish): depth gather, records -> LDS -------------------------------------------
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int i = tid + k * kBlock;
    if (i < npts) {
      const float dv = depth[l_rd[k]];
      const int last = (l_row[k] != l_nxt[k]) ? (int)0x80000000 : 0;
      s_rfd[i] = make_int2(l_rf[k] | last, __float_as_int(dv));
      s_row[i] = l_row[k];
    }
  }
  if (tid < G) s_head_row[tid] = -1;
  __syncthreads();

  // ---- phase P: equal pieces of the point list, one per group ----------------------------------
  const int w = (npts + G - 1) / G;
  const int i0 = min(grp * w, npts);
  const int i1 = min(i0 + w, npts);
  // my first point continues a row that an earlier piece started
  bool head_pending = (i0 > 0) && (i0 < i1) && (s_rfd[i0 - 1].x >= 0);
  for (int i = i0; i < i1; i += U) {
    float4 v[U];
    float d[U];
    int fl[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int2 rc = s_rfd[min(i + u, i1 - 1)];
      d[u] = __int_as_float(rc.y);
      fl[u] = rc.x;
      v[u] = feat4[(size_t)(rc.x & 0x7fffffff) * C4 + sub];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i + u < i1) {
        acc = fma4(d[u], v[u], acc);
        if (fl[u] < 0) {   // this point closes its output row
          const int row = s_row[i + u];
          if (head_pending) {
            s_head[tid] = acc;
            if (sub == 0) s_head_row[grp] = row;
            head_pending = false;
          } else {
            store_row(out4 + (size_t)row * C4 + sub, acc, true);
          }
          acc = zero4;
        }
      }
    }
  }
  const int tail_row = (i1 > i0 && s_rfd[i1 - 1].x >= 0) ? s_row[i1 - 1] : -2;
  __syncthreads();   // every group is done with the records: their LDS is reused for the tails
  s_tail[tid] = acc;
  if (sub == 0) s_tail_row[grp] = tail_row;
  __syncthreads();

  const int hr = s_head_row[grp];
  if (hr >= 0) {
    int g0 = grp;
    while (g0 > 0 && s_tail_row[g0 - 1] == hr) --g0;
    float4 tsum = zero4;
    for (int g = g0; g < grp; ++g) tsum = add4(tsum, s_tail[g * C4 + sub]);
    tsum = add4(tsum, s_head[tid]);
    store_row(out4 + (size_t)hr * C4 + sub, tsum, true);
  }
}

// ---------------------------------------------------------------------------------------------
// "Two-table" variant of the tiled forward: reads ONE per-point table (ranks_depth) instead of three.
//   * the pixel row of a point is a function of its depth index: rf = (rd / (D*fHW)) * fHW + rd % fHW
//     (two divisions by launch constants, done in fp32 with an exact fix-up);
//   * which point closes which output row comes from the CSR boundaries the zero-fill phase reads anyway:
//     row r closes at point row_ptr[r+1]-1, so the lane that looks at row r sets the flag (and the row id)
//     of that one record after the records are in LDS (one extra barrier).
// Per launch at R1 that is 16 MB less table traffic (225 -> 209 MB) and one load stream instead of four in phase L.
// Partials of rows cut by a piece boundary are combined by adjacency (piece g's open tail belongs to the row that a
// later piece closes first), so no per-point row id is needed at all.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int div_const(int x, int d, float inv) {
  int q = (int)((float)x * inv);
  int r = x - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) { ++q; }
  return q;
}

template <int C4, int U>
__global__ __launch_bounds__(kBlock) void k_pool_fwd_lean(
    const float* __restrict__ depth, const float4* __restrict__ feat4, const int* __restrict__ ranks_depth,
    const int* __restrict__ row_ptr, const int4* __restrict__ tile_desc, float4* __restrict__ out4,
    int tiles_per_xcd, int fhw, int dfhw, float inv_fhw, float inv_dfhw) {
  constexpr int G = kBlock / C4;
  constexpr int GPW = 64 / C4;
  constexpr int kRecInts = kCap * 3;
  constexpr int kRowsPerLane = 3;                  // a staged tile holds <= 768 rows
  __shared__ int s_mem[kRecInts > kBlock * 4 ? kRecInts : kBlock * 4];
  __shared__ float4 s_head[kBlock];
  __shared__ int s_head_row[G];
  __shared__ int s_tail_flags[G];                  // bit0: piece ends inside a row, bit1: piece closes no row
  int2* s_rfd = reinterpret_cast<int2*>(s_mem);
  int* s_row = s_mem + 2 * kCap;
  float4* s_tail = reinterpret_cast<float4*>(s_mem);

  if ((int)(blockIdx.x >> 3) >= tiles_per_xcd) return;
  const int4 desc = tile_desc[(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3)];
  const int Ra = desc.x, nrows = desc.y, Pa = desc.z, npts = desc.w;
  if (nrows <= 0) return;

  const int tid = threadIdx.x;
  const int sub = tid % C4;
  const int grp = tid / C4;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool long_row = (nrows == 1 && npts > kCap);
  const bool staged = (npts > 0 && npts <= kCap && nrows <= kRowsPerLane * kBlock);

  auto pixel_row = [&](int rd) {
    const int n = div_const(rd, dfhw, inv_dfhw);
    const int q = div_const(rd, fhw, inv_fhw);
    return n * fhw + (rd - q * fhw);
  };

  // ---- phase L (issue): ONE rank table -> registers -------------------------------------------
  constexpr int kPer = (kCap + kBlock - 1) / kBlock;
  int l_rd[kPer];
  if (staged) {
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int i = tid + k * kBlock;
      if (i < npts) l_rd[k] = ranks_depth[Pa + i];
    }
  }

  // ---- phase Z: zero-fill the empty rows; remember where the non-empty rows of this lane close --
  int l_close[kRowsPerLane], l_crow[kRowsPerLane];
#pragma unroll
  for (int j = 0; j < kRowsPerLane; ++j) l_close[j] = -1;
  if (!(nrows == 1 && npts > 0)) {
    const int lane = tid & 63;
    const int gw = lane / C4;
    int j = 0;
    for (int base = 0; base < nrows; base += kBlock, ++j) {
      const int i = base + tid;
      bool empty = false;
      if (i < nrows) {
        const int s0 = row_ptr[Ra + i], e0 = row_ptr[Ra + i + 1];
        empty = e0 == s0;
        if (!empty && j < kRowsPerLane) {
          l_close[j] = e0 - 1 - Pa;
          l_crow[j] = Ra + i;
        }
      }
      const unsigned long long m = __ballot(empty);
      if (m == 0ull) continue;
      const int wave_row0 = Ra + base + (tid & ~63);
      for (int k = 0; k < 64; k += GPW) {
        const unsigned long long window = (GPW >= 64) ? m : ((m >> k) & ((1ull << GPW) - 1ull));
        if (window == 0ull) continue;
        if ((m >> (k + gw)) & 1ull)
          store_row(out4 + (size_t)(wave_row0 + k + gw) * C4 + sub, zero4, true);
      }
    }
  } else if (tid == 0) {
    l_close[0] = npts - 1;                         // the single row of the tile closes at its last point
    l_crow[0] = Ra;
  }
  if (npts == 0) return;

  float4 acc = zero4;

  // ---- a single long row: windows of kCap points, every group accumulates, one combine --------
  if (long_row) {
    for (int base = 0; base < npts; base += kCap) {
      const int n = min(kCap, npts - base);
      for (int i = tid; i < n; i += kBlock) {
        const int rd = ranks_depth[Pa + base + i];
        s_rfd[i] = make_int2(pixel_row(rd), __float_as_int(depth[rd]));
      }
      __syncthreads();
      const int cw = (n + G - 1) / G;
      const int j0 = min(grp * cw, n), j1 = min(j0 + cw, n);
      for (int j = j0; j < j1; j += U) {
        float4 v[U];
        float d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int2 rc = s_rfd[min(j + u, j1 - 1)];
          d[u] = (j + u < j1) ? __int_as_float(rc.y) : 0.f;
          v[u] = feat4[(size_t)rc.x * C4 + sub];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = fma4(d[u], v[u], acc);
      }
      __syncthreads();
    }
    s_tail[tid] = acc;
    __syncthreads();
    if (grp == 0) {
      float4 tsum = s_tail[sub];
      for (int g = 1; g < G; ++g) tsum = add4(tsum, s_tail[g * C4 + sub]);
      store_row(out4 + (size_t)Ra * C4 + sub, tsum, true);
    }
    return;
  }

  // ---- a tile that does not fit the LDS window (only for foreign tile tables): row by row -----
  if (!staged) {
    for (int r = Ra + grp; r < Ra + nrows; r += G) {
      const int s0 = row_ptr[r], len = row_ptr[r + 1] - s0;
      if (len <= 0) continue;
      float4 a4 = zero4;
      for (int q = s0; q < s0 + len; ++q) {
        const int rd = ranks_depth[q];
        a4 = fma4(depth[rd], feat4[(size_t)pixel_row(rd) * C4 + sub], a4);
      }
      store_row(out4 + (size_t)r * C4 + sub, a4, true);
    }
    return;
  }

  // ---- phase L (finish): pixel row, depth gather, records -> LDS --------------------------------
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int i = tid + k * kBlock;
    if (i < npts) {
      s_rfd[i] = make_int2(pixel_row(l_rd[k]), __float_as_int(depth[l_rd[k]]));
    }
  }
  if (tid < G) s_head_row[tid] = -1;
  __syncthreads();
  // ---- closing flags + row ids from the CSR boundaries (each closing record has exactly one owner) ----
#pragma unroll
  for (int j = 0; j < kRowsPerLane; ++j)
    if (l_close[j] >= 0) {
      s_rfd[l_close[j]].x |= (int)0x80000000;
      s_row[l_close[j]] = l_crow[j];
    }
  __syncthreads();

  // ---- phase P: equal pieces of the point list, one per group ----------------------------------
  const int w = (npts + G - 1) / G;
  const int i0 = min(grp * w, npts);
  const int i1 = min(i0 + w, npts);
  bool head_pending = (i0 > 0) && (i0 < i1) && (s_rfd[i0 - 1].x >= 0);
  bool closed_any = false;
  for (int i = i0; i < i1; i += U) {
    float4 v[U];
    float d[U];
    int fl[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int2 rc = s_rfd[min(i + u, i1 - 1)];
      d[u] = __int_as_float(rc.y);
      fl[u] = rc.x;
      v[u] = feat4[(size_t)(rc.x & 0x7fffffff) * C4 + sub];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i + u < i1) {
        acc = fma4(d[u], v[u], acc);
        if (fl[u] < 0) {   // this point closes its output row
          const int row = s_row[i + u];
          closed_any = true;
          if (head_pending) {
            s_head[tid] = acc;
            if (sub == 0) s_head_row[grp] = row;
            head_pending = false;
          } else {
            store_row(out4 + (size_t)row * C4 + sub, acc, true);
          }
          acc = zero4;
        }
      }
    }
  }
  // piece ends inside a row (or is empty: then it is "inside" whatever row surrounds it, with a zero partial)
  const bool open_end = (i1 <= i0) || (s_rfd[i1 - 1].x >= 0);
  __syncthreads();
  s_tail[tid] = acc;
  if (sub == 0) s_tail_flags[grp] = (open_end ? 1 : 0) | (closed_any ? 0 : 2);
  __syncthreads();

  const int hr = s_head_row[grp];
  if (hr >= 0) {
    // the row I close first started in earlier pieces: add their open tails, walking back through pieces that lie
    // entirely inside the row and stopping after the first one that closed a row of its own
    int g0 = grp;
    while (g0 > 0) {
      const int f = s_tail_flags[g0 - 1];
      if (!(f & 1)) break;
      --g0;
      if (!(f & 2)) break;
    }
    float4 tsum = zero4;                             // point order, like the three-table kernel
    for (int g = g0; g < grp; ++g) tsum = add4(tsum, s_tail[g * C4 + sub]);
    tsum = add4(tsum, s_head[tid]);
    store_row(out4 + (size_t)hr * C4 + sub, tsum, true);
  }
}

// ---------------------------------------------------------------------------------------------
// k_pool_fwd_lean2: the same tiles, plan and one-table phase L as k_pool_fwd_lean, with the point loop rewritten for
// ISSUE cost.  Phase timelines of k_pool_fwd_lean (scripts/lab/pool_trace.py, wall_clock64 stamps per workgroup) showed the
// point loop taking 10.6 of a workgroup's 20.2 us — and still 6.0 us with every feature gather AND every store compiled
// out: the loop was bound by instruction issue (about 40 wave-instructions per step of 4 points, most of them exec-mask
// bookkeeping for bounds checks and the three-way "first close of a continued row / later close / no close" branch), not
// by memory.  Changes:
//   * every group runs the SAME number of full steps: the record list is padded to G * Wp entries whose pixel offset lies
//     outside the feature buffer — a raw buffer load returns zeros there (hardware range check), so a pad point adds
//     0 * 0 and no bounds test exists in the loop; the trip count is a scalar;
//   * gathers are `buffer_load_dwordx4 ... offen` with a 32-bit offset (pixel << 8 | lane*16: ONE v_lshl_or_b32, the
//     closing flag in bit 31 shifts out) instead of 64-bit address arithmetic;
//   * the closing flag and the output row travel together in ONE LDS word (s_row[i] = row | 1<<31, zero otherwise), read
//     four at a time with one ds_read_b128; two ds_read_b128 fetch four {pixel, depth} records;
//   * the head partial of a row continued from an earlier piece stays in registers (no LDS head slots);
//   * the closing words are written by the lanes that look at row_ptr in the zero-fill phase BEFORE the depth gather
//     returns (the LDS row words are zeroed behind an early barrier), so the second barrier of the old kernel and its
//     phase are gone.
// Results: same tiles, same per-piece fma order, pieces of Wp = roundup(ceil(n/G), U) points (the old kernel: ceil(n/G)),
// so rows cut by a piece boundary may differ from k_pool_fwd_lean in the last bit; run-to-run identical, no atomics.
// ---------------------------------------------------------------------------------------------
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

template <int C4, int U>
__global__ __launch_bounds__(kBlock) void k_pool_fwd_lean2(
    const float* __restrict__ depth, const float* __restrict__ feat, unsigned feat_bytes,
    const int* __restrict__ ranks_depth, const int* __restrict__ row_ptr, const int4* __restrict__ tile_desc,
    float* __restrict__ out, int tiles_per_xcd, int fhw, int dfhw, float inv_fhw, float inv_dfhw, int empty_rows_kept) {
  static_assert(U == 4, "the record reads below are written for 4 points per step");
  constexpr int G = kBlock / C4;
  constexpr int GPW = 64 / C4;
  constexpr int SH = (C4 == 1 ? 4 : C4 == 2 ? 5 : C4 == 4 ? 6 : C4 == 8 ? 7 : C4 == 16 ? 8 : C4 == 32 ? 9 : 10);  // log2(row bytes)
  constexpr int kPad = G * U;
  constexpr int kRec = kCap + kPad;
  constexpr int kRowsPerLane = 3;
  constexpr int kMemInts = (kRec * 3 > kBlock * 4) ? kRec * 3 : kBlock * 4;
  __shared__ __attribute__((aligned(16))) int s_mem[kMemInts];   // [0, 2*kRec) records {pixel, depth}; [2*kRec, 3*kRec) row words
  __shared__ int s_tail_flags[G];
  __shared__ float4 s_head[kBlock];                               // head partials (rare path), outside the aliased area
  __shared__ int s_head_row[G];
  int2* s_rec = reinterpret_cast<int2*>(s_mem);
  int* s_row = s_mem + 2 * kRec;
  float4* s_tail = reinterpret_cast<float4*>(s_mem);              // after the point loop

  if ((int)(blockIdx.x >> 3) >= tiles_per_xcd) return;
  const int4 desc = tile_desc[(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3)];
  const int Ra = desc.x, nrows = desc.y, Pa = desc.z, npts = desc.w;
  const int tid = threadIdx.x;
  if (nrows <= 0) return;

  const int sub = tid % C4;
  const int grp = tid / C4;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4* out4 = reinterpret_cast<float4*>(out);
  const float4* feat4 = reinterpret_cast<const float4*>(feat);
  const bool long_row = (nrows == 1 && npts > kCap);
  const bool staged = (npts > 0 && npts <= kCap && nrows <= kRowsPerLane * kBlock);

  auto pixel_row = [&](int rd) {
    const int n = div_const(rd, dfhw, inv_dfhw);
    const int q = div_const(rd, fhw, inv_fhw);
    return n * fhw + (rd - q * fhw);
  };

  // ---- phase L (issue): the one rank table -> registers -----------------------------------------
  constexpr int kPer = (kCap + kBlock - 1) / kBlock;
  int l_rd[kPer];
  if (staged) {
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int i = tid + k * kBlock;
      if (i < npts) l_rd[k] = ranks_depth[Pa + i];
    }
    // row words of this tile start at zero (a closing record overwrites its word below)
    for (int i = tid; i < kRec; i += kBlock) s_row[i] = 0;
    __syncthreads();
  }

  // ---- phase Z: zero-fill the empty rows; the lane that sees a non-empty row marks its closing record --------
  if (!(nrows == 1 && npts > 0)) {
    const int lane = tid & 63;
    const int gw = lane / C4;
    for (int base = 0; base < nrows; base += kBlock) {
      const int i = base + tid;
      bool empty = false;
      if (i < nrows) {
        const int s0 = row_ptr[Ra + i], e0 = row_ptr[Ra + i + 1];
        empty = e0 == s0;
        if (!empty && staged) s_row[e0 - 1 - Pa] = i | (int)0x80000000;   // row offset inside the tile
      }
      // empty_rows_kept: the caller hands in a buffer whose empty rows (a property of the plan) are zero already — the
      // buffer of an earlier launch of the SAME plan, which only ever wrote the non-empty rows — so they are not stored again
      const unsigned long long m = empty_rows_kept ? 0ull : __ballot(empty);
      if (m == 0ull) continue;
      const int wave_row0 = Ra + base + (tid & ~63);
      for (int k = 0; k < 64; k += GPW) {
        const unsigned long long window = (GPW >= 64) ? m : ((m >> k) & ((1ull << GPW) - 1ull));
        if (window == 0ull) continue;
        if ((m >> (k + gw)) & 1ull)
          store_row(out4 + (size_t)(wave_row0 + k + gw) * C4 + sub, zero4, true);
      }
    }
  } else if (tid == 0 && staged) {
    s_row[npts - 1] = (int)0x80000000;             // the single row of the tile (offset 0) closes at its last point
  }
  if (npts == 0) return;

  float4 acc = zero4;

  // ---- a single long row: windows of kCap points, every group accumulates, one combine --------
  if (long_row) {
    for (int base = 0; base < npts; base += kCap) {
      const int n = min(kCap, npts - base);
      for (int i = tid; i < n; i += kBlock) {
        const int rd = ranks_depth[Pa + base + i];
        s_rec[i] = make_int2(pixel_row(rd), __float_as_int(depth[rd]));
      }
      __syncthreads();
      const int cw = (n + G - 1) / G;
      const int j0 = min(grp * cw, n), j1 = min(j0 + cw, n);
      for (int j = j0; j < j1; j += U) {
        float4 v[U];
        float d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int2 rc = s_rec[min(j + u, j1 - 1)];
          d[u] = (j + u < j1) ? __int_as_float(rc.y) : 0.f;
          v[u] = feat4[(size_t)rc.x * C4 + sub];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = fma4(d[u], v[u], acc);
      }
      __syncthreads();
    }
    s_tail[tid] = acc;
    __syncthreads();
    if (grp == 0) {
      float4 tsum = s_tail[sub];
      for (int g = 1; g < G; ++g) tsum = add4(tsum, s_tail[g * C4 + sub]);
      store_row(out4 + (size_t)Ra * C4 + sub, tsum, true);
    }
    return;
  }

  // ---- a tile that does not fit the LDS window (only for foreign tile tables): row by row -----
  if (!staged) {
    for (int r = Ra + grp; r < Ra + nrows; r += G) {
      const int s0 = row_ptr[r], len = row_ptr[r + 1] - s0;
      if (len <= 0) continue;
      float4 a4 = zero4;
      for (int q = s0; q < s0 + len; ++q) {
        const int rd = ranks_depth[q];
        a4 = fma4(depth[rd], feat4[(size_t)pixel_row(rd) * C4 + sub], a4);
      }
      store_row(out4 + (size_t)r * C4 + sub, a4, true);
    }
    return;
  }

  // ---- phase L (finish): pixel row, depth gather, records -> LDS; pad records gather zeros ---------
  const int w = (npts + G - 1) / G;
  const int Wp = (w + U - 1) / U * U;              // points per group, the same for every group
  const int n_rec = G * Wp;                        // <= npts + G*U - 1 < kRec
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int i = tid + k * kBlock;
    if (i < npts) {
      s_rec[i] = make_int2(pixel_row(l_rd[k]), __float_as_int(depth[l_rd[k]]));
    }
  }
  for (int i = npts + tid; i < n_rec; i += kBlock) s_rec[i] = make_int2(0x7fffffff, 0);   // offset beyond the buffer
  __syncthreads();

  // ---- phase P: Wp points per group, U per step, no bounds tests -------------------------------
  const __amdgpu_buffer_rsrc_t feat_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)feat_bytes, 0x00020000);
  // output window of this tile: rows [Ra, Ra + nrows); the row words hold offsets relative to Ra
  const __amdgpu_buffer_rsrc_t out_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(out + (size_t)Ra * (C4 * 4)), 0, nrows << SH, 0x00020000);
  const unsigned lane_off = (unsigned)sub << 4;
  const int i0 = grp * Wp;
  // `pend`: the wave's mask of lanes whose first point continues a row that an earlier piece started; the first row such a
  // group closes holds only a HEAD partial, kept in LDS and completed with the earlier pieces' tails after the loop.
  const bool was_pending = (i0 > 0 && i0 < npts && s_row[i0 - 1] >= 0);
  unsigned long long pend = __builtin_amdgcn_ballot_w64(was_pending);
  const int lane = tid & 63;
  const int4* rec4 = reinterpret_cast<const int4*>(s_rec + i0);      // 2 records per int4 (i0 is a multiple of U)
  const int4* row4 = reinterpret_cast<const int4*>(s_row + i0);
  const int steps = Wp / U;

  // (A variant that issued the gathers of step b+1 before the stores of step b — so that the in-order vmcnt wait for a
  // step's last gather would not cover that step's own stores — needed 72-78 VGPRs and measured the same: 41.7-43.0 us
  // against 42.3-42.6 us in alternating runs; the simple loop is kept.)
  for (int b = 0; b < steps; ++b) {
    const int4 ra = rec4[2 * b], rb = rec4[2 * b + 1];
    const int4 rw = row4[b];
    const int px[U] = {ra.x, ra.z, rb.x, rb.z};
    const float d[U] = {__int_as_float(ra.y), __int_as_float(ra.w), __int_as_float(rb.y), __int_as_float(rb.w)};
    const int cl[U] = {rw.x, rw.y, rw.z, rw.w};
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const u32x4v raw = __builtin_amdgcn_raw_buffer_load_b128(feat_rsrc, ((unsigned)px[u] << SH) | lane_off, 0, 0);
      v[u] = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc = fma4(d[u], v[u], acc);
      const bool closing = cl[u] < 0;                    // this point closes its output row
      const unsigned long long cm = __builtin_amdgcn_ballot_w64(closing);
      if (closing) {
        if ((pend >> lane) & 1ull) {                     // head partial of a continued row: completed after the loop
          s_head[tid] = acc;
          if (sub == 0) s_head_row[grp] = cl[u];
        } else {
          const u32x4v o = {__float_as_uint(acc.x), __float_as_uint(acc.y), __float_as_uint(acc.z), __float_as_uint(acc.w)};
          __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, ((unsigned)cl[u] << SH) | lane_off, 0, 2 /* nt */);
        }
        acc = zero4;
      }
      pend &= ~cm;
    }
  }
  const bool pending = ((pend >> lane) & 1ull) != 0ull;   // started inside a row and never closed it
  // piece ends inside a row (or is empty: then it lies "inside" whatever row surrounds it, with a zero partial)
  const int i1 = min(i0 + Wp, npts);
  const bool open_end = (i1 <= i0) || (s_row[i1 - 1] >= 0);
  __syncthreads();   // every group is done with the records: their LDS is reused for the tails
  s_tail[tid] = acc;
  // bit0: the piece ends inside a row; bit1: the piece lies entirely inside one row that started before it
  if (sub == 0) s_tail_flags[grp] = (open_end ? 1 : 0) | ((pending || i1 <= i0) ? 2 : 0);
  __syncthreads();

  if (was_pending && !pending) {
    // the row I closed first started in earlier pieces: add their open tails, walking back through pieces that lie
    // entirely inside the row and stopping after the first one that started a row of its own
    int g0 = grp;
    while (g0 > 0) {
      const int f = s_tail_flags[g0 - 1];
      if (!(f & 1)) break;
      --g0;
      if (!(f & 2)) break;
    }
    float4 tsum = zero4;                             // point order
    for (int g = g0; g < grp; ++g) tsum = add4(tsum, s_tail[g * C4 + sub]);
    tsum = add4(tsum, s_head[tid]);
    const u32x4v o = {__float_as_uint(tsum.x), __float_as_uint(tsum.y), __float_as_uint(tsum.z), __float_as_uint(tsum.w)};
    __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, ((unsigned)s_head_row[grp] << SH) | lane_off, 0, 2);
  }
}

// schedule slot -> {first row, #rows, first point, #points}; idle slots get #rows = 0
__global__ __launch_bounds__(kBlock) void k_tile_desc(const int* __restrict__ row_ptr,
                                                      const int* __restrict__ tile_row,
                                                      const int* __restrict__ tile_order,
                                                      int n_slots, int n_tiles,
                                                      int4* __restrict__ desc) {
  for (int s = blockIdx.x * kBlock + threadIdx.x; s < n_slots; s += gridDim.x * kBlock) {
    const int t = tile_order ? tile_order[s] : (s < n_tiles ? s : -1);
    int4 d = make_int4(0, 0, 0, 0);
    if (t >= 0 && t < n_tiles) {
      const int ra = tile_row[t], rb = tile_row[t + 1];
      d = make_int4(ra, rb - ra, row_ptr[ra], row_ptr[rb] - row_ptr[ra]);
    }
    desc[s] = d;
  }
}

// Any channel count / any alignment: one thread per (unit, channel), serial over the points.
template <bool DENSE>
__global__ __launch_bounds__(kBlock) void k_pool_fwd_generic(
    const float* __restrict__ depth, const float* __restrict__ feat,
    const int* __restrict__ ranks_depth, const int* __restrict__ ranks_feat,
    const int* __restrict__ ranks_bev, const int* __restrict__ starts,
    const int* __restrict__ lengths_or_rowptr, float* __restrict__ out, int c, int n_units) {
  const int64_t total = (int64_t)n_units * c;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * kBlock) {
    const int u = (int)(t / c);
    const int ch = (int)(t % c);
    int s, len, row;
    if (DENSE) {
      s = lengths_or_rowptr[u];
      len = lengths_or_rowptr[u + 1] - s;
      row = u;
    } else {
      s = starts[u];
      len = lengths_or_rowptr[u];
      if (len <= 0) continue;
      row = ranks_bev[s];
    }
    float acc = 0.f;
    for (int i = 0; i < len; ++i)
      acc = fmaf(feat[(size_t)ranks_feat[s + i] * c + ch], depth[ranks_depth[s + i]], acc);
    out[(size_t)row * c + ch] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// backward: one group of C4 lanes per backward interval (= one image-feature pixel).
// ---------------------------------------------------------------------------------------------
template <int C4>
__global__ __launch_bounds__(kBlock) void k_pool_bwd(
    const float4* __restrict__ og4, const float* __restrict__ depth,
    const float4* __restrict__ feat4, const int* __restrict__ ranks_depth,
    const int* __restrict__ ranks_feat, const int* __restrict__ ranks_bev,
    const int* __restrict__ starts, const int* __restrict__ lengths,
    float* __restrict__ depth_grad, float4* __restrict__ feat_grad4, int n_intervals) {
  constexpr int G = kBlock / C4;
  const int tid = threadIdx.x;
  const int sub = tid % C4;
  const int grp = tid / C4;
  for (int base = blockIdx.x * G; base < n_intervals; base += gridDim.x * G) {
    const int iv = base + grp;
    if (iv >= n_intervals) continue;
    const int s = starts[iv];
    const int len = lengths[iv];
    if (len <= 0) continue;
    float4 fg = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int cb = 0; cb < len; cb += C4) {
      const int mine = cb + sub;
      int my_rb = 0, my_rf = 0, my_rd = 0;
      float my_d = 0.f, my_dot = 0.f;
      if (mine < len) {
        my_rb = ranks_bev[s + mine];
        my_rf = ranks_feat[s + mine];
        my_rd = ranks_depth[s + mine];
        my_d = depth[my_rd];
      }
      const int n = min(C4, len - cb);
      for (int j = 0; j < n; ++j) {
        const int v = __shfl(my_rb, j, C4);
        const int f = __shfl(my_rf, j, C4);
        const float d = __shfl(my_d, j, C4);
        const float4 g = og4[(size_t)v * C4 + sub];
        const float4 x = feat4[(size_t)f * C4 + sub];
        float dot = fmaf(g.w, x.w, fmaf(g.z, x.z, fmaf(g.y, x.y, g.x * x.x)));
#pragma unroll
        for (int o = C4 / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, C4);
        if (sub == j) my_dot = dot;
        fg = fma4(d, g, fg);
      }
      if (mine < len) depth_grad[my_rd] = my_dot;
    }
    feat_grad4[(size_t)ranks_feat[s] * C4 + sub] = fg;
  }
}

// ---------------------------------------------------------------------------------------------
// Scheduled backward: one group of C4 lanes per image-feature pixel, pixels taken from a plan-made
// schedule of 16-byte descriptors {pixel row, first point, #points, -}.
//   * the schedule lists EVERY pixel (also those without points), so feat_grad is written densely
//     and needs no zero-fill; it walks the image in 4x4 pixel patches and gives each XCD one
//     contiguous run of patches: neighbouring pixels hit the same BEV rows (a voxel collects its
//     points from adjacent pixels), so an out_grad row fetched once is reused from L1/L2;
//   * per chunk of C4 points the three tables are read coalesced, depth is gathered C4-wide,
//     U out_grad rows (16 B x C4 lanes) are in flight per group;
//   * depth_grad needs a dot product over the C channels for every point: each lane keeps its
//     4-channel partial for the C4 points of the chunk and ONE log2(C4)-stage butterfly
//     (C4-1 exchanges instead of C4*log2(C4)) leaves lane j with the sum of point j, which
//     it then scatters to depth_grad.
// feat_grad: fg += depth * g in table order (same fma chain as the reference kernel).
// ---------------------------------------------------------------------------------------------
template <int C4, int U>
__global__ __launch_bounds__(kBlock) void k_pool_bwd_sched(
    const float4* __restrict__ og4, const float* __restrict__ depth,
    const float4* __restrict__ feat4, const int* __restrict__ ranks_depth,
    const int* __restrict__ ranks_row, const int4* __restrict__ pix_desc,
    float* __restrict__ depth_grad, float4* __restrict__ feat_grad4, int groups_per_xcd) {
  static_assert(U >= 1 && U <= C4 && (C4 % U) == 0 && (U & (U - 1)) == 0, "U: power of two dividing C4");
  constexpr int G = kBlock / C4;
  const int sub = threadIdx.x % C4;
  const int grp = threadIdx.x / C4;
  const int gi = (int)(blockIdx.x >> 3) * G + grp;     // group slot inside this XCD's run
  if (gi >= groups_per_xcd) return;
  const int4 desc = pix_desc[(size_t)(blockIdx.x & 7) * groups_per_xcd + gi];
  const int f = desc.x, s = desc.y, len = desc.z;
  if (f < 0) return;
  float4 fg = make_float4(0.f, 0.f, 0.f, 0.f);
  if (len > 0) {
    const float4 x = feat4[(size_t)f * C4 + sub];
    for (int cb = 0; cb < len; cb += C4) {
      const int n = min(C4, len - cb);
      int my_rb = 0, my_rd = 0;
      float my_d = 0.f;
      if (sub < n) {
        my_rb = ranks_row[s + cb + sub];
        my_rd = ranks_depth[s + cb + sub];
        my_d = depth[my_rd];
      }
      for (int j = 0; j < n; j += U) {
        float4 g[U];
        float d[U];
        float part[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int jj = min(j + u, n - 1);
          const int v = __shfl(my_rb, jj, C4);
          d[u] = (j + u < n) ? __shfl(my_d, jj, C4) : 0.f;
          g[u] = og4[(size_t)v * C4 + sub];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          part[u] = fmaf(g[u].w, x.w, fmaf(g[u].z, x.z, fmaf(g[u].y, x.y, g[u].x * x.x)));
          fg = fma4(d[u], g[u], fg);
        }
        // Channel sums of the U points: log2(U) halving stages over the TOP lane bits (a lane keeps
        // the partials whose index matches its bits), then a plain xor-reduction over the remaining
        // low bits.  Afterwards every lane with (sub / (C4/U)) == u holds the full dot of point j+u.
        int m = C4 / 2;
#pragma unroll
        for (int h = U / 2; h >= 1; h >>= 1, m >>= 1) {
          const bool hi = (sub & m) != 0;
#pragma unroll
          for (int k = 0; k < h; ++k) {
            const float send = hi ? part[k] : part[k + h];
            const float keep = hi ? part[k + h] : part[k];
            part[k] = keep + __shfl_xor(send, m, C4);
          }
        }
#pragma unroll
        for (; m >= 1; m >>= 1) part[0] += __shfl_xor(part[0], m, C4);
        constexpr int kLanesPerPoint = C4 / U;
        const int u_mine = sub / kLanesPerPoint;
        const int rd_mine = __shfl(my_rd, min(j + u_mine, n - 1), C4);
        if ((sub % kLanesPerPoint) == 0 && j + u_mine < n) depth_grad[rd_mine] = part[0];
      }
    }
  }
  feat_grad4[(size_t)f * C4 + sub] = fg;
}

// ---------------------------------------------------------------------------------------------
// Patch backward (C = 64): one workgroup per PATCH of 16 consecutive image-feature pixels (a run along the image row), one
// group of 16 lanes per pixel.
//   * depth_grad is written DENSELY: the D x 16 block of depth gradients of the patch is assembled in LDS (zero where a
//     frustum point falls outside the grid) and stored as D segments of 64 contiguous bytes.  The scheduled kernel scattered
//     2 million 4-byte stores, 45 KB apart along a ray, into a buffer that a separate 16 MB memset had cleared;
//   * the depth values of the patch are read the same way (D coalesced 64-byte segments into LDS) instead of one
//     4-byte gather per point;
//   * the tables of a pixel are read 16 points at a time, one per lane, and handed to the row's lanes by DPP row
//     broadcasts (v_mov_dpp row_newbcast, no LDS round trip); the channel sum of a point's depth gradient is four
//     v_add_f32_dpp row_ror adds;
//   * out_grad rows are gathered with range-checked buffer loads (32-bit offsets, a lane past the end of its pixel's point
//     list asks for a row beyond the buffer and gets zeros: no bounds branches in the point loop).
// feat_grad: fg += depth * g in table order (the reference's fma chain, bit-exact); depth_grad: channel sums in a fixed
// lane order (run-to-run identical; differs from the reference's serial channel loop by fp32 rounding).
// ---------------------------------------------------------------------------------------------
template <int J>
__device__ __forceinline__ int dpp_row_bcast_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xf, 0xf, true);        // row_newbcast:J (16-lane rows); bound_ctrl: no `old` to initialise
}
template <int J>
__device__ __forceinline__ float dpp_row_bcast_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + J, 0xf, 0xf, true));
}
template <int N>
__device__ __forceinline__ float dpp_ror_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}

constexpr int kPatch = 16;     // pixels per patch = groups per workgroup at C = 64

typedef unsigned u32x4t __attribute__((ext_vector_type(4)));

// one point of the current 16-point chunk: g = its out_grad row (already gathered), d = its depth value
template <int JJ>
__device__ __forceinline__ void patch_point(const u32x4t a, float d, const float4 x, float4& fg, float& mydot, int sub) {
  const float4 g = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
  fg = fma4(d, g, fg);
  float p = fmaf(g.w, x.w, fmaf(g.z, x.z, fmaf(g.y, x.y, g.x * x.x)));
  p = dpp_ror_add<8>(p); p = dpp_ror_add<4>(p); p = dpp_ror_add<2>(p); p = dpp_ror_add<1>(p);   // channel sum over the 16 lanes
  mydot = (sub == JJ) ? p : mydot;
}

// points J0 .. J0+7 of the current chunk: eight gathers in flight per lane group
template <int J0>
__device__ __forceinline__ void patch_batch8(const __amdgpu_buffer_rsrc_t og_rsrc, unsigned lane_off, int rr, float dval,
                                             const float4 x, float4& fg, float& mydot, int sub) {
#define OMNIHD_G(K) const u32x4t a##K = __builtin_amdgcn_raw_buffer_load_b128(og_rsrc, ((unsigned)dpp_row_bcast_i<J0 + K>(rr) << 8) | lane_off, 0, 0);
  OMNIHD_G(0) OMNIHD_G(1) OMNIHD_G(2) OMNIHD_G(3) OMNIHD_G(4) OMNIHD_G(5) OMNIHD_G(6) OMNIHD_G(7)
#undef OMNIHD_G
  patch_point<J0 + 0>(a0, dpp_row_bcast_f<J0 + 0>(dval), x, fg, mydot, sub);
  patch_point<J0 + 1>(a1, dpp_row_bcast_f<J0 + 1>(dval), x, fg, mydot, sub);
  patch_point<J0 + 2>(a2, dpp_row_bcast_f<J0 + 2>(dval), x, fg, mydot, sub);
  patch_point<J0 + 3>(a3, dpp_row_bcast_f<J0 + 3>(dval), x, fg, mydot, sub);
  patch_point<J0 + 4>(a4, dpp_row_bcast_f<J0 + 4>(dval), x, fg, mydot, sub);
  patch_point<J0 + 5>(a5, dpp_row_bcast_f<J0 + 5>(dval), x, fg, mydot, sub);
  patch_point<J0 + 6>(a6, dpp_row_bcast_f<J0 + 6>(dval), x, fg, mydot, sub);
  patch_point<J0 + 7>(a7, dpp_row_bcast_f<J0 + 7>(dval), x, fg, mydot, sub);
}

// PACKED: `ranks_row` holds (output row | depth bin << 24) per point and `ranks_depth` is not read: one table word per point
// instead of two (8.1 MB less traffic per launch at R1, one table load per chunk instead of two).
template <bool PACKED>
__global__ __launch_bounds__(kBlock, 8) void k_pool_bwd_patch(
    const float* __restrict__ og, unsigned og_bytes, const float* __restrict__ depth, const float4* __restrict__ feat4,
    const int* __restrict__ ranks_depth, const int* __restrict__ ranks_row, const int* __restrict__ pix_ptr,
    const int* __restrict__ patch_order, int patches_per_xcd, int patches_per_img, int fhw, int d_bins, float inv_fhw,
    float* __restrict__ depth_grad, float4* __restrict__ feat_grad4) {
  constexpr int C4 = 16;
  extern __shared__ float s_dyn[];                 // [0, D*16) depth values of the patch, [D*16, 2*D*16) depth gradients
  const int slot = (int)(blockIdx.x >> 3);
  if (slot >= patches_per_xcd) return;
  const int patch = patch_order[(size_t)(blockIdx.x & 7) * patches_per_xcd + slot];
  if (patch < 0) return;
  const int tid = threadIdx.x;
  const int sub = tid % C4;
  const int grp = tid / C4;
  const int img = patch / patches_per_img;
  const int hw0 = (patch - img * patches_per_img) * kPatch;
  const int npx = min(kPatch, fhw - hw0);
  float* s_dv = s_dyn;
  float* s_dg = s_dyn + d_bins * kPatch;
  const size_t img_base = (size_t)img * d_bins * fhw + hw0;     // index of (img, d = 0, pixel hw0) in depth / depth_grad

  // ---- depth values of the patch -> LDS (D segments of 64 contiguous bytes), gradients start at zero ----------
  const int n_cell = d_bins * kPatch;
  for (int i = tid; i < n_cell; i += kBlock) {
    const int d = i / kPatch, px = i % kPatch;
    s_dv[i] = (px < npx) ? depth[img_base + (size_t)d * fhw + px] : 0.f;
    s_dg[i] = 0.f;
  }
  const bool valid = grp < npx;
  const int f = img * fhw + hw0 + grp;                          // my pixel (feature row)
  int s = 0, len = 0;
  float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid) {
    s = pix_ptr[f];
    len = pix_ptr[f + 1] - s;
    x = feat4[(size_t)f * C4 + sub];
  }
  // the longest point list among the four pixels of this wavefront bounds the (wave-uniform) trip count
  int ml = len;
  ml = max(ml, __shfl_xor(ml, 16));
  ml = max(ml, __shfl_xor(ml, 32));
  const int wave_len = __builtin_amdgcn_readfirstlane(ml);
  __syncthreads();

  const __amdgpu_buffer_rsrc_t og_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)og, 0, (int)og_bytes, 0x00020000);
  const unsigned lane_off = (unsigned)sub << 4;
  const int rd_base = (img * d_bins) * fhw + hw0 + grp;         // ranks_depth of (my pixel, d = 0)
  float4 fg = make_float4(0.f, 0.f, 0.f, 0.f);
  // tables of the first chunk; each chunk's tables are requested one chunk ahead
  int rr_n = 0x00ffffff, rd_n = rd_base;                        // row beyond the buffer: the gather returns zeros
  if (sub < len) {
    rr_n = ranks_row[s + sub];
    if (!PACKED) rd_n = ranks_depth[s + sub];
  }
  for (int cb = 0; cb < wave_len; cb += kPatch) {
    const int mine = cb + sub;
    const bool inb = mine < len;
    const int rr = PACKED ? (rr_n & 0x00ffffff) : rr_n;
    const int dk = PACKED ? (int)((unsigned)rr_n >> 24) : div_const(rd_n - rd_base, fhw, inv_fhw);
    rr_n = 0x00ffffff;
    rd_n = rd_base;
    if (mine + kPatch < len) {
      rr_n = ranks_row[s + mine + kPatch];
      if (!PACKED) rd_n = ranks_depth[s + mine + kPatch];
    }
    const float dval = inb ? s_dv[dk * kPatch + grp] : 0.f;
    float mydot = 0.f;
    patch_batch8<0>(og_rsrc, lane_off, rr, dval, x, fg, mydot, sub);
    if (cb + 8 < wave_len) patch_batch8<8>(og_rsrc, lane_off, rr, dval, x, fg, mydot, sub);
    if (inb) s_dg[dk * kPatch + grp] = mydot;
  }
  if (valid) feat_grad4[(size_t)f * C4 + sub] = fg;
  __syncthreads();
  for (int i = tid; i < n_cell; i += kBlock) {
    const int d = i / kPatch, px = i % kPatch;
    if (px < npx) depth_grad[img_base + (size_t)d * fhw + px] = s_dg[i];
  }
}

// ---------------------------------------------------------------------------------------------
// Stream backward (round 4, C = 64; OPT-IN, OMNIHD_POOL_BWD_STREAM=1 — measured slower than k_pool_bwd_patch at 256x704 and equal at
// 544x960): every out_grad row that the 16 pixels of a patch share is fetched from global memory ONCE and handed
// on through LDS.  Why it was built: k_pool_bwd_patch issues one 256-byte row gather per frustum point and neighbouring pixels ask
// for the same rows — points per DISTINCT row inside a 16-pixel patch: 2.3 (16x1 run) / 2.9 (8x2) / 3.2 (4x4) at 256x704, 3.1 / 4.3 /
// 4.8 at 544x960 (profiles/round4/pool_bwd_row_reuse.txt); the L1 serves the repeats, but each still occupies the vector-memory path.
// The plan lists, per patch, the distinct output rows its points touch (sorted: `uniq`) and cuts that list into STAGES of R rows; a
// pixel's points are sorted by output row (the backward tables are a stable sort of the row-sorted forward tables by pixel), so its
// points of one stage are a contiguous piece of its list.  A first form with one workgroup per patch and a barrier per stage
// (scripts/lab/records/pool_bwd_shared_workgroup.inc.txt) paid 5-7 us of dependent loads in front of every workgroup's first stage;
// this form removes that and halves the VALU work per point:
//   * ONE WAVEFRONT walks a STREAM of stages (the stages of the patches the plan dealt to it, one after the other): no
//     workgroup barrier anywhere, and the loads of a stage are issued while earlier stages are consumed, across patch
//     boundaries.  In iteration t the wave stores the rows of stage t-2 (registers -> LDS) and consumes that stage, gathers
//     the rows and the first table words of stage t-1, and reads the row ids and per-pixel offsets of stage t and the stream
//     entry t+1.  Every load is unconditional (range-checked buffer loads: a missing row / word / entry is an address beyond
//     the buffer and returns zeros), so no register that a pending load writes is copied before the point loop: the only
//     place the wave waits for memory is the top of an iteration.
//   * 4 lanes per pixel (16 channels each) instead of 16: one wave = the 16 pixels of a patch, one point-loop step = one point of
//     every pixel: 16 packed FMAs for feat_grad + 9 packed ops and 2 quad exchanges for the depth gradient per 16 points (the
//     16-lane version: 17 VALU per 4 points).  The four 16-byte slots a lane owns of a row are rotated by the pixel's position
//     inside its ds_read_b128 lane group, so the 16 lanes of a group hit 16 distinct 16-byte bank slots whatever rows they read.
// feat_grad: fg += depth * g per channel in table order (the reference's fma chain, bit-exact); depth_grad: a fixed-order
// channel sum (16 in-lane packed FMAs, then two quad exchanges): run-to-run identical, differs from the reference's serial
// channel loop by fp32 rounding.
//   stream[e]   = {patch | flags of stage e-2 (bit 30: first stage of its patch, 29: last, 28: entry has a stage to consume),
//                  first uniq entry of stage e, (row of stage e in px_off) | #rows << 24, patch whose first stage is e-1 or -1}
//   px_off      = per stage 16 ints (+ the 16 of the next row): index into pt_word of pixel g's first point of the stage
// ---------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kStreamFirst = 1 << 30, kStreamLast = 1 << 29, kStreamValid = 1 << 28;
constexpr int kStreamCellRegs = 16;       // D * 16 cells of a patch over 64 lanes, D <= 64

template <int CTRL>
__device__ __forceinline__ int dpp_quad_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);             // quad_perm
}
template <int CTRL>
__device__ __forceinline__ float dpp_quad_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// One point-loop step = point t of every pixel's piece, in two halves so that the LDS reads of step t+1 are in flight while step
// t is accumulated (with two waves per SIMD nothing else hides the LDS latency).
struct StreamStep {
  float4 g0, g1, g2, g3;      // my 16 channels of the point's out_grad row
  float dv;                   // its depth value
  int cell;                   // (depth bin * 16 + pixel): where its depth gradient goes
  bool act;                   // the pixel's piece has a point t
};

// J = t % 4: lane q of a quad holds the table word of step 4*(t/4) + q in `wblk`
template <int J, int R>
__device__ __forceinline__ StreamStep stream_load(const char* s_rows, const float* s_dv, int wblk, int t, int cnt, int p, const int (&o)[4]) {
  StreamStep st;
  const int w = dpp_quad_i<J * 0x55>(wblk);
  st.act = t < cnt;
  const int off = st.act ? (w & 0x00ffffff) : R * 256;                         // past the piece: the zero row
  st.cell = (int)(((unsigned)w >> 24) << 4) | p;
  st.dv = s_dv[st.act ? st.cell : p];
  st.g0 = *reinterpret_cast<const float4*>(s_rows + off + o[0]);
  st.g1 = *reinterpret_cast<const float4*>(s_rows + off + o[1]);
  st.g2 = *reinterpret_cast<const float4*>(s_rows + off + o[2]);
  st.g3 = *reinterpret_cast<const float4*>(s_rows + off + o[3]);
  return st;
}

__device__ __forceinline__ void stream_math(const StreamStep& st, float* s_dg, int q, const f32x2 (&x)[8], f32x2 (&fg)[8]) {
  const float dv = st.act ? st.dv : 0.f;
  // {dv, dv} as a REAL register pair: the compiler's own form is a packed op that broadcasts the low register and leaves the high
  // one of the pair to whatever lives there — here the destination of a pending load, which the point loop then waited for
  float dv_hi;
  asm volatile("v_mov_b32 %0, %1" : "=v"(dv_hi) : "v"(dv));
  const f32x2 d2 = {dv, dv_hi};
  const f32x2 ga[8] = {{st.g0.x, st.g0.y}, {st.g0.z, st.g0.w}, {st.g1.x, st.g1.y}, {st.g1.z, st.g1.w},
                       {st.g2.x, st.g2.y}, {st.g2.z, st.g2.w}, {st.g3.x, st.g3.y}, {st.g3.z, st.g3.w}};
#pragma unroll
  for (int i = 0; i < 8; ++i) fg[i] = __builtin_elementwise_fma(d2, ga[i], fg[i]);
  f32x2 acc = ga[0] * x[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) acc = __builtin_elementwise_fma(ga[i], x[i], acc);
  float sum = acc.x + acc.y;
  sum += dpp_quad_f<0xB1>(sum);                                                // lanes 1,0,3,2
  sum += dpp_quad_f<0x4E>(sum);                                                // lanes 2,3,0,1
  if (st.act && q == 0) s_dg[st.cell] = sum;
}

template <int RQ>          // RQ = rows per stage / 4 = row gathers per lane and stage
__global__ __launch_bounds__(64, 2) void k_pool_bwd_stream(
    const float* __restrict__ og, unsigned og_bytes, const float* __restrict__ depth, unsigned depth_bytes,
    const float* __restrict__ feat, unsigned feat_bytes, const int* __restrict__ pt_word, unsigned word_bytes,
    const int* __restrict__ uniq, unsigned uniq_bytes, const int* __restrict__ px_off, unsigned off_bytes,
    const int4* __restrict__ stream, unsigned stream_bytes, const int* __restrict__ stream_ptr, int streams_per_xcd,
    int fh, int fw, int pw_shift, int d_bins, float* __restrict__ depth_grad, float4* __restrict__ feat_grad4) {
  constexpr int R = RQ * 4;
  extern __shared__ float s_dyn[];                 // [(R+1) rows x 64][64*16 depth values][64*16 depth gradients]
  char* s_rows = reinterpret_cast<char*>(s_dyn);
  float* s_dv = s_dyn + (R + 1) * 64;
  float* s_dg = s_dv + 64 * kPatch;
  const int lane = threadIdx.x;
  const int p = lane >> 2, q = lane & 3;           // point loop: pixel of the patch, quarter of its channels
  const int gq = lane & 15, gr = lane >> 4;        // row gathers and depth cells: 16-byte slot / pixel, row of a 4-row group
  const int sid = (int)(blockIdx.x & 7) * streams_per_xcd + (int)(blockIdx.x >> 3);
  const int sbeg = stream_ptr[sid], send = stream_ptr[sid + 1];
  if (sbeg >= send) return;
  if (lane < 16) reinterpret_cast<float4*>(s_rows)[R * 16 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);

  const __amdgpu_buffer_rsrc_t og_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)og, 0, (int)og_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t depth_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)depth, 0, (int)depth_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t feat_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)feat_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t word_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)pt_word, 0, (int)word_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t uniq_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)uniq, 0, (int)uniq_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t off_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)px_off, 0, (int)off_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t stream_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)stream, 0, (int)stream_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dgrad_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)depth_grad, 0, (int)depth_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t fgrad_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)feat_grad4, 0, (int)feat_bytes, 0x00020000);
  constexpr unsigned kBeyond = 0xfffffff0u;        // an offset no buffer reaches: the load returns zeros, the store is dropped

  const int fhw = fh * fw;
  const int plane4 = 16 * fhw;                     // bytes of four depth planes
  const int pw = 1 << pw_shift, ph = kPatch >> pw_shift;
  const int pcols = (fw + pw - 1) >> pw_shift, prows = (fh + ph - 1) / ph;
  int o[4], slot[4];                               // my four 16-byte slots of a 256-byte row, rotated by the pixel's place in its lane group
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    slot[j] = 4 * ((((p >> 1) & 3) + j) & 3) + q;
    o[j] = 16 * slot[j];
  }

  // ---- state carried from iteration to iteration ---------------------------------------------------------------
  u32x4t pre[RQ];                                  // rows of the stage consumed NEXT iteration
#pragma unroll
  for (int m = 0; m < RQ; ++m) pre[m] = u32x4t{0u, 0u, 0u, 0u};
  int w_next[4] = {0, 0, 0, 0};                    // first 16 table words of my pixel's piece of that stage
  int ids_prev = 0, nrows_prev = 0;                // row ids of the stage whose rows are gathered this iteration
  int oa1 = 0, ob1 = 0, oa2 = 0, ob2 = 0;          // my pixel's piece [oa, ob) of pt_word: stage t-1, stage t-2
  f32x2 x[8], fg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = fg[i] = f32x2{0.f, 0.f};
  u32x4t xn[4];                                    // feature row of my pixel in the NEXT patch, its depth cells
  unsigned cn[kStreamCellRegs];
#pragma unroll
  for (int j = 0; j < 4; ++j) xn[j] = u32x4t{0u, 0u, 0u, 0u};
#pragma unroll
  for (int it = 0; it < kStreamCellRegs; ++it) cn[it] = 0u;
  int f_cur = -1, f_next = -1;                     // my pixel's feature row (-1: outside the image)
  int cell_cur = -1, cell_next = -1;               // index of (image, d = 0, pixel gq of the patch) in depth / depth_grad (-1: outside)

  int ex, ey, ez, ew;
  {
    const u32x4t e = __builtin_amdgcn_raw_buffer_load_b128(stream_rsrc, (unsigned)sbeg << 4, 0, 0);
    ex = __builtin_amdgcn_readfirstlane((int)e.x); ey = __builtin_amdgcn_readfirstlane((int)e.y);
    ez = __builtin_amdgcn_readfirstlane((int)e.z); ew = __builtin_amdgcn_readfirstlane((int)e.w);
  }
  for (int j = sbeg; j < send; ++j) {
    // ---- (1) what the previous iteration requested has arrived: rows of stage t-2 -> LDS, a new patch's pixel data ----
#pragma unroll
    for (int m = 0; m < RQ; ++m) reinterpret_cast<u32x4t*>(s_rows)[(4 * m + gr) * 16 + gq] = pre[m];
    if (ex & kStreamFirst) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        x[2 * i] = f32x2{__uint_as_float(xn[i].x), __uint_as_float(xn[i].y)};
        x[2 * i + 1] = f32x2{__uint_as_float(xn[i].z), __uint_as_float(xn[i].w)};
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) fg[i] = f32x2{0.f, 0.f};
      f_cur = f_next;
      cell_cur = cell_next;
#pragma unroll
      for (int it = 0; it < kStreamCellRegs; ++it) {       // (64 bins are laid out: cells past d_bins hold the zeros their loads returned)
        s_dv[(gr + 4 * it) * kPatch + gq] = __uint_as_float(cn[it]);
        s_dg[(gr + 4 * it) * kPatch + gq] = 0.f;
      }
    }
    int wq[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) wq[c] = w_next[c];

    // ---- (2) requests: entry t+1, ids + offsets of stage t, pixel data of a patch that starts at t-1, rows + words of t-1 ----
    const u32x4t e_n = __builtin_amdgcn_raw_buffer_load_b128(stream_rsrc, (unsigned)(j + 1) << 4, 0, 0);
    const int ids_n = (int)__builtin_amdgcn_raw_buffer_load_b32(uniq_rsrc, (unsigned)(ey + lane) << 2, 0, 0);
    const int so = ez & 0x00ffffff;
    const int oa_n = (int)__builtin_amdgcn_raw_buffer_load_b32(off_rsrc, (unsigned)(so * kPatch + p) << 2, 0, 0);
    const int ob_n = (int)__builtin_amdgcn_raw_buffer_load_b32(off_rsrc, (unsigned)((so + 1) * kPatch + p) << 2, 0, 0);
    if (ew >= 0) {
      const int img = ew / (pcols * prows);
      const int pr = (ew - img * pcols * prows) / pcols, pc = ew - (img * prows + pr) * pcols;
      const int h0 = pr * ph, w0 = pc << pw_shift;
      const int hh = h0 + (p >> pw_shift), ww = w0 + (p & (pw - 1));
      f_next = (hh < fh && ww < fw) ? img * fhw + hh * fw + ww : -1;
      const int hc = h0 + (gq >> pw_shift), wc = w0 + (gq & (pw - 1));
      cell_next = (hc < fh && wc < fw) ? img * d_bins * fhw + hc * fw + wc : -1;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        xn[i] = __builtin_amdgcn_raw_buffer_load_b128(feat_rsrc, f_next >= 0 ? ((unsigned)f_next << 8) | (unsigned)o[i] : kBeyond, 0, 0);
      // cell (d = gr + 4*it, pixel gq): the lane part of the address in the VGPR offset, 4*it depth planes in the scalar offset
      const unsigned c_off = (unsigned)(cell_next + gr * fhw) << 2;
#pragma unroll
      for (int it = 0; it < kStreamCellRegs; ++it)
        cn[it] = __builtin_amdgcn_raw_buffer_load_b32(depth_rsrc, (cell_next >= 0 && gr + 4 * it < d_bins) ? c_off : kBeyond, it * plane4, 0);
    }
    {
      const int id_use = (lane < nrows_prev) ? ids_prev : 0x00ffffff;          // a row beyond out_grad: the gather returns zeros
#pragma unroll
      for (int m = 0; m < RQ; ++m) {
        const int r = __builtin_amdgcn_ds_bpermute((4 * m + gr) << 2, id_use);
        pre[m] = __builtin_amdgcn_raw_buffer_load_b128(og_rsrc, ((unsigned)r << 8) | ((unsigned)gq << 4), 0, 0);
      }
      const int cnt1 = ob1 - oa1;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        w_next[c] = (int)__builtin_amdgcn_raw_buffer_load_b32(word_rsrc, (4 * c + q < cnt1) ? (unsigned)(oa1 + 4 * c + q) << 2 : kBeyond, 0, 0);
    }

    // ---- (3) the points of stage t-2 ------------------------------------------------------------------------------
    if (ex & kStreamValid) {
      const int cnt = ob2 - oa2;
      int ml = cnt;
      ml = max(ml, __shfl_xor(ml, 4));
      ml = max(ml, __shfl_xor(ml, 8));
      ml = max(ml, __shfl_xor(ml, 16));
      ml = max(ml, __shfl_xor(ml, 32));
      const int trip = __builtin_amdgcn_readfirstlane(ml);
      StreamStep sa = stream_load<0, R>(s_rows, s_dv, wq[0], 0, cnt, p, o), sb;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (4 * c < trip) {
          sb = stream_load<1, R>(s_rows, s_dv, wq[c], 4 * c + 1, cnt, p, o);
          stream_math(sa, s_dg, q, x, fg);
          sa = stream_load<2, R>(s_rows, s_dv, wq[c], 4 * c + 2, cnt, p, o);
          stream_math(sb, s_dg, q, x, fg);
          sb = stream_load<3, R>(s_rows, s_dv, wq[c], 4 * c + 3, cnt, p, o);
          stream_math(sa, s_dg, q, x, fg);
          if (c < 3) sa = stream_load<0, R>(s_rows, s_dv, wq[c < 3 ? c + 1 : 3], 4 * c + 4, cnt, p, o);
          stream_math(sb, s_dg, q, x, fg);
        }
      }
      for (int c = 4; 4 * c < trip; ++c) {           // a piece longer than 16 points: its further words are read here
        const int wl = (int)__builtin_amdgcn_raw_buffer_load_b32(word_rsrc, (4 * c + q < cnt) ? (unsigned)(oa2 + 4 * c + q) << 2 : kBeyond, 0, 0);
        sa = stream_load<0, R>(s_rows, s_dv, wl, 4 * c + 0, cnt, p, o);
        sb = stream_load<1, R>(s_rows, s_dv, wl, 4 * c + 1, cnt, p, o);
        stream_math(sa, s_dg, q, x, fg);
        sa = stream_load<2, R>(s_rows, s_dv, wl, 4 * c + 2, cnt, p, o);
        stream_math(sb, s_dg, q, x, fg);
        sb = stream_load<3, R>(s_rows, s_dv, wl, 4 * c + 3, cnt, p, o);
        stream_math(sa, s_dg, q, x, fg);
        stream_math(sb, s_dg, q, x, fg);
      }
      if (ex & kStreamLast) {                        // the patch is complete: its feature gradient rows, its D x 16 block of depth gradients
#pragma unroll
        for (int i = 0; i < 4; ++i)                  // a store beyond the buffer is dropped
          __builtin_amdgcn_raw_buffer_store_b128(u32x4t{__float_as_uint(fg[2 * i].x), __float_as_uint(fg[2 * i].y), __float_as_uint(fg[2 * i + 1].x),
                                                        __float_as_uint(fg[2 * i + 1].y)},
                                                 fgrad_rsrc, f_cur >= 0 ? ((unsigned)f_cur << 8) | (unsigned)o[i] : kBeyond, 0, 0);
        const unsigned c_off = (unsigned)(cell_cur + gr * fhw) << 2;
#pragma unroll
        for (int it = 0; it < kStreamCellRegs; ++it)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s_dg[(gr + 4 * it) * kPatch + gq]), dgrad_rsrc,
                                                (cell_cur >= 0 && gr + 4 * it < d_bins) ? c_off : kBeyond, it * plane4, 0);
      }
    }

    // ---- (4) rotate (the only copies of registers that pending loads write: after the point loop) -------------------
    oa2 = oa1; ob2 = ob1; oa1 = oa_n; ob1 = ob_n;
    ids_prev = ids_n;
    nrows_prev = (int)((unsigned)ez >> 24);
    ex = __builtin_amdgcn_readfirstlane((int)e_n.x); ey = __builtin_amdgcn_readfirstlane((int)e_n.y);
    ez = __builtin_amdgcn_readfirstlane((int)e_n.z); ew = __builtin_amdgcn_readfirstlane((int)e_n.w);
  }
}

// ---------------------------------------------------------------------------------------------
// k_pool_fwd_direct (round 4, C = 64): the tiled dense forward WITHOUT the LDS staging of the point records.
//
// Why: at the repo's own resolution (544x960: 4.5 M points, 6 552 tiles) k_pool_fwd_lean2 takes 93 us with every input
// resident in the Infinity Cache, 86 us with ALL feature gathers compiled out and 98 us with all stores compiled out
// (profiles/round4/pool_fwd_r2_ablation.txt): neither bytes nor gathers bound it.  Its per-workgroup timeline (same
// file) is a chain of dependent phases — descriptor 1.8-5 us -> rank table + row_ptr loads 4.8 -> depth gather -> LDS
// records -> barrier 3.3 -> closing flags -> barrier 1.3 -> 11 point steps of four gathers each 10 -> tail 2 = 27 us per
// tile, 3.4 rounds of 2 048 resident workgroups: the launch is the latency of that chain times the rounds.
//
// Here a group of 16 lanes (one output row at a time, as before) reads ITS OWN piece of the tile's point list straight from
// global memory, 16 points at a time, one per lane: the lane gathers the depth value of its point, derives the pixel row,
// and hands {pixel offset, depth, closing flag, output row} to the row's lanes with DPP row broadcasts — the scheme of
// k_pool_bwd_patch.  No record staging, no barrier in front of the point loop, 8 feature-row gathers in flight per group
// instead of 4, tables of chunk k+2 / depth values of chunk k+1 requested while chunk k is being accumulated.
//   * per-point table: ONE int32 `pt` = ranks_depth | closing << 31 (closing = last point of its output row);
//   * the output row of the k-th non-empty row of the launch is `ivl_rel[k]`, relative to its tile's first row; a group finds
//     its first k in the tile descriptor (number of rows closed before its piece), later ones by counting closing flags;
//   * pad lanes hold a sentinel whose depth offset and pixel offset lie beyond their buffers: range-checked buffer loads
//     return 0 * 0, the point loop has no bounds test (as in k_pool_fwd_lean2);
//   * rows cut by a piece boundary are combined after ONE barrier in piece order (same rule as k_pool_fwd_lean2); pieces
//     are ceil(n/16) points, so such rows may differ from k_pool_fwd_lean2 in the last bit; no atomics, run-to-run identical.
// Tile descriptor: 32 ints {first row, #rows, first point, #points, 0,0,0,0, g[16], 0 x 8}, g[j] = number of non-empty rows
// of the launch closed before group j's piece | (piece starts inside a row) << 31.
// ---------------------------------------------------------------------------------------------
constexpr int kPtSentinel = 0x3fffffff;

struct DirectChunk {       // what a lane holds about ITS point of a 16-point chunk
  float dval;              // depth value (0 for a pad lane)
  int px;                  // byte offset of the pixel's feature row (beyond the buffer for a pad lane)
  int flag;                // < 0: the point closes its output row
  int row;                 // output row of a closing point, relative to the tile's first row
  unsigned long long cm;   // wave ballot of `flag < 0`
};

// row broadcasts whose `old` operand is never read (bound_ctrl: a disabled source lane would yield 0; every lane is active where
// these are used): ONE v_mov_b32_dpp, without the zero-initialising move the plain form needs in front of it
template <int J>
__device__ __forceinline__ int dpp_bcast_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xf, 0xf, true);
}
template <int J>
__device__ __forceinline__ float dpp_bcast_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + J, 0xf, 0xf, true));
}

template <int J>
__device__ __forceinline__ void direct_point(const u32x4t a, const DirectChunk& ck, const __amdgpu_buffer_rsrc_t out_rsrc,
                                             unsigned lane_off, float4& acc, bool& pend, float4* s_head, int* s_head_row,
                                             int tid, int grp, int sub) {
  const float d = dpp_bcast_f<J>(ck.dval);
  acc = fma4(d, make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)), acc);
  if (ck.cm & (0x0001000100010001ull << J)) {          // wave-uniform: one of the wave's four groups closes a row at its point J
    const int cj = dpp_bcast_i<J>(ck.flag);        // every lane is active here (DPP reads other lanes)
    const int rj = dpp_bcast_i<J>(ck.row);
    if (cj < 0) {
      if (pend) {                                      // head partial of a row that an earlier piece started
        s_head[tid] = acc;
        if (sub == 0) s_head_row[grp] = rj;
        pend = false;
      } else {
        const u32x4t o = {__float_as_uint(acc.x), __float_as_uint(acc.y), __float_as_uint(acc.z), __float_as_uint(acc.w)};
        __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, ((unsigned)rj << 8) | lane_off, 0, 2 /* nt */);
      }
      acc = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

template <int J0>
__device__ __forceinline__ void direct_batch8(const __amdgpu_buffer_rsrc_t feat_rsrc, const __amdgpu_buffer_rsrc_t out_rsrc,
                                              unsigned lane_off, const DirectChunk& ck, float4& acc, bool& pend, float4* s_head,
                                              int* s_head_row, int tid, int grp, int sub) {
#define OMNIHD_G(K) const u32x4t a##K = __builtin_amdgcn_raw_buffer_load_b128(feat_rsrc, (unsigned)dpp_bcast_i<J0 + K>(ck.px) | lane_off, 0, 0);
  OMNIHD_G(0) OMNIHD_G(1) OMNIHD_G(2) OMNIHD_G(3) OMNIHD_G(4) OMNIHD_G(5) OMNIHD_G(6) OMNIHD_G(7)
#undef OMNIHD_G
  direct_point<J0 + 0>(a0, ck, out_rsrc, lane_off, acc, pend, s_head, s_head_row, tid, grp, sub);
  direct_point<J0 + 1>(a1, ck, out_rsrc, lane_off, acc, pend, s_head, s_head_row, tid, grp, sub);
  direct_point<J0 + 2>(a2, ck, out_rsrc, lane_off, acc, pend, s_head, s_head_row, tid, grp, sub);
  direct_point<J0 + 3>(a3, ck, out_rsrc, lane_off, acc, pend, s_head, s_head_row, tid, grp, sub);
  direct_point<J0 + 4>(a4, ck, out_rsrc, lane_off, acc, pend, s_head, s_head_row, tid, grp, sub);
  direct_point<J0 + 5>(a5, ck, out_rsrc, lane_off, acc, pend, s_head, s_head_row, tid, grp, sub);
  direct_point<J0 + 6>(a6, ck, out_rsrc, lane_off, acc, pend, s_head, s_head_row, tid, grp, sub);
  direct_point<J0 + 7>(a7, ck, out_rsrc, lane_off, acc, pend, s_head, s_head_row, tid, grp, sub);
}

// One tile of the direct forward.  `prev_row_ptr` (device-built plans, empty_rows_kept == 2): the CSR of the tables that filled
// `out` last — a row that is empty now is zero-filled only if it was NOT empty then.
__device__ __forceinline__ void direct_tile(
    const float* __restrict__ depth, unsigned depth_bytes, const float* __restrict__ feat, unsigned feat_bytes,
    const int* __restrict__ pt, const int* __restrict__ ivl_rel, unsigned ivl_bytes, const int* __restrict__ dsc,
    const int* __restrict__ row_ptr, const int* __restrict__ prev_row_ptr, float* __restrict__ out, int fhw, int dfhw,
    float inv_fhw, float inv_dfhw, int empty_rows_kept, float4* s_tail, float4* s_head, int* s_head_row, int* s_tail_flags) {
  constexpr int C4 = 16, G = kBlock / C4, GPW = 64 / C4;
  const int Ra = dsc[0], nrows = dsc[1], Pa = dsc[2], npts = dsc[3];
  if (nrows <= 0) return;
  const int tid = threadIdx.x;
  const int sub = tid % C4;
  const int grp = tid / C4;
  const int lane = tid & 63;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4* out4 = reinterpret_cast<float4*>(out);

  // ---- my piece of the point list; its first table words are requested before anything else --------------------
  const int Wp = (npts + G - 1) / G;
  const int q0 = Pa + min(grp * Wp, npts);
  const int q1 = Pa + min(grp * Wp + Wp, npts);
  const int nchunks = (Wp + kPatch - 1) / kPatch;            // the same for every group (scalar trip count)
  int pt_a = kPtSentinel, pt_b = kPtSentinel, pt_last = -1, gi = 0;
  if (npts > 0) {
    if (q0 + sub < q1) pt_a = pt[q0 + sub];
    if (q0 + kPatch + sub < q1) pt_b = pt[q0 + kPatch + sub];
    if (q1 > q0) pt_last = pt[q1 - 1];
    gi = dsc[8 + grp];
  }

  // ---- zero-fill the rows no point falls into (unless the caller's buffer holds those zeros already) -----------
  if (empty_rows_kept != 1 && !(nrows == 1 && npts > 0)) {
    const int gw = lane / C4;
    for (int base = 0; base < nrows; base += kBlock) {
      const int i = base + tid;
      bool empty = false;
      if (i < nrows) {
        empty = row_ptr[Ra + i + 1] == row_ptr[Ra + i];
        if (empty_rows_kept == 2 && empty) empty = prev_row_ptr[Ra + i + 1] != prev_row_ptr[Ra + i];
      }
      const unsigned long long m = __ballot(empty);
      if (m == 0ull) continue;
      const int wave_row0 = Ra + base + (tid & ~63);
      for (int k = 0; k < 64; k += GPW) {
        if (((m >> k) & ((1ull << GPW) - 1ull)) == 0ull) continue;
        if ((m >> (k + gw)) & 1ull) store_row(out4 + (size_t)(wave_row0 + k + gw) * C4 + sub, zero4, true);
      }
    }
  }
  if (npts == 0) return;

  const __amdgpu_buffer_rsrc_t feat_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)feat_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t depth_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)depth, 0, (int)depth_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ivl_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ivl_rel, 0, (int)ivl_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t out_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(out + (size_t)Ra * (C4 * 4)), 0, nrows << 8, 0x00020000);
  const unsigned lane_off = (unsigned)sub << 4;
  int ivl = gi & 0x7fffffff;                                 // non-empty rows of the launch closed before my piece
  const bool was_pending = (gi < 0) && (q0 < q1);            // my first point continues a row an earlier piece started
  bool pend = was_pending;
  const unsigned below = (1u << sub) - 1u;
  const int gsh = lane & 48;

  // lane's own point of a chunk: depth gather + pixel offset + output row of a closing point (all requests, no waits)
  auto stage = [&](int p) {
    DirectChunk ck;
    const int rd = p & 0x7fffffff;
    ck.flag = p;
    ck.dval = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(depth_rsrc, (unsigned)rd << 2, 0, 0));   // pad: beyond -> 0
    const int n = div_const(rd, dfhw, inv_dfhw);
    const int q = div_const(rd, fhw, inv_fhw);
    const int pixel = n * fhw + (rd - q * fhw);
    ck.px = (p == kPtSentinel) ? (int)0x80000000 : (pixel << 8);
    ck.cm = __builtin_amdgcn_ballot_w64(p < 0);
    const unsigned m16 = (unsigned)(ck.cm >> gsh) & 0xffffu;
    const int k = ivl + __builtin_popcount(m16 & below);
    ck.row = (int)__builtin_amdgcn_raw_buffer_load_b32(ivl_rsrc, (p < 0) ? ((unsigned)k << 2) : 0xfffffffcu, 0, 0);
    ivl += __builtin_popcount(m16);
    return ck;
  };

  float4 acc = zero4;
  DirectChunk nxt = stage(pt_a);
  for (int c = 0; c < nchunks; ++c) {
    const DirectChunk ck = nxt;
    pt_a = pt_b;
    pt_b = kPtSentinel;
    if (q0 + (c + 2) * kPatch + sub < q1) pt_b = pt[q0 + (c + 2) * kPatch + sub];
    if (c + 1 < nchunks) nxt = stage(pt_a);
    direct_batch8<0>(feat_rsrc, out_rsrc, lane_off, ck, acc, pend, s_head, s_head_row, tid, grp, sub);
    if (c * kPatch + 8 < Wp)                                 // (scalar) the last chunk of a piece may hold <= 8 points: no pad gathers
      direct_batch8<8>(feat_rsrc, out_rsrc, lane_off, ck, acc, pend, s_head, s_head_row, tid, grp, sub);
  }

  // ---- rows cut by a piece boundary: tails of earlier pieces + my head partial, in piece order ------------------
  const bool pending = pend;                                 // started inside a row and never closed it
  const bool open_end = (q1 <= q0) || (pt_last >= 0);        // the piece ends inside a row (an empty piece lies inside one)
  s_tail[tid] = acc;
  if (sub == 0) s_tail_flags[grp] = (open_end ? 1 : 0) | ((pending || q1 <= q0) ? 2 : 0);
  __syncthreads();
  if (was_pending && !pending) {
    int g0 = grp;
    while (g0 > 0) {
      const int f = s_tail_flags[g0 - 1];
      if (!(f & 1)) break;
      --g0;
      if (!(f & 2)) break;
    }
    float4 tsum = zero4;
    for (int g = g0; g < grp; ++g) tsum = add4(tsum, s_tail[g * C4 + sub]);
    tsum = add4(tsum, s_head[tid]);
    const u32x4t o = {__float_as_uint(tsum.x), __float_as_uint(tsum.y), __float_as_uint(tsum.z), __float_as_uint(tsum.w)};
    __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, ((unsigned)s_head_row[grp] << 8) | lane_off, 0, 2);
  }
}

// DEV = false: the schedule has n_slots = 8 * tiles_per_xcd descriptors known to the host, one workgroup each.
// DEV = true (plans built by csrc/pool_plan.hip): the number of slots per XCD is hdr[3] ON THE DEVICE; the host launches 8 * k
// workgroups with k >= hdr[3] (the plan's capacity, or the exact count once it has travelled to the host) and the workgroups
// of slots beyond hdr[3] leave at once.  (A form that lets a workgroup walk several slots was tried: the loop costs 40 VGPRs.)
template <bool DEV>
__global__ __launch_bounds__(kBlock) void k_pool_fwd_direct(
    const float* __restrict__ depth, unsigned depth_bytes, const float* __restrict__ feat, unsigned feat_bytes,
    const int* __restrict__ pt, const int* __restrict__ ivl_rel, unsigned ivl_bytes, const int* __restrict__ desc32,
    const int* __restrict__ row_ptr, float* __restrict__ out, int tiles_per_xcd, int fhw, int dfhw, float inv_fhw,
    float inv_dfhw, int empty_rows_kept, const int* __restrict__ hdr, const int* __restrict__ prev_row_ptr) {
  constexpr int G = kBlock / 16;
  __shared__ float4 s_tail[kBlock];
  __shared__ float4 s_head[kBlock];
  __shared__ int s_head_row[G];
  __shared__ int s_tail_flags[G];
  const int per = DEV ? hdr[3] : tiles_per_xcd;
  if ((int)(blockIdx.x >> 3) >= per) return;
  const int* dsc = desc32 + ((size_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3)) * 32;
  direct_tile(depth, depth_bytes, feat, feat_bytes, pt, ivl_rel, ivl_bytes, dsc, row_ptr, DEV ? prev_row_ptr : nullptr, out, fhw,
              dfhw, inv_fhw, inv_dfhw, empty_rows_kept, s_tail, s_head, s_head_row, s_tail_flags);
}

__global__ __launch_bounds__(kBlock) void k_pool_bwd_generic(
    const float* __restrict__ og, const float* __restrict__ depth, const float* __restrict__ feat,
    const int* __restrict__ ranks_depth, const int* __restrict__ ranks_feat,
    const int* __restrict__ ranks_bev, const int* __restrict__ starts,
    const int* __restrict__ lengths, float* __restrict__ depth_grad,
    float* __restrict__ feat_grad, int c, int n_intervals) {
  // thread per (interval, channel) for feat_grad; channel 0's thread also does depth_grad.
  const int64_t total = (int64_t)n_intervals * c;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * kBlock) {
    const int iv = (int)(t / c);
    const int ch = (int)(t % c);
    const int s = starts[iv];
    const int len = lengths[iv];
    if (len <= 0) continue;
    float acc = 0.f;
    for (int i = 0; i < len; ++i)
      acc = fmaf(og[(size_t)ranks_bev[s + i] * c + ch], depth[ranks_depth[s + i]], acc);
    feat_grad[(size_t)ranks_feat[s] * c + ch] = acc;
    if (ch == 0) {
      for (int i = 0; i < len; ++i) {
        const float* g = og + (size_t)ranks_bev[s + i] * c;
        const float* x = feat + (size_t)ranks_feat[s + i] * c;
        float dsum = 0.f;
        for (int k = 0; k < c; ++k) dsum = fmaf(g[k], x[k], dsum);
        depth_grad[ranks_depth[s + i]] = dsum;
      }
    }
  }
}

inline bool vec_ok(int c, const void* a, const void* b, const void* c3 = nullptr) {
  if (c % 4 != 0) return false;
  const int c4 = c / 4;
  if (c4 > 64 || (64 % c4) != 0) return false;
  auto al = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  return al(a) && al(b) && al(c3);
}

template <bool DENSE>
int launch_fwd(const float* depth, const float* feat, const int* ranks_depth,
               const int* ranks_feat, const int* ranks_bev, const int* starts,
               const int* lengths_or_rowptr, float* out, int c, int n_units,
               hipStream_t st) {
  if (n_units == 0) return OMNIHD_OK;
  if (vec_ok(c, feat, out)) {
    const int c4 = c / 4;
    const int G = kBlock / c4;
    // ~4 rows per group and workgroup so a deferred long interval is found early.
    const int grid = grid_for(n_units, G * 4);
    const float4* f4 = reinterpret_cast<const float4*>(feat);
    float4* o4 = reinterpret_cast<float4*>(out);
#define OMNIHD_FWD_CASE(C4)                                                                  \
  case C4:                                                                                   \
    hipLaunchKernelGGL((k_pool_fwd<C4, DENSE>), dim3(grid), dim3(kBlock), 0, st, depth, f4,  \
                       ranks_depth, ranks_feat, ranks_bev, starts, lengths_or_rowptr, o4,    \
                       n_units);                                                             \
    break;
    switch (c4) {
      OMNIHD_FWD_CASE(1)
      OMNIHD_FWD_CASE(2)
      OMNIHD_FWD_CASE(4)
      OMNIHD_FWD_CASE(8)
      OMNIHD_FWD_CASE(16)
      OMNIHD_FWD_CASE(32)
      OMNIHD_FWD_CASE(64)
      default:
        set_error("unreachable c4=%d", c4);
        return OMNIHD_ERR_ARG;
    }
#undef OMNIHD_FWD_CASE
  } else {
    const int grid = grid_for((int64_t)n_units * c, kBlock);
    hipLaunchKernelGGL((k_pool_fwd_generic<DENSE>), dim3(grid), dim3(kBlock), 0, st, depth, feat,
                       ranks_depth, ranks_feat, ranks_bev, starts, lengths_or_rowptr, out, c,
                       n_units);
  }
  return check_launch(DENSE ? "bev_pool_v2_fwd_csr" : "bev_pool_v2_fwd");
}

}  // namespace
}  // namespace omnihd

using namespace omnihd;

extern "C" int omnihd_bev_pool_v2_fwd(const float* depth, const float* feat,
                                      const int* ranks_depth, const int* ranks_feat,
                                      const int* ranks_bev, const int* interval_starts,
                                      const int* interval_lengths, float* out, int c,
                                      int n_intervals, void* stream) {
  OMNIHD_REQUIRE(c > 0 && n_intervals >= 0, "c > 0 and n_intervals >= 0");
  if (n_intervals == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(depth && feat && ranks_depth && ranks_feat && ranks_bev && interval_starts &&
                     interval_lengths && out,
                 "null pointer");
  return launch_fwd<false>(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                           interval_lengths, out, c, n_intervals, (hipStream_t)stream);
}

extern "C" int omnihd_tile_desc(const int* row_ptr, const int* tile_row, const int* tile_order,
                                int n_tiles, int* tile_desc, void* stream) {
  OMNIHD_REQUIRE(n_tiles > 0 && row_ptr && tile_row && tile_desc, "arguments");
  const int n_slots = 8 * ((n_tiles + 7) / 8);
  hipLaunchKernelGGL(k_tile_desc, dim3(grid_for(n_slots, kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, row_ptr, tile_row, tile_order, n_slots, n_tiles,
                     reinterpret_cast<int4*>(tile_desc));
  return check_launch("tile_desc");
}

extern "C" int omnihd_bev_pool_v2_fwd_lean(const float* depth, const float* feat, const int* ranks_depth,
                                           const int* row_ptr, const int* tile_desc, int n_tiles, float* out, int c,
                                           int n_rows, int n_points, int d_bins, int fhw, int n_feat_rows,
                                           int empty_rows_kept, void* stream) {
  OMNIHD_REQUIRE(c > 0 && n_rows >= 0 && n_tiles > 0 && n_points >= 0 && d_bins > 0 && fhw > 0 && n_feat_rows >= 0, "sizes");
  if (n_rows == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(depth && feat && row_ptr && out && tile_desc && (n_points == 0 || ranks_depth), "null pointer");
  OMNIHD_REQUIRE(vec_ok(c, feat, out) && (reinterpret_cast<uintptr_t>(tile_desc) & 15u) == 0 && n_tiles % 8 == 0,
                 "C % 4 == 0, 16-byte aligned pointers, 8*k schedule slots");
  OMNIHD_REQUIRE((long long)d_bins * fhw < (1ll << 30), "D * fH * fW too large");
  hipStream_t st = (hipStream_t)stream;
  const int tiles_per_xcd = n_tiles / 8;
  const dim3 grid(n_tiles);
  const float4* f4 = reinterpret_cast<const float4*>(feat);
  const int4* td = reinterpret_cast<const int4*>(tile_desc);
  float4* o4 = reinterpret_cast<float4*>(out);
  const int dfhw = d_bins * fhw;
  // k_pool_fwd_lean2 (issue-lean point loop) is the default; OMNIHD_POOL_LEAN2=0 selects the first lean kernel.  It needs
  // 32-bit byte offsets into the feature table (buffer loads): the caller's feature table ends at the last pixel a rank
  // names, and the wrapper passes its size in rows through n_feat_rows (0: unknown -> first lean kernel).
  static const bool lean2 = [] { const char* e = getenv("OMNIHD_POOL_LEAN2"); return !(e && e[0] == '0'); }();
  const long long feat_bytes = (long long)n_feat_rows * c * 4;
  OMNIHD_REQUIRE(!empty_rows_kept || (lean2 && n_feat_rows > 0 && feat_bytes < (1ll << 31)),
                 "empty_rows_kept needs the second-generation kernel (n_feat_rows given, feature table below 2 GiB)");
  if (lean2 && n_feat_rows > 0 && feat_bytes < (1ll << 31)) {
#define OMNIHD_LEAN2_CASE(C4)                                                                                     \
  case C4:                                                                                                        \
    hipLaunchKernelGGL((k_pool_fwd_lean2<C4, 4>), grid, dim3(kBlock), 0, st, depth, feat, (unsigned)feat_bytes,   \
                       ranks_depth, row_ptr, td, out, tiles_per_xcd, fhw, dfhw, 1.0f / (float)fhw,               \
                       1.0f / (float)dfhw, empty_rows_kept);                                                      \
    break;
    switch (c / 4) {
      OMNIHD_LEAN2_CASE(1)
      OMNIHD_LEAN2_CASE(2)
      OMNIHD_LEAN2_CASE(4)
      OMNIHD_LEAN2_CASE(8)
      OMNIHD_LEAN2_CASE(16)
      OMNIHD_LEAN2_CASE(32)
      OMNIHD_LEAN2_CASE(64)
      default:
        set_error("bev_pool_v2_fwd_lean: C/4 must be a power of two <= 64");
        return OMNIHD_ERR_ARG;
    }
#undef OMNIHD_LEAN2_CASE
    return check_launch("bev_pool_v2_fwd_lean(2)");
  }
#define OMNIHD_LEAN_CASE(C4)                                                                                     \
  case C4:                                                                                                       \
    hipLaunchKernelGGL((k_pool_fwd_lean<C4, 4>), grid, dim3(kBlock), 0, st, depth, f4, ranks_depth, row_ptr, td, \
                       o4, tiles_per_xcd, fhw, dfhw, 1.0f / (float)fhw, 1.0f / (float)dfhw);                     \
    break;
  switch (c / 4) {
    OMNIHD_LEAN_CASE(1)
    OMNIHD_LEAN_CASE(2)
    OMNIHD_LEAN_CASE(4)
    OMNIHD_LEAN_CASE(8)
    OMNIHD_LEAN_CASE(16)
    OMNIHD_LEAN_CASE(32)
    OMNIHD_LEAN_CASE(64)
    default:
      set_error("bev_pool_v2_fwd_lean: C/4 must be a power of two <= 64");
      return OMNIHD_ERR_ARG;
  }
#undef OMNIHD_LEAN_CASE
  return check_launch("bev_pool_v2_fwd_lean");
}

extern "C" int omnihd_bev_pool_v2_fwd_direct(const float* depth, const float* feat, const int* pt, const int* ivl_rel,
                                             int n_intervals, const int* desc32, int n_slots, const int* row_ptr, float* out,
                                             int c, int n_rows, int n_points, int d_bins, int fhw, int n_feat_rows,
                                             int empty_rows_kept, void* stream) {
  OMNIHD_REQUIRE(c == 64, "the direct forward is written for C = 64 (use omnihd_bev_pool_v2_fwd_lean otherwise)");
  OMNIHD_REQUIRE(n_rows >= 0 && n_slots > 0 && n_slots % 8 == 0 && n_points >= 0 && n_intervals >= 0 && d_bins > 0 && fhw > 0 &&
                     n_feat_rows > 0, "sizes");
  if (n_rows == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(depth && feat && out && desc32 && (n_points == 0 || (pt && ivl_rel)), "null pointer");
  OMNIHD_REQUIRE(empty_rows_kept || row_ptr, "row_ptr is needed to zero-fill the empty rows");
  OMNIHD_REQUIRE(((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(desc32)) & 15u) == 0,
                 "16-byte aligned feat / out / desc32");
  const long long feat_bytes = (long long)n_feat_rows * c * 4;
  const long long depth_bytes = (long long)n_feat_rows * d_bins * 4;
  OMNIHD_REQUIRE(feat_bytes < (1ll << 31) && depth_bytes < (1ll << 32) - 8 && (long long)n_feat_rows * d_bins < kPtSentinel,
                 "feature table below 2 GiB and depth tensor below 4 GiB (32-bit gather offsets)");
  OMNIHD_REQUIRE((long long)d_bins * fhw < (1ll << 30), "D * fH * fW too large");
  const int dfhw = d_bins * fhw;
  hipLaunchKernelGGL(k_pool_fwd_direct<false>, dim3(n_slots), dim3(kBlock), 0, (hipStream_t)stream, depth, (unsigned)depth_bytes, feat,
                     (unsigned)feat_bytes, pt, ivl_rel, (unsigned)((long long)n_intervals * 4), desc32, row_ptr, out, n_slots / 8, fhw,
                     dfhw, 1.0f / (float)fhw, 1.0f / (float)dfhw, empty_rows_kept ? 1 : 0, (const int*)nullptr, (const int*)nullptr);
  return check_launch("bev_pool_v2_fwd_direct");
}

extern "C" int omnihd_bev_pool_v2_fwd_direct_dev(const float* depth, const float* feat, const int* pt, const int* ivl_rel,
                                                 long long ivl_capacity, const int* desc32, const int* hdr, int launch_slots,
                                                 const int* row_ptr, const int* prev_row_ptr, float* out, int c, int n_rows,
                                                 int d_bins, int fhw, int n_feat_rows, int empty_rows_mode, void* stream) {
  OMNIHD_REQUIRE(c == 64, "the direct forward is written for C = 64");
  OMNIHD_REQUIRE(n_rows > 0 && launch_slots > 0 && launch_slots % 8 == 0 && ivl_capacity > 0 && d_bins > 0 && fhw > 0 &&
                     n_feat_rows > 0, "sizes");
  OMNIHD_REQUIRE(depth && feat && out && desc32 && hdr && pt && ivl_rel && row_ptr, "null pointer");
  OMNIHD_REQUIRE(empty_rows_mode == 0 || empty_rows_mode == 1 || (empty_rows_mode == 2 && prev_row_ptr),
                 "empty_rows_mode: 0 fill, 1 kept, 2 kept relative to prev_row_ptr");
  OMNIHD_REQUIRE(((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(desc32)) & 15u) == 0,
                 "16-byte aligned feat / out / desc32");
  const long long feat_bytes = (long long)n_feat_rows * c * 4;
  const long long depth_bytes = (long long)n_feat_rows * d_bins * 4;
  OMNIHD_REQUIRE(feat_bytes < (1ll << 31) && depth_bytes < (1ll << 32) - 8 && (long long)n_feat_rows * d_bins < kPtSentinel,
                 "feature table below 2 GiB and depth tensor below 4 GiB (32-bit gather offsets)");
  OMNIHD_REQUIRE((long long)d_bins * fhw < (1ll << 30) && ivl_capacity * 4 < (1ll << 32) - 8, "D * fH * fW / interval table too large");
  const int dfhw = d_bins * fhw;
  hipLaunchKernelGGL(k_pool_fwd_direct<true>, dim3(launch_slots), dim3(kBlock), 0, (hipStream_t)stream, depth, (unsigned)depth_bytes,
                     feat, (unsigned)feat_bytes, pt, ivl_rel, (unsigned)(ivl_capacity * 4), desc32, row_ptr, out, 0, fhw, dfhw,
                     1.0f / (float)fhw, 1.0f / (float)dfhw, empty_rows_mode, hdr, prev_row_ptr);
  return check_launch("bev_pool_v2_fwd_direct_dev");
}

extern "C" int omnihd_bev_pool_v2_fwd_csr(const float* depth, const float* feat,
                                          const int* ranks_depth, const int* ranks_feat,
                                          const int* ranks_row, const int* row_ptr,
                                          const int* tile_desc, int n_tiles, float* out, int c,
                                          int n_rows, int n_points, void* stream) {
  OMNIHD_REQUIRE(c > 0 && n_rows >= 0 && n_tiles >= 0 && n_points >= 0, "sizes");
  if (n_rows == 0) return OMNIHD_OK;
  // ranks_* may be null when the plan holds no point at all (every row is then written as zeros)
  OMNIHD_REQUIRE(depth && feat && row_ptr && out, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (tile_desc != nullptr && ranks_row != nullptr && n_tiles > 0 && vec_ok(c, feat, out) &&
      (reinterpret_cast<uintptr_t>(tile_desc) & 15u) == 0) {
    const int tiles_per_xcd = (n_tiles + 7) / 8;
    const dim3 grid(tiles_per_xcd * 8);
    const float4* f4 = reinterpret_cast<const float4*>(feat);
    const int4* td = reinterpret_cast<const int4*>(tile_desc);
    float4* o4 = reinterpret_cast<float4*>(out);
    static const int unroll = [] { const char* e = getenv("OMNIHD_FWD_UNROLL"); const int u = e ? atoi(e) : 4; return (u == 8) ? 8 : 4; }();
#define OMNIHD_TILE_CASE(C4)                                                                   \
  case C4:                                                                                     \
    if (unroll == 8)                                                                           \
      hipLaunchKernelGGL((k_pool_fwd_tiles<C4, 8>), grid, dim3(kBlock), 0, st, depth, f4,      \
                         ranks_depth, ranks_feat, ranks_row, row_ptr, td, o4, tiles_per_xcd,   \
                         n_points);                                                            \
    else                                                                                       \
      hipLaunchKernelGGL((k_pool_fwd_tiles<C4, 4>), grid, dim3(kBlock), 0, st, depth, f4,      \
                         ranks_depth, ranks_feat, ranks_row, row_ptr, td, o4, tiles_per_xcd,   \
                         n_points);                                                            \
    break;
    switch (c / 4) {
      OMNIHD_TILE_CASE(1)
      OMNIHD_TILE_CASE(2)
      OMNIHD_TILE_CASE(4)
      OMNIHD_TILE_CASE(8)
      OMNIHD_TILE_CASE(16)
      OMNIHD_TILE_CASE(32)
      OMNIHD_TILE_CASE(64)
      default:
        set_error("unreachable c4=%d", c / 4);
        return OMNIHD_ERR_ARG;
    }
#undef OMNIHD_TILE_CASE
    return check_launch("bev_pool_v2_fwd_csr(tiles)");
  }
  return launch_fwd<true>(depth, feat, ranks_depth, ranks_feat, nullptr, nullptr, row_ptr, out, c,
                          n_rows, st);
}

extern "C" int omnihd_bev_pool_v2_bwd(const float* out_grad, const float* depth,
                                      const float* feat, const int* ranks_depth,
                                      const int* ranks_feat, const int* ranks_bev,
                                      const int* interval_starts, const int* interval_lengths,
                                      float* depth_grad, float* feat_grad, int c,
                                      int n_intervals, void* stream) {
  OMNIHD_REQUIRE(c > 0 && n_intervals >= 0, "c > 0 and n_intervals >= 0");
  if (n_intervals == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(out_grad && depth && feat && ranks_depth && ranks_feat && ranks_bev &&
                     interval_starts && interval_lengths && depth_grad && feat_grad,
                 "null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (vec_ok(c, out_grad, feat, feat_grad)) {
    const int c4 = c / 4;
    const int G = kBlock / c4;
    const int grid = grid_for(n_intervals, G * 2);
    const float4* og4 = reinterpret_cast<const float4*>(out_grad);
    const float4* f4 = reinterpret_cast<const float4*>(feat);
    float4* fg4 = reinterpret_cast<float4*>(feat_grad);
#define OMNIHD_BWD_CASE(C4)                                                                   \
  case C4:                                                                                    \
    hipLaunchKernelGGL((k_pool_bwd<C4>), dim3(grid), dim3(kBlock), 0, st, og4, depth, f4,     \
                       ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths, \
                       depth_grad, fg4, n_intervals);                                         \
    break;
    switch (c4) {
      OMNIHD_BWD_CASE(1)
      OMNIHD_BWD_CASE(2)
      OMNIHD_BWD_CASE(4)
      OMNIHD_BWD_CASE(8)
      OMNIHD_BWD_CASE(16)
      OMNIHD_BWD_CASE(32)
      OMNIHD_BWD_CASE(64)
      default:
        set_error("unreachable c4=%d", c4);
        return OMNIHD_ERR_ARG;
    }
#undef OMNIHD_BWD_CASE
  } else {
    const int grid = grid_for((int64_t)n_intervals * c, kBlock);
    hipLaunchKernelGGL(k_pool_bwd_generic, dim3(grid), dim3(kBlock), 0, st, out_grad, depth, feat,
                       ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths,
                       depth_grad, feat_grad, c, n_intervals);
  }
  return check_launch("bev_pool_v2_bwd");
}

extern "C" int omnihd_bev_pool_v2_bwd_sched(const float* out_grad, const float* depth,
                                            const float* feat, const int* ranks_depth,
                                            const int* ranks_row, const int* pix_desc,
                                            int groups_per_xcd, float* depth_grad,
                                            float* feat_grad, int c, void* stream) {
  OMNIHD_REQUIRE(c > 0 && groups_per_xcd >= 0, "sizes");
  if (groups_per_xcd == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(out_grad && depth && feat && pix_desc && depth_grad && feat_grad, "null pointer");
  OMNIHD_REQUIRE(c % 4 == 0 && (c / 4 == 16 || c / 4 == 8 || c / 4 == 4 || c / 4 == 2 || c / 4 == 1),
                 "scheduled backward supports C in {4,8,16,32,64}");
  OMNIHD_REQUIRE(((reinterpret_cast<uintptr_t>(out_grad) | reinterpret_cast<uintptr_t>(feat) |
                   reinterpret_cast<uintptr_t>(feat_grad) | reinterpret_cast<uintptr_t>(pix_desc)) & 15u) == 0,
                 "16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  const float4* og4 = reinterpret_cast<const float4*>(out_grad);
  const float4* f4 = reinterpret_cast<const float4*>(feat);
  float4* fg4 = reinterpret_cast<float4*>(feat_grad);
  const int4* pd = reinterpret_cast<const int4*>(pix_desc);
#define OMNIHD_BWDS_CASE(C4, U)                                                                 \
  case C4: {                                                                                    \
    const int G = kBlock / C4;                                                                  \
    const dim3 grid(8 * ((groups_per_xcd + G - 1) / G));                                        \
    hipLaunchKernelGGL((k_pool_bwd_sched<C4, U>), grid, dim3(kBlock), 0, st, og4, depth, f4,    \
                       ranks_depth, ranks_row, pd, depth_grad, fg4, groups_per_xcd);            \
  } break;
  switch (c / 4) {
    OMNIHD_BWDS_CASE(1, 1)
    OMNIHD_BWDS_CASE(2, 2)
    OMNIHD_BWDS_CASE(4, 4)
    OMNIHD_BWDS_CASE(8, 4)
    OMNIHD_BWDS_CASE(16, 4)
    default:
      return OMNIHD_ERR_ARG;
  }
#undef OMNIHD_BWDS_CASE
  return check_launch("bev_pool_v2_bwd_sched");
}

extern "C" int omnihd_bev_pool_v2_bwd_patch(const float* out_grad, const float* depth, const float* feat,
                                            const int* ranks_depth, const int* ranks_row, const int* pix_ptr,
                                            const int* patch_order, int n_slots, int n_img, int d_bins, int fhw,
                                            long long n_rows, float* depth_grad, float* feat_grad, int c, void* stream) {
  OMNIHD_REQUIRE(c == 64, "the patch backward is written for C = 64 (use omnihd_bev_pool_v2_bwd_sched otherwise)");
  OMNIHD_REQUIRE(n_slots >= 0 && n_slots % 8 == 0 && n_img > 0 && d_bins > 0 && fhw > 0 && n_rows > 0, "sizes");
  if (n_slots == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(out_grad && depth && feat && ranks_row && pix_ptr && patch_order && depth_grad && feat_grad, "null pointer");
  OMNIHD_REQUIRE(ranks_depth || d_bins <= 127, "packed (row | bin << 24) tables hold at most 127 depth bins");
  OMNIHD_REQUIRE(((reinterpret_cast<uintptr_t>(out_grad) | reinterpret_cast<uintptr_t>(feat) |
                   reinterpret_cast<uintptr_t>(feat_grad)) & 15u) == 0, "16-byte alignment");
  OMNIHD_REQUIRE(n_rows * 256 < (1ll << 32) && n_rows < 0x00ffffff, "out_grad must stay below 4 GiB (32-bit gather offsets)");
  OMNIHD_REQUIRE((long long)d_bins * fhw * n_img < (1ll << 31), "depth tensor too large for int32 ranks");
  const size_t lds = (size_t)2 * d_bins * kPatch * sizeof(float);
  OMNIHD_REQUIRE(lds <= 64 * 1024, "too many depth bins for the LDS patch buffers");
  hipStream_t st = (hipStream_t)stream;
  const int patches_per_img = (fhw + kPatch - 1) / kPatch;
  if (ranks_depth == nullptr)     // `ranks_row` is the packed table (row | depth bin << 24)
    hipLaunchKernelGGL(k_pool_bwd_patch<true>, dim3(n_slots), dim3(kBlock), lds, st, out_grad, (unsigned)(n_rows * 256), depth,
                       reinterpret_cast<const float4*>(feat), ranks_depth, ranks_row, pix_ptr, patch_order, n_slots / 8,
                       patches_per_img, fhw, d_bins, 1.0f / (float)fhw, depth_grad, reinterpret_cast<float4*>(feat_grad));
  else
    hipLaunchKernelGGL(k_pool_bwd_patch<false>, dim3(n_slots), dim3(kBlock), lds, st, out_grad, (unsigned)(n_rows * 256), depth,
                       reinterpret_cast<const float4*>(feat), ranks_depth, ranks_row, pix_ptr, patch_order, n_slots / 8,
                       patches_per_img, fhw, d_bins, 1.0f / (float)fhw, depth_grad, reinterpret_cast<float4*>(feat_grad));
  return check_launch("bev_pool_v2_bwd_patch");
}

extern "C" int omnihd_bev_pool_v2_bwd_stream_lds_bytes(int rows_per_stage, int d_bins) {
  (void)d_bins;                                    // both depth blocks are laid out for 64 bins (16 cells per lane)
  return (rows_per_stage + 1) * 256 + 2 * 64 * kPatch * (int)sizeof(float);
}

extern "C" int omnihd_bev_pool_v2_bwd_stream(const float* out_grad, const float* depth, const float* feat, const int* pt_word,
                                             long long n_points, const int* uniq_rows, long long n_uniq, const int* px_off,
                                             long long n_off, const int* stream, long long n_entries, const int* stream_ptr,
                                             int n_streams, int n_img, int d_bins, int fh, int fw, int patch_w,
                                             int rows_per_stage, long long n_rows, float* depth_grad, float* feat_grad, int c,
                                             void* stream_handle) {
  OMNIHD_REQUIRE(c == 64, "the stream backward is written for C = 64 (use omnihd_bev_pool_v2_bwd_sched otherwise)");
  OMNIHD_REQUIRE(n_streams >= 0 && n_streams % 8 == 0 && n_img > 0 && d_bins > 0 && fh > 0 && fw > 0 && n_rows > 0, "sizes");
  OMNIHD_REQUIRE(patch_w == 16 || patch_w == 8 || patch_w == 4, "patch width must be 16, 8 or 4 pixels");
  OMNIHD_REQUIRE(rows_per_stage == 32 || rows_per_stage == 48 || rows_per_stage == 64, "rows per stage: 32, 48 or 64");
  if (n_streams == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(out_grad && depth && feat && pt_word && uniq_rows && px_off && stream && stream_ptr && depth_grad && feat_grad,
                 "null pointer");
  OMNIHD_REQUIRE(d_bins <= 64, "the depth cells of a patch are held as 16 registers x 64 lanes: at most 64 depth bins");
  OMNIHD_REQUIRE(((reinterpret_cast<uintptr_t>(out_grad) | reinterpret_cast<uintptr_t>(feat) |
                   reinterpret_cast<uintptr_t>(feat_grad) | reinterpret_cast<uintptr_t>(stream)) & 15u) == 0, "16-byte alignment");
  const long long n_px = (long long)n_img * fh * fw;
  OMNIHD_REQUIRE(n_rows * 256 < (1ll << 32) - 256 && n_rows < 0x00ffffff, "out_grad must stay below 4 GiB (32-bit gather offsets)");
  OMNIHD_REQUIRE(n_px * d_bins * 4 < (1ll << 32) - 256 && n_px * 256 < (1ll << 32) - 256, "depth / feat must stay below 4 GiB");
  OMNIHD_REQUIRE(n_points * 4 < (1ll << 31) && n_uniq * 4 < (1ll << 31) && n_off * 4 < (1ll << 31) && n_entries * 16 < (1ll << 31),
                 "tables must stay below 2 GiB");
  const size_t lds = (size_t)omnihd_bev_pool_v2_bwd_stream_lds_bytes(rows_per_stage, d_bins);
  hipStream_t st = (hipStream_t)stream_handle;
  const int pw_shift = patch_w == 16 ? 4 : (patch_w == 8 ? 3 : 2);
#define OMNIHD_STREAM_CASE(RQ)                                                                                             \
  case RQ:                                                                                                                 \
    hipLaunchKernelGGL(k_pool_bwd_stream<RQ>, dim3(n_streams), dim3(64), lds, st, out_grad, (unsigned)(n_rows * 256), depth, \
                       (unsigned)(n_px * d_bins * 4), feat, (unsigned)(n_px * 256), pt_word, (unsigned)(n_points * 4),      \
                       uniq_rows, (unsigned)(n_uniq * 4), px_off, (unsigned)(n_off * 4), reinterpret_cast<const int4*>(stream), \
                       (unsigned)(n_entries * 16), stream_ptr, n_streams / 8, fh, fw, pw_shift, d_bins, depth_grad,          \
                       reinterpret_cast<float4*>(feat_grad));                                                              \
    break;
  switch (rows_per_stage / 4) {
    OMNIHD_STREAM_CASE(8) OMNIHD_STREAM_CASE(12) OMNIHD_STREAM_CASE(16)
    default:
      return OMNIHD_ERR_ARG;
  }
#undef OMNIHD_STREAM_CASE
  return check_launch("bev_pool_v2_bwd_stream");
}
