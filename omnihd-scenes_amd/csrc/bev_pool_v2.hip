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
// Tiles.  A TILE is a run of whole output rows of ~tile_items points + rows, or one single row of any length (the plan cuts
// them so that every workgroup gets the same amount of work although 40 % of the BEV rows are empty and a few near-ego rows hold
// thousands of points).  The launch schedule is an array of descriptors, one per (XCD, slot): workgroup b runs on XCD b % 8
// (observed dispatch rule, used for locality only), and the plan orders the schedule so that one XCD works on tiles that gather from
// the same image columns.  The kernels that staged a tile's point records in LDS (k_pool_fwd_tiles, k_pool_fwd_lean,
// k_pool_fwd_lean2: rounds 2-4) were superseded by k_pool_fwd_direct below and live on as scripts/lab/patches/pool_superseded_kernels.patch.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 add4(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}


// x / d for a launch constant d (fp32 reciprocal with an exact fix-up)
__device__ __forceinline__ int div_const(int x, int d, float inv) {
  int q = (int)((float)x * inv);
  int r = x - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) { ++q; }
  return q;
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

extern "C" int omnihd_bev_pool_v2_fwd_csr(const float* depth, const float* feat, const int* ranks_depth, const int* ranks_feat,
                                          const int* row_ptr, float* out, int c, int n_rows, int n_points, void* stream) {
  OMNIHD_REQUIRE(c > 0 && n_rows >= 0 && n_points >= 0, "sizes");
  if (n_rows == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(depth && feat && row_ptr && out && (n_points == 0 || (ranks_depth && ranks_feat)), "null pointer");
  return launch_fwd<true>(depth, feat, ranks_depth, ranks_feat, nullptr, nullptr, row_ptr, out, c, n_rows, (hipStream_t)stream);
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
