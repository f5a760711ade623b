// Bilinear sampling of the deformable 3x3 convolution in DepthNet (deform_groups = 1) on gfx950.
//
// Reference: build_conv_layer(dict(type='DCN', groups=4, ...)) at
// bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:587-595; the op itself is mmcv's
// DeformConv2dPack (an un-vendored extension: deformable im2col + grouped GEMM, its backward scatters
// with atomicAdd).  Here the im2col is a ROW gather — the same access pattern as bev_pool: a feature
// row (C bf16 channels of one pixel, 512 B for C=256) is owned by C/8 lanes with 16 B each:
//
//   fwd      col[p][t][:]  = sum_{4 corners k} w_k(p,t) * x[corner_k(p,t)][:]        (p = output pixel, t = tap)
//   bwd-off  d off[p][t]   = chain rule over  D_k = <gcol[p][t][:], x[corner_k][:]>      (4 dot products)
//   bwd-in   gx[q][:]      = sum over the (p,t) whose sample lies within one pixel of q of
//                            (1-|py-qy|)(1-|px-qx|) * gcol[p][t][:]
//
// bwd-in is a GATHER over a bounded window of candidate output pixels (window radius from the largest
// |offset| of this call, read from device memory): deterministic, no atomics.  The contraction with
// the (grouped) weight stays a bf16 GEMM in the caller.  Zero padding outside the image, offset
// channel order (tap, (dy, dx)) as in mmcv.
#include "common.h"
#include "bn_vec.h"

namespace omnihd {
namespace {

constexpr int kBlock = 256;

// eight consecutive channels (vector i of a row-major array) as fp32: bf16 rows 16 B per lane, fp32 rows 32 B per lane;
// plain (cached) accesses — every input row is gathered by ~36 (pixel, tap, corner) samples
__device__ __forceinline__ void ld8(const bf16_t* __restrict__ p, size_t i, float (&f)[8]) {
  const uint4 v = reinterpret_cast<const uint4*>(p)[i];
  f[0] = bf2f(v.x & 0xffff); f[1] = bf2f(v.x >> 16); f[2] = bf2f(v.y & 0xffff); f[3] = bf2f(v.y >> 16);
  f[4] = bf2f(v.z & 0xffff); f[5] = bf2f(v.z >> 16); f[6] = bf2f(v.w & 0xffff); f[7] = bf2f(v.w >> 16);
}
__device__ __forceinline__ void ld8(const float* __restrict__ p, size_t i, float (&f)[8]) {
  const float4 a = reinterpret_cast<const float4*>(p)[2 * i], b = reinterpret_cast<const float4*>(p)[2 * i + 1];
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
__device__ __forceinline__ void st8(bf16_t* __restrict__ p, size_t i, const float (&f)[8]) {
  uint4 v;
  v.x = f2bf(f[0]) | ((unsigned)f2bf(f[1]) << 16); v.y = f2bf(f[2]) | ((unsigned)f2bf(f[3]) << 16);
  v.z = f2bf(f[4]) | ((unsigned)f2bf(f[5]) << 16); v.w = f2bf(f[6]) | ((unsigned)f2bf(f[7]) << 16);
  reinterpret_cast<uint4*>(p)[i] = v;
}
__device__ __forceinline__ void st8(float* __restrict__ p, size_t i, const float (&f)[8]) {
  reinterpret_cast<float4*>(p)[2 * i] = make_float4(f[0], f[1], f[2], f[3]);
  reinterpret_cast<float4*>(p)[2 * i + 1] = make_float4(f[4], f[5], f[6], f[7]);
}

struct Geo { int B, H, W, Ho, Wo, stride, pad, dil; };

struct Sample {        // bilinear footprint of one (output pixel, tap)
  bool any;
  int row[4];          // flat input pixel index of the 4 corners, -1 = outside
  float w[4];
  float hh, lh, hw, lw;
};

__device__ __forceinline__ Sample make_sample(const Geo& g, const float* __restrict__ off, long p, int t) {
  Sample s;
  const int hw_o = g.Ho * g.Wo;
  const int b = (int)(p / hw_o);
  const int rem = (int)(p % hw_o);
  const int yo = rem / g.Wo, xo = rem % g.Wo;
  const float py = (float)(yo * g.stride - g.pad + (t / 3) * g.dil) + off[p * 18 + 2 * t];
  const float px = (float)(xo * g.stride - g.pad + (t % 3) * g.dil) + off[p * 18 + 2 * t + 1];
  s.any = py > -1.f && px > -1.f && py < (float)g.H && px < (float)g.W;
  const float fy = floorf(py), fx = floorf(px);
  const int y0 = (int)fy, x0 = (int)fx;
  s.lh = py - fy; s.lw = px - fx; s.hh = 1.f - s.lh; s.hw = 1.f - s.lw;
  const float wy[2] = {s.hh, s.lh}, wx[2] = {s.hw, s.lw};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int yy = y0 + (k >> 1), xx = x0 + (k & 1);
    const bool ok = s.any && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
    s.row[k] = ok ? (b * g.H + yy) * g.W + xx : -1;
    s.w[k] = ok ? wy[k >> 1] * wx[k & 1] : 0.f;
  }
  return s;
}

template <typename T, int C8>
__global__ __launch_bounds__(kBlock) void k_dcn_fwd(const T* __restrict__ x, const float* __restrict__ off,
                                                    Geo g, long n_bags, T* __restrict__ col) {
  constexpr int G = kBlock / C8;
  const int sub = threadIdx.x % C8, grp = threadIdx.x / C8;
  for (long bag = (long)blockIdx.x * G + grp; bag < n_bags; bag += (long)gridDim.x * G) {
    const Sample s = make_sample(g, off, bag / 9, (int)(bag % 9));
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float f[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) ld8(x, (size_t)max(s.row[k], 0) * C8 + sub, f[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = fmaf(s.w[k], f[k][c], acc[c]);
    }
    st8(col, (size_t)bag * C8 + sub, acc);
  }
}

template <typename T, int C8>
__global__ __launch_bounds__(kBlock) void k_dcn_bwd_off(const T* __restrict__ x, const float* __restrict__ off,
                                                        const T* __restrict__ gcol, Geo g, long n_bags,
                                                        float* __restrict__ goff) {
  constexpr int G = kBlock / C8;
  const int sub = threadIdx.x % C8, grp = threadIdx.x / C8;
  for (long bag = (long)blockIdx.x * G + grp; bag < n_bags; bag += (long)gridDim.x * G) {
    const long p = bag / 9;
    const int t = (int)(bag % 9);
    const Sample s = make_sample(g, off, p, t);
    float gc[8];
    ld8(gcol, (size_t)bag * C8 + sub, gc);
    float d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float f[8];
      ld8(x, (size_t)max(s.row[k], 0) * C8 + sub, f);
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) a = fmaf(gc[c], f[c], a);
      d[k] = s.row[k] >= 0 ? a : 0.f;
    }
#pragma unroll
    for (int o = C8 / 2; o > 0; o >>= 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] += __shfl_xor(d[k], o, C8);
    }
    if (sub == 0) {
      // corners: 0 = (y0,x0) hh*hw, 1 = (y0,x1) hh*lw, 2 = (y1,x0) lh*hw, 3 = (y1,x1) lh*lw
      goff[p * 18 + 2 * t] = -s.hw * d[0] - s.lw * d[1] + s.hw * d[2] + s.lw * d[3];
      goff[p * 18 + 2 * t + 1] = -s.hh * d[0] + s.hh * d[1] - s.lh * d[2] + s.lh * d[3];
    }
  }
}

// The C8 lanes that own one input pixel search its candidate window TOGETHER: lane l tests candidates l, l + C8, ...
// (candidate = (tap, window position), in the order (tap, yo, xo) ascending), the hits of a round are collected with a
// ballot and then visited in ascending order by all lanes — four gradient rows in flight.  The summation order is the
// serial one, so the result does not depend on C8.
template <typename T, int C8>
__global__ __launch_bounds__(kBlock) void k_dcn_bwd_in(const float* __restrict__ off, const T* __restrict__ gcol,
                                                       Geo g, const int* __restrict__ radius_ptr, long n_in,
                                                       T* __restrict__ gx) {
  constexpr int G = kBlock / C8;
  const int sub = threadIdx.x % C8, grp = threadIdx.x / C8;
  const int R = *radius_ptr + 1;                      // candidate window half-width
  const int win = 2 * R + 1;
  const int n_cand = 9 * win * win;
  const int shift = (threadIdx.x & 63) / C8 * C8;     // first lane of this group inside its wavefront
  const unsigned long long gmask = (C8 == 64) ? ~0ull : ((1ull << C8) - 1ull);
  for (long q0 = (long)blockIdx.x * G; q0 < n_in; q0 += (long)gridDim.x * G) {
    const long q = q0 + grp;
    const bool live = q < n_in;                       // whole groups drop out together; ballots stay wave-wide
    const long qq = live ? q : 0;
    const int b = (int)(qq / (g.H * g.W));
    const int rem = (int)(qq % (g.H * g.W));
    const int y = rem / g.W, x = rem % g.W;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < n_cand; c0 += C8) {
      const int idx = c0 + sub;
      float w = 0.f;
      long row = 0;
      if (live && idx < n_cand) {
        const int t = idx / (win * win), r2 = idx - t * win * win;
        const int wy = r2 / win, wx = r2 - wy * win;
        const int ky = t / 3, kx = t - ky * 3;
        // stride 1: the undeformed sample of output pixel (yo, xo) sits at (yo - pad + ky*dil, ...)
        const int yo = y + g.pad - ky * g.dil - R + wy, xo = x + g.pad - kx * g.dil - R + wx;
        if (yo >= 0 && yo < g.Ho && xo >= 0 && xo < g.Wo) {
          const long p = ((long)b * g.Ho + yo) * g.Wo + xo;
          const float2 o = *reinterpret_cast<const float2*>(off + p * 18 + 2 * t);
          const float py = (float)(yo * g.stride - g.pad + ky * g.dil) + o.x;
          const float px = (float)(xo * g.stride - g.pad + kx * g.dil) + o.y;
          const float ay = fabsf(py - (float)y), ax = fabsf(px - (float)x);
          if (ay < 1.f && ax < 1.f) { w = (1.f - ay) * (1.f - ax); row = p * 9 + t; }
        }
      }
      // a hit with weight exactly 0 adds nothing in the serial order either
      unsigned long long m = (__ballot(w != 0.f) >> shift) & gmask;
      while (m) {
        int src[4];
        float ws[4];
        long rows[4];
        float f[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          src[u] = m ? __builtin_ctzll(m) : -1;
          if (m) m &= m - 1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int sl = src[u] < 0 ? 0 : src[u];
          ws[u] = __shfl(w, sl, C8);
          rows[u] = __shfl(row, sl, C8);
          if (src[u] >= 0) ld8(gcol, (size_t)rows[u] * C8 + sub, f[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (src[u] >= 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = fmaf(ws[u], f[u][c], acc[c]);
          }
        }
      }
    }
    if (live) st8(gx, (size_t)q * C8 + sub, acc);
  }
}

bool c8_ok(int c) { return c == 256 || c == 128 || c == 64 || c == 32; }

}  // namespace
}  // namespace omnihd

using namespace omnihd;

#define OMNIHD_DCN_DISPATCH(KERNEL, T, GRID, ...)                                                        \
  switch (c / 8) {                                                                                       \
    case 32: hipLaunchKernelGGL((KERNEL<T, 32>), GRID, dim3(kBlock), 0, st, __VA_ARGS__); break;        \
    case 16: hipLaunchKernelGGL((KERNEL<T, 16>), GRID, dim3(kBlock), 0, st, __VA_ARGS__); break;        \
    case 8: hipLaunchKernelGGL((KERNEL<T, 8>), GRID, dim3(kBlock), 0, st, __VA_ARGS__); break;          \
    default: hipLaunchKernelGGL((KERNEL<T, 4>), GRID, dim3(kBlock), 0, st, __VA_ARGS__); break;         \
  }

static int dcn_geo(Geo* g, int b, int h, int w, int stride, int pad, int dil) {
  g->B = b; g->H = h; g->W = w; g->stride = stride; g->pad = pad; g->dil = dil;
  g->Ho = (h + 2 * pad - dil * 2 - 1) / stride + 1;
  g->Wo = (w + 2 * pad - dil * 2 - 1) / stride + 1;
  return g->Ho > 0 && g->Wo > 0;
}

template <typename T>
static int dcn_fwd_t(const void* x_nhwc, const float* offset_nhwc, void* col, int batch, int h, int w, int c, int stride,
                     int pad, int dil, void* stream, const char* what) {
  hipStream_t st = (hipStream_t)stream;
  Geo g;
  OMNIHD_REQUIRE(batch > 0 && c8_ok(c) && dcn_geo(&g, batch, h, w, stride, pad, dil), "shape (C in {32,64,128,256})");
  OMNIHD_REQUIRE(x_nhwc && offset_nhwc && col, "null pointer");
  const long n_bags = (long)batch * g.Ho * g.Wo * 9;
  const dim3 grid(grid_for(n_bags, kBlock / (c / 8) * 4));
  OMNIHD_DCN_DISPATCH(k_dcn_fwd, T, grid, static_cast<const T*>(x_nhwc), offset_nhwc, g, n_bags, static_cast<T*>(col))
  return check_launch(what);
}

template <typename T>
static int dcn_bwd_t(const void* x_nhwc, const float* offset_nhwc, const void* gcol, const int* max_abs_offset_ceil,
                     void* gx_nhwc, float* goffset_nhwc, int batch, int h, int w, int c, int stride, int pad, int dil,
                     void* stream, const char* what) {
  hipStream_t st = (hipStream_t)stream;
  Geo g;
  OMNIHD_REQUIRE(batch > 0 && c8_ok(c) && dcn_geo(&g, batch, h, w, stride, pad, dil), "shape (C in {32,64,128,256})");
  OMNIHD_REQUIRE(stride == 1, "the input-gradient gather assumes stride 1");
  OMNIHD_REQUIRE(x_nhwc && offset_nhwc && gcol && max_abs_offset_ceil, "null pointer");
  const long n_bags = (long)batch * g.Ho * g.Wo * 9;
  const long n_in = (long)batch * h * w;
  if (goffset_nhwc) {
    const dim3 grid(grid_for(n_bags, kBlock / (c / 8) * 4));
    OMNIHD_DCN_DISPATCH(k_dcn_bwd_off, T, grid, static_cast<const T*>(x_nhwc), offset_nhwc, static_cast<const T*>(gcol), g,
                        n_bags, goffset_nhwc)
  }
  if (gx_nhwc) {
    const dim3 grid(grid_for(n_in, kBlock / (c / 8)));
    OMNIHD_DCN_DISPATCH(k_dcn_bwd_in, T, grid, offset_nhwc, static_cast<const T*>(gcol), g, max_abs_offset_ceil, n_in,
                        static_cast<T*>(gx_nhwc))
  }
  return check_launch(what);
}

extern "C" int omnihd_dcn3x3_sample_fwd(const void* x_nhwc_bf16, const float* offset_nhwc, void* col_bf16, int batch,
                                        int h, int w, int c, int stride, int pad, int dil, void* stream) {
  return dcn_fwd_t<bf16_t>(x_nhwc_bf16, offset_nhwc, col_bf16, batch, h, w, c, stride, pad, dil, stream, "dcn3x3_sample_fwd");
}
extern "C" int omnihd_dcn3x3_sample_fwd_f32(const float* x_nhwc, const float* offset_nhwc, float* col, int batch, int h,
                                            int w, int c, int stride, int pad, int dil, void* stream) {
  return dcn_fwd_t<float>(x_nhwc, offset_nhwc, col, batch, h, w, c, stride, pad, dil, stream, "dcn3x3_sample_fwd_f32");
}

extern "C" int omnihd_dcn3x3_sample_bwd(const void* x_nhwc_bf16, const float* offset_nhwc, const void* gcol_bf16,
                                        const int* max_abs_offset_ceil, void* gx_nhwc_bf16, float* goffset_nhwc,
                                        int batch, int h, int w, int c, int stride, int pad, int dil, void* stream) {
  return dcn_bwd_t<bf16_t>(x_nhwc_bf16, offset_nhwc, gcol_bf16, max_abs_offset_ceil, gx_nhwc_bf16, goffset_nhwc, batch, h, w,
                           c, stride, pad, dil, stream, "dcn3x3_sample_bwd");
}
extern "C" int omnihd_dcn3x3_sample_bwd_f32(const float* x_nhwc, const float* offset_nhwc, const float* gcol,
                                            const int* max_abs_offset_ceil, float* gx_nhwc, float* goffset_nhwc, int batch,
                                            int h, int w, int c, int stride, int pad, int dil, void* stream) {
  return dcn_bwd_t<float>(x_nhwc, offset_nhwc, gcol, max_abs_offset_ceil, gx_nhwc, goffset_nhwc, batch, h, w, c, stride, pad,
                          dil, stream, "dcn3x3_sample_bwd_f32");
}
