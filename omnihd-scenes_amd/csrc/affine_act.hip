// Fused per-channel affine (+ residual) (+ ReLU) on channels-last bf16 activations, forward and backward.
//
// Where it sits: the image backbone of the reference config is a ResNet-50 whose BatchNorm layers are
// frozen (projects/configs/bevfusion_NewScenes/bevfusion.py:76-85: norm_cfg requires_grad=False,
// norm_eval=True), i.e. y = x * scale[c] + shift[c] with constants scale = gamma / sqrt(var + eps),
// shift = beta - mean * scale.  Under torch that is a BN-inference pass, an add pass (residual) and a ReLU
// pass forward, a ReLU-backward and a native_batch_norm_backward pass backward — the latter also emits
// NCHW gradients, forcing layout copies in front of every NHWC convolution kernel behind it.  Here the
// whole epilogue of a convolution is ONE pass each way, channels-last in and out:
//     forward : y = act(x * scale + shift [+ res])
//     backward: gres = gy * [y > 0],  gx = gres * scale          (scale/shift are constants: no gradient)
// HBM-bound streaming: 16 B per lane (8 channels), fp32 arithmetic, one rounding to bf16.
#include "common.h"

namespace omnihd {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {              // round to nearest even, NaN kept quiet
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(256) void k_affine_fwd(const u32x4* __restrict__ x, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, const u32x4* __restrict__ res,
                                                    u32x4* __restrict__ y, int64_t n_vec, int c8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % c8) * 8;
    const u32x4 xv = __builtin_nontemporal_load(x + i);
    u32x4 rv = {0u, 0u, 0u, 0u};
    if (RES) rv = __builtin_nontemporal_load(res + i);
    const float4 s0 = *reinterpret_cast<const float4*>(scale + c), s1 = *reinterpret_cast<const float4*>(scale + c + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(shift + c), b1 = *reinterpret_cast<const float4*>(shift + c + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float sh[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    const unsigned short* xe = reinterpret_cast<const unsigned short*>(&xv);
    const unsigned short* re = reinterpret_cast<const unsigned short*>(&rv);
    u32x4 out;
    unsigned short* oe = reinterpret_cast<unsigned short*>(&out);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = fmaf(bf2f(xe[k]), sc[k], sh[k]);
      if (RES) v += bf2f(re[k]);
      if (RELU) v = fmaxf(v, 0.f);
      oe[k] = f2bf(v);
    }
    y[i] = out;
  }
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(256) void k_affine_bwd(const u32x4* __restrict__ gy, const u32x4* __restrict__ y,
                                                    const float* __restrict__ scale, u32x4* __restrict__ gx,
                                                    u32x4* __restrict__ gres, int64_t n_vec, int c8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % c8) * 8;
    const u32x4 gv = __builtin_nontemporal_load(gy + i);
    u32x4 yv = {0u, 0u, 0u, 0u};
    if (RELU) yv = __builtin_nontemporal_load(y + i);
    const float4 s0 = *reinterpret_cast<const float4*>(scale + c), s1 = *reinterpret_cast<const float4*>(scale + c + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const unsigned short* ge = reinterpret_cast<const unsigned short*>(&gv);
    const unsigned short* ye = reinterpret_cast<const unsigned short*>(&yv);
    u32x4 o1, o2;
    unsigned short* e1 = reinterpret_cast<unsigned short*>(&o1);
    unsigned short* e2 = reinterpret_cast<unsigned short*>(&o2);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool pass = !RELU || bf2f(ye[k]) > 0.f;
      const unsigned short g = pass ? ge[k] : (unsigned short)0;
      e2[k] = g;
      e1[k] = f2bf(bf2f(g) * sc[k]);
    }
    gx[i] = o1;
    if (RES) gres[i] = o2;
  }
}

}  // namespace
}  // namespace omnihd

using namespace omnihd;

extern "C" int omnihd_affine_act_fwd(const void* x, const float* scale, const float* shift, const void* res, void* y,
                                     long long n_rows, int c, int relu, void* stream) {
  OMNIHD_REQUIRE(n_rows >= 0 && c > 0 && c % 8 == 0, "C must be a positive multiple of 8");
  if (n_rows == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(x && scale && shift && y, "null pointer");
  const int64_t n_vec = (int64_t)n_rows * (c / 8);
  const dim3 grid(grid_for(n_vec, 256 * 2)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const u32x4 *xv = (const u32x4*)x, *rv = (const u32x4*)res;
  if (relu && res) hipLaunchKernelGGL((k_affine_fwd<true, true>), grid, block, 0, st, xv, scale, shift, rv, (u32x4*)y, n_vec, c / 8);
  else if (relu) hipLaunchKernelGGL((k_affine_fwd<true, false>), grid, block, 0, st, xv, scale, shift, rv, (u32x4*)y, n_vec, c / 8);
  else if (res) hipLaunchKernelGGL((k_affine_fwd<false, true>), grid, block, 0, st, xv, scale, shift, rv, (u32x4*)y, n_vec, c / 8);
  else hipLaunchKernelGGL((k_affine_fwd<false, false>), grid, block, 0, st, xv, scale, shift, rv, (u32x4*)y, n_vec, c / 8);
  return check_launch("affine_act_fwd");
}

extern "C" int omnihd_affine_act_bwd(const void* gy, const void* y, const float* scale, void* gx, void* gres,
                                     long long n_rows, int c, int relu, void* stream) {
  OMNIHD_REQUIRE(n_rows >= 0 && c > 0 && c % 8 == 0, "C must be a positive multiple of 8");
  if (n_rows == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(gy && scale && gx && (!relu || y), "null pointer");
  const int64_t n_vec = (int64_t)n_rows * (c / 8);
  const dim3 grid(grid_for(n_vec, 256 * 2)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const u32x4 *gv = (const u32x4*)gy, *yv = (const u32x4*)y;
  if (relu && gres) hipLaunchKernelGGL((k_affine_bwd<true, true>), grid, block, 0, st, gv, yv, scale, (u32x4*)gx, (u32x4*)gres, n_vec, c / 8);
  else if (relu) hipLaunchKernelGGL((k_affine_bwd<true, false>), grid, block, 0, st, gv, yv, scale, (u32x4*)gx, (u32x4*)gres, n_vec, c / 8);
  else if (gres) hipLaunchKernelGGL((k_affine_bwd<false, true>), grid, block, 0, st, gv, yv, scale, (u32x4*)gx, (u32x4*)gres, n_vec, c / 8);
  else hipLaunchKernelGGL((k_affine_bwd<false, false>), grid, block, 0, st, gv, yv, scale, (u32x4*)gx, (u32x4*)gres, n_vec, c / 8);
  return check_launch("affine_act_bwd");
}
