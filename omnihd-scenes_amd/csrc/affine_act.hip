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
// HBM-bound streaming: 8 channels per lane (16 B of bf16 or 32 B of fp32), fp32 arithmetic, one rounding to bf16.
// The element type is a template parameter (bn_vec.h): bf16 rows under autocast, fp32 rows for the reference-precision
// step — the *_f32 entry points.
#include "common.h"
#include "bn_vec.h"

namespace omnihd {
namespace {

template <typename T, bool RELU, bool RES>
__global__ __launch_bounds__(256) void k_affine_fwd(const T* __restrict__ x, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, const T* __restrict__ res,
                                                    T* __restrict__ y, int64_t n_vec, int c8, bf16_t* __restrict__ y_hi = nullptr,
                                                    bf16_t* __restrict__ y_lo = nullptr) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % c8) * 8;
    float xv[8], rv[8], out[8];
    load8(x, i, xv);
    if (RES) load8(res, i, rv);
    const float4 s0 = *reinterpret_cast<const float4*>(scale + c), s1 = *reinterpret_cast<const float4*>(scale + c + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(shift + c), b1 = *reinterpret_cast<const float4*>(shift + c + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float sh[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = fmaf(xv[k], sc[k], sh[k]);
      if (RES) v += rv[k];
      if (RELU) v = fmaxf(v, 0.f);
      out[k] = v;
    }
    store8(y, i, out);
    if (y_hi) {                                      // the next convolution reads these instead of a split / cast pass over y:
      if (y_lo) store_planes8(y_hi, y_lo, i, out);   //   the two bf16 planes of the fp32-grade form,
      else store_half8(y_hi, i, out);                //   or (y_lo NULL) the IEEE-half plane of the TF32-grade form
    }
  }
}

template <typename T, bool RELU, bool RES, bool AMAX = false>
__global__ __launch_bounds__(256) void k_affine_bwd(const T* __restrict__ gy, const T* __restrict__ y,
                                                    const float* __restrict__ scale, T* __restrict__ gx,
                                                    T* __restrict__ gres, int64_t n_vec, int c8, unsigned* __restrict__ amax_bits = nullptr) {
  float amax = 0.f;                                // max |gx| for the TF32-grade convolution in front of this layer (its cast needs the scale)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % c8) * 8;
    float gv[8], yv[8], o1[8], o2[8];
    load8(gy, i, gv);
    if (RELU) load8(y, i, yv);
    const float4 s0 = *reinterpret_cast<const float4*>(scale + c), s1 = *reinterpret_cast<const float4*>(scale + c + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool pass = !RELU || yv[k] > 0.f;
      const float g = pass ? gv[k] : 0.f;
      o2[k] = g;
      o1[k] = g * sc[k];
      if (AMAX) amax = fmaxf(amax, fabsf(o1[k]));
    }
    store8(gx, i, o1);
    if (RES) store8(gres, i, o2);
  }
  if (AMAX) block_amax_256(amax, amax_bits);
}

template <typename T>
int affine_fwd_t(const void* x, const float* scale, const float* shift, const void* res, void* y, long long n_rows, int c,
                 int relu, void* stream, bf16_t* y_hi = nullptr, bf16_t* y_lo = nullptr) {
  const int64_t n_vec = (int64_t)n_rows * (c / 8);
  const dim3 grid(grid_for(n_vec, 256 * 2)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const T *xv = (const T*)x, *rv = (const T*)res;
  if (relu && res) hipLaunchKernelGGL((k_affine_fwd<T, true, true>), grid, block, 0, st, xv, scale, shift, rv, (T*)y, n_vec, c / 8, y_hi, y_lo);
  else if (relu) hipLaunchKernelGGL((k_affine_fwd<T, true, false>), grid, block, 0, st, xv, scale, shift, rv, (T*)y, n_vec, c / 8, y_hi, y_lo);
  else if (res) hipLaunchKernelGGL((k_affine_fwd<T, false, true>), grid, block, 0, st, xv, scale, shift, rv, (T*)y, n_vec, c / 8, y_hi, y_lo);
  else hipLaunchKernelGGL((k_affine_fwd<T, false, false>), grid, block, 0, st, xv, scale, shift, rv, (T*)y, n_vec, c / 8, y_hi, y_lo);
  return check_launch("affine_act_fwd");
}

template <typename T>
int affine_bwd_t(const void* gy, const void* y, const float* scale, void* gx, void* gres, long long n_rows, int c, int relu,
                 void* stream, unsigned* amax_bits = nullptr) {
  const int64_t n_vec = (int64_t)n_rows * (c / 8);
  const dim3 grid(grid_for(n_vec, 256 * 2)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const T *gv = (const T*)gy, *yv = (const T*)y;
#define OMNIHD_AFFINE_BWD(R, G, A)                                                                                          \
  hipLaunchKernelGGL((k_affine_bwd<T, R, G, A>), grid, block, 0, st, gv, yv, scale, (T*)gx, (T*)gres, n_vec, c / 8, amax_bits)
  if (amax_bits) {
    if constexpr (sizeof(T) == 4) {
      if (relu && gres) OMNIHD_AFFINE_BWD(true, true, true);
      else if (relu) OMNIHD_AFFINE_BWD(true, false, true);
      else if (gres) OMNIHD_AFFINE_BWD(false, true, true);
      else OMNIHD_AFFINE_BWD(false, false, true);
    } else {
      OMNIHD_REQUIRE(false, "the amax form takes fp32 rows");
    }
  } else if (relu && gres) OMNIHD_AFFINE_BWD(true, true, false);
  else if (relu) OMNIHD_AFFINE_BWD(true, false, false);
  else if (gres) OMNIHD_AFFINE_BWD(false, true, false);
  else OMNIHD_AFFINE_BWD(false, false, false);
#undef OMNIHD_AFFINE_BWD
  return check_launch("affine_act_bwd");
}

}  // namespace
}  // namespace omnihd

using namespace omnihd;

extern "C" int omnihd_affine_act_fwd(const void* x, const float* scale, const float* shift, const void* res, void* y,
                                     long long n_rows, int c, int relu, void* stream) {
  OMNIHD_REQUIRE(n_rows >= 0 && c > 0 && c % 8 == 0, "C must be a positive multiple of 8");
  if (n_rows == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(x && scale && shift && y, "null pointer");
  return affine_fwd_t<bf16_t>(x, scale, shift, res, y, n_rows, c, relu, stream);
}

extern "C" int omnihd_affine_act_fwd_f32(const float* x, const float* scale, const float* shift, const float* res, float* y,
                                         long long n_rows, int c, int relu, void* stream) {
  OMNIHD_REQUIRE(n_rows >= 0 && c > 0 && c % 8 == 0, "C must be a positive multiple of 8");
  if (n_rows == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(x && scale && shift && y, "null pointer");
  return affine_fwd_t<float>(x, scale, shift, res, y, n_rows, c, relu, stream);
}

/* omnihd_affine_act_fwd_f32 that ALSO writes the two bf16 planes of y (omnihd_split_f32's output) for the next split convolution */
extern "C" int omnihd_affine_act_fwd_f32_planes(const float* x, const float* scale, const float* shift, const float* res, float* y,
                                                void* y_hi, void* y_lo, long long n_rows, int c, int relu, void* stream) {
  OMNIHD_REQUIRE(n_rows >= 0 && c > 0 && c % 8 == 0, "C must be a positive multiple of 8");
  if (n_rows == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(x && scale && shift && y && y_hi, "null pointer");          // (y_lo NULL: y_hi receives the IEEE-half plane)
  OMNIHD_REQUIRE(((reinterpret_cast<uintptr_t>(y_hi) | reinterpret_cast<uintptr_t>(y_lo)) & 15u) == 0, "16-byte alignment");
  return affine_fwd_t<float>(x, scale, shift, res, y, n_rows, c, relu, stream, static_cast<bf16_t*>(y_hi), static_cast<bf16_t*>(y_lo));
}

/* omnihd_affine_act_bwd_f32 that ALSO accumulates max |gx| into *amax_bits (bit pattern of a float >= 0; the caller zeroes it): the
 * scale of the half cast of gx for the TF32-grade convolution in front of this layer (omnihd_cast_f16, scaled == 3) */
extern "C" int omnihd_affine_act_bwd_f32_amax(const float* gy, const float* y, const float* scale, float* gx, float* gres,
                                              void* amax_bits, long long n_rows, int c, int relu, void* stream) {
  OMNIHD_REQUIRE(n_rows >= 0 && c > 0 && c % 8 == 0, "C must be a positive multiple of 8");
  if (n_rows == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(gy && scale && gx && (!relu || y) && amax_bits, "null pointer");
  return affine_bwd_t<float>(gy, y, scale, gx, gres, n_rows, c, relu, stream, static_cast<unsigned*>(amax_bits));
}

extern "C" int omnihd_affine_act_bwd(const void* gy, const void* y, const float* scale, void* gx, void* gres,
                                     long long n_rows, int c, int relu, void* stream) {
  OMNIHD_REQUIRE(n_rows >= 0 && c > 0 && c % 8 == 0, "C must be a positive multiple of 8");
  if (n_rows == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(gy && scale && gx && (!relu || y), "null pointer");
  return affine_bwd_t<bf16_t>(gy, y, scale, gx, gres, n_rows, c, relu, stream);
}

extern "C" int omnihd_affine_act_bwd_f32(const float* gy, const float* y, const float* scale, float* gx, float* gres,
                                         long long n_rows, int c, int relu, void* stream) {
  OMNIHD_REQUIRE(n_rows >= 0 && c > 0 && c % 8 == 0, "C must be a positive multiple of 8");
  if (n_rows == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(gy && scale && gx && (!relu || y), "null pointer");
  return affine_bwd_t<float>(gy, y, scale, gx, gres, n_rows, c, relu, stream);
}
