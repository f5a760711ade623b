// Eight consecutive channels of one channels-last row as fp32 values, for bf16 rows (16 B per lane) and fp32 rows
// (two 16-B accesses per lane).  Streaming accesses: non-temporal loads, plain stores.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace omnihd {

typedef unsigned short bf16_t;                       // storage type of a bf16 element
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {              // round to nearest even, NaN kept quiet
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// vector i = elements [8*i, 8*i + 8) of the array
__device__ __forceinline__ void load8(const bf16_t* __restrict__ p, int64_t i, float (&v)[8]) {
  const u32x4 raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p) + i);
  const unsigned short* e = reinterpret_cast<const unsigned short*>(&raw);
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = bf2f(e[k]);
}
__device__ __forceinline__ void load8(const float* __restrict__ p, int64_t i, float (&v)[8]) {
  const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + 2 * i);
  const f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + 2 * i + 1);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(bf16_t* __restrict__ p, int64_t i, const float (&v)[8]) {
  u32x4 raw;
  unsigned short* e = reinterpret_cast<unsigned short*>(&raw);
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] = f2bf(v[k]);
  reinterpret_cast<u32x4*>(p)[i] = raw;
}
__device__ __forceinline__ void store8(float* __restrict__ p, int64_t i, const float (&v)[8]) {
  reinterpret_cast<f32x4*>(p)[2 * i] = f32x4{v[0], v[1], v[2], v[3]};
  reinterpret_cast<f32x4*>(p)[2 * i + 1] = f32x4{v[4], v[5], v[6], v[7]};
}

// the two bf16 planes of 8 fp32 values (hi = bf16(v), lo = bf16(v - hi); inf / nan stay in the hi plane alone): what
// omnihd_split_f32 writes, here from the registers of the kernel that produced v
__device__ __forceinline__ void store_planes8(bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int64_t i, const float (&v)[8]) {
  u32x4 rh, rl;
  unsigned short* eh = reinterpret_cast<unsigned short*>(&rh);
  unsigned short* el = reinterpret_cast<unsigned short*>(&rl);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    eh[k] = f2bf(v[k]);
    const bool finite = (eh[k] & 0x7f80) != 0x7f80;
    el[k] = finite ? f2bf(v[k] - bf2f(eh[k])) : (unsigned short)0;
  }
  reinterpret_cast<u32x4*>(hi)[i] = rh;
  reinterpret_cast<u32x4*>(lo)[i] = rl;
}

// the IEEE-half plane of 8 fp32 values (round to nearest even; finite values beyond +-65504 saturate, infinities / NaNs pass): what
// omnihd_cast_f16 writes without a scale — the operand of the TF32-grade convolutions (conv_igemm.hip, F16), from the producer's registers
__device__ __forceinline__ void store_half8(bf16_t* __restrict__ h, int64_t i, const float (&v)[8]) {
  u32x4 raw;
  unsigned short* e = reinterpret_cast<unsigned short*>(&raw);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float a = fabsf(v[k]);
    const float s = (a > 65504.f && a < __builtin_huge_valf()) ? copysignf(65504.f, v[k]) : v[k];
    e[k] = __builtin_bit_cast(unsigned short, (_Float16)s);
  }
  reinterpret_cast<u32x4*>(h)[i] = raw;
}

// max |v| of a 256-thread workgroup into *amax_bits (bit pattern of a non-negative float; atomicMax is order-independent): ONE atomic
// per workgroup, and only where it can still raise the value.  Every thread of the workgroup calls it, once, after its loop.
__device__ __forceinline__ void block_amax_256(float m, unsigned* __restrict__ amax_bits) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float wave_max[4];
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
    const unsigned bits = __float_as_uint(m);
    if (bits > __hip_atomic_load(amax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax_bits, bits);
  }
}

}  // namespace omnihd
