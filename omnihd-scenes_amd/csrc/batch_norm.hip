// Training-mode BatchNorm (+ ReLU) on channels-last bf16 activations, split so that the per-channel
// statistics can be exchanged between ranks in the middle ("naive" SyncBN of the reference:
// projects/mmdet3d_plugin/ops/norm.py:28-82 — mean and mean-of-squares are AVERAGED OVER RANKS, each rank
// weighted 1/R whatever its pixel count; the 1-D/2-D variants come from mmdet3d, un-vendored).
//
// Under torch this is, per layer, MIOpen's 3 forward + 3 backward kernels plus separate ReLU passes on one
// GPU, and on several GPUs the reference algorithm spelled as a dozen fp32 elementwise/reduction passes
// (x.float(), mean, x*x, mean, x*scale+bias, cast and their autograd mirrors).  Here, on any number of GPUs:
//   forward : k_bn_partial (sum x, sum x^2 per channel, one read of x)  -> k_bn_reduce -> [all-reduce]
//             -> k_bn_fwd_consts (scale, shift, running stats) -> affine_act_fwd (one read, one write, ReLU fused)
//   backward: k_bn_partial (sum g', sum g'x with the ReLU mask; reads gy, y, x) -> k_bn_reduce -> [all-reduce]
//             -> k_bn_bwd_consts (dgamma, dbeta and the three per-channel coefficients)
//             -> k_bn_bwd_apply  gx = g' * A[c] + x * B[c] + C[c]   (one pass)
// HBM-bound streaming; reductions are two-stage with a fixed order (deterministic, no atomics).
#include "common.h"
#include "bn_vec.h"

namespace omnihd {
namespace {

// MODE 0: s1 = sum x,  s2 = sum x*x                       (a = x)
// MODE 1: s1 = sum g', s2 = sum g'*x,  g' = gy * [y > 0]  (a = gy, b = x, m = y or nullptr)
// partial[block][0][c] = s1, partial[block][1][c] = s2 over the block's rows.  T: bf16 or fp32 rows.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_bn_partial(const T* __restrict__ a, const T* __restrict__ b,
                                                    const T* __restrict__ m, const float* __restrict__ fss,
                                                    float* __restrict__ partial, long long rows, int c8,
                                                    long long rows_per_block) {
  __shared__ float red[2][256][8];
  const int t = threadIdx.x;
  const int rpi = 256 / c8;                      // rows handled per iteration by the workgroup
  const int vcol = t % c8, r0 = t / c8;
  float s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s1[k] = s2[k] = 0.f;
  const long long begin = (long long)blockIdx.x * rows_per_block;
  const long long end = min(begin + rows_per_block, rows);
  if (r0 < rpi) {
    constexpr int U = (sizeof(T) == 2) ? 4 : 2;   // independent row loads in flight per lane (16 B resp. 32 B each)
    for (long long rb = begin + r0; rb < end; rb += (long long)U * rpi) {
      float av[U][8], bv[U][8], mv[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long r = rb + (long long)u * rpi;
        const bool ok = r < end;
        const long long i = (ok ? r : begin + r0) * c8 + vcol;
        load8(a, i, av[u]);
        if (MODE == 1) {
          load8(b, i, bv[u]);
          if (m) load8(m, i, mv[u]);
          else {
#pragma unroll
            for (int k = 0; k < 8; ++k) mv[u][k] = 1.f;                                   // mask passes
          }
        }
        if (!ok) {
#pragma unroll
          for (int k = 0; k < 8; ++k) av[u][k] = 0.f;                                     // a zero row adds nothing to either sum
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (MODE == 0) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float v = av[u][k];
            s1[k] += v;
            s2[k] = fmaf(v, v, s2[k]);
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float xv = bv[u][k];
            // ReLU mask: from the saved output, or recomputed from x with the forward's own fmaf (no y to read)
            const bool pass = (m || !fss) ? mv[u][k] > 0.f : fmaf(xv, fss[vcol * 8 + k], fss[c8 * 8 + vcol * 8 + k]) > 0.f;
            const float g = pass ? av[u][k] : 0.f;
            s1[k] += g;
            s2[k] = fmaf(g, xv, s2[k]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    red[0][t][k] = s1[k];
    red[1][t][k] = s2[k];
  }
  __syncthreads();
  if (t < c8) {
    float o1[8], o2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o1[k] = o2[k] = 0.f;
    for (int g = 0; g < rpi; ++g)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        o1[k] += red[0][g * c8 + t][k];
        o2[k] += red[1][g * c8 + t][k];
      }
    float* dst = partial + (size_t)blockIdx.x * 2 * c8 * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      dst[t * 8 + k] = o1[k];
      dst[c8 * 8 + t * 8 + k] = o2[k];
    }
  }
}

// sums[0][c], sums[1][c] = (sum over blocks of partial) * mult.  One wavefront per output: the 64 lanes stride
// over the blocks (a handful of independent loads each) and combine with a fixed shuffle tree.
__global__ __launch_bounds__(256) void k_bn_reduce(const float* __restrict__ partial, int n_blocks, int c, float mult,
                                                   float* __restrict__ sums) {
  const int out = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (out >= 2 * c) return;
  float s = 0.f;
  for (int b = lane; b < n_blocks; b += 64) s += partial[(size_t)b * 2 * c + out];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) sums[out] = s * mult;
}

// Sum of the per-block partials of one output column, in block order (the same order for every launch: deterministic).
// 16 lanes per column: lane l adds blocks l, l+16, ... and a fixed xor tree combines the 16 strands — every lane of the
// group returns the total.
__device__ __forceinline__ float sum_partials(const float* __restrict__ partial, int n_blocks, int c2, int col, int l16) {
  float s = 0.f;
#pragma unroll 4
  for (int b = l16; b < n_blocks; b += 16) s += partial[(size_t)b * c2 + col];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
  return s;
}

// stats = (mean, meansqr) [2c] possibly summed over ranks -> * rank_mult; constants of the forward pass.
// With n_blocks > 0 `stats` is the PARTIAL array of k_bn_partial ([n_blocks][2c]) and the second reduction stage happens
// here (single-rank path: one launch less per layer and direction; the reduced statistics are also written to stats_out).
__global__ __launch_bounds__(256) void k_bn_fwd_consts(const float* __restrict__ stats, float rank_mult, int n_blocks,
                                                       float* __restrict__ stats_out,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, float momentum, float var_correction, int c,
                                                       float* __restrict__ running_mean, float* __restrict__ running_var,
                                                       float* __restrict__ scale, float* __restrict__ shift,
                                                       float* __restrict__ mean_out, float* __restrict__ invstd_out) {
  // launch: n_blocks > 0 -> 16 channels per workgroup (16 lanes each), else 256 channels per workgroup
  const int i = n_blocks > 0 ? blockIdx.x * 16 + threadIdx.x / 16 : blockIdx.x * 256 + threadIdx.x;
  const int l16 = threadIdx.x % 16;
  if (i >= c) return;
  float mean, msq;
  if (n_blocks > 0) {
    mean = sum_partials(stats, n_blocks, 2 * c, i, l16) * rank_mult;
    msq = sum_partials(stats, n_blocks, 2 * c, c + i, l16) * rank_mult;
    if (l16 != 0) return;
    if (stats_out) { stats_out[i] = mean; stats_out[c + i] = msq; }
  } else {
    mean = stats[i] * rank_mult;
    msq = stats[c + i] * rank_mult;
  }
  const float var = fmaxf(msq - mean * mean, 0.f);          // E[x^2] - E[x]^2 can cancel below zero for |mean| >> std
  const float invstd = rsqrtf(var + eps);
  const float sc = gamma[i] * invstd;
  scale[i] = sc;
  shift[i] = beta[i] - mean * sc;
  mean_out[i] = mean;
  invstd_out[i] = invstd;
  if (running_mean) {
    running_mean[i] += momentum * (mean - running_mean[i]);
    running_var[i] += momentum * (var * var_correction - running_var[i]);
  }
}

// local [2c] = this rank's (sum g', sum g'x); global [2c] = the same summed over ranks.
// dgamma, dbeta from the local sums; coefficients of gx = g'*A + x*B + C from the global ones.
__global__ __launch_bounds__(256) void k_bn_bwd_consts(const float* __restrict__ local, const float* __restrict__ global,
                                                       int n_blocks, const float* __restrict__ gamma,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, float inv_count, int c,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                       float* __restrict__ coefA, float* __restrict__ coefB,
                                                       float* __restrict__ coefC) {
  const int i = n_blocks > 0 ? blockIdx.x * 16 + threadIdx.x / 16 : blockIdx.x * 256 + threadIdx.x;
  const int l16 = threadIdx.x % 16;
  if (i >= c) return;
  const float mu = mean[i], is = invstd[i], g = gamma[i];
  float l1, l2, t1, t2;
  if (n_blocks > 0) {                                          // single rank: local == global == the partials, reduced here
    l1 = t1 = sum_partials(local, n_blocks, 2 * c, i, l16);
    l2 = t2 = sum_partials(local, n_blocks, 2 * c, c + i, l16);
    if (l16 != 0) return;
  } else {
    l1 = local[i]; l2 = local[c + i];
    t1 = global[i]; t2 = global[c + i];
  }
  dbeta[i] = l1;
  dgamma[i] = is * (l2 - mu * l1);
  const float dot = g * (t2 - mu * t1) * is * is * is;      // gamma * sum g'(x - mu) * invstd^3
  const float dmu = -g * is * t1 + dot * mu;                 // dL/dmu
  const float dq = -0.5f * dot;                              // dL/d(meansqr)
  coefA[i] = g * is;
  coefB[i] = 2.f * dq * inv_count;
  coefC[i] = dmu * inv_count;
}

// MASK 0: no ReLU in front of the gradient; 1: mask from the saved output y; 2: mask recomputed from x with the forward's
// scale / shift (fss).  The per-channel coefficients are read as two 16-byte vectors per array (round 3's form read them one
// float at a time inside a three-way conditional: 24 dependent 4-byte loads per 8 elements, 2.4 TB/s; this form 4.8 TB/s on
// a 157 MB tensor, scripts/micro/bn_apply_bw.hip).
template <typename T, int MASK, bool AMAX = false>      // (AMAX: its own instantiation — two more registers would cost the plain form a wave of occupancy)
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const T* __restrict__ gy, const T* __restrict__ y,
                                                      const T* __restrict__ x, const float* __restrict__ coefA,
                                                      const float* __restrict__ coefB, const float* __restrict__ coefC,
                                                      const float* __restrict__ fss, T* __restrict__ gx,
                                                      T* __restrict__ gres, int64_t n_vec, int c8, bf16_t* __restrict__ gx_hi = nullptr,
                                                      bf16_t* __restrict__ gx_lo = nullptr, unsigned* __restrict__ amax_bits = nullptr) {
  float amax = 0.f;                                // max |gx| (amax_bits given: the TF32-grade convolution in front casts gx with that scale)
  auto vec8 = [](const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  };
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * 256) {
    const int c = (int)((uint64_t)i % (unsigned)c8) * 8;
    float gv[8], xv[8], mv[8], out[8], masked[8], ca[8], cb[8], cc[8], fs[8], fh[8];
    load8(gy, i, gv);
    load8(x, i, xv);
    if (MASK == 1) load8(y, i, mv);
    vec8(coefA + c, ca);
    vec8(coefB + c, cb);
    vec8(coefC + c, cc);
    if (MASK == 2) {
      vec8(fss + c, fs);
      vec8(fss + c8 * 8 + c, fh);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool pass = MASK == 0 ? true : MASK == 1 ? mv[k] > 0.f : fmaf(xv[k], fs[k], fh[k]) > 0.f;
      const float g = pass ? gv[k] : 0.f;
      masked[k] = g;
      out[k] = fmaf(g, ca[k], fmaf(xv[k], cb[k], cc[k]));
      if (AMAX) amax = fmaxf(amax, fabsf(out[k]));
    }
    if (gx) store8(gx, i, out);                      // (gx NULL: the convolution behind this layer takes the planes ONLY)
    if (gx_hi) store_planes8(gx_hi, gx_lo, i, out);  // the convolution behind this layer reads these instead of a split pass over gx
    if (gres) store8(gres, i, masked);             // gradient of a residual added before the ReLU
  }
  if (AMAX) block_amax_256(amax, amax_bits);
}

int plan_blocks(long long rows, int c8, long long* rows_per_block) {
  const int rpi = 256 / c8;
  long long iters = (rows + rpi - 1) / rpi;
  long long blocks = (iters + 15) / 16;                     // >= 16 row-iterations per workgroup
  static const long long cap = [] {                          // the second stage reads blocks x 2c floats
    const char* e = getenv("OMNIHD_BN_MAX_BLOCKS");         // (lab override, scripts/lab/bn_parts_time.py)
    const long long v = e ? atoll(e) : 0;
    return v >= 1 && v <= 8192 ? v : 512ll;
  }();
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  long long per = (rows + blocks - 1) / blocks;
  per = (per + rpi - 1) / rpi * rpi;
  *rows_per_block = per;
  return (int)((rows + per - 1) / per);
}

}  // namespace
}  // namespace omnihd

using namespace omnihd;

// ---------------------------------------------------------------------------------------------------------------
// Column sums of a row-major [rows][c] matrix for ANY c (scalar element loads, lanes along the columns): the bias gradient
// of a convolution whose channel count is not a multiple of 8.  torch's reduction of an NHWC gradient with an odd channel
// count (DepthNet's 59 depth logits) takes 0.36 ms for 8 MB.  Two stages in a fixed order, like the vector kernels above.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float elem_f32(const T* p, size_t i);
template <>
__device__ __forceinline__ float elem_f32<bf16_t>(const bf16_t* p, size_t i) { return bf2f(p[i]); }
template <>
__device__ __forceinline__ float elem_f32<float>(const float* p, size_t i) { return p[i]; }

template <typename T>
__global__ __launch_bounds__(256) void k_col_partial(const T* __restrict__ a, float* __restrict__ partial, long long rows,
                                                     int c, int cw, long long rows_per_block) {
  __shared__ float red[256];
  const int t = threadIdx.x;
  const int rl_n = 256 / cw;                        // row lanes of the workgroup
  const int cl = t % cw, rl = t / cw;
  const long long begin = (long long)blockIdx.x * rows_per_block;
  const long long end = min(begin + rows_per_block, rows);
  for (int c0 = 0; c0 < c; c0 += cw) {              // one pass per strip of cw columns (one pass for c <= 128)
    const int col = c0 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (col < c && rl < rl_n) {
      long long r = begin + rl;
      for (; r + 3 * rl_n < end; r += 4 * rl_n) {
        s0 += elem_f32<T>(a, (size_t)r * c + col);
        s1 += elem_f32<T>(a, (size_t)(r + rl_n) * c + col);
        s2 += elem_f32<T>(a, (size_t)(r + 2 * rl_n) * c + col);
        s3 += elem_f32<T>(a, (size_t)(r + 3 * rl_n) * c + col);
      }
      for (; r < end; r += rl_n) s0 += elem_f32<T>(a, (size_t)r * c + col);
    }
    red[t] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && col < c) {
      float s = red[cl];
      for (int k = 1; k < rl_n; ++k) s += red[k * cw + cl];
      partial[(size_t)blockIdx.x * c + col] = s;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_col_reduce(const float* __restrict__ partial, int n_blocks, int c, float* __restrict__ sums) {
  const int out = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (out >= c) return;
  float s = 0.f;
  for (int b = lane; b < n_blocks; b += 64) s += partial[(size_t)b * c + out];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) sums[out] = s;
}

constexpr int kColBlocks = 512;

extern "C" size_t omnihd_column_sums_workspace_bytes(long long rows, int c) {
  (void)rows;
  return c > 0 ? (size_t)kColBlocks * c * sizeof(float) : 0;
}

extern "C" int omnihd_column_sums(const void* a, int is_f32, long long rows, int c, float* sums, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  OMNIHD_REQUIRE(rows > 0 && c > 0, "sizes");
  OMNIHD_REQUIRE(a && sums && workspace, "null pointer");
  OMNIHD_REQUIRE(workspace_bytes >= omnihd_column_sums_workspace_bytes(rows, c), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  int cw = 32;
  while (cw < c && cw < 256) cw <<= 1;              // lanes along the columns: 32, 64, 128 or 256
  long long per = (rows + kColBlocks - 1) / kColBlocks;
  const int rl_n = 256 / cw;
  if (per < 4 * rl_n) per = 4 * rl_n;               // at least one unrolled trip per row lane
  const int blocks = (int)((rows + per - 1) / per);
  float* partial = static_cast<float*>(workspace);
  if (is_f32) {
    hipLaunchKernelGGL((k_col_partial<float>), dim3(blocks), dim3(256), 0, st, static_cast<const float*>(a), partial, rows, c, cw,
                       per);
  } else {
    hipLaunchKernelGGL((k_col_partial<bf16_t>), dim3(blocks), dim3(256), 0, st, static_cast<const bf16_t*>(a), partial, rows, c,
                       cw, per);
  }
  hipLaunchKernelGGL(k_col_reduce, dim3((c + 3) / 4), dim3(256), 0, st, partial, blocks, c, sums);
  return check_launch("column_sums");
}

extern "C" size_t omnihd_bn_workspace_bytes(long long rows, int c) {
  if (rows <= 0 || c <= 0 || c % 8 || c > 2048) return 0;
  long long per;
  const int blocks = plan_blocks(rows, c / 8, &per);
  return align_up((size_t)blocks * 2 * c * sizeof(float), 256);
}

namespace {
template <typename T>
int channel_sums_t(const void* a, const void* b, const void* mask, const float* fwd_scale_shift, float* sums, long long rows,
                   int c, int mode, float mult, void* workspace, size_t workspace_bytes, void* stream,
                   int* partial_blocks = nullptr) {
  OMNIHD_REQUIRE(rows > 0 && c > 0 && c % 8 == 0 && c <= 2048, "rows > 0, C a multiple of 8, C <= 2048");
  OMNIHD_REQUIRE(a && sums && workspace && (mode == 0 || b), "null pointer");
  OMNIHD_REQUIRE(workspace_bytes >= omnihd_bn_workspace_bytes(rows, c), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  long long per;
  const int blocks = plan_blocks(rows, c / 8, &per);
  float* partial = static_cast<float*>(workspace);
  if (mode == 0)
    hipLaunchKernelGGL((k_bn_partial<T, 0>), dim3(blocks), dim3(256), 0, st, (const T*)a, (const T*)nullptr, (const T*)nullptr,
                       (const float*)nullptr, partial, rows, c / 8, per);
  else
    hipLaunchKernelGGL((k_bn_partial<T, 1>), dim3(blocks), dim3(256), 0, st, (const T*)a, (const T*)b, (const T*)mask,
                       fwd_scale_shift, partial, rows, c / 8, per);
  if (partial_blocks) *partial_blocks = blocks;                 // the caller's constants kernel reduces the partials itself
  else hipLaunchKernelGGL(k_bn_reduce, dim3((2 * c + 3) / 4), dim3(256), 0, st, partial, blocks, c, mult, sums);
  return check_launch("bn_channel_sums");
}

template <typename T>
int bwd_apply_t(const void* gy, const void* y_mask, const float* fwd_scale_shift, const void* x, const float* coef_a,
                const float* coef_b, const float* coef_c, void* gx, void* gres, long long rows, int c, void* stream,
                bf16_t* gx_hi = nullptr, bf16_t* gx_lo = nullptr, unsigned* amax_bits = nullptr) {
  OMNIHD_REQUIRE(rows >= 0 && c > 0 && c % 8 == 0, "C must be a positive multiple of 8");
  if (rows == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(gy && x && coef_a && coef_b && coef_c && (gx || (gx_hi && gx_lo)), "null pointer");
  const int64_t n_vec = (int64_t)rows * (c / 8);
  const dim3 grid(grid_for(n_vec, 256 * 2));
#define OMNIHD_APPLY(M, A)                                                                                                  \
  hipLaunchKernelGGL((k_bn_bwd_apply<T, M, A>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)gy, (const T*)y_mask,    \
                     (const T*)x, coef_a, coef_b, coef_c, fwd_scale_shift, (T*)gx, (T*)gres, n_vec, c / 8, gx_hi, gx_lo, amax_bits)
  if (amax_bits) {
    if constexpr (sizeof(T) == 4) {
      if (y_mask) OMNIHD_APPLY(1, true);
      else if (fwd_scale_shift) OMNIHD_APPLY(2, true);
      else OMNIHD_APPLY(0, true);
    } else {
      OMNIHD_REQUIRE(false, "the amax form takes fp32 rows");
    }
  } else if (y_mask) OMNIHD_APPLY(1, false);
  else if (fwd_scale_shift) OMNIHD_APPLY(2, false);
  else OMNIHD_APPLY(0, false);
#undef OMNIHD_APPLY
  return check_launch("bn_bwd_apply");
}
}  // namespace

/* mode 0: sums = (sum x, sum x^2) * mult over a [rows, c] bf16;  mode 1: (sum g', sum g' x) * mult with
 * g' = a * [mask > 0] (mask may be NULL), x = b.  The _f32 forms take fp32 rows.                          */
extern "C" int omnihd_bn_channel_sums(const void* a, const void* b, const void* mask, const float* fwd_scale_shift,
                                      float* sums, long long rows, int c, int mode, float mult, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  return channel_sums_t<bf16_t>(a, b, mask, fwd_scale_shift, sums, rows, c, mode, mult, workspace, workspace_bytes, stream);
}

extern "C" int omnihd_bn_channel_sums_f32(const float* a, const float* b, const float* mask, const float* fwd_scale_shift,
                                          float* sums, long long rows, int c, int mode, float mult, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  return channel_sums_t<float>(a, b, mask, fwd_scale_shift, sums, rows, c, mode, mult, workspace, workspace_bytes, stream);
}

extern "C" int omnihd_bn_fwd_consts(const float* stats, float rank_mult, const float* gamma, const float* beta, float eps,
                                    float momentum, float var_correction, int c, float* running_mean,
                                    float* running_var, float* scale, float* shift, float* mean, float* invstd,
                                    void* stream) {
  OMNIHD_REQUIRE(c > 0 && stats && gamma && beta && scale && shift && mean && invstd, "null pointer");
  OMNIHD_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running stats: both or neither");
  hipLaunchKernelGGL(k_bn_fwd_consts, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, rank_mult, 0,
                     (float*)nullptr, gamma, beta, eps, momentum, var_correction, c, running_mean, running_var, scale, shift,
                     mean, invstd);
  return check_launch("bn_fwd_consts");
}

extern "C" int omnihd_bn_bwd_consts(const float* local_sums, const float* global_sums, const float* gamma,
                                    const float* mean, const float* invstd, float inv_count, int c, float* dgamma,
                                    float* dbeta, float* coef_a, float* coef_b, float* coef_c, void* stream) {
  OMNIHD_REQUIRE(c > 0 && local_sums && global_sums && gamma && mean && invstd && dgamma && dbeta && coef_a && coef_b &&
                     coef_c, "null pointer");
  hipLaunchKernelGGL(k_bn_bwd_consts, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, local_sums, global_sums, 0,
                     gamma, mean, invstd, inv_count, c, dgamma, dbeta, coef_a, coef_b, coef_c);
  return check_launch("bn_bwd_consts");
}

extern "C" int omnihd_bn_bwd_apply(const void* gy, const void* y_mask, const float* fwd_scale_shift, const void* x,
                                   const float* coef_a, const float* coef_b, const float* coef_c, void* gx, void* gres,
                                   long long rows, int c, void* stream) {
  return bwd_apply_t<bf16_t>(gy, y_mask, fwd_scale_shift, x, coef_a, coef_b, coef_c, gx, gres, rows, c, stream);
}

extern "C" int omnihd_bn_bwd_apply_f32(const float* gy, const float* y_mask, const float* fwd_scale_shift, const float* x,
                                       const float* coef_a, const float* coef_b, const float* coef_c, float* gx, float* gres,
                                       long long rows, int c, void* stream) {
  return bwd_apply_t<float>(gy, y_mask, fwd_scale_shift, x, coef_a, coef_b, coef_c, gx, gres, rows, c, stream);
}

namespace {
template <typename T>
int train_fwd_t(const void* x, const void* res, const float* gamma, const float* beta, float* running_mean,
                float* running_var, float momentum, float eps, float var_correction, int relu, void* y, float* stats2c,
                float* consts4c, long long rows, int c, void* workspace, size_t workspace_bytes, void* stream,
                void* y_hi = nullptr, void* y_lo = nullptr) {
  int blocks = 0;
  int rc = channel_sums_t<T>(x, nullptr, nullptr, nullptr, stats2c, rows, c, 0, 1.0f, workspace, workspace_bytes, stream,
                             &blocks);
  if (rc) return rc;
  OMNIHD_REQUIRE(gamma && beta && consts4c && stats2c, "null pointer");
  hipLaunchKernelGGL(k_bn_fwd_consts, dim3((c + 15) / 16), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const float*>(workspace), 1.0f / (float)rows, blocks, stats2c, gamma, beta, eps, momentum,
                     var_correction, c, running_mean, running_var, consts4c, consts4c + c, consts4c + 2 * c, consts4c + 3 * c);
  rc = check_launch("bn_fwd_consts");
  if (rc) return rc;
  if (sizeof(T) == 2) return omnihd_affine_act_fwd(x, consts4c, consts4c + c, res, y, rows, c, relu, stream);
  if (y_hi)
    return omnihd_affine_act_fwd_f32_planes((const float*)x, consts4c, consts4c + c, (const float*)res, (float*)y, y_hi, y_lo, rows, c,
                                            relu, stream);
  return omnihd_affine_act_fwd_f32((const float*)x, consts4c, consts4c + c, (const float*)res, (float*)y, rows, c, relu, stream);
}

template <typename T>
int train_bwd_t(const void* gy, const void* y_mask, int relu_from_x, const void* x, const float* gamma, const float* consts4c,
                void* gx, void* gres, float* sums2c, float* out5c, long long rows, int c, void* workspace,
                size_t workspace_bytes, void* stream, void* gx_hi = nullptr, void* gx_lo = nullptr, void* amax_bits = nullptr) {
  const float* fss = (relu_from_x && !y_mask) ? consts4c : nullptr;    // consts4c starts with scale, shift
  int blocks = 0;
  int rc = channel_sums_t<T>(gy, x, y_mask, fss, sums2c, rows, c, 1, 1.0f, workspace, workspace_bytes, stream, &blocks);
  if (rc) return rc;
  const float* partial = static_cast<const float*>(workspace);
  hipLaunchKernelGGL(k_bn_bwd_consts, dim3((c + 15) / 16), dim3(256), 0, (hipStream_t)stream, partial, partial, blocks, gamma,
                     consts4c + 2 * c, consts4c + 3 * c, 1.0f / (float)rows, c, out5c, out5c + c, out5c + 2 * c, out5c + 3 * c,
                     out5c + 4 * c);
  rc = check_launch("bn_bwd_consts");
  if (rc) return rc;
  return bwd_apply_t<T>(gy, y_mask, fss, x, out5c + 2 * c, out5c + 3 * c, out5c + 4 * c, gx, gres, rows, c, stream,
                        static_cast<bf16_t*>(gx_hi), static_cast<bf16_t*>(gx_lo), static_cast<unsigned*>(amax_bits));
}
}  // namespace

/* Single-rank training step of BatchNorm in one call (no exchange between the statistics and their use):
 * forward  = channel sums -> constants (+ running stats) -> y = act(x * scale + shift (+ res));
 * consts [4, c] receives scale, shift, mean, invstd (kept for the backward).                            */
extern "C" int omnihd_bn_train_fwd(const void* x, const void* res, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum, float eps,
                                   float var_correction, int relu, void* y, float* stats2c, float* consts4c,
                                   long long rows, int c, void* workspace, size_t workspace_bytes, void* stream) {
  return train_fwd_t<bf16_t>(x, res, gamma, beta, running_mean, running_var, momentum, eps, var_correction, relu, y, stats2c,
                             consts4c, rows, c, workspace, workspace_bytes, stream);
}

extern "C" int omnihd_bn_train_fwd_f32(const float* x, const float* res, const float* gamma, const float* beta,
                                       float* running_mean, float* running_var, float momentum, float eps,
                                       float var_correction, int relu, float* y, float* stats2c, float* consts4c,
                                       long long rows, int c, void* workspace, size_t workspace_bytes, void* stream) {
  return train_fwd_t<float>(x, res, gamma, beta, running_mean, running_var, momentum, eps, var_correction, relu, y, stats2c,
                            consts4c, rows, c, workspace, workspace_bytes, stream);
}

/* ..._f32_planes: the same call, y additionally written as its two bf16 planes (omnihd_split_f32's output) for the next
 * fp32-grade convolution.                                                                                 */
extern "C" int omnihd_bn_train_fwd_f32_planes(const float* x, const float* res, const float* gamma, const float* beta,
                                              float* running_mean, float* running_var, float momentum, float eps,
                                              float var_correction, int relu, float* y, void* y_hi, void* y_lo, float* stats2c,
                                              float* consts4c, long long rows, int c, void* workspace, size_t workspace_bytes,
                                              void* stream) {
  // (y_lo NULL: y_hi receives the IEEE-half plane of y for the next TF32-grade convolution instead of the two bf16 planes)
  OMNIHD_REQUIRE(y_hi && ((reinterpret_cast<uintptr_t>(y_hi) | reinterpret_cast<uintptr_t>(y_lo)) & 15u) == 0, "plane pointers");
  return train_fwd_t<float>(x, res, gamma, beta, running_mean, running_var, momentum, eps, var_correction, relu, y, stats2c,
                            consts4c, rows, c, workspace, workspace_bytes, stream, y_hi, y_lo);
}

/* backward = masked channel sums -> dgamma/dbeta + coefficients -> gx (and gres).  out5c [5, c] receives
 * dgamma, dbeta and the three coefficient vectors; sums2c is scratch.                                     */
extern "C" int omnihd_bn_train_bwd(const void* gy, const void* y_mask, int relu_from_x, const void* x, const float* gamma,
                                   const float* consts4c, void* gx, void* gres, float* sums2c, float* out5c,
                                   long long rows, int c, void* workspace, size_t workspace_bytes, void* stream) {
  return train_bwd_t<bf16_t>(gy, y_mask, relu_from_x, x, gamma, consts4c, gx, gres, sums2c, out5c, rows, c, workspace,
                             workspace_bytes, stream);
}

extern "C" int omnihd_bn_train_bwd_f32(const float* gy, const float* y_mask, int relu_from_x, const float* x,
                                       const float* gamma, const float* consts4c, float* gx, float* gres, float* sums2c,
                                       float* out5c, long long rows, int c, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  return train_bwd_t<float>(gy, y_mask, relu_from_x, x, gamma, consts4c, gx, gres, sums2c, out5c, rows, c, workspace,
                            workspace_bytes, stream);
}

/* ..._f32_amax: max |gx| additionally accumulated into *amax_bits (bit pattern of a float >= 0, zeroed by the caller): the scale of
 * the half cast of gx for the TF32-grade convolution in front of this layer (omnihd_cast_f16, scaled == 3). */
extern "C" int omnihd_bn_train_bwd_f32_amax(const float* gy, const float* y_mask, int relu_from_x, const float* x,
                                            const float* gamma, const float* consts4c, float* gx, void* amax_bits, float* gres,
                                            float* sums2c, float* out5c, long long rows, int c, void* workspace,
                                            size_t workspace_bytes, void* stream) {
  OMNIHD_REQUIRE(gx && amax_bits, "null pointer");
  return train_bwd_t<float>(gy, y_mask, relu_from_x, x, gamma, consts4c, gx, gres, sums2c, out5c, rows, c, workspace,
                            workspace_bytes, stream, nullptr, nullptr, amax_bits);
}

/* ..._f32_planes: gx additionally written as its two bf16 planes for the convolution in front of this layer. */
extern "C" int omnihd_bn_train_bwd_f32_planes(const float* gy, const float* y_mask, int relu_from_x, const float* x,
                                              const float* gamma, const float* consts4c, float* gx, void* gx_hi, void* gx_lo,
                                              float* gres, float* sums2c, float* out5c, long long rows, int c, void* workspace,
                                              size_t workspace_bytes, void* stream) {
  OMNIHD_REQUIRE(gx_hi && gx_lo && ((reinterpret_cast<uintptr_t>(gx_hi) | reinterpret_cast<uintptr_t>(gx_lo)) & 15u) == 0, "plane pointers");
  return train_bwd_t<float>(gy, y_mask, relu_from_x, x, gamma, consts4c, gx, gres, sums2c, out5c, rows, c, workspace,
                            workspace_bytes, stream, gx_hi, gx_lo);
}
