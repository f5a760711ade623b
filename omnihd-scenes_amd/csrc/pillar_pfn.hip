// Fused pillar feature net for gfx950: decorate -> Linear (no bias) -> BatchNorm over the channel dim -> ReLU -> max over the
// points of a pillar, forward and backward, without ever materialising the (pillars x points x channels) activations.
//
// Reference: PillarFeatureNetV1.forward (projects/mmdet3d_plugin/rcfusion/voxel_encoders/pillar_encoder.py:378-432) +
// PFNLayer.forward (.../utils.py:144-181); radar variant RadarPillarFeatureNet.forward (pillar_encoder.py:91-153) +
// PFNLayer_Radar.forward (utils.py:229-280), whose three Linear/BatchNorm branches are ONE Linear with a block-sparse weight
// followed by a per-channel BatchNorm.  The reference runs ~25 small torch kernels forward (cat / mask / linear / two permuted
// copies around BatchNorm1d / relu / max) and their autograd mirrors backward.
//
// What makes one pass possible: the layer's input is only K <= 16 "decorated" channels per point
//     x = [ base (raw point, x/y replaced by the pillar-centre offset when legacy) | xyz - mean_xyz | xy - pillar centre |
//           (|base xyz|) | (radar: v_x v_y power snr - their mean) ],        zero for the padded slots of a pillar,
// and y = W x is linear in it, so the BatchNorm statistics of y over all N = pillars * slots rows follow from the first and
// second moments of x:   mean_c = W_c . E[x],   E[y_c^2] = W_c . E[x x^T] . W_c^T   (K + K*K numbers instead of N*C).
//   forward : k_pfn_moments  per-workgroup partial sums of x and x x^T (256 slots per workgroup)
//             k_pfn_reduce   fixed-order reduction of the partials in double -> moments / N
//             (multi-GPU: the K + K*K moments are averaged over the ranks between these two — mean of rank means, the
//              semantics of the reference's naiveSyncBN, ops/norm.py:65-72)
//             k_pfn_consts   mean / variance / scale / shift per channel (+ running statistics)
//             k_pfn_apply    out[p][c] = max_k relu(scale_c * (W_c . x_pk) + shift_c): one lane per channel, W_c in registers
//   backward: k_pfn_bwd_sums recomputes y, the ReLU mask and the arg-max slot (first maximum, as torch.max), accumulates per
//             channel A = sum g, B = sum g * yhat and G[c][j] = sum g * x_j (only the arg-max slot of a pillar carries gradient)
//             k_pfn_reduce   the same fixed-order reduction
//             k_pfn_bwd_final dW, dgamma, dbeta from A, B, G and the moments (the dense "minus mean" terms of the BatchNorm
//             backward are again linear in the moments of x)
// No atomics; every reduction has a fixed order (run-to-run identical).
#include "common.h"

#pragma clang fp contract(off)   // the decoration follows torch's separate multiply / add / subtract roundings

namespace omnihd {
namespace {

constexpr int kBlock = 256;
constexpr int kMaxK = 16;        // decorated channels
constexpr int kC = 64;           // output channels (one lane each)

struct PfnGeom {
  int m, p, f, k;                // pillars, slots per pillar, raw channels, decorated channels
  float vx, vy, x_off, y_off;
  int cluster, center, distance, legacy, radar;
};

// decorated channels of slot `s` of pillar `pil` -> x[0..k); returns false for a padded slot (x = 0)
__device__ __forceinline__ bool decorate(const PfnGeom& g, const float* __restrict__ voxels, const int* __restrict__ num_points,
                                         const int* __restrict__ coors, int pil, int s, float* x) {
#pragma unroll
  for (int j = 0; j < kMaxK; ++j) x[j] = 0.f;
  const int n = num_points[pil];
  if (s >= n) return false;
  const float* pts = voxels + (size_t)pil * g.p * g.f;
  const float* me = pts + (size_t)s * g.f;
  const float cnt = (float)n;
  int o = 0;
  float cx = 0.f, cy = 0.f;
  if (g.center) {
    cx = (float)coors[pil * 4 + 3] * g.vx + g.x_off;
    cy = (float)coors[pil * 4 + 2] * g.vy + g.y_off;
  }
  // base: the raw point; legacy: x, y replaced by the pillar-centre offsets (the reference's f_center is a view of them)
  for (int j = 0; j < g.f; ++j) x[o + j] = me[j];
  if (g.center && g.legacy) {
    x[o + 0] = me[0] - cx;
    x[o + 1] = me[1] - cy;
  }
  const int base = o;
  o += g.f;
  if (g.cluster) {
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int q = 0; q < g.p; ++q) {          // all slots: padded ones hold zeros (pillar_encoder.py:393-397)
      sx += pts[q * g.f + 0];
      sy += pts[q * g.f + 1];
      sz += pts[q * g.f + 2];
    }
    x[o + 0] = me[0] - sx / cnt;
    x[o + 1] = me[1] - sy / cnt;
    x[o + 2] = me[2] - sz / cnt;
    o += 3;
  }
  if (g.center) {
    x[o + 0] = me[0] - cx;
    x[o + 1] = me[1] - cy;
    o += 2;
  }
  if (g.distance) {
    x[o] = sqrtf(x[base] * x[base] + x[base + 1] * x[base + 1] + x[base + 2] * x[base + 2]);
    o += 1;
  }
  if (g.radar) {                               // v_x, v_y, power, snr minus their pillar means (pillar_encoder.py:137-141)
    for (int j = 0; j < 4; ++j) {
      float sj = 0.f;
      for (int q = 0; q < g.p; ++q) sj += pts[q * g.f + 3 + j];
      x[o + j] = me[3 + j] - sj / cnt;
    }
    o += 4;
  }
  return true;
}

// partial[q * n_blocks + b]: q < k: sum of x_q; q = k + i*k + j: sum of x_i x_j over the 256 slots of workgroup b
__global__ __launch_bounds__(kBlock) void k_pfn_moments(PfnGeom g, const float* __restrict__ voxels,
                                                        const int* __restrict__ num_points, const int* __restrict__ coors,
                                                        float* __restrict__ partial, int n_blocks) {
  __shared__ float s_x[kBlock][kMaxK + 1];
  const int tid = threadIdx.x;
  const long long slot = (long long)blockIdx.x * kBlock + tid;
  const long long total = (long long)g.m * g.p;
  float x[kMaxK];
  if (slot < total) {
    decorate(g, voxels, num_points, coors, (int)(slot / g.p), (int)(slot % g.p), x);
  } else {
#pragma unroll
    for (int j = 0; j < kMaxK; ++j) x[j] = 0.f;
  }
#pragma unroll
  for (int j = 0; j < kMaxK; ++j) s_x[tid][j] = x[j];
  __syncthreads();
  const int i = tid / kMaxK, j = tid % kMaxK;      // 256 threads = 16 x 16 pairs
  float s2 = 0.f, s1 = 0.f;
  for (int r = 0; r < kBlock; ++r) {
    const float a = s_x[r][i], b = s_x[r][j];
    s2 = fmaf(a, b, s2);
    if (i == 0) s1 += b;
  }
  if (i < g.k && j < g.k) partial[(size_t)(g.k + i * g.k + j) * n_blocks + blockIdx.x] = s2;
  if (i == 0 && j < g.k) partial[(size_t)j * n_blocks + blockIdx.x] = s1;
}

// out[q] = scale * sum_b partial[q * n_blocks + b]   (64 lanes, lane-strided then a fixed butterfly; double)
template <typename T>
__global__ __launch_bounds__(64) void k_pfn_reduce(const float* __restrict__ partial, int n_blocks, double scale,
                                                   T* __restrict__ out) {
  const int q = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < n_blocks; b += 64) s += (double)partial[(size_t)q * n_blocks + b];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (threadIdx.x == 0) out[q] = (T)(s * scale);
}

// consts[0..C) mean, [C..2C) inverse standard deviation, [2C..3C) scale, [3C..4C) shift
__global__ __launch_bounds__(kC) void k_pfn_consts(const float* __restrict__ weight, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, const double* __restrict__ moments, int k,
                                                   double n_rows, float eps, float momentum, int unbiased, int use_running,
                                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                                   float* __restrict__ consts) {
  const int c = threadIdx.x;
  double mean, var;
  if (use_running) {
    mean = running_mean[c];
    var = running_var[c];
  } else {
    double m1 = 0.0, m2 = 0.0;
    for (int i = 0; i < k; ++i) {
      const double wi = weight[c * k + i];
      m1 += wi * moments[i];
      double row = 0.0;
      for (int j = 0; j < k; ++j) row += (double)weight[c * k + j] * moments[k + i * k + j];
      m2 += wi * row;
    }
    mean = m1;
    var = m2 - m1 * m1;
    if (var < 0.0) var = 0.0;
    if (running_mean != nullptr) {
      const double uv = (unbiased && n_rows > 1.0) ? var * n_rows / (n_rows - 1.0) : var;
      running_mean[c] = (float)((double)running_mean[c] + (double)momentum * (mean - (double)running_mean[c]));
      running_var[c] = (float)((double)running_var[c] + (double)momentum * (uv - (double)running_var[c]));
    }
  }
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float scale = gamma[c] * invstd;
  consts[c] = (float)mean;
  consts[kC + c] = invstd;
  consts[2 * kC + c] = scale;
  consts[3 * kC + c] = beta[c] - (float)mean * scale;
}

constexpr int kPillarsPerBlock = 16;   // 4 wavefronts x 4 pillars

// decorated slots of the workgroup's pillars -> LDS; returns the number of pillars staged
__device__ __forceinline__ int stage_pillars(const PfnGeom& g, const float* voxels, const int* num_points, const int* coors,
                                             int first, float (*s_x)[kMaxK + 1], int max_slots) {
  const int npil = min(kPillarsPerBlock, g.m - first);
  for (int t = threadIdx.x; t < npil * g.p; t += kBlock) {
    float x[kMaxK];
    decorate(g, voxels, num_points, coors, first + t / g.p, t % g.p, x);
#pragma unroll
    for (int j = 0; j < kMaxK; ++j) s_x[t][j] = x[j];
  }
  return npil;
}

constexpr int kMaxSlots = 64;          // slots per pillar the apply / backward kernels stage (P <= 64)

__global__ __launch_bounds__(kBlock) void k_pfn_apply(PfnGeom g, const float* __restrict__ voxels,
                                                      const int* __restrict__ num_points, const int* __restrict__ coors,
                                                      const float* __restrict__ weight, const float* __restrict__ consts,
                                                      float* __restrict__ out) {
  extern __shared__ float s_dyn[];
  float (*s_x)[kMaxK + 1] = reinterpret_cast<float (*)[kMaxK + 1]>(s_dyn);
  const int c = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float w[kMaxK];
#pragma unroll
  for (int j = 0; j < kMaxK; ++j) w[j] = j < g.k ? weight[c * g.k + j] : 0.f;
  const float scale = consts[2 * kC + c], shift = consts[3 * kC + c];
  for (int first = blockIdx.x * kPillarsPerBlock; first < g.m; first += gridDim.x * kPillarsPerBlock) {
    __syncthreads();
    const int npil = stage_pillars(g, voxels, num_points, coors, first, s_x, kMaxSlots);
    __syncthreads();
    for (int q = wv; q < npil; q += 4) {
      float best = -INFINITY;
      for (int s = 0; s < g.p; ++s) {
        const float* x = s_x[q * g.p + s];
        float y = 0.f;
#pragma unroll
        for (int j = 0; j < kMaxK; ++j) y = fmaf(x[j], w[j], y);
        const float z = fmaxf(fmaf(y, scale, shift), 0.f);
        best = fmaxf(best, z);
      }
      out[(size_t)(first + q) * kC + c] = best;
    }
  }
}

// per workgroup b: partial[q * n_blocks + b], q = c (A), C + c (B), 2C + c*k + j (G)
__global__ __launch_bounds__(kBlock) void k_pfn_bwd_sums(PfnGeom g, const float* __restrict__ voxels,
                                                         const int* __restrict__ num_points, const int* __restrict__ coors,
                                                         const float* __restrict__ weight, const float* __restrict__ consts,
                                                         const float* __restrict__ grad_out, float* __restrict__ partial,
                                                         int n_blocks) {
  extern __shared__ float s_dyn[];
  float (*s_x)[kMaxK + 1] = reinterpret_cast<float (*)[kMaxK + 1]>(s_dyn);
  __shared__ float s_red[kC][kMaxK + 2];
  const int c = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float w[kMaxK], G[kMaxK];
#pragma unroll
  for (int j = 0; j < kMaxK; ++j) {
    w[j] = j < g.k ? weight[c * g.k + j] : 0.f;
    G[j] = 0.f;
  }
  const float mean = consts[c], invstd = consts[kC + c], scale = consts[2 * kC + c], shift = consts[3 * kC + c];
  float A = 0.f, B = 0.f;
  for (int first = blockIdx.x * kPillarsPerBlock; first < g.m; first += gridDim.x * kPillarsPerBlock) {
    __syncthreads();
    const int npil = stage_pillars(g, voxels, num_points, coors, first, s_x, kMaxSlots);
    __syncthreads();
    for (int q = wv; q < npil; q += 4) {
      float best = -INFINITY, ybest = 0.f;
      int sbest = 0;
      for (int s = 0; s < g.p; ++s) {
        const float* x = s_x[q * g.p + s];
        float y = 0.f;
#pragma unroll
        for (int j = 0; j < kMaxK; ++j) y = fmaf(x[j], w[j], y);
        const float z = fmaxf(fmaf(y, scale, shift), 0.f);
        if (z > best) {                              // strict: the FIRST maximum keeps the gradient (torch.max)
          best = z; ybest = y; sbest = s;
        }
      }
      const float go = best > 0.f ? grad_out[(size_t)(first + q) * kC + c] : 0.f;    // ReLU'(0) = 0
      A += go;
      B = fmaf(go, (ybest - mean) * invstd, B);
      const float* x = s_x[q * g.p + sbest];
#pragma unroll
      for (int j = 0; j < kMaxK; ++j) G[j] = fmaf(go, x[j], G[j]);
    }
  }
  // wavefronts 1..3 hand their sums to wavefront 0 (fixed order)
  for (int src = 1; src < 4; ++src) {
    __syncthreads();
    if (wv == src) {
      s_red[c][0] = A; s_red[c][1] = B;
#pragma unroll
      for (int j = 0; j < kMaxK; ++j) s_red[c][2 + j] = G[j];
    }
    __syncthreads();
    if (wv == 0) {
      A += s_red[c][0]; B += s_red[c][1];
#pragma unroll
      for (int j = 0; j < kMaxK; ++j) G[j] += s_red[c][2 + j];
    }
  }
  if (wv == 0) {
    partial[(size_t)c * n_blocks + blockIdx.x] = A;
    partial[(size_t)(kC + c) * n_blocks + blockIdx.x] = B;
    for (int j = 0; j < g.k; ++j) partial[(size_t)(2 * kC + c * g.k + j) * n_blocks + blockIdx.x] = G[j];
  }
}

// sums: [A (C) | B (C) | G (C x k)] of THIS rank; ab: [A | B] summed over all ranks (the same array on one rank).
// moments: this rank's E[x], E[x x^T]; consts: the (global) statistics the forward used; n_eff = ranks * rows of this rank.
__global__ __launch_bounds__(kC) void k_pfn_bwd_final(const float* __restrict__ sums, const float* __restrict__ ab,
                                                      const double* __restrict__ moments, const float* __restrict__ weight,
                                                      const float* __restrict__ gamma, const float* __restrict__ consts, int k,
                                                      double n_rows, double n_eff, float* __restrict__ dweight,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = threadIdx.x;
  const double mean = consts[c], invstd = consts[kC + c];
  const double A = ab[c], B = ab[kC + c];
  dgamma[c] = sums[kC + c];
  dbeta[c] = sums[c];
  const double gs = (double)gamma[c] * invstd;
  for (int j = 0; j < k; ++j) {
    // sum over this rank's rows of yhat_c * x_j = (W_c . S2[:, j] - mean_c S1_j) * invstd, S = n_rows * moments
    double wy = 0.0;
    for (int i = 0; i < k; ++i) wy += (double)weight[c * k + i] * moments[k + i * k + j];
    const double s1j = moments[j] * n_rows;
    const double yhx = (wy * n_rows - mean * s1j) * invstd;
    const double corr = n_eff > 0.0 ? (A * s1j + B * yhx) / n_eff : 0.0;     // n_eff <= 0: statistics that did not depend on the batch
    dweight[c * k + j] = (float)(gs * ((double)sums[2 * kC + c * k + j] - corr));
  }
}

int geom_of(PfnGeom* g, int m, int p, int f, float vx, float vy, float x_off, float y_off, int flags) {
  g->m = m; g->p = p; g->f = f;
  g->vx = vx; g->vy = vy; g->x_off = x_off; g->y_off = y_off;
  g->cluster = flags & 1; g->center = (flags >> 1) & 1; g->distance = (flags >> 2) & 1; g->legacy = (flags >> 3) & 1;
  g->radar = (flags >> 4) & 1;
  g->k = f + 3 * g->cluster + 2 * g->center + g->distance + 4 * g->radar;
  return g->k;
}

}  // namespace
}  // namespace omnihd

using namespace omnihd;

extern "C" int omnihd_pfn_channels(int f, int flags) {
  PfnGeom g;
  return geom_of(&g, 0, 1, f, 0, 0, 0, 0, flags);
}

extern "C" size_t omnihd_pfn_workspace_bytes(int m, int p, int k) {
  if (m <= 0 || p <= 0 || k <= 0) return 256;
  const size_t nb_m = ((size_t)m * p + kBlock - 1) / kBlock;
  const size_t nb_b = 1024;
  const size_t a = nb_m * (size_t)(k + k * k), b = nb_b * (size_t)(2 * kC + kC * k);
  return align_up((a > b ? a : b) * sizeof(float), 256);
}

#define OMNIHD_PFN_GEOM()                                                                                               \
  PfnGeom g;                                                                                                            \
  const int k = geom_of(&g, m, p, f, vx, vy, x_off, y_off, flags);                                                      \
  OMNIHD_REQUIRE(m >= 0 && p > 0 && p <= kMaxSlots && f >= 3 && k <= kMaxK, "pillars >= 0, 1..64 slots, >= 3 raw channels, <= 16 decorated channels"); \
  OMNIHD_REQUIRE(!g.radar || f >= 7, "the radar decoration needs 7 raw channels (x y z vx vy power snr)")

extern "C" int omnihd_pfn_moments(const float* voxels, const int* num_points, const int* coors, int m, int p, int f, float vx,
                                  float vy, float x_off, float y_off, int flags, double* moments, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  OMNIHD_PFN_GEOM();
  OMNIHD_REQUIRE(m > 0 && voxels && num_points && coors && moments && workspace, "null pointer / empty input");
  OMNIHD_REQUIRE(workspace_bytes >= omnihd_pfn_workspace_bytes(m, p, k), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)(((long long)m * p + kBlock - 1) / kBlock);
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(k_pfn_moments, dim3(nb), dim3(kBlock), 0, st, g, voxels, num_points, coors, partial, nb);
  hipLaunchKernelGGL(k_pfn_reduce<double>, dim3(k + k * k), dim3(64), 0, st, partial, nb, 1.0 / ((double)m * p), moments);
  return check_launch("pfn_moments");
}

extern "C" int omnihd_pfn_consts(const float* weight, const float* gamma, const float* beta, const double* moments, int k,
                                 long long n_rows, float eps, float momentum, int unbiased, int use_running,
                                 float* running_mean, float* running_var, float* consts, void* stream) {
  OMNIHD_REQUIRE(k > 0 && k <= kMaxK && weight && gamma && beta && consts, "arguments");
  OMNIHD_REQUIRE(use_running ? (running_mean && running_var) : (moments != nullptr), "statistics source");
  OMNIHD_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running_mean and running_var come together");
  hipLaunchKernelGGL(k_pfn_consts, dim3(1), dim3(kC), 0, (hipStream_t)stream, weight, gamma, beta, moments, k, (double)n_rows, eps,
                     momentum, unbiased, use_running, running_mean, running_var, consts);
  return check_launch("pfn_consts");
}

extern "C" int omnihd_pfn_apply(const float* voxels, const int* num_points, const int* coors, int m, int p, int f, float vx,
                                float vy, float x_off, float y_off, int flags, const float* weight, const float* consts,
                                float* out, void* stream) {
  OMNIHD_PFN_GEOM();
  if (m == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(voxels && num_points && coors && weight && consts && out, "null pointer");
  const size_t lds = (size_t)kPillarsPerBlock * p * (kMaxK + 1) * sizeof(float);
  OMNIHD_REQUIRE(lds <= 96 * 1024, "too many slots per pillar");
  const int nb = min((m + kPillarsPerBlock - 1) / kPillarsPerBlock, 4096);
  hipLaunchKernelGGL(k_pfn_apply, dim3(nb), dim3(kBlock), lds, (hipStream_t)stream, g, voxels, num_points, coors, weight, consts, out);
  return check_launch("pfn_apply");
}

extern "C" int omnihd_pfn_bwd_sums(const float* voxels, const int* num_points, const int* coors, int m, int p, int f, float vx,
                                   float vy, float x_off, float y_off, int flags, const float* weight, const float* consts,
                                   const float* grad_out, float* sums, void* workspace, size_t workspace_bytes, void* stream) {
  OMNIHD_PFN_GEOM();
  OMNIHD_REQUIRE(m > 0 && voxels && num_points && coors && weight && consts && grad_out && sums && workspace, "null pointer / empty input");
  OMNIHD_REQUIRE(workspace_bytes >= omnihd_pfn_workspace_bytes(m, p, k), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)kPillarsPerBlock * p * (kMaxK + 1) * sizeof(float);
  OMNIHD_REQUIRE(lds <= 96 * 1024, "too many slots per pillar");
  const int nb = min((m + kPillarsPerBlock - 1) / kPillarsPerBlock, 1024);
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(k_pfn_bwd_sums, dim3(nb), dim3(kBlock), lds, st, g, voxels, num_points, coors, weight, consts, grad_out, partial, nb);
  hipLaunchKernelGGL(k_pfn_reduce<float>, dim3(2 * kC + kC * k), dim3(64), 0, st, partial, nb, 1.0, sums);
  return check_launch("pfn_bwd_sums");
}

extern "C" int omnihd_pfn_bwd_final(const float* sums, const float* ab_all_ranks, const double* moments, const float* weight,
                                    const float* gamma, const float* consts, int k, long long n_rows, int n_ranks, float* dweight,
                                    float* dgamma, float* dbeta, void* stream) {
  OMNIHD_REQUIRE(k > 0 && k <= kMaxK && n_rows > 0 && sums && ab_all_ranks && moments && weight && gamma && consts &&
                     dweight && dgamma && dbeta, "arguments");
  hipLaunchKernelGGL(k_pfn_bwd_final, dim3(1), dim3(kC), 0, (hipStream_t)stream, sums, ab_all_ranks, moments, weight, gamma, consts, k,
                     (double)n_rows, (double)n_rows * n_ranks, dweight, dgamma, dbeta);
  return check_launch("pfn_bwd_final");
}
