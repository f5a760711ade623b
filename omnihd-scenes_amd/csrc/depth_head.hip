// Depth-head epilogue of the LSS camera stream: softmax over the D depth logits + depth / context split + the two layouts
// the pooling kernels read, in ONE pass each way.
//
// What the reference does (cam_stream_lss_bevpoolv2_depthnet.py:134-143, :290, ops/bev_pool_v2/bev_pool.py:20-21):
//     x = depthnet(x)                           (B*N, D + C, fH, fW)
//     depth = x[:, :D].softmax(dim=1)           -> later .float().contiguous()          (B, N, D, fH, fW) fp32
//     feat  = x[:, D:D+C]                       -> permute(0,1,3,4,2).contiguous().float()  (B, N, fH, fW, C) fp32
// i.e. concat -> slice -> softmax -> cast -> two layout copies (5-7 launches, ~130 MB of traffic at 6 x 64 x 176), and the KL
// depth loss transposes the distribution back to pixel-major rows.  With channels-last activations the logits of one pixel
// are D contiguous values: a workgroup takes 64 consecutive pixels of one image, stages their logits in LDS (row stride D:
// D = 59 is odd, column reads are conflict-free), four lanes per pixel reduce max and sum with DPP-free sub-wave shuffles,
// and the tile is written D-major (one 256-byte segment per depth bin: the (B,N,D,fH,fW) tensor the pooling gathers from),
// optionally pixel-major as well (what the depth loss reads), next to the fp32 context rows (B,N,fH,fW,C).  The tensors the
// pooling forward gathers from are therefore written by the launch directly in front of it.
//
// Backward: g_logits = y * (g - sum_d g*y) with g = g_dmajor (from the pooling backward) + g_rows (from the depth loss), both
// optional; g_context = cast(g_feat).  HBM-bound streaming, fp32 arithmetic, no atomics.
#include "common.h"
#include "bn_vec.h"

namespace omnihd {
namespace {

constexpr int kTile = 64;          // pixels per workgroup
constexpr int kThreads = 256;

__device__ __forceinline__ float ld(const float* p) { return *p; }
__device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ void st(float* p, float v) { *p = v; }
__device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }

// context rows (pitch ld_ctx elements of T) -> packed fp32 rows; 4 channels per lane
template <typename T>
__device__ __forceinline__ void copy_rows_to_f32(const T* __restrict__ src, long long ld_src, float* __restrict__ dst, int c,
                                                 long long row0, int n_rows, int tid) {
  const int c4 = c >> 2;
  for (int i = tid; i < n_rows * c4; i += kThreads) {
    const int r = i / c4, q = i - r * c4;
    const T* s = src + (row0 + r) * ld_src + 4 * q;
    float4 v;
    if constexpr (sizeof(T) == 4) {
      v = *reinterpret_cast<const float4*>(s);
    } else {
      const uint2 raw = *reinterpret_cast<const uint2*>(s);
      v = make_float4(__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u), __uint_as_float(raw.y << 16),
                      __uint_as_float(raw.y & 0xffff0000u));
    }
    *reinterpret_cast<float4*>(dst + (row0 + r) * (long long)c + 4 * q) = v;
  }
}

template <typename T>
__device__ __forceinline__ void copy_f32_to_rows(const float* __restrict__ src, T* __restrict__ dst, long long ld_dst, int c,
                                                 long long row0, int n_rows, int tid) {
  const int c4 = c >> 2;
  for (int i = tid; i < n_rows * c4; i += kThreads) {
    const int r = i / c4, q = i - r * c4;
    const float4 v = *reinterpret_cast<const float4*>(src + (row0 + r) * (long long)c + 4 * q);
    T* d = dst + (row0 + r) * ld_dst + 4 * q;
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(d) = v;
    } else {
      uint2 raw;
      raw.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
      raw.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
      *reinterpret_cast<uint2*>(d) = raw;
    }
  }
}

// grid: (tiles per image, images).  LDS: kTile * D floats.
template <typename T>
__global__ __launch_bounds__(kThreads) void k_depth_head_fwd(const T* __restrict__ logits, long long ld_logits,
                                                             const T* __restrict__ context, long long ld_ctx, int d_bins, int c,
                                                             int fhw, float* __restrict__ depth, float* __restrict__ depth_rows,
                                                             float* __restrict__ feat) {
  extern __shared__ __attribute__((aligned(16))) float s_t[];             // [kTile][d_bins]
  const int tid = threadIdx.x;
  const int img = blockIdx.y;
  const int p0 = blockIdx.x * kTile;
  const int np = min(kTile, fhw - p0);
  const long long row0 = (long long)img * fhw + p0;
  // ---- logits of the tile -> LDS (coalesced along the pixel rows) ----------------------------------
  const int n_el = np * d_bins;
  for (int e = tid; e < n_el; e += kThreads) {
    const int r = e / d_bins, d = e - r * d_bins;
    s_t[e] = ld(logits + (row0 + r) * ld_logits + d);
  }
  // the context rows do not depend on the softmax: their loads are in flight while the LDS phase runs
  if (context) copy_rows_to_f32(context, ld_ctx, feat, c, row0, np, tid);
  __syncthreads();
  // ---- softmax per pixel: 4 lanes per pixel, lane j takes bins j, j+4, ... -------------------------
  {
    const int r = tid >> 2, j = tid & 3;
    if (r < np) {
      float* row = s_t + r * d_bins;
      float m = -INFINITY;
      for (int d = j; d < d_bins; d += 4) m = fmaxf(m, row[d]);
      m = fmaxf(m, __shfl_xor(m, 1, 4));
      m = fmaxf(m, __shfl_xor(m, 2, 4));
      float sum = 0.f;
      for (int d = j; d < d_bins; d += 4) {
        const float e = expf(row[d] - m);
        row[d] = e;
        sum += e;
      }
      sum += __shfl_xor(sum, 1, 4);
      sum += __shfl_xor(sum, 2, 4);
      const float inv = 1.0f / sum;
      for (int d = j; d < d_bins; d += 4) row[d] *= inv;
    }
  }
  __syncthreads();
  // ---- D-major output: one segment of np floats per depth bin -------------------------------------
  {
    const int lane = tid & 63, w = tid >> 6;
    float* base = depth + (long long)img * d_bins * fhw + p0;
    if (lane < np)
      for (int d = w; d < d_bins; d += kThreads / 64) base[(long long)d * fhw + lane] = s_t[lane * d_bins + d];
  }
  // ---- pixel-major output (rows of D values), the layout of the tile in LDS ------------------------
  if (depth_rows) {
    float* base = depth_rows + row0 * d_bins;
    for (int e = tid; e < n_el; e += kThreads) base[e] = s_t[e];
  }
}

// g_logits[p][d] = y * (g - sum_d g*y),  g = g_depth[d][p] (+ g_rows[p][d]);  g_context = cast(g_feat)
template <typename T>
__global__ __launch_bounds__(kThreads) void k_depth_head_bwd(const float* __restrict__ depth, const float* __restrict__ g_depth,
                                                             const float* __restrict__ g_rows, const float* __restrict__ g_feat,
                                                             int d_bins, int c, int fhw, T* __restrict__ g_logits,
                                                             long long ld_gl, T* __restrict__ g_ctx, long long ld_gc) {
  extern __shared__ __attribute__((aligned(16))) float s_t[];             // y [kTile][d_bins], g [kTile][d_bins]
  const int tid = threadIdx.x;
  const int img = blockIdx.y;
  const int p0 = blockIdx.x * kTile;
  const int np = min(kTile, fhw - p0);
  const long long row0 = (long long)img * fhw + p0;
  float* s_y = s_t;
  float* s_g = s_t + kTile * d_bins;
  const int n_el = np * d_bins;
  {
    const int lane = tid & 63, w = tid >> 6;
    const long long off = (long long)img * d_bins * fhw + p0;
    if (lane < np)
      for (int d = w; d < d_bins; d += kThreads / 64) {
        s_y[lane * d_bins + d] = depth[off + (long long)d * fhw + lane];
        s_g[lane * d_bins + d] = g_depth ? g_depth[off + (long long)d * fhw + lane] : 0.f;
      }
  }
  if (g_ctx) copy_f32_to_rows(g_feat, g_ctx, ld_gc, c, row0, np, tid);
  __syncthreads();
  if (g_rows) {
    const float* base = g_rows + row0 * d_bins;
    for (int e = tid; e < n_el; e += kThreads) s_g[e] += base[e];          // element e is touched by this thread only
  }
  __syncthreads();
  {
    const int r = tid >> 2, j = tid & 3;
    if (r < np) {
      const float* y = s_y + r * d_bins;
      float* g = s_g + r * d_bins;
      float dot = 0.f;
      for (int d = j; d < d_bins; d += 4) dot = fmaf(g[d], y[d], dot);
      dot += __shfl_xor(dot, 1, 4);
      dot += __shfl_xor(dot, 2, 4);
      for (int d = j; d < d_bins; d += 4) g[d] = y[d] * (g[d] - dot);
    }
  }
  __syncthreads();
  for (int e = tid; e < n_el; e += kThreads) {
    const int r = e / d_bins, d = e - r * d_bins;
    st(g_logits + (row0 + r) * ld_gl + d, s_g[e]);
  }
}

template <typename T>
int fwd_t(const void* logits, long long ld_logits, const void* context, long long ld_ctx, int n_img, int fhw, int d_bins, int c,
          float* depth, float* depth_rows, float* feat, void* stream) {
  const dim3 grid((fhw + kTile - 1) / kTile, n_img);
  const size_t lds = (size_t)kTile * d_bins * sizeof(float);
  hipLaunchKernelGGL((k_depth_head_fwd<T>), grid, dim3(kThreads), lds, (hipStream_t)stream, (const T*)logits, ld_logits,
                     (const T*)context, ld_ctx, d_bins, c, fhw, depth, depth_rows, feat);
  return check_launch("depth_head_fwd");
}

template <typename T>
int bwd_t(const float* depth, const float* g_depth, const float* g_rows, const float* g_feat, int n_img, int fhw, int d_bins,
          int c, void* g_logits, long long ld_gl, void* g_ctx, long long ld_gc, void* stream) {
  const dim3 grid((fhw + kTile - 1) / kTile, n_img);
  const size_t lds = (size_t)2 * kTile * d_bins * sizeof(float);
  hipLaunchKernelGGL((k_depth_head_bwd<T>), grid, dim3(kThreads), lds, (hipStream_t)stream, depth, g_depth, g_rows, g_feat,
                     d_bins, c, fhw, (T*)g_logits, ld_gl, (T*)g_ctx, ld_gc);
  return check_launch("depth_head_bwd");
}

int check_common(int n_img, int fhw, int d_bins, int c, long long ld_a, long long ld_b, bool has_ctx) {
  OMNIHD_REQUIRE(n_img > 0 && n_img <= 65535 && fhw > 0 && d_bins > 0 && d_bins <= 160, "1 <= images <= 65535, 1 <= D <= 160");
  OMNIHD_REQUIRE(ld_a >= d_bins, "logit row pitch smaller than D");
  if (has_ctx) OMNIHD_REQUIRE(c > 0 && c % 4 == 0 && ld_b >= c && ld_b % 4 == 0, "C % 4 == 0 and a context row pitch that is a multiple of 4");
  return OMNIHD_OK;
}

}  // namespace
}  // namespace omnihd

extern "C" int omnihd_depth_head_fwd(const void* logits, long long ld_logits, const void* context, long long ld_ctx,
                                     int is_f32, int n_img, int fhw, int d_bins, int c, float* depth, float* depth_rows,
                                     float* feat, void* stream) {
  using namespace omnihd;
  if (int e = check_common(n_img, fhw, d_bins, c, ld_logits, ld_ctx, context != nullptr)) return e;
  OMNIHD_REQUIRE(logits && depth && (!context || feat), "null pointer");
  if (context) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(context), b = reinterpret_cast<uintptr_t>(feat);
    OMNIHD_REQUIRE((a % (is_f32 ? 16 : 8)) == 0 && (b % 16) == 0, "context rows must be 16-byte (fp32) / 8-byte (bf16) aligned");
  }
  return is_f32 ? fwd_t<float>(logits, ld_logits, context, ld_ctx, n_img, fhw, d_bins, c, depth, depth_rows, feat, stream)
                : fwd_t<bf16_t>(logits, ld_logits, context, ld_ctx, n_img, fhw, d_bins, c, depth, depth_rows, feat, stream);
}

extern "C" int omnihd_depth_head_bwd(const float* depth, const float* g_depth, const float* g_rows, const float* g_feat,
                                     int is_f32, int n_img, int fhw, int d_bins, int c, void* g_logits, long long ld_g_logits,
                                     void* g_context, long long ld_g_context, void* stream) {
  using namespace omnihd;
  if (int e = check_common(n_img, fhw, d_bins, c, ld_g_logits, ld_g_context, g_context != nullptr)) return e;
  OMNIHD_REQUIRE(depth && g_logits && (!g_context || g_feat), "null pointer");
  if (g_context) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(g_context), b = reinterpret_cast<uintptr_t>(g_feat);
    OMNIHD_REQUIRE((a % (is_f32 ? 16 : 8)) == 0 && (b % 16) == 0, "context-gradient rows must be 16-byte (fp32) / 8-byte (bf16) aligned");
  }
  return is_f32 ? bwd_t<float>(depth, g_depth, g_rows, g_feat, n_img, fhw, d_bins, c, g_logits, ld_g_logits, g_context,
                               ld_g_context, stream)
                : bwd_t<bf16_t>(depth, g_depth, g_rows, g_feat, n_img, fhw, d_bins, c, g_logits, ld_g_logits, g_context,
                                ld_g_context, stream);
}
