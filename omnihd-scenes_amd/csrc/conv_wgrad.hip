// Weight gradient of the 3x3 / stride 1 / pad 1 BEV convolutions on the gfx950 matrix cores.
//
// Where it sits: the BEV encoder of the camera stream (4 convs 1024->1024->512->512->256 at
// 160x240, reference bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:201-214) and the
// fusion conv 640->384 (bevf_faster_rcnn_bevdepth.py:61-72) hold ~1.5 of the ~3 TFLOP of a forward
// pass; their weight gradients are the slowest dense kernels of the training step under MIOpen
// (154-205 TFLOP/s measured on MI355X, 9 ms of a 70 ms step; scripts/conv_bench.py).
//
//   dW[n][dy][dx][c] = sum over pixels m=(b,y,x) of  G[m][n] * X[(b, y+dy-1, x+dx-1)][c]
//
// is, per tap, a GEMM whose REDUCTION dimension is the pixel index — the slow dimension of both
// NHWC operands.  So the operands are first re-laid out pixel-contiguous (k_to_kmajor: one pass
// through LDS, which also bakes the three dx shifts of X with their zero borders into three
// copies; a dy shift is then a 16-byte-aligned offset of W pixels), and the GEMM itself is a plain
// "NT" bf16 MFMA kernel: 128x128 output tile per workgroup, 4 wavefronts each owning a 64x64
// quadrant as 2x2 v_mfma_f32_32x32x16_bf16 tiles, K (pixels) stepped by 64 through double-buffered
// LDS (row pitch 144 B: conflict-free ds_read_b128 fragments), register-staged global loads issued
// one step ahead, fp32 accumulation, split-K over pixel ranges into fp32 slabs that a second tiny
// kernel adds in a fixed order (deterministic, no atomics).
#include "common.h"
#include <stdlib.h>

namespace omnihd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBlock = 256;
constexpr int kTile = 128;     // output tile is kTile x kTile
constexpr int kBK = 64;        // pixels per K-step
constexpr int kPitch = 72;     // bf16 elements per LDS row (64 + 8 pad -> 144 B)

// ---------------------------------------------------------------------------------------------
// NHWC (M, C) bf16  ->  n_shifts x (C, Mp) bf16, pixel-contiguous; shift s in {0} or {-1,0,+1}:
//   out[s][c][m] = in[m + dx_s][c]  if 0 <= x(m) + dx_s < W  else 0;   m >= M: 0
// One workgroup: 64 pixels x 64 channels through LDS.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_to_kmajor(const unsigned short* __restrict__ in, int M,
                                                      int C, int W, int Mp, int n_shifts,
                                                      unsigned short* __restrict__ out) {
  __shared__ unsigned short s[64][66 + 2];   // [channel][pixel + halo], +2 pad
  const int m0 = blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  // load pixels m0-1 .. m0+64 (66 of them), 64 channels each: 8 lanes x 16 B per pixel
  for (int i = tid; i < 66 * 8; i += kBlock) {
    const int p = i / 8, oc = i % 8;
    const int m = m0 - 1 + p;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m >= 0 && m < M) v = *reinterpret_cast<const uint4*>(in + (size_t)m * C + c0 + oc * 8);
    const unsigned short* e = reinterpret_cast<const unsigned short*>(&v);
#pragma unroll
    for (int k = 0; k < 8; ++k) s[oc * 8 + k][p] = e[k];
  }
  __syncthreads();
  // write: per (channel, chunk of 8 pixels, shift) one 16 B store
  for (int i = tid; i < 64 * 8 * n_shifts; i += kBlock) {
    const int sh = i / (64 * 8);
    const int c = (i / 8) % 64;
    const int q = i % 8;
    const int dx = (n_shifts == 1) ? 0 : sh - 1;
    unsigned short e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int m = m0 + q * 8 + k;
      const int x = m % W;
      const bool ok = (m < M) && (x + dx >= 0) && (x + dx < W);
      e[k] = ok ? s[c][q * 8 + k + 1 + dx] : (unsigned short)0;
    }
    *reinterpret_cast<uint4*>(out + ((size_t)sh * C + c0 + c) * Mp + m0 + q * 8) =
        *reinterpret_cast<const uint4*>(e);
  }
}

// ---------------------------------------------------------------------------------------------
// slab[split][n][tap][c] = sum over the split's pixels of Gt[n][m] * Xt[dx][c][m + dy*W]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_wgrad_mfma(const unsigned short* __restrict__ Gt,
                                                       const unsigned short* __restrict__ Xt,
                                                       float* __restrict__ slab, int Cout, int Cin,
                                                       int M, int Mp, int H, int W, int n_split,
                                                       int k_per_split) {
  __shared__ __attribute__((aligned(16))) unsigned short sA[2][kTile][kPitch];
  __shared__ __attribute__((aligned(16))) unsigned short sB[2][kTile][kPitch];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int tiles_c = Cin / kTile, tiles_n = Cout / kTile;
  int bid = blockIdx.x;
  const int ct = bid % tiles_c; bid /= tiles_c;
  const int nt = bid % tiles_n; bid /= tiles_n;
  const int tap = bid % 9;
  const int split = bid / 9;
  const int dy = tap / 3 - 1, dx = tap % 3 - 1;
  const int k0 = split * k_per_split;
  const int k1 = min(k0 + k_per_split, Mp);

  // loader mapping: 8 lanes x 16 B cover the 64 pixels of one row; rows r0 + 32*i
  const int lrow = tid >> 3;          // 0..31
  const int lcol = tid & 7;           // chunk of 8 pixels
  const unsigned short* a_src = Gt + (size_t)(nt * kTile + lrow) * Mp + lcol * 8;
  const unsigned short* b_src = Xt + ((size_t)(dx + 1) * Cin + ct * kTile + lrow) * Mp + lcol * 8 + dy * W;
  // image row of this lane's chunk (chunks never straddle rows: W % 8 == 0)
  int px = (k0 + lcol * 8) % W;
  int py = ((k0 + lcol * 8) / W) % H;

  uint4 ra[4], rb[4];
  auto issue = [&](int k) {
    const bool ok = (k + lcol * 8 < M) && (py + dy >= 0) && (py + dy < H);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const uint4*>(a_src + (size_t)(32 * i) * Mp + k);
      rb[i] = *reinterpret_cast<const uint4*>(b_src + (size_t)(32 * i) * Mp + k);   // always in bounds (guards)
    }
    // mask AFTER the loads: a load behind a runtime condition would be branched around and waited for
    // one by one (4 dependent memory round trips per K-step)
    if (!ok) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rb[i] = make_uint4(0, 0, 0, 0);
    }
    px += kBK;
    while (px >= W) { px -= W; py = (py + 1 == H) ? 0 : py + 1; }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(&sA[buf][lrow + 32 * i][lcol * 8]) = ra[i];
      *reinterpret_cast<uint4*>(&sB[buf][lrow + 32 * i][lcol * 8]) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 31;
  const int fk = (lane >> 5) * 8;

  if (k0 < k1) {
    issue(k0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int k = k0; k < k1; k += kBK) {
      const bool more = k + kBK < k1;
      if (more) issue(k + kBK);
#pragma unroll
      for (int ks = 0; ks < kBK / 16; ++ks) {
        bf16x8 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i] = *reinterpret_cast<const bf16x8*>(&sA[buf][wm * 64 + i * 32 + frow][ks * 16 + fk]);
          b[i] = *reinterpret_cast<const bf16x8*>(&sB[buf][wn * 64 + i * 32 + frow][ks * 16 + fk]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if (more) stash(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float* dst = slab + (size_t)split * Cout * 9 * Cin;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nt * kTile + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int c = ct * kTile + wn * 64 + j * 32 + (lane & 31);
        dst[((size_t)n * 9 + tap) * Cin + c] = acc[i][j][r];
      }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant of the NT GEMM.  The register-staged kernel above keeps only 64 KB of operands in
// flight per CU (2 workgroups x one K-step) against a loaded-memory latency of several microseconds
// -> 13 % MFMA utilisation (profiles/round1: SQ_WAIT_ANY 54 %).  Here operands go global -> LDS with
// `global_load_lds` (no VGPR round trip, 16 B per lane, 1 KiB per wave instruction) into a ring of
// kStages K-steps, three of which are in flight while the fourth is multiplied: 96 KB in flight per CU
// with one 4-wave workgroup per CU.  LDS-DMA writes lane-linear images, so rows are unpadded (128 B
// pitch) and the 16-byte chunks of a row are XOR-swizzled by ((row >> 1) & 7) — applied to the GLOBAL
// source chunk a lane fetches and again when a fragment is read: conflict-free ds_read_b128.
// Waits are counted by hand (s_waitcnt vmcnt(16) leaves two K-steps in flight) and the barrier is the
// raw s_barrier: __syncthreads() would drain the DMA queue.
// ---------------------------------------------------------------------------------------------
constexpr int kStages = 4;
typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef __attribute__((address_space(1))) const void gbl_ptr_t;

__global__ __launch_bounds__(kBlock) void k_wgrad_mfma_glds(const unsigned short* __restrict__ Gt,
                                                            const unsigned short* __restrict__ Xt,
                                                            const unsigned short* __restrict__ zero_page,
                                                            float* __restrict__ slab, int Cout, int Cin,
                                                            int M, int Mp, int H, int W, int n_split,
                                                            int k_per_split, int taps) {
  // one LDS object only (a second one makes hipcc drain the DMA queue before every ds_read)
  __shared__ __attribute__((aligned(16))) unsigned short sm[kStages][2][kTile][kBK];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int tiles_c = Cin / kTile, tiles_n = Cout / kTile;
  int bid = blockIdx.x;
  const int ct = bid % tiles_c; bid /= tiles_c;
  const int nt = bid % tiles_n; bid /= tiles_n;
  const int tap = bid % taps;
  const int split = bid / taps;
  const int dy = (taps == 9) ? tap / 3 - 1 : 0, dx = (taps == 9) ? tap % 3 - 1 : 0;
  const int k0 = split * k_per_split;
  const int k1 = min(k0 + k_per_split, Mp);

  // loader: wave w owns rows [32w, 32w+32) of both operands, 4 DMA calls of 8 rows each per operand.
  const int lr = lane >> 3;                       // row inside the 8-row call
  const int pos = lane & 7;                       // 16-byte slot inside the LDS row
  const int c_even = pos ^ ((lane >> 4) & 7);     // global chunk for calls 0,2 ; calls 1,3 use c_even ^ 4
  const unsigned short* a_row = Gt + (size_t)(nt * kTile + wave * 32 + lr) * Mp;
  const unsigned short* b_row = Xt + ((size_t)((taps == 9) ? dx + 1 : 0) * Cin + ct * kTile + wave * 32 + lr) * Mp + dy * W;
  // image row of this lane's two chunk positions at K-step k0 (chunks never straddle rows: W % 8 == 0)
  int px0 = (k0 + c_even * 8) % W, py0 = ((k0 + c_even * 8) / W) % H;
  int px1 = (k0 + (c_even ^ 4) * 8) % W, py1 = ((k0 + (c_even ^ 4) * 8) / W) % H;

  auto issue = [&](int k, int stage) {
    const bool real = k < k1;
    const bool ok0 = real && (k + c_even * 8 < M) && (py0 + dy >= 0) && (py0 + dy < H);
    const bool ok1 = real && (k + (c_even ^ 4) * 8 < M) && (py1 + dy >= 0) && (py1 + dy < H);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = (i & 1) ? (c_even ^ 4) : c_even;
      const bool ok = (i & 1) ? ok1 : ok0;
      const unsigned short* ga = real ? a_row + (size_t)(8 * i) * Mp + k + c * 8 : zero_page;
      const unsigned short* gb = ok ? b_row + (size_t)(8 * i) * Mp + k + c * 8 : zero_page;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t*)ga, (lds_ptr_t*)&sm[stage][0][wave * 32 + 8 * i][0], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t*)gb, (lds_ptr_t*)&sm[stage][1][wave * 32 + 8 * i][0], 16, 0, 0);
    }
    px0 += kBK; while (px0 >= W) { px0 -= W; py0 = (py0 + 1 == H) ? 0 : py0 + 1; }
    px1 += kBK; while (px1 >= W) { px1 -= W; py1 = (py1 + 1 == H) ? 0 : py1 + 1; }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 31;
  const int fhalf = lane >> 5;

  // prologue: three K-steps in flight
  issue(k0, 0);
  issue(k0 + kBK, 1);
  issue(k0 + 2 * kBK, 2);
  int stage = 0;
  for (int k = k0; k < k1; k += kBK) {
    // the oldest K-step (8 DMA calls per wave) has landed when at most 16 younger calls are outstanding
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(k + 3 * kBK, (stage + 3) % kStages);      // its buffer was last read two barriers ago
#pragma unroll
    for (int ks = 0; ks < kBK / 16; ++ks) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + frow;
        const int rb = wn * 64 + i * 32 + frow;
        const int c = ks * 2 + fhalf;
        a[i] = *reinterpret_cast<const bf16x8*>(&sm[stage][0][ra][((c ^ ((ra >> 1) & 7)) * 8)]);
        b[i] = *reinterpret_cast<const bf16x8*>(&sm[stage][1][rb][((c ^ ((rb >> 1) & 7)) * 8)]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    stage = (stage + 1) % kStages;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the dummy tail loads before the epilogue stores

  float* dst = slab + (size_t)split * Cout * taps * Cin;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nt * kTile + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int c = ct * kTile + wn * 64 + j * 32 + (lane & 31);
        dst[((size_t)n * taps + tap) * Cin + c] = acc[i][j][r];
      }
}

__global__ __launch_bounds__(kBlock) void k_sum_slabs(const float* __restrict__ slab, int n_split,
                                                      size_t n, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    float s = slab[i];
    for (int k = 1; k < n_split; ++k) s += slab[(size_t)k * n + i];
    out[i] = s;
  }
}

// Row pitch of the pixel-major operands.  A pitch that is a multiple of 1 KiB (38400 px * 2 B = 75 KiB)
// maps every row of a tile onto the same few L2 channels; an ODD number of 128-byte lines per row
// rotates the rows over all channels.
int padded_pixels(size_t m) {
  size_t mp = (m + kBK - 1) / kBK * kBK;
  static const int extra = [] { const char* e = getenv("OMNIHD_WGRAD_PAD"); return e ? atoi(e) : 1; }();
  if (extra > 0 && (mp / 64) % 2 == 0) mp += 64 * (size_t)extra;
  return (int)mp;
}

int pick_split(int cout, int cin, int mp, int taps = 9) {
  const int tiles = (cout / kTile) * (cin / kTile) * taps;
  int s = (3 * kCUs + tiles - 1) / tiles;          // aim at >= 3 workgroups per CU
  const int max_s = mp / (kBK * 8);                 // at least 8 K-steps per split
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 16) s = 16;
  return s;
}

}  // namespace
}  // namespace omnihd

using namespace omnihd;

extern "C" size_t omnihd_conv3x3_wgrad_workspace_bytes(int batch, int h, int w, int cin, int cout) {
  if (batch <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return 256;
  const size_t M = (size_t)batch * h * w;
  const size_t Mp = padded_pixels(M);
  const int S = pick_split(cout, cin, (int)Mp);
  // [pad row of W pixels] Gt [Cout][Mp] | guard | Xt [3][Cin][Mp] | guard | slabs
  return 256 + align_up((size_t)cout * Mp * 2, 256) + align_up((size_t)3 * cin * Mp * 2, 256) +
         align_up((size_t)S * cout * 9 * cin * 4, 256) + 4 * align_up((size_t)w * 2 + 256, 256);
}

extern "C" int omnihd_conv3x3_wgrad_bf16(const void* x_nhwc, const void* gout_nhwc, float* dw,
                                         int batch, int h, int w, int cin, int cout,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  OMNIHD_REQUIRE(batch > 0 && h > 0 && w > 0, "shape");
  OMNIHD_REQUIRE(cin % kTile == 0 && cout % kTile == 0, "Cin and Cout must be multiples of 128");
  OMNIHD_REQUIRE(w % 8 == 0, "W must be a multiple of 8");
  OMNIHD_REQUIRE(x_nhwc && gout_nhwc && dw && workspace, "null pointer");
  const size_t need = omnihd_conv3x3_wgrad_workspace_bytes(batch, h, w, cin, cout);
  if (workspace_bytes < need) {
    set_error("conv3x3_wgrad: workspace %zu < required %zu", workspace_bytes, need);
    return OMNIHD_ERR_WORKSPACE;
  }
  const int M = batch * h * w;
  const int Mp = padded_pixels(M);
  const int S = pick_split(cout, cin, Mp);
  int k_per_split = ((Mp / kBK + S - 1) / S) * kBK;
  const size_t guard = align_up((size_t)w * 2 + 256, 256);   // a dy = -1 read at k = 0 lands here, masked anyway
  unsigned short* zero_page = static_cast<unsigned short*>(workspace);       // 256 zero bytes
  OMNIHD_HIP_TRY(hipMemsetAsync(zero_page, 0, 256, st));
  char* p = static_cast<char*>(workspace) + 256 + guard;
  unsigned short* Gt = reinterpret_cast<unsigned short*>(p);
  p += align_up((size_t)cout * Mp * 2, 256) + guard;
  unsigned short* Xt = reinterpret_cast<unsigned short*>(p);
  p += align_up((size_t)3 * cin * Mp * 2, 256) + guard;
  float* slab = reinterpret_cast<float*>(p);

  const dim3 gG((Mp + 63) / 64, cout / 64), gX((Mp + 63) / 64, cin / 64);
  hipLaunchKernelGGL(k_to_kmajor, gG, dim3(kBlock), 0, st, static_cast<const unsigned short*>(gout_nhwc), M,
                     cout, w, Mp, 1, Gt);
  hipLaunchKernelGGL(k_to_kmajor, gX, dim3(kBlock), 0, st, static_cast<const unsigned short*>(x_nhwc), M, cin,
                     w, Mp, 3, Xt);
  const int blocks = (cin / kTile) * (cout / kTile) * 9 * S;
  static const int use_glds = [] { const char* e = getenv("OMNIHD_WGRAD_GLDS"); return e ? atoi(e) : 1; }();
  if (use_glds)
    hipLaunchKernelGGL(k_wgrad_mfma_glds, dim3(blocks), dim3(kBlock), 0, st, Gt, Xt, zero_page, S > 1 ? slab : dw,
                       cout, cin, M, Mp, h, w, S, k_per_split, 9);
  else
    hipLaunchKernelGGL(k_wgrad_mfma, dim3(blocks), dim3(kBlock), 0, st, Gt, Xt, S > 1 ? slab : dw, cout, cin, M,
                       Mp, h, w, S, k_per_split);
  if (S > 1) {
    const size_t n = (size_t)cout * 9 * cin;
    hipLaunchKernelGGL(k_sum_slabs, dim3(grid_for((int64_t)n, kBlock * 4)), dim3(kBlock), 0, st, slab, S, n, dw);
  }
  return check_launch("conv3x3_wgrad_bf16");
}

// 1x1 convolution (taps = 1): dW[n][c] = sum_m G[m][n] * X[m][c]; x and gout are [m, c] / [m, cout] bf16 rows.
extern "C" size_t omnihd_conv1x1_wgrad_workspace_bytes(int m, int cin, int cout) {
  if (m <= 0 || cin <= 0 || cout <= 0) return 256;
  const size_t Mp = padded_pixels((size_t)m);
  const int S = pick_split(cout, cin, (int)Mp, 1);
  return 256 + align_up((size_t)cout * Mp * 2, 256) + align_up((size_t)cin * Mp * 2, 256) +
         align_up((size_t)S * cout * cin * 4, 256) + 4 * 1024;
}

extern "C" int omnihd_conv1x1_wgrad_bf16(const void* x_rows, const void* gout_rows, float* dw, int m, int cin,
                                         int cout, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  OMNIHD_REQUIRE(m > 0 && cin % kTile == 0 && cout % kTile == 0, "Cin and Cout must be multiples of 128");
  OMNIHD_REQUIRE(x_rows && gout_rows && dw && workspace, "null pointer");
  if (workspace_bytes < omnihd_conv1x1_wgrad_workspace_bytes(m, cin, cout)) {
    set_error("conv1x1_wgrad: workspace too small");
    return OMNIHD_ERR_WORKSPACE;
  }
  const int Mp = padded_pixels((size_t)m);
  const int S = pick_split(cout, cin, Mp, 1);
  const int k_per_split = ((Mp / kBK + S - 1) / S) * kBK;
  unsigned short* zero_page = static_cast<unsigned short*>(workspace);
  OMNIHD_HIP_TRY(hipMemsetAsync(zero_page, 0, 256, st));
  char* p = static_cast<char*>(workspace) + 256 + 1024;
  unsigned short* Gt = reinterpret_cast<unsigned short*>(p);
  p += align_up((size_t)cout * Mp * 2, 256) + 1024;
  unsigned short* Xt = reinterpret_cast<unsigned short*>(p);
  p += align_up((size_t)cin * Mp * 2, 256) + 1024;
  float* slab = reinterpret_cast<float*>(p);
  const dim3 gG(Mp / 64, cout / 64), gX(Mp / 64, cin / 64);
  // W = Mp: no image-row wrap, so the (single) dx = 0 shift never masks anything
  hipLaunchKernelGGL(k_to_kmajor, gG, dim3(kBlock), 0, st, static_cast<const unsigned short*>(gout_rows), m, cout, Mp, Mp, 1, Gt);
  hipLaunchKernelGGL(k_to_kmajor, gX, dim3(kBlock), 0, st, static_cast<const unsigned short*>(x_rows), m, cin, Mp, Mp, 1, Xt);
  const int blocks = (cin / kTile) * (cout / kTile) * S;
  hipLaunchKernelGGL(k_wgrad_mfma_glds, dim3(blocks), dim3(kBlock), 0, st, Gt, Xt, zero_page, S > 1 ? slab : dw, cout, cin,
                     m, Mp, 1 << 30, 64, S, k_per_split, 1);
  if (S > 1) {
    const size_t n = (size_t)cout * cin;
    hipLaunchKernelGGL(k_sum_slabs, dim3(grid_for((int64_t)n, kBlock * 4)), dim3(kBlock), 0, st, slab, S, n, dw);
  }
  return check_launch("conv1x1_wgrad_bf16");
}
