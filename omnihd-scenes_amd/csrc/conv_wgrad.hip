// Weight gradient of the dense convolutions on the gfx950 matrix cores.
//
// Where it sits: the BEV encoder of the camera stream (4 convs 1024->1024->512->512->256 at
// 160x240, reference bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:201-214) and the
// fusion conv 640->384 (bevf_faster_rcnn_bevdepth.py:61-72) hold ~1.5 of the ~3 TFLOP of a forward
// pass; their weight gradients were the slowest dense kernels of the training step under MIOpen
// (154-205 TFLOP/s measured on MI355X, 9 ms of a 70 ms step; scripts/conv_bench.py).  The same kernel
// now serves every 1x1 / 3x3 convolution of the detector (image backbone, FPN, DepthNet incl. the
// dilated ASPP branches, SECOND incl. its stride-2 entries, SECONDFPN, head).
//
//   dW[n][ky][kx][c] = sum over output pixels m=(b,y,x) of  G[m][n] * X[(b, y*s+ky*d-p, x*s+kx*d-p)][c]
//
// is, per tap, a GEMM whose REDUCTION dimension is the pixel index — the slow dimension of both
// NHWC operands.  So the operands are first re-laid out pixel-contiguous ("k-major") in one pass through
// LDS.  For stride-1 3x3 convolutions that pass also bakes the three dx shifts of X with their zero
// borders into three copies (a dy shift is then a 16-byte-aligned offset of dil*Wp pixels, Wp = the row
// pitch padded to a multiple of 8); strided convolutions get one staged copy per tap.  The GEMM itself
// is an "NT" bf16 MFMA kernel: 128x128 output tile per workgroup, 4 wavefronts each owning a 64x64
// quadrant as 2x2 v_mfma_f32_32x32x16_bf16 tiles, K (pixels) stepped by 64 through a 4-stage LDS ring
// filled by LDS-DMA, fp32 accumulation, split-K over pixel ranges into fp32 slabs that a second tiny
// kernel adds in a fixed order (deterministic, no atomics).  Channel counts that are not multiples of
// 128 are zero-padded in the staged operands.
#include "common.h"
#include <stdlib.h>

namespace omnihd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBlock = 256;
constexpr int kTile = 128;     // output tile is kTile x kTile
constexpr int kBK = 64;        // pixels per K-step

// ---------------------------------------------------------------------------------------------
// NHWC (M, C) bf16  ->  n_shifts x (Cp, Mp) bf16, pixel-contiguous ("k-major").
// The pixel axis of the output is the PADDED image raster m' = (b*H + y) * Wp + x with Wp = W rounded up
// to a multiple of 8 (so that a row shift is a 16-byte-aligned offset and an 8-pixel chunk never
// straddles image rows); pad columns, pixels past the last image and channel rows c >= C are zero.
// shift s in {0} or {-dil, 0, +dil}:
//   out[s][c][m'] = in[(row(m'), x(m') + dx_s)][c]   if x(m') < W and 0 <= x(m') + dx_s < W   else 0
// One workgroup: 64 padded pixels x 64 channels through LDS (halo of `dil` pixels each side).
// ---------------------------------------------------------------------------------------------
constexpr int kMaxDil = 18;

struct StageArgs {
  const unsigned short* in;
  unsigned short* out;
  int C, Cp, n_shifts;
};

__device__ __forceinline__ void stage_kmajor(const unsigned short* __restrict__ in, int M, int C, int Cp, int W, int Wp,
                                             int Mp, int n_shifts, int dil, unsigned short* __restrict__ out,
                                             unsigned short (*s)[64 + 2 * kMaxDil + 2]) {
  const int m0 = blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  if (c0 >= Cp) return;
  const int tid = threadIdx.x;
  const int halo = (n_shifts == 1) ? 0 : dil;
  const int span = 64 + 2 * halo;
  const int rows = M / W;                                   // image rows over the whole batch
  // load padded pixels m0-halo .. m0+63+halo, 64 channels each: 8 lanes x 16 B per pixel
  for (int i = tid; i < span * 8; i += kBlock) {
    const int p = i / 8, oc = i % 8;
    const int mp = m0 - halo + p;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (mp >= 0 && c0 + oc * 8 < C) {
      const int row = mp / Wp, x = mp - row * Wp;
      if (row < rows && x < W) v = *reinterpret_cast<const uint4*>(in + ((size_t)row * W + x) * C + c0 + oc * 8);
    }
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
    const int dx = (n_shifts == 1) ? 0 : (sh - 1) * dil;
    unsigned short e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int mp = m0 + q * 8 + k;
      const int x = mp % Wp;
      const bool ok = (x < W) && (x + dx >= 0) && (x + dx < W);   // rows past the batch were loaded as zeros
      e[k] = ok ? s[c][q * 8 + k + halo + dx] : (unsigned short)0;
    }
    *reinterpret_cast<uint4*>(out + ((size_t)sh * Cp + c0 + c) * Mp + m0 + q * 8) =
        *reinterpret_cast<const uint4*>(e);
  }
}

// One launch stages BOTH operands of a stride-1 convolution: blockIdx.z = 0 -> output gradient (one copy),
// 1 -> input activations (n_shifts copies).  grid.y covers the wider of the two channel counts.
__global__ __launch_bounds__(kBlock) void k_to_kmajor(StageArgs a0, StageArgs a1, int M, int W, int Wp, int Mp, int dil) {
  __shared__ unsigned short s[64][64 + 2 * kMaxDil + 2];   // [channel][pixel + halo]
  const StageArgs& a = blockIdx.z == 0 ? a0 : a1;
  stage_kmajor(a.in, M, a.C, a.Cp, W, Wp, Mp, a.n_shifts, dil, a.out, s);
}

// ---------------------------------------------------------------------------------------------
// Per-tap staging for strided convolutions (im2col in k-major form): one copy per tap t = (ky, kx),
//   out[t][c][m] = in[(b, oy*stride + ky*dil - pad, ox*stride + kx*dil - pad)][c]   or 0 outside,
// m = (b*Ho + oy)*Wo + ox the OUTPUT raster (the reduction axis of the weight gradient).
// One workgroup: 64 output pixels x 64 channels of one tap; each pixel's 64 channels are one 128-byte read.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_taps_kmajor(const unsigned short* __restrict__ in, int B, int H, int W,
                                                        int C, int Cp, int Ho, int Wo, int Mp, int KW, int stride,
                                                        int pad, int dil, unsigned short* __restrict__ out) {
  __shared__ unsigned short s[64][64 + 2];
  const int m0 = blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tap = blockIdx.z;
  const int ky = tap / KW, kx = tap % KW;
  const int tid = threadIdx.x;
  const int M = B * Ho * Wo;
  for (int i = tid; i < 64 * 8; i += kBlock) {
    const int p = i / 8, oc = i % 8;
    const int m = m0 + p;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m < M && c0 + oc * 8 < C) {
      const int ox = m % Wo, t = m / Wo, oy = t % Ho, b = t / Ho;
      const int iy = oy * stride + ky * dil - pad, ix = ox * stride + kx * dil - pad;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W)
        v = *reinterpret_cast<const uint4*>(in + (((size_t)b * H + iy) * W + ix) * C + c0 + oc * 8);
    }
    const unsigned short* e = reinterpret_cast<const unsigned short*>(&v);
#pragma unroll
    for (int k = 0; k < 8; ++k) s[oc * 8 + k][p] = e[k];
  }
  __syncthreads();
  for (int i = tid; i < 64 * 8; i += kBlock) {
    const int c = i / 8, q = i % 8;
    unsigned short e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = s[c][q * 8 + k];
    *reinterpret_cast<uint4*>(out + ((size_t)tap * Cp + c0 + c) * Mp + m0 + q * 8) = *reinterpret_cast<const uint4*>(e);
  }
}

// ---------------------------------------------------------------------------------------------
// slab[split][n][tap][c] = sum over the split's pixels of Gt[n][m] * Xt[copy(tap)][c][m + off(tap)]
//
// "NT" GEMM on the matrix cores with LDS-DMA operand staging.  A first, register-staged version of this
// kernel kept only 64 KB of operands in flight per CU against a loaded-memory latency of several
// microseconds -> 13 % MFMA utilisation (profiles/round1: SQ_WAIT_ANY 54 %).  Here operands go
// global -> LDS with `global_load_lds` (no VGPR round trip, 16 B per lane, 1 KiB per wave instruction)
// into a ring of kStages K-steps, three of which are in flight while the fourth is multiplied: 96 KB in
// flight per CU with one 4-wave workgroup per CU.  LDS-DMA writes lane-linear images, so rows are
// unpadded (128 B pitch) and the 16-byte chunks of a row are XOR-swizzled by ((row >> 1) & 7) — applied
// to the GLOBAL source chunk a lane fetches and again when a fragment is read: conflict-free
// ds_read_b128.  Waits are counted by hand (s_waitcnt vmcnt(16) leaves two K-steps in flight) and the
// barrier is the raw s_barrier: __syncthreads() would drain the DMA queue.
//
// mode 0: taps share three dx-shifted copies of X; tap (ky,kx) reads copy kx at pixel offset
//         (ky-1)*dil*Wp and masks image rows whose source row is outside [0,H)   (taps = 9 or 1)
// mode 1: one staged copy per tap, nothing to mask                                (any taps)
// Channel counts are padded to multiples of 128 in the staged operands (zero rows); only the real
// [Cout][taps][Cin] block is written.
// ---------------------------------------------------------------------------------------------
constexpr int kStages = 4;
typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef __attribute__((address_space(1))) const void gbl_ptr_t;

__global__ __launch_bounds__(kBlock) void k_wgrad_mfma_glds(const unsigned short* __restrict__ Gt,
                                                            const unsigned short* __restrict__ Xt,
                                                            const unsigned short* __restrict__ zero_page,
                                                            float* __restrict__ slab, int Cout, int Cin,
                                                            int Coutp, int Cinp, int M, int Mp, int H, int W,
                                                            int n_split, int k_per_split, int taps, int mode,
                                                            int dil, const unsigned short* __restrict__ Gt2,
                                                            const unsigned short* __restrict__ Xt2, int n_terms) {
  // n_terms == 3 (fp32-grade split operands): Gt / Xt are the hi planes, Gt2 / Xt2 the lo planes, and the K loop runs three
  // times over the split's pixel range, accumulating  G_hi*X_hi + G_hi*X_lo + G_lo*X_hi  into the same tile
  // one LDS object only (a second one makes hipcc drain the DMA queue before every ds_read)
  __shared__ __attribute__((aligned(16))) unsigned short sm[kStages][2][kTile][kBK];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int tiles_c = Cinp / kTile, tiles_n = Coutp / kTile;
  int bid = blockIdx.x;
  const int ct = bid % tiles_c; bid /= tiles_c;
  const int nt = bid % tiles_n; bid /= tiles_n;
  const int tap = bid % taps;
  const int split = bid / taps;
  const bool shared = (mode == 0) && (taps == 9);
  const int dy = shared ? (tap / 3 - 1) * dil : 0;
  const int copy = shared ? tap % 3 : ((mode == 1) ? tap : 0);
  const int k0 = split * k_per_split;
  const int k1 = min(k0 + k_per_split, Mp);

  // loader: wave w owns rows [32w, 32w+32) of both operands, 4 DMA calls of 8 rows each per operand.
  const int lr = lane >> 3;                       // row inside the 8-row call
  const int pos = lane & 7;                       // 16-byte slot inside the LDS row
  const int c_even = pos ^ ((lane >> 4) & 7);     // global chunk for calls 0,2 ; calls 1,3 use c_even ^ 4
  const size_t a_off = (size_t)(nt * kTile + wave * 32 + lr) * Mp;
  const ptrdiff_t b_off = (ptrdiff_t)(((size_t)copy * Cinp + ct * kTile + wave * 32 + lr) * Mp) + (ptrdiff_t)dy * W;
  const unsigned short* a_row = Gt + a_off;
  const unsigned short* b_row = Xt + b_off;
  // image row of this lane's two chunk positions at K-step k0 (chunks never straddle rows: W % 8 == 0)
  const int px0_0 = (k0 + c_even * 8) % W, py0_0 = ((k0 + c_even * 8) / W) % H;
  const int px1_0 = (k0 + (c_even ^ 4) * 8) % W, py1_0 = ((k0 + (c_even ^ 4) * 8) / W) % H;
  int px0 = px0_0, py0 = py0_0, px1 = px1_0, py1 = py1_0;
  const int n_it = (k1 - k0 + kBK - 1) / kBK;      // K-steps per term
  const int total = n_it * n_terms;
  int it_issue = 0, it_in_term = 0, term = 0;

  // issues the next K-step in (term, pixel) order; beyond the last one: zero-page dummies (keeps the wait counts uniform)
  auto issue = [&](int stage) {
    const bool real = it_issue < total;
    const int k = k0 + it_in_term * kBK;
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
    ++it_issue;
    if (++it_in_term == n_it && it_issue < total) {   // next term: rewind the pixel range, switch operand planes
      it_in_term = 0; ++term;
      px0 = px0_0; py0 = py0_0; px1 = px1_0; py1 = py1_0;
      a_row = (term == 2 ? Gt2 : Gt) + a_off;
      b_row = (term == 1 ? Xt2 : Xt) + b_off;
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
  const int fhalf = lane >> 5;

  // prologue: three K-steps in flight
  issue(0);
  issue(1);
  issue(2);
  int stage = 0;
  for (int it = 0; it < total; ++it) {
    // the oldest K-step (8 DMA calls per wave) has landed when at most 16 younger calls are outstanding
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue((stage + 3) % kStages);                   // its buffer was last read two barriers ago
    // all fragments of the K-step are requested before its first MFMA (see k_wgrad_mfma_glds3)
    bf16x8 a[kBK / 16][2], b[kBK / 16][2];
#pragma unroll
    for (int ks = 0; ks < kBK / 16; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + frow;
        const int rb = wn * 64 + i * 32 + frow;
        const int c = ks * 2 + fhalf;
        a[ks][i] = *reinterpret_cast<const bf16x8*>(&sm[stage][0][ra][((c ^ ((ra >> 1) & 7)) * 8)]);
        b[ks][i] = *reinterpret_cast<const bf16x8*>(&sm[stage][1][rb][((c ^ ((rb >> 1) & 7)) * 8)]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < kBK / 16; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
    stage = (stage + 1) % kStages;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the dummy tail loads before the epilogue stores

  // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float* dst = slab + (size_t)split * Cout * taps * Cin;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nt * kTile + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int c = ct * kTile + wn * 64 + j * 32 + (lane & 31);
        if (n < Cout && c < Cin) dst[((size_t)n * taps + tap) * Cin + c] = acc[i][j][r];
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

// ---------------------------------------------------------------------------------------------
// Three taps per workgroup (3x3, stride 1, pad == dil): the taps (ky, 0..2) of one kernel row multiply the SAME G tile with
// the three dx-shifted copies of X at the same row offset, so G is streamed once per three taps.  The one-tap kernel keeps
// 96 KB of operands in flight per CU and is bound by that (12 TB/s out of the L2s at ~2 us of loaded latency = what 24 MB
// in flight can deliver): the same bytes in flight now feed 1.5x the MFMAs (4 operand tiles per 3 tile products instead
// of 6).  K-step of 32 pixels (64-byte LDS rows, 16-byte chunks XOR-swizzled by ((row >> 2) & 3): conflict-free
// ds_read_b128), ring of 4 stages x 4 tiles x 8 KB = 128 KB, one 4-wave workgroup per CU, 192 accumulator registers.
// Per wave and stage: 2 DMA calls for G + 3 x 2 for X = 8, so vmcnt(16) leaves two younger stages in flight.
// ---------------------------------------------------------------------------------------------
constexpr int kBK3 = 32;
constexpr int kStages3 = 4;

__global__ __launch_bounds__(kBlock) void k_wgrad_mfma_glds3(const unsigned short* __restrict__ Gt,
                                                             const unsigned short* __restrict__ Xt,
                                                             const unsigned short* __restrict__ zero_page,
                                                             float* __restrict__ slab, int Cout, int Cin, int Coutp,
                                                             int Cinp, int M, int Mp, int H, int W, int n_split,
                                                             int k_per_split, int dil, const unsigned short* __restrict__ Gt2,
                                                             const unsigned short* __restrict__ Xt2, int n_terms) {
  // n_terms == 3: split operands, three passes over the pixel range (see k_wgrad_mfma_glds)
  __shared__ __attribute__((aligned(16))) unsigned short sm[kStages3][4][kTile][kBK3];   // tile 0 = G, 1..3 = X copies
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int tiles_c = Cinp / kTile, tiles_n = Coutp / kTile;
  int bid = blockIdx.x;
  const int ct = bid % tiles_c; bid /= tiles_c;
  const int nt = bid % tiles_n; bid /= tiles_n;
  const int ky = bid % 3;
  const int split = bid / 3;
  const int dy = (ky - 1) * dil;
  const int k0 = split * k_per_split;
  const int k1 = min(k0 + k_per_split, Mp);

  // loader: one DMA call = 16 rows of 64 B; wave w owns rows [32w, 32w+32) of every tile: two calls per tile
  const int lr = lane >> 2;                        // row inside the call
  const int gch = (lane & 3) ^ ((lane >> 4) & 3);  // global 16-byte chunk this lane fetches (LDS slot lane & 3)
  const size_t a_off = (size_t)(nt * kTile + wave * 32 + lr) * Mp;
  const ptrdiff_t b_off = (ptrdiff_t)(((size_t)ct * kTile + wave * 32 + lr) * Mp) + (ptrdiff_t)dy * W;
  const unsigned short* a_row = Gt + a_off;
  const unsigned short* b_row = Xt + b_off;
  const size_t copy_pitch = (size_t)Cinp * Mp;
  const int px_0 = (k0 + gch * 8) % W, py_0 = ((k0 + gch * 8) / W) % H;
  int px = px_0, py = py_0;
  const int n_it = (k1 - k0 + kBK3 - 1) / kBK3;
  const int total = n_it * n_terms;
  int it_issue = 0, it_in_term = 0, term = 0;

  auto issue = [&](int stage) {
    const bool real = it_issue < total;
    const int k = k0 + it_in_term * kBK3;
    const bool ok = real && (k + gch * 8 < M) && (py + dy >= 0) && (py + dy < H);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned short* ga = real ? a_row + (size_t)(16 * i) * Mp + k + gch * 8 : zero_page;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t*)ga, (lds_ptr_t*)&sm[stage][0][wave * 32 + 16 * i][0], 16, 0, 0);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const unsigned short* gb = ok ? b_row + c * copy_pitch + (size_t)(16 * i) * Mp + k + gch * 8 : zero_page;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t*)gb, (lds_ptr_t*)&sm[stage][1 + c][wave * 32 + 16 * i][0], 16, 0, 0);
      }
    }
    px += kBK3; while (px >= W) { px -= W; py = (py + 1 == H) ? 0 : py + 1; }
    ++it_issue;
    if (++it_in_term == n_it && it_issue < total) {
      it_in_term = 0; ++term;
      px = px_0; py = py_0;
      a_row = (term == 2 ? Gt2 : Gt) + a_off;
      b_row = (term == 1 ? Xt2 : Xt) + b_off;
    }
  };

  f32x16 acc[3][2][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 31;
  const int fhalf = lane >> 5;

  issue(0);
  issue(1);
  issue(2);
  int stage = 0;
  for (int it = 0; it < total; ++it) {
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue((stage + 3) % kStages3);
    // both 16-deep slices are requested before the first MFMA and slice 1's reads are in flight while slice 0 multiplies
    // (hipcc's own schedule interleaved reads and MFMAs with full lgkmcnt(0) waits)
    bf16x8 a[2][2], b[2][3][2];
#pragma unroll
    for (int ks = 0; ks < kBK3 / 16; ++ks) {
      const int c = ks * 2 + fhalf;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + frow;
        const int rb = wn * 64 + i * 32 + frow;
        a[ks][i] = *reinterpret_cast<const bf16x8*>(&sm[stage][0][ra][(c ^ ((ra >> 2) & 3)) * 8]);
#pragma unroll
        for (int t = 0; t < 3; ++t)
          b[ks][t][i] = *reinterpret_cast<const bf16x8*>(&sm[stage][1 + t][rb][(c ^ ((rb >> 2) & 3)) * 8]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < kBK3 / 16; ++ks) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks][t][j], acc[t][i][j], 0, 0, 0);
    }
    stage = (stage + 1) % kStages3;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  float* dst = slab + (size_t)split * Cout * 9 * Cin;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int tap = ky * 3 + t;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = nt * kTile + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int cc = ct * kTile + wn * 64 + j * 32 + (lane & 31);
          if (n < Cout && cc < Cin) dst[((size_t)n * 9 + tap) * Cin + cc] = acc[t][i][j][r];
        }
  }
}

// ---------------------------------------------------------------------------------------------
// Three taps per workgroup on SPLIT operands (fp32-grade weight gradient of the reference-precision step): per K-step of
// 16 pixels one stage holds G_hi, G_lo and the three dx-shifted copies of X_hi and X_lo (8 tiles of 128 rows x 32 B = 32 KB,
// the ring of 4 stages is the same 128 KB) and feeds  3 taps x 3 terms x 4 = 36 MFMAs  from 16 fragment reads:
//     dW[tap] += G_hi*X_hi[tap] + G_hi*X_lo[tap] + G_lo*X_hi[tap].
// Running k_wgrad_mfma_glds3 three times over the pixel range (one launch, n_terms = 3) streams 12 operand tiles through LDS
// for the same 9 tile products and measured 2.97 ms on 1024->1024 at 160x240; here it is 8.
// 32-byte LDS rows: slot = 16-byte chunk ^ ((row >> 3) & 1)  (conflict-free ds_read_b128, checked per 16-lane group).
// Per wave and stage: 8 DMA calls (one per tile), so vmcnt(16) leaves two younger stages in flight.
// ---------------------------------------------------------------------------------------------
constexpr int kBKS = 16;

__global__ __launch_bounds__(kBlock) void k_wgrad_split3(const unsigned short* __restrict__ Gt, const unsigned short* __restrict__ Gt2,
                                                         const unsigned short* __restrict__ Xt, const unsigned short* __restrict__ Xt2,
                                                         const unsigned short* __restrict__ ws_base, unsigned ws_bytes,
                                                         float* __restrict__ slab, int Cout, int Cin, int Coutp, int Cinp, int M,
                                                         int Mp, int H, int W, int n_split, int k_per_split, int dil) {
  // tiles: 0 = G_hi, 1 = G_lo, 2..4 = X_hi copies, 5..7 = X_lo copies
  __shared__ __attribute__((aligned(16))) unsigned short sm[kStages3][8][kTile][kBKS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_c = Cinp / kTile, tiles_n = Coutp / kTile;
  // (integer division runs on the vector ALU: without readfirstlane the tile indices, and every K-step counter derived from
  // them, live in VGPRs and each LDS-DMA call below turns into a waterfall loop over its "divergent" scalar offset)
  int bid = blockIdx.x;
  const int ct = __builtin_amdgcn_readfirstlane(bid % tiles_c); bid /= tiles_c;
  const int nt = __builtin_amdgcn_readfirstlane(bid % tiles_n); bid /= tiles_n;
  const int ky = __builtin_amdgcn_readfirstlane(bid % 3);
  const int split = __builtin_amdgcn_readfirstlane(bid / 3);
  const int dy = (ky - 1) * dil;
  const int k0 = split * k_per_split;
  const int k1 = min(k0 + k_per_split, Mp);
  const int k_last = k1 - kBKS;

  // loader: one DMA call = 32 rows of 32 B; wave w owns rows [32w, 32w+32) of every tile: one call per tile.  The calls
  // are `buffer_load_dwordx4 ... offen lds` through ONE range-checked descriptor over the staging workspace (all four
  // planes live in it): the lane's offset is static, the K-step position travels in the scalar offset, and an X row outside
  // the image is an offset beyond the buffer (the hardware writes zeros) — no zero page, no 64-bit address arithmetic.
  constexpr unsigned kOOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ws_base, 0, (int)ws_bytes, 0x00020000);
  const int lr = lane >> 1;                          // row inside the call
  const int gch = (lane & 1) ^ ((lr >> 3) & 1);      // global 16-byte chunk this lane fetches (LDS slot lane & 1)
  const size_t a_row = (size_t)(nt * kTile + wave * 32 + lr) * Mp;
  const ptrdiff_t b_row = (ptrdiff_t)(((size_t)ct * kTile + wave * 32 + lr) * Mp) + (ptrdiff_t)dy * W;   // may reach into the guard
  const unsigned g_off0 = (unsigned)((const char*)(Gt + a_row) - (const char*)ws_base) + gch * 16;
  const unsigned g_off1 = (unsigned)((const char*)(Gt2 + a_row) - (const char*)ws_base) + gch * 16;
  const unsigned x_off0 = (unsigned)((const char*)(Xt + b_row) - (const char*)ws_base) + gch * 16;
  const unsigned x_off1 = (unsigned)((const char*)(Xt2 + b_row) - (const char*)ws_base) + gch * 16;
  const unsigned copy_pitch = (unsigned)((size_t)Cinp * Mp * 2);
  int px = (k0 + gch * 8) % W, py = ((k0 + gch * 8) / W) % H;
  int k_issue = k0;

  // call c (0..7) of the fill of the next K-step in issue order; the dummy fills behind the last K-step re-read it (their
  // ring slot is never read again; the wait counts stay uniform)
  unsigned xm0 = 0, xm1 = 0;
  auto fill_begin = [&]() {
    const int kk = min(k_issue, k_last);
    const bool ok = (kk + gch * 8 < M) && (py + dy >= 0) && (py + dy < H);
    xm0 = ok ? x_off0 : kOOB;
    xm1 = ok ? x_off1 : kOOB;
  };
  auto fill_call = [&](int stage, int c) {
    const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(min(k_issue, k_last) * 2);   // scalar offset: the K-step position
    if (c == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t*)&sm[stage][0][wave * 32][0], 16, g_off0, so, 0, 0);
    else if (c == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t*)&sm[stage][1][wave * 32][0], 16, g_off1, so, 0, 0);
    else if (c < 5) {
      const unsigned v = xm0 + (unsigned)(c - 2) * copy_pitch;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t*)&sm[stage][c][wave * 32][0], 16, v, so, 0, 0);
    } else {
      const unsigned v = xm1 + (unsigned)(c - 5) * copy_pitch;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t*)&sm[stage][c][wave * 32][0], 16, v, so, 0, 0);
    }
  };
  auto fill_end = [&]() {
    k_issue += kBKS;
    px += kBKS; while (px >= W) { px -= W; py = (py + 1 == H) ? 0 : py + 1; }
  };
  auto issue = [&](int stage) {
    fill_begin();
#pragma unroll
    for (int c = 0; c < 8; ++c) fill_call(stage, c);
    fill_end();
  };

  f32x16 acc[3][2][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 31;
  const int fhalf = lane >> 5;

  // Software pipeline over the K-steps (one wavefront per SIMD has nobody to hide its LDS latency behind).  A step is three
  // groups of 12 MFMAs:  A = G_hi*X_hi,  B = G_lo*X_hi,  C = G_hi*X_lo.  The fragment reads run ONE GROUP ahead of their use:
  //     during A(k): G_lo(k)          during B(k): X_lo(k)          during C(k): G_hi(k+1), X_hi(k+1)
  // so that only G_hi needs a second register set (72 fragment registers instead of 128 for whole-step double buffering, which
  // spilled).  Ring: in front of C(k) one wait + barrier makes stage k+1 visible (vmcnt(16): the two younger stages may be in
  // flight) and proves that every wave has read stage k (lgkmcnt(0)), whose slot then takes the 8 fill calls of stage k+4.
  bf16x8 ah0[2], ah1[2], al[2], bh[3][2], bl[3][2];
  int stage = 0;                                      // ring slot of step k
  const int ra0 = wm * 64 + frow, rb0 = wn * 64 + frow;
  auto slot_of = [&](int r) { return (fhalf ^ ((r >> 3) & 1)) * 8; };
  auto read_g = [&](bf16x8 (&g)[2], int st, int tile) {
#pragma unroll
    for (int i = 0; i < 2; ++i) g[i] = *reinterpret_cast<const bf16x8*>(&sm[st][tile][ra0 + i * 32][slot_of(ra0 + i * 32)]);
  };
  auto read_x = [&](bf16x8 (&x)[3][2], int st, int tile0) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i) x[t][i] = *reinterpret_cast<const bf16x8*>(&sm[st][tile0 + t][rb0 + i * 32][slot_of(rb0 + i * 32)]);
  };
  auto mma12 = [&](const bf16x8 (&g)[2], const bf16x8 (&x)[3][2]) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[i], x[t][j], acc[t][i][j], 0, 0, 0);
  };
  auto step = [&](const bf16x8 (&ah_cur)[2], bf16x8 (&ah_nxt)[2]) {
    __builtin_amdgcn_sched_barrier(0);
    read_g(al, stage, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma12(ah_cur, bh);                                // A
    __builtin_amdgcn_sched_barrier(0);
    read_x(bl, stage, 5);
    __builtin_amdgcn_sched_barrier(0);
    mma12(al, bh);                                    // B
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int nxt = (stage + 1) % kStages3;
    read_g(ah_nxt, nxt, 0);
    read_x(bh, nxt, 2);
    __builtin_amdgcn_sched_barrier(0);
    fill_begin();
#pragma unroll
    for (int c = 0; c < 8; ++c) fill_call(stage, c);   // stage k+4 into the slot every wave has just finished reading
    fill_end();
    __builtin_amdgcn_sched_barrier(0);
    mma12(ah_cur, bl);                                // C
    __builtin_amdgcn_sched_barrier(0);
    stage = nxt;
  };
  issue(0);
  issue(1);
  issue(2);
  issue(3);
  asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_g(ah0, 0, 0);
  read_x(bh, 0, 2);
  for (int k = k0; k < k1; k += 2 * kBKS) {
    step(ah0, ah1);
    if (k + kBKS < k1) step(ah1, ah0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  float* dst = slab + (size_t)split * Cout * 9 * Cin;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int tap = ky * 3 + t;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = nt * kTile + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int cc = ct * kTile + wn * 64 + j * 32 + (lane & 31);
          if (n < Cout && cc < Cin) dst[((size_t)n * 9 + tap) * Cin + cc] = acc[t][i][j][r];
        }
  }
}

// ---------------------------------------------------------------------------------------------
// Three taps per workgroup on SPLIT operands, the dx shifts made IN REGISTERS (dilation 1; round 3).
// Counters of k_wgrad_split3 on 1024->1024 at 160x240 (profiles/round3/wgrad_split3_pmc.txt): matrix pipes busy 40 %, 64 % of the
// wave time waiting for fills, 475 M L2 requests per launch = 12.7 TB/s of 32-byte requests — its 32-byte LDS rows (K-step 16, the
// only way 8 operand tiles x 4 stages fit) turn every row of every tile into one half-used L2 request, and three of the eight
// tiles per plane are dx-shifted copies of the same X.  Here ONE copy of X is staged, on a raster whose image rows are padded
// with at least one zero column (Wp = roundup(W + 1, 8)), so the element next to a row's end is a zero and
//     dW[ky][0] = sum_q G[q+1] X[q]      dW[ky][1] = sum_q G[q] X[q]      dW[ky][2] = sum_q G[q] X[q+1]
// (q = padded pixel index + the row offset of ky): the outer taps use a fragment SHIFTED BY ONE ELEMENT — four v_alignbit over
// the fragment's dwords and the first dword of the following 16-byte chunk, read from LDS as one extra ds_read_b32 (for the last
// chunk of a K-step it lies in the NEXT stage of the ring, which is resident by then).  Per K-step of 32 pixels a stage holds
// G_hi, G_lo, X_hi, X_lo = 4 tiles of 128 rows x 64 B (ring of 4 stages = 128 KB): 4x fewer L2 requests per MFMA than
// k_wgrad_split3, half the LDS fragment reads, one staged copy of X instead of three.
// Pipeline (one wavefront per SIMD): a K-step is two 16-pixel slices of 36 MFMAs; the fragments of the next slice are requested
// while the current slice multiplies (two register sets), the 8 fill calls of stage k+3 are spread over the step, one barrier
// per step: behind it stage k+2 has landed (vmcnt(8)) and every wave is done with stage k (lgkmcnt(0)).
// The step after a split's last one is fetched for real (k_lim): its first elements are the "next elements" of the last step.
// ---------------------------------------------------------------------------------------------
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 shift_in_next(bf16x8 f, unsigned next_dword) {
  const u32x4v d = __builtin_bit_cast(u32x4v, f);
  u32x4v o;
  o.x = __builtin_amdgcn_alignbit(d.y, d.x, 16);
  o.y = __builtin_amdgcn_alignbit(d.z, d.y, 16);
  o.z = __builtin_amdgcn_alignbit(d.w, d.z, 16);
  o.w = __builtin_amdgcn_alignbit(next_dword, d.w, 16);
  return __builtin_bit_cast(bf16x8, o);
}

// SPLIT = true : fp32-grade split operands, K-step 32 (64-byte LDS rows), tiles G_hi, G_lo, X_hi, X_lo, 36 MFMAs per 16-pixel slice.
// SPLIT = false: plain bf16 operands, K-step 64 (128-byte rows = whole cache lines), tiles G, X, 12 MFMAs per slice.
// Either way a stage is 32 KB (ring of 4 = 128 KB) and a wave issues 8 fill calls per stage.
template <bool SPLIT>
__global__ __launch_bounds__(kBlock) void k_wgrad_shift(const unsigned short* __restrict__ Gt, const unsigned short* __restrict__ Gt2,
                                                        const unsigned short* __restrict__ Xt, const unsigned short* __restrict__ Xt2,
                                                        const unsigned short* __restrict__ ws_base, unsigned ws_bytes,
                                                        float* __restrict__ slab, int Cout, int Cin, int Coutp, int Cinp, int M,
                                                        int Mp, int H, int Wp, int n_split, int k_per_split) {
  constexpr int kStep = SPLIT ? 32 : 64;               // pixels per K-step = elements per LDS row
  constexpr int kTiles = SPLIT ? 4 : 2;                // SPLIT: 0 = G_hi, 1 = G_lo, 2 = X_hi, 3 = X_lo ; else 0 = G, 1 = X
  constexpr int kX = SPLIT ? 2 : 1;                    // first X tile
  constexpr int kChunks = kStep / 8;                   // 16-byte chunks per row
  constexpr int kSlices = kStep / 16;
  constexpr int kRowsPerCall = 64 / kChunks;           // 16 or 8
  constexpr int kCallsPerTile = 32 / kRowsPerCall;     // 2 or 4
  __shared__ __attribute__((aligned(16))) unsigned short sm[kStages3][kTiles][kTile][kStep];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_c = Cinp / kTile, tiles_n = Coutp / kTile;
  int bid = blockIdx.x;
  const int ct = __builtin_amdgcn_readfirstlane(bid % tiles_c); bid /= tiles_c;
  const int nt = __builtin_amdgcn_readfirstlane(bid % tiles_n); bid /= tiles_n;
  const int ky = __builtin_amdgcn_readfirstlane(bid % 3);
  const int split = __builtin_amdgcn_readfirstlane(bid / 3);
  const int dy = ky - 1;
  const int k0 = split * k_per_split;
  const int k1 = min(k0 + k_per_split, Mp);
  const int k_lim = min(k1, Mp - kStep);

  // loader: one DMA call = kRowsPerCall rows of one tile; wave w owns rows [32w, 32w+32) of every tile.  Chunks are XOR-swizzled
  // (64-byte rows: by (row >> 2) & 3, 128-byte rows: by (row >> 1) & 7) on the global source and again on the fragment read.
  auto swz = [](int row) { return SPLIT ? ((row >> 2) & 3) : ((row >> 1) & 7); };
  constexpr unsigned kOOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ws_base, 0, (int)ws_bytes, 0x00020000);
  // (128-byte rows: the swizzle of a row depends on bit 2 of the call's first row, i.e. on the parity of the call — two chunk
  //  positions per lane, `[part & 1]`; 64-byte rows: one)
  const int lr = lane / kChunks;
  const int gch = (lane % kChunks) ^ swz(lr);
  const int gch2[2] = {gch, SPLIT ? gch : (gch ^ 4)};
  const size_t a_row = (size_t)(nt * kTile + wave * 32 + lr) * Mp;
  const ptrdiff_t b_row = (ptrdiff_t)(((size_t)ct * kTile + wave * 32 + lr) * Mp) + (ptrdiff_t)dy * Wp;   // may reach into the guard
  const unsigned g_off0 = (unsigned)((const char*)(Gt + a_row) - (const char*)ws_base);       // row starts: multiples of 128 B
  const unsigned g_off1 = SPLIT ? (unsigned)((const char*)(Gt2 + a_row) - (const char*)ws_base) : 0u;
  const unsigned x_off0 = (unsigned)((const char*)(Xt + b_row) - (const char*)ws_base);
  const unsigned x_off1 = SPLIT ? (unsigned)((const char*)(Xt2 + b_row) - (const char*)ws_base) : 0u;
  const unsigned call_pitch = (unsigned)Mp * 2u * kRowsPerCall;   // bytes between two calls of a tile
  constexpr int kPos = SPLIT ? 1 : 2;
  int px[kPos], py[kPos];
#pragma unroll
  for (int e = 0; e < kPos; ++e) { px[e] = (k0 + gch2[e] * 8) % Wp; py[e] = ((k0 + gch2[e] * 8) / Wp) % H; }
  int k_issue = k0;
  bool x_ok[kPos];
  auto fill_begin = [&]() {
    const int kk = min(k_issue, k_lim);
#pragma unroll
    for (int e = 0; e < kPos; ++e) x_ok[e] = (kk + gch2[e] * 8 < M) && (py[e] + dy >= 0) && (py[e] + dy < H);
  };
  auto fill_call = [&](int stage, int c) {              // c = tile * kCallsPerTile + part
    const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(min(k_issue, k_lim) * 2);
    const int tile = c / kCallsPerTile, part = c % kCallsPerTile;
    const int e = SPLIT ? 0 : (part & 1);
    const bool is_x = tile >= kX;
    const unsigned base = SPLIT ? (tile == 0 ? g_off0 : tile == 1 ? g_off1 : tile == 2 ? x_off0 : x_off1) : (tile == 0 ? g_off0 : x_off0);
    const unsigned v = (is_x && !x_ok[e]) ? kOOB : base + (unsigned)gch2[e] * 16u + (unsigned)part * call_pitch;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t*)&sm[stage][tile][wave * 32 + kRowsPerCall * part][0], 16, v, so, 0, 0);
  };
  auto fill_end = [&]() {
    k_issue += kStep;
#pragma unroll
    for (int e = 0; e < kPos; ++e) {
      px[e] += kStep;
      while (px[e] >= Wp) { px[e] -= Wp; py[e] = (py[e] + 1 == H) ? 0 : py[e] + 1; }
    }
  };

  f32x16 acc[3][2][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 31;
  const int fhalf = lane >> 5;
  const int ra0 = wm * 64 + frow, rb0 = wn * 64 + frow;

  struct Slice { bf16x8 gh[2], gl[2], xh[2], xl[2]; unsigned ngh[2], ngl[2], nxh[2], nxl[2]; };
  // fragments of slice s of the step in ring slot st, and the first dword of the chunk behind each of them
  auto read_slice = [&](Slice& f, int st, int s) {
    const int c = 2 * s + fhalf;
    const bool wrap = c == kChunks - 1;                 // the following chunk is chunk 0 of the next stage
    const int st_n = wrap ? (st + 1) % kStages3 : st;
    const int c_n = wrap ? 0 : c + 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ra = ra0 + i * 32, rb = rb0 + i * 32;
      const int sa = (c ^ swz(ra)) * 8, sb = (c ^ swz(rb)) * 8;
      const int na = (c_n ^ swz(ra)) * 8, nb = (c_n ^ swz(rb)) * 8;
      f.gh[i] = *reinterpret_cast<const bf16x8*>(&sm[st][0][ra][sa]);
      f.xh[i] = *reinterpret_cast<const bf16x8*>(&sm[st][kX][rb][sb]);
      f.ngh[i] = *reinterpret_cast<const unsigned*>(&sm[st_n][0][ra][na]);
      f.nxh[i] = *reinterpret_cast<const unsigned*>(&sm[st_n][kX][rb][nb]);
      if (SPLIT) {
        f.gl[i] = *reinterpret_cast<const bf16x8*>(&sm[st][kTiles - 3][ra][sa]);
        f.xl[i] = *reinterpret_cast<const bf16x8*>(&sm[st][kTiles - 1][rb][sb]);
        f.ngl[i] = *reinterpret_cast<const unsigned*>(&sm[st_n][kTiles - 3][ra][na]);
        f.nxl[i] = *reinterpret_cast<const unsigned*>(&sm[st_n][kTiles - 1][rb][nb]);
      }
    }
  };
  auto mma4 = [&](int t, const bf16x8 (&g)[2], const bf16x8 (&x)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[i], x[j], acc[t][i][j], 0, 0, 0);
  };
  // the MFMAs of one slice; kFillsPerSlice of the stage's 8 fill calls are issued between its MFMA groups
  constexpr int kFillsPerSlice = 8 / kSlices;           // 4 or 2
  auto slice_mma = [&](const Slice& f, int fill_stage, int fill_lo) {
    mma4(1, f.gh, f.xh);
    if (SPLIT) mma4(1, f.gl, f.xh);
    fill_call(fill_stage, fill_lo);
    if (SPLIT) mma4(1, f.gh, f.xl);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 sgh[2], sgl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { sgh[i] = shift_in_next(f.gh[i], f.ngh[i]); if (SPLIT) sgl[i] = shift_in_next(f.gl[i], f.ngl[i]); }
    mma4(0, sgh, f.xh);
    fill_call(fill_stage, fill_lo + 1);
    if (SPLIT) {
      mma4(0, sgl, f.xh);
      fill_call(fill_stage, fill_lo + 2);
      mma4(0, sgh, f.xl);
    }
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 sxh[2], sxl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { sxh[i] = shift_in_next(f.xh[i], f.nxh[i]); if (SPLIT) sxl[i] = shift_in_next(f.xl[i], f.nxl[i]); }
    mma4(2, f.gh, sxh);
    if (SPLIT) {
      fill_call(fill_stage, fill_lo + 3);
      mma4(2, f.gl, sxh); mma4(2, f.gh, sxl);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  auto issue = [&](int stage) {
    fill_begin();
#pragma unroll
    for (int c = 0; c < 8; ++c) fill_call(stage, c);
    fill_end();
  };
  issue(0);
  issue(1);
  issue(2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // stages 0 and 1 have landed
  __builtin_amdgcn_s_barrier();
  Slice fa, fb;
  read_slice(fa, 0, 0);
  int stage = 0;
  for (int k = k0; k < k1; k += kStep) {
    const int fill = (stage + 3) % kStages3;            // slot of step k-1: every wave left it before the last barrier
    const int nxt = (stage + 1) % kStages3;
    fill_begin();
#pragma unroll
    for (int s = 0; s < kSlices; s += 2) {
      __builtin_amdgcn_sched_barrier(0);
      read_slice(fb, stage, s + 1);
      __builtin_amdgcn_sched_barrier(0);
      slice_mma(fa, fill, s * kFillsPerSlice);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < kSlices) read_slice(fa, stage, s + 2); else read_slice(fa, nxt, 0);
      __builtin_amdgcn_sched_barrier(0);
      slice_mma(fb, fill, (s + 1) * kFillsPerSlice);
    }
    fill_end();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage = nxt;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  float* dst = slab + (size_t)split * Cout * 9 * Cin;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int tap = ky * 3 + t;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = nt * kTile + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int cc = ct * kTile + wn * 64 + j * 32 + (lane & 31);
          if (n < Cout && cc < Cin) dst[((size_t)n * 9 + tap) * Cin + cc] = acc[t][i][j][r];
        }
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

int pick_split(int coutp, int cinp, int mp, int taps) {
  const int tiles = (coutp / kTile) * (cinp / kTile) * taps;
  int s = (3 * kCUs + tiles - 1) / tiles;          // aim at >= 3 workgroups per CU
  const int max_s = mp / (kBK * 8);                 // at least 8 K-steps per split
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 16) s = 16;
  return s;
}

inline int round_up(int x, int a) { return (x + a - 1) / a * a; }

// Split of the three-taps kernel: one workgroup per CU is resident (128 KB of LDS), so the launch is a whole number of
// "rounds" of kCUs workgroups; pick the smallest split whose rounds are within 10 % of the fullest.  0 = too few
// workgroups to fill three quarters of the chip: the one-tap kernel (two workgroups per CU, three times the tiles) is used.
int pick_split3(int coutp, int cinp, int mp) {
  const int tiles = (coutp / kTile) * (cinp / kTile) * 3;
  int max_s = mp / (kBK * 8);
  if (max_s > 16) max_s = 16;
  if (max_s < 1) max_s = 1;
  if (tiles * max_s < (3 * kCUs) / 4) return 0;
  auto eff_of = [&](int sp) {
    const int blocks = tiles * sp;
    const int rounds = (blocks + kCUs - 1) / kCUs;
    return blocks >= (3 * kCUs) / 4 ? (double)blocks / ((double)rounds * kCUs) : 0.0;
  };
  double best_eff = 0.0;
  for (int sp = 1; sp <= max_s; ++sp) best_eff = eff_of(sp) > best_eff ? eff_of(sp) : best_eff;
  if (best_eff <= 0.0) return 0;
  // every split costs one more fp32 slab written and read back: the smallest split within 10 % of the fullest rounds
  for (int sp = 1; sp <= max_s; ++sp)
    if (eff_of(sp) >= 0.9 * best_eff) return sp;
  return 0;
}

// 256 zero bytes per device, allocated once: the source of every masked LDS-DMA row (borders, K tail).
const unsigned short* zero_page_for_current_device() {
  static void* pages[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!pages[dev]) {
    void* p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 256) != hipSuccess) return nullptr;      // synchronous, once per device and process
    pages[dev] = p;
  }
  return static_cast<const unsigned short*>(pages[dev]);
}

// How one convolution is laid out for the GEMM.
struct WgradPlan {
  int mode;        // 0: three dx copies + row offsets (3x3, stride 1, pad == dil) or plain 1x1 ; 1: one copy per tap
  int taps, copies;
  int wp;          // row pitch of the padded raster (mode 0, 3x3) / unused
  int rows_h;      // H for the row mask (mode 0, 3x3) / huge
  int mpix;        // valid extent of the pixel axis
  int mp;          // its padded pitch
  int coutp, cinp, split;
  int split3;      // split of the three-taps-per-workgroup kernel (mode 0, 3x3), 0 = not applicable
  size_t gt_bytes, xt_bytes, slab_bytes, guard;
};

// shift_form: one staged copy of X on a raster with >= 1 zero column per image row (k_wgrad_shift makes the dx shifts in
// registers); only for split operands, 3x3, stride 1, pad == dil == 1.
bool shift_form_enabled() {
  static const bool on = [] { const char* e = getenv("OMNIHD_WGRAD_SHIFT"); return !(e && e[0] == '0'); }();
  return on;
}

bool make_plan(int batch, int h, int w, int cin, int ho, int wo, int cout, int kh, int kw, int stride, int pad,
               int dil, WgradPlan* p, bool shift_form = false) {
  if (batch <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || ho <= 0 || wo <= 0) return false;
  if (kh != kw || (kh != 1 && kh != 3) || stride < 1 || dil < 1 || pad < 0) return false;
  if (cin % 8 || cout % 8) return false;
  if (ho != (h + 2 * pad - dil * (kh - 1) - 1) / stride + 1 || wo != (w + 2 * pad - dil * (kw - 1) - 1) / stride + 1) return false;
  p->taps = kh * kw;
  p->coutp = round_up(cout, kTile);
  p->cinp = round_up(cin, kTile);
  if (kh == 3 && stride == 1 && pad == dil && dil <= kMaxDil) {
    p->mode = 0; p->copies = 3; p->wp = round_up(w, 8); p->rows_h = h;
    if (shift_form && dil == 1) { p->copies = 1; p->wp = round_up(w + 1, 8); }
    p->mpix = batch * h * p->wp;
  } else if (kh == 1 && stride == 1 && pad == 0) {
    p->mode = 0; p->copies = 1; p->wp = 64; p->rows_h = 1 << 30;
    p->mpix = batch * h * w;
  } else {
    p->mode = 1; p->copies = p->taps; p->wp = 64; p->rows_h = 1 << 30;
    p->mpix = batch * ho * wo;
  }
  p->mp = padded_pixels((size_t)p->mpix);
  p->split = pick_split(p->coutp, p->cinp, p->mp, p->taps);
  p->split3 = (p->mode == 0 && p->taps == 9) ? pick_split3(p->coutp, p->cinp, p->mp) : 0;
  p->gt_bytes = align_up((size_t)p->coutp * p->mp * 2, 256);
  p->xt_bytes = align_up((size_t)p->copies * p->cinp * p->mp * 2, 256);
  p->slab_bytes = align_up((size_t)(p->split3 > p->split ? p->split3 : p->split) * cout * p->taps * cin * 4, 256);
  p->guard = align_up((size_t)dil * p->wp * 2 + 256, 256);
  return true;
}

// The plan of one call: the shift form where it applies (split operands, 3x3, stride 1, pad == dil == 1, and the geometry fills
// the chip with three-tap workgroups), the three-copy form otherwise.
bool plan_for(int batch, int h, int w, int cin, int ho, int wo, int cout, int kh, int kw, int stride, int pad, int dil,
              bool split, WgradPlan* p, bool* shift_form) {
  static const bool three_taps = [] { const char* e = getenv("OMNIHD_WGRAD_3TAPS"); return !(e && e[0] == '0'); }();
  *shift_form = false;
  (void)split;
  if (three_taps && shift_form_enabled() && kh == 3 && kw == 3 && stride == 1 && pad == 1 && dil == 1 &&
      make_plan(batch, h, w, cin, ho, wo, cout, kh, kw, stride, pad, dil, p, true) && p->split3 > 0) {
    *shift_form = true;
    return true;
  }
  return make_plan(batch, h, w, cin, ho, wo, cout, kh, kw, stride, pad, dil, p, false);
}

}  // namespace
}  // namespace omnihd

using namespace omnihd;

extern "C" size_t omnihd_conv_wgrad_workspace_bytes(int batch, int h, int w, int cin, int ho, int wo, int cout,
                                                    int kh, int kw, int stride, int pad, int dil) {
  WgradPlan p;
  bool shift_form = false;
  if (!plan_for(batch, h, w, cin, ho, wo, cout, kh, kw, stride, pad, dil, false, &p, &shift_form)) return 0;
  return 256 + p.gt_bytes + p.xt_bytes + p.slab_bytes + 4 * p.guard;
}

namespace {
// x_lo / g_lo == nullptr: plain bf16 operands.  Otherwise the fp32-grade three-term form on split operands: all four planes
// are staged k-major once and ONE GEMM launch runs the K loop three times (hi*hi, hi*lo, lo*hi) into the same tiles.
int wgrad_impl(const void* x_nhwc, const void* x_lo, const void* gout_nhwc, const void* g_lo, float* dw, int batch, int h, int w,
               int cin, int ho, int wo, int cout, int kh, int kw, int stride, int pad, int dil, void* workspace,
               size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const bool split = x_lo != nullptr;
  WgradPlan p;
  bool shift_form = false;
  if (!plan_for(batch, h, w, cin, ho, wo, cout, kh, kw, stride, pad, dil, split, &p, &shift_form)) {
    set_error("conv_wgrad: unsupported geometry (square 1x1/3x3 kernels, channels multiples of 8, consistent output size)");
    return OMNIHD_ERR_ARG;
  }
  OMNIHD_REQUIRE(x_nhwc && gout_nhwc && dw && workspace && (!split || g_lo), "null pointer");
  const size_t need = (split ? 2 : 1) * (p.gt_bytes + p.xt_bytes + 2 * p.guard) + 256 + p.slab_bytes + 2 * p.guard;
  if (workspace_bytes < need) {
    set_error("conv_wgrad: workspace %zu < required %zu", workspace_bytes, need);
    return OMNIHD_ERR_WORKSPACE;
  }
  const int k_per_split = ((p.mp / kBK + p.split - 1) / p.split) * kBK;
  const unsigned short* zero_page = zero_page_for_current_device();
  OMNIHD_REQUIRE(zero_page != nullptr, "could not allocate the zero page");
  char* q = static_cast<char*>(workspace) + 256 + p.guard;
  unsigned short* Gt = reinterpret_cast<unsigned short*>(q);
  q += p.gt_bytes + p.guard;
  unsigned short* Xt = reinterpret_cast<unsigned short*>(q);
  q += p.xt_bytes + p.guard;
  unsigned short *Gt2 = nullptr, *Xt2 = nullptr;
  if (split) {
    Gt2 = reinterpret_cast<unsigned short*>(q);
    q += p.gt_bytes + p.guard;
    Xt2 = reinterpret_cast<unsigned short*>(q);
    q += p.xt_bytes + p.guard;
  }
  float* slab = reinterpret_cast<float*>(q);
  const int n_terms = split ? 3 : 1;

  const int cmax = p.coutp > p.cinp ? p.coutp : p.cinp;
  for (int plane = 0; plane < (split ? 2 : 1); ++plane) {
    const unsigned short* xs = static_cast<const unsigned short*>(plane ? x_lo : x_nhwc);
    const unsigned short* gs = static_cast<const unsigned short*>(plane ? g_lo : gout_nhwc);
    unsigned short *gt = plane ? Gt2 : Gt, *xt = plane ? Xt2 : Xt;
    const StageArgs aG{gs, gt, cout, p.coutp, 1};
    if (p.mode == 0 && p.taps == 9) {
      // G and X share the padded raster (b, y, x) with row pitch wp
      const StageArgs aX{xs, xt, cin, p.cinp, p.copies};
      hipLaunchKernelGGL(k_to_kmajor, dim3(p.mp / 64, cmax / 64, 2), dim3(kBlock), 0, st, aG, aX, batch * h * w, w, p.wp, p.mp, dil);
    } else if (p.mode == 0) {
      // 1x1: the raster is the plain pixel index (one "row" of mp pixels, nothing to shift)
      const StageArgs aX{xs, xt, cin, p.cinp, 1};
      hipLaunchKernelGGL(k_to_kmajor, dim3(p.mp / 64, cmax / 64, 2), dim3(kBlock), 0, st, aG, aX, p.mpix, p.mpix, p.mp, p.mp, 1);
    } else {
      hipLaunchKernelGGL(k_to_kmajor, dim3(p.mp / 64, p.coutp / 64, 1), dim3(kBlock), 0, st, aG, aG, p.mpix, p.mpix, p.mp, p.mp, 1);
      hipLaunchKernelGGL(k_taps_kmajor, dim3(p.mp / 64, p.cinp / 64, p.taps), dim3(kBlock), 0, st, xs, batch, h, w, cin,
                         p.cinp, ho, wo, p.mp, kw, stride, pad, dil, xt);
    }
  }
  static const bool three_taps = [] { const char* e = getenv("OMNIHD_WGRAD_3TAPS"); return !(e && e[0] == '0'); }();
  int n_split = p.split;
  if (p.split3 > 0 && three_taps) {
    n_split = p.split3;
    const int k_per_split3 = ((p.mp / kBK + n_split - 1) / n_split) * kBK;
    const int blocks = (p.cinp / kTile) * (p.coutp / kTile) * 3 * n_split;
    static const bool fused8 = [] { const char* e = getenv("OMNIHD_WGRAD_SPLIT8"); return !(e && e[0] == '0'); }();
    // (the 8-tile kernel reaches all four staged planes through one 32-bit buffer descriptor over the workspace)
    const bool one_desc = (size_t)(reinterpret_cast<char*>(slab) - static_cast<char*>(workspace)) < (1ull << 31);
    if (shift_form) {
      OMNIHD_REQUIRE(one_desc && p.copies == 1, "conv_wgrad_split: staged operands beyond 2 GiB");
      if (split)
        hipLaunchKernelGGL(k_wgrad_shift<true>, dim3(blocks), dim3(kBlock), 0, st, Gt, Gt2, Xt, Xt2,
                           reinterpret_cast<const unsigned short*>(workspace), (unsigned)(reinterpret_cast<char*>(slab) - static_cast<char*>(workspace)),
                           n_split > 1 ? slab : dw, cout, cin, p.coutp, p.cinp, p.mpix, p.mp, p.rows_h, p.wp, n_split, k_per_split3);
      else
        hipLaunchKernelGGL(k_wgrad_shift<false>, dim3(blocks), dim3(kBlock), 0, st, Gt, Gt, Xt, Xt,
                           reinterpret_cast<const unsigned short*>(workspace), (unsigned)(reinterpret_cast<char*>(slab) - static_cast<char*>(workspace)),
                           n_split > 1 ? slab : dw, cout, cin, p.coutp, p.cinp, p.mpix, p.mp, p.rows_h, p.wp, n_split, k_per_split3);
    } else if (split && fused8 && one_desc)
      hipLaunchKernelGGL(k_wgrad_split3, dim3(blocks), dim3(kBlock), 0, st, Gt, Gt2, Xt, Xt2,
                         reinterpret_cast<const unsigned short*>(workspace), (unsigned)(reinterpret_cast<char*>(slab) - static_cast<char*>(workspace)),
                         n_split > 1 ? slab : dw, cout, cin, p.coutp, p.cinp, p.mpix, p.mp, p.rows_h, p.wp, n_split, k_per_split3, dil);
    else
      hipLaunchKernelGGL(k_wgrad_mfma_glds3, dim3(blocks), dim3(kBlock), 0, st, Gt, Xt, zero_page, n_split > 1 ? slab : dw, cout,
                         cin, p.coutp, p.cinp, p.mpix, p.mp, p.rows_h, p.wp, n_split, k_per_split3, dil, Gt2, Xt2, n_terms);
  } else {
    const int blocks = (p.cinp / kTile) * (p.coutp / kTile) * p.taps * p.split;
    hipLaunchKernelGGL(k_wgrad_mfma_glds, dim3(blocks), dim3(kBlock), 0, st, Gt, Xt, zero_page, p.split > 1 ? slab : dw,
                       cout, cin, p.coutp, p.cinp, p.mpix, p.mp, p.rows_h, p.wp, p.split, k_per_split, p.taps, p.mode, dil,
                       Gt2, Xt2, n_terms);
  }
  if (n_split > 1) {
    const size_t n = (size_t)cout * p.taps * cin;
    hipLaunchKernelGGL(k_sum_slabs, dim3(grid_for((int64_t)n, kBlock * 4)), dim3(kBlock), 0, st, slab, n_split, n, dw);
  }
  return check_launch(split ? "conv_wgrad_split" : "conv_wgrad_bf16");
}
}  // namespace

extern "C" int omnihd_conv_wgrad_bf16(const void* x_nhwc, const void* gout_nhwc, float* dw, int batch, int h, int w,
                                      int cin, int ho, int wo, int cout, int kh, int kw, int stride, int pad,
                                      int dil, void* workspace, size_t workspace_bytes, void* stream) {
  return wgrad_impl(x_nhwc, nullptr, gout_nhwc, nullptr, dw, batch, h, w, cin, ho, wo, cout, kh, kw, stride, pad, dil, workspace,
                    workspace_bytes, stream);
}

extern "C" size_t omnihd_conv_wgrad_split_workspace_bytes(int batch, int h, int w, int cin, int ho, int wo, int cout, int kh,
                                                          int kw, int stride, int pad, int dil) {
  WgradPlan p;
  bool shift_form = false;
  if (!plan_for(batch, h, w, cin, ho, wo, cout, kh, kw, stride, pad, dil, true, &p, &shift_form)) return 0;
  return 256 + 2 * (p.gt_bytes + p.xt_bytes + 2 * p.guard) + p.slab_bytes + 2 * p.guard;
}

extern "C" int omnihd_conv_wgrad_split(const void* x_hi, const void* x_lo, const void* g_hi, const void* g_lo, float* dw,
                                       int batch, int h, int w, int cin, int ho, int wo, int cout, int kh, int kw, int stride,
                                       int pad, int dil, void* workspace, size_t workspace_bytes, void* stream) {
  OMNIHD_REQUIRE(x_lo && g_lo, "null pointer");
  return wgrad_impl(x_hi, x_lo, g_hi, g_lo, dw, batch, h, w, cin, ho, wo, cout, kh, kw, stride, pad, dil, workspace,
                    workspace_bytes, stream);
}

// The two original entry points, kept as names for the common cases.
extern "C" size_t omnihd_conv3x3_wgrad_workspace_bytes(int batch, int h, int w, int cin, int cout) {
  const size_t n = omnihd_conv_wgrad_workspace_bytes(batch, h, w, cin, h, w, cout, 3, 3, 1, 1, 1);
  return n ? n : 256;
}

extern "C" int omnihd_conv3x3_wgrad_bf16(const void* x_nhwc, const void* gout_nhwc, float* dw, int batch, int h,
                                         int w, int cin, int cout, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  return omnihd_conv_wgrad_bf16(x_nhwc, gout_nhwc, dw, batch, h, w, cin, h, w, cout, 3, 3, 1, 1, 1, workspace,
                                workspace_bytes, stream);
}

extern "C" size_t omnihd_conv1x1_wgrad_workspace_bytes(int m, int cin, int cout) {
  const size_t n = omnihd_conv_wgrad_workspace_bytes(1, 1, m, cin, 1, m, cout, 1, 1, 1, 0, 1);
  return n ? n : 256;
}

extern "C" int omnihd_conv1x1_wgrad_bf16(const void* x_rows, const void* gout_rows, float* dw, int m, int cin,
                                         int cout, void* workspace, size_t workspace_bytes, void* stream) {
  return omnihd_conv_wgrad_bf16(x_rows, gout_rows, dw, 1, 1, m, cin, 1, m, cout, 1, 1, 1, 0, 1, workspace,
                                workspace_bytes, stream);
}
