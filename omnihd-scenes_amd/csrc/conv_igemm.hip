// Forward and data gradient of the dense stride-1 convolutions on the gfx950 matrix cores: implicit GEMM over NHWC.
//
// Where it sits: the BEV encoder of the camera stream (3x3 convs 1024->1024->512->512->256 at 160x240, reference
// bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:201-214), the fusion conv 640->384
// (bevf_faster_rcnn_bevdepth.py:61-72) and every other "same" 1x1 / 3x3 convolution of the detector.  north_star: "the
// BEV conv encoder written for gfx950 ... MFMA used only for the dense BEV convs".
//
//   y[m][n] = sum over taps t = (ky, kx) and channels c of  x[(b, y + (ky-k/2)*d, x + (kx-k/2)*d)][c] * w[n][t][c]
//
// With channels-last activations AND channels-last weights ((Cout, kh, kw, Cin) memory) both operands of that GEMM are
// already contiguous along the reduction index (t, c): no im2col buffer, no layout pass.  M = pixels, N = Cout,
// K = taps * Cin.  A-tile rows are the pixels of the output tile shifted by the tap; a lane whose source pixel lies outside
// the image fetches from a 256-byte zero page instead (LDS-DMA takes a per-lane global address).
// The data gradient is the same kernel applied to the output gradient with the weights laid out (Cin, kh, kw, Cout) and
// the taps mirrored: gx[m][c] = sum_{t,n} g[shift_{-t}(m)][n] * w[n][t][c]  (omnihd_conv_dgrad_weights does that re-layout).
//
// Kernel: TM x TN output tile per workgroup of WM x WN wavefronts, each wavefront a 64x64 quadrant as 2x2
// v_mfma_f32_32x32x16_bf16 tiles; K stepped by 64 through a ring of LDS stages filled by LDS-DMA
// (`global_load_lds`, 16 B per lane, no VGPR round trip), rows XOR-swizzled on the global source chunk and again on the
// fragment read (conflict-free ds_read_b128), hand-counted s_waitcnt vmcnt + raw s_barrier (as csrc/conv_wgrad.hip).
// Tile shapes: 256x128 with 8 wavefronts (two per SIMD: one multiplies while the other waits on LDS; 85 FLOP per
// operand byte) for the BEV-sized layers, 128x128 with 4 wavefronts for narrow ones.  fp32 accumulation, one rounding to
// bf16 in the epilogue (+ optional fp32 bias).  Workgroups are numbered so that one XCD keeps ONE band of output channels:
// its weight slab (TN x K bf16 = 2.4 MB for K = 9216) stays in that XCD's 4 MiB L2 while the activations stream by.
//
// SPLIT = true (round 3): the same kernels at fp32-grade accuracy for the reference-precision (fp32) training step.  Every
// fp32 operand is handed over as TWO bf16 planes, hi = bf16(v) and lo = bf16(v - hi) (omnihd_split_f32), and a product is
//     x * w  ~=  x_hi*w_hi + x_hi*w_lo + x_lo*w_hi            (fp32 accumulation; the dropped x_lo*w_lo is 2^-16 of the product)
// i.e. three bf16 MFMAs instead of one fp32 MFMA that runs at 1/16 of their rate.  One LDS row (128 B) then holds 32 channels
// of the hi plane followed by the same 32 channels of the lo plane — the LDS-DMA source address is per lane, so the loader
// only picks the plane by the 16-byte slot — and everything else (stage ring, swizzle, DMA call counts, waits) is unchanged:
// a K-step covers 32 channels, reads 8 fragments (4 hi, 4 lo) per 16-deep slice and issues 12 MFMAs on them (0.67 LDS reads
// per MFMA against 1.0 in the bf16 kernel, which is bound by the bytes streamed into LDS).  The result is written in fp32.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace omnihd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// one 32x32x16 matrix product on 16-bit fragments: bf16 operands, or (F16: the TF32-grade form of round 6) IEEE half operands —
// the same bits in the same registers, another instruction
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef __attribute__((address_space(1))) const void gbl_ptr_t;

constexpr int kBK = 64;        // reduction elements per K-step

__device__ __forceinline__ unsigned short f2bf_rn(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// F16 (with SPLIT = false): operands are IEEE half planes, the output is fp32 = alpha * (x * w) + bias with alpha read from the
// device (the inverse of the power-of-two scale a gradient plane was converted with; NULL = 1).
template <int WM, int WN, int STAGES, bool SPREAD, bool SPLIT = false, bool F16 = false>
__global__ __launch_bounds__(64 * WM * WN) void k_conv_igemm(
    const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wt, const unsigned short* __restrict__ zero_page,
    const float* __restrict__ bias, void* __restrict__ Yv, int M, int H, int W, int Cin, int Cout, int ksize,
    int dil, int tiles_m, int tiles_n, int tiles_per_xcd, const unsigned short* __restrict__ X2 = nullptr,
    const unsigned short* __restrict__ Wt2 = nullptr, const float* __restrict__ alpha = nullptr) {
  static_assert(!(SPLIT && F16), "the half form has one plane per operand");
  constexpr int kCS = SPLIT ? 32 : 64;            // channels per K-step
  constexpr int TM = 64 * WM, TN = 64 * WN, NW = WM * WN;
  constexpr int A_CALLS = TM / (8 * NW), B_CALLS = TN / (8 * NW);     // 8-row LDS-DMA calls per wavefront and stage
  constexpr int CALLS = A_CALLS + B_CALLS;
  static_assert(TM % (8 * NW) == 0 && TN % (8 * NW) == 0, "rows must split evenly over the wavefronts");
  // one LDS object only (a second one makes hipcc drain the DMA queue before every ds_read)
  __shared__ __attribute__((aligned(16))) unsigned short sm[STAGES][TM + TN][kBK];

  // workgroup -> tile: XCD x (= block % 8, observed dispatch rule, speed only) walks a contiguous run of the n-major tile
  // list, i.e. stays on one band of output channels
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int t = xcd * tiles_per_xcd + slot;
  if (slot >= tiles_per_xcd || t >= tiles_m * tiles_n) return;
  const int nt = t / tiles_m, mt = t - nt * tiles_m;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: LDS destinations of the fills are SALU math
  const int taps = ksize * ksize;
  const int half = ksize / 2;
  const int K = taps * Cin;

  // ---- loader state: this lane's rows of A (pixels) and of B (output channels) -------------------------------
  // Fills are `buffer_load_dwordx4 ... offen lds` through one range-checked descriptor per operand (see k_conv_igemm_rs): a
  // row that does not exist is an offset beyond the buffer and arrives as zeros.
  constexpr unsigned kOOB = 0x80000000u;
  const unsigned short* xbase = (SPLIT && X2 < X) ? X2 : X;
  const unsigned short* wbase = (SPLIT && Wt2 < Wt) ? Wt2 : Wt;
  const size_t x_plane = (size_t)M * Cin * 2, w_plane = (size_t)Cout * K * 2;
  const unsigned x_hi_off = (unsigned)((const char*)X - (const char*)xbase), w_hi_off = (unsigned)((const char*)Wt - (const char*)wbase);
  const unsigned x_lo_off = SPLIT ? (unsigned)((const char*)X2 - (const char*)xbase) : 0u;
  const unsigned w_lo_off = SPLIT ? (unsigned)((const char*)Wt2 - (const char*)wbase) : 0u;
  const unsigned x_bytes = (unsigned)((SPLIT ? (x_lo_off > x_hi_off ? x_lo_off : x_hi_off) : 0u) + x_plane);
  const unsigned w_bytes = (unsigned)((SPLIT ? (w_lo_off > w_hi_off ? w_lo_off : w_hi_off) : 0u) + w_plane);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, (int)w_bytes, 0x00020000);

  const int lr = lane >> 3;                       // row inside an 8-row call
  const int pos = lane & 7;                       // 16-byte slot inside the 128-byte LDS row
  const int c_even = pos ^ ((lane >> 4) & 7);     // global chunk for even calls; odd calls use c_even ^ 4
  // per call: the lane's byte offset at tap (0,0), channel slice 0 (SPLIT: slots 0-3 of a row come from the hi plane, slots
  // 4-7 from the lo plane of the same 32 channels); a fill adds ONE wave-uniform offset to it
  int a_y[A_CALLS], a_x[A_CALLS];                 // pixel coordinates (y = -huge: row beyond M)
  unsigned a_off[A_CALLS];
#pragma unroll
  for (int i = 0; i < A_CALLS; ++i) {
    const int c = (i & 1) ? (c_even ^ 4) : c_even;
    const int cc = SPLIT ? (c & 3) : c;
    const int m = mt * TM + wave * (8 * A_CALLS) + 8 * i + lr;
    if (m < M) {
      const int xx = m % W, r = m / W;
      a_x[i] = xx; a_y[i] = r % H;
      a_off[i] = ((SPLIT && c >= 4) ? x_lo_off : x_hi_off) + (unsigned)((size_t)m * Cin * 2) + cc * 16;
    } else {
      a_x[i] = 0; a_y[i] = -(1 << 20); a_off[i] = kOOB;
    }
  }
  unsigned b_off[B_CALLS];                        // kOOB + any tap / slice offset stays beyond the buffer
#pragma unroll
  for (int i = 0; i < B_CALLS; ++i) {
    const int call = A_CALLS + i;                 // A_CALLS is even: the parity of a B call is that of its index
    const int c = (call & 1) ? (c_even ^ 4) : c_even;
    const int cc = SPLIT ? (c & 3) : c;
    const int n = nt * TN + wave * (8 * B_CALLS) + 8 * i + lr;
    b_off[i] = n < Cout ? ((SPLIT && c >= 4) ? w_lo_off : w_hi_off) + (unsigned)((size_t)n * K * 2) + cc * 16 : kOOB;
  }

  int k_tap = 0, k_c = 0;                         // position of the NEXT K-step to issue
  // one LDS-DMA call (8 rows x 128 B) of the K-step at (k_tap, k_c): calls 0..A_CALLS-1 fetch A rows, the rest B rows
  auto issue_call = [&](int stage, bool real, int call) {
    if (call < A_CALLS) {
      const int i = call;
      const int dy = (k_tap / ksize - half) * dil, dx = (k_tap % ksize - half) * dil;
      const bool ok = real && (unsigned)(a_y[i] + dy) < (unsigned)H && (unsigned)(a_x[i] + dx) < (unsigned)W;
      const int goff = ((dy * W + dx) * Cin + k_c) * 2;                        // wave-uniform, may be negative
      const unsigned voff = ok ? a_off[i] + (unsigned)goff : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_ptr_t*)&sm[stage][wave * (8 * A_CALLS) + 8 * i][0], 16, voff, 0, 0, 0);
    } else {
      const int i = call - A_CALLS;
      const unsigned goff = real ? (unsigned)((k_tap * Cin + k_c) * 2) : kOOB;   // wave-uniform
      // (the offset in a named variable: with the sum written in the argument list hipcc's host pass silently dropped the
      // kernel's launch stub)
      const unsigned voff = b_off[i] + goff;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t*)&sm[stage][TM + wave * (8 * B_CALLS) + 8 * i][0], 16, voff, 0, 0, 0);
    }
  };
  // channels OUTER, taps INNER: the nine taps of one 64-channel slice read the same pixels' 128-byte lines (shifted), so a
  // line fetched for the first tap is an L2 hit for the other eight; with taps outer every tap re-fetched 2 KB-strided lines
  // that had long left the 4 MiB L2 (753 -> 793 TFLOP/s on 1024->1024 at 160x240; L2 hit rate 90 %)
  auto advance = [&]() { if (++k_tap == taps) { k_tap = 0; k_c += kCS; } };
  auto issue = [&](int stage, bool real) {
#pragma unroll
    for (int call = 0; call < CALLS; ++call) issue_call(stage, real, call);
    advance();
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wm = wave / WN, wn = wave % WN;
  const int frow = lane & 31;
  const int fhalf = lane >> 5;
  const int n_steps = K / kCS;

  // one 16-deep slice of fragments: SPLIT: [0..1] a_hi, [2..3] a_lo, [4..5] b_hi, [6..7] b_lo; else [0..1] a, [2..3] b
  constexpr int NS = kCS / 16;
  constexpr int NF = SPLIT ? 8 : 4;
  auto load_slice = [&](int stage, int ks, bf16x8 (&f)[NF]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ra = wm * 64 + i * 32 + frow;
      const int rb = wn * 64 + i * 32 + frow;
      const int c = ks * 2 + fhalf;
      f[i] = *reinterpret_cast<const bf16x8*>(&sm[stage][ra][((c ^ ((ra >> 1) & 7)) * 8)]);
      if constexpr (SPLIT) {
        f[2 + i] = *reinterpret_cast<const bf16x8*>(&sm[stage][ra][(((c + 4) ^ ((ra >> 1) & 7)) * 8)]);
        f[4 + i] = *reinterpret_cast<const bf16x8*>(&sm[stage][TM + rb][((c ^ ((rb >> 1) & 7)) * 8)]);
        f[6 + i] = *reinterpret_cast<const bf16x8*>(&sm[stage][TM + rb][(((c + 4) ^ ((rb >> 1) & 7)) * 8)]);
      } else {
        f[2 + i] = *reinterpret_cast<const bf16x8*>(&sm[stage][TM + rb][((c ^ ((rb >> 1) & 7)) * 8)]);
      }
    }
  };
  auto mma_slice = [&](const bf16x8 (&f)[NF]) {
    if constexpr (SPLIT) {
      // term-major order: consecutive MFMAs go to different accumulators
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2 + i], f[4 + j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[i], f[6 + j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[i], f[4 + j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<F16>(f[i], f[2 + j], acc[i][j]);
    }
  };
  auto wait_stage = [&]() {   // this wave's fills of the next K-step have landed: (STAGES-2) younger K-steps may be outstanding
    __builtin_amdgcn_sched_barrier(0);
    if (STAGES == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (STAGES == 4) {
      if (CALLS == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else {
      if (CALLS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // prologue: STAGES-1 K-steps in flight
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) issue(s, s < n_steps);
  bf16x8 fr[NS][NF];
  int stage = 0;
  // staggered two-group schedule for the 8-wavefront shapes (see k_conv_igemm_rs): waves 0-3 read and fill while waves 4-7
  // multiply, then the roles swap.  (The segment lambdas live at function scope: hipcc drops the host stub of the kernel
  // when they are declared inside the `if constexpr` branch.)
  auto seg_load = [&](int step) {
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) load_slice(stage, ks, fr[ks]);
    __builtin_amdgcn_sched_barrier(0);
    issue((stage + STAGES - 1) % STAGES, step + STAGES - 1 < n_steps);
    stage = (stage + 1) % STAGES;
  };
  auto seg_compute = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) mma_slice(fr[ks]);
    __builtin_amdgcn_s_setprio(0);
  };
  if (NW == 8) {
    wait_stage();
    __builtin_amdgcn_s_barrier();
    if (wave < 4) {
      for (int step = 0; step < n_steps; ++step) { seg_load(step); bar(); seg_compute(); wait_stage(); bar(); }
    } else {
      bar();
      for (int step = 0; step < n_steps; ++step) {
        seg_load(step); wait_stage(); bar(); seg_compute();
        if (step + 1 < n_steps) bar();
      }
    }
  } else {
    for (int step = 0; step < n_steps; ++step) {
      wait_stage();
      __builtin_amdgcn_s_barrier();
      const int fill = (stage + STAGES - 1) % STAGES;             // its buffer was last read before this barrier
      const bool fill_real = step + STAGES - 1 < n_steps;
      // all fragment reads of the K-step first, then its MFMAs with the fills of the next K-steps spread behind them
#pragma unroll
      for (int ks = 0; ks < NS; ++ks) load_slice(stage, ks, fr[ks]);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int PER = (CALLS + NS - 1) / NS;
#pragma unroll
      for (int ks = 0; ks < NS; ++ks) {
        mma_slice(fr[ks]);
#pragma unroll
        for (int q = 0; q < PER; ++q)
          if (ks * PER + q < CALLS) issue_call(fill, fill_real, ks * PER + q);
        __builtin_amdgcn_sched_barrier(0);
      }
      advance();
      stage = (stage + 1) % STAGES;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the dummy tail loads before the epilogue stores

  // epilogue: C/D layout of 32x32 MFMA: col = lane & 31 (B row = output channel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const float alpha_v = (F16 && alpha) ? *alpha : 1.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = nt * TN + wn * 64 + j * 32 + (lane & 31);
    const float bv = (bias && n < Cout) ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mt * TM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M && n < Cout) {
          if constexpr (SPLIT) static_cast<float*>(Yv)[(size_t)m * Cout + n] = acc[i][j][r] + bv;
          else if constexpr (F16) static_cast<float*>(Yv)[(size_t)m * Cout + n] = acc[i][j][r] * alpha_v + bv;
          else static_cast<unsigned short*>(Yv)[(size_t)m * Cout + n] = f2bf_rn(acc[i][j][r] + bv);
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------
// 3x3 variant with ROW-SHIFT REUSE of the activation tile (k_conv_igemm_rs).
// Ablation of k_conv_igemm on 1024->1024 at 160x240 (tile codes 251/252): operand streaming alone 0.688 ms, MFMAs + LDS
// reads alone 0.337 ms, together 0.854 ms — the kernel is bound by the bytes it streams into LDS (12 TB/s out of the L2s,
// 90 % hits), not by the matrix pipes.  Two thirds of those bytes are the A tile, and the three taps of one kernel row
// read the SAME pixels shifted by one: here the A tile of a (channel slice, ky) pair is staged ONCE with a halo of 8 pixels
// on either side (272 rows) and the taps kx = 0,1,2 read it at row offsets 8-d, 8, 8+d; a fragment row whose source pixel
// x + dx falls outside the image row is zeroed in registers (the raster neighbour belongs to another image row).  Only the
// weights are streamed per tap.  Bytes per three K-steps: 40 + 3*16 = 88 KB instead of 144 KB.
// LDS: two A buffers (272 rows + 8 junk rows for the padding calls) + a ring of four B buffers = 134 KB; per wavefront and
// group of three K-steps 5 A calls + 3 x 2 B calls in a fixed program order, so the waits are literal counts:
// vmcnt(4) before the first K-step of a group, vmcnt(9) before the other two.
// ---------------------------------------------------------------------------------------------
constexpr int kHalo = 8;

// Round 3: the slice loop is software-pipelined by hand.  hipcc's schedule of the straightforward loop read two fragments,
// waited lgkmcnt(0), issued one MFMA, read the next fragment, waited again ... five exposed LDS round trips per 16-deep slice
// (ISA dump of the first split build); and it computed every LDS-DMA source address with 64-bit multiply chains behind an
// exec-mask branch.  Now: all fragments of slice s+1 are requested before the MFMAs of slice s are issued (two register
// sets, pinned with sched_barrier), the per-lane source pointers are precomputed and a fill call adds one scalar offset and
// selects the zero page with two v_cndmask.
template <bool SPREAD, bool SPLIT = false, bool F16 = false>
__global__ __launch_bounds__(512) void k_conv_igemm_rs(
    const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wt, const unsigned short* __restrict__ zero_page,
    const float* __restrict__ bias, void* __restrict__ Yv, int M, int H, int W, int Cin, int Cout, int dil,
    int tiles_m, int tiles_n, int tiles_per_xcd, const unsigned short* __restrict__ X2 = nullptr,
    const unsigned short* __restrict__ Wt2 = nullptr, const float* __restrict__ alpha = nullptr) {
  static_assert(!(SPLIT && F16), "the half form has one plane per operand");
  constexpr int TM = 256, TN = 128, WN = 2;
  constexpr int kCS = SPLIT ? 32 : 64;                // channels per K-step (SPLIT: 32 of the hi plane + the same 32 of the lo plane)
  constexpr int NS = kCS / 16;                        // 16-deep slices per K-step
  constexpr int A_ROWS = TM + 2 * kHalo;              // 272 = 34 calls; every wavefront issues 5 (the last 6 are padding)
  constexpr int A_BUF = A_ROWS + 8;                   // + one junk block the padding calls write to
  __shared__ __attribute__((aligned(16))) unsigned short sm[2 * A_BUF + 4 * TN][kBK];
  auto a_buf = [&](int slot) { return sm + slot * A_BUF; };
  auto b_buf = [&](int slot) { return sm + 2 * A_BUF + slot * TN; };

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int t = xcd * tiles_per_xcd + slot;
  if (slot >= tiles_per_xcd || t >= tiles_m * tiles_n) return;
  const int nt = t / tiles_m, mt = t - nt * tiles_m;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: LDS destinations of the fills are SALU math
  const int K = 9 * Cin;
  const int lr = lane >> 3;
  const int pos = lane & 7;
  const int chunk = pos ^ (((wave & 1) << 2) | ((lane >> 4) & 3));   // call index parity == wave parity for A and B calls
  // SPLIT: slots 0-3 of a row come from the hi plane, slots 4-7 from the lo plane
  const int cchunk = SPLIT ? (chunk & 3) : chunk;
  const bool lo_plane = SPLIT && chunk >= 4;

  // The fills are `buffer_load_dwordx4 ... offen lds` through ONE range-checked descriptor per operand (both planes of a
  // split operand lie in it): a lane whose source row does not exist — image border, rows beyond M / Cout, the dummy fills
  // behind the last K-step — asks for an offset beyond the buffer and the hardware writes ZEROS into its LDS slot (checked:
  // scripts/micro/buffer_lds_oob.hip).  No zero page, no pointer select, no 64-bit address arithmetic: a fill call is one
  // v_add_u32 (+ compare / select for the dy border) and the load.
  constexpr unsigned kOOB = 0x80000000u;               // every buffer is smaller than 2 GiB
  const unsigned short* xbase = (SPLIT && X2 < X) ? X2 : X;
  const unsigned short* wbase = (SPLIT && Wt2 < Wt) ? Wt2 : Wt;
  const size_t x_plane = (size_t)M * Cin * 2, w_plane = (size_t)Cout * K * 2;
  const unsigned x_hi_off = (unsigned)((const char*)X - (const char*)xbase), w_hi_off = (unsigned)((const char*)Wt - (const char*)wbase);
  const unsigned x_lo_off = SPLIT ? (unsigned)((const char*)X2 - (const char*)xbase) : 0u;
  const unsigned w_lo_off = SPLIT ? (unsigned)((const char*)Wt2 - (const char*)wbase) : 0u;
  const unsigned x_bytes = (unsigned)((SPLIT ? (x_lo_off > x_hi_off ? x_lo_off : x_hi_off) : 0u) + x_plane);
  const unsigned w_bytes = (unsigned)((SPLIT ? (w_lo_off > w_hi_off ? w_lo_off : w_hi_off) : 0u) + w_plane);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, (int)w_bytes, 0x00020000);

  // A calls of this wavefront: call index ca = wave + 8*q (q = 0..4), rows 8*ca .. 8*ca+7 of the A buffer,
  // row j <-> pixel m0 - kHalo + j.  a_off = byte offset of (pixel, channel chunk) at channel slice 0, dy = 0.
  int a_y[5];
  unsigned a_off[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int ca = wave + 8 * q;
    const long long m = (long long)mt * TM - kHalo + 8 * ca + lr;
    if (ca < A_ROWS / 8 && m >= 0 && m < M) {
      a_off[q] = (lo_plane ? x_lo_off : x_hi_off) + (unsigned)((size_t)m * Cin * 2) + cchunk * 16;
      a_y[q] = (int)((m / W) % H);
    } else {
      a_off[q] = kOOB;
      a_y[q] = -(1 << 20);
    }
  }
  unsigned b_off[2];                                   // kOOB + any tap / slice offset stays beyond the buffer
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int n = nt * TN + 8 * (wave + 8 * q) + lr;
    b_off[q] = n < Cout ? (lo_plane ? w_lo_off : w_hi_off) + (unsigned)((size_t)n * K * 2) + cchunk * 16 : kOOB;
  }

  // fragment rows of this lane and the x coordinate of their pixels (for the border mask)
  const int wm = wave / WN, wn = wave % WN;
  const int frow = lane & 31;
  const int fhalf = lane >> 5;
  int fx[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const long long m = (long long)mt * TM + wm * 64 + i * 32 + frow;
    fx[i] = (int)(m % W);
  }

  const int n_c = Cin / kCS;
  const int n_groups = n_c * 3;
  const int n_steps = n_groups * 3;
  const int row_pitch = W * Cin * 2;                  // bytes per image row

  auto issue_a = [&](int g, int q) {                  // call q (0..4) of the A fill of group g = (c, ky)
    const int c = g / 3, ky = g - 3 * c;
    const int dy = (ky - 1) * dil;
    const int ca = wave + 8 * q;
    const int goff = dy * row_pitch + c * (kCS * 2);                        // wave-uniform, may be negative
    const bool ok = (g < n_groups) && (unsigned)(a_y[q] + dy) < (unsigned)H;
    const unsigned voff = ok ? a_off[q] + (unsigned)goff : kOOB;
    unsigned short(*dst)[kBK] = a_buf(g & 1) + (ca < A_ROWS / 8 ? 8 * ca : A_ROWS);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_ptr_t*)&dst[0][0], 16, voff, 0, 0, 0);
  };
  auto issue_b = [&](int k, int q) {                  // call q (0..1) of the B fill of K-step k = (c, ky, kx)
    const int g = k / 3, kx = k - 3 * g;
    const int c = g / 3, ky = g - 3 * c;
    const unsigned goff = k < n_steps ? (unsigned)(((ky * 3 + kx) * Cin + c * kCS) * 2) : kOOB;   // wave-uniform
    unsigned short(*dst)[kBK] = b_buf(k & 3) + 8 * (wave + 8 * q);
    const unsigned voff = b_off[q] + goff;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t*)&dst[0][0], 16, voff, 0, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // prologue, program order A(0), B(0), B(1), B(2)
#pragma unroll
  for (int q = 0; q < 5; ++q) issue_a(0, q);
#pragma unroll
  for (int kk = 0; kk < 3; ++kk) {
    issue_b(kk, 0);
    issue_b(kk, 1);
  }

  // one 16-deep slice of fragments: SPLIT: [0..1] a_hi, [2..3] a_lo, [4..5] b_hi, [6..7] b_lo; else [0..1] a, [2..3] b
  constexpr int NF = SPLIT ? 8 : 4;
  auto load_slice = [&](const unsigned short(*A)[kBK], const unsigned short(*B)[kBK], int dx, int ks, bf16x8 (&f)[NF]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ra = wm * 64 + i * 32 + frow + kHalo + dx;          // row of the A buffer
      const int rb = wn * 64 + i * 32 + frow;
      const int c = ks * 2 + fhalf;
      f[i] = *reinterpret_cast<const bf16x8*>(&A[ra][((c ^ ((ra >> 1) & 7)) * 8)]);
      if constexpr (SPLIT) {
        f[2 + i] = *reinterpret_cast<const bf16x8*>(&A[ra][(((c + 4) ^ ((ra >> 1) & 7)) * 8)]);
        f[4 + i] = *reinterpret_cast<const bf16x8*>(&B[rb][((c ^ ((rb >> 1) & 7)) * 8)]);
        f[6 + i] = *reinterpret_cast<const bf16x8*>(&B[rb][(((c + 4) ^ ((rb >> 1) & 7)) * 8)]);
      } else {
        f[2 + i] = *reinterpret_cast<const bf16x8*>(&B[rb][((c ^ ((rb >> 1) & 7)) * 8)]);
      }
    }
  };
  auto mask_slice = [&](bf16x8 (&f)[NF], const bool (&ok)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (!ok[i]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          f[i][e] = (__bf16)0.0f;
          if constexpr (SPLIT) f[2 + i][e] = (__bf16)0.0f;
        }
      }
  };
  auto mma_slice = [&](const bf16x8 (&f)[NF]) {
    if constexpr (SPLIT) {
      // term-major order: consecutive MFMAs go to different accumulators
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2 + i], f[4 + j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[i], f[6 + j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[i], f[4 + j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<F16>(f[i], f[2 + j], acc[i][j]);
    }
  };

  // ---- staggered two-group schedule ------------------------------------------------------------------------------
  // The two wavefronts of a SIMD (wave w and w + 4) used to run in lockstep behind one barrier per K-step: both read their
  // fragments, both issued their MFMAs into the one matrix pipe, both issued their LDS-DMA fills (60-185 cycles each, during
  // which an in-order wavefront issues no MFMA) — the pipe was busy 44 % of the time with the reads already hoisted.  Now a
  // K-step is two half-steps separated by barriers: in one, group A (waves 0-3) is in its LOAD segment (all fragment reads of
  // the K-step, its share of the fills, the border masks) while group B (waves 4-7) is in its COMPUTE segment (nothing but
  // MFMAs), in the other the roles are swapped:
  //     A:  L(0) | C(0) w(1) | L(1) | C(1) w(2) | ...            w(k) = wait for this wave's fills of K-step k
  //     B:   -   | L(0) w(1) | C(0) | L(1) w(2) | C(1) | ...
  // Stage k is read by A in the half-step before BARa_k and by B in the one after it; its fills were waited for by every
  // wave before BARb_{k-1}; the fills issued in L(k) go to buffers whose last read (K-step k-1, group B) lies before
  // BARb_{k-1}.  Per wavefront the program order of fills and waits is unchanged, so are the literal vmcnt counts.
  const bool grp_b = wave >= 4;
  bf16x8 fr[NS][NF];
  auto seg_load = [&](int k, int g, auto kx_tag) {
    constexpr int KX = decltype(kx_tag)::value;
    const int dx = (KX - 1) * dil;
    const unsigned short(*A)[kBK] = a_buf(g & 1);
    const unsigned short(*B)[kBK] = b_buf(k & 3);
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) load_slice(A, B, dx, ks, fr[ks]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SPREAD) {
      // the A tile of the next group in two instalments (3 + 2 calls behind kx = 0 and kx = 1) instead of all five behind
      // kx = 0: the LOAD segments then carry 5 / 4 / 2 LDS-DMA calls instead of 7 / 2 / 2 against COMPUTE segments of
      // equal length (an LDS-DMA call costs the issuing wave 60-185 cycles: the 7-call segment was the longest of the three)
      if (KX == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) issue_a(g + 1, q);
      } else if (KX == 1) {
#pragma unroll
        for (int q = 3; q < 5; ++q) issue_a(g + 1, q);
      }
    } else if (KX == 0) {
#pragma unroll
      for (int q = 0; q < 5; ++q) issue_a(g + 1, q);
    }
    issue_b(k + 3, 0);
    issue_b(k + 3, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (KX != 1) {
      bool ok[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) ok[i] = (unsigned)(fx[i] + dx) < (unsigned)W;
      if (__builtin_amdgcn_ballot_w64(!(ok[0] && ok[1])) != 0ull) {          // most wavefronts touch no image border column
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) mask_slice(fr[ks], ok);
      }
    }
  };
  auto seg_compute = [&]() {
    // the computing wavefront outranks its partner's LOAD segment at the SIMD's issue arbiter (the role split gives
    // s_setprio something to arbitrate: cdna_hip_programming.md T5)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) mma_slice(fr[ks]);
    __builtin_amdgcn_s_setprio(0);
  };
  // (sched_barrier(0) on both sides: hipcc moves register-only MFMAs across s_barrier and inline-asm waits — the first
  // build of this schedule had the MFMAs of one segment sunk into the fragment reads of the next)
  // Program order of a wavefront's fills in the steady state (g = group, k = 3g + kx):
  //   all-at-once:  L(3g): A(g+1) x5, B(3g+3) x2 | L(3g+1): B(3g+4) x2 | L(3g+2): B(3g+5) x2
  //                 before K-step 3g+1 / 3g+2 the fills of B(3g+1) / B(3g+2) must have landed: 9 younger calls may be in flight;
  //                 before 3g+3: A(g+1) and B(3g+3): 4 younger calls
  //   spread:       L(3g): A(g+1) q0-2, B(3g+3) x2 | L(3g+1): A(g+1) q3-4, B(3g+4) x2 | L(3g+2): B(3g+5) x2
  //                 before 3g+1: B(3g+1) was issued in L(3g-2); younger: L(3g-1) 2 + L(3g) 5 = 7
  //                 before 3g+2: B(3g+2) from L(3g-1); younger: L(3g) 5 + L(3g+1) 4 = 9
  //                 before 3g+3: A(g+1) q4 from L(3g+1); younger: B(3g+4) x2 + L(3g+2) 2 = 4
  auto wait_stage = [&](int kx_next) {                 // this wave's fills of the NEXT K-step have landed
    __builtin_amdgcn_sched_barrier(0);
    if (kx_next == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (SPREAD && kx_next == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  wait_stage(0);
  __builtin_amdgcn_s_barrier();
  if (!grp_b) {
    for (int g = 0; g < n_groups; ++g) {
      seg_load(3 * g + 0, g, std::integral_constant<int, 0>{}); bar(); seg_compute(); wait_stage(1); bar();
      seg_load(3 * g + 1, g, std::integral_constant<int, 1>{}); bar(); seg_compute(); wait_stage(2); bar();
      seg_load(3 * g + 2, g, std::integral_constant<int, 2>{}); bar(); seg_compute(); wait_stage(0); bar();
    }
  } else {
    bar();
    for (int g = 0; g < n_groups; ++g) {
      seg_load(3 * g + 0, g, std::integral_constant<int, 0>{}); wait_stage(1); bar(); seg_compute(); bar();
      seg_load(3 * g + 1, g, std::integral_constant<int, 1>{}); wait_stage(2); bar(); seg_compute(); bar();
      seg_load(3 * g + 2, g, std::integral_constant<int, 2>{}); wait_stage(0); bar(); seg_compute();
      if (g + 1 < n_groups) bar();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const float alpha_v = (F16 && alpha) ? *alpha : 1.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = nt * TN + wn * 64 + j * 32 + (lane & 31);
    const float bv = (bias && n < Cout) ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long m = (long long)mt * TM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M && n < Cout) {
          if constexpr (SPLIT) static_cast<float*>(Yv)[(size_t)m * Cout + n] = acc[i][j][r] + bv;
          else if constexpr (F16) static_cast<float*>(Yv)[(size_t)m * Cout + n] = acc[i][j][r] * alpha_v + bv;
          else static_cast<unsigned short*>(Yv)[(size_t)m * Cout + n] = f2bf_rn(acc[i][j][r] + bv);
        }
      }
  }
}

// w (Cout, k, k, Cin) -> wt (Cin, k, k, Cout) with mirrored taps: the weights of the data-gradient convolution
__global__ __launch_bounds__(256) void k_dgrad_weights(const unsigned short* __restrict__ w, unsigned short* __restrict__ wt,
                                                       int cout, int cin, int taps) {
  __shared__ unsigned short s[64][64 + 2];
  const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tap = blockIdx.z;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int n = i / 64, c = i % 64;
    s[n][c] = (n0 + n < cout && c0 + c < cin) ? w[((size_t)(n0 + n) * taps + tap) * cin + c0 + c] : (unsigned short)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i / 64, n = i % 64;
    if (c0 + c < cin && n0 + n < cout) wt[((size_t)(c0 + c) * taps + (taps - 1 - tap)) * cout + n0 + n] = s[n][c];
  }
}

// v (fp32) -> hi = bf16(v) (round to nearest even), lo = bf16(v - hi): v = hi + lo up to 2^-17 |v|.  8 values per lane.
__global__ __launch_bounds__(256) void k_split_f32(const float* __restrict__ x, long long n, unsigned short* __restrict__ hi,
                                                   unsigned short* __restrict__ lo) {
  const long long n8 = n / 8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned short h[8], l[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      h[k] = f2bf_rn(v[k]);
      const float hv = __uint_as_float((unsigned)h[k] << 16);
      const bool finite = (h[k] & 0x7f80) != 0x7f80;                    // inf / nan stay in the hi plane alone
      l[k] = finite ? f2bf_rn(v[k] - hv) : (unsigned short)0;
    }
    uint4 ph, pl;
    ph.x = h[0] | ((unsigned)h[1] << 16); ph.y = h[2] | ((unsigned)h[3] << 16); ph.z = h[4] | ((unsigned)h[5] << 16); ph.w = h[6] | ((unsigned)h[7] << 16);
    pl.x = l[0] | ((unsigned)l[1] << 16); pl.y = l[2] | ((unsigned)l[3] << 16); pl.z = l[4] | ((unsigned)l[5] << 16); pl.w = l[6] | ((unsigned)l[7] << 16);
    reinterpret_cast<uint4*>(hi)[i] = ph;
    reinterpret_cast<uint4*>(lo)[i] = pl;
  }
  if (blockIdx.x == 0) {                                                 // tail of fewer than 8 values
    const long long i = n8 * 8 + threadIdx.x;
    if (i < n) {
      const unsigned short h = f2bf_rn(x[i]);
      hi[i] = h;
      lo[i] = ((h & 0x7f80) != 0x7f80) ? f2bf_rn(x[i] - __uint_as_float((unsigned)h << 16)) : (unsigned short)0;
    }
  }
}

// ---- TF32-grade form (round 6): fp32 -> IEEE half planes -------------------------------------------------------------------------
// A half has the 11 significant bits of TF32 (what the reference's cuDNN convolutions compute with: tools/train.py:150-153 leaves
// allow_tf32 on) but 5 exponent bits: activations and weights fit as they are; a GRADIENT tensor is multiplied by a power of two that
// brings its largest magnitude just below 2^15 (exact: no rounding is added), and the consuming kernel's epilogue multiplies by the
// inverse.  k_amax_f32: the largest |x| as the bit pattern of a non-negative float (atomicMax: order-independent, deterministic).
__global__ __launch_bounds__(256) void k_amax_f32(const float* __restrict__ x, long long n, unsigned* __restrict__ amax_bits) {
  float m = 0.f;
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(x)[i];
    m = fmaxf(fmaxf(m, fabsf(a.x)), fmaxf(fabsf(a.y), fmaxf(fabsf(a.z), fabsf(a.w))));
  }
  if (blockIdx.x == 0 && n4 * 4 + threadIdx.x < n) m = fmaxf(m, fabsf(x[n4 * 4 + threadIdx.x]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  // ONE atomic per workgroup, and only where it can still raise the value: every wavefront of a 2048-workgroup grid on the same
  // word was 40 us of serialised L2 atomics per call (profiles/round6/step_f16_first.txt: 3.75 ms per step in 81 calls)
  __shared__ float wave_max[4];
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
    const unsigned bits = __float_as_uint(m);
    if (bits > __hip_atomic_load(amax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax_bits, bits);
  }
}

__device__ __forceinline__ unsigned short f2h_rn(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }

// out = half(x * s), s = 2^(15 - e) for amax in [2^(e-1), 2^e) (amax_bits given) or 1; *inv_scale = 1 / s.  8 values per lane.
// Without a scale, finite values beyond the half range SATURATE at +-65504 instead of becoming infinities (TF32 has fp32's
// exponent range: an activation of 1e5 is not an overflow there); infinities and NaNs pass through.
__device__ __forceinline__ float sat_half(float v) {
  const float a = fabsf(v);
  return (a > 65504.f && a < __builtin_huge_valf()) ? copysignf(65504.f, v) : v;
}

__global__ __launch_bounds__(256) void k_cast_f16(const float* __restrict__ x, long long n, const unsigned* __restrict__ amax_bits,
                                                  unsigned short* __restrict__ out, float* __restrict__ inv_scale) {
  float s = 1.f, inv = 1.f;
  if (amax_bits) {
    const unsigned b = *amax_bits;
    if (b != 0u && b < 0x7f800000u) {
      const int e = (int)(b >> 23) - 126;                       // amax < 2^e
      s = ldexpf(1.f, 15 - e);
      inv = ldexpf(1.f, e - 15);
    }
  }
  if (inv_scale && blockIdx.x == 0 && threadIdx.x == 0) *inv_scale = inv;
  const long long n8 = n / 8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned short h[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = f2h_rn(amax_bits ? v[k] * s : sat_half(v[k]));
    uint4 ph;
    ph.x = h[0] | ((unsigned)h[1] << 16); ph.y = h[2] | ((unsigned)h[3] << 16); ph.z = h[4] | ((unsigned)h[5] << 16); ph.w = h[6] | ((unsigned)h[7] << 16);
    reinterpret_cast<uint4*>(out)[i] = ph;
  }
  if (blockIdx.x == 0) {
    const long long i = n8 * 8 + threadIdx.x;
    if (i < n) out[i] = f2h_rn(amax_bits ? x[i] * s : sat_half(x[i]));
  }
}

// The row-shift kernel addresses each operand through ONE buffer descriptor with 32-bit offsets: both planes of a split operand
// must lie within 2 GiB of each other (the wrappers allocate them back to back).
bool rs_addressable(const void* hi, const void* lo, size_t plane_bytes) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(hi), b = lo ? reinterpret_cast<uintptr_t>(lo) : a;
  const uintptr_t span = (a > b ? a - b : b - a) + plane_bytes;
  return span < (1ull << 31);
}

// OMNIHD_CONV_RS_SPREAD=0: the A fill of a group issued at once (read once per process)
bool rs_spread() {
  static const bool on = [] { const char* e = getenv("OMNIHD_CONV_RS_SPREAD"); return !(e && e[0] == '0'); }();
  return on;
}

const unsigned short* igemm_zero_page() {
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

}  // namespace
}  // namespace omnihd

using namespace omnihd;

extern "C" int omnihd_conv_fwd_supported(int batch, int h, int w, int cin, int cout, int ksize, int dil) {
  return batch > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && (ksize == 1 || ksize == 3) && dil >= 1 && cin % 64 == 0 &&
         cout % 8 == 0 && (long long)batch * h * w < (1ll << 31) / 2;
}

extern "C" int omnihd_conv_fwd_bf16(const void* x_nhwc, const void* w_ohwi, const float* bias, void* y_nhwc, int batch,
                                    int h, int w, int cin, int cout, int ksize, int dil, int tile, void* stream) {
  OMNIHD_REQUIRE(omnihd_conv_fwd_supported(batch, h, w, cin, cout, ksize, dil),
                 "conv_fwd: square 1x1 / 3x3 kernel, stride 1, 'same' padding, Cin a multiple of 64, Cout of 8");
  OMNIHD_REQUIRE(x_nhwc && w_ohwi && y_nhwc, "null pointer");
  OMNIHD_REQUIRE(((reinterpret_cast<uintptr_t>(x_nhwc) | reinterpret_cast<uintptr_t>(w_ohwi)) & 15u) == 0, "16-byte alignment");
  const unsigned short* zero_page = igemm_zero_page();
  OMNIHD_REQUIRE(zero_page != nullptr, "could not allocate the zero page");
  hipStream_t st = (hipStream_t)stream;
  const int M = batch * h * w;
  const unsigned short* X = static_cast<const unsigned short*>(x_nhwc);
  const unsigned short* Wt = static_cast<const unsigned short*>(w_ohwi);
  unsigned short* Y = static_cast<unsigned short*>(y_nhwc);
  // tile 0 = choose: 256x128 when that still fills the chip several times over, else 128x128
  const long long big_tiles = (long long)((M + 255) / 256) * ((cout + 127) / 128);
  const bool big = tile == 256 || (tile == 0 && big_tiles >= 2 * kCUs);
  OMNIHD_REQUIRE(rs_addressable(X, nullptr, (size_t)M * cin * 2) && rs_addressable(Wt, nullptr, (size_t)cout * ksize * ksize * cin * 2),
                 "operands of 2 GiB and more are not addressable (32-bit buffer offsets)");
  const bool rs_ok = ksize == 3 && dil <= kHalo;
  if (tile == 300 || tile == 301 || (tile == 0 && rs_ok && big_tiles >= 2 * kCUs)) {
    // 3x3 with row-shift reuse of the activation tile (301: the A fill of a group issued at once, round 3's schedule)
    OMNIHD_REQUIRE(rs_ok, "the row-shift kernel takes 3x3 kernels with dilation <= 8");
    const int tiles_m = (M + 255) / 256, tiles_n = (cout + 127) / 128;
    const int per = (tiles_m * tiles_n + 7) / 8;
    if (tile == 301 || !rs_spread())
      hipLaunchKernelGGL((k_conv_igemm_rs<false>), dim3(8 * per), dim3(512), 0, st, X, Wt, zero_page, bias, (void*)Y, M, h, w, cin, cout, dil,
                         tiles_m, tiles_n, per, (const unsigned short*)nullptr, (const unsigned short*)nullptr);
    else
      hipLaunchKernelGGL((k_conv_igemm_rs<true>), dim3(8 * per), dim3(512), 0, st, X, Wt, zero_page, bias, (void*)Y, M, h, w, cin, cout, dil,
                         tiles_m, tiles_n, per, (const unsigned short*)nullptr, (const unsigned short*)nullptr);
  } else if (tile == 129 || (tile == 0 && !big && (long long)((M + 127) / 128) * ((cout + 127) / 128) >= kCUs)) {
    // 128x128 tile, 2-stage ring (64 KB): two workgroups per CU cover each other's fills — where there are enough tiles for
    // two per CU (scripts/lab/conv_small_tiles.py: 1.4x on the 6 x 64 x 176 1x1 layers, a loss on deep small maps)
    const int tiles_m = (M + 127) / 128, tiles_n = (cout + 127) / 128;
    const int per = (tiles_m * tiles_n + 7) / 8;
    hipLaunchKernelGGL((k_conv_igemm<2, 2, 2, true>), dim3(8 * per), dim3(256), 0, st, X, Wt, zero_page, bias, (void*)Y, M, h, w, cin, cout,
                       ksize, dil, tiles_m, tiles_n, per, (const unsigned short*)nullptr, (const unsigned short*)nullptr);
  } else if (tile == 254) {   // 128x256 tile (2 x 4 wavefronts): half the A traffic per flop, twice the weights'
    const int tiles_m = (M + 127) / 128, tiles_n = (cout + 255) / 256;
    const int per = (tiles_m * tiles_n + 7) / 8;
    hipLaunchKernelGGL((k_conv_igemm<2, 4, 3, true>), dim3(8 * per), dim3(512), 0, st, X, Wt, zero_page, bias, (void*)Y, M, h, w, cin, cout,
                       ksize, dil, tiles_m, tiles_n, per, (const unsigned short*)nullptr, (const unsigned short*)nullptr);
  } else if (big) {
    const int tiles_m = (M + 255) / 256, tiles_n = (cout + 127) / 128;
    const int per = (tiles_m * tiles_n + 7) / 8;
    hipLaunchKernelGGL((k_conv_igemm<4, 2, 3, true>), dim3(8 * per), dim3(512), 0, st, X, Wt, zero_page, bias, (void*)Y, M, h, w, cin, cout,
                       ksize, dil, tiles_m, tiles_n, per, (const unsigned short*)nullptr, (const unsigned short*)nullptr);
  } else {
    const int tiles_m = (M + 127) / 128, tiles_n = (cout + 127) / 128;
    const int per = (tiles_m * tiles_n + 7) / 8;
    hipLaunchKernelGGL((k_conv_igemm<2, 2, 4, true>), dim3(8 * per), dim3(256), 0, st, X, Wt, zero_page, bias, (void*)Y, M, h, w, cin, cout,
                       ksize, dil, tiles_m, tiles_n, per, (const unsigned short*)nullptr, (const unsigned short*)nullptr);
  }
  return check_launch("conv_fwd_bf16");
}

// ---------------------------------------------------------------------------------------------
// Weight images of MANY convolution layers in ONE launch (round 3).  Per layer and step the fp32 master weight (any strides:
// OIHW or channels_last parameters) has to become: its bf16 plane(s) in (Cout,k,k,Cin) memory for the forward kernel and the
// mirrored / transposed plane(s) in (Cin,k,k,Cout) memory for the data gradient — done layer by layer that was a layout copy,
// a split pass and two transposes (4 launches x 76 layers per fp32 step).  One workgroup = a 32 x 32 (cout, cin) tile of one
// layer, all taps; the layer of a workgroup is found in the table by its first-block prefix.
// ---------------------------------------------------------------------------------------------
struct WeightImageEntry {
  const float* src;                 // fp32 master weight
  long long so, si, sy, sx;         // its element strides (cout, cin, ky, kx)
  unsigned short* f_hi;             // (Cout,k,k,Cin) bf16: hi plane (or the plain bf16 image)
  unsigned short* f_lo;             // lo plane, or null
  unsigned short* d_hi;             // (Cin,k,k,Cout) bf16, taps mirrored: data-gradient image, or null
  unsigned short* d_lo;             // its lo plane, or null
  int cout, cin, k, first_block;
};

__global__ __launch_bounds__(256) void k_weight_images(const WeightImageEntry* __restrict__ table, int n_entries) {
  __shared__ float s[32][16][33];       // up to 4 x 4 taps (SECONDFPN's transposed convolutions with kernel == stride 4)
  // binary search for the layer of this workgroup (first_block is increasing; the table has a closing sentinel entry)
  int lo_e = 0, hi_e = n_entries - 1;
  while (lo_e < hi_e) {
    const int mid = (lo_e + hi_e + 1) >> 1;
    if (table[mid].first_block <= (int)blockIdx.x) lo_e = mid; else hi_e = mid - 1;
  }
  WeightImageEntry e = table[lo_e];
  const bool f16 = e.k < 0;                          // round 6: k < 0 asks for IEEE half images (f_hi / d_hi only) of a |k| x |k| kernel
  e.k = f16 ? -e.k : e.k;
  const int taps = e.k * e.k;
  const int tiles_i = (e.cin + 31) / 32;
  const int b = (int)blockIdx.x - e.first_block;
  const int o0 = (b / tiles_i) * 32, i0 = (b % tiles_i) * 32;
  const int tid = threadIdx.x;
  // read order follows the master weight's memory: channels_last parameters (si == 1: what model.to(channels_last) leaves) have
  // the input channel fastest, OIHW ones the tap — consecutive lanes read consecutive floats either way (round 5: the tap-fastest
  // order on channels_last weights read 4 bytes per 4 KB stride: 0.68 ms per step for 66 M parameters)
  const bool cin_fastest = e.si == 1;
  for (int idx = tid; idx < 32 * 32 * taps; idx += 256) {
    int t, i, o;
    if (cin_fastest) { i = idx % 32; t = (idx / 32) % taps; o = idx / (32 * taps); }
    else { t = idx % taps; i = (idx / taps) % 32; o = idx / (taps * 32); }
    float v = 0.f;
    if (o0 + o < e.cout && i0 + i < e.cin)
      v = e.src[(long long)(o0 + o) * e.so + (long long)(i0 + i) * e.si + (long long)(t / e.k) * e.sy + (long long)(t % e.k) * e.sx];
    s[o][t][i] = v;
  }
  __syncthreads();
  auto split = [f16](float v, unsigned short& h, unsigned short& l) {
    if (f16) { h = f2h_rn(v); l = 0; return; }
    h = f2bf_rn(v);
    const float hv = __uint_as_float((unsigned)h << 16);
    l = ((h & 0x7f80) != 0x7f80) ? f2bf_rn(v - hv) : (unsigned short)0;
  };
  // forward image: (o, tap, i), i fastest
  for (int idx = tid; idx < 32 * 32 * taps; idx += 256) {
    const int i = idx % 32, t = (idx / 32) % taps, o = idx / (32 * taps);
    if (o0 + o < e.cout && i0 + i < e.cin) {
      unsigned short h, l;
      split(s[o][t][i], h, l);
      const size_t at = ((size_t)(o0 + o) * taps + t) * e.cin + i0 + i;
      e.f_hi[at] = h;
      if (e.f_lo) e.f_lo[at] = l;
    }
  }
  // data-gradient image: (i, mirrored tap, o), o fastest
  if (e.d_hi) {
    for (int idx = tid; idx < 32 * 32 * taps; idx += 256) {
      const int o = idx % 32, t = (idx / 32) % taps, i = idx / (32 * taps);
      if (o0 + o < e.cout && i0 + i < e.cin) {
        unsigned short h, l;
        split(s[o][t][i], h, l);
        const size_t at = ((size_t)(i0 + i) * taps + (taps - 1 - t)) * e.cout + o0 + o;
        e.d_hi[at] = h;
        if (e.d_lo) e.d_lo[at] = l;
      }
    }
  }
}

// The same for master weights in channels_last memory (input channel fastest: what model.to(channels_last) leaves — every layer
// of the training step): one workgroup = a 64 x 64 (cout, cin) tile of ONE tap.  Reads are 256-byte runs, both images are written
// in 128-byte runs (the all-taps tile above: 128-byte reads, 64-byte writes, 68 KB of LDS = two workgroups per CU; 0.65 ms per fp32
// step for 66 M parameters, round 5 profile), 17 KB of LDS.  first_block counts tiles x taps (mode 1 of omnihd_weight_images).
__global__ __launch_bounds__(256) void k_weight_images_cl(const WeightImageEntry* __restrict__ table, int n_entries) {
  __shared__ float s[64][65];
  int lo_e = 0, hi_e = n_entries - 1;
  while (lo_e < hi_e) {
    const int mid = (lo_e + hi_e + 1) >> 1;
    if (table[mid].first_block <= (int)blockIdx.x) lo_e = mid; else hi_e = mid - 1;
  }
  WeightImageEntry e = table[lo_e];
  const bool f16 = e.k < 0;
  e.k = f16 ? -e.k : e.k;
  const int taps = e.k * e.k;
  const int tiles_i = (e.cin + 63) / 64;
  int b = (int)blockIdx.x - e.first_block;
  const int t = b % taps; b /= taps;
  const int o0 = (b / tiles_i) * 64, i0 = (b % tiles_i) * 64;
  const int tid = threadIdx.x;
  const long long tap_off = (long long)(t / e.k) * e.sy + (long long)(t % e.k) * e.sx;
  for (int idx = tid; idx < 64 * 64; idx += 256) {
    const int i = idx & 63, o = idx >> 6;
    float v = 0.f;
    if (o0 + o < e.cout && i0 + i < e.cin) v = e.src[(long long)(o0 + o) * e.so + tap_off + (long long)(i0 + i) * e.si];
    s[o][i] = v;
  }
  __syncthreads();
  auto split = [f16](float v, unsigned short& h, unsigned short& l) {
    if (f16) { h = f2h_rn(v); l = 0; return; }
    h = f2bf_rn(v);
    const float hv = __uint_as_float((unsigned)h << 16);
    l = ((h & 0x7f80) != 0x7f80) ? f2bf_rn(v - hv) : (unsigned short)0;
  };
  for (int idx = tid; idx < 64 * 64; idx += 256) {            // forward image (o, tap, i): i fastest
    const int i = idx & 63, o = idx >> 6;
    if (o0 + o < e.cout && i0 + i < e.cin) {
      unsigned short h, l;
      split(s[o][i], h, l);
      const size_t at = ((size_t)(o0 + o) * taps + t) * e.cin + i0 + i;
      e.f_hi[at] = h;
      if (e.f_lo) e.f_lo[at] = l;
    }
  }
  if (e.d_hi) {
    for (int idx = tid; idx < 64 * 64; idx += 256) {          // data-gradient image (i, mirrored tap, o): o fastest
      const int o = idx & 63, i = idx >> 6;
      if (o0 + o < e.cout && i0 + i < e.cin) {
        unsigned short h, l;
        split(s[o][i], h, l);
        const size_t at = ((size_t)(i0 + i) * taps + (taps - 1 - t)) * e.cout + o0 + o;
        e.d_hi[at] = h;
        if (e.d_lo) e.d_lo[at] = l;
      }
    }
  }
}

extern "C" int omnihd_weight_images(const void* table_dev, int n_entries, int total_blocks, void* stream) {
  OMNIHD_REQUIRE(n_entries >= 0 && total_blocks >= 0 && (n_entries == 0 || table_dev), "arguments");
  if (n_entries == 0 || total_blocks == 0) return OMNIHD_OK;
  hipLaunchKernelGGL(k_weight_images, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const WeightImageEntry*>(table_dev), n_entries);
  return check_launch("weight_images");
}

extern "C" int omnihd_weight_images_cl(const void* table_dev, int n_entries, int total_blocks, void* stream) {
  OMNIHD_REQUIRE(n_entries >= 0 && total_blocks >= 0 && (n_entries == 0 || table_dev), "arguments");
  if (n_entries == 0 || total_blocks == 0) return OMNIHD_OK;
  hipLaunchKernelGGL(k_weight_images_cl, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const WeightImageEntry*>(table_dev), n_entries);
  return check_launch("weight_images_cl");
}

extern "C" int omnihd_conv_dgrad_weights(const void* w_ohwi, void* wt_ihwo, int cout, int cin, int ksize, void* stream) {
  OMNIHD_REQUIRE(w_ohwi && wt_ihwo && cout > 0 && cin > 0 && ksize >= 1 && ksize <= 4, "arguments");
  hipLaunchKernelGGL(k_dgrad_weights, dim3((cout + 63) / 64, (cin + 63) / 64, ksize * ksize), dim3(256), 0,
                     (hipStream_t)stream, static_cast<const unsigned short*>(w_ohwi), static_cast<unsigned short*>(wt_ihwo),
                     cout, cin, ksize * ksize);
  return check_launch("conv_dgrad_weights");
}

extern "C" int omnihd_split_f32(const float* x, long long n, void* hi, void* lo, void* stream) {
  OMNIHD_REQUIRE(n >= 0 && (n == 0 || (x && hi && lo)), "arguments");
  if (n == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(hi) | reinterpret_cast<uintptr_t>(lo)) & 15u) == 0,
                 "16-byte alignment");
  hipLaunchKernelGGL(k_split_f32, dim3(grid_for(n / 8 + 1, 256 * 2)), dim3(256), 0, (hipStream_t)stream, x, n,
                     static_cast<unsigned short*>(hi), static_cast<unsigned short*>(lo));
  return check_launch("split_f32");
}

extern "C" int omnihd_cast_f16(const float* x, long long n, int scaled, void* out16, float* scratch2, void* stream) {
  OMNIHD_REQUIRE(n >= 0 && (n == 0 || (x && out16)) && (!scaled || scratch2), "arguments");
  if (n == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out16)) & 15u) == 0, "16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  unsigned* amax = nullptr;
  if (scaled) {                                   // scratch2[0]: amax bits (working value), scratch2[1]: the inverse scale (result)
    amax = reinterpret_cast<unsigned*>(scratch2);
    if (scaled == 1) OMNIHD_HIP_TRY(hipMemsetAsync(amax, 0, sizeof(unsigned), st));      // (2: the caller hands a zeroed word)
    if (scaled != 3) {                                                                    // (3: the word already holds max |x|, from x's producer)
      const int blocks = grid_for(n / 4 + 1, 256 * 4);
      hipLaunchKernelGGL(k_amax_f32, dim3(blocks > 4 * kCUs ? 4 * kCUs : blocks), dim3(256), 0, st, x, n, amax);
    }
  }
  hipLaunchKernelGGL(k_cast_f16, dim3(grid_for(n / 8 + 1, 256 * 2)), dim3(256), 0, st, x, n, amax, static_cast<unsigned short*>(out16),
                     scaled ? scratch2 + 1 : (float*)nullptr);
  return check_launch("cast_f16");
}

extern "C" int omnihd_conv_fwd_f16(const void* x16, const void* w16, const float* bias, float* y_nhwc, const float* alpha, int batch,
                                   int h, int w, int cin, int cout, int ksize, int dil, int tile, void* stream) {
  OMNIHD_REQUIRE(omnihd_conv_fwd_supported(batch, h, w, cin, cout, ksize, dil),
                 "conv_fwd_f16: square 1x1 / 3x3 kernel, stride 1, 'same' padding, Cin a multiple of 64, Cout of 8");
  OMNIHD_REQUIRE(x16 && w16 && y_nhwc, "null pointer");
  OMNIHD_REQUIRE(((reinterpret_cast<uintptr_t>(x16) | reinterpret_cast<uintptr_t>(w16)) & 15u) == 0, "16-byte alignment");
  const unsigned short* zero_page = igemm_zero_page();
  OMNIHD_REQUIRE(zero_page != nullptr, "could not allocate the zero page");
  hipStream_t st = (hipStream_t)stream;
  const int M = batch * h * w;
  const unsigned short* X = static_cast<const unsigned short*>(x16);
  const unsigned short* Wt = static_cast<const unsigned short*>(w16);
  const unsigned short* nul = nullptr;
  const long long big_tiles = (long long)((M + 255) / 256) * ((cout + 127) / 128);
  const bool big = tile == 256 || (tile == 0 && big_tiles >= 2 * kCUs);
  OMNIHD_REQUIRE(rs_addressable(X, nullptr, (size_t)M * cin * 2) && rs_addressable(Wt, nullptr, (size_t)cout * ksize * ksize * cin * 2),
                 "operands of 2 GiB and more are not addressable (32-bit buffer offsets)");
  const bool rs_ok = ksize == 3 && dil <= kHalo;
  if (tile == 300 || (tile == 0 && rs_ok && big_tiles >= 2 * kCUs)) {
    OMNIHD_REQUIRE(rs_ok, "the row-shift kernel takes 3x3 kernels with dilation <= 8");
    const int tiles_m = (M + 255) / 256, tiles_n = (cout + 127) / 128;
    const int per = (tiles_m * tiles_n + 7) / 8;
    hipLaunchKernelGGL((k_conv_igemm_rs<true, false, true>), dim3(8 * per), dim3(512), 0, st, X, Wt, zero_page, bias, (void*)y_nhwc, M, h, w,
                       cin, cout, dil, tiles_m, tiles_n, per, nul, nul, alpha);
  } else if (tile == 129 || (tile == 0 && !big && (long long)((M + 127) / 128) * ((cout + 127) / 128) >= kCUs)) {
    const int tiles_m = (M + 127) / 128, tiles_n = (cout + 127) / 128;
    const int per = (tiles_m * tiles_n + 7) / 8;
    hipLaunchKernelGGL((k_conv_igemm<2, 2, 2, true, false, true>), dim3(8 * per), dim3(256), 0, st, X, Wt, zero_page, bias, (void*)y_nhwc,
                       M, h, w, cin, cout, ksize, dil, tiles_m, tiles_n, per, nul, nul, alpha);
  } else if (big) {
    const int tiles_m = (M + 255) / 256, tiles_n = (cout + 127) / 128;
    const int per = (tiles_m * tiles_n + 7) / 8;
    hipLaunchKernelGGL((k_conv_igemm<4, 2, 3, true, false, true>), dim3(8 * per), dim3(512), 0, st, X, Wt, zero_page, bias, (void*)y_nhwc,
                       M, h, w, cin, cout, ksize, dil, tiles_m, tiles_n, per, nul, nul, alpha);
  } else {
    const int tiles_m = (M + 127) / 128, tiles_n = (cout + 127) / 128;
    const int per = (tiles_m * tiles_n + 7) / 8;
    hipLaunchKernelGGL((k_conv_igemm<2, 2, 4, true, false, true>), dim3(8 * per), dim3(256), 0, st, X, Wt, zero_page, bias, (void*)y_nhwc,
                       M, h, w, cin, cout, ksize, dil, tiles_m, tiles_n, per, nul, nul, alpha);
  }
  return check_launch("conv_fwd_f16");
}

extern "C" int omnihd_conv_fwd_split(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                                     float* y_nhwc, int batch, int h, int w, int cin, int cout, int ksize, int dil, int tile,
                                     void* stream) {
  OMNIHD_REQUIRE(omnihd_conv_fwd_supported(batch, h, w, cin, cout, ksize, dil),
                 "conv_fwd_split: square 1x1 / 3x3 kernel, stride 1, 'same' padding, Cin a multiple of 64, Cout of 8");
  OMNIHD_REQUIRE(x_hi && x_lo && w_hi && w_lo && y_nhwc, "null pointer");
  OMNIHD_REQUIRE(((reinterpret_cast<uintptr_t>(x_hi) | reinterpret_cast<uintptr_t>(x_lo) | reinterpret_cast<uintptr_t>(w_hi) |
                   reinterpret_cast<uintptr_t>(w_lo)) & 15u) == 0, "16-byte alignment");
  const unsigned short* zero_page = igemm_zero_page();
  OMNIHD_REQUIRE(zero_page != nullptr, "could not allocate the zero page");
  hipStream_t st = (hipStream_t)stream;
  const int M = batch * h * w;
  const unsigned short *X = static_cast<const unsigned short*>(x_hi), *X2 = static_cast<const unsigned short*>(x_lo);
  const unsigned short *Wt = static_cast<const unsigned short*>(w_hi), *Wt2 = static_cast<const unsigned short*>(w_lo);
  const long long big_tiles = (long long)((M + 255) / 256) * ((cout + 127) / 128);
  const bool big = tile == 256 || (tile == 0 && big_tiles >= 2 * kCUs);
  OMNIHD_REQUIRE(rs_addressable(X, X2, (size_t)M * cin * 2) && rs_addressable(Wt, Wt2, (size_t)cout * ksize * ksize * cin * 2),
                 "the two planes of a split operand must lie within 2 GiB of each other (32-bit buffer offsets)");
  const bool rs_ok = ksize == 3 && dil <= kHalo;
  if (tile == 300 || tile == 301 || (tile == 0 && rs_ok && big_tiles >= 2 * kCUs)) {
    OMNIHD_REQUIRE(rs_ok, "the row-shift kernel takes 3x3 kernels with dilation <= 8");
    const int tiles_m = (M + 255) / 256, tiles_n = (cout + 127) / 128;
    const int per = (tiles_m * tiles_n + 7) / 8;
    if (tile == 301 || !rs_spread())
      hipLaunchKernelGGL((k_conv_igemm_rs<false, true>), dim3(8 * per), dim3(512), 0, st, X, Wt, zero_page, bias, (void*)y_nhwc, M, h, w, cin,
                         cout, dil, tiles_m, tiles_n, per, X2, Wt2);
    else
      hipLaunchKernelGGL((k_conv_igemm_rs<true, true>), dim3(8 * per), dim3(512), 0, st, X, Wt, zero_page, bias, (void*)y_nhwc, M, h, w, cin,
                         cout, dil, tiles_m, tiles_n, per, X2, Wt2);
  } else if (tile == 129 || (tile == 0 && !big && (long long)((M + 127) / 128) * ((cout + 127) / 128) >= kCUs)) {
    const int tiles_m = (M + 127) / 128, tiles_n = (cout + 127) / 128;
    const int per = (tiles_m * tiles_n + 7) / 8;
    hipLaunchKernelGGL((k_conv_igemm<2, 2, 2, true, true>), dim3(8 * per), dim3(256), 0, st, X, Wt, zero_page, bias, (void*)y_nhwc,
                       M, h, w, cin, cout, ksize, dil, tiles_m, tiles_n, per, X2, Wt2);
  } else if (tile == 254) {
    const int tiles_m = (M + 127) / 128, tiles_n = (cout + 255) / 256;
    const int per = (tiles_m * tiles_n + 7) / 8;
    hipLaunchKernelGGL((k_conv_igemm<2, 4, 3, true, true>), dim3(8 * per), dim3(512), 0, st, X, Wt, zero_page, bias, (void*)y_nhwc,
                       M, h, w, cin, cout, ksize, dil, tiles_m, tiles_n, per, X2, Wt2);
  } else if (big) {
    const int tiles_m = (M + 255) / 256, tiles_n = (cout + 127) / 128;
    const int per = (tiles_m * tiles_n + 7) / 8;
    hipLaunchKernelGGL((k_conv_igemm<4, 2, 3, true, true>), dim3(8 * per), dim3(512), 0, st, X, Wt, zero_page, bias, (void*)y_nhwc,
                       M, h, w, cin, cout, ksize, dil, tiles_m, tiles_n, per, X2, Wt2);
  } else {
    const int tiles_m = (M + 127) / 128, tiles_n = (cout + 127) / 128;
    const int per = (tiles_m * tiles_n + 7) / 8;
    hipLaunchKernelGGL((k_conv_igemm<2, 2, 4, true, true>), dim3(8 * per), dim3(256), 0, st, X, Wt, zero_page, bias, (void*)y_nhwc,
                       M, h, w, cin, cout, ksize, dil, tiles_m, tiles_n, per, X2, Wt2);
  }
  return check_launch("conv_fwd_split");
}
