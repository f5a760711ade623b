// Strided and transposed convolutions on the gfx950 matrix cores: the GENERAL form of the implicit GEMM of csrc/conv_igemm.hip.
//
// Where it sits: the stage-entry convolutions of the radar backbone (SECOND, 3x3 stride 2: reference config
// projects/configs/bevfusion_NewScenes/bevfusion.py:62-68, layer_strides=[2,2,2]), the strided 3x3 / 1x1 layers of the image
// backbone (ResNet-50, bevfusion.py:77-85), the up-sampling blocks of SECONDFPN (transposed convolutions with kernel == stride,
// bevfusion.py:69-74) and every small layer whose forward / data gradient used to run on library kernels that accumulate with
// atomics (round 4: two runs of one training step differed by 2-3e-2 in the backbone's weight gradients).  No atomics here:
// every output element is one workgroup's fixed-order sum.
//
// One kernel, a table of CLASSES.  A class is a sub-grid of GEMM rows (b, j, i), j < Hm, i < Wm, with
//     source pixel of tap t   = (j * ss + ty[t], i * ss + tx[t])        (outside [0,Hs) x [0,Ws): zeros)
//     weight K-block of tap t = wt[t]                                     (weights: rows n, K = (tap, source channel))
//     destination pixel       = (j * ds + oy0, i * ds + ox0)
//   forward of conv2d(stride s, padding p, dilation d):  ONE class, ss = s, ty = ky*d - p, wt = ky*k + kx, ds = 1;
//   data gradient of the same convolution:  s*s classes = the residues (py, px) of the INPUT pixel modulo s; class (py, px)
//     keeps the taps with (py + p - ky*d) % s == 0 (1, 2, 2 and 4 of the 9 taps for 3x3 / stride 2), reads the output gradient
//     at (j + (py + p - ky*d)/s, ...) with ss = 1 and writes input pixels (py + s*j, px + s*i); a class without taps (1x1,
//     stride 2: three of four) writes zeros.  No zero-stuffed gradient tensor, no atomics, each input pixel written once.
//   a transposed convolution with kernel == stride is the data gradient of a stride-k convolution (k*k classes of one tap);
//   its data gradient is that convolution's forward.
// Channel counts need only be multiples of 8 (one 16-byte chunk): chunks beyond the source channel count are fetched with an
// out-of-range buffer offset and arrive as zeros, so the head's 16/32/72-channel layers and 3x3 layers on 32 channels take it.
//
// Tile: 128 x 128 per workgroup of 4 wavefronts (2 x 2 quadrants of 64 x 64 as 2 x 2 v_mfma_f32_32x32x16_bf16), K stepped by 64
// through a ring of LDS stages filled by `buffer_load ... lds` (16 B per lane), rows XOR-swizzled on the source chunk and on the
// fragment read, counted vmcnt + raw s_barrier — the narrow-layer schedule of k_conv_igemm.  SPLIT: fp32-grade result from
// hi / lo bf16 planes (x*w = hi*hi + hi*lo + lo*hi, fp32 accumulation), fp32 output.
#include "common.h"

namespace omnihd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_ptr_t;

constexpr int kBK = 64;
constexpr int kMaxTaps = 16;
constexpr int kMaxClasses = 16;

struct GenClass {
  int Hm, Wm;              // rows of the class: (b, j < Hm, i < Wm)
  int oy0, ox0;            // destination pixel = (j * ds + oy0, i * ds + ox0)
  int n_taps;
  int tap[kMaxTaps];       // per tap: ty (bits 0-11, signed) | tx (bits 12-23, signed) | wt (bits 24-31) — whole dwords: scalar loads
};

__host__ __device__ inline int pack_tap(int ty, int tx, int wt) { return (ty & 0xfff) | ((tx & 0xfff) << 12) | (wt << 24); }

struct GenGeom {
  int n_classes, tiles_m, tiles_n, tiles_per_xcd;
  int B, Hs, Ws, Cs;       // source tensor (B, Hs, Ws, Cs)
  int Hd, Wd, N;           // destination tensor (B, Hd, Wd, N)
  int ss, ds;              // source / destination pixel strides of a row step
  int taps_total;          // K blocks per weight row: K = taps_total * Cs
  GenClass cls[kMaxClasses];
};

__device__ __forceinline__ unsigned short f2bf_rn(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <int STAGES, bool SPLIT>
__global__ __launch_bounds__(256) void k_conv_gen(const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wt,
                                                  const unsigned short* __restrict__ X2, const unsigned short* __restrict__ Wt2,
                                                  const float* __restrict__ bias, void* __restrict__ Yv, const GenGeom g) {
  constexpr int kCS = SPLIT ? 32 : 64;            // source channels per K-step
  constexpr int TM = 128, TN = 128, NW = 4, WN = 2;
  constexpr int A_CALLS = TM / (8 * NW), B_CALLS = TN / (8 * NW);     // 4 + 4 LDS-DMA calls (8 rows x 128 B) per wavefront and stage
  constexpr int CALLS = A_CALLS + B_CALLS;
  // one LDS object only (a second one makes hipcc drain the DMA queue before every ds_read); the destination offsets of the
  // tile's 128 rows live behind the ring
  __shared__ __attribute__((aligned(16))) unsigned short sm[STAGES * (TM + TN) * kBK + TM * 2];
  auto stage_row = [&](int stage, int row) { return sm + ((size_t)stage * (TM + TN) + row) * kBK; };
  unsigned* row_dst = reinterpret_cast<unsigned*>(sm + STAGES * (TM + TN) * kBK);

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int t = xcd * g.tiles_per_xcd + slot;
  const int per_class = g.tiles_m * g.tiles_n;
  if (slot >= g.tiles_per_xcd || t >= per_class * g.n_classes) return;
  const int ci = t / per_class, rem = t - ci * per_class;
  const int nt = rem / g.tiles_m, mt = rem - nt * g.tiles_m;
  const GenClass& cl = g.cls[ci];
  const int M = g.B * cl.Hm * cl.Wm;
  if (mt * TM >= M) return;                        // a smaller class than the largest one

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_taps = cl.n_taps;
  const int K = g.taps_total * g.Cs;

  constexpr unsigned kOOB = 0x80000000u;
  const unsigned short* xbase = (SPLIT && X2 < X) ? X2 : X;
  const unsigned short* wbase = (SPLIT && Wt2 < Wt) ? Wt2 : Wt;
  const size_t x_plane = (size_t)g.B * g.Hs * g.Ws * g.Cs * 2, w_plane = (size_t)g.N * K * 2;
  const unsigned x_hi_off = (unsigned)((const char*)X - (const char*)xbase), w_hi_off = (unsigned)((const char*)Wt - (const char*)wbase);
  const unsigned x_lo_off = SPLIT ? (unsigned)((const char*)X2 - (const char*)xbase) : 0u;
  const unsigned w_lo_off = SPLIT ? (unsigned)((const char*)Wt2 - (const char*)wbase) : 0u;
  const unsigned x_bytes = (unsigned)((SPLIT ? (x_lo_off > x_hi_off ? x_lo_off : x_hi_off) : 0u) + x_plane);
  const unsigned w_bytes = (unsigned)((SPLIT ? (w_lo_off > w_hi_off ? w_lo_off : w_hi_off) : 0u) + w_plane);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, (int)w_bytes, 0x00020000);

  const int lr = lane >> 3;                       // row inside an 8-row call
  const int pos = lane & 7;                       // 16-byte slot inside the 128-byte LDS row
  const int c_even = pos ^ ((lane >> 4) & 7);     // source chunk for even calls; odd calls use c_even ^ 4 (the read-side swizzle)
  // per call: pixel coordinates of the row at tap offset (0, 0) and the byte offset of (pixel, chunk) there (mod 2^32: the
  // pixel may lie outside the image, the tap offset brings it back inside or the lane is masked)
  int a_y[A_CALLS], a_x[A_CALLS], a_ch[A_CALLS];
  unsigned a_off[A_CALLS];
#pragma unroll
  for (int i = 0; i < A_CALLS; ++i) {
    const int c = (i & 1) ? (c_even ^ 4) : c_even;
    const int cc = SPLIT ? (c & 3) : c;
    a_ch[i] = cc * 8;
    const int row = wave * (8 * A_CALLS) + 8 * i + lr;
    const int m = mt * TM + row;
    if (m < M) {
      const int ii = m % cl.Wm, r = m / cl.Wm;
      const int jj = r % cl.Hm, b = r / cl.Hm;
      a_y[i] = jj * g.ss; a_x[i] = ii * g.ss;
      a_off[i] = ((SPLIT && c >= 4) ? x_lo_off : x_hi_off) +
                 (unsigned)(((long long)(b * g.Hs + a_y[i]) * g.Ws + a_x[i]) * g.Cs * 2) + cc * 16;
      if (pos == 0) row_dst[row] = (unsigned)((b * g.Hd + jj * g.ds + cl.oy0) * g.Wd + ii * g.ds + cl.ox0);
    } else {
      a_y[i] = -(1 << 24); a_x[i] = 0; a_off[i] = kOOB;
      if (pos == 0) row_dst[row] = 0xffffffffu;
    }
  }
  unsigned b_off[B_CALLS];
  int b_ch[B_CALLS];
#pragma unroll
  for (int i = 0; i < B_CALLS; ++i) {
    const int call = A_CALLS + i;
    const int c = (call & 1) ? (c_even ^ 4) : c_even;
    const int cc = SPLIT ? (c & 3) : c;
    b_ch[i] = cc * 8;
    const int n = nt * TN + wave * (8 * B_CALLS) + 8 * i + lr;
    b_off[i] = n < g.N ? ((SPLIT && c >= 4) ? w_lo_off : w_hi_off) + (unsigned)((size_t)n * K * 2) + cc * 16 : kOOB;
  }

  int k_tap = 0, k_c = 0;                         // position of the NEXT K-step to issue
  int tapw = cl.tap[0];                           // its tap word (a scalar load from the kernel arguments, one K-step ahead of its use)
  auto issue_call = [&](int stage, bool real, int call) {
    const int dy = (tapw << 20) >> 20, dx = (tapw << 8) >> 20, wtap = (int)((unsigned)tapw >> 24);
    if (call < A_CALLS) {
      const int i = call;
      const bool ok = real && (unsigned)(a_y[i] + dy) < (unsigned)g.Hs && (unsigned)(a_x[i] + dx) < (unsigned)g.Ws &&
                      k_c + a_ch[i] < g.Cs;
      const int goff = ((dy * g.Ws + dx) * g.Cs + k_c) * 2;                  // wave-uniform, may be negative
      const unsigned voff = ok ? a_off[i] + (unsigned)goff : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_ptr_t*)stage_row(stage, wave * (8 * A_CALLS) + 8 * i), 16, voff, 0, 0, 0);
    } else {
      const int i = call - A_CALLS;
      const bool ok = real && k_c + b_ch[i] < g.Cs;
      const unsigned goff = (unsigned)((wtap * g.Cs + k_c) * 2);             // wave-uniform
      const unsigned voff = ok ? b_off[i] + goff : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t*)stage_row(stage, TM + wave * (8 * B_CALLS) + 8 * i), 16, voff, 0, 0, 0);
    }
  };
  // channels OUTER, taps INNER (the taps of one channel slice read neighbouring lines: L2 hits)
  auto advance = [&]() {
    if (++k_tap >= n_taps) { k_tap = 0; k_c += kCS; }
    tapw = cl.tap[k_tap];
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
  const int n_steps = n_taps * ((g.Cs + kCS - 1) / kCS);

  constexpr int NS = kCS / 16;
  constexpr int NF = SPLIT ? 8 : 4;
  auto load_slice = [&](int stage, int ks, bf16x8 (&f)[NF]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ra = wm * 64 + i * 32 + frow;
      const int rb = wn * 64 + i * 32 + frow;
      const int c = ks * 2 + fhalf;
      f[i] = *reinterpret_cast<const bf16x8*>(stage_row(stage, ra) + ((c ^ ((ra >> 1) & 7)) * 8));
      if constexpr (SPLIT) {
        f[2 + i] = *reinterpret_cast<const bf16x8*>(stage_row(stage, ra) + (((c + 4) ^ ((ra >> 1) & 7)) * 8));
        f[4 + i] = *reinterpret_cast<const bf16x8*>(stage_row(stage, TM + rb) + ((c ^ ((rb >> 1) & 7)) * 8));
        f[6 + i] = *reinterpret_cast<const bf16x8*>(stage_row(stage, TM + rb) + (((c + 4) ^ ((rb >> 1) & 7)) * 8));
      } else {
        f[2 + i] = *reinterpret_cast<const bf16x8*>(stage_row(stage, TM + rb) + ((c ^ ((rb >> 1) & 7)) * 8));
      }
    }
  };
  auto mma_slice = [&](const bf16x8 (&f)[NF]) {
    if constexpr (SPLIT) {
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
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[i], f[2 + j], acc[i][j], 0, 0, 0);
    }
  };
  auto wait_stage = [&]() {   // this wave's fills of the next K-step have landed: (STAGES-2) younger K-steps may be outstanding
    __builtin_amdgcn_sched_barrier(0);
    if (STAGES == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (STAGES == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  static_assert(CALLS == 8, "the literal vmcnt counts above assume 8 calls per wavefront and K-step");

  // prologue: STAGES-1 K-steps in flight (steps beyond the last one are dummy fills of zeros: the counts stay literal)
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) {
#pragma unroll
    for (int call = 0; call < CALLS; ++call) issue_call(s, s < n_steps, call);
    advance();
  }
  bf16x8 fr[NS][NF];
  int stage = 0;
  for (int step = 0; step < n_steps; ++step) {
    wait_stage();
    __builtin_amdgcn_s_barrier();
    const int fill = (stage + STAGES - 1) % STAGES;             // its buffer was last read before this barrier
    const bool fill_real = step + STAGES - 1 < n_steps;
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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the dummy tail loads before the epilogue
  __syncthreads();                                   // (row_dst was written before the first barrier of the loop; n_steps may be 0)

  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31 (output channel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = nt * TN + wn * 64 + j * 32 + (lane & 31);
    const float bv = (bias && n < g.N) ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const unsigned d = row_dst[row];
        if (d != 0xffffffffu && n < g.N) {
          if constexpr (SPLIT) static_cast<float*>(Yv)[(size_t)d * g.N + n] = acc[i][j][r] + bv;
          else static_cast<unsigned short*>(Yv)[(size_t)d * g.N + n] = f2bf_rn(acc[i][j][r] + bv);
        }
      }
  }
}

int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// classes of one pass; returns false when the geometry does not fit the tables
bool build_geom(int mode, int batch, int h, int w, int cin, int ho, int wo, int cout, int k, int s, int p, int d, GenGeom& g) {
  g = GenGeom();
  const int taps = k * k;
  if (taps > kMaxTaps || s < 1 || s * s > kMaxClasses || k < 1 || d < 1 || p < 0) return false;
  g.B = batch;
  g.taps_total = taps;
  g.tiles_m = 0;
  if (mode == 0) {             // forward: source x (B,h,w,cin) -> y (B,ho,wo,cout)
    g.Hs = h; g.Ws = w; g.Cs = cin; g.Hd = ho; g.Wd = wo; g.N = cout; g.ss = s; g.ds = 1;
    g.n_classes = 1;
    GenClass& c = g.cls[0];
    c.Hm = ho; c.Wm = wo; c.oy0 = c.ox0 = 0; c.n_taps = taps;
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx) {
        const int t = ky * k + kx;
        c.tap[t] = pack_tap(ky * d - p, kx * d - p, t);
      }
  } else {                     // data gradient: source gout (B,ho,wo,cout) -> gx (B,h,w,cin); weight rows = cin, taps mirrored
    g.Hs = ho; g.Ws = wo; g.Cs = cout; g.Hd = h; g.Wd = w; g.N = cin; g.ss = 1; g.ds = s;
    int n = 0;
    for (int py = 0; py < s; ++py)
      for (int px = 0; px < s; ++px) {
        const int Hm = (h - py + s - 1) / s, Wm = (w - px + s - 1) / s;
        if (py >= h || px >= w || Hm <= 0 || Wm <= 0) continue;
        GenClass& c = g.cls[n++];
        c.Hm = Hm; c.Wm = Wm; c.oy0 = py; c.ox0 = px; c.n_taps = 0;
        for (int ky = 0; ky < k; ++ky) {
          const int ny = py + p - ky * d;
          if (ny - floordiv(ny, s) * s != 0) continue;
          for (int kx = 0; kx < k; ++kx) {
            const int nx = px + p - kx * d;
            if (nx - floordiv(nx, s) * s != 0) continue;
            const int q = c.n_taps++;
            c.tap[q] = pack_tap(floordiv(ny, s), floordiv(nx, s), taps - 1 - (ky * k + kx));
          }
        }
      }
    g.n_classes = n;
  }
  for (int i = 0; i < g.n_classes; ++i) {
    const long long m = (long long)batch * g.cls[i].Hm * g.cls[i].Wm;
    const int tm = (int)((m + 127) / 128);
    if (tm > g.tiles_m) g.tiles_m = tm;
  }
  g.tiles_n = (g.N + 127) / 128;
  g.tiles_per_xcd = (g.tiles_m * g.tiles_n * g.n_classes + 7) / 8;
  return g.n_classes > 0 && g.tiles_m > 0;
}

bool addressable(const void* hi, const void* lo, size_t plane_bytes) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(hi), b = lo ? reinterpret_cast<uintptr_t>(lo) : a;
  const uintptr_t span = (a > b ? a - b : b - a) + plane_bytes;
  return span < (1ull << 31);
}

}  // namespace
}  // namespace omnihd

using namespace omnihd;

extern "C" int omnihd_conv_gen_supported(int mode, int batch, int h, int w, int cin, int ho, int wo, int cout, int ksize, int stride,
                                         int pad, int dil) {
  if (!(mode == 0 || mode == 1) || batch <= 0 || h <= 0 || w <= 0 || cin <= 0 || ho <= 0 || wo <= 0 || cout <= 0) return 0;
  if (ksize < 1 || ksize * ksize > kMaxTaps || stride < 1 || stride * stride > kMaxClasses || pad < 0 || dil < 1) return 0;
  if (ho != (h + 2 * pad - dil * (ksize - 1) - 1) / stride + 1 || wo != (w + 2 * pad - dil * (ksize - 1) - 1) / stride + 1) return 0;
  const int cs = mode == 0 ? cin : cout;            // source channels: whole 16-byte chunks
  if (cs % 8 != 0) return 0;
  if (pad > 2000 || dil * ksize > 2000) return 0;         // tap offsets are 12-bit signed fields
  const long long src = (long long)batch * (mode == 0 ? (long long)h * w : (long long)ho * wo) * cs * 2;
  const long long dst_px = (long long)batch * (mode == 0 ? (long long)ho * wo : (long long)h * w);
  const long long wbytes = (long long)(mode == 0 ? cout : cin) * ksize * ksize * cs * 2;
  return src < (1ll << 30) && wbytes < (1ll << 30) && dst_px < (1ll << 31) - 1;
}

extern "C" int omnihd_conv_gen(int mode, const void* src_hi, const void* src_lo, const void* w_hi, const void* w_lo, const float* bias,
                               void* dst, int batch, int h, int w, int cin, int ho, int wo, int cout, int ksize, int stride, int pad,
                               int dil, void* stream) {
  OMNIHD_REQUIRE(omnihd_conv_gen_supported(mode, batch, h, w, cin, ho, wo, cout, ksize, stride, pad, dil),
                 "conv_gen: mode 0 (forward) / 1 (data gradient), square kernel of <= 16 taps, stride^2 <= 16, source channels a "
                 "multiple of 8, (ho, wo) = the convolution's output size, operands below 1 GiB");
  OMNIHD_REQUIRE(src_hi && w_hi && dst && ((src_lo == nullptr) == (w_lo == nullptr)), "null pointer / both or neither lo plane");
  OMNIHD_REQUIRE(((reinterpret_cast<uintptr_t>(src_hi) | reinterpret_cast<uintptr_t>(src_lo) | reinterpret_cast<uintptr_t>(w_hi) |
                   reinterpret_cast<uintptr_t>(w_lo)) & 15u) == 0, "16-byte alignment");
  GenGeom g;
  OMNIHD_REQUIRE(build_geom(mode, batch, h, w, cin, ho, wo, cout, ksize, stride, pad, dil, g), "conv_gen: geometry");
  const bool split = src_lo != nullptr;
  const unsigned short *X = static_cast<const unsigned short*>(src_hi), *X2 = static_cast<const unsigned short*>(src_lo);
  const unsigned short *Wt = static_cast<const unsigned short*>(w_hi), *Wt2 = static_cast<const unsigned short*>(w_lo);
  OMNIHD_REQUIRE(addressable(X, X2, (size_t)g.B * g.Hs * g.Ws * g.Cs * 2) && addressable(Wt, Wt2, (size_t)g.N * g.taps_total * g.Cs * 2),
                 "the two planes of a split operand must lie within 2 GiB of each other (32-bit buffer offsets)");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = 8 * g.tiles_per_xcd;
  // two workgroups per CU (2-stage ring, 64 KB) cover each other's fills where there are tiles for them; a 4-stage ring otherwise
  const bool two_per_cu = (long long)g.tiles_m * g.tiles_n * g.n_classes >= kCUs;
  if (split) {
    if (two_per_cu) hipLaunchKernelGGL((k_conv_gen<2, true>), dim3(blocks), dim3(256), 0, st, X, Wt, X2, Wt2, bias, dst, g);
    else hipLaunchKernelGGL((k_conv_gen<4, true>), dim3(blocks), dim3(256), 0, st, X, Wt, X2, Wt2, bias, dst, g);
  } else {
    if (two_per_cu) hipLaunchKernelGGL((k_conv_gen<2, false>), dim3(blocks), dim3(256), 0, st, X, Wt, X2, Wt2, bias, dst, g);
    else hipLaunchKernelGGL((k_conv_gen<4, false>), dim3(blocks), dim3(256), 0, st, X, Wt, X2, Wt2, bias, dst, g);
  }
  return check_launch("conv_gen");
}
