// Weight gradient of a convolution straight from the NHWC operands: no pixel-major staging pass (round 5).
//
// Where it sits: the weight gradients of the small and middle-sized layers of the detector — ResNet-50's bottlenecks
// (reference config projects/configs/bevfusion_NewScenes/bevfusion.py:77-85), FPN / FPNC adapters, DepthNet's 3x3 layers
// (bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:563-609), SECOND / SECONDFPN (bevfusion.py:62-74), the anchor head —
// 58 geometries per step on which the staged chain of csrc/conv_wgrad.hip (two pixel-major staging launches per plane pair + GEMM
// + slab sum) lost to the library's fp32 kernels, which cost 4-5 launches each and accumulate with atomics.
//
//   dW[n][tap][c] = sum over output pixels m of  G[m][n] * X[src_tap(m)][c]
//
// is a GEMM whose reduction index is the PIXEL — the slow dimension of both NHWC operands.  csrc/conv_wgrad.hip re-lays both operands
// pixel-major first; here the tiles go into LDS as they lie in memory ([pixel][channel], LDS-DMA, 256-byte rows) and the MFMA
// fragments — 8 consecutive PIXELS of one channel per lane — are fetched with gfx950's transposing LDS read
// `ds_read_b64_tr_b16`: each 16-lane group reads a [4 pixel][16 channel] block (lane i supplies the address of pixel i/4, channels
// 4*(i%4)..+3) and lane i receives channel i of the 4 pixels (checked on the device: scripts/micro/tr_read.hip, 0 mismatches).
// A workgroup computes a 128 x 128 tile of (Cout, Cin) for ONE tap over its share of the pixels (split-K, fp32 slabs summed in a
// fixed order: deterministic).  The source pixel of every tile row is computed by the lane that fetches it (any stride, padding,
// dilation); a source pixel outside the image is an out-of-range buffer offset and arrives as a row of zeros, so no fragment is
// ever masked.  SPLIT: fp32-grade sums from hi / lo planes (G_hi X_hi + G_hi X_lo + G_lo X_hi).
// Two LDS stages (32 / 64 KB per workgroup): the next K-step is in flight while this one multiplies, and 2-4 workgroups per CU cover
// each other's fills.  (A deeper ring does not pay with the transposing read as a compiler builtin: hipcc puts `s_waitcnt vmcnt(0)`
// in front of every `ds_read_b64_tr_b16` while an LDS-DMA is pending — it does not for plain LDS loads — so every fill is drained at
// the top of a K-step anyway; ISA checked.  Inline-asm reads with hand-counted lgkmcnt would lift that.)
#include "common.h"

namespace omnihd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int kKP = 32;          // pixels per K-step
constexpr int kT = 128;          // tile edge (channels)

struct WgNhwcArgs {
  int B, H, W, Cin, Ho, Wo, Cout;
  int k, stride, pad, dil;
  int tiles_n, tiles_c, n_split, px_per_split;   // px_per_split: a multiple of kKP
  long long slab_stride;                          // floats between the slabs of consecutive splits (0: n_split == 1, dst = dw)
};

template <bool SPLIT, int STAGES>
__global__ __launch_bounds__(256) void k_wgrad_nhwc(const unsigned short* __restrict__ G, const unsigned short* __restrict__ G2,
                                                    const unsigned short* __restrict__ X, const unsigned short* __restrict__ X2,
                                                    float* __restrict__ dst, const WgNhwcArgs a) {
  constexpr int PLANES = SPLIT ? 2 : 1;
  constexpr int TILE = kKP * kT;                                  // elements of one [32 pixel][128 channel] tile (8 KB)
  constexpr int STAGE = 2 * PLANES * TILE;                        // G_hi [, G_lo], X_hi [, X_lo]
  constexpr int CALLS = 4 * PLANES;                               // LDS-DMA calls (1 KB = 4 pixel rows) per wavefront and K-step
  __shared__ __attribute__((aligned(16))) unsigned short sm[STAGES * STAGE];

  const int taps = a.k * a.k;
  int t = blockIdx.x;
  const int sp = t % a.n_split; t /= a.n_split;
  const int tap = t % taps; t /= taps;
  const int ct = t % a.tiles_c, nt = t / a.tiles_c;
  const int ky = tap / a.k, kx = tap - ky * a.k;
  const int M = a.B * a.Ho * a.Wo;
  const int m_begin = sp * a.px_per_split;
  const int m_end = min(M, m_begin + a.px_per_split);
  const int n_steps = m_begin < m_end ? (m_end - m_begin + kKP - 1) / kKP : 0;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = nt * kT, c0 = ct * kT;

  constexpr unsigned kOOB = 0x80000000u;
  const unsigned g_bytes = (unsigned)((size_t)M * a.Cout * 2);
  const unsigned x_bytes = (unsigned)((size_t)a.B * a.H * a.W * a.Cin * 2);
  const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)G, 0, (int)g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t g2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(SPLIT ? G2 : G), 0, (int)g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(SPLIT ? X2 : X), 0, (int)x_bytes, 0x00020000);

  // ---- loader state: this lane fetches chunk q (16 bytes = 8 channels) of the tile rows p0 = 4 * wave + lane / 16 and p0 + 16 -----
  const int q = lane & 15;
  const int prow = 4 * wave + (lane >> 4);
  const bool g_chan_ok = n0 + q * 8 < a.Cout, x_chan_ok = c0 + q * 8 < a.Cin;
  int ox[2], oy[2], ob[2];                        // output pixel of the two rows at the NEXT K-step to issue
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int m = m_begin + prow + 16 * r;
    ox[r] = m % a.Wo;
    const int rest = m / a.Wo;
    oy[r] = rest % a.Ho;
    ob[r] = rest / a.Ho;
  }
  int m_issue = m_begin;                          // first pixel of the next K-step to issue
  auto issue = [&](int stage, bool real) {
    unsigned short* base = sm + stage * STAGE;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int p = prow + 16 * r;
      const int m = m_issue + p;
      const bool live = real && m < m_end;
      // G row: output pixel m, channels n0 + 8 q ..
      const unsigned goff = (live && g_chan_ok) ? (unsigned)(((size_t)m * a.Cout + n0 + q * 8) * 2) : kOOB;
      // X row: the tap's source pixel of output pixel m (outside the image: zeros)
      const int iy = oy[r] * a.stride - a.pad + ky * a.dil, ix = ox[r] * a.stride - a.pad + kx * a.dil;
      const bool inside = live && x_chan_ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const unsigned xoff = inside ? (unsigned)((((size_t)ob[r] * a.H + iy) * a.W + ix) * a.Cin * 2 + (c0 + q * 8) * 2) : kOOB;
      // LDS destination of the call: 4 rows x 256 B starting at row 4 * wave + 16 r (lane-linear: row lane / 16, chunk lane % 16)
      unsigned short* gd = base + (4 * wave + 16 * r) * kT;
      unsigned short* xd = base + PLANES * TILE + (4 * wave + 16 * r) * kT;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(g_rsrc, (lds_ptr_t*)gd, 16, goff, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_ptr_t*)xd, 16, xoff, 0, 0, 0);
      if constexpr (SPLIT) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(g2_rsrc, (lds_ptr_t*)(gd + TILE), 16, goff, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x2_rsrc, (lds_ptr_t*)(xd + TILE), 16, xoff, 0, 0, 0);
      }
    }
    // advance the two rows by one K-step of pixels
    m_issue += kKP;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      ox[r] += kKP;
      while (ox[r] >= a.Wo) {
        ox[r] -= a.Wo;
        if (++oy[r] == a.Ho) { oy[r] = 0; ++ob[r]; }
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- fragment reads: wave (wm, wn) owns rows n = wm*64 .. +63 of the tile's Cout range and c = wn*64 .. +63 of its Cin range ----
  const int wm = wave >> 1, wn = wave & 1;
  const int grp = lane >> 4, li = lane & 15;
  // byte offset inside a [32][128] tile of this lane's 8-byte piece for slice 0, first half (pixels 0-3 of the lane's 8), block 0:
  // pixel 8 * (grp / 2) + li / 4, channel 16 * (grp % 2) + 4 * (li % 4)
  const int frag_px = 8 * (grp >> 1) + (li >> 2);
  const int frag_ch = 16 * (grp & 1) + 4 * (li & 3);
  auto tr8 = [&](const unsigned short* tile, int slice, int ch_base) {
    // 8 pixels (16 * slice + 8 * (lane / 32) ..+7) of channel ch_base + lane % 32, as one MFMA operand
    const unsigned short* p0 = tile + (16 * slice + frag_px) * kT + ch_base + frag_ch;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * kT));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
  };
  auto wait_stage = [&]() {   // this wave's fills of the next K-step have landed: (STAGES - 2) younger K-steps may be outstanding
    __builtin_amdgcn_sched_barrier(0);
    if (STAGES == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (CALLS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // STAGES == 3
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  static_assert(STAGES == 2 || STAGES == 3, "literal vmcnt counts");

#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) issue(s, s < n_steps);
  int stage = 0;
  for (int step = 0; step < n_steps; ++step) {
    wait_stage();
    __builtin_amdgcn_s_barrier();
    const unsigned short* base = sm + stage * STAGE;
    const unsigned short* g_hi = base;
    const unsigned short* x_hi = base + PLANES * TILE;
    bf16x8 fa[2][2], fb[2][2], fa2[2][2], fb2[2][2];                  // [slice][block]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[s][i] = tr8(g_hi, s, wm * 64 + i * 32);
        fb[s][i] = tr8(x_hi, s, wn * 64 + i * 32);
        if constexpr (SPLIT) {
          fa2[s][i] = tr8(g_hi + TILE, s, wm * 64 + i * 32);
          fb2[s][i] = tr8(x_hi + TILE, s, wn * 64 + i * 32);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
    issue((stage + STAGES - 1) % STAGES, step + STAGES - 1 < n_steps);   // its buffer was last read before this barrier
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if constexpr (SPLIT) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa2[s][i], fb[s][j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][i], fb2[s][j], acc[i][j], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][i], fb[s][j], acc[i][j], 0, 0, 0);
    }
    stage = (stage + 1) % STAGES;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // drain the dummy tail fills before the epilogue

  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31 (B row = input channel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float* out = dst + (size_t)sp * a.slab_stride;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int c = c0 + wn * 64 + j * 32 + (lane & 31);
        if (n < a.Cout && c < a.Cin) out[((size_t)n * taps + tap) * a.Cin + c] = acc[i][j][r];
      }
}

__global__ __launch_bounds__(256) void k_sum_slabs_nhwc(const float* __restrict__ slab, int n_split, size_t n, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float s = slab[i];
    for (int k = 1; k < n_split; ++k) s += slab[(size_t)k * n + i];           // fixed order: run-to-run identical
    out[i] = s;
  }
}

bool nhwc_plan(int batch, int h, int w, int cin, int ho, int wo, int cout, int k, int stride, int pad, int dil, WgNhwcArgs* a) {
  if (batch <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || ho <= 0 || wo <= 0) return false;
  if (k < 1 || k > 4 || stride < 1 || dil < 1 || pad < 0 || cin % 8 || cout % 8) return false;
  if (ho != (h + 2 * pad - dil * (k - 1) - 1) / stride + 1 || wo != (w + 2 * pad - dil * (k - 1) - 1) / stride + 1) return false;
  const long long M = (long long)batch * ho * wo;
  if (M * cout * 2 >= (1ll << 31) || (long long)batch * h * w * cin * 2 >= (1ll << 31)) return false;    // 32-bit buffer offsets
  a->B = batch; a->H = h; a->W = w; a->Cin = cin; a->Ho = ho; a->Wo = wo; a->Cout = cout;
  a->k = k; a->stride = stride; a->pad = pad; a->dil = dil;
  a->tiles_n = (cout + kT - 1) / kT; a->tiles_c = (cin + kT - 1) / kT;
  const long long tiles = (long long)a->tiles_n * a->tiles_c * k * k;
  const int steps = (int)((M + kKP - 1) / kKP);
  long long sp = (3 * kCUs + tiles - 1) / tiles;                 // aim at >= 3 workgroups per CU
  const long long max_sp = steps / 4 > 0 ? steps / 4 : 1;         // at least 4 K-steps per split
  if (sp > max_sp) sp = max_sp;
  if (sp > 64) sp = 64;
  if (sp < 1) sp = 1;
  const int steps_per = (int)((steps + sp - 1) / sp);
  a->n_split = (steps + steps_per - 1) / steps_per;               // no empty split
  a->px_per_split = steps_per * kKP;
  a->slab_stride = a->n_split > 1 ? (long long)cout * k * k * cin : 0;
  return true;
}

}  // namespace
}  // namespace omnihd

using namespace omnihd;

extern "C" size_t omnihd_conv_wgrad_nhwc_workspace_bytes(int batch, int h, int w, int cin, int ho, int wo, int cout, int ksize,
                                                         int stride, int pad, int dil) {
  WgNhwcArgs a;
  if (!nhwc_plan(batch, h, w, cin, ho, wo, cout, ksize, stride, pad, dil, &a)) return 0;
  return 256 + (a.n_split > 1 ? align_up((size_t)a.n_split * cout * ksize * ksize * cin * sizeof(float), 256) : 0);
}

extern "C" int omnihd_conv_wgrad_nhwc(const void* x_hi, const void* x_lo, const void* g_hi, const void* g_lo, float* dw, int batch,
                                      int h, int w, int cin, int ho, int wo, int cout, int ksize, int stride, int pad, int dil,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  WgNhwcArgs a;
  OMNIHD_REQUIRE(nhwc_plan(batch, h, w, cin, ho, wo, cout, ksize, stride, pad, dil, &a),
                 "conv_wgrad_nhwc: square kernel <= 4x4, channels multiples of 8, consistent output size, operands below 2 GiB");
  OMNIHD_REQUIRE(x_hi && g_hi && dw && ((x_lo == nullptr) == (g_lo == nullptr)), "null pointer / both or neither lo plane");
  OMNIHD_REQUIRE(((reinterpret_cast<uintptr_t>(x_hi) | reinterpret_cast<uintptr_t>(x_lo) | reinterpret_cast<uintptr_t>(g_hi) |
                   reinterpret_cast<uintptr_t>(g_lo)) & 15u) == 0, "16-byte alignment");
  const size_t need = omnihd_conv_wgrad_nhwc_workspace_bytes(batch, h, w, cin, ho, wo, cout, ksize, stride, pad, dil);
  if (a.n_split > 1 && (!workspace || workspace_bytes < need)) {
    set_error("conv_wgrad_nhwc: workspace %zu < required %zu", workspace_bytes, need);
    return OMNIHD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const bool split = x_lo != nullptr;
  float* slab = a.n_split > 1 ? reinterpret_cast<float*>(static_cast<char*>(workspace) + 256) : dw;
  const int blocks = a.tiles_n * a.tiles_c * ksize * ksize * a.n_split;
  const unsigned short *G = static_cast<const unsigned short*>(g_hi), *G2 = static_cast<const unsigned short*>(g_lo);
  const unsigned short *X = static_cast<const unsigned short*>(x_hi), *X2 = static_cast<const unsigned short*>(x_lo);
  if (split) hipLaunchKernelGGL((k_wgrad_nhwc<true, 2>), dim3(blocks), dim3(256), 0, st, G, G2, X, X2, slab, a);
  else hipLaunchKernelGGL((k_wgrad_nhwc<false, 2>), dim3(blocks), dim3(256), 0, st, G, G2, X, X2, slab, a);
  if (a.n_split > 1) {
    const size_t n = (size_t)cout * ksize * ksize * cin;
    hipLaunchKernelGGL(k_sum_slabs_nhwc, dim3(grid_for((int64_t)n, 256 * 4)), dim3(256), 0, st, slab, a.n_split, n, dw);
  }
  return check_launch("conv_wgrad_nhwc");
}
