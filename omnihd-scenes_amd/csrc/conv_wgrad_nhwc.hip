// Weight gradient of a convolution straight from the NHWC operands: no pixel-major staging pass (round 5).
//
// Where it sits: the weight gradients of the detector's convolutions — ResNet-50's bottlenecks (reference config
// projects/configs/bevfusion_NewScenes/bevfusion.py:77-85), FPN / FPNC adapters, DepthNet's 3x3 layers
// (bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:563-609), SECOND / SECONDFPN (bevfusion.py:62-74), the BEV encoder and
// the anchor head.
//
//   dW[n][tap][c] = sum over output pixels m of  G[m][n] * X[src_tap(m)][c]
//
// is a GEMM whose reduction index is the PIXEL — the slow dimension of both NHWC operands.  csrc/conv_wgrad.hip re-lays both operands
// pixel-major first (k_to_kmajor); here the tiles go into LDS as they lie in memory ([pixel][channel], LDS-DMA, 256-byte rows) and the
// MFMA fragments — 8 consecutive PIXELS of one channel per lane — are fetched with gfx950's transposing LDS read
// `ds_read_b64_tr_b16`: each 16-lane group reads a [4 pixel][16 channel] block (lane i supplies the address of pixel i/4, channels
// 4*(i%4)..+3) and lane i receives channel i of the 4 pixels (checked on the device: scripts/micro/tr_read.hip, 0 mismatches).
//
// A workgroup computes a 128 x 128 tile of (Cout, Cin) over its share of the pixels (split-K, fp32 slabs summed in a fixed order:
// deterministic) in one of two forms:
//   one tap    any kernel <= 4x4, stride, padding, dilation: the source pixel of every tile row is computed by the lane that
//              fetches it; a source pixel outside the image is an out-of-range buffer offset and arrives as a row of zeros.
//   three taps 3x3 / stride 1 / pad 1 / dilation 1: the reduction runs over a PADDED raster (one zero column behind every image row,
//              one zero row above every image — both are out-of-range fetches, nothing is materialised), on which the three taps
//              of a kernel row are the same X rows shifted by one pixel: a 34-row X tile serves three 128 x 128 accumulators, so
//              a K-step's MFMAs triple over nearly the same L2 -> LDS traffic.
// LDS rows are 256 bytes = the bank period, so the four pixel rows a 16-lane group reads would collide on the same banks: the
// 32-byte units of a row are XOR-swizzled by f(row) = 2 * (row % 4) + (row / 8) % 2 on the way in (each loader lane simply fetches
// a different 16-byte chunk) and again on the fragment reads — any four consecutive rows x two adjacent units then cover all eight
// units, for shifted taps as well.
// The transposing reads are inline assembly: as a compiler builtin every one of them is preceded by `s_waitcnt vmcnt(0)` while an
// LDS-DMA is pending (hipcc 7.2; it does not do that for plain LDS loads), which drains the fill ring at the top of every K-step.
// With hand-counted lgkmcnt / vmcnt the ring runs STAGES deep and the reads of the next tap are in flight under the MFMAs of this one.
// SPLIT: fp32-grade sums from hi / lo planes (G_hi X_hi + G_hi X_lo + G_lo X_hi).
// The launch is persistent: one workgroup per CU (three taps) or two (one tap) walk the item list (split, Cout tile, Cin tile, tap
// group), cut into one contiguous run per XCD.  Measured (profiles/round5/wgrad_nhwc_vs_chain.txt, wgrad_nhwc_pmc.txt): 1024 -> 1024
// 3x3 at 160 x 240, split form, 1.60 ms = 453 effective TFLOP/s (staged chain of csrc/conv_wgrad.hip 1.84 ms, library fp32 5.2 ms),
// MFMA pipes busy 68 % of the cycles at the 1.88 GHz the chip holds under this load, no LDS bank conflicts.
#include <type_traits>
#include <utility>

#include "common.h"

namespace omnihd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_ptr_t;

constexpr int kKP = 32;          // pixels per K-step
constexpr int kT = 128;          // tile edge (channels)
constexpr int kRowBytes = kT * 2;

struct WgNhwcArgs {
  int B, H, W, Cin, Ho, Wo, Cout;
  int k, stride, pad, dil;
  int tiles_n, tiles_c, n_split, px_per_split;   // px_per_split: a multiple of kKP
  long long slab_stride;                          // floats between the slabs of consecutive splits (0: n_split == 1, dst = dw)
  int three;                                      // three-taps form
  int total, per_xcd;                             // work items; items per XCD run (launched: 8 * min(per_xcd, resident workgroups per XCD))
  unsigned long long* trace;                      // lab (OMNIHD_WGRAD_NHWC_TRACE): per workgroup {start, end} in 10 ns ticks, hardware id, XCC id
  int RW, RH;                                     // the raster the reduction runs over: (Wo, Ho), or (W + 1, H + 1) padded
  const float* alpha;                             // F16 form: device scalar the result is multiplied by (NULL: 1)
};

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int OFF>
__device__ __forceinline__ void tr_read(v2i& dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void tie(v2i& r) { asm volatile("" : "+v"(r)); }       // "r was written by the asm above": orders its users
__device__ __forceinline__ bf16x8 frag_of(v2i lo, v2i hi) {
  const v4i v = {lo.x, lo.y, hi.x, hi.y};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ int swz(int row) { return ((row & 3) << 1) | ((row >> 3) & 1); }

struct RowPos { int cx, cy, cb; };

// F16 (round 6, with SPLIT = false): the operands are IEEE half planes (the TF32-grade form); the gradient plane carries a power-of-two
// scale whose inverse `a.alpha` (device) multiplies the result — in this kernel's epilogue when the pixels are not split, in the slab sum
// otherwise.
template <bool SPLIT, bool T3, int STAGES, bool F16 = false>
__global__ __launch_bounds__(256) void k_wgrad_nhwc(const unsigned short* __restrict__ G, const unsigned short* __restrict__ G2,
                                                    const unsigned short* __restrict__ X, const unsigned short* __restrict__ X2,
                                                    float* __restrict__ dst, const WgNhwcArgs a) {
  constexpr int PLANES = SPLIT ? 2 : 1;
  constexpr int NT = T3 ? 3 : 1;                                  // taps (accumulator sets) per workgroup
  constexpr int XROWS = T3 ? 48 : 32;                             // rows of an X tile (three taps: 34 used, 36 fetched, 48 addressed)
  constexpr int XR = T3 ? 3 : 2;                                  // X rows per loader lane
  constexpr int GB = kKP * kRowBytes;                             // bytes of a G tile (8 KB)
  constexpr int XB = XROWS * kRowBytes;                           // bytes of an X tile
  constexpr int XBASE = PLANES * GB;                              // first X tile inside a stage
  constexpr int STAGE_B = PLANES * (GB + XB);
  constexpr int CALLS = PLANES * (2 + XR);                        // LDS-DMA calls (1 KB = 4 pixel rows) per wavefront and K-step
  constexpr int PER = 4 * PLANES;                                 // transposing reads per fragment group (2 blocks x planes x 2 halves)
  __shared__ __attribute__((aligned(16))) unsigned char sm[STAGES * STAGE_B];

  const int taps = a.k * a.k;
  // Work order: the list of (split, Cout tile, Cin tile, tap group) — tap group fastest — is cut into 8 contiguous runs, one per
  // XCD (workgroup b runs on XCD b % 8): the ~32 workgroups an XCD runs at a time are then neighbours in that list, i.e. the same
  // pixel range and mostly the same G / X channel slices, which they fetch into their shared L2 once.  (Split-fastest order, the
  // first version, had every workgroup of an XCD stream its own pixel range: 5 TB/s of L2 misses on 1024 -> 1024 at 160 x 240.)
  // PERSISTENT: the launch holds at most as many workgroups as the chip runs at a time (a.wgs_per_xcd per XCD); workgroup j of
  // XCD x walks the items x * per_xcd + j, + wgs_per_xcd, ... of its XCD's run.  (Launched as one workgroup per item, 768 items on
  // 1024 -> 1024 at 160 x 240 kept only ~182 of 256 CUs occupied on average — SQ_WAVE_CYCLES against the kernel's duration — and a
  // launch of 192 took exactly as long as one of 768 or 1536: rounds of large-LDS workgroups do not back-fill.)
  const unsigned long long t_start = a.trace ? wall_clock64() : 0ull;
  const int xcd = (int)(blockIdx.x & 7), wg_j = (int)(blockIdx.x >> 3), wgs_per_xcd = (int)(gridDim.x >> 3);
  const int item_end = min(a.total, (xcd + 1) * a.per_xcd);
  for (int t_lin = xcd * a.per_xcd + wg_j; t_lin < item_end; t_lin += wgs_per_xcd) {
  const int n_tg = T3 ? 3 : taps;
  int t = t_lin;
  const int tg = t % n_tg; t /= n_tg;                            // tap group: kernel row (three taps) or tap
  const int ct = t % a.tiles_c; t /= a.tiles_c;
  const int nt = t % a.tiles_n;
  const int sp = t / a.tiles_n;
  const int ky = T3 ? tg : tg / a.k, kx0 = T3 ? 0 : tg - ky * a.k;
  const int RW = a.RW, RH = a.RH;
  const int M = a.B * RH * RW;                                    // raster positions
  const int m_begin = sp * a.px_per_split;
  const int m_end = min(M, m_begin + a.px_per_split);
  const int n_steps = m_begin < m_end ? (m_end - m_begin + kKP - 1) / kKP : 0;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = nt * kT, c0 = ct * kT;

  constexpr unsigned kOOB = 0x80000000u;
  const unsigned g_bytes = (unsigned)((size_t)a.B * a.Ho * a.Wo * a.Cout * 2);
  const unsigned x_bytes = (unsigned)((size_t)a.B * a.H * a.W * a.Cin * 2);
  const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)G, 0, (int)g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t g2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(SPLIT ? G2 : G), 0, (int)g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(SPLIT ? X2 : X), 0, (int)x_bytes, 0x00020000);

  // ---- loader: this lane fills LDS chunk q of the tile rows prow + 16 r; the chunk it fetches is the swizzled one -------------
  const int q = lane & 15;
  const int prow = 4 * wave + (lane >> 4);
  auto src_chunk = [&](int row) { return ((((q >> 1) ^ swz(row)) << 1) | (q & 1)) * 8; };     // first channel of the fetched chunk
  auto locate = [&](long long u) {            // raster position (may be up to one image before the first) -> (column, row, image)
    RowPos p;
    const long long img = (long long)RH * RW;
    const long long u2 = u + img;
    p.cx = (int)(u2 % RW);
    const long long r = u2 / RW;
    p.cy = (int)(r % RH);
    p.cb = (int)(r / RH) - 1;
    return p;
  };
  // one K-step further: + 32 positions in the mixed radix (RW, RH), branch-free (32 = (sb * RH + sy) * RW + sx)
  const int sx = kKP % RW, sy = (kKP / RW) % RH, sb = (kKP / RW) / RH;
  auto advance = [&](RowPos& p) {
    p.cx += sx;
    const int c1 = p.cx >= RW;
    p.cx -= c1 ? RW : 0;
    p.cy += sy + c1;
    const int c2 = p.cy >= RH;
    p.cy -= c2 ? RH : 0;
    p.cb += sb + c2;
  };
  // Per fetched row: its position on the raster (column, row[, image]) and — three-taps form — the byte offset of its source
  // pixel, carried from K-step to K-step: with p = (image * H + row - 1) * W + column a step of 32 raster positions changes p by
  // (sb * H * W + sy * W + sx) - [column wrapped] - W * [row wrapped], so the update is adds and selects (no multiply in the loop).
  RowPos gp[2], xp[XR];
  int g_ch[2], x_ch[XR];
  unsigned g_off[2], x_off[XR];                 // three taps: byte offset of the row's chunk (meaningful where the position is real)
  int g_du[2], x_du[XR];                        // three taps: raster position of the row minus the K-step start
  bool x_row_ok[XR];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    g_du[r] = prow + 16 * r;
    gp[r] = locate((long long)m_begin + g_du[r]);
    g_ch[r] = n0 + src_chunk(prow + 16 * r);
    g_off[r] = (unsigned)(((((long long)gp[r].cb * a.H + (gp[r].cy - 1)) * a.W + gp[r].cx) * a.Cout + g_ch[r]) * 2);
  }
#pragma unroll
  for (int r = 0; r < XR; ++r) {
    // three taps: X tile row j holds raster position (K-step start) + j - 1, one raster row up / down for the outer kernel rows
    x_du[r] = prow + 16 * r - 1 + (ky - 1) * RW;
    xp[r] = T3 ? locate((long long)m_begin + x_du[r]) : gp[r < 2 ? r : 0];
    x_ch[r] = c0 + src_chunk(prow + 16 * r);
    x_off[r] = (unsigned)(((((long long)xp[r].cb * a.H + (xp[r].cy - 1)) * a.W + xp[r].cx) * a.Cin + x_ch[r]) * 2);
    x_row_ok[r] = x_ch[r] < a.Cin && (r < 2 || wave == 0);      // (rows 36..47 of the tile: dummy calls, equal DMA counts per wave)
  }
  const unsigned g_step = (unsigned)(((long long)sb * a.H * a.W + (long long)sy * a.W + sx) * a.Cout * 2);
  const unsigned x_step = (unsigned)(((long long)sb * a.H * a.W + (long long)sy * a.W + sx) * a.Cin * 2);
  const unsigned g_c1 = (unsigned)a.Cout * 2u, g_c2 = (unsigned)a.W * (unsigned)a.Cout * 2u;
  const unsigned x_c1 = (unsigned)a.Cin * 2u, x_c2 = (unsigned)a.W * (unsigned)a.Cin * 2u;
  auto advance3 = [&](RowPos& p, unsigned& off, unsigned step, unsigned c1b, unsigned c2b) {
    p.cx += sx;
    const bool c1 = p.cx >= RW;
    p.cx -= c1 ? RW : 0;
    p.cy += sy + (c1 ? 1 : 0);
    const bool c2 = p.cy >= RH;
    p.cy -= c2 ? RH : 0;
    off += step - (c1 ? c1b : 0u) - (c2 ? c2b : 0u);
  };
  int m_issue = m_begin;                          // first raster position of the next K-step to issue
  // A fill in three parts, so that each can be placed among the MFMAs: (1) the offsets of this lane's 2 + XR rows (VALU only),
  // (2) the CALLS LDS-DMA calls one by one, (3) the position update.
  unsigned fill_off[2 + XR];                      // byte offset of each row's chunk for the fill being issued, or out of range
  auto fill_addr = [&](bool real) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {                 // G rows
      const RowPos p = gp[r];
      bool ok = real & ((m_issue + prow + 16 * r) < m_end) & (g_ch[r] < a.Cout);
      unsigned off;
      if constexpr (T3) {
        ok = ok & (p.cx < a.W) & (p.cy >= 1);     // (image index < B follows from the position < M)
        off = g_off[r];
      } else {
        off = (unsigned)((((size_t)p.cb * RH + p.cy) * RW + p.cx) * a.Cout + g_ch[r]) * 2u;
      }
      fill_off[r] = ok ? off : kOOB;
    }
#pragma unroll
    for (int r = 0; r < XR; ++r) {                // X rows
      const RowPos p = xp[r];
      bool ok = real & x_row_ok[r];
      unsigned off;
      if constexpr (T3) {
        ok = ok & (p.cx < a.W) & (p.cy >= 1) & ((unsigned)(m_issue + x_du[r]) < (unsigned)M);
        off = x_off[r];
      } else {
        const int iy = p.cy * a.stride - a.pad + ky * a.dil, ix = p.cx * a.stride - a.pad + kx0 * a.dil;
        ok = ok & ((m_issue + prow + 16 * r) < m_end) & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
        off = (unsigned)((((size_t)p.cb * a.H + iy) * a.W + ix) * a.Cin + x_ch[r]) * 2u;
      }
      fill_off[2 + r] = ok ? off : kOOB;
    }
  };
  // (the DMA builtin sits in a NON-generic lambda: inside a generic one hipcc 7.2's host pass silently drops the kernel's launch stub)
  auto dma16 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned char* dst, unsigned off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t*)dst, 16, off, 0, 0, 0);
  };
  auto fill_call = [&](auto K, int stage) {       // call K of CALLS: row K / PLANES (G rows first), plane K % PLANES
    constexpr int ROW = K / PLANES, PL = K % PLANES;
    unsigned char* base = sm + stage * STAGE_B;
    if constexpr (ROW < 2) {
      dma16(PL ? g2_rsrc : g_rsrc, base + (4 * wave + 16 * ROW) * kRowBytes + PL * GB, fill_off[ROW]);
    } else {
      dma16(PL ? x2_rsrc : x_rsrc, base + XBASE + (4 * wave + 16 * (ROW - 2)) * kRowBytes + PL * XB, fill_off[ROW]);
    }
  };
  auto issue_adv = [&]() {                        // ... and every row one K-step further
    m_issue += kKP;
    if constexpr (T3) {
#pragma unroll
      for (int r = 0; r < 2; ++r) advance3(gp[r], g_off[r], g_step, g_c1, g_c2);
#pragma unroll
      for (int r = 0; r < XR; ++r) advance3(xp[r], x_off[r], x_step, x_c1, x_c2);
    } else {
#pragma unroll
      for (int r = 0; r < 2; ++r) advance(gp[r]);
#pragma unroll
      for (int r = 0; r < XR; ++r) xp[r] = gp[r];
    }
  };
  auto issue = [&](int stage, bool real) {
    fill_addr(real);
    static_for<CALLS>([&](auto K) { fill_call(K, stage); });
    issue_adv();
  };

  f32x16 acc[NT][2][2];
#pragma unroll
  for (int tp = 0; tp < NT; ++tp)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][i][j][r] = 0.f;

  // ---- fragment reads: wave (wm, wn) owns rows n = wm*64 .. +63 of the tile's Cout range and c = wn*64 .. +63 of its Cin range ----
  // A 16-lane group fetches a [4 pixel][16 channel] block: this lane's 8-byte piece is pixel row 8 * (grp / 2) + li / 4 (+ 4 for the
  // second half of its 8 pixels, + the tap's shift), 32-byte unit (block base / 16 + grp % 2) ^ swz(row), bytes 8 * (li % 4) ..
  const int wm = wave >> 1, wn = wave & 1;
  const int grp = lane >> 4, li = lane & 15;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)sm;
  auto piece = [&](int row, int unit0) {       // byte offset inside a tile, MFMA block i = 0 (block 1: ^ 64)
    return (unsigned)(row * kRowBytes + (((unit0 + (grp & 1)) ^ swz(row)) << 5) + 8 * (li & 3));
  };
  unsigned ga[2][2], xa[NT][2][2];              // [half][block], [tap][half][block]
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * (grp >> 1) + (li >> 2) + 4 * h;
    ga[h][0] = piece(row, wm * 4);
    ga[h][1] = ga[h][0] ^ 64u;
#pragma unroll
    for (int tp = 0; tp < NT; ++tp) {
      xa[tp][h][0] = piece(row + tp, wn * 4);
      xa[tp][h][1] = xa[tp][h][0] ^ 64u;
    }
  }

  v2i gq[2][2][PLANES][2];                       // [slice][block][plane][half]
  v2i xq[2][2][PLANES][2];                       // [buffer][block][plane][half]

  auto read_g = [&](auto S, unsigned sbase) {
    static_for<2>([&](auto I) {
      static_for<PLANES>([&](auto P) {
        static_for<2>([&](auto Hh) { tr_read<P * GB + S * 16 * kRowBytes>(gq[S][I][P][Hh], sbase + ga[Hh][I]); });
      });
    });
  };
  auto read_g_half = [&](auto S, auto I, unsigned sbase) {      // MFMA block I of slice S
    static_for<PLANES>([&](auto P) {
      static_for<2>([&](auto Hh) { tr_read<P * GB + S * 16 * kRowBytes>(gq[S][I][P][Hh], sbase + ga[Hh][I]); });
    });
  };
  auto read_x = [&](auto Q, unsigned sbase) {    // group Q = slice * NT + tap -> buffer Q % 2
    constexpr int S = Q / NT, TP = Q % NT, BUF = Q % 2;
    static_for<2>([&](auto I) {
      static_for<PLANES>([&](auto P) {
        static_for<2>([&](auto Hh) { tr_read<XBASE + P * XB + S * 16 * kRowBytes>(xq[BUF][I][P][Hh], sbase + xa[TP][Hh][I]); });
      });
    });
  };
  auto tie_x = [&](auto BUF) {
    static_for<2>([&](auto I) { static_for<PLANES>([&](auto P) { static_for<2>([&](auto Hh) { tie(xq[BUF][I][P][Hh]); }); }); });
  };
  auto tie_g = [&](auto S) {
    static_for<2>([&](auto I) { static_for<PLANES>([&](auto P) { static_for<2>([&](auto Hh) { tie(gq[S][I][P][Hh]); }); }); });
  };
  auto mma = [&](auto Q, auto&& hook) {          // hook(K) runs behind MFMA K of the group, pinned there
    constexpr int S = Q / NT, TP = Q % NT, BUF = Q % 2;
    bf16x8 fa[2], fb[2], fa2[2], fb2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fa[i] = frag_of(gq[S][i][0][0], gq[S][i][0][1]);
      fb[i] = frag_of(xq[BUF][i][0][0], xq[BUF][i][0][1]);
      if constexpr (SPLIT) {
        fa2[i] = frag_of(gq[S][i][PLANES - 1][0], gq[S][i][PLANES - 1][1]);
        fb2[i] = frag_of(xq[BUF][i][PLANES - 1][0], xq[BUF][i][PLANES - 1][1]);
      }
    }
    static_for<SPLIT ? 3 : 1>([&](auto TERM) {     // split: lo*hi, hi*lo, hi*hi
      static_for<4>([&](auto IJ) {
        constexpr int i = IJ / 2, j = IJ % 2;
        const bf16x8 av = (SPLIT && TERM == 0) ? fa2[i] : fa[i];
        const bf16x8 bv = (SPLIT && TERM == 1) ? fb2[j] : fb[j];
        if constexpr (F16) acc[TP][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), acc[TP][i][j], 0, 0, 0);
        else acc[TP][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[TP][i][j], 0, 0, 0);
        hook(std::integral_constant<int, TERM * 4 + IJ>{});
      });
    });
  };
  auto no_hook = [](auto) {};
  // one MFMA, then a few of the fill's address / DMA instructions: the pattern handed to the scheduler for the MFMA groups that
  // carry a part of the fill (left to itself it clusters the address arithmetic behind the MFMAs and the matrix pipe idles)
  auto interleave = [&](auto N_VALU) {
#pragma unroll
    for (int e = 0; e < 4 * (SPLIT ? 3 : 1); ++e) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // MFMA
      __builtin_amdgcn_sched_group_barrier(0x016, N_VALU, 0);     // VALU | SALU | VMEM (the LDS-DMA calls and their m0 moves)
    }
  };
  constexpr int LAST = 2 * NT - 1;
  constexpr int Q_ADDR = 0, Q_CALLS = 1, Q_ADV = 2;               // three taps: which MFMA group carries which part of the fill
  constexpr int MPG = 4 * (SPLIT ? 3 : 1);                        // MFMAs per group

  // The pipeline, per K-step (stage = step % STAGES):
  //   [G slice 1 fragments requested]  group 0 .. LAST: { request the X fragments of the next group; wait for this group's; MFMAs }
  //   groups 0 / 1 / 2 also carry the fills of K-step + STAGES - 1: G rows / X rows / position update;
  //   in front of the LAST group's MFMAs: this wave's fills of the next K-step have landed (vmcnt), barrier (everybody's have, and
  //   everybody is done reading this stage — every read of it was waited for), then the next step's first fragments (G slice 0, X
  //   group 0) are requested: their latency, and the barrier's, hide behind the LAST group's 12 MFMAs.
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) issue(s, s < n_steps);
  __builtin_amdgcn_sched_barrier(0);
  wait_vm<CALLS*(STAGES - 2)>();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  read_g(std::integral_constant<int, 0>{}, lds0);
  read_x(std::integral_constant<int, 0>{}, lds0);
  int stage = 0;
  for (int step = 0; step < n_steps; ++step) {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned sbase = lds0 + (unsigned)stage * (unsigned)STAGE_B;
    const int next_stage = stage + 1 == STAGES ? 0 : stage + 1;
    const int fill_stage = stage == 0 ? STAGES - 1 : stage - 1;        // (stage + STAGES - 1) % STAGES: left by everybody one barrier ago
    const bool fill_real = step + STAGES - 1 < n_steps;
    static_for<2 * NT>([&](auto Q) {
      // the G fragments of slice 1 (needed from group NT on) are requested in two halves IN FRONT of the X requests of groups
      // NT - 2 and NT - 1: every wait below then leaves at most PER + PER / 2 <= 12 younger reads outstanding (lgkmcnt is 4 bits)
      // and never waits for a read that was only just requested
      constexpr bool G1A = NT >= 3 && Q == NT - 2, G1B = NT >= 3 && Q == NT - 1;
      if constexpr (G1A) read_g_half(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, sbase);
      if constexpr (G1B) read_g_half(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, sbase);
      if constexpr (Q < LAST) {
        read_x(std::integral_constant<int, Q + 1>{}, sbase);
        wait_lgkm<PER + ((G1A || G1B) ? PER / 2 : 0)>();       // all but the X group just requested (and the G half requested with it)
      } else {
        wait_lgkm<0>();
      }
      if constexpr (Q == 0) tie_g(std::integral_constant<int, 0>{});
      if constexpr (NT < 3 && Q == 0) read_g(std::integral_constant<int, 1>{}, sbase);   // one tap: under group 0's MFMAs, waited for by group 1
      if constexpr (Q == NT) tie_g(std::integral_constant<int, 1>{});
      tie_x(std::integral_constant<int, Q % 2>{});
      if constexpr (Q == LAST) {
        __builtin_amdgcn_sched_barrier(0);
        wait_vm<CALLS*(STAGES - 2)>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned nbase = lds0 + (unsigned)next_stage * (unsigned)STAGE_B;
        read_g(std::integral_constant<int, 0>{}, nbase);          // (behind the last K-step: a stage of dummy fills, never used)
        read_x(std::integral_constant<int, 0>{}, nbase);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NT > 1 && Q == Q_ADDR) {
        // offsets of the fill's rows: VALU / SALU only, a few behind every MFMA
        fill_addr(fill_real);
        mma(Q, no_hook);
        interleave(std::integral_constant<int, 4>{});
      } else if constexpr (NT > 1 && Q == Q_CALLS) {
        // the DMA calls, one behind each of the first CALLS MFMAs (pinned: the scheduler keeps m0-chained calls in one clump)
        mma(Q, [&](auto K) {
          if constexpr (K < CALLS) {
            __builtin_amdgcn_sched_barrier(0);
            fill_call(K, fill_stage);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        if constexpr (CALLS > MPG) static_for<CALLS - MPG>([&](auto K) { fill_call(std::integral_constant<int, MPG + K>{}, fill_stage); });
      } else if constexpr (NT > 1 && Q == Q_ADV) {
        issue_adv();
        mma(Q, no_hook);
        interleave(std::integral_constant<int, 7>{});
      } else if constexpr (NT == 1 && Q == 0) {
        mma(Q, no_hook);
        issue(fill_stage, fill_real);
      } else {
        mma(Q, no_hook);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    stage = next_stage;
  }
  wait_vm<0>();                                  // drain the dummy tail fills before the epilogue

  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31 (B row = input channel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float* out = dst + (size_t)sp * a.slab_stride;
  const float alpha_v = (F16 && a.alpha) ? *a.alpha : 1.f;
#pragma unroll
  for (int tp = 0; tp < NT; ++tp) {
    const int tap = T3 ? ky * 3 + tp : tg;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int c = c0 + wn * 64 + j * 32 + (lane & 31);
          if (n < a.Cout && c < a.Cin) out[((size_t)n * taps + tap) * a.Cin + c] = (F16 && a.n_split == 1) ? acc[tp][i][j][r] * alpha_v : acc[tp][i][j][r];
        }
  }
  __builtin_amdgcn_s_barrier();                  // the next item's first fills overwrite stages a slower wave may still be reading
  }                                              // item loop
  if (a.trace && threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* t = a.trace + 4 * (size_t)blockIdx.x;
    t[0] = t_start; t[1] = wall_clock64(); t[2] = hw; t[3] = xcc;
  }
}

__global__ __launch_bounds__(256) void k_sum_slabs_nhwc(const float* __restrict__ slab, int n_split, size_t n, float* __restrict__ out,
                                                        const float* __restrict__ alpha) {
  const float al = alpha ? *alpha : 1.f;                                       // (F16 form: the inverse of the gradient plane's scale)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float s = slab[i];
    for (int k = 1; k < n_split; ++k) s += slab[(size_t)k * n + i];           // fixed order: run-to-run identical
    out[i] = alpha ? s * al : s;
  }
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

bool nhwc_plan(int batch, int h, int w, int cin, int ho, int wo, int cout, int k, int stride, int pad, int dil, WgNhwcArgs* a) {
  if (batch <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || ho <= 0 || wo <= 0) return false;
  if (k < 1 || k > 4 || stride < 1 || dil < 1 || pad < 0 || cin % 8 || cout % 8) return false;
  if (ho != (h + 2 * pad - dil * (k - 1) - 1) / stride + 1 || wo != (w + 2 * pad - dil * (k - 1) - 1) / stride + 1) return false;
  const long long Mo = (long long)batch * ho * wo;
  if (Mo * cout * 2 >= (1ll << 31) || (long long)batch * h * w * cin * 2 >= (1ll << 31)) return false;    // 32-bit buffer offsets
  a->B = batch; a->H = h; a->W = w; a->Cin = cin; a->Ho = ho; a->Wo = wo; a->Cout = cout;
  a->k = k; a->stride = stride; a->pad = pad; a->dil = dil;
  a->tiles_n = (cout + kT - 1) / kT; a->tiles_c = (cin + kT - 1) / kT;
  static const int three_mode = env_int("OMNIHD_WGRAD_NHWC_THREE", 1);
  const long long tiles_cn = (long long)a->tiles_n * a->tiles_c;
  // three taps: where the (Cout, Cin) tiles x 3 kernel rows still give the chip work (one workgroup per CU: 120 KB of LDS)
  a->three = three_mode && k == 3 && stride == 1 && pad == 1 && dil == 1 && (long long)(h + 1) * (w + 1) * batch < (1ll << 30);
  if (a->three) {
    a->RW = w + 1; a->RH = h + 1;
  } else {
    a->RW = wo; a->RH = ho;
  }
  const long long M = (long long)batch * a->RH * a->RW;
  const int steps = (int)((M + kKP - 1) / kKP);
  const long long groups = tiles_cn * (a->three ? 3 : k * k);
  const long long slab_floats = (long long)cout * k * k * cin;
  long long sp;
  if (a->three) {
    // whole rounds of kCUs workgroups; the fewest rounds that fill >= 93 % of their slots, at least 8 K-steps per split
    const long long max_sp = steps / 8 > 0 ? steps / 8 : 1;
    sp = 1;
    double best = 0.0;
    for (int rounds = 1; rounds <= 4; ++rounds) {
      long long s = (long long)kCUs * rounds / groups;
      if (s < 1) s = 1;
      if (s > max_sp) s = max_sp;
      const long long wgs = groups * s;
      const double eff = (double)wgs / (double)(((wgs + kCUs - 1) / kCUs) * kCUs);
      if (eff > best + 1e-9) { best = eff; sp = s; }
      if (eff >= 0.93 || s == max_sp) break;
    }
  } else {
    sp = (3 * kCUs + groups - 1) / groups;                           // aim at >= 3 workgroups per CU
    const long long max_sp = steps / 4 > 0 ? steps / 4 : 1;           // at least 4 K-steps per split
    if (sp > max_sp) sp = max_sp;
  }
  if (sp > (a->three ? 512 : 64)) sp = a->three ? 512 : 64;    // (the slab sum is one thread per element over all splits)
  static const int forced = env_int("OMNIHD_WGRAD_NHWC_SPLITS", 0);  // (lab: scripts/lab/wgrad_nhwc_bench.py sweeps it)
  if (forced > 0) sp = forced;
  while (sp > 1 && sp * slab_floats * 4 > (256ll << 20)) --sp;       // slabs of at most 256 MB (1024 -> 1024 3x3: 4 x 38 MB)
  if (sp < 1) sp = 1;
  const int steps_per = (int)((steps + sp - 1) / sp);
  a->n_split = (steps + steps_per - 1) / steps_per;               // no empty split
  a->px_per_split = steps_per * kKP;
  a->slab_stride = a->n_split > 1 ? slab_floats : 0;
  a->trace = nullptr;
  a->alpha = nullptr;
  a->total = (int)(groups * a->n_split);
  a->per_xcd = (a->total + 7) / 8;
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

static int wgrad_nhwc_launch(const void* x_hi, const void* x_lo, const void* g_hi, const void* g_lo, float* dw, int batch, int h, int w,
                             int cin, int ho, int wo, int cout, int ksize, int stride, int pad, int dil, void* workspace,
                             size_t workspace_bytes, bool f16, const float* alpha, void* stream);

extern "C" int omnihd_conv_wgrad_nhwc(const void* x_hi, const void* x_lo, const void* g_hi, const void* g_lo, float* dw, int batch,
                                      int h, int w, int cin, int ho, int wo, int cout, int ksize, int stride, int pad, int dil,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  return wgrad_nhwc_launch(x_hi, x_lo, g_hi, g_lo, dw, batch, h, w, cin, ho, wo, cout, ksize, stride, pad, dil, workspace,
                           workspace_bytes, false, nullptr, stream);
}

extern "C" int omnihd_conv_wgrad_nhwc_f16(const void* x16, const void* g16, float* dw, const float* alpha, int batch, int h, int w,
                                          int cin, int ho, int wo, int cout, int ksize, int stride, int pad, int dil,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  return wgrad_nhwc_launch(x16, nullptr, g16, nullptr, dw, batch, h, w, cin, ho, wo, cout, ksize, stride, pad, dil, workspace,
                           workspace_bytes, true, alpha, stream);
}

static int wgrad_nhwc_launch(const void* x_hi, const void* x_lo, const void* g_hi, const void* g_lo, float* dw, int batch, int h, int w,
                             int cin, int ho, int wo, int cout, int ksize, int stride, int pad, int dil, void* workspace,
                             size_t workspace_bytes, bool f16, const float* alpha, void* stream) {
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
  // persistent launch: one workgroup per CU for the three-taps form (192 accumulator registers: one wave per SIMD) and for the
  // three-stage one-tap split form (96 KB of LDS), two otherwise
  static const int one_tap_stages = env_int("OMNIHD_WGRAD_NHWC_STAGES", 2);
  const int resident_per_xcd = (kCUs / 8) * ((a.three || (one_tap_stages == 3 && split)) ? 1 : 2);
  const int blocks = 8 * (a.per_xcd < resident_per_xcd ? a.per_xcd : resident_per_xcd);
  const unsigned short *G = static_cast<const unsigned short*>(g_hi), *G2 = static_cast<const unsigned short*>(g_lo);
  const unsigned short *X = static_cast<const unsigned short*>(x_hi), *X2 = static_cast<const unsigned short*>(x_lo);
  static const char* trace_path = getenv("OMNIHD_WGRAD_NHWC_TRACE");        // lab only: synchronises and dumps every launch
  if (trace_path) {
    if (hipMalloc((void**)&a.trace, (size_t)blocks * 32) != hipSuccess) a.trace = nullptr;
    else (void)hipMemsetAsync(a.trace, 0, (size_t)blocks * 32, st);
  }
  a.alpha = f16 ? alpha : nullptr;
  if (f16) {
    OMNIHD_REQUIRE(!split, "the half form takes one plane per operand");
    if (a.three) hipLaunchKernelGGL((k_wgrad_nhwc<false, true, 3, true>), dim3(blocks), dim3(256), 0, st, G, G2, X, X2, slab, a);
    else if (one_tap_stages == 3) hipLaunchKernelGGL((k_wgrad_nhwc<false, false, 3, true>), dim3(blocks), dim3(256), 0, st, G, G2, X, X2, slab, a);
    else hipLaunchKernelGGL((k_wgrad_nhwc<false, false, 2, true>), dim3(blocks), dim3(256), 0, st, G, G2, X, X2, slab, a);
  } else if (a.three) {
    if (split) hipLaunchKernelGGL((k_wgrad_nhwc<true, true, 3>), dim3(blocks), dim3(256), 0, st, G, G2, X, X2, slab, a);
    else hipLaunchKernelGGL((k_wgrad_nhwc<false, true, 3>), dim3(blocks), dim3(256), 0, st, G, G2, X, X2, slab, a);
  } else if (one_tap_stages == 3) {
    if (split) hipLaunchKernelGGL((k_wgrad_nhwc<true, false, 3>), dim3(blocks), dim3(256), 0, st, G, G2, X, X2, slab, a);
    else hipLaunchKernelGGL((k_wgrad_nhwc<false, false, 3>), dim3(blocks), dim3(256), 0, st, G, G2, X, X2, slab, a);
  } else {
    if (split) hipLaunchKernelGGL((k_wgrad_nhwc<true, false, 2>), dim3(blocks), dim3(256), 0, st, G, G2, X, X2, slab, a);
    else hipLaunchKernelGGL((k_wgrad_nhwc<false, false, 2>), dim3(blocks), dim3(256), 0, st, G, G2, X, X2, slab, a);
  }
  if (a.trace) {
    unsigned long long* h = (unsigned long long*)malloc((size_t)blocks * 32);
    if (h && hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h, a.trace, (size_t)blocks * 32, hipMemcpyDeviceToHost) == hipSuccess) {
      if (FILE* f = fopen(trace_path, "a")) {
        fprintf(f, "launch cin %d cout %d k %d three %d blocks %d items %d\n", cin, cout, ksize, a.three, blocks, a.total);
        for (int b = 0; b < blocks; ++b)
          fprintf(f, "%d %llu %llu %llu %llu\n", b, h[4 * b], h[4 * b + 1], h[4 * b + 2], h[4 * b + 3]);
        fclose(f);
      }
    }
    free(h);
    (void)hipFree(a.trace);
  }
  if (a.n_split > 1) {
    const size_t n = (size_t)cout * ksize * ksize * cin;
    hipLaunchKernelGGL(k_sum_slabs_nhwc, dim3(grid_for((int64_t)n, 256 * 4)), dim3(256), 0, st, slab, a.n_split, n, dw, a.alpha);
  }
  return check_launch("conv_wgrad_nhwc");
}
