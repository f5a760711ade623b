// Rotated-box BEV NMS for the detector's test-time path.
//
// Replaces mmdet3d v0.17.1 `iou3d_cuda.nms_gpu` (mmdet3d/ops/iou3d/src/iou3d_kernel.cu: iou_bev,
// box_overlap, nms_kernel + the host-side mask reduction in iou3d.cpp), reached from the reference
// through `Anchor3DHead.get_bboxes` -> `box3d_multiclass_nms` -> `nms_gpu` with the test_cfg of
// projects/configs/bevfusion_NewScenes/bevfusion.py:147-155 (use_rotate_nms, nms_thr 0.2).
// mmdet3d is not vendored in the reference tree, so the polygon-clipping algorithm is restated
// from its published source; oracle/nms_oracle.c restates it again on the CPU.
//
// Differences of shape, not of result:
//  * 64-thread workgroups = one wavefront per (row block, column block); only blocks on or above
//    the diagonal are launched (the reduction never reads the others);
//  * the suppression masks are reduced ON THE DEVICE by one wavefront (lane w owns word w of the
//    "removed" bitmap), so no mask is copied to the host; only the count comes back when the
//    caller asks for it;
//  * sin/cos/atan2 are evaluated in double and rounded to float, and floating-point contraction is
//    off, so the CPU oracle reproduces every IoU bit for bit.
#include "common.h"

#pragma clang fp contract(off)

namespace omnihd {
namespace {

constexpr int kNmsBlock = 64;          // boxes per block = bits per mask word = wavefront size
constexpr float kEps = 1e-8f;
constexpr float kMargin = 1e-5f;

struct Pt {
  float x, y;
};

__device__ __forceinline__ float cross2(Pt a, Pt b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ float cross3(Pt p1, Pt p2, Pt p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
__device__ __forceinline__ float cosd(float a) { return (float)cos((double)a); }
__device__ __forceinline__ float sind(float a) { return (float)sin((double)a); }

__device__ __forceinline__ bool rect_cross(Pt p1, Pt p2, Pt q1, Pt q2) {
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

// A box with its trigonometry done once: (x1,y1,x2,y2) + cos/sin of the angle.  cos(-t) = cos(t) and
// sin(-t) = -sin(t) hold bit for bit, so upstream's cos(-angle)/sin(-angle) in the point-in-box test
// reuse the same two numbers.
struct RBox {
  float x1, y1, x2, y2, c, s;
};

__device__ __forceinline__ RBox make_rbox(const float* b) {
  return RBox{b[0], b[1], b[2], b[3], cosd(b[4]), sind(b[4])};
}

// Is p inside the rotated box?  The point is turned back by -angle.
__device__ __forceinline__ bool in_box(const RBox& box, Pt p) {
  float cx = (box.x1 + box.x2) / 2, cy = (box.y1 + box.y2) / 2;
  float c = box.c, s = -box.s;
  float rx = (p.x - cx) * c + (p.y - cy) * s + cx;
  float ry = -(p.x - cx) * s + (p.y - cy) * c + cy;
  return rx > box.x1 - kMargin && rx < box.x2 + kMargin && ry > box.y1 - kMargin && ry < box.y2 + kMargin;
}

__device__ __forceinline__ bool seg_intersection(Pt p1, Pt p0, Pt q1, Pt q0, Pt& ans) {
  if (!rect_cross(p0, p1, q0, q1)) return false;
  float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0);
  float s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > kEps) {
    ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float d = a0 * b1 - a1 * b0;
    ans.x = (b0 * c1 - b1 * c0) / d;
    ans.y = (a1 * c0 - a0 * c1) / d;
  }
  return true;
}

__device__ __forceinline__ void box_corners(const RBox& box, Pt* c5) {
  float cx = (box.x1 + box.x2) / 2, cy = (box.y1 + box.y2) / 2;
  const float xs[4] = {box.x1, box.x2, box.x2, box.x1};
  const float ys[4] = {box.y1, box.y1, box.y2, box.y2};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float dx = xs[k] - cx, dy = ys[k] - cy;
    c5[k].x = dx * box.c + dy * box.s + cx;
    c5[k].y = -dx * box.s + dy * box.c + cy;
  }
  c5[4] = c5[0];
}

__device__ float overlap_area(const RBox& a, const RBox& b) {
  // Not upstream: boxes whose circumscribed circles are a millimetre apart share no point, every
  // test below would fail and the area would be 0 -- skip the work (most pairs of a frame).
  {
    float dx = (a.x1 + a.x2) / 2 - (b.x1 + b.x2) / 2, dy = (a.y1 + a.y2) / 2 - (b.y1 + b.y2) / 2;
    float ra = 0.5f * sqrtf((a.x2 - a.x1) * (a.x2 - a.x1) + (a.y2 - a.y1) * (a.y2 - a.y1));
    float rb = 0.5f * sqrtf((b.x2 - b.x1) * (b.x2 - b.x1) + (b.y2 - b.y1) * (b.y2 - b.y1));
    float reach = ra + rb + 1e-3f;
    if (dx * dx + dy * dy > reach * reach * 1.0001f) return 0.f;
  }
  Pt ca[5], cb[5];
  box_corners(a, ca);
  box_corners(b, cb);
  Pt pts[16];
  float ang[16];
  Pt centre{0.f, 0.f};
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      Pt x;
      if (seg_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], x)) {
        centre.x = centre.x + x.x;
        centre.y = centre.y + x.y;
        pts[cnt++] = x;
      }
    }
  for (int k = 0; k < 4; ++k) {
    if (in_box(a, cb[k])) {
      centre.x = centre.x + cb[k].x;
      centre.y = centre.y + cb[k].y;
      pts[cnt++] = cb[k];
    }
    if (in_box(b, ca[k])) {
      centre.x = centre.x + ca[k].x;
      centre.y = centre.y + ca[k].y;
      pts[cnt++] = ca[k];
    }
  }
  if (cnt == 0) return 0.f;            // upstream divides 0/0 here and then sums nothing: area 0
  centre.x = centre.x / cnt;
  centre.y = centre.y / cnt;
  // Upstream bubble-sorts with atan2 recomputed inside the comparator; the angle of a point is a
  // pure function of the point, so computing it once gives the same permutation.
  for (int k = 0; k < cnt; ++k) ang[k] = (float)atan2((double)(pts[k].y - centre.y), (double)(pts[k].x - centre.x));
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (ang[i] > ang[i + 1]) {
        Pt t = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = t;
        float u = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = u;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k) {
    Pt u{pts[k].x - pts[0].x, pts[k].y - pts[0].y};
    Pt v{pts[k + 1].x - pts[0].x, pts[k + 1].y - pts[0].y};
    area = area + cross2(u, v);
  }
  return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float iou_bev(const RBox& a, const RBox& b) {
  float sa = (a.x2 - a.x1) * (a.y2 - a.y1);
  float sb = (b.x2 - b.x1) * (b.y2 - b.y1);
  float so = overlap_area(a, b);
  return so / fmaxf(sa + sb - so, kEps);
}

// mask[i][cb] bit j set <=> IoU(box i, box cb*64+j) > thresh, for j later in score order than i.
__global__ void __launch_bounds__(kNmsBlock)
k_nms_mask(const float* __restrict__ boxes, int n, float thresh, unsigned long long* __restrict__ mask,
           int col_blocks) {
  // Upper-triangular block enumeration: blockIdx.x -> (rb, cb) with cb >= rb.
  int t = blockIdx.x, rb = 0;
  while (t >= col_blocks - rb) {
    t -= col_blocks - rb;
    ++rb;
  }
  int cb = rb + t;
  __shared__ RBox colbox[kNmsBlock];
  int col_size = min(n - cb * kNmsBlock, kNmsBlock);
  int row_size = min(n - rb * kNmsBlock, kNmsBlock);
  if ((int)threadIdx.x < col_size) colbox[threadIdx.x] = make_rbox(boxes + (size_t)(cb * kNmsBlock + threadIdx.x) * 5);
  __syncthreads();
  if ((int)threadIdx.x < row_size) {
    int i = rb * kNmsBlock + threadIdx.x;
    RBox cur = make_rbox(boxes + (size_t)i * 5);
    unsigned long long bits = 0;
    int start = (rb == cb) ? (int)threadIdx.x + 1 : 0;
    for (int j = start; j < col_size; ++j)
      if (iou_bev(cur, colbox[j]) > thresh) bits |= 1ULL << j;
    mask[(size_t)i * col_blocks + cb] = bits;
  }
}

// Pairwise IoU matrix (row box a_i against column box b_j), the boxes_iou_bev_gpu analogue; used by
// tests to compare every IoU with the oracle, not only the thresholded bits.
__global__ void k_iou_matrix(const float* __restrict__ a, int na, const float* __restrict__ b, int nb,
                             float* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)na * nb) return;
  int i = (int)(idx / nb), j = (int)(idx % nb);
  out[idx] = iou_bev(make_rbox(a + (size_t)i * 5), make_rbox(b + (size_t)j * 5));
}

// One wavefront walks the boxes in score order.  Lane w keeps word w of the removed-set (n <= 4096
// -> at most 64 words).  The mask rows of 64 boxes at a time are staged in LDS with one burst of
// coalesced loads, so the serial walk never waits on global memory.
__global__ void __launch_bounds__(64)
k_nms_reduce(const unsigned long long* __restrict__ mask, int n, int col_blocks, long long* __restrict__ keep,
             int* __restrict__ num_out) {
  __shared__ unsigned long long rows[64][64];          // [row in chunk][mask word]  32 KiB
  const int lane = threadIdx.x;
  unsigned long long removed = 0;
  int kept = 0;
  for (int base = 0; base < n; base += 64) {
    const int w0 = base >> 6;
    const int nrows = min(64, n - base);
    const bool live = lane < col_blocks && lane >= w0;
    for (int r = 0; r < nrows; ++r) rows[r][lane] = live ? mask[(size_t)(base + r) * col_blocks + lane] : 0ULL;
    __syncthreads();
    for (int r = 0; r < nrows; ++r) {
      unsigned long long word = __shfl(removed, w0, 64);
      if (!((word >> r) & 1ULL)) {
        if (lane == 0) keep[kept] = base + r;
        ++kept;
        removed |= rows[r][lane];
      }
    }
    __syncthreads();
  }
  if (lane == 0) *num_out = kept;
}

}  // namespace
}  // namespace omnihd

extern "C" {

size_t omnihd_nms_rotated_workspace_bytes(int n) {
  if (n < 0) n = 0;
  size_t cb = (size_t)(n + 63) / 64;
  return omnihd::align_up((size_t)n * cb * sizeof(unsigned long long) + 256, 256);
}

int omnihd_nms_rotated(const float* boxes, int n, float thresh, long long* keep, int* num_out,
                       void* workspace, size_t workspace_bytes, void* stream) {
  using namespace omnihd;
  OMNIHD_REQUIRE(n >= 0, "n >= 0");
  OMNIHD_REQUIRE(num_out != nullptr, "num_out is null");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    OMNIHD_HIP_TRY(hipMemsetAsync(num_out, 0, sizeof(int), s));
    return OMNIHD_OK;
  }
  OMNIHD_REQUIRE(boxes && keep, "boxes/keep is null");
  OMNIHD_REQUIRE(n <= 4096, "n <= 4096 (one window of 64 mask words)");
  OMNIHD_REQUIRE(workspace && workspace_bytes >= omnihd_nms_rotated_workspace_bytes(n), "workspace too small");
  int cb = (n + 63) / 64;
  unsigned long long* mask = (unsigned long long*)workspace;
  // Lower-triangular words are never read, but a row's words lane>=w are; zero nothing: every word
  // with cb >= rb is written by k_nms_mask.
  int blocks = cb * (cb + 1) / 2;
  hipLaunchKernelGGL(k_nms_mask, dim3(blocks), dim3(kNmsBlock), 0, s, boxes, n, thresh, mask, cb);
  int rc = check_launch("k_nms_mask");
  if (rc) return rc;
  hipLaunchKernelGGL(k_nms_reduce, dim3(1), dim3(64), 0, s, mask, n, cb, keep, num_out);
  return check_launch("k_nms_reduce");
}

int omnihd_iou_bev_matrix(const float* boxes_a, int na, const float* boxes_b, int nb, float* out, void* stream) {
  using namespace omnihd;
  OMNIHD_REQUIRE(na >= 0 && nb >= 0, "sizes >= 0");
  if (na == 0 || nb == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(boxes_a && boxes_b && out, "null pointer");
  int64_t total = (int64_t)na * nb;
  hipLaunchKernelGGL(k_iou_matrix, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     boxes_a, na, boxes_b, nb, out);
  return check_launch("k_iou_matrix");
}

}  // extern "C"
