// Anchor target assignment + the three detection losses of Anchor3DHead in THREE launches (and one for the backward).
//
// Where it sits: the tail of BEVFUSION_depth.forward_train (reference bevfusion/detectors/bevf_faster_rcnn_bevdepth.py:178-232 ->
// forward_pts_train -> Anchor3DHead.loss; the vendored copy of the head: bevfusion/dense_heads/det_anchor3d_head.py:192-372;
// config projects/configs/bevfusion_NewScenes/bevfusion.py:96-155).  The torch formulation of this library already avoids every
// host synchronisation (mask-based assigner, losses over all anchors with zero weight off the positives) but costs ~150 launches
// of 5-20 us per step on 307 200 anchors x 30 boxes; the bf16 step is host-bound, so only FEWER launches help there.
//
//   k_anchor_gt_max   per (sample, anchor): IoU of the nearest axis-aligned BEV boxes (BboxOverlapsNearest3D) against the sample's
//                     boxes; per-box maximum over all anchors (atomicMax on the bit pattern of a non-negative float: the maximum
//                     does not depend on the order of arrival)
//   k_anchor_loss     per (sample, anchor): the same IoUs again, MaxIoUAssigner (pos / neg thresholds, every anchor that reaches a
//                     box's best IoU is matched to it, later boxes win), DeltaXYZWLHRBBoxCoder targets, direction bin; sigmoid focal
//                     loss over the classes, smooth-L1 over the box code with the sine-difference encoding of the yaw, 2-way
//                     cross-entropy of the direction bin; the UNSCALED gradients of the three prediction maps in the same pass;
//                     fixed-order block sums of the three losses and of the positives
//   k_anchor_finalize one workgroup: positives per sample (clamped to >= 1, summed over the batch: the reference's avg_factor),
//                     the three losses
//   k_anchor_scale    backward: the three gradient maps times (upstream gradient x loss weight / avg_factor)
// No atomics on sums: results are run-to-run identical.
#include "common.h"

// No fused multiply-adds in this file: the assigner compares IoUs computed in two different kernels for EQUALITY (an anchor that
// reaches a box's best IoU), so both must round every operation the same way — and the torch formulation this replaces rounds
// every operation separately too.
#pragma clang fp contract(off)

namespace omnihd {
namespace {

constexpr int kBlk = 256;
constexpr int kMaxGt = 128;      // boxes per sample held in LDS
constexpr int kMaxCode = 12;     // box code size (7 + custom values)
constexpr int kMaxCls = 8;

struct AnchorLossCfg {
  int B, A, NA, H, W, K, CS;                 // samples, anchors per sample (= H*W*NA), anchors per location, map size, classes, code size
  long long cls_sb, cls_sc, cls_sy, cls_sx;  // element strides of the (B, NA*K, H, W) class map
  long long box_sb, box_sc, box_sy, box_sx;  // ... of the (B, NA*CS, H, W) regression map
  long long dir_sb, dir_sc, dir_sy, dir_sx;  // ... of the (B, NA*2, H, W) direction map
  float pos_thr, neg_thr, min_pos;
  float gamma, alpha, beta, dir_offset;
  int sin_diff;
  float code_w[kMaxCode];
};

__device__ __forceinline__ float limit_period_f(float v, float offset, float period) {
  return v - floorf(v / period + offset) * period;
}

// nearest axis-aligned BEV box of (x, y, w, l, yaw): (x1, y1, x2, y2)
__device__ __forceinline__ float4 nearest_bev(float x, float y, float w, float l, float yaw) {
  const float kPi = 3.14159265358979323846f;
  const float rot = fabsf(limit_period_f(yaw, 0.5f, kPi));
  const bool swap = rot > kPi / 4;
  const float d0 = swap ? l : w, d1 = swap ? w : l;
  return make_float4(x - d0 / 2, y - d1 / 2, x + d0 / 2, y + d1 / 2);
}

__device__ __forceinline__ float iou_aligned(const float4 g, float area_g, const float4 a, float area_a) {
  const float lx = fmaxf(g.x, a.x), ly = fmaxf(g.y, a.y);
  const float rx = fminf(g.z, a.z), ry = fminf(g.w, a.w);
  const float w = fmaxf(rx - lx, 0.f), h = fmaxf(ry - ly, 0.f);
  const float overlap = w * h;
  const float uni = fmaxf(area_g + area_a - overlap, 1e-6f);
  return overlap / uni;
}

struct GtLds {
  float4 box[kMaxGt];
  float area[kMaxGt];
};

__device__ __forceinline__ int load_gts(GtLds& s, const float* __restrict__ gt, const int* __restrict__ gt_off, int b, int cs) {
  const int g0 = gt_off[b], n = min(gt_off[b + 1] - g0, kMaxGt);
  for (int g = threadIdx.x; g < n; g += kBlk) {
    const float* r = gt + (size_t)(g0 + g) * cs;
    const float4 bx = nearest_bev(r[0], r[1], r[3], r[4], r[6]);
    s.box[g] = bx;
    s.area[g] = (bx.z - bx.x) * (bx.w - bx.y);
  }
  __syncthreads();
  return n;
}

__global__ __launch_bounds__(kBlk) void k_anchor_gt_max(const float* __restrict__ anchors, const float* __restrict__ gt,
                                                         const int* __restrict__ gt_off, AnchorLossCfg c, int blocks_per_sample,
                                                         int* __restrict__ gt_max_bits) {
  __shared__ GtLds s;
  __shared__ int blk_max[kMaxGt];
  const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x - b * blocks_per_sample;
  const int n_gt = load_gts(s, gt, gt_off, b, c.CS);
  if (n_gt == 0) return;
  for (int g = threadIdx.x; g < n_gt; g += kBlk) blk_max[g] = 0;
  __syncthreads();
  const int n = blk * kBlk + threadIdx.x;
  if (n < c.A) {
    const float* a = anchors + (size_t)n * c.CS;
    const float4 ab = nearest_bev(a[0], a[1], a[3], a[4], a[6]);
    const float area_a = (ab.z - ab.x) * (ab.w - ab.y);
    for (int g = 0; g < n_gt; ++g) {
      const float v = iou_aligned(s.box[g], s.area[g], ab, area_a);
      if (v > 0.f) atomicMax(&blk_max[g], __float_as_int(v));
    }
  }
  __syncthreads();
  const int g0 = gt_off[b];
  for (int g = threadIdx.x; g < n_gt; g += kBlk)
    if (blk_max[g] > 0) atomicMax(&gt_max_bits[g0 + g], blk_max[g]);
}

__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

__global__ __launch_bounds__(kBlk) void k_anchor_loss(
    const float* __restrict__ anchors, const float* __restrict__ gt, const int* __restrict__ gt_labels, const int* __restrict__ gt_off,
    const int* __restrict__ gt_max_bits, const float* __restrict__ cls, const float* __restrict__ box, const float* __restrict__ dir,
    AnchorLossCfg c, int blocks_per_sample, float* __restrict__ g_cls, float* __restrict__ g_box, float* __restrict__ g_dir,
    float* __restrict__ partials /* [blocks][4]: loss_cls, loss_bbox, loss_dir, positives */) {
  __shared__ GtLds s;
  __shared__ float gmax[kMaxGt];
  __shared__ float red[4][kBlk];
  const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x - b * blocks_per_sample;
  const int n_gt = load_gts(s, gt, gt_off, b, c.CS);
  const int g0 = gt_off[b];
  for (int g = threadIdx.x; g < n_gt; g += kBlk) gmax[g] = __int_as_float(gt_max_bits[g0 + g]);
  __syncthreads();
  const int n = blk * kBlk + threadIdx.x;
  float l_cls = 0.f, l_box = 0.f, l_dir = 0.f, n_pos = 0.f;
  if (n < c.A) {
    const float* a = anchors + (size_t)n * c.CS;
    float av[kMaxCode];
#pragma unroll
    for (int k = 0; k < kMaxCode; ++k) av[k] = k < c.CS ? a[k] : 0.f;
    // ---- MaxIoUAssigner (mmdet 2.14 semantics restated in mm/anchor_head.py::MaxIoUAssigner.assign) ----------------------
    int assigned = 0;                                      // no boxes in the sample: every anchor is a negative
    if (n_gt > 0) {
      const float4 ab = nearest_bev(av[0], av[1], av[3], av[4], av[6]);
      const float area_a = (ab.z - ab.x) * (ab.w - ab.y);
      float max_ov = -1.f;
      int argmax = 0, last = 0;
      for (int g = 0; g < n_gt; ++g) {
        const float v = iou_aligned(s.box[g], s.area[g], ab, area_a);
        if (v > max_ov) { max_ov = v; argmax = g; }
        if (v == gmax[g] && gmax[g] >= c.min_pos) last = g + 1;      // every anchor reaching box g's best IoU; later boxes win
      }
      assigned = -1;
      if (max_ov >= 0.f && max_ov < c.neg_thr) assigned = 0;
      if (max_ov >= c.pos_thr) assigned = argmax + 1;
      if (last > 0) assigned = last;
    }
    const bool pos = assigned > 0, neg = assigned == 0;
    const int y = n / (c.W * c.NA), rem = n - y * (c.W * c.NA);
    const int x = rem / c.NA, na = rem - x * c.NA;
    // ---- classification: sigmoid focal loss over the K classes of this anchor, weight = pos | neg ------------------------
    const int label = pos ? gt_labels[g0 + assigned - 1] : c.K;
    const float lw = (pos || neg) ? 1.f : 0.f;
    const size_t cls_at = (size_t)b * c.cls_sb + (size_t)y * c.cls_sy + (size_t)x * c.cls_sx;
    for (int k = 0; k < c.K; ++k) {
      const size_t at = cls_at + (size_t)(na * c.K + k) * c.cls_sc;
      const float xv = cls[at];
      const float p = 1.f / (1.f + expf(-xv));
      float loss, grad;
      if (k == label) {
        const float q = 1.f - p;                            // pt
        const float nlogp = softplus_f(-xv);                // -log p
        const float f = c.alpha * powf(q, c.gamma);
        loss = nlogp * f;
        grad = f * (-c.gamma * p * nlogp - q);              // alpha (1-p)^g [g p log p - (1-p)]
      } else {
        const float nlog1p = softplus_f(xv);                // -log (1-p)
        const float f = (1.f - c.alpha) * powf(p, c.gamma);
        loss = nlog1p * f;
        grad = f * (p + c.gamma * (1.f - p) * nlog1p);      // (1-alpha) p^g [p - g (1-p) log(1-p)]
      }
      l_cls += loss * lw;
      g_cls[at] = grad * lw;
    }
    // ---- regression + direction: zero weight off the positives ------------------------------------------------------------
    const size_t box_at = (size_t)b * c.box_sb + (size_t)y * c.box_sy + (size_t)x * c.box_sx;
    const size_t dir_at = (size_t)b * c.dir_sb + (size_t)y * c.dir_sy + (size_t)x * c.dir_sx;
    if (pos) {
      n_pos = 1.f;
      const float* gr = gt + (size_t)(g0 + assigned - 1) * c.CS;
      float t[kMaxCode];
      {   // DeltaXYZWLHRBBoxCoder.encode(anchor, box)
        const float xa = av[0], ya = av[1], ha = av[5], wa = av[3], la = av[4], ra = av[6];
        const float za = av[2] + ha / 2;
        const float hg = gr[5];
        const float zg = gr[2] + hg / 2;
        const float diag = sqrtf(la * la + wa * wa);
        t[0] = (gr[0] - xa) / diag; t[1] = (gr[1] - ya) / diag; t[2] = (zg - za) / ha;
        t[3] = logf(gr[3] / wa); t[4] = logf(gr[4] / la); t[5] = logf(hg / ha); t[6] = gr[6] - ra;
#pragma unroll
        for (int k = 7; k < kMaxCode; ++k) t[k] = k < c.CS ? gr[k] - av[k] : 0.f;
      }
      // direction bin of the target yaw
      const float kPi = 3.14159265358979323846f;
      const float rot_gt = t[6] + av[6];
      const float off = limit_period_f(rot_gt - c.dir_offset, 0.f, 2 * kPi);
      int dbin = (int)floorf(off / kPi);
      dbin = dbin < 0 ? 0 : (dbin > 1 ? 1 : dbin);
#pragma unroll
      for (int k = 0; k < kMaxCode; ++k) {
        if (k < c.CS) {
          const size_t at = box_at + (size_t)(na * c.CS + k) * c.box_sc;
          const float pv = box[at];
          float d, dd = 1.f;                                 // d = pred' - target', dd = d(pred')/d(pred)
          if (k == 6 && c.sin_diff) {
            float sp, cp, stv, ctv;
            sincosf(pv, &sp, &cp);
            sincosf(t[6], &stv, &ctv);
            d = sp * ctv - cp * stv;                         // sin(p) cos(t) - cos(p) sin(t)
            dd = cp * ctv + sp * stv;
          } else {
            d = pv - t[k];
          }
          const float ad = fabsf(d);
          const float w = c.code_w[k];
          float loss, grad;
          if (ad < c.beta) { loss = 0.5f * ad * ad / c.beta; grad = d / c.beta; }
          else { loss = ad - 0.5f * c.beta; grad = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
          l_box += loss * w;
          g_box[at] = grad * dd * w;
        }
      }
      const size_t a0 = dir_at + (size_t)(na * 2 + 0) * c.dir_sc, a1 = dir_at + (size_t)(na * 2 + 1) * c.dir_sc;
      const float d0 = dir[a0], d1 = dir[a1];
      const float m = fmaxf(d0, d1);
      const float e0 = expf(d0 - m), e1 = expf(d1 - m);
      const float lse = m + logf(e0 + e1);
      l_dir = lse - (dbin == 0 ? d0 : d1);
      const float inv = 1.f / (e0 + e1);
      g_dir[a0] = e0 * inv - (dbin == 0 ? 1.f : 0.f);
      g_dir[a1] = e1 * inv - (dbin == 1 ? 1.f : 0.f);
    } else {
#pragma unroll
      for (int k = 0; k < kMaxCode; ++k)
        if (k < c.CS) g_box[box_at + (size_t)(na * c.CS + k) * c.box_sc] = 0.f;
      g_dir[dir_at + (size_t)(na * 2 + 0) * c.dir_sc] = 0.f;
      g_dir[dir_at + (size_t)(na * 2 + 1) * c.dir_sc] = 0.f;
    }
  }
  red[0][threadIdx.x] = l_cls; red[1][threadIdx.x] = l_box; red[2][threadIdx.x] = l_dir; red[3][threadIdx.x] = n_pos;
  __syncthreads();
  for (int off = kBlk / 2; off > 0; off >>= 1) {            // fixed-order tree: run-to-run identical sums
    if (threadIdx.x < off) {
#pragma unroll
      for (int q = 0; q < 4; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) partials[(size_t)blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

// one workgroup: out[0..2] = the three losses (x their weights), out[3] = avg_factor, out[4 + b] = positives of sample b
__global__ __launch_bounds__(kBlk) void k_anchor_finalize(const float* __restrict__ partials, int B, int blocks_per_sample, float w_cls,
                                                           float w_box, float w_dir, float* __restrict__ out) {
  __shared__ double red[4][kBlk];
  __shared__ double tot[4];
  if (threadIdx.x < 4) tot[threadIdx.x] = 0.0;
  __syncthreads();
  for (int b = 0; b < B; ++b) {
    double acc[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < blocks_per_sample; i += kBlk)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += (double)partials[((size_t)b * blocks_per_sample + i) * 4 + q];
#pragma unroll
    for (int q = 0; q < 4; ++q) red[q][threadIdx.x] = acc[q];
    __syncthreads();
    for (int off = kBlk / 2; off > 0; off >>= 1) {
      if (threadIdx.x < off)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      tot[0] += red[0][0]; tot[1] += red[1][0]; tot[2] += red[2][0];
      tot[3] += red[3][0] < 1.0 ? 1.0 : red[3][0];          // num_total_samples: per-sample count clamped to >= 1
      out[4 + b] = (float)red[3][0];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float avg = (float)tot[3];
    out[0] = w_cls * ((float)tot[0] / avg);
    out[1] = w_box * ((float)tot[1] / avg);
    out[2] = w_dir * ((float)tot[2] / avg);
    out[3] = avg;
  }
}

// gradient maps (dense copies of the three maps' memory, n elements each) *= upstream[k] * weight[k] / avg
__global__ __launch_bounds__(kBlk) void k_anchor_scale(float* __restrict__ g_cls, long long n_cls, float* __restrict__ g_box, long long n_box,
                                                        float* __restrict__ g_dir, long long n_dir, const float* __restrict__ up_cls,
                                                        const float* __restrict__ up_box, const float* __restrict__ up_dir,
                                                        const float* __restrict__ fin, float w_cls, float w_box, float w_dir) {
  const float avg = fin[3];
  const float s0 = (up_cls ? *up_cls : 0.f) * w_cls / avg, s1 = (up_box ? *up_box : 0.f) * w_box / avg, s2 = (up_dir ? *up_dir : 0.f) * w_dir / avg;
  const long long total = n_cls + n_box + n_dir;
  for (long long i = (long long)blockIdx.x * kBlk + threadIdx.x; i < total; i += (long long)gridDim.x * kBlk) {
    if (i < n_cls) g_cls[i] *= s0;
    else if (i < n_cls + n_box) g_box[i - n_cls] *= s1;
    else g_dir[i - n_cls - n_box] *= s2;
  }
}

}  // namespace
}  // namespace omnihd

using namespace omnihd;

extern "C" size_t omnihd_anchor_loss_workspace_bytes(int batch, int anchors_per_sample, int total_gt) {
  if (batch <= 0 || anchors_per_sample <= 0 || total_gt < 0) return 0;
  const size_t blocks = (size_t)batch * ((anchors_per_sample + kBlk - 1) / kBlk);
  return align_up(blocks * 4 * sizeof(float), 256) + align_up((size_t)(total_gt + 1) * sizeof(int), 256);
}

extern "C" int omnihd_anchor_loss_fwd(const float* anchors, const float* gt_boxes, const int* gt_labels, const int* gt_offsets,
                                      int total_gt, const float* cls_score, const float* bbox_pred, const float* dir_pred, int batch,
                                      int h, int w, int anchors_per_loc, int num_classes, int code_size, const long long* h_strides12,
                                      const float* h_params7 /* pos_thr, neg_thr, min_pos, gamma, alpha, beta, dir_offset */,
                                      int sin_diff, const float* h_code_weight, const float* h_loss_weights3, float* g_cls,
                                      float* g_box, float* g_dir, float* out /* 4 + batch floats */, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  OMNIHD_REQUIRE(batch > 0 && h > 0 && w > 0 && anchors_per_loc > 0 && num_classes > 0 && num_classes <= kMaxCls && code_size >= 7 &&
                 code_size <= kMaxCode && total_gt >= 0, "sizes (classes <= 8, 7 <= code size <= 12)");
  OMNIHD_REQUIRE(anchors && gt_offsets && cls_score && bbox_pred && dir_pred && h_strides12 && h_params7 && h_loss_weights3 && g_cls &&
                 g_box && g_dir && out && workspace && (total_gt == 0 || (gt_boxes && gt_labels)), "null pointer");
  const long long A = (long long)h * w * anchors_per_loc;
  OMNIHD_REQUIRE(A < (1ll << 31), "anchors per sample");
  const size_t need = omnihd_anchor_loss_workspace_bytes(batch, (int)A, total_gt);
  if (workspace_bytes < need) { set_error("anchor_loss: workspace %zu < required %zu", workspace_bytes, need); return OMNIHD_ERR_WORKSPACE; }
  AnchorLossCfg c;
  c.B = batch; c.A = (int)A; c.NA = anchors_per_loc; c.H = h; c.W = w; c.K = num_classes; c.CS = code_size;
  c.cls_sb = h_strides12[0]; c.cls_sc = h_strides12[1]; c.cls_sy = h_strides12[2]; c.cls_sx = h_strides12[3];
  c.box_sb = h_strides12[4]; c.box_sc = h_strides12[5]; c.box_sy = h_strides12[6]; c.box_sx = h_strides12[7];
  c.dir_sb = h_strides12[8]; c.dir_sc = h_strides12[9]; c.dir_sy = h_strides12[10]; c.dir_sx = h_strides12[11];
  c.pos_thr = h_params7[0]; c.neg_thr = h_params7[1]; c.min_pos = h_params7[2];
  c.gamma = h_params7[3]; c.alpha = h_params7[4]; c.beta = h_params7[5]; c.dir_offset = h_params7[6];
  c.sin_diff = sin_diff;
  for (int k = 0; k < kMaxCode; ++k) c.code_w[k] = (h_code_weight && k < code_size) ? h_code_weight[k] : 1.f;
  hipStream_t st = (hipStream_t)stream;
  const int bps = (int)((A + kBlk - 1) / kBlk);
  float* partials = static_cast<float*>(workspace);
  int* gt_max = reinterpret_cast<int*>(static_cast<char*>(workspace) + align_up((size_t)batch * bps * 4 * sizeof(float), 256));
  OMNIHD_HIP_TRY(hipMemsetAsync(gt_max, 0, (size_t)(total_gt + 1) * sizeof(int), st));
  if (total_gt > 0)
    hipLaunchKernelGGL(k_anchor_gt_max, dim3(batch * bps), dim3(kBlk), 0, st, anchors, gt_boxes, gt_offsets, c, bps, gt_max);
  hipLaunchKernelGGL(k_anchor_loss, dim3(batch * bps), dim3(kBlk), 0, st, anchors, gt_boxes, gt_labels, gt_offsets, gt_max, cls_score, bbox_pred,
                     dir_pred, c, bps, g_cls, g_box, g_dir, partials);
  hipLaunchKernelGGL(k_anchor_finalize, dim3(1), dim3(kBlk), 0, st, partials, batch, bps, h_loss_weights3[0], h_loss_weights3[1],
                     h_loss_weights3[2], out);
  return check_launch("anchor_loss_fwd");
}

extern "C" int omnihd_anchor_loss_bwd(float* g_cls, long long n_cls, float* g_box, long long n_box, float* g_dir, long long n_dir,
                                      const float* up_cls, const float* up_box, const float* up_dir, const float* fin,
                                      const float* h_loss_weights3, void* stream) {
  OMNIHD_REQUIRE(g_cls && g_box && g_dir && fin && h_loss_weights3 && n_cls >= 0 && n_box >= 0 && n_dir >= 0, "arguments");
  const long long total = n_cls + n_box + n_dir;
  if (total == 0) return OMNIHD_OK;
  hipLaunchKernelGGL(k_anchor_scale, dim3(grid_for(total, kBlk * 4)), dim3(kBlk), 0, (hipStream_t)stream, g_cls, n_cls, g_box, n_box, g_dir,
                     n_dir, up_cls, up_box, up_dir, fin, h_loss_weights3[0], h_loss_weights3[1], h_loss_weights3[2]);
  return check_launch("anchor_loss_bwd");
}
