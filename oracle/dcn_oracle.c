/* oracle/dcn_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement of deformable convolution v1 as mmcv-full 1.4.0 computes it (`DeformConv2dPack` = the "DCN" layer the
 * reference builds inside DepthNet: projects/mmdet3d_plugin/bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:587-595,
 * `build_conv_layer(dict(type='DCN', ..., groups=4, im2col_step=128))`).  mmcv is NOT vendored in /root/reference and is not
 * installed in this image, so this file restates the PUBLISHED algorithm of mmcv/ops/csrc/common/cuda/deform_conv_cuda_kernel.cuh
 * (v1.4.0, the version README.md:147 pins) from its documentation and the reference's call site: PARITY UNPINNED against
 * upstream (no reference test or fixture exercises the op); pinned instead by hand-computed known answers in
 * tests/test_oracle.py (zero offsets = plain grouped convolution, integer offsets = shifted taps, the `> -1 / < H` border
 * rule, the (dy, dx) channel order).
 *
 *   deformable_im2col:   for output pixel (b, ho, wo), tap (i, j), input channel c (deformable group g = c / (C / DG)):
 *       off_h = offset[b][g*2*KH*KW + 2*(i*KW + j)    ][ho][wo]
 *       off_w = offset[b][g*2*KH*KW + 2*(i*KW + j) + 1][ho][wo]
 *       h = ho*stride - pad + i*dil + off_h ;  w = wo*stride - pad + j*dil + off_w
 *       col[c][i][j][b][ho][wo] = (h > -1 && w > -1 && h < H && w < W) ? bilinear(x[b][c], h, w) : 0
 *   bilinear: corners (floor, floor + 1); a corner outside [0, H-1] x [0, W-1] contributes 0.
 *   output[b][n][ho][wo] = sum over the input channels of n's group, taps:  weight[n][c'][i][j] * col[...]
 *
 * Backward = the analytic gradient of exactly this function (what mmcv's deformable_col2im / col2im_coord compute):
 *   grad_x      scatter of weight^T * grad_out onto the four corners with their bilinear weights,
 *   grad_offset d bilinear / d h, d w with the same border gate (derivative of the piecewise-linear interpolant inside the
 *               cell [floor, floor + 1]),
 *   grad_weight grad_out x col.
 * Plain loops, double accumulators rounded once to float; layouts are torch's NCHW contiguous.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static float px(const float* img, int H, int W, int y, int x) {
  return (y >= 0 && y < H && x >= 0 && x < W) ? img[(size_t)y * W + x] : 0.0f;
}

/* value and the two partial derivatives of the gated bilinear sample */
static void sample(const float* img, int H, int W, float h, float w, double* val, double* dh, double* dw) {
  *val = *dh = *dw = 0.0;
  if (!(h > -1 && w > -1 && h < H && w < W)) return;
  const int hl = (int)floorf(h), wl = (int)floorf(w);
  const double lh = (double)h - hl, lw = (double)w - wl, hh = 1.0 - lh, hw = 1.0 - lw;
  const double v1 = px(img, H, W, hl, wl), v2 = px(img, H, W, hl, wl + 1);
  const double v3 = px(img, H, W, hl + 1, wl), v4 = px(img, H, W, hl + 1, wl + 1);
  *val = hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
  *dh = hw * (v3 - v1) + lw * (v4 - v2);
  *dw = hh * (v2 - v1) + lh * (v4 - v3);
}

/* x (B,C,H,W), offset (B,DG*2*K*K,Ho,Wo), weight (N, C/G, K, K) -> out (B,N,Ho,Wo) */
void oracle_deform_conv_fwd(const float* x, const float* offset, const float* weight, float* out, int B, int C, int H, int W,
                            int N, int K, int stride, int pad, int dil, int G, int DG, int Ho, int Wo) {
  const int cg = C / G, ng = N / G, cdg = C / DG;
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n) {
      const int g = n / ng;
      for (int ho = 0; ho < Ho; ++ho)
        for (int wo = 0; wo < Wo; ++wo) {
          double acc = 0.0;
          for (int cc = 0; cc < cg; ++cc) {
            const int c = g * cg + cc, dg = c / cdg;
            const float* img = x + ((size_t)b * C + c) * H * W;
            for (int i = 0; i < K; ++i)
              for (int j = 0; j < K; ++j) {
                const size_t ob = (((size_t)b * DG + dg) * 2 * K * K + 2 * (i * K + j)) * Ho * Wo + (size_t)ho * Wo + wo;
                const float h = (float)(ho * stride - pad + i * dil) + offset[ob];
                const float w = (float)(wo * stride - pad + j * dil) + offset[ob + (size_t)Ho * Wo];
                double v, dh, dw;
                sample(img, H, W, h, w, &v, &dh, &dw);
                acc += (double)weight[(((size_t)n * cg + cc) * K + i) * K + j] * v;
              }
          }
          out[(((size_t)b * N + n) * Ho + ho) * Wo + wo] = (float)acc;
        }
    }
}

/* grad_out (B,N,Ho,Wo) -> grad_x (B,C,H,W), grad_offset (B,DG*2*K*K,Ho,Wo), grad_weight (N,C/G,K,K); any output may be NULL */
void oracle_deform_conv_bwd(const float* x, const float* offset, const float* weight, const float* grad_out, float* grad_x,
                            float* grad_offset, float* grad_weight, int B, int C, int H, int W, int N, int K, int stride,
                            int pad, int dil, int G, int DG, int Ho, int Wo) {
  const int cg = C / G, ng = N / G, cdg = C / DG;
  const size_t nx = (size_t)B * C * H * W, no = (size_t)B * DG * 2 * K * K * Ho * Wo, nw = (size_t)N * cg * K * K;
  double* gx = grad_x ? (double*)calloc(nx, sizeof(double)) : NULL;
  double* go = grad_offset ? (double*)calloc(no, sizeof(double)) : NULL;
  double* gw = grad_weight ? (double*)calloc(nw, sizeof(double)) : NULL;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      const int g = c / cg, cc = c - g * cg, dg = c / cdg;
      const float* img = x + ((size_t)b * C + c) * H * W;
      for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j)
          for (int ho = 0; ho < Ho; ++ho)
            for (int wo = 0; wo < Wo; ++wo) {
              const size_t ob = (((size_t)b * DG + dg) * 2 * K * K + 2 * (i * K + j)) * Ho * Wo + (size_t)ho * Wo + wo;
              const float h = (float)(ho * stride - pad + i * dil) + offset[ob];
              const float w = (float)(wo * stride - pad + j * dil) + offset[ob + (size_t)Ho * Wo];
              double v, dh, dw;
              sample(img, H, W, h, w, &v, &dh, &dw);
              /* gradient arriving at col[c][i][j][b][ho][wo]: sum over the output channels of the group */
              double gcol = 0.0;
              for (int q = 0; q < ng; ++q) {
                const int n = g * ng + q;
                const double gout = grad_out[(((size_t)b * N + n) * Ho + ho) * Wo + wo];
                gcol += (double)weight[(((size_t)n * cg + cc) * K + i) * K + j] * gout;
                if (gw) gw[(((size_t)n * cg + cc) * K + i) * K + j] += gout * v;
              }
              if (go) {
                go[ob] += gcol * dh;
                go[ob + (size_t)Ho * Wo] += gcol * dw;
              }
              if (gx && h > -1 && w > -1 && h < H && w < W) {
                const int hl = (int)floorf(h), wl = (int)floorf(w);
                const double lh = (double)h - hl, lw = (double)w - wl;
                const double wt[4] = {(1 - lh) * (1 - lw), (1 - lh) * lw, lh * (1 - lw), lh * lw};
                const int ys[4] = {hl, hl, hl + 1, hl + 1}, xs[4] = {wl, wl + 1, wl, wl + 1};
                for (int k = 0; k < 4; ++k)
                  if (ys[k] >= 0 && ys[k] < H && xs[k] >= 0 && xs[k] < W)
                    gx[((size_t)b * C + c) * H * W + (size_t)ys[k] * W + xs[k]] += gcol * wt[k];
              }
            }
    }
  if (gx) { for (size_t k = 0; k < nx; ++k) grad_x[k] = (float)gx[k]; free(gx); }
  if (go) { for (size_t k = 0; k < no; ++k) grad_offset[k] = (float)go[k]; free(go); }
  if (gw) { for (size_t k = 0; k < nw; ++k) grad_weight[k] = (float)gw[k]; free(gw); }
}
