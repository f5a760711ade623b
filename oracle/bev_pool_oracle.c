/*
 * oracle/bev_pool_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement of the reference's four pooling kernels, one C loop nest per CUDA kernel,
 * iterating "threads" in index order.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (omnihd-scenes_amd/) never does.
 *
 * Pinning: the reference's only known-answer test, test_bev_pool_v2()
 * (projects/mmdet3d_plugin/ops/bev_pool_v2/bev_pool.py:145-176), is replayed against these
 * functions by tests/test_oracle.py, together with golden tables captured by importing the
 * reference Python (tests/golden/make_golden.py).  On the GPU box the functions are also
 * compared with the reference's own kernels compiled by hipcc (oracle/_ref, see
 * oracle/Makefile).
 *
 * Arithmetic note: the CUDA source writes `psum += *cur_feat * *cur_depth;` which nvcc (and
 * hipcc) contract into one fused multiply-add; fmaf() below states that explicitly so the
 * rounding chain is the same on every compiler.
 */
#include <math.h>
#include <stddef.h>

/* ref: ops/bev_pool_v2/src/bev_pool_cuda.cu:21-48 (bev_pool_v2_kernel).
 * thread idx -> (index = idx / c, cur_c = idx % c); serial loop over the interval; plain store
 * at out[ranks_bev[interval_start] * c + cur_c]. */
void oracle_bev_pool_v2_fwd(int c, int n_intervals, const float* depth, const float* feat,
                            const int* ranks_depth, const int* ranks_feat, const int* ranks_bev,
                            const int* interval_starts, const int* interval_lengths, float* out) {
  for (int index = 0; index < n_intervals; ++index) {
    const int interval_start = interval_starts[index];
    const int interval_length = interval_lengths[index];
    for (int cur_c = 0; cur_c < c; ++cur_c) {
      float psum = 0.f;
      for (int i = 0; i < interval_length; ++i) {
        const float* cur_depth = depth + ranks_depth[interval_start + i];
        const float* cur_feat = feat + (size_t)ranks_feat[interval_start + i] * c + cur_c;
        psum = fmaf(*cur_feat, *cur_depth, psum);
      }
      out[(size_t)ranks_bev[interval_start] * c + cur_c] = psum;
    }
  }
}

/* ref: ops/bev_pool_v2/src/bev_pool_cuda.cu:67-121 (bev_pool_grad_kernel).
 * One thread per (backward) interval.  Loop 1 (:91-105): depth_grad of every point = dot over
 * channels.  Loop 2 (:109-120): feat_grad of the interval's pixel = sum over its points. */
void oracle_bev_pool_v2_bwd(int c, int n_intervals, const float* out_grad, const float* depth,
                            const float* feat, const int* ranks_depth, const int* ranks_feat,
                            const int* ranks_bev, const int* interval_starts,
                            const int* interval_lengths, float* depth_grad, float* feat_grad) {
  for (int idx = 0; idx < n_intervals; ++idx) {
    const int interval_start = interval_starts[idx];
    const int interval_length = interval_lengths[idx];
    for (int i = 0; i < interval_length; ++i) {
      const float* cur_out_grad_start = out_grad + (size_t)ranks_bev[interval_start + i] * c;
      const float* cur_feat_start = feat + (size_t)ranks_feat[interval_start + i] * c;
      float grad_sum = 0.f;
      for (int cur_c = 0; cur_c < c; ++cur_c)
        grad_sum = fmaf(cur_out_grad_start[cur_c], cur_feat_start[cur_c], grad_sum);
      depth_grad[ranks_depth[interval_start + i]] = grad_sum;
    }
    for (int cur_c = 0; cur_c < c; ++cur_c) {
      float grad_sum = 0.f;
      for (int i = 0; i < interval_length; ++i) {
        const float* cur_out_grad = out_grad + (size_t)ranks_bev[interval_start + i] * c + cur_c;
        const float* cur_depth = depth + ranks_depth[interval_start + i];
        grad_sum = fmaf(*cur_out_grad, *cur_depth, grad_sum);
      }
      feat_grad[(size_t)ranks_feat[interval_start] * c + cur_c] = grad_sum;
    }
  }
}

/* ref: ops/bev_pool/src/bev_pool_cuda.cu:20-42 (bev_pool_kernel, v1).
 * geom_feats row = (g0,g1,g2,g3) = (h_idx, w_idx, d_idx, b_idx); out is [b,d,h,w,c]. */
void oracle_bev_pool_v1_fwd(int b, int d, int h, int w, int n, int c, int n_intervals,
                            const float* x, const int* geom_feats, const int* interval_starts,
                            const int* interval_lengths, float* out) {
  (void)b; (void)n;
  for (int index = 0; index < n_intervals; ++index) {
    const int interval_start = interval_starts[index];
    const int interval_length = interval_lengths[index];
    const int* g = geom_feats + (size_t)interval_start * 4;
    for (int cur_c = 0; cur_c < c; ++cur_c) {
      const float* cur_x = x + (size_t)interval_start * c + cur_c;
      float* cur_out = out + (size_t)g[3] * d * h * w * c + (size_t)g[2] * h * w * c +
                       (size_t)g[0] * w * c + (size_t)g[1] * c + cur_c;
      float psum = 0.f;
      for (int i = 0; i < interval_length; ++i) psum += cur_x[(size_t)i * c];
      *cur_out = psum;
    }
  }
}

/* ref: ops/bev_pool/src/bev_pool_cuda.cu:61-84 (bev_pool_grad_kernel, v1): broadcast. */
void oracle_bev_pool_v1_bwd(int b, int d, int h, int w, int n, int c, int n_intervals,
                            const float* out_grad, const int* geom_feats,
                            const int* interval_starts, const int* interval_lengths,
                            float* x_grad) {
  (void)b; (void)n;
  for (int index = 0; index < n_intervals; ++index) {
    const int interval_start = interval_starts[index];
    const int interval_length = interval_lengths[index];
    const int* g = geom_feats + (size_t)interval_start * 4;
    for (int cur_c = 0; cur_c < c; ++cur_c) {
      float* cur_x_grad = x_grad + (size_t)interval_start * c + cur_c;
      const float* cur_out_grad = out_grad + (size_t)g[3] * d * h * w * c +
                                  (size_t)g[2] * h * w * c + (size_t)g[0] * w * c +
                                  (size_t)g[1] * c + cur_c;
      for (int i = 0; i < interval_length; ++i) cur_x_grad[(size_t)i * c] = *cur_out_grad;
    }
  }
}

/* Same arithmetic as oracle_bev_pool_v2_fwd / _bwd, OpenMP over intervals: the cpu_baseline leg
 * of bench.py times these on all host cores. */
void oracle_bev_pool_v2_fwd_omp(int c, int n_intervals, const float* depth, const float* feat,
                                const int* ranks_depth, const int* ranks_feat,
                                const int* ranks_bev, const int* interval_starts,
                                const int* interval_lengths, float* out) {
#pragma omp parallel for schedule(dynamic, 256)
  for (int index = 0; index < n_intervals; ++index)
    oracle_bev_pool_v2_fwd(c, 1, depth, feat, ranks_depth, ranks_feat, ranks_bev,
                           interval_starts + index, interval_lengths + index, out);
}

void oracle_bev_pool_v2_bwd_omp(int c, int n_intervals, const float* out_grad,
                                const float* depth, const float* feat, const int* ranks_depth,
                                const int* ranks_feat, const int* ranks_bev,
                                const int* interval_starts, const int* interval_lengths,
                                float* depth_grad, float* feat_grad) {
#pragma omp parallel for schedule(dynamic, 256)
  for (int idx = 0; idx < n_intervals; ++idx)
    oracle_bev_pool_v2_bwd(c, 1, out_grad, depth, feat, ranks_depth, ranks_feat, ranks_bev,
                           interval_starts + idx, interval_lengths + idx, depth_grad, feat_grad);
}
