/*
 * omnihd_hip.h — C ABI of libomnihd_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary for the camera + 4D-radar BEV-fusion hot path of
 * TJRadarLab/OmniHD-Scenes.  Every entry point takes plain device pointers, sizes and a
 * hipStream_t passed as void*; nothing here depends on torch.  All functions return
 * OMNIHD_OK (0) or a negative omnihd_status and never throw across the boundary.
 * A human readable message for the last failure on the calling thread is available from
 * omnihd_last_error().
 *
 * "ref:" lines cite the reference interface (path relative to the reference repo root)
 * that each entry point replaces.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name starts with h_;
 *   - `stream` is a hipStream_t (0 = the null stream); work is only enqueued, never
 *     synchronised, except where a function documents a host read-back;
 *   - tables are int32, features are float32 (the reference forces both:
 *     ops/bev_pool_v2/bev_pool.py:19-25);
 *   - workspaces are caller-owned scratch; ask the matching *_workspace_bytes() first.
 */
#ifndef OMNIHD_HIP_H_
#define OMNIHD_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum omnihd_status {
  OMNIHD_OK = 0,
  OMNIHD_ERR_ARG = -1,       /* null pointer / negative size / unsupported shape   */
  OMNIHD_ERR_LAUNCH = -2,    /* hipGetLastError() after a launch was not hipSuccess */
  OMNIHD_ERR_WORKSPACE = -3, /* workspace too small                                  */
  OMNIHD_ERR_RUNTIME = -4    /* any other HIP runtime failure                        */
} omnihd_status;

/* Library identification / diagnostics. */
const char* omnihd_version(void);
const char* omnihd_last_error(void);
/* Number of HIP devices visible to the library (0 when no GPU): lets host code fail loudly. */
int omnihd_device_count(void);

/* Streaming read of up to 4 device buffers (16-byte aligned, `bytes[i]` each) on `stream`: brings static tables (the
 * pooling plan's rank tables) into the L2 / Infinity Cache ahead of the kernel that walks them.  A hint only — results
 * never depend on it.  (No reference counterpart: the reference rebuilds its tables every forward.)                  */
int omnihd_prefetch(const void* const* ptrs, const size_t* bytes, int n, void* stream);

/* ------------------------------------------------------------------------------------------
 * bev_pool_v2 — LSS "BEVPoolv2" (depth x feature gather by rank tables, segment-sum per voxel)
 * ---------------------------------------------------------------------------------------- */

/* ref: projects/mmdet3d_plugin/ops/bev_pool_v2/src/bev_pool.cpp:30-57 (bev_pool_v2_forward)
 *      -> bev_pool_cuda.cu:21-48 (bev_pool_v2_kernel), launcher :125-131.
 * out[ranks_bev[s]*c + ch] = sum_{i<len} depth[ranks_depth[s+i]] * feat[ranks_feat[s+i]*c + ch]
 * for every interval (s = interval_starts[k], len = interval_lengths[k]).  `out` rows that no
 * interval names are NOT touched (the caller zero-fills, as bev_pool.py:27 does).
 * Summation runs in table order with one fused multiply-add per point (same chain as the
 * reference kernel), except for intervals longer than 512 points, which are split over the
 * lanes of one workgroup and combined in a fixed order (run-to-run deterministic).        */
int omnihd_bev_pool_v2_fwd(const float* depth, const float* feat,
                           const int* ranks_depth, const int* ranks_feat, const int* ranks_bev,
                           const int* interval_starts, const int* interval_lengths,
                           float* out, int c, int n_intervals, void* stream);

/* ref: ops/bev_pool_v2/src/bev_pool.cpp:74-104 (bev_pool_v2_backward)
 *      -> bev_pool_cuda.cu:67-121 (bev_pool_grad_kernel), launcher :133-140.
 * Tables are the BACKWARD tables: sorted so that an interval is a run of equal ranks_feat
 * (bev_pool.py:47-57).  depth_grad[ranks_depth[p]] = sum_ch out_grad[ranks_bev[p]*c+ch] *
 * feat[ranks_feat[p]*c+ch];  feat_grad[ranks_feat[s]*c+ch] = sum_p out_grad[ranks_bev[p]*c+ch]
 * * depth[ranks_depth[p]].  Entries not named by the tables are NOT touched.               */
int omnihd_bev_pool_v2_bwd(const float* out_grad, const float* depth, const float* feat,
                           const int* ranks_depth, const int* ranks_feat, const int* ranks_bev,
                           const int* interval_starts, const int* interval_lengths,
                           float* depth_grad, float* feat_grad, int c, int n_intervals,
                           void* stream);

/* Dense forward for ANY channel count (no reference counterpart: it removes the reference's zero-fill (bev_pool.py:27),
 * permute copy (:91) and s2c concat copy (bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:374-376)).
 * The points are grouped by OUTPUT ROW in CSR form: row r owns points [row_ptr[r], row_ptr[r+1]) of ranks_depth / ranks_feat.
 * Every one of the n_rows rows is written (zeros for empty rows), so `out` needs no initialisation.  The row numbering is
 * whatever the plan builder chose (reference (b,z,y,x) order, or (b,y,x,z) = channels-last of the s2c tensor).  One group
 * of C/4 lanes per row, long rows split over the workgroup (the kernel of omnihd_bev_pool_v2_fwd); C = 64 — every
 * configuration of the reference — runs omnihd_bev_pool_v2_fwd_direct instead.  (The LDS-staged tiled kernels of rounds 2-4
 * behind this entry point and omnihd_bev_pool_v2_fwd_lean were superseded: scripts/lab/patches/pool_superseded_kernels.patch.) */
int omnihd_bev_pool_v2_fwd_csr(const float* depth, const float* feat, const int* ranks_depth, const int* ranks_feat,
                               const int* row_ptr, float* out, int c, int n_rows, int n_points, void* stream);

/* The dense forward of the product, C = 64 (round 4): the row sequence is cut into TILES of ~768 rows + points (runs of whole
 * rows, a row of more than 512 points alone: omnihd_csr_tiles) scheduled per XCD (omnihd_tile_desc); every group of 16 lanes
 * walks its own piece of the tile's point list straight from global memory (csrc/bev_pool_v2.hip, k_pool_fwd_direct).
 * empty_rows_kept != 0: `out` is the buffer of an earlier launch with the SAME tables (or a zero-filled one) that nobody has
 * written to since — its empty rows are zero already and are not stored again.
 *   pt        (n_points ints)   ranks_depth | closing << 31, closing = the point is the last one of its output row;
 *   ivl_rel   (n_intervals ints) output row of the k-th non-empty row of the launch, relative to the first row of its tile;
 *   desc32    (n_slots x 32 ints, n_slots = 8*k, 16-byte aligned) launch schedule, entry [x*k + i] = the i-th tile of XCD x:
 *             {first row, #rows, first point, #points, 0, 0, 0, 0, g[16], 0 x 8} with g[j] = number of non-empty rows of the
 *             launch that close before point first_point + min(j*w, #points), w = ceil(#points/16), | 1<<31 when that
 *             point continues the row of the point in front of it inside the same tile;
 *   row_ptr   CSR over all rows (only read to zero-fill empty rows: may be NULL when empty_rows_kept != 0).
 * Rows cut by the in-tile split are combined in a fixed order (a row's sum may be associated differently from table order;
 * run-to-run identical; no atomics); tile ORDER only affects speed.  ref: replaces ops/bev_pool_v2/src/bev_pool_cuda.cu:21-48 + bev_pool.py:27,91. */
int omnihd_bev_pool_v2_fwd_direct(const float* depth, const float* feat, const int* pt, const int* ivl_rel, int n_intervals,
                                  const int* desc32, int n_slots, const int* row_ptr, float* out, int c, int n_rows,
                                  int n_points, int d_bins, int fhw, int n_feat_rows, int empty_rows_kept, void* stream);

/* The same kernel on a plan that was built on the device (omnihd_pool_plan_build below): the number of schedule slots is not
 * known to the host.  hdr (device, int[32]): hdr[3] = slots per XCD.  launch_slots = 8*k workgroups are launched, k >= hdr[3]
 * (the plan's capacity, or the exact value once the host has learned it); surplus workgroups leave at once.
 * ivl_capacity: ints allocated behind ivl_rel.  empty_rows_mode: 0 = zero-fill every empty row; 1 = `out` was filled by the
 * SAME tables last (empty rows are zero already); 2 = `out` was filled last by tables whose row CSR is prev_row_ptr: a row
 * that is empty now is zero-filled only if it was not empty then (per-frame calibrations: a handful of rows per launch).
 * ref: replaces ops/bev_pool_v2/src/bev_pool_cuda.cu:21-48 + bev_pool.py:27,91 for tables rebuilt every forward
 * (bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:283-300). */
int omnihd_bev_pool_v2_fwd_direct_dev(const float* depth, const float* feat, const int* pt, const int* ivl_rel,
                                      long long ivl_capacity, const int* desc32, const int* hdr, int launch_slots,
                                      const int* row_ptr, const int* prev_row_ptr, float* out, int c, int n_rows, int d_bins,
                                      int fhw, int n_feat_rows, int empty_rows_mode, void* stream);

/* Pooling plan for a NEW camera calibration, built entirely on the device: no host read-back, no synchronisation; every
 * launch goes to `stream`.  ref: voxel_pooling_prepare_v2 (bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:302-362,
 * called per forward at :283-300) + the re-sort of QuickCumsumCuda.backward (ops/bev_pool_v2/bev_pool.py:47-57) + the
 * host-side scheduling of omnihd_amd/plan.py (tiles, XCD runs, patch order).
 *   geometry: either geom (B,N,D,fH,fW,3) fp32, or geom == NULL and rots (B*N,3,3), trans (B*N,3), xs (fW), ys (fH), ds (D)
 *             (device): the frustum point is formed in the kernel with the rounding steps of get_geometry (:235-264);
 *   h_off3 = bx - dx/2, h_dx3, h_nx3: host triples of gen_dx_bx (:80-85); layout_yxz: output row = ((b*Y+y)*X+x)*Z+z
 *             (else ((b*Z+z)*Y+y)*X+x);
 *   walk / wblock (device, n_patch ints each): the static walk of the patch backward over the 16-pixel patches of the
 *             frustum shape and the 4-row band of every walk position (omnihd_amd/pool_plan.py: patch_walk);
 *   outputs (device; capacities from omnihd_pool_plan_sizes, n_total = B*N*D*fH*fW, n_rows = B*X*Y*Z):
 *     pt [n_total], ivl_rel [n_rows], desc32 [tiles_cap*32] (16-byte aligned), row_ptr [n_rows+1]: the tables of
 *     omnihd_bev_pool_v2_fwd_direct_dev; row_bin [n_total], pix_ptr [B*N*fH*fW+1], patch_order [8*patch_per]: the tables of
 *     omnihd_bev_pool_v2_bwd_patch (packed form); hdr int[32] = {points, non-empty rows, tiles, tiles per XCD, longest
 *     patch run, status (0 = ok), ...}; rows_sorted / ranks_depth_sorted [n_total]: the sorted (row, point index) pairs
 *     = ranks_bev (in the chosen numbering) / ranks_depth of the reference, valid for the first hdr[0] entries.
 * omnihd_pool_plan_sizes: out4 = {workspace bytes, tiles_cap, n_patch, patch_per}. */
int omnihd_pool_plan_sizes(long long n_total, int n_rows, int n_pix, int fhw, int tile_items, int long_len, long long* out4);
int omnihd_pool_plan_build(const float* geom, const float* rots, const float* trans, const float* xs, const float* ys,
                           const float* ds, int B, int N, int D, int fH, int fW, const float* h_off3, const float* h_dx3,
                           const int* h_nx3, int layout_yxz, const int* walk, const int* wblock, int tile_items, int long_len,
                           int* pt, int* ivl_rel, int* desc32, int* row_ptr, int* row_bin, int* pix_ptr, int* patch_order,
                           int* hdr, uint32_t* rows_sorted, int* ranks_depth_sorted, void* workspace, size_t workspace_bytes,
                           void* stream);

/* Schedule descriptors for the call above from a tile table (omnihd_csr_tiles) and an optional
 * tile order (8*ceil(n_tiles/8) ints, -1 = idle slot, NULL = tiles in index order).          */
int omnihd_tile_desc(const int* row_ptr, const int* tile_row, const int* tile_order,
                     int n_tiles, int* tile_desc, void* stream);

/* Patch backward used by our own LSS module for C = 64 (same arithmetic as omnihd_bev_pool_v2_bwd; replaces the
 * re-sort + one-thread-per-pixel kernel of ops/bev_pool_v2/bev_pool.py:43-83 / src/bev_pool_cuda.cu:67-121).
 * depth (n_img, d_bins, fhw) and feat (n_img*fhw, 64) of the same frames; out_grad (n_rows, 64).  The backward tables
 * ranks_depth / ranks_row are sorted by pixel (omnihd_sort_ranks by ranks_feat); pix_ptr (n_img*fhw + 1 ints) is the CSR
 * of that order: points [pix_ptr[f], pix_ptr[f+1]) belong to pixel f.  patch_order (n_slots = 8*k ints): entry
 * [x*k + i] = the i-th patch handled on XCD x, patch p = 16 consecutive pixels [16*(p % ppi), ...) of image p / ppi with
 * ppi = ceil(fhw/16); -1 = idle slot; every patch exactly once.  BOTH outputs are written densely (depth_grad zero where
 * no frustum point lies, feat_grad zero for pixels without points): the caller does not clear them.
 * ranks_depth == NULL selects the one-table form: ranks_row[i] = output row | (depth bin << 24) of point i (rows < 2^24 - 1,
 * d_bins <= 127) — one table word per point instead of two.                                                              */
int omnihd_bev_pool_v2_bwd_patch(const float* out_grad, const float* depth, const float* feat,
                                 const int* ranks_depth, const int* ranks_row, const int* pix_ptr,
                                 const int* patch_order, int n_slots, int n_img, int d_bins, int fhw,
                                 long long n_rows, float* depth_grad, float* feat_grad, int c, void* stream);

/* Work partition for the tiled forward: tile_row[0..n_tiles] (capacity n_rows+1 ints) with
 * tile k = rows [tile_row[k], tile_row[k+1]).  A tile closes when rows+points reach a multiple
 * of tile_items; a row with more than long_len points is a tile of its own.  The kernel needs
 * tile_items + long_len <= 1280 for its fast path.  count (device int) = n_tiles; h_count, if
 * non-null, receives it after a stream synchronisation.                                      */
size_t omnihd_csr_tiles_workspace_bytes(int n_rows);
int omnihd_csr_tiles(const int* row_ptr, int n_rows, int tile_items, int long_len,
                     int* tile_row, int* count, int* h_count, void* workspace,
                     size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * bev_pool (v1) — imported by the plugin at load time (projects/mmdet3d_plugin/__init__.py:20)
 * ---------------------------------------------------------------------------------------- */

/* ref: ops/bev_pool/src/bev_pool.cpp:22-47 (bev_pool_forward) -> bev_pool_cuda.cu:20-42.
 * x is [n,c] sorted by rank; geom_feats is [n,4] = (h_idx, w_idx, d_idx, b_idx);
 * out is [b,d,h,w,c]; out rows not named are NOT touched (reference allocates zeros).     */
int omnihd_bev_pool_v1_fwd(const float* x, const int* geom_feats,
                           const int* interval_starts, const int* interval_lengths,
                           float* out, int b, int d, int h, int w, int n, int c,
                           int n_intervals, void* stream);

/* ref: ops/bev_pool/src/bev_pool.cpp:60-87 (bev_pool_backward) -> bev_pool_cuda.cu:61-84.
 * x_grad[(s+i)*c + ch] = out_grad[voxel(s), ch].                                         */
int omnihd_bev_pool_v1_bwd(const float* out_grad, const int* geom_feats,
                           const int* interval_starts, const int* interval_lengths,
                           float* x_grad, int b, int d, int h, int w, int n, int c,
                           int n_intervals, void* stream);

/* ------------------------------------------------------------------------------------------
 * Rank-table preparation (voxel_pooling_prepare_v2 and the backward re-sort)
 * ---------------------------------------------------------------------------------------- */

/* ref: bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:302-346.
 * geom is the (B,N,D,H,W,3) fp32 frustum geometry; n_total = B*N*D*H*W;
 * pts_per_batch = N*D*H*W.  off[a] = bx[a]-dx[a]/2 and dx[a] are fp32 values computed by the
 * host exactly as the reference's tensor arithmetic does.  For point i:
 *   t = (geom[i][a] - off[a]) / dx[a]   (IEEE fp32, no contraction, true division)
 *   coor = trunc(t)  (toward zero — reference defect D3 is kept)
 *   kept = 0 <= coor[a] < nx[a] for a in x,y,z  (NaN is dropped)
 *   keys[i] = b*nz*ny*nx + z*ny*nx + y*nx + x   or  sentinel (= B*nz*ny*nx) when not kept
 *   idx[i]  = i
 * keys/idx have n_total entries.                                                           */
int omnihd_bev_rank_keys(const float* geom, int64_t n_total, int64_t pts_per_batch,
                         const float* h_off3, const float* h_dx3, const int* h_nx3,
                         uint32_t* keys, int* idx, uint32_t sentinel, void* stream);

/* Stable sort of (key, payload...) by key + run-length encoding of the sorted keys.
 * ref: the argsort + RLE blocks at cam_stream_lss_bevpoolv2_depthnet.py:347-359 and
 *      ops/bev_pool_v2/bev_pool.py:47-57 (torch argsort is not guaranteed stable, reference
 *      defect D6; we always produce the stable = canonical order).
 * keys_in[n] (uint32, < 2^key_bits), up to three int payloads (null = absent) are permuted
 * alike into *_out.  Keys equal to `sentinel` (pass 0xFFFFFFFF for "none") sort last and are
 * excluded from the result.  On return (after stream completes):
 *   h_counts[0] = number of non-sentinel elements (n_points)
 *   h_counts[1] = number of runs among them (n_intervals)
 * interval_starts/lengths receive the runs (capacity n each).  counts is a 2-int device
 * buffer; if h_counts is non-null the function synchronises the stream and copies it back. */
size_t omnihd_sort_ranks_workspace_bytes(int64_t n);
int omnihd_sort_ranks(const uint32_t* keys_in, const int* p0_in, const int* p1_in,
                      const int* p2_in, int64_t n, int key_bits, uint32_t sentinel,
                      uint32_t* keys_out, int* p0_out, int* p1_out, int* p2_out,
                      int* interval_starts, int* interval_lengths, int* counts, int* h_counts,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ranks_feat[i] = (ranks_depth[i] / (D*HW)) * HW + ranks_depth[i] % HW
 * ref: cam_stream_lss_bevpoolv2_depthnet.py:321-326 (ranks_feat = pixel index of a frustum
 * point whose flat (B,N,D,H,W) index is ranks_depth).                                      */
int omnihd_ranks_feat_from_depth(const int* ranks_depth, int64_t n, int d, int hw,
                                 int* ranks_feat, void* stream);

/* row_ptr[r] (r in [0,n_rows]) = index of the first point whose sorted key is >= r
 * (lower bound), i.e. CSR offsets over output rows for omnihd_bev_pool_v2_fwd_csr.
 * sorted_keys are the n_points non-sentinel keys produced by omnihd_sort_ranks.            */
int omnihd_csr_from_sorted_keys(const uint32_t* sorted_keys, int n_points, int n_rows,
                                int* row_ptr, void* stream);

/* rows_out[i] = row index of voxel rows_in[i] after moving from the reference (b,z,y,x)
 * order to (b,y,x,z) order — the row numbering of a channels-last s2c tensor
 * (cam_stream_lss_bevpoolv2_depthnet.py:374-376 puts channel = z*C + c).                  */
int omnihd_permute_rows_zyx_to_yxz(const int* rows_in, int64_t n, int nz, int ny, int nx,
                                   int* rows_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Radar point cloud: hard voxelisation and pillar scatter
 * ---------------------------------------------------------------------------------------- */

/* ref: call site bevfusion/detectors/bevf_faster_rcnn_bevdepth.py:97 (self.voxelize), config
 * projects/configs/bevfusion_NewScenes/bevfusion.py:46-50; the op itself is mmdet3d v0.17.1
 * Voxelization/hard_voxelize (un-vendored; semantics restated in oracle/voxelize.c).
 * points [n, f] fp32 (x,y,z first).  Voxels are numbered in first-occurrence point order; the
 * first max_points points of a voxel are kept in point order; voxels beyond max_voxels are
 * refused.  Outputs (capacity max_voxels): voxels [M,max_points,f] zero padded,
 * coors [M,3] = (z,y,x), num_points [M].  voxel_num (device int) = M; if h_voxel_num is
 * non-null the stream is synchronised and M copied back.                                   */
size_t omnihd_voxelize_workspace_bytes(int n_points);
/* The same result in THREE launches for sparse clouds on small grids (round 5: the radar stream, <= 20 k returns on 480x320x1
 * cells, 10 points per pillar): a per-cell atomicMin / list push, a flag pass with per-workgroup counts, and a writer that numbers the voxels in point order from those counts and
 * that keeps each voxel's first max_points points by index.  No sort, no memset, nothing read back; bit-identical outputs for
 * rows < voxel_num (rows beyond it are NOT defined here — the sort path zero-fills them).
 * cell_state: a PERSISTENT device buffer of omnihd_voxelize_grid_state_bytes() bytes (0 = grid too large for this path),
 * initialised once by omnihd_voxelize_grid_state_init and left idle again by every completed call (one call at a time per
 * buffer).  max_points <= 16.  workspace: omnihd_voxelize_grid_workspace_bytes(n) bytes.                                  */
size_t omnihd_voxelize_grid_state_bytes(const float* h_voxel_size3, const float* h_range6);
int omnihd_voxelize_grid_state_init(void* cell_state, size_t state_bytes, void* stream);
size_t omnihd_voxelize_grid_workspace_bytes(int n_points);
int omnihd_voxelize_hard_grid(const float* points, int n_points, int n_feat, const float* h_voxel_size3,
                              const float* h_range6, int max_points, int max_voxels, float* voxels, int* coors,
                              int* num_points, int* voxel_num, void* cell_state, size_t state_bytes, void* workspace,
                              size_t workspace_bytes, void* stream);
int omnihd_voxelize_hard(const float* points, int n_points, int n_feat,
                         const float* h_voxel_size3, const float* h_range6,
                         int max_points, int max_voxels,
                         float* voxels, int* coors, int* num_points, int* voxel_num,
                         int* h_voxel_num, void* workspace, size_t workspace_bytes,
                         void* stream);

/* ref: call site bevf_faster_rcnn_bevdepth.py:101 (pts_middle_encoder), config
 * bevfusion.py:60-61; op = mmdet3d PointPillarsScatter.forward_batch (un-vendored).
 * feats [m,c]; coors [m,4] = (b,z,y,x).  Writes the WHOLE canvas (zeros where no pillar):
 *   channels_last == 0: canvas [batch, c, ny, nx]   (reference layout)
 *   channels_last == 1: canvas [batch, ny, nx, c]
 * cell_map is scratch of batch*ny*nx ints.                                                 */
size_t omnihd_pillar_scatter_workspace_bytes(int batch, int ny, int nx);
int omnihd_pillar_scatter(const float* feats, const int* coors, int m, int c,
                          int batch, int ny, int nx, int channels_last, float* canvas,
                          void* workspace, size_t workspace_bytes, void* stream);

/* The same scatter in two calls around a CALLER-OWNED cell map (batch*ny*nx ints, all -1 = "clean"): omnihd_pillar_cell_map
 * enters the pillars (last pillar of a cell wins), omnihd_pillar_canvas writes the dense canvas and — reset_map != 0 — leaves
 * the map clean again, so a steady-state scatter is two launches without a memset (channels-last canvases with c % 4 == 0: one
 * 16-byte non-temporal store per lane).                                                                                       */
int omnihd_pillar_cell_map(const int* coors, int m, int batch, int ny, int nx, int* cell_map, void* stream);
int omnihd_pillar_canvas(const float* feats, int* cell_map, int c, int batch, int ny, int nx, int channels_last, int reset_map,
                         float* canvas, void* stream);

/* Backward of the scatter: feats_grad[v, ch] = canvas_grad[b, ch, y, x] (gather).          */
int omnihd_pillar_gather(const float* canvas_grad, const int* coors, int m, int c,
                         int batch, int ny, int nx, int channels_last, float* feats_grad,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused pillar feature net (decorate -> Linear -> BatchNorm -> ReLU -> max over the points of a pillar), C = 64 channels
 * ref: PillarFeatureNetV1.forward (projects/mmdet3d_plugin/rcfusion/voxel_encoders/pillar_encoder.py:378-432) +
 *      PFNLayer.forward (rcfusion/voxel_encoders/utils.py:144-181); radar variant RadarPillarFeatureNet.forward
 *      (pillar_encoder.py:91-153) + PFNLayer_Radar.forward (utils.py:229-280: its three Linear / BatchNorm branches are one
 *      Linear with a block-sparse (64 x K) weight and a per-channel BatchNorm).
 * voxels (m, p, f) fp32 zero padded, num_points (m) int32, coors (m, 4) int32 = (b, z, y, x) — the hard voxeliser's output.
 * flags: 1 cluster offset (xyz - mean), 2 pillar-centre offset, 4 distance, 8 legacy (centre offset also REPLACES raw x, y:
 * pillar_encoder.py:410-416), 16 radar offsets (v_x v_y power snr - mean).  Decorated channels K = omnihd_pfn_channels(f, flags)
 * <= 16, in the reference's concatenation order; padded slots are all-zero rows that DO take part in the statistics and in
 * the maximum, as in the reference.  weight (64, K) row-major fp32.
 *   omnihd_pfn_moments  moments (K + K*K DOUBLES): [0..K) = E[x_j], [K + i*K + j] = E[x_i x_j] over the m*p rows of this call (two launches,
 *                       fixed-order reduction); a multi-GPU caller averages them over the ranks (naiveSyncBN: mean of rank means);
 *   omnihd_pfn_consts   consts[0..64) mean, [64..128) 1/sqrt(var + eps), [128..192) scale, [192..256) shift per channel from
 *                       the moments (use_running == 0; running statistics updated when given: `unbiased` selects torch's
 *                       n/(n-1) variance for them) or from the running statistics (use_running != 0: inference);
 *   omnihd_pfn_apply    out (m, 64) = max over slots of relu(scale * (W x) + shift);
 *   omnihd_pfn_bwd_sums sums = [A (64) | B (64) | G (64 x K)]: A = sum g, B = sum g * yhat, G = sum g * x over the arg-max slot
 *                       (first maximum) of every (pillar, channel) with a positive output, g = grad_out (m, 64);
 *   omnihd_pfn_bwd_final dweight (64, K), dgamma, dbeta from this rank's sums, [A | B] summed over all ranks (= sums on one
 *                       rank), this rank's moments and the constants of the forward; n_ranks <= 0: the constants came from the
 *                       running statistics (inference-mode BatchNorm inside a differentiated graph) — the batch-statistics
 *                       terms vanish and `moments` is not read beyond being non-NULL.  No gradient flows to the points.      */
int omnihd_pfn_channels(int f, int flags);
size_t omnihd_pfn_workspace_bytes(int m, int p, int k);
int omnihd_pfn_moments(const float* voxels, const int* num_points, const int* coors, int m, int p, int f, float vx, float vy,
                       float x_off, float y_off, int flags, double* moments, void* workspace, size_t workspace_bytes, void* stream);
int omnihd_pfn_consts(const float* weight, const float* gamma, const float* beta, const double* moments, int k, long long n_rows,
                      float eps, float momentum, int unbiased, int use_running, float* running_mean, float* running_var,
                      float* consts, void* stream);
int omnihd_pfn_apply(const float* voxels, const int* num_points, const int* coors, int m, int p, int f, float vx, float vy,
                     float x_off, float y_off, int flags, const float* weight, const float* consts, float* out, void* stream);
int omnihd_pfn_bwd_sums(const float* voxels, const int* num_points, const int* coors, int m, int p, int f, float vx, float vy,
                        float x_off, float y_off, int flags, const float* weight, const float* consts, const float* grad_out,
                        float* sums, void* workspace, size_t workspace_bytes, void* stream);
int omnihd_pfn_bwd_final(const float* sums, const float* ab_all_ranks, const double* moments, const float* weight,
                         const float* gamma, const float* consts, int k, long long n_rows, int n_ranks, float* dweight,
                         float* dgamma, float* dbeta, void* stream);

/* Bilinear sampling ("deformable im2col") of the 3x3 deformable convolution of DepthNet, deform_groups = 1.
 * ref: build_conv_layer(dict(type='DCN', ...)) at cam_stream_lss_bevpoolv2_depthnet.py:587-595 (mmcv
 * DeformConv2dPack, un-vendored).  x [batch,h,w,c] bf16 channels-last; offset [batch,ho,wo,18] fp32 with
 * channel = tap*2 + (0: dy, 1: dx); col / gcol [batch*ho*wo, 9, c] bf16; c in {32,64,128,256}.
 * fwd:  col[p][t][:] = zero-padded bilinear sample of x at (yo*stride - pad + ky*dil + dy, ...).
 * bwd:  gx (same layout as x, may be NULL) and goffset (same layout as offset, may be NULL) from gcol;
 *       max_abs_offset_ceil is a DEVICE int >= ceil(max |offset|) (bounds the gather window of gx;
 *       no atomics, deterministic); stride must be 1 for gx.                                        */
int omnihd_dcn3x3_sample_fwd(const void* x_nhwc_bf16, const float* offset_nhwc, void* col_bf16, int batch,
                             int h, int w, int c, int stride, int pad, int dil, void* stream);
int omnihd_dcn3x3_sample_bwd(const void* x_nhwc_bf16, const float* offset_nhwc, const void* gcol_bf16,
                             const int* max_abs_offset_ceil, void* gx_nhwc_bf16, float* goffset_nhwc,
                             int batch, int h, int w, int c, int stride, int pad, int dil, void* stream);
/* The same with fp32 x / col / gcol / gx (the reference's arithmetic). */
int omnihd_dcn3x3_sample_fwd_f32(const float* x_nhwc, const float* offset_nhwc, float* col, int batch, int h, int w,
                                 int c, int stride, int pad, int dil, void* stream);
int omnihd_dcn3x3_sample_bwd_f32(const float* x_nhwc, const float* offset_nhwc, const float* gcol,
                                 const int* max_abs_offset_ceil, float* gx_nhwc, float* goffset_nhwc, int batch, int h,
                                 int w, int c, int stride, int pad, int dil, void* stream);

/* ------------------------------------------------------------------------------------------
 * Test-time post-process: rotated BEV NMS
 * ---------------------------------------------------------------------------------------- */

/* Replaces mmdet3d v0.17.1 `iou3d_cuda.nms_gpu(boxes, keep, thresh, device_id)`
 * (mmdet3d/ops/iou3d/src/iou3d.cpp + iou3d_kernel.cu, un-vendored), reached from the reference via
 * Anchor3DHead.get_bboxes -> box3d_multiclass_nms with test_cfg
 * projects/configs/bevfusion_NewScenes/bevfusion.py:147-155.
 *   boxes    [n,5] f32 (x1, y1, x2, y2, angle), ALREADY in descending score order (the Python wrapper
 *            sorts, as upstream's does);  n <= 4096
 *   keep     [n] i64: positions (into the sorted order) of the surviving boxes, ascending
 *   num_out  [1] i32 on the device: number of survivors (upstream returns it to the host; here the
 *            mask reduction also runs on the device, the caller reads the count when it needs it)
 *   workspace >= omnihd_nms_rotated_workspace_bytes(n)                                              */
size_t omnihd_nms_rotated_workspace_bytes(int n);
int omnihd_nms_rotated(const float* boxes, int n, float thresh, long long* keep, int* num_out,
                       void* workspace, size_t workspace_bytes, void* stream);

/* out[i*nb+j] = rotated BEV IoU(boxes_a[i], boxes_b[j]) with the same arithmetic as the NMS
 * (upstream `boxes_iou_bev_gpu`).                                                                   */
int omnihd_iou_bev_matrix(const float* boxes_a, int na, const float* boxes_b, int nb, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dense stride-1 "same" convolutions on the matrix cores: forward and data gradient (implicit GEMM, NHWC)
 * ref: the BEV encoder  bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:201-214 (nn.Conv2d 3x3, padding 1,
 *      bias=False, 1024->1024->512->512->256 at 160x240), the fusion conv bevf_faster_rcnn_bevdepth.py:61-72 (640->384),
 *      and every other 1x1 / 3x3 stride-1 convolution with padding = dilation * (k / 2) — torch.nn.functional.conv2d.
 * ---------------------------------------------------------------------------------------- */

/* 1 when omnihd_conv_fwd_bf16 takes the geometry: k in {1, 3}, Cin % 64 == 0, Cout % 8 == 0.                   */
int omnihd_conv_fwd_supported(int batch, int h, int w, int cin, int cout, int ksize, int dil);
/* y[b,y,x,n] = bias[n] + sum_{ky,kx,c} x[b, y+(ky-k/2)*dil, x+(kx-k/2)*dil, c] * w[n,ky,kx,c]   (zero outside the image)
 *   x_nhwc (batch,h,w,cin) bf16, w_ohwi (cout,k,k,cin) bf16 (= a torch weight in channels_last memory format),
 *   bias (cout) f32 or NULL, y_nhwc (batch,h,w,cout) bf16.  fp32 accumulation, one rounding to bf16.
 *   tile: 0 = choose, 128 = 128x128 tile / 4 wavefronts, 256 = 256x128 tile / 8 wavefronts, 254 = 128x256 tile,
 *         300 = 3x3 row-shift kernel (256x128 tile, dilation <= 8).                                                 */
int omnihd_conv_fwd_bf16(const void* x_nhwc, const void* w_ohwi, const float* bias, void* y_nhwc, int batch, int h,
                         int w, int cin, int cout, int ksize, int dil, int tile, void* stream);
/* Weight images of many convolution layers in one launch (fp32 master weights -> what the kernels above read): `table` is a
 * DEVICE array of n_entries records
 *   { const float* src; long long so, si, sy, sx;            fp32 weight and its element strides (cout, cin, ky, kx)
 *     uint16* f_hi, *f_lo;                                    (Cout,k,k,Cin) bf16 image: hi plane (= the plain bf16 rounding) and,
 *                                                             unless NULL, lo = bf16(w - hi) for the split kernels
 *     uint16* d_hi, *d_lo;                                    (Cin,k,k,Cout) images with mirrored taps for the data gradient, or NULL
 *     int cout, cin, k, first_block; }                        first_block = sum over the earlier records of ceil(cout/32)*ceil(cin/32)
 * in increasing first_block order; total_blocks = that sum over all records.  Replaces, per layer and step, a layout copy +
 * omnihd_split_f32 + two omnihd_conv_dgrad_weights launches.                                                              */
int omnihd_weight_images(const void* table, int n_entries, int total_blocks, void* stream);
/* The same for records whose weights lie in channels_last memory (si == 1; what the training step holds): one workgroup per
 * 64 x 64 (cout, cin) tile of ONE tap — 256-byte reads, 128-byte writes.  first_block = sum over the earlier records of
 * ceil(cout/64)*ceil(cin/64)*k*k.                                                                                           */
int omnihd_weight_images_cl(const void* table, int n_entries, int total_blocks, void* stream);

/* wt[c,k-1-ky,k-1-kx,n] = w[n,ky,kx,c]: the weights with which the DATA GRADIENT of the convolution above is the same
 * convolution applied to the output gradient:  omnihd_conv_fwd_bf16(gout, wt, NULL, gx, batch, h, w, cout, cin, ...).  */
int omnihd_conv_dgrad_weights(const void* w_ohwi, void* wt_ihwo, int cout, int cin, int ksize, void* stream);

/* The same convolutions at fp32-grade accuracy on the bf16 matrix cores (the reference trains in fp32:
 * projects/configs/bevfusion_NewScenes/bevfusion.py:223-268 has no fp16 hook; gfx950's fp32 MFMA runs at 1/16 of the bf16
 * rate and has no TF32 form).  Every fp32 operand is split into two bf16 planes, hi = bf16(v), lo = bf16(v - hi), and
 *     x * w ~= x_hi*w_hi + x_hi*w_lo + x_lo*w_hi   with fp32 accumulation
 * (the dropped lo*lo term is 2^-16 of a product: results agree with an fp32 convolution to ~1e-5 relative).               */

/* hi[i] = bf16(x[i]) (round to nearest even), lo[i] = bf16(x[i] - hi[i]) for n fp32 values (any layout: element-wise);
 * inf / nan stay in the hi plane.  16-byte aligned buffers.                                                               */
int omnihd_split_f32(const float* x, long long n, void* hi, void* lo, void* stream);
/* omnihd_conv_fwd_bf16 on split operands: x_hi/x_lo (batch,h,w,cin) bf16, w_hi/w_lo (cout,k,k,cin) bf16, bias f32 or NULL,
 * y (batch,h,w,cout) F32.  Same geometries (omnihd_conv_fwd_supported) and tile codes.  The data gradient is the same call
 * on the split output gradient with both weight planes re-laid by omnihd_conv_dgrad_weights.                              */
int omnihd_conv_fwd_split(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                          float* y_nhwc, int batch, int h, int w, int cin, int cout, int ksize, int dil, int tile,
                          void* stream);

/* ---- TF32-grade form of the dense convolutions (round 6) ----------------------------------------------------------------------
 * The reference trains with TF32 left on (tools/train.py:150-153): its cuDNN convolutions round both operands to 11 significant
 * bits.  An IEEE half has the same 11 bits, and gfx950's half MFMA runs at the bf16 rate: ONE matrix product per fp32 product
 * instead of the three of the fp32-grade split form.  Half has 5 exponent bits, so a GRADIENT tensor is converted with a
 * power-of-two scale (exact) that brings its largest magnitude just below 2^15, and the consuming kernel multiplies by the inverse.
 *   omnihd_cast_f16     out16[i] = half(x[i] * s) (round to nearest even).  scaled == 0: s = 1.  scaled != 0: s = 2^(15 - e) with
 *                       max|x| in [2^(e-1), 2^e) found by a first pass; scratch2 (2 device words) receives the working maximum in
 *                       [0] and 1 / s in [1] (what the consumers take as `alpha`).  scaled == 1: [0] is zeroed by the call (a
 *                       memset node); scaled == 2: the caller passes it zeroed (one memset for many calls); scaled == 3: [0] already
 *                       holds max|x| (accumulated by x's producer: omnihd_*_bwd_f32_amax) — no first pass.  No synchronisation.
 *   omnihd_conv_fwd_f16 omnihd_conv_fwd_bf16's geometries and tile codes on half operands: x16 (batch,h,w,cin), w16 (cout,k,k,cin)
 *                       -> y (batch,h,w,cout) F32 = alpha * conv + bias (alpha: device scalar or NULL = 1).  The data gradient is
 *                       the same call on the scaled half output gradient with the mirrored weight image and alpha = its 1 / s.
 *   omnihd_conv_wgrad_nhwc_f16  omnihd_conv_wgrad_nhwc on half operands, dw = alpha * sum (same split-K slabs, fixed order).
 *   omnihd_weight_images / _cl  an entry with k < 0 asks for HALF images of a |k| x |k| kernel (f_hi / d_hi; the lo pointers unused).
 * Parity contract: tests/test_conv_f16_gpu.py (per kernel: exact against an fp32 convolution of the half-rounded operands up to
 * fp32 summation order; 2^-10-grade against the fp32 convolution).  ref: the cuDNN TF32 convolutions behind every nn.Conv2d of
 * bevfusion.py:62-123 and cam_stream_lss_bevpoolv2_depthnet.py:201-214. */
int omnihd_cast_f16(const float* x, long long n, int scaled, void* out16, float* scratch2, void* stream);
int omnihd_conv_fwd_f16(const void* x16, const void* w16, const float* bias, float* y_nhwc, const float* alpha, int batch, int h,
                        int w, int cin, int cout, int ksize, int dil, int tile, void* stream);
int omnihd_conv_wgrad_nhwc_f16(const void* x16, const void* g16, float* dw, const float* alpha, int batch, int h, int w, int cin,
                               int ho, int wo, int cout, int ksize, int stride, int pad, int dil, void* workspace,
                               size_t workspace_bytes, void* stream);

/* Weight gradient straight from the NHWC operands (round 5, csrc/conv_wgrad_nhwc.hip): no pixel-major staging pass — the
 * [pixel][channel] tiles are read transposed from LDS (ds_read_b64_tr_b16); a workgroup computes a 128x128 (Cout, Cin) tile for
 * one tap — or, for 3x3 / stride 1 / pad 1 / dilation 1, for the three taps of a kernel row over a padded raster — with split-K
 * over the pixels into fp32 slabs summed in a fixed order (deterministic).  The launch is persistent (one or two workgroups per
 * CU walk the tile list).  Any stride / padding / dilation, square kernels up to 4x4, channel counts multiples of 8, operands
 * below 2 GiB.  x_lo / g_lo NULL: bf16 operands; both non-NULL: the fp32-grade split form.  dw (cout,k,k,cin) F32.  Two launches
 * (one when the pixels are not split); the workspace holds the slabs (up to 256 MB: 1024->1024 3x3 takes 4 x 38 MB).  Replaces
 * the weight-gradient half of cuDNN's convolution backward for every Conv2d of the detector it applies to: ResNet-50, FPN / FPNC,
 * DepthNet, the BEV encoder (cam_stream_lss_bevpoolv2_depthnet.py:201-214), SECOND / SECONDFPN, the anchor head — reference
 * layers bevfusion.py:62-85,96-123.                                                                                             */
size_t omnihd_conv_wgrad_nhwc_workspace_bytes(int batch, int h, int w, int cin, int ho, int wo, int cout, int ksize, int stride,
                                              int pad, int dil);
int omnihd_conv_wgrad_nhwc(const void* x_hi, const void* x_lo, const void* g_hi, const void* g_lo, float* dw, int batch, int h,
                           int w, int cin, int ho, int wo, int cout, int ksize, int stride, int pad, int dil, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Strided and transposed convolutions on the matrix cores (round 5): the general form of the implicit GEMM above
 * ref: the stage-entry convolutions of SECOND (3x3, stride 2: projects/configs/bevfusion_NewScenes/bevfusion.py:62-68,
 *      layer_strides=[2,2,2]), the strided 3x3 / 1x1 layers of ResNet-50 (bevfusion.py:77-85), SECONDFPN's transposed
 *      convolutions with kernel == stride (bevfusion.py:69-74) — torch.nn.functional.conv2d / conv_transpose2d and their
 *      data gradients (torch.ops.aten.convolution_backward).  No atomics: results are run-to-run identical.
 * ---------------------------------------------------------------------------------------- */

/* 1 when omnihd_conv_gen takes the pass: mode 0 / 1, square kernel with k*k <= 16, stride*stride <= 16, (ho, wo) the
 * convolution's output size for (h, w, k, stride, pad, dil), SOURCE channels (mode 0: cin, mode 1: cout) a multiple of 8,
 * operands below 1 GiB per plane.                                                                                        */
int omnihd_conv_gen_supported(int mode, int batch, int h, int w, int cin, int ho, int wo, int cout, int ksize, int stride,
                              int pad, int dil);
/* The convolution  y = conv2d(x (batch,h,w,cin), w, stride, pad, dil) -> (batch,ho,wo,cout), all tensors NHWC:
 *   mode 0, forward:        src = x,    w = (cout,k,k,cin) image,                     dst = y  (+ bias (cout) f32 or NULL)
 *   mode 1, data gradient:  src = gout (batch,ho,wo,cout), w = (cin,k,k,cout) image with MIRRORED taps
 *                           (omnihd_conv_dgrad_weights / the d images of omnihd_weight_images),  dst = gx (batch,h,w,cin):
 *                           every input pixel is written exactly once (zeros where no tap reaches it), no atomics.
 * A transposed convolution with kernel == stride (weight (Cin_t,Cout_t,k,k)) is mode 1 of the stride-k convolution whose
 * weight is that tensor read as (cout = Cin_t, cin = Cout_t); its data gradient is mode 0 of the same convolution.
 * src_lo / w_lo NULL: bf16 operands, dst bf16, fp32 accumulation.  Both non-NULL: the fp32-grade split form
 * (hi*hi + hi*lo + lo*hi), dst F32.  128x128 tiles, one launch for all stride classes.                                   */
int omnihd_conv_gen(int mode, const void* src_hi, const void* src_lo, const void* w_hi, const void* w_lo, const float* bias,
                    void* dst, int batch, int h, int w, int cin, int ho, int wo, int cout, int ksize, int stride, int pad,
                    int dil, void* stream);

/* ------------------------------------------------------------------------------------------
 * Anchor target assignment + the three detection losses of Anchor3DHead, fused (round 5)
 * ref: Anchor3DHead.loss / loss_single as the reference vendors it (bevfusion/dense_heads/det_anchor3d_head.py:192-372),
 *      config projects/configs/bevfusion_NewScenes/bevfusion.py:96-155: MaxIoUAssigner over BboxOverlapsNearest3D (pos 0.6 /
 *      neg 0.3 / min_pos 0.3, every anchor reaching a box's best IoU matched, later boxes win), DeltaXYZWLHRBBoxCoder targets,
 *      sigmoid FocalLoss, SmoothL1Loss with the sine-difference yaw encoding and code weights, 2-way CrossEntropyLoss of the
 *      direction bin; avg_factor = sum over samples of max(positives, 1).  One feature level.
 * anchors (A, code_size) f32 in (y, x, anchor-per-location) order, A = h*w*anchors_per_loc; gt_boxes (total_gt, code_size) f32
 * and gt_labels (total_gt) i32 concatenated over the batch, gt_offsets (batch+1) i32 (at most 128 boxes per sample);
 * cls_score (batch, anchors_per_loc*num_classes, h, w), bbox_pred (batch, anchors_per_loc*code_size, h, w), dir_pred
 * (batch, anchors_per_loc*2, h, w) f32 with the 12 element strides (b, c, y, x per map) in h_strides12.
 * Writes the UNSCALED gradient maps g_cls / g_box / g_dir (same strides as the inputs) and out[0..2] = the three losses times
 * h_loss_weights3, out[3] = avg_factor, out[4 + b] = positives of sample b.  Three launches (+ one memset); the backward
 * (omnihd_anchor_loss_bwd) scales the three gradient maps (dense memory of n_* floats) by upstream * weight / avg_factor in one.
 * No atomic sums: run-to-run identical.                                                                                   */
size_t omnihd_anchor_loss_workspace_bytes(int batch, int anchors_per_sample, int total_gt);
int omnihd_anchor_loss_fwd(const float* anchors, const float* gt_boxes, const int* gt_labels, const int* gt_offsets,
                           int total_gt, const float* cls_score, const float* bbox_pred, const float* dir_pred, int batch,
                           int h, int w, int anchors_per_loc, int num_classes, int code_size, const long long* h_strides12,
                           const float* h_params7, int sin_diff, const float* h_code_weight, const float* h_loss_weights3,
                           float* g_cls, float* g_box, float* g_dir, float* out, void* workspace, size_t workspace_bytes,
                           void* stream);
int omnihd_anchor_loss_bwd(float* g_cls, long long n_cls, float* g_box, long long n_box, float* g_dir, long long n_dir,
                           const float* up_cls, const float* up_box, const float* up_dir, const float* fin,
                           const float* h_loss_weights3, void* stream);

/* ------------------------------------------------------------------------------------------
 * Depth-head epilogue of the LSS camera stream: softmax over D + depth / context split + pooling layouts
 * ref: CamEncode.get_depth_dist / get_depth_feat  bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:134-143
 *      (x[:, :D].softmax(dim=1), x[:, D:D+C]), the layout copy in front of the pooling  :290
 *      (feat.permute(0,1,3,4,2).contiguous()) and the fp32 casts of ops/bev_pool_v2/bev_pool.py:20-21.
 * ---------------------------------------------------------------------------------------- */

/* logits  [n_img*fhw rows, d_bins] with row pitch ld_logits ELEMENTS (channels-last depth logits; the pitch lets a channel
 *         slice of a wider tensor be passed), context [rows, c] with pitch ld_context, both bf16 (is_f32 = 0) or fp32.
 * depth      [n_img, d_bins, fhw] f32 = softmax over d_bins per pixel  (the (B,N,D,fH,fW) tensor bev_pool_v2 gathers from)
 * depth_rows [rows, d_bins]       f32, the same values pixel-major (what the KL depth loss reads), or NULL
 * feat       [rows, c]            f32 = the context rows               (the (B,N,fH,fW,C) tensor bev_pool_v2 gathers from);
 *            context == NULL skips it (an fp32, packed context tensor IS feat already).
 * fp32 arithmetic: exp(x - max) / sum.  c % 4 == 0, d_bins <= 160, n_img <= 65535.                                        */
int omnihd_depth_head_fwd(const void* logits, long long ld_logits, const void* context, long long ld_context, int is_f32,
                          int n_img, int fhw, int d_bins, int c, float* depth, float* depth_rows, float* feat,
                          void* stream);
/* Backward: g = g_depth [n_img, d_bins, fhw] (+ g_rows [rows, d_bins]), either may be NULL;
 * g_logits[p][d] = y * (g - sum_d g * y) with y = depth (the forward output), written with row pitch ld_g_logits in the
 * logits' element type; g_context = g_feat cast to that type (pitch ld_g_context), skipped when g_context == NULL.       */
int omnihd_depth_head_bwd(const float* depth, const float* g_depth, const float* g_rows, const float* g_feat, int is_f32,
                          int n_img, int fhw, int d_bins, int c, void* g_logits, long long ld_g_logits, void* g_context,
                          long long ld_g_context, void* stream);

/* ------------------------------------------------------------------------------------------
 * Frozen-BatchNorm epilogue of a convolution (image backbone, bevfusion.py:76-85)
 * ---------------------------------------------------------------------------------------- */

/* y = act(x * scale[c] + shift[c] (+ res)) on channels-last bf16 rows: x, res, y [n_rows, c] bf16
 * (res may be NULL), scale/shift [c] f32 (= gamma/sqrt(var+eps), beta - mean*scale of a BatchNorm in
 * eval mode, mmcv/torch `F.batch_norm(training=False)` followed by `+ identity` and `ReLU`);
 * relu != 0 applies max(., 0).  c % 8 == 0.                                                        */
int omnihd_affine_act_fwd(const void* x, const float* scale, const float* shift, const void* res, void* y,
                          long long n_rows, int c, int relu, void* stream);
/* Backward of the above for constant scale/shift: gres = gy * [y > 0] (all ones when relu == 0),
 * gx = gres * scale[c].  y is the forward output (only read when relu != 0); gres may be NULL.       */
int omnihd_affine_act_bwd(const void* gy, const void* y, const float* scale, void* gx, void* gres,
                          long long n_rows, int c, int relu, void* stream);
/* The same two passes on fp32 rows (the reference-precision step: the reference trains in fp32).       */
int omnihd_affine_act_fwd_f32(const float* x, const float* scale, const float* shift, const float* res, float* y,
                              long long n_rows, int c, int relu, void* stream);
/* The same pass, y additionally written as its two bf16 planes (hi = bf16(y), lo = bf16(y - hi): what omnihd_split_f32 produces)
 * for the fp32-grade convolution that reads y next — saves that split pass over y.                                          */
int omnihd_affine_act_fwd_f32_planes(const float* x, const float* scale, const float* shift, const float* res, float* y,
                                     void* y_hi, void* y_lo, long long n_rows, int c, int relu, void* stream);
int omnihd_affine_act_bwd_f32(const float* gy, const float* y, const float* scale, float* gx, float* gres,
                              long long n_rows, int c, int relu, void* stream);
/* Round 6, for the TF32-grade convolutions next to these layers (omnihd_conv_fwd_f16): omnihd_affine_act_fwd_f32_planes with
 * y_lo == NULL writes ONE plane to y_hi — the IEEE half of y (omnihd_cast_f16's unscaled output: saves that cast pass);
 * omnihd_affine_act_bwd_f32_amax additionally accumulates max |gx| into *amax_bits (bit pattern of a float >= 0, zeroed by the caller;
 * one atomic per workgroup), so that the cast of gx needs no pass of its own for the scale (omnihd_cast_f16 with scaled == 3). */
int omnihd_affine_act_bwd_f32_amax(const float* gy, const float* y, const float* scale, float* gx, float* gres, void* amax_bits,
                                   long long n_rows, int c, int relu, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training-mode BatchNorm (+ReLU) with the statistics exchange left to the caller ("naive" SyncBN:
 * projects/mmdet3d_plugin/ops/norm.py:28-82 and mmdet3d's 1-D/2-D variants; plain BatchNorm on one rank)
 * All activations are channels-last rows [rows, c], bf16 — or fp32 in the *_f32 forms, whose argument meaning is the
 * same; c % 8 == 0, c <= 2048.  Why fp32 rows do not go to torch: torch's native channels-last BatchNorm kernels combine
 * their per-block partial sums inside ONE launch behind a semaphore without an acquire on the reading side, and on this
 * 8-XCD part (per-XCD L2s, not coherent inside a launch) that returned wrong input gradients in some processes
 * (tests/test_lss_plain_gpu.py, round 1's red suite).  Here every reduction crosses a kernel boundary.
 * ---------------------------------------------------------------------------------------- */

size_t omnihd_bn_workspace_bytes(long long rows, int c);
/* mode 0: sums[0:c] = mult * sum_rows a,  sums[c:2c] = mult * sum_rows a^2           (forward statistics)
 * mode 1: g' = a * [mask > 0]; sums[0:c] = mult * sum g', sums[c:2c] = mult * sum g' * b.  With mask NULL the ReLU
 *         mask is recomputed as [b * scale + shift > 0] from fwd_scale_shift [2c] (the forward's constants: no
 *         saved output to read), or everything passes when that is NULL too.
 * Two-stage reduction in a fixed order (deterministic).                                              */
int omnihd_bn_channel_sums(const void* a, const void* b, const void* mask, const float* fwd_scale_shift, float* sums,
                           long long rows, int c, int mode, float mult, void* workspace, size_t workspace_bytes,
                           void* stream);
/* stats = (mean, mean of squares) [2c], possibly summed over ranks: multiplied by rank_mult (1/R) here.
 * Writes scale = gamma * invstd, shift = beta - mean * scale (feed omnihd_affine_act_fwd), mean, invstd,
 * and updates running_mean / running_var (both NULL to skip) with `momentum`;
 * running_var takes var * var_correction (n/(n-1) for torch BatchNorm, 1 for the reference's SyncBN).   */
int omnihd_bn_fwd_consts(const float* stats, float rank_mult, const float* gamma, const float* beta, float eps,
                         float momentum, float var_correction, int c, float* running_mean, float* running_var,
                         float* scale, float* shift, float* mean, float* invstd, void* stream);
/* local_sums / global_sums: mode-1 sums of this rank / summed over ranks (the same array on one rank).
 * dgamma, dbeta from the local sums; coef_a/b/c such that gx = g' * a[c] + x * b[c] + c[c];
 * inv_count = 1 / (ranks * rows of THIS rank).                                                          */
int omnihd_bn_bwd_consts(const float* local_sums, const float* global_sums, const float* gamma, const float* mean,
                         const float* invstd, float inv_count, int c, float* dgamma, float* dbeta, float* coef_a,
                         float* coef_b, float* coef_c, void* stream);
/* gx = g' * a + x * b + c;  gres (may be NULL) = g', the gradient of a residual added before the ReLU.   */
int omnihd_bn_bwd_apply(const void* gy, const void* y_mask, const float* fwd_scale_shift, const void* x,
                        const float* coef_a, const float* coef_b, const float* coef_c, void* gx, void* gres,
                        long long rows, int c, void* stream);

int omnihd_bn_channel_sums_f32(const float* a, const float* b, const float* mask, const float* fwd_scale_shift,
                               float* sums, long long rows, int c, int mode, float mult, void* workspace,
                               size_t workspace_bytes, void* stream);
int omnihd_bn_bwd_apply_f32(const float* gy, const float* y_mask, const float* fwd_scale_shift, const float* x,
                            const float* coef_a, const float* coef_b, const float* coef_c, float* gx, float* gres,
                            long long rows, int c, void* stream);

/* One-call single-rank forms of the above (no statistics exchange): see csrc/batch_norm.hip.             */
int omnihd_bn_train_fwd(const void* x, const void* res, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, float momentum, float eps, float var_correction, int relu, void* y,
                        float* stats2c, float* consts4c, long long rows, int c, void* workspace,
                        size_t workspace_bytes, void* stream);
int omnihd_bn_train_bwd(const void* gy, const void* y_mask, int relu_from_x, const void* x, const float* gamma,
                        const float* consts4c, void* gx, void* gres, float* sums2c, float* out5c, long long rows, int c,
                        void* workspace, size_t workspace_bytes, void* stream);
int omnihd_bn_train_fwd_f32(const float* x, const float* res, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, float momentum, float eps, float var_correction, int relu, float* y,
                            float* stats2c, float* consts4c, long long rows, int c, void* workspace,
                            size_t workspace_bytes, void* stream);
int omnihd_bn_train_bwd_f32(const float* gy, const float* y_mask, int relu_from_x, const float* x, const float* gamma,
                            const float* consts4c, float* gx, float* gres, float* sums2c, float* out5c, long long rows,
                            int c, void* workspace, size_t workspace_bytes, void* stream);
/* One-call forms that also write the bf16 planes of their output (y resp. gx) for the adjacent fp32-grade convolution.
 * omnihd_bn_train_bwd_f32_planes with gx == NULL writes the planes ONLY (round 5: the convolution in front of the layer is the
 * sole consumer of that gradient and reads planes anyway — no fp32 copy, no split pass).                                  */
int omnihd_bn_train_fwd_f32_planes(const float* x, const float* res, const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, float momentum, float eps, float var_correction, int relu, float* y,
                                   void* y_hi, void* y_lo, float* stats2c, float* consts4c, long long rows, int c, void* workspace,
                                   size_t workspace_bytes, void* stream);
int omnihd_bn_train_bwd_f32_planes(const float* gy, const float* y_mask, int relu_from_x, const float* x, const float* gamma,
                                   const float* consts4c, float* gx, void* gx_hi, void* gx_lo, float* gres, float* sums2c,
                                   float* out5c, long long rows, int c, void* workspace, size_t workspace_bytes, void* stream);
/* The TF32-grade neighbours (round 6): omnihd_bn_train_fwd_f32_planes with y_lo == NULL writes the IEEE-half plane of y to y_hi;
 * omnihd_bn_train_bwd_f32_amax = omnihd_bn_train_bwd_f32 that also accumulates max |gx| into *amax_bits (see the affine form). */
int omnihd_bn_train_bwd_f32_amax(const float* gy, const float* y_mask, int relu_from_x, const float* x, const float* gamma,
                                 const float* consts4c, float* gx, void* amax_bits, float* gres, float* sums2c, float* out5c,
                                 long long rows, int c, void* workspace, size_t workspace_bytes, void* stream);

/* Column sums of a row-major [rows][c] bf16 (is_f32 = 0) or fp32 matrix for ANY c, in fp32, two stages in a fixed order:
 * the bias gradient of a convolution whose channel count is not a multiple of 8 (sum over N, H, W of the NHWC output
 * gradient; torch's reduction takes 0.36 ms for DepthNet's 59 depth logits at 6 x 64 x 176).                          */
size_t omnihd_column_sums_workspace_bytes(long long rows, int c);
int omnihd_column_sums(const void* a, int is_f32, long long rows, int c, float* sums, void* workspace,
                       size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Radar input format: sweep merge + ego-motion compensation on the device
 * ---------------------------------------------------------------------------------------- */

/* ref: LoadRadarPointsMultiSweeps.__call__, projects/mmdet3d_plugin/datasets/pipelines/loading.py:229-309.
 *   raw            [n, load_dim] f32: the returns of all used sweeps, concatenated in the reference's order
 *                  (radar by radar, newest sweep first); columns x, y, z, v_r, power, motion_state, SNR, valid
 *   sweep_offsets  [n_sweeps+1] i32: first row of each sweep in `raw`
 *   sweep_consts   [n_sweeps, 17] f64 per sweep: ego velocity in the SENSOR frame (3), sensor2lidar rotation
 *                  (9, row major), sensor2lidar translation (3), time lag dt (1), radar id (1)
 *   out10          [n, 10] f32: x, y, z, vx_comp, vy_comp, power, snr, dt, Vr_comp, radar_id (LiDAR frame)
 *   in_range       [n] u8 (may be NULL): strict range test against pc_range6 (device f32[6]), as RadarPoints.in_range_3d
 * Positions are bit-identical to the host loader; velocity columns to float32 trigonometry accuracy.   */
int omnihd_radar_merge(const float* raw, int n, int load_dim, const int* sweep_offsets, int n_sweeps,
                       const double* sweep_consts, const float* pc_range6, float* out10, unsigned char* in_range,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OMNIHD_HIP_H_ */
