/*
 * oracle/voxelize_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Sequential restatement of the radar-side ops the reference takes from mmdet3d v0.17.1
 * (pinned in /root/reference/README.md:153-156; the package is NOT vendored in the reference):
 *   - Voxelization / hard_voxelize   (call site bevfusion/detectors/bevf_faster_rcnn_bevdepth.py:97,
 *                                     config projects/configs/bevfusion_NewScenes/bevfusion.py:46-50)
 *   - PointPillarsScatter            (call site bevf_faster_rcnn_bevdepth.py:101, config :60-61)
 *
 * PARITY UNPINNED for these two: the algorithm below is the published one-pass CPU algorithm of
 * that release (a dense coor_to_voxelidx map, voxels numbered at first occurrence, max_points and
 * max_voxels caps with `continue`), restated from its documented behaviour; the reference repo
 * holds no test, fixture or golden vector for it, so it is anchored only on the call sites above
 * and on hand-computed cases in tests/test_oracle.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* points [n,f]; voxels [max_voxels,max_points,f] (caller zero-fills); coors [max_voxels,3]=(z,y,x);
 * num_points_per_voxel [max_voxels] (caller zero-fills).  Returns voxel_num. */
int oracle_hard_voxelize(const float* points, int n, int f, const float* voxel_size,
                         const float* coors_range, int max_points, int max_voxels, float* voxels,
                         int* coors, int* num_points_per_voxel) {
  int grid[3];
  for (int j = 0; j < 3; ++j)
    grid[j] = (int)lroundf((coors_range[j + 3] - coors_range[j]) / voxel_size[j]);
  const size_t cells = (size_t)grid[0] * grid[1] * grid[2];
  int* coor_to_voxelidx = (int*)malloc(cells * sizeof(int));
  for (size_t i = 0; i < cells; ++i) coor_to_voxelidx[i] = -1;
  int voxel_num = 0;
  for (int i = 0; i < n; ++i) {
    int coor[3]; /* (z,y,x) */
    int failed = 0;
    for (int j = 0; j < 3; ++j) {
      const float t = floorf((points[(size_t)i * f + j] - coors_range[j]) / voxel_size[j]);
      /* NaN / overflow behave like the x86 float->int conversion (INT_MIN): dropped. */
      if (!(t >= 0.f && t < (float)grid[j])) { failed = 1; break; }
      coor[2 - j] = (int)t;
    }
    if (failed) continue;
    const size_t cell = ((size_t)coor[0] * grid[1] + coor[1]) * grid[0] + coor[2];
    int voxelidx = coor_to_voxelidx[cell];
    if (voxelidx == -1) {
      voxelidx = voxel_num;
      if (max_voxels != -1 && voxel_num >= max_voxels) continue;
      voxel_num += 1;
      coor_to_voxelidx[cell] = voxelidx;
      for (int k = 0; k < 3; ++k) coors[(size_t)voxelidx * 3 + k] = coor[k];
    }
    const int num = num_points_per_voxel[voxelidx];
    if (max_points == -1 || num < max_points) {
      memcpy(voxels + ((size_t)voxelidx * max_points + num) * f, points + (size_t)i * f,
             (size_t)f * sizeof(float));
      num_points_per_voxel[voxelidx] = num + 1;
    }
  }
  free(coor_to_voxelidx);
  return voxel_num;
}

/* PointPillarsScatter.forward_batch: per sample canvas = zeros(C, ny*nx);
 * canvas[:, y*nx + x] = feats.t(); stacked to [B, C, ny, nx].  coors rows are (b,z,y,x). */
void oracle_pillar_scatter(const float* feats, const int* coors, int m, int c, int batch, int ny,
                           int nx, float* canvas) {
  memset(canvas, 0, (size_t)batch * c * ny * nx * sizeof(float));
  for (int v = 0; v < m; ++v) {
    const int b = coors[v * 4 + 0], y = coors[v * 4 + 2], x = coors[v * 4 + 3];
    for (int ch = 0; ch < c; ++ch)
      canvas[(((size_t)b * c + ch) * ny + y) * nx + x] = feats[(size_t)v * c + ch];
  }
}
