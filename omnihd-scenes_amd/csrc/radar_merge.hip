// Radar sweep merge on the device: raw sensor-frame returns of up to 3 sweeps x 6 radars ->
// rows [x, y, z, vx_comp, vy_comp, power, snr, dt, Vr_comp, radar_id] in the current LiDAR frame, plus the
// strict-inequality range mask — the arithmetic of the reference's LoadRadarPointsMultiSweeps.__call__
// (projects/mmdet3d_plugin/datasets/pipelines/loading.py:229-309), one thread per return.
//
// The reference mixes precisions (numpy promotion): geometry of the return in float32 (range, azimuth, elevation
// and their cos/sin), everything multiplied by the float64 ego velocity / rotation in float64, the position rotated in
// float64 and stored back as float32.  The same promotions are made here, with floating-point contraction off, so
// the positions (which decide the voxel of a point) are bit-identical to the host loader; the velocity columns can
// differ in the last bits of the float32 trigonometry (numpy's SIMD atan2f/asinf/cosf/sinf vs the device's).
// Latency-bound: ~20 k returns per frame, one launch.
#include "common.h"

#pragma clang fp contract(off)

namespace omnihd {
namespace {

// per sweep: v_sensor[3], R[9] (sensor -> lidar, row major), t[3], dt, radar_id  = 17 doubles
constexpr int kSweepDoubles = 17;

__global__ __launch_bounds__(256) void k_radar_merge(const float* __restrict__ raw, int load_dim,
                                                     const int* __restrict__ sweep_offsets, int n_sweeps,
                                                     const double* __restrict__ sweep_consts,
                                                     const float* __restrict__ pc_range, float* __restrict__ out,
                                                     unsigned char* __restrict__ in_range, int n) {
  __shared__ int s_off[65];
  for (int i = threadIdx.x; i <= n_sweeps && i < 65; i += 256) s_off[i] = sweep_offsets[i];
  __syncthreads();
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  int s = 0;
  while (s + 1 < n_sweeps && p >= s_off[s + 1]) ++s;
  const double* c = sweep_consts + (size_t)s * kSweepDoubles;
  const float* r = raw + (size_t)p * load_dim;
  const float x = r[0], y = r[1], z = r[2], vr = r[3];
  // float32 geometry, as numpy computes it on the float32 columns
  const float rng = sqrtf((x * x + y * y) + z * z);
  const float az = atan2f(y, x);
  const float el = asinf(z / rng);
  const float caz = cosf(az), saz = sinf(az), cel = cosf(el), sel = sinf(el);
  // float64 from here on (float64 ego velocity x float32 trigonometry)
  const double vr_comp = ((c[0] * (double)caz) * (double)cel + (c[1] * (double)saz) * (double)cel) + c[2] * (double)sel + (double)vr;
  const double vx = (vr_comp * (double)cel) * (double)caz;
  const double vy = (vr_comp * (double)cel) * (double)saz;
  // (vx, vy, 0) @ R^T and the position @ R^T + t; the position is rounded to float32 BEFORE the translation is added
  const double* R = c + 3;
  const double vlx = (vx * R[0] + vy * R[1]) + 0.0 * R[2];
  const double vly = (vx * R[3] + vy * R[4]) + 0.0 * R[5];
  float px = (float)(((double)x * R[0] + (double)y * R[1]) + (double)z * R[2]);
  float py = (float)(((double)x * R[3] + (double)y * R[4]) + (double)z * R[5]);
  float pz = (float)(((double)x * R[6] + (double)y * R[7]) + (double)z * R[8]);
  px = (float)((double)px + c[12]);
  py = (float)((double)py + c[13]);
  pz = (float)((double)pz + c[14]);
  float* o = out + (size_t)p * 10;
  o[0] = px; o[1] = py; o[2] = pz;
  o[3] = (float)vlx; o[4] = (float)vly;
  o[5] = r[4]; o[6] = r[6];
  o[7] = (float)c[15];
  o[8] = (float)vr_comp;
  o[9] = (float)c[16];
  if (in_range)
    in_range[p] = (px > pc_range[0]) && (py > pc_range[1]) && (pz > pc_range[2]) && (px < pc_range[3]) &&
                  (py < pc_range[4]) && (pz < pc_range[5]);
}

}  // namespace
}  // namespace omnihd

using namespace omnihd;

extern "C" int omnihd_radar_merge(const float* raw, int n, int load_dim, const int* sweep_offsets, int n_sweeps,
                                  const double* sweep_consts, const float* pc_range6, float* out10,
                                  unsigned char* in_range, void* stream) {
  OMNIHD_REQUIRE(n >= 0 && load_dim >= 7 && n_sweeps >= 1 && n_sweeps <= 64, "n >= 0, load_dim >= 7, 1..64 sweeps");
  if (n == 0) return OMNIHD_OK;
  OMNIHD_REQUIRE(raw && sweep_offsets && sweep_consts && out10 && (in_range == nullptr || pc_range6), "null pointer");
  hipLaunchKernelGGL(k_radar_merge, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, raw, load_dim, sweep_offsets,
                     n_sweeps, sweep_consts, pc_range6, out10, in_range, n);
  return check_launch("radar_merge");
}
