// Micro-benchmark: variants of the BatchNorm backward-apply pass (gx = g' * A[c] + x * B[c] + C[c], g' = gy * [y > 0]) on an fp32
// NHWC tensor of 1024 channels x 38400 pixels (157 MB): which form reaches copy bandwidth?
// Build: hipcc --offload-arch=gfx950 -O3 bn_apply_bw.hip -o bn_apply_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// V0: the product kernel's form: 8 channels per lane, non-temporal loads, coefficients from global memory, grid-stride
template <bool NT, int VPT>
__global__ __launch_bounds__(256) void k_v0(const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ x,
                                            const float* __restrict__ cA, const float* __restrict__ cB, const float* __restrict__ cC,
                                            float* __restrict__ gx, long n_vec, int c8) {
  for (long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * VPT; i0 < n_vec; i0 += (long)gridDim.x * 256 * VPT) {
    f32x4 g[VPT][2], m[VPT][2], v[VPT][2];
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
      const long i = i0 + u;
      const f32x4* pg = reinterpret_cast<const f32x4*>(gy) + 2 * i;
      const f32x4* pm = reinterpret_cast<const f32x4*>(y) + 2 * i;
      const f32x4* px = reinterpret_cast<const f32x4*>(x) + 2 * i;
      if (NT) {
        g[u][0] = __builtin_nontemporal_load(pg); g[u][1] = __builtin_nontemporal_load(pg + 1);
        m[u][0] = __builtin_nontemporal_load(pm); m[u][1] = __builtin_nontemporal_load(pm + 1);
        v[u][0] = __builtin_nontemporal_load(px); v[u][1] = __builtin_nontemporal_load(px + 1);
      } else {
        g[u][0] = pg[0]; g[u][1] = pg[1]; m[u][0] = pm[0]; m[u][1] = pm[1]; v[u][0] = px[0]; v[u][1] = px[1];
      }
    }
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
      const long i = i0 + u;
      const int c = (int)(i % c8) * 8;
      f32x4 o[2];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float gg = m[u][h][k] > 0.f ? g[u][h][k] : 0.f;
          o[h][k] = fmaf(gg, cA[c + 4 * h + k], fmaf(v[u][h][k], cB[c + 4 * h + k], cC[c + 4 * h + k]));
        }
      reinterpret_cast<f32x4*>(gx)[2 * i] = o[0];
      reinterpret_cast<f32x4*>(gx)[2 * i + 1] = o[1];
    }
  }
}

// V1: one float4 per lane (16 B), a row of 1024 channels = 256 lanes = one workgroup; coefficients in registers per lane;
// each workgroup walks rows blockIdx, blockIdx + grid, ...
template <int U>
__global__ __launch_bounds__(256) void k_v1(const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ x,
                                            const float* __restrict__ cA, const float* __restrict__ cB, const float* __restrict__ cC,
                                            float* __restrict__ gx, long rows, int c) {
  const int lane_c = threadIdx.x * 4;
  const f32x4 a = *reinterpret_cast<const f32x4*>(cA + lane_c), b = *reinterpret_cast<const f32x4*>(cB + lane_c),
              cc = *reinterpret_cast<const f32x4*>(cC + lane_c);
  for (long r0 = (long)blockIdx.x * U; r0 < rows; r0 += (long)gridDim.x * U) {
    f32x4 g[U], m[U], v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long off = (r0 + u < rows ? r0 + u : r0) * (c / 4) + threadIdx.x;
      g[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gy) + off);
      m[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(y) + off);
      v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x) + off);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r0 + u >= rows) break;
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = fmaf(m[u][k] > 0.f ? g[u][k] : 0.f, a[k], fmaf(v[u][k], b[k], cc[k]));
      __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(gx) + (r0 + u) * (c / 4) + threadIdx.x);
    }
  }
}

__global__ void k_copy(const f32x4* __restrict__ a, f32x4* __restrict__ b, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b[i] = a[i];
}

template <typename F>
float timeit(F f) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); for (int i = 0; i < 20; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms * 50.f;
}

int main() {
  const int c = 1024; const long rows = 38400, n = rows * c;
  float *gy, *y, *x, *gx, *cA, *cB, *cC;
  (void)hipMalloc(&gy, n * 4); (void)hipMalloc(&y, n * 4); (void)hipMalloc(&x, n * 4); (void)hipMalloc(&gx, n * 4);
  (void)hipMalloc(&cA, c * 4); (void)hipMalloc(&cB, c * 4); (void)hipMalloc(&cC, c * 4);
  {  // random data (a zero-filled input measures a different clock / different DRAM behaviour)
    float* h = (float*)malloc(n * 4);
    unsigned s = 12345u;
    for (int t = 0; t < 3; ++t) {
      for (long i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 22)); }
      (void)hipMemcpy(t == 0 ? gy : t == 1 ? y : x, h, n * 4, hipMemcpyHostToDevice);
    }
    free(h);
  } (void)hipMemset(cA, 0, c * 4); (void)hipMemset(cB, 0, c * 4); (void)hipMemset(cC, 0, c * 4);
  const long n_vec = n / 8; const double bytes = 4.0 * n * 4;
  auto rep = [&](const char* name, float us) { printf("%-58s %7.1f us  %5.2f TB/s\n", name, us, bytes / us / 1e6); };
  rep("copy of 2 x 157 MB (for scale: bytes counted as 4 tensors)", timeit([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const f32x4*)gy, (f32x4*)gx, n / 4); hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)y, n / 4); }));
  for (int grid : {2048, 4096, 8192, 16384})
    { char nm[96]; snprintf(nm, 96, "V0 product form (nt loads, 8 ch/lane), grid %d", grid);
      rep(nm, timeit([&] { hipLaunchKernelGGL((k_v0<true, 1>), dim3(grid), dim3(256), 0, 0, gy, y, x, cA, cB, cC, gx, n_vec, c / 8); })); }
  rep("V0 plain loads, grid 2048", timeit([&] { hipLaunchKernelGGL((k_v0<false, 1>), dim3(2048), dim3(256), 0, 0, gy, y, x, cA, cB, cC, gx, n_vec, c / 8); }));
  rep("V0 nt, 2 vectors per lane in flight, grid 2048", timeit([&] { hipLaunchKernelGGL((k_v0<true, 2>), dim3(2048), dim3(256), 0, 0, gy, y, x, cA, cB, cC, gx, n_vec, c / 8); }));
  for (int grid : {2048, 4096})
    { char nm[96]; snprintf(nm, 96, "V1 row per workgroup, coefficients in registers, U=2, grid %d", grid);
      rep(nm, timeit([&] { hipLaunchKernelGGL((k_v1<2>), dim3(grid), dim3(256), 0, 0, gy, y, x, cA, cB, cC, gx, rows, c); })); }
  rep("V1 U=4, grid 2048", timeit([&] { hipLaunchKernelGGL((k_v1<4>), dim3(2048), dim3(256), 0, 0, gy, y, x, cA, cB, cC, gx, rows, c); }));
  rep("V1 U=8, grid 2048", timeit([&] { hipLaunchKernelGGL((k_v1<8>), dim3(2048), dim3(256), 0, 0, gy, y, x, cA, cB, cC, gx, rows, c); }));
  return 0;
}
