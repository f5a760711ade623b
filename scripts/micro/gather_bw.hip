// Micro-benchmark: achievable rate of 256-byte row gathers (16 lanes x 16 B) on MI355X as a
// function of the table size (L2-resident .. Infinity-Cache .. HBM), with U gathers in flight per
// group.  Build: hipcc --offload-arch=gfx950 -O3 gather_bw.hip -o gather_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int U>
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ table, const int* __restrict__ idx,
                                                int n_per_group, float4* __restrict__ out) {
  const int sub = threadIdx.x & 15;
  const long g = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int* my = idx + g * n_per_group;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int i = 0; i < n_per_group; i += U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = table[(size_t)my[i + u] * 16 + sub];
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  out[g * 16 + sub] = acc;
}

int main() {
  const int blocks = 2048, groups = blocks * 16, n_per_group = 64;   // 2.1 M row gathers = 537 MB
  const long total = (long)groups * n_per_group;
  float4 *table, *out; int* idx;
  const size_t max_rows = 1 << 20;
  hipMalloc(&table, max_rows * 256); hipMalloc(&out, (size_t)groups * 256); hipMalloc(&idx, total * 4);
  hipMemset(table, 0, max_rows * 256);
  std::vector<int> h(total);
  for (size_t rows : {4096ul, 8192ul, 16384ul, 67584ul, 262144ul, 1048576ul}) {
    for (int mode = 0; mode < 2; ++mode) {   // 0: random rows; 1: locality (each block walks a 512-row window)
      srand(1);
      for (long i = 0; i < total; ++i) {
        if (mode == 0) h[i] = rand() % rows;
        else { long blk = i / (16L * n_per_group); h[i] = (int)((blk * 37 % (rows / 512 ? rows / 512 : 1)) * 512 % rows + rand() % 512) % rows; }
      }
      hipMemcpy(idx, h.data(), total * 4, hipMemcpyHostToDevice);
      for (int U : {4, 8, 16}) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        auto launch = [&]() {
          if (U == 4) hipLaunchKernelGGL(k_gather<4>, dim3(blocks), dim3(256), 0, 0, table, idx, n_per_group, out);
          else if (U == 8) hipLaunchKernelGGL(k_gather<8>, dim3(blocks), dim3(256), 0, 0, table, idx, n_per_group, out);
          else hipLaunchKernelGGL(k_gather<16>, dim3(blocks), dim3(256), 0, 0, table, idx, n_per_group, out);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(a); for (int k = 0; k < 10; ++k) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double us = ms * 100.0;
        printf("table %7zu rows (%6.1f MB) %s U=%2d: %7.1f us  %6.2f TB/s gathered  %5.1f rows/ns\n", rows, rows * 256 / 1e6,
               mode ? "windowed" : "random  ", U, us, total * 256.0 / us / 1e6, total / us / 1e3);
      }
    }
  }
  return 0;
}
