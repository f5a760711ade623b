// Micro-benchmark: do 256-byte row gathers (16 lanes x 16 B, L2-resident table) and VALU work overlap on gfx950?
// Each group of 16 lanes gathers N rows in batches of 8; per gathered row it executes M extra v_pk_fma_f32 pairs on registers
// that do not depend on the loaded data (plus the 2 that consume it).  If gathers and VALU overlap, time ~ max(T_gather(M=0),
// T_valu); if they do not, time ~ sum.  Build: hipcc --offload-arch=gfx950 -O3 gather_valu_overlap.hip -o gather_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int M, bool GATHER>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ table, const int* __restrict__ idx, int n_per_group, float4* __restrict__ out) {
  const int sub = threadIdx.x & 15;
  const long g = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int* my = idx + g * n_per_group;
  f2 acc0 = {0, 0}, acc1 = {0, 0}, x0 = {1.0f + sub, 0.5f}, x1 = {0.25f, 2.0f + sub};
  for (int i = 0; i < n_per_group; i += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (GATHER) v[u] = table[(size_t)my[i + u] * 16 + sub];
      else v[u] = make_float4(my[i + u], sub, i, u);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      f2 a = {v[u].x, v[u].y}, b = {v[u].z, v[u].w};
      acc0 = __builtin_elementwise_fma(a, x0, acc0);
      acc1 = __builtin_elementwise_fma(b, x1, acc1);
#pragma unroll
      for (int m = 0; m < M; ++m) {          // independent VALU work (does not wait for the gather)
        x0 = __builtin_elementwise_fma(x0, x1, x0);
        x1 = __builtin_elementwise_fma(x1, x0, x1);
      }
    }
  }
  out[g * 16 + sub] = make_float4(acc0.x + x0.x, acc0.y + x0.y, acc1.x + x1.x, acc1.y + x1.y);
}

template <int M, bool GATHER>
float run(const float4* table, const int* idx, int npg, float4* out, int blocks) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k<M, GATHER>), dim3(blocks), dim3(256), 0, 0, table, idx, npg, out); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<M, GATHER>), dim3(blocks), dim3(256), 0, 0, table, idx, npg, out);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms * 100.f;
}

int main() {
  const int blocks = 2048, groups = blocks * 16, npg = 64;
  const long total = (long)groups * npg;
  const size_t rows = 8192;                               // 2 MB: L2-resident
  float4 *table, *out; int* idx;
  (void)hipMalloc(&table, rows * 256); (void)hipMalloc(&out, (size_t)groups * 256); (void)hipMalloc(&idx, total * 4);
  (void)hipMemset(table, 0, rows * 256);
  std::vector<int> h(total); srand(1);
  for (long i = 0; i < total; ++i) h[i] = rand() % rows;
  (void)hipMemcpy(idx, h.data(), total * 4, hipMemcpyHostToDevice);
#define LINE(M) printf("M=%2d extra pk_fma pairs per row: gathers+valu %7.1f us | valu only %7.1f us | (%.1f rows/ns)\n", M, \
    run<M, true>(table, idx, npg, out, blocks), run<M, false>(table, idx, npg, out, blocks), total / run<M, true>(table, idx, npg, out, blocks) / 1e3);
  LINE(0) LINE(2) LINE(4) LINE(8) LINE(16) LINE(32)
  return 0;
}
