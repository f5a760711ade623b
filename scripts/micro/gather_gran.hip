// Micro-benchmark: does the L1/TA path deliver more bytes per clock when a wave's 1 KB request is contiguous?
// L lanes x 16 B per gathered chunk (L = 16: 256-byte rows, 4 per wave instruction; 32: 512 B; 64: one contiguous 1 KB), random
// chunk indices in an L2-resident (2 MB) or Infinity-Cache-resident (64 MB) table, 8 requests in flight per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
template <int L>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ table, const int* __restrict__ idx, int n_per_group, float4* __restrict__ out) {
  const int sub = threadIdx.x % L;
  const long g = ((long)blockIdx.x * 256 + threadIdx.x) / L;
  const int* my = idx + g * n_per_group;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int i = 0; i < n_per_group; i += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = table[(size_t)my[i + u] * L + sub];
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  out[(long)blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int L>
void run(const float4* table, size_t table_bytes, int* idx, std::vector<int>& h, float4* out) {
  const int blocks = 2048, npg = 64;
  const long groups = (long)blocks * 256 / L, total = groups * npg;
  const long chunks = table_bytes / (L * 16);
  srand(1);
  for (long i = 0; i < total; ++i) h[i] = rand() % chunks;
  (void)hipMemcpy(idx, h.data(), total * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<L>, dim3(blocks), dim3(256), 0, 0, table, idx, npg, out); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k<L>, dim3(blocks), dim3(256), 0, 0, table, idx, npg, out);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double us = ms * 100.0, bytes = (double)total * L * 16;
  printf("table %6.1f MB, chunk %4d B: %7.1f us  %6.2f TB/s\n", table_bytes / 1e6, L * 16, us, bytes / us / 1e6);
}
int main() {
  float4 *table, *out; int* idx;
  (void)hipMalloc(&table, 64u << 20); (void)hipMalloc(&out, 2048 * 256 * 16); (void)hipMalloc(&idx, 2048l * 16 * 64 * 4);
  (void)hipMemset(table, 0, 64u << 20);
  std::vector<int> h(2048l * 16 * 64);
  for (size_t tb : {2ul << 20, 64ul << 20}) { run<16>(table, tb, idx, h, out); run<32>(table, tb, idx, h, out); run<64>(table, tb, idx, h, out); }
  return 0;
}
