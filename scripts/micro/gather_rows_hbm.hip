// Micro-benchmark: what does HBM deliver for the pooling backward's access pattern?  256-byte rows (16 lanes x 16 B) gathered from
// a table much larger than the Infinity Cache (1 GiB), every row read ONCE per launch in a random order (so nothing is served
// from a cache: every byte comes from HBM), 8 requests in flight per lane, 8 waves per SIMD.  Variants: the rows in sequential
// order (a streaming read through the same kernel), random rows of 512 B / 1 KiB, and a fraction of the rows only.
// build: hipcc --offload-arch=gfx950 -O3 -o gather_rows_hbm gather_rows_hbm.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <numeric>
#include <random>
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
  if (acc.x == 1.2345e30f) out[(long)blockIdx.x * 256 + threadIdx.x] = acc;      // never true: no store traffic
}
template <int L>
void run(const float4* table, size_t table_bytes, int* idx, float4* out, bool random, double fraction, const char* what) {
  const long chunks = table_bytes / (L * 16);
  const int npg = 64;
  long use = (long)(chunks * fraction) / npg * npg;
  const long groups = use / npg;
  const int gpb = 256 / L;
  const int blocks = (int)(groups / gpb);
  use = (long)blocks * gpb * npg;
  std::vector<int> h(chunks);
  std::iota(h.begin(), h.end(), 0);
  if (random) { std::mt19937 rng(7); std::shuffle(h.begin(), h.end(), rng); }
  (void)hipMemcpy(idx, h.data(), use * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<L>, dim3(blocks), dim3(256), 0, 0, table, idx, npg, out); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<L>, dim3(blocks), dim3(256), 0, 0, table, idx, npg, out);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1000.0 / reps, bytes = (double)use * L * 16;
  printf("%-34s chunk %4d B, %7.1f MB read (+ %5.1f MB of indices): %8.1f us  %5.2f TB/s\n", what, L * 16, bytes / 1e6, use * 4 / 1e6, us, bytes / us / 1e6);
}
int main() {
  const size_t tb = 1ul << 30;
  float4 *table, *out; int* idx;
  (void)hipMalloc(&table, tb); (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&idx, (tb / 256) * 4);
  (void)hipMemset(table, 0, tb);
  run<16>(table, tb, idx, out, false, 1.0, "sequential rows (streaming)");
  run<16>(table, tb, idx, out, true, 1.0, "random rows, each once");
  run<32>(table, tb, idx, out, true, 1.0, "random rows, each once");
  run<64>(table, tb, idx, out, true, 1.0, "random rows, each once");
  run<16>(table, tb, idx, out, true, 0.25, "random quarter of the rows");
  run<16>(table, tb, idx, out, true, 0.15, "random 15 % of the rows (157 MB)");
  return 0;
}
