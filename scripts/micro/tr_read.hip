// ds_read_b64_tr_b16 on gfx950: which element does which lane get?  (round 5: groundwork for reading NHWC tiles transposed in the
// weight gradient.)  An LDS tile [64 pixels][128 channels] of 16-bit values v = pixel * 128 + channel; every 16-lane group g of a wave
// reads the [4 pixel][16 channel] block at pixel base 8 * (g >> 1) (+ 4 for the second read), channel base 16 * (g & 1): lane i of the
// group supplies the address of pixel (i >> 2), channels 4 * (i & 3) .. +3.  Hypothesis (cdna_hip_programming.md T10): lane i receives
// channel i of the block for the 4 pixels, i.e. 4 consecutive k of an MFMA A/B fragment row.
// Second part: time a loop of such reads on rows 256 B apart, plain vs with the 32-byte units XOR-swizzled by the pixel (bank conflicts).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_v;

__global__ void k_map(const short* in, short* out) {
  __shared__ __attribute__((aligned(16))) short sm[64 * 128];
  for (int i = threadIdx.x; i < 64 * 128; i += 64) sm[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, li = l & 15;
  for (int half = 0; half < 2; ++half) {
    const int row = 8 * (g >> 1) + 4 * half + (li >> 2), col = 16 * (g & 1) + 4 * (li & 3);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)&sm[row * 128 + col]);
    for (int j = 0; j < 4; ++j) out[(l * 2 + half) * 4 + j] = v[j];
  }
}

template <bool SWZ>
__global__ void k_time(const short* in, int iters, long long* cycles, int* sink) {
  __shared__ __attribute__((aligned(16))) short sm[64 * 128];
  for (int i = threadIdx.x; i < 64 * 128; i += blockDim.x) sm[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x & 63, g = l >> 4, li = l & 15;
  int acc = 0;
  const long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {           // 8 fragments: pixels kk*8 ..
      const int row = kk * 8 + 8 * 0 + 4 * (g >> 1) + (li >> 2);
      int unit = (g & 1) + 2 * (it & 3);       // 32-byte unit of the row (16 channels)
      if (SWZ) unit ^= (row & 7);
      const int col = unit * 16 + 4 * (li & 3);
      s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)&sm[row * 128 + col]);
      acc += v[0] + v[3];
    }
  }
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
  if (acc == 0x12345678) *sink = acc;
}

int main() {
  std::vector<short> h(64 * 128);
  for (int p = 0; p < 64; ++p) for (int c = 0; c < 128; ++c) h[p * 128 + c] = (short)(p * 128 + c);
  short *d_in, *d_out; long long* d_cyc; int* d_sink;
  hipMalloc(&d_in, h.size() * 2); hipMalloc(&d_out, 64 * 8 * 2); hipMalloc(&d_cyc, 8); hipMalloc(&d_sink, 4);
  hipMemcpy(d_in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, d_in, d_out);
  std::vector<short> o(64 * 8);
  hipMemcpy(o.data(), d_out, o.size() * 2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int g = l >> 4, i = l & 15;
    for (int half = 0; half < 2; ++half)
      for (int j = 0; j < 4; ++j) {
        const int want = (8 * (g >> 1) + 4 * half + j) * 128 + 16 * (g & 1) + i;      // pixel k, channel of lane i
        if (o[(l * 2 + half) * 4 + j] != want) ++bad;
      }
  }
  printf("mapping: %d mismatches of 512 (hypothesis: lane i of a 16-lane group gets channel i, elements = the block's 4 pixels)\n", bad);
  for (int l : {0, 1, 5, 16, 33, 63}) {
    printf("  lane %2d:", l);
    for (int q = 0; q < 8; ++q) printf(" (p%d,c%d)", o[l * 8 + q] / 128, o[l * 8 + q] % 128);
    printf("\n");
  }
  for (int swz = 0; swz < 2; ++swz) {
    long long cyc = 0;
    for (int rep = 0; rep < 3; ++rep) {
      if (swz) hipLaunchKernelGGL((k_time<true>), dim3(256), dim3(256), 0, 0, d_in, 2000, d_cyc, d_sink);
      else hipLaunchKernelGGL((k_time<false>), dim3(256), dim3(256), 0, 0, d_in, 2000, d_cyc, d_sink);
      hipDeviceSynchronize();
      hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    }
    printf("%s: %.1f wall-clock ticks per ds_read_b64_tr_b16 wave-instruction (4 waves per workgroup, 256 workgroups)\n",
           swz ? "units XOR-swizzled by the pixel" : "plain rows 256 B apart      ", (double)cyc / (2000.0 * 8));
  }
  return 0;
}
