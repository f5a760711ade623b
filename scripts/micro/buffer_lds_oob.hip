// Does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` write ZEROS into its LDS slot, or skip the write?
// (The convolution kernels select the zero padding of a row through the buffer's range check.)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_ptr_t;
__global__ void k(const unsigned short* x, int nbytes, float* out) {
  __shared__ __attribute__((aligned(16))) unsigned short sm[64 * 8];
  for (int i = threadIdx.x; i < 64 * 8; i += 64) sm[i] = 7;             // stale content
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
  unsigned voff = threadIdx.x * 16;
  if (threadIdx.x == 5) voff = 0xfffffff0u;                               // beyond num_records
  if (threadIdx.x == 9) voff = nbytes - 8;                                // straddles the end
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t*)&sm[0], 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = 0; e < 8; ++e) out[threadIdx.x * 8 + e] = (float)sm[threadIdx.x * 8 + e];
}
int main() {
  unsigned short h[64 * 8];
  for (int i = 0; i < 64 * 8; ++i) h[i] = 100 + i;
  unsigned short* d; float* o; float ho[64 * 8];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, (int)sizeof(h), o);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  printf("lane 4 :"); for (int e = 0; e < 8; ++e) printf(" %g", ho[4 * 8 + e]); printf("   (expect 132..139)\n");
  printf("lane 5 :"); for (int e = 0; e < 8; ++e) printf(" %g", ho[5 * 8 + e]); printf("   (beyond the buffer: zeros = written, 7 = skipped)\n");
  printf("lane 9 :"); for (int e = 0; e < 8; ++e) printf(" %g", ho[9 * 8 + e]); printf("   (straddling the end)\n");
  return 0;
}
