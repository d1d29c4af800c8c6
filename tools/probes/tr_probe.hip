// Prints what ds_read_b64_tr_b16 returns on this GPU: LDS holds lds[i] = i (16-bit), lane l reads at
// byte address 8*l; output = the 4 element indices each lane received.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ void k(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = i;
  __syncthreads();
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)((char*)lds + threadIdx.x * 8));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)t[j];
}
int main() {
  uint16_t* d; uint16_t h[256];
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) { printf("no gpu\n"); return 1; }
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
