// Micro-test: is the scalar offset of a raw buffer access part of the range check on gfx950? (tools/r05 notes)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *p, int *res) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *) p, 0, 16, 0x00020000);   // 4 floats
  if (threadIdx.x == 0) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(1.f), rs, 0, 16, 0);      // voffset 0, soffset 16 -> element 4
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(2.f), rs, 16, 0, 0);      // voffset 16 -> element 4, out of range
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(3.f), rs, 12, 16, 0);     // voffset 12 (in range) + soffset 16 -> element 7
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(4.f), rs, 8, 0, 0);       // element 2, in range
  }
}
int main() {
  float *d; int *r;
  hipMalloc(&d, 64); hipMalloc(&r, 4); hipMemset(d, 0, 64);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, r);
  float h[16]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  for (int i = 0; i < 8; i++) printf("%g ", h[i]);
  printf("\n(element 4 = 1: soffset not range-checked; element 7 = 3: the check is on voffset alone)\n");
  return 0;
}
