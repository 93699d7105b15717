// Probe of ds_read_b64_tr_b16 lane/element semantics on gfx950 (prints which LDS element each lane receives).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, int mode) {
  __shared__ unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  int l = threadIdx.x;
  int a;
  if (mode == 0) a = l * 4;                                                  // lane-linear 8-byte pieces
  else if (mode == 1) a = (l & 15) * 64 + (l >> 4) * 4;                       // one row (pitch 64) per lane of a 16-group
  else if (mode == 2) a = (l & 3) * 4 + ((l >> 2) & 3) * 64 + (l >> 4) * 256; // 4 lanes per row x 4 rows per 16-group
  else a = ((l & 15) >> 2) * 64 + (l & 3) * 4 + (l >> 4) * 1024;            // same as 2 with other group stride
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + a));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("  lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %5d", h[l * 4 + j]); if (l % 4 == 3) printf("\n"); }
  }
  return 0;
}
