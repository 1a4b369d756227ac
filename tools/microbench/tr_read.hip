// What ds_read_b64_tr_b16 returns (gfx950): every lane of a 16-lane group hands in the address of 4 consecutive 16-bit elements; the 16 x 4
// elements are taken as a [4 rows][16 columns] block (lane i = row i / 4, columns 4 (i % 4) ..) and lane i gets column i back, rows 0..3.
// Prints, for the image s[k] = k, the four values every lane receives.   hipcc --offload-arch=gfx950 -O2 tr_read.hip -o tr_read && ./tr_read
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short s[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  const unsigned short* p = s + g * 1024 + (i >> 2) * 72 + 4 * (i & 3);   // rows 72 elements (144 bytes) apart, as in conv_wgrad.hip
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; j++) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d;
  unsigned short h[256];
  hipMalloc(&d, sizeof(h));
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) {
    const int g = l >> 4, i = l & 15;
    printf("lane %2d:", l);
    for (int j = 0; j < 4; j++) {
      printf(" %4d", h[l * 4 + j]);
      bad += h[l * 4 + j] != g * 1024 + j * 72 + i;   // row j, column i of the group's block
    }
    printf("\n");
  }
  printf("%s\n", bad ? "UNEXPECTED LAYOUT" : "as documented: lane i of a group receives column i, rows 0..3");
  return bad != 0;
}
