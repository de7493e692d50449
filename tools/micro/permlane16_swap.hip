// What v_permlane16_swap_b32 does (gfx950): prints the two result registers for x = lane, y = 100 + lane.
// Measured: r[0] = [x.row0, y.row0, x.row2, y.row2], r[1] = [x.row1, y.row1, x.row3, y.row3] (rows of 16 lanes): the odd rows
// of the first operand trade places with the even rows of the second (used by conv_mfma.hip's epilogue).
// Build: hipcc -O3 --offload-arch=gfx950 permlane16_swap.hip -o permlane16_swap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  unsigned x = threadIdx.x, y = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  unsigned r0 = r[0], r1 = r[1];
  asm("" : "+v"(r0), "+v"(r1));
  o[threadIdx.x] = r0;
  o[64 + threadIdx.x] = r1;
}
int main() {
  unsigned* d; hipMalloc(&d, 512); k<<<1, 64>>>(d); unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int j = 0; j < 2; ++j) { printf("r[%d]:", j); for (int i = 0; i < 64; ++i) printf(" %u", h[64 * j + i]); printf("\n"); }
  return 0;
}
