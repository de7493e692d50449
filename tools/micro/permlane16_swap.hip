// What v_permlane16_swap_b32 does (gfx950): prints the two result registers for x = lane, y = 100 + lane.
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
