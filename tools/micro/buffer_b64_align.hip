// Does an 8-byte raw buffer load from a 4-byte-aligned (not 8-byte-aligned) offset return the
// two dwords at that offset on gfx950?  (hipcc --offload-arch=gfx950 buffer_b64_align.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void k(const unsigned* src, unsigned* dst, int n, unsigned soff) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, n * 4, 0x00020000);
  const int i = threadIdx.x;
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, 4u * (unsigned)i, soff, 0);
  dst[2 * i] = v.x;
  dst[2 * i + 1] = v.y;
}

int main() {
  const int n = 256;
  unsigned h[n], *d, *o, r[128];
  for (int i = 0; i < n; ++i) h[i] = 1000 + i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (unsigned soff : {0u, 4u, 64u}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n, soff);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i)
      if (r[2 * i] != 1000 + i + soff / 4 || r[2 * i + 1] != 1001 + i + soff / 4) ++bad;
    printf("soffset %u: %d bad lanes; lane1 -> %u %u (want %u %u)\n", soff, bad, r[2], r[3], 1001 + soff / 4, 1002 + soff / 4);
  }
  return 0;
}
