// Microbenchmark: LDS atomic throughput per wave-instruction on gfx950 (f32 vs u32 vs u64 vs store).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int stride) {
  __shared__ unsigned long long buf[4096];
  float* f = (float*)buf;
  unsigned* u = (unsigned*)buf;
  for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int idx = (w * 512 + lane * stride) & 2047;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) atomicAdd(&f[idx], 1.0f);
    if (MODE == 1) atomicAdd(&u[idx], 1u);
    if (MODE == 2) atomicAdd(&buf[idx], 1ull);
    if (MODE == 3) f[idx] = (float)i;
    if (MODE == 4) { float v = f[idx]; f[idx] = v + 1.f; }
    idx = (idx + 67) & 2047;
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = f[3] + (float)u[5];
}

template <int MODE>
float run(int iters, int stride) {
  float* out; hipMalloc(&out, 4096 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, out, iters, stride);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, out, iters, stride);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipFree(out);
  return ms;
}

int main() {
  const int iters = 2000;
  const char* names[5] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "ds_write_b32", "read+write"};
  for (int stride = 1; stride <= 2; ++stride) {
    float ms[5] = {run<0>(iters, stride), run<1>(iters, stride), run<2>(iters, stride), run<3>(iters, stride), run<4>(iters, stride)};
    for (int m = 0; m < 5; ++m) {
      // 1024 blocks x 4 waves x iters wave-instructions over 256 CUs (4 blocks/CU resident)
      double winst_per_cu = 1024.0 * 4 * iters / 256.0;
      double cyc = ms[m] * 1e-3 * 2.4e9 / winst_per_cu;
      printf("stride %d %-14s %8.3f ms  ~%6.1f cycles per wave-instruction per CU\n", stride, names[m], ms[m], cyc);
    }
  }
  return 0;
}
