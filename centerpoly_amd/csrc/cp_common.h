// Shared helpers for libcenterpoly_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "centerpoly_hip.h"

#define CP_WAVE 64

#define CP_CHECK_ARG(cond) \
  do {                     \
    if (!(cond)) return CP_EINVAL; \
  } while (0)

static inline int cp_launch_status() {
  return hipGetLastError() == hipSuccess ? CP_OK : CP_EHIP;
}

static inline size_t cp_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Wave-wide sum (64 lanes) through DPP/shuffles.
__device__ __forceinline__ float cp_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ double cp_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
