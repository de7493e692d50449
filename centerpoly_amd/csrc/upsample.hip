// Depth-wise transposed convolution (kernel 2f, stride f, padding f/2, groups = C) fused
// with the following skip add, for gfx950.
//
// Replaces IDAUp's `up` = nn.ConvTranspose2d(o, o, 2f, stride=f, padding=f//2, groups=o,
// bias=False) followed by `layers[i] + layers[i-1]`
// (reference: src/lib/models/networks/pose_dla_dcn.py:372-375, 381-387).  The library path
// (col2im + batched GEMM + transposes) spends ~100 us per layer on a memory-bound op; this is
// one streaming pass: each output pixel reads its 2x2 contributing inputs (L1/L2 resident)
// and the skip map, float4 stores along x.  HBM-bound: 4*(C*H*W + 2*C*f*f*H*W) bytes.
#include "cp_common.h"

namespace {

template <int F>
__global__ __launch_bounds__(256) void dw_up_kernel(const float* __restrict__ x,
                                                    const float* __restrict__ w,
                                                    const float* __restrict__ skip,
                                                    float* __restrict__ out, int C, int H, int W) {
  constexpr int KS = 2 * F, PAD = F / 2;
  const int Ho = H * F, Wo = W * F;
  const int bc = blockIdx.z;                       // b*C + c
  const int c = bc % C;
  const int oy = blockIdx.y;
  const int ox0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (ox0 >= Wo) return;
  const float* wc = w + (long long)c * KS * KS;
  const float* xc = x + (long long)bc * H * W;
  // rows: oy = iy*F - PAD + ky  ->  ky in {ky0, ky0 + F}, ky0 = (oy + PAD) % F
  const int ky0 = (oy + PAD) % F;
  const int iy0 = (oy + PAD - ky0) / F;            // pairs (iy0, ky0), (iy0 - 1, ky0 + F)
  float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int iy = iy0 - r, ky = ky0 + r * F;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ox = ox0 + q;
      const int kx0 = (ox + PAD) % F;
      const int ix0 = (ox + PAD - kx0) / F;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ix = ix0 - t, kx = kx0 + t * F;
        if (ix < 0 || ix >= W || ox >= Wo) continue;
        v[q] += xc[iy * W + ix] * wc[ky * KS + kx];
      }
    }
  }
  const long long o = ((long long)bc * Ho + oy) * Wo + ox0;
  if (skip) {
    if (ox0 + 3 < Wo) {
      const f32x4 s = *reinterpret_cast<const f32x4*>(skip + o);
      v[0] += s[0]; v[1] += s[1]; v[2] += s[2]; v[3] += s[3];
    } else {
      for (int q = 0; q < 4 && ox0 + q < Wo; ++q) v[q] += skip[o + q];
    }
  }
  if (ox0 + 3 < Wo) *reinterpret_cast<f32x4*>(out + o) = f32x4{v[0], v[1], v[2], v[3]};
  else
    for (int q = 0; q < 4 && ox0 + q < Wo; ++q) out[o + q] = v[q];
}

}  // namespace

extern "C" int cp_depthwise_up_forward(const float* x, const float* weight, const float* skip,
                                       float* out, int32_t B, int32_t C, int32_t H, int32_t W,
                                       int32_t f, void* stream) {
  CP_CHECK_ARG(x && weight && out && B > 0 && C > 0 && H > 0 && W > 0);
  if (f != 2 && f != 4 && f != 8) return CP_EUNSUPPORTED;
  if ((W * f) % 4 != 0 || (long long)B * C > 65535 || (long long)H * f > 65535) return CP_EUNSUPPORTED;
  const int Wo = W * f, Ho = H * f;
  dim3 grid((Wo / 4 + 255) / 256, Ho, B * C);
  hipStream_t st = (hipStream_t)stream;
  if (f == 2) hipLaunchKernelGGL(dw_up_kernel<2>, grid, dim3(256), 0, st, x, weight, skip, out, C, H, W);
  else if (f == 4) hipLaunchKernelGGL(dw_up_kernel<4>, grid, dim3(256), 0, st, x, weight, skip, out, C, H, W);
  else hipLaunchKernelGGL(dw_up_kernel<8>, grid, dim3(256), 0, st, x, weight, skip, out, C, H, W);
  return cp_launch_status();
}
