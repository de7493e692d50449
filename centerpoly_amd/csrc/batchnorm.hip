// Fused training-mode BatchNorm2d (+ residual add) (+ ReLU), forward and backward, for gfx950.
//
// Replaces the BatchNorm -> (add) -> ReLU chains of the path in TRAINING:
// DeformConv.actf after DCN (reference: src/lib/models/networks/pose_dla_dcn.py:347-359),
// BasicBlock bn1/relu and bn2/+=residual/relu (:32-60), Root (:148-166), the conv levels
// (:266-277).  The library path reads/writes each activation 3x (BN) + 2x (ReLU) + 3x (add)
// forward and again backward; here forward = one statistics pass + one apply pass, backward =
// one reduction pass + one apply pass.  HBM bound: fp32 NCHW, float4 per lane.
//
// Semantics = torch.nn.BatchNorm2d in training mode: biased variance for normalisation,
// running_mean/var updated with `momentum` (unbiased variance), then y = relu(bn(x) + residual).
#include "cp_common.h"

namespace {

constexpr int SEG = 8192;          // floats per workgroup segment (256 threads x 8 float4)
constexpr int THREADS = 256;
constexpr int NV = SEG / (THREADS * 4);   // float4 per thread per segment

struct BnDims {
  int B, C;
  long long HW;
  int nseg;                        // segments per (b, c) plane
};

__device__ __forceinline__ void block_reduce2(double& a, double& b) {
  __shared__ double red[2][THREADS / 64];
  a = cp_wave_sum_d(a);
  b = cp_wave_sum_d(b);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { red[0][wid] = a; red[1][wid] = b; }
  __syncthreads();
  a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
}

// Sum of a channel's B * nseg partial pairs, by every workgroup of that channel in the same order (so all of them --
// and the backward pass, which reads what workgroup (0, c, 0) stores -- use bit-identical statistics).  Replaces the
// one-thread-per-channel finalize launches (two per BatchNorm and step, ~10 us each) by a 4 KB L2-resident read and
// one block reduction inside the apply passes.
__device__ __forceinline__ void channel_sums(const double* __restrict__ partial, const BnDims& d, int c, double& s,
                                             double& q) {
  const int n = d.B * d.nseg;
  const double* p = partial + 2 * (long long)c * n;
  s = 0;
  q = 0;
  for (int i = threadIdx.x; i < n; i += THREADS) {
    s += p[2 * i];
    q += p[2 * i + 1];
  }
  block_reduce2(s, q);
}

// grid = (nseg, C, B): partial[(c * B + b) * nseg + seg] = (sum, sum of squares)
__global__ __launch_bounds__(THREADS) void bn_stats_kernel(const float* __restrict__ x, BnDims d,
                                                           double* __restrict__ partial) {
  const int seg = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
  const float* p = x + ((long long)b * d.C + c) * d.HW;
  const long long i0 = (long long)seg * SEG, i1 = min(d.HW, i0 + SEG);
  float s = 0.f, q = 0.f;
  if ((d.HW & 3) == 0) {
    // all NV float4 loads of the segment are issued before the first use (a rolled loop waits out one HBM
    // latency per iteration)
    f32x4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const long long i = i0 + (k * THREADS + threadIdx.x) * 4;
      v[k] = i < i1 ? *reinterpret_cast<const f32x4*>(p + i) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
      q += (v[k][0] * v[k][0] + v[k][1] * v[k][1]) + (v[k][2] * v[k][2] + v[k][3] * v[k][3]);
    }
  } else {
    for (long long i = i0 + threadIdx.x; i < i1; i += THREADS) { const float v = p[i]; s += v; q += v * v; }
  }
  double ds = s, dq = q;
  block_reduce2(ds, dq);
  if (threadIdx.x == 0) {
    const long long o = ((long long)c * d.B + b) * d.nseg + seg;
    partial[2 * o] = ds;
    partial[2 * o + 1] = dq;
  }
}

// (the statistics are finalised here from the partial sums: mean / invstd, and by workgroup (0, c, 0) the saved and
//  running statistics)
__global__ __launch_bounds__(THREADS) void bn_apply_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ res,
                                                           float* __restrict__ y, BnDims d,
                                                           const double* __restrict__ partial, float eps,
                                                           float momentum, float* __restrict__ mean,
                                                           float* __restrict__ invstd,
                                                           float* __restrict__ running_mean,
                                                           float* __restrict__ running_var,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ bias, int relu) {
  const int seg = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
  const long long base = ((long long)b * d.C + c) * d.HW;
  double ds, dq;
  channel_sums(partial, d, c, ds, dq);
  const double N = (double)d.B * (double)d.HW;
  const double m = ds / N;
  double var = dq / N - m * m;
  if (var < 0) var = 0;
  const float meanf = (float)m, invf = (float)(1.0 / sqrt(var + (double)eps));
  if (seg == 0 && b == 0 && threadIdx.x == 0) {
    mean[c] = meanf;
    invstd[c] = invf;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
    if (running_var) {
      const double unbiased = N > 1 ? var * N / (N - 1) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
  const float sc = invf * (w ? w[c] : 1.f);
  const float sh = (bias ? bias[c] : 0.f) - meanf * sc;
  const long long i0 = (long long)seg * SEG, i1 = min(d.HW, i0 + SEG);
  if ((d.HW & 3) == 0) {
    f32x4 xv[NV], rv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const long long i = i0 + (j * THREADS + threadIdx.x) * 4;
      const bool ok = i < i1;
      xv[j] = ok ? *reinterpret_cast<const f32x4*>(x + base + i) : f32x4{0.f, 0.f, 0.f, 0.f};
      rv[j] = (ok && res) ? *reinterpret_cast<const f32x4*>(res + base + i) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const long long i = i0 + (j * THREADS + threadIdx.x) * 4;
      if (i >= i1) continue;
      f32x4 v = xv[j];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = v[k] * sc + sh;
        if (res) v[k] += rv[j][k];
        if (relu) v[k] = fmaxf(v[k], 0.f);
      }
      *reinterpret_cast<f32x4*>(y + base + i) = v;
    }
  } else {
    for (long long i = i0 + threadIdx.x; i < i1; i += THREADS) {
      float v = x[base + i] * sc + sh;
      if (res) v += res[base + i];
      y[base + i] = relu ? fmaxf(v, 0.f) : v;
    }
  }
}

// backward reduction: partial (sum g, sum g * xhat), g = gy * [y > 0] when relu
// RECOMP (ReLU without a residual): y is not read -- [y > 0] is the sign of x * sc + sh, the very expression (same
// operands, same contraction) the forward kernel evaluated, so the mask is bit-identical at a third less traffic.
template <bool RECOMP>
__global__ __launch_bounds__(THREADS) void bn_bwd_reduce_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ y,
                                                                const float* __restrict__ gy, BnDims d,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ w,
                                                                const float* __restrict__ bias,
                                                                int relu, double* __restrict__ partial) {
  const int seg = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
  const long long base = ((long long)b * d.C + c) * d.HW;
  const float m = mean[c], is = invstd[c];
  const float sc = is * (w ? w[c] : 1.f);
  const float sh = (bias ? bias[c] : 0.f) - m * sc;
  const long long i0 = (long long)seg * SEG, i1 = min(d.HW, i0 + SEG);
  float s = 0.f, q = 0.f;
  if ((d.HW & 3) == 0) {
    f32x4 xv[NV], gv[NV], yv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const long long i = i0 + (j * THREADS + threadIdx.x) * 4;
      const bool ok = i < i1;
      xv[j] = ok ? *reinterpret_cast<const f32x4*>(x + base + i) : f32x4{0.f, 0.f, 0.f, 0.f};
      gv[j] = ok ? *reinterpret_cast<const f32x4*>(gy + base + i) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (RECOMP) {
#pragma unroll
        for (int k = 0; k < 4; ++k) yv[j][k] = ok ? xv[j][k] * sc + sh : 0.f;
      } else {
        yv[j] = (ok && relu) ? *reinterpret_cast<const f32x4*>(y + base + i) : f32x4{1.f, 1.f, 1.f, 1.f};
      }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float g = yv[j][k] > 0.f ? gv[j][k] : 0.f;
        s += g;
        q += g * ((xv[j][k] - m) * is);
      }
    }
  } else {
    for (long long i = i0 + threadIdx.x; i < i1; i += THREADS) {
      float g = gy[base + i];
      if (relu && !((RECOMP ? x[base + i] * sc + sh : y[base + i]) > 0.f)) g = 0.f;
      s += g;
      q += g * ((x[base + i] - m) * is);
    }
  }
  double ds = s, dq = q;
  block_reduce2(ds, dq);
  if (threadIdx.x == 0) {
    const long long o = ((long long)c * d.B + b) * d.nseg + seg;
    partial[2 * o] = ds;
    partial[2 * o + 1] = dq;
  }
}

// dx = w * invstd * (g - dbeta/N - xhat * dgamma/N);  dres = g
template <bool RECOMP>
__global__ __launch_bounds__(THREADS) void bn_bwd_apply_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gy, BnDims d,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ w,
    const float* __restrict__ bias, const double* __restrict__ partial, float* __restrict__ grad_w,
    float* __restrict__ grad_b, int relu, float* __restrict__ gx, float* __restrict__ gres) {
  const int seg = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
  const long long base = ((long long)b * d.C + c) * d.HW;
  const float m = mean[c], is = invstd[c];
  const float sc = is * (w ? w[c] : 1.f);
  const float sh = (bias ? bias[c] : 0.f) - m * sc;
  double ds, dq;                                     // (sum g, sum g * xhat) of the channel, finalised here
  channel_sums(partial, d, c, ds, dq);
  if (seg == 0 && b == 0 && threadIdx.x == 0) {
    if (grad_b) grad_b[c] = (float)ds;
    if (grad_w) grad_w[c] = (float)dq;
  }
  const double N = (double)d.B * (double)d.HW;
  const float k0 = (w ? w[c] : 1.f) * is, db = (float)(ds / N), dg = (float)(dq / N);
  const long long i0 = (long long)seg * SEG, i1 = min(d.HW, i0 + SEG);
  if ((d.HW & 3) == 0) {
    f32x4 xv[NV], gv[NV], yv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const long long i = i0 + (j * THREADS + threadIdx.x) * 4;
      const bool ok = i < i1;
      xv[j] = ok ? *reinterpret_cast<const f32x4*>(x + base + i) : f32x4{0.f, 0.f, 0.f, 0.f};
      gv[j] = ok ? *reinterpret_cast<const f32x4*>(gy + base + i) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (RECOMP) {
#pragma unroll
        for (int k = 0; k < 4; ++k) yv[j][k] = ok ? xv[j][k] * sc + sh : 0.f;
      } else {
        yv[j] = (ok && relu) ? *reinterpret_cast<const f32x4*>(y + base + i) : f32x4{1.f, 1.f, 1.f, 1.f};
      }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const long long i = i0 + (j * THREADS + threadIdx.x) * 4;
      if (i >= i1) continue;
      f32x4 g, o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        g[k] = yv[j][k] > 0.f ? gv[j][k] : 0.f;
        o[k] = k0 * (g[k] - db - (xv[j][k] - m) * is * dg);
      }
      *reinterpret_cast<f32x4*>(gx + base + i) = o;
      if (gres) *reinterpret_cast<f32x4*>(gres + base + i) = g;
    }
  } else {
    for (long long i = i0 + threadIdx.x; i < i1; i += THREADS) {
      float g = gy[base + i];
      if (relu && !((RECOMP ? x[base + i] * sc + sh : y[base + i]) > 0.f)) g = 0.f;
      gx[base + i] = k0 * (g - db - (x[base + i] - m) * is * dg);
      if (gres) gres[base + i] = g;
    }
  }
}

int make_dims(BnDims& d, int32_t B, int32_t C, int64_t HW) {
  CP_CHECK_ARG(B > 0 && C > 0 && HW > 0);
  if (B > 65535 || C > 65535) return CP_EUNSUPPORTED;
  d.B = B; d.C = C; d.HW = HW;
  d.nseg = (int)((HW + SEG - 1) / SEG);
  return CP_OK;
}

}  // namespace

extern "C" size_t cp_bn_workspace_bytes(int32_t B, int32_t C, int64_t HW) {
  if (B <= 0 || C <= 0 || HW <= 0) return 0;
  const size_t nseg = (size_t)((HW + SEG - 1) / SEG);
  return (size_t)C * B * nseg * 2 * sizeof(double) + (size_t)C * 2 * sizeof(float);
}

extern "C" int cp_bn_act_forward_train(const float* x, const float* weight, const float* bias,
                                       const float* residual, float* y, float* save_mean,
                                       float* save_invstd, float* running_mean, float* running_var,
                                       float momentum, float eps, int32_t relu, int32_t B, int32_t C,
                                       int64_t HW, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  BnDims d;
  const int rc = make_dims(d, B, C, HW);
  if (rc != CP_OK) return rc;
  CP_CHECK_ARG(x && y && save_mean && save_invstd && workspace);
  if (workspace_bytes < cp_bn_workspace_bytes(B, C, HW)) return CP_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)workspace;
  const dim3 grid(d.nseg, C, B);
  hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(THREADS), 0, st, x, d, partial);
  hipLaunchKernelGGL(bn_apply_kernel, grid, dim3(THREADS), 0, st, x, residual, y, d, partial, eps, momentum,
                     save_mean, save_invstd, running_mean, running_var, weight, bias, relu);
  return cp_launch_status();
}

extern "C" int cp_bn_act_backward(const float* x, const float* y, const float* grad_y,
                                  const float* weight, const float* bias, const float* save_mean,
                                  const float* save_invstd, int32_t relu, float* grad_x,
                                  float* grad_residual, float* grad_weight, float* grad_bias,
                                  int32_t B, int32_t C, int64_t HW, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  BnDims d;
  const int rc = make_dims(d, B, C, HW);
  if (rc != CP_OK) return rc;
  // y == NULL with ReLU (forwards WITHOUT a residual only): the mask is recomputed from x, weight and `bias` (the
  // forward's, NULL if it had none) -- y is never read
  const bool recomp = relu && !y;
  CP_CHECK_ARG(x && grad_y && save_mean && save_invstd && grad_x && workspace && !(recomp && grad_residual));
  if (workspace_bytes < cp_bn_workspace_bytes(B, C, HW)) return CP_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)workspace;
  const dim3 grid(d.nseg, C, B);
  if (recomp) {
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, grid, dim3(THREADS), 0, st, x, y, grad_y, d, save_mean, save_invstd,
                       weight, bias, relu, partial);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, grid, dim3(THREADS), 0, st, x, y, grad_y, d, save_mean, save_invstd,
                       weight, bias, partial, grad_weight, grad_bias, relu, grad_x, grad_residual);
  } else {
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, grid, dim3(THREADS), 0, st, x, y, grad_y, d, save_mean, save_invstd,
                       weight, bias, relu, partial);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, grid, dim3(THREADS), 0, st, x, y, grad_y, d, save_mean, save_invstd,
                       weight, bias, partial, grad_weight, grad_bias, relu, grad_x, grad_residual);
  }
  return cp_launch_status();
}
