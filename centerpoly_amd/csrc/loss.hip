// Fused sigmoid-clamp + CornerNet focal loss, and gathered L1 losses, for gfx950.
//
// Replaces _sigmoid (src/lib/models/utils.py:8-10), _neg_loss / FocalLoss
// (src/lib/models/losses.py:146-171, 792-799), RegL1Loss (:817-830) and the
// regression part of PolyLoss (:910-949) with the NHWC permute-copy of
// _transpose_and_gather_feat (src/lib/models/utils.py:22-26) replaced by direct
// strided gathers.  Both are HBM/latency bound: one streaming pass each way for
// the focal term (float4 per lane), a handful of scattered reads for the rest.
#include "cp_common.h"

namespace {

constexpr int FOCAL_THREADS = 256;
constexpr int FOCAL_MAX_BLOCKS = 2048;
constexpr float CLAMP_LO = 1e-4f;
constexpr float CLAMP_HI = (float)(1 - 1e-4);

__device__ __forceinline__ void focal_elem(float x, float g, float& p_out, float& pos, float& neg,
                                           float& npos) {
  float p = 1.f / (1.f + expf(-x));
  p = fminf(fmaxf(p, CLAMP_LO), CLAMP_HI);
  p_out = p;
  if (g == 1.f) {
    const float q = 1.f - p;
    pos += logf(p) * q * q;
    npos += 1.f;
  } else if (g < 1.f) {
    const float w = (1.f - g) * (1.f - g);
    neg += logf(1.f - p) * p * p * (w * w);
  }
}

__global__ __launch_bounds__(FOCAL_THREADS) void focal_fwd_kernel(float* __restrict__ hm,
                                                                  const float* __restrict__ gt,
                                                                  long long n,
                                                                  double* __restrict__ partial) {
  float pos = 0.f, neg = 0.f, npos = 0.f;
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * FOCAL_THREADS;
  f32x4* hm4 = reinterpret_cast<f32x4*>(hm);
  const f32x4* gt4 = reinterpret_cast<const f32x4*>(gt);
  for (long long i = (long long)blockIdx.x * FOCAL_THREADS + threadIdx.x; i < n4; i += stride) {
    f32x4 x = hm4[i];
    const f32x4 g = gt4[i];
    float p0, p1, p2, p3;
    focal_elem(x[0], g[0], p0, pos, neg, npos);
    focal_elem(x[1], g[1], p1, pos, neg, npos);
    focal_elem(x[2], g[2], p2, pos, neg, npos);
    focal_elem(x[3], g[3], p3, pos, neg, npos);
    hm4[i] = f32x4{p0, p1, p2, p3};
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {      // tail
    const long long i = (n4 << 2) + threadIdx.x;
    float p;
    focal_elem(hm[i], gt[i], p, pos, neg, npos);
    hm[i] = p;
  }
  __shared__ double red[3][FOCAL_THREADS / 64];
  double dp = cp_wave_sum_d((double)pos), dn = cp_wave_sum_d((double)neg),
         dc = cp_wave_sum_d((double)npos);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wid] = dp;
    red[1][wid] = dn;
    red[2][wid] = dc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0, c = 0;
    for (int w = 0; w < FOCAL_THREADS / 64; ++w) {
      a += red[0][w];
      b += red[1][w];
      c += red[2][w];
    }
    partial[blockIdx.x * 3 + 0] = a;
    partial[blockIdx.x * 3 + 1] = b;
    partial[blockIdx.x * 3 + 2] = c;
  }
}

__global__ __launch_bounds__(256) void focal_finalize_kernel(const double* __restrict__ partial,
                                                             int nblocks,
                                                             float* __restrict__ loss_out,
                                                             float* __restrict__ stats_out) {
  double a = 0, b = 0, c = 0;
  for (int i = threadIdx.x; i < nblocks; i += 256) {
    a += partial[i * 3 + 0];
    b += partial[i * 3 + 1];
    c += partial[i * 3 + 2];
  }
  __shared__ double red[3][4];
  a = cp_wave_sum_d(a);
  b = cp_wave_sum_d(b);
  c = cp_wave_sum_d(c);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wid] = a;
    red[1][wid] = b;
    red[2][wid] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    c = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    // losses.py:167-170: -neg if num_pos == 0 else -(pos+neg)/num_pos
    const double loss = (c == 0) ? -b : -(a + b) / c;
    loss_out[0] = (float)loss;
    if (stats_out) {
      stats_out[0] = (float)a;
      stats_out[1] = (float)b;
      stats_out[2] = (float)c;
    }
  }
}

__device__ __forceinline__ float focal_grad_elem(float p, float g, float scale) {
  // clamp passes gradient only strictly inside (boundary equality is measure zero)
  if (!(p > CLAMP_LO && p < CLAMP_HI)) return 0.f;
  const float q = 1.f - p;
  float dldp;
  if (g == 1.f) dldp = q * q / p - 2.f * q * logf(p);
  else if (g < 1.f) {
    const float w = (1.f - g) * (1.f - g);
    dldp = (w * w) * (2.f * p * logf(q) - p * p / q);
  } else return 0.f;
  return scale * dldp * p * q;
}

__global__ __launch_bounds__(FOCAL_THREADS) void focal_bwd_kernel(
    const float* __restrict__ hm, const float* __restrict__ gt, long long n,
    const float* __restrict__ stats, const float* __restrict__ grad_loss,
    float* __restrict__ grad) {
  const float npos = stats[2];
  const float scale = -grad_loss[0] / (npos == 0.f ? 1.f : npos);
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * FOCAL_THREADS;
  const f32x4* hm4 = reinterpret_cast<const f32x4*>(hm);
  const f32x4* gt4 = reinterpret_cast<const f32x4*>(gt);
  f32x4* gr4 = reinterpret_cast<f32x4*>(grad);
  for (long long i = (long long)blockIdx.x * FOCAL_THREADS + threadIdx.x; i < n4; i += stride) {
    const f32x4 p = hm4[i], g = gt4[i];
    f32x4 o;
    o[0] = focal_grad_elem(p[0], g[0], scale);
    o[1] = focal_grad_elem(p[1], g[1], scale);
    o[2] = focal_grad_elem(p[2], g[2], scale);
    o[3] = focal_grad_elem(p[3], g[3], scale);
    gr4[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    grad[i] = focal_grad_elem(hm[i], gt[i], scale);
  }
}

inline int focal_blocks(long long n) {
  long long b = ((n >> 2) + FOCAL_THREADS - 1) / FOCAL_THREADS;
  if (b < 1) b = 1;
  if (b > FOCAL_MAX_BLOCKS) b = FOCAL_MAX_BLOCKS;
  return (int)b;
}

// ------------------------------------------------------------ gathered L1 ---
struct GatherArgs {
  const float* feat;
  const long long* ind;
  const unsigned char* mask;
  const float* target;
  const float* pred_add;
  int B, D, HW, M, mode;
  float eps;
};

__device__ __forceinline__ float gl1_term(int mode, int d, float p, float t) {
  const float diff = p - t;
  switch (mode) {
    case CP_L1_POLAR:
      return (d & 1) ? 1.f - cosf(diff) : fabsf(diff);
    case CP_L1_POLAR_FIXED:
      return (d & 1) ? 0.f : fabsf(diff);
    case CP_L1_RELU20: {
      const float a = fabsf(diff);
      return a >= 20.f ? a : 0.f;
    }
    case CP_L1_SMOOTH: {                     // smooth_l1 (beta 1) of RegLoss / _reg_loss, losses.py:201-216
      const float a = fabsf(diff);
      return a < 1.f ? 0.5f * diff * diff : a - 0.5f;
    }
    default:
      return fabsf(diff);
  }
}

__device__ __forceinline__ float gl1_dterm(int mode, int d, float p, float t) {
  const float diff = p - t;
  const float sgn = (diff > 0.f) ? 1.f : (diff < 0.f ? -1.f : 0.f);
  switch (mode) {
    case CP_L1_POLAR:
      return (d & 1) ? sinf(diff) : sgn;
    case CP_L1_POLAR_FIXED:
      return (d & 1) ? 0.f : sgn;
    case CP_L1_RELU20:
      return fabsf(diff) >= 20.f ? sgn : 0.f;
    case CP_L1_SMOOTH:
      return fabsf(diff) < 1.f ? diff : sgn;
    default:
      return sgn;
  }
}

// One 1024-thread workgroup: deterministic, no workspace.  Work = (#masked objects) x D
// scattered reads; tens of thousands at most.
__global__ __launch_bounds__(1024) void gather_l1_fwd_kernel(GatherArgs a, float* loss_out) {
  __shared__ double red[16];
  __shared__ int cnt_red[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  double s = 0;
  int cnt = 0;
  const int BM = a.B * a.M;
  for (int o = wid; o < BM; o += 16) {       // one wave per object: lanes over D
    if (!a.mask[o]) continue;
    if (lane == 0) ++cnt;
    const int b = o / a.M;
    const long long sp = a.ind[o];
    for (int d = lane; d < a.D; d += 64) {
      float p = a.feat[((long long)b * a.D + d) * a.HW + sp];
      if (a.pred_add) p += a.pred_add[(long long)o * a.D + d];
      s += (double)gl1_term(a.mode, d, p, a.target[(long long)o * a.D + d]);
    }
  }
  s = cp_wave_sum_d(s);
  if (lane == 0) {
    red[wid] = s;
    cnt_red[wid] = cnt;
  }
  __syncthreads();
  if (tid == 0) {
    double t = 0;
    int c = 0;
    for (int w = 0; w < 16; ++w) {
      t += red[w];
      c += cnt_red[w];
    }
    // RegL1Loss / PolyLoss divide by the EXPANDED mask sum (objects x D), RegLoss by the object count
    const float denom = (float)(a.mode == CP_L1_SMOOTH ? (long long)c : (long long)c * a.D) + a.eps;
    loss_out[0] = (float)t / denom;
  }
}

__global__ __launch_bounds__(256) void gather_l1_bwd_kernel(GatherArgs a, const float* grad_loss,
                                                            float* grad_feat) {
  __shared__ int cnt_sh;
  // mask count (needed for the denominator) -- B*M <= a few thousand bytes
  if (threadIdx.x == 0) cnt_sh = 0;
  __syncthreads();
  int c = 0;
  const int BM = a.B * a.M;
  for (int o = threadIdx.x; o < BM; o += 256) c += a.mask[o] ? 1 : 0;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(&cnt_sh, c);
  __syncthreads();
  const float denom = (float)(a.mode == CP_L1_SMOOTH ? (long long)cnt_sh : (long long)cnt_sh * a.D) + a.eps;
  const float g = grad_loss[0] / denom;
  const long long total = (long long)BM * a.D;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total;
       q += (long long)gridDim.x * 256) {
    const int o = (int)(q / a.D), d = (int)(q - (long long)o * a.D);
    if (!a.mask[o]) continue;
    const int b = o / a.M;
    const long long sp = a.ind[o];
    const long long fi = ((long long)b * a.D + d) * a.HW + sp;
    float p = a.feat[fi];
    if (a.pred_add) p += a.pred_add[q];
    const float dv = gl1_dterm(a.mode, d, p, a.target[q]);
    if (dv != 0.f) atomicAdd(&grad_feat[fi], g * dv);
  }
}

}  // namespace

extern "C" size_t cp_sigmoid_focal_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return (size_t)FOCAL_MAX_BLOCKS * 3 * sizeof(double);
}

extern "C" int cp_sigmoid_focal_forward(float* hm_inout, const float* gt, int64_t n,
                                        float* loss_out, float* stats_out, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  CP_CHECK_ARG(hm_inout && gt && loss_out && workspace && n > 0);
  CP_CHECK_ARG(((uintptr_t)hm_inout & 15) == 0 && ((uintptr_t)gt & 15) == 0);
  if (workspace_bytes < cp_sigmoid_focal_workspace_bytes(n)) return CP_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int nb = focal_blocks(n);
  hipLaunchKernelGGL(focal_fwd_kernel, dim3(nb), dim3(FOCAL_THREADS), 0, st, hm_inout, gt,
                     (long long)n, (double*)workspace);
  hipLaunchKernelGGL(focal_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)workspace,
                     nb, loss_out, stats_out);
  return cp_launch_status();
}

extern "C" int cp_sigmoid_focal_backward(const float* hm_act, const float* gt, int64_t n,
                                         const float* stats, const float* grad_loss,
                                         float* grad_logits, void* stream) {
  CP_CHECK_ARG(hm_act && gt && stats && grad_loss && grad_logits && n > 0);
  CP_CHECK_ARG(((uintptr_t)hm_act & 15) == 0 && ((uintptr_t)gt & 15) == 0 &&
               ((uintptr_t)grad_logits & 15) == 0);
  hipLaunchKernelGGL(focal_bwd_kernel, dim3(focal_blocks(n)), dim3(FOCAL_THREADS), 0,
                     (hipStream_t)stream, hm_act, gt, (long long)n, stats, grad_loss, grad_logits);
  return cp_launch_status();
}

static int fill_gather(GatherArgs& a, const float* feat, const int64_t* ind, const uint8_t* mask,
                       const float* target, const float* pred_add, int32_t B, int32_t D,
                       int32_t H, int32_t W, int32_t M, int32_t mode, float eps) {
  CP_CHECK_ARG(feat && ind && mask && target);
  CP_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0 && M > 0);
  CP_CHECK_ARG(mode >= CP_L1_PLAIN && mode <= CP_L1_SMOOTH);
  if ((long long)H * W >= (1ll << 31) || (long long)B * M >= (1ll << 24)) return CP_EUNSUPPORTED;
  a.feat = feat; a.ind = (const long long*)ind; a.mask = mask; a.target = target;
  a.pred_add = pred_add; a.B = B; a.D = D; a.HW = H * W; a.M = M; a.mode = mode; a.eps = eps;
  return CP_OK;
}

extern "C" int cp_gather_l1_forward(const float* feat, const int64_t* ind, const uint8_t* mask,
                                    const float* target, const float* pred_add, int32_t B,
                                    int32_t D, int32_t H, int32_t W, int32_t M, int32_t mode,
                                    float eps, float* loss_out, void* stream) {
  GatherArgs a;
  int rc = fill_gather(a, feat, ind, mask, target, pred_add, B, D, H, W, M, mode, eps);
  if (rc != CP_OK) return rc;
  CP_CHECK_ARG(loss_out);
  hipLaunchKernelGGL(gather_l1_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a,
                     loss_out);
  return cp_launch_status();
}

extern "C" int cp_gather_l1_backward(const float* feat, const int64_t* ind, const uint8_t* mask,
                                     const float* target, const float* pred_add, int32_t B,
                                     int32_t D, int32_t H, int32_t W, int32_t M, int32_t mode,
                                     float eps, const float* grad_loss, float* grad_feat,
                                     void* stream) {
  GatherArgs a;
  int rc = fill_gather(a, feat, ind, mask, target, pred_add, B, D, H, W, M, mode, eps);
  if (rc != CP_OK) return rc;
  CP_CHECK_ARG(grad_loss && grad_feat);
  const long long total = (long long)B * M * D;
  int nb = (int)((total + 255) / 256);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(gather_l1_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a,
                     grad_loss, grad_feat);
  return cp_launch_status();
}


// ------------------------------------------------------------------ MSE heat-map loss ---
// `--mse_loss`: crit = torch.nn.MSELoss() on the RAW heat-map head (no sigmoid), trains/polydet.py:23,44-46,84:
// loss = mean((x - gt)^2).  One streaming pass each way; double block partials, fixed-order final sum.
namespace {
constexpr int MSE_BLOCKS = 1024;

__global__ __launch_bounds__(256) void mse_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gt,
                                                      long long n, double* __restrict__ part) {
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float d = x[i] - gt[i];
    s += (double)(d * d);
  }
  s = cp_wave_sum_d(s);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void mse_final_kernel(const double* __restrict__ part, int nparts, long long n,
                                                        float* __restrict__ loss) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
  s = cp_wave_sum_d(s);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (float)((red[0] + red[1] + red[2] + red[3]) / (double)n);
}

__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gt,
                                                      long long n, const float* __restrict__ grad_loss,
                                                      float* __restrict__ grad) {
  const float g = 2.f * grad_loss[0] / (float)n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    grad[i] = g * (x[i] - gt[i]);
}
}  // namespace

extern "C" size_t cp_mse_workspace_bytes(void) { return MSE_BLOCKS * sizeof(double); }

extern "C" int cp_mse_forward(const float* x, const float* gt, int64_t n, float* loss_out, void* workspace,
                              size_t workspace_bytes, void* stream) {
  CP_CHECK_ARG(x && gt && loss_out && n > 0 && workspace && workspace_bytes >= cp_mse_workspace_bytes());
  long long nb = (n + 255) / 256;
  if (nb > MSE_BLOCKS) nb = MSE_BLOCKS;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mse_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, gt, (long long)n, (double*)workspace);
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, st, (const double*)workspace, (int)nb, (long long)n, loss_out);
  return cp_launch_status();
}

extern "C" int cp_mse_backward(const float* x, const float* gt, int64_t n, const float* grad_loss, float* grad_x,
                               void* stream) {
  CP_CHECK_ARG(x && gt && grad_loss && grad_x && n > 0);
  long long nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, gt, (long long)n,
                     grad_loss, grad_x);
  return cp_launch_status();
}


// ------------------------------------------------------------------ dense masked L1 (--dense_poly) ---
// trains/polydet.py:107-110: mask_weight = dense_poly_mask.sum() + 1e-4;
//   poly_loss = L1Loss(reduction='sum')(output['poly'] * mask, dense_poly * mask) / mask_weight
// One streaming pass each way over the [B, 2N, h, w] maps; double block partials (|diff| and mask), fixed-order final sum.
namespace {
__global__ __launch_bounds__(256) void dense_l1_fwd_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                           const float* __restrict__ m, long long n,
                                                           double* __restrict__ part) {
  double s = 0.0, sm = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float mk = m[i];
    s += (double)fabsf(p[i] * mk - t[i] * mk);          // (the reference multiplies both operands by the mask first)
    sm += (double)mk;
  }
  s = cp_wave_sum_d(s);
  sm = cp_wave_sum_d(sm);
  __shared__ double red[8];
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = s;
    red[4 + (threadIdx.x >> 6)] = sm;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    part[2 * blockIdx.x + 1] = red[4] + red[5] + red[6] + red[7];
  }
}

__global__ __launch_bounds__(256) void dense_l1_final_kernel(const double* __restrict__ part, int nparts, float eps,
                                                             float* __restrict__ out2) {
  double s = 0.0, sm = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) {
    s += part[2 * i];
    sm += part[2 * i + 1];
  }
  s = cp_wave_sum_d(s);
  sm = cp_wave_sum_d(sm);
  __shared__ double red[8];
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = s;
    red[4 + (threadIdx.x >> 6)] = sm;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float den = (float)(red[4] + red[5] + red[6] + red[7]) + eps;     // float32 sum + 1e-4, as the tensor expression
    out2[1] = den;
    out2[0] = (float)(red[0] + red[1] + red[2] + red[3]) / den;
  }
}

__global__ __launch_bounds__(256) void dense_l1_bwd_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                           const float* __restrict__ m, long long n,
                                                           const float* __restrict__ den,
                                                           const float* __restrict__ grad_loss, float* __restrict__ g) {
  const float sc = grad_loss[0] / den[0];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float mk = m[i];
    const float d = p[i] * mk - t[i] * mk;
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);                // torch's sign(0) = 0
    g[i] = sg * mk * sc;
  }
}
}  // namespace

extern "C" size_t cp_dense_l1_workspace_bytes(void) { return 2 * MSE_BLOCKS * sizeof(double); }

extern "C" int cp_dense_l1_forward(const float* pred, const float* target, const float* mask, int64_t n, float eps,
                                   float* out2, void* workspace, size_t workspace_bytes, void* stream) {
  CP_CHECK_ARG(pred && target && mask && out2 && n > 0 && workspace && workspace_bytes >= cp_dense_l1_workspace_bytes());
  long long nb = (n + 255) / 256;
  if (nb > MSE_BLOCKS) nb = MSE_BLOCKS;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(dense_l1_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, st, pred, target, mask, (long long)n,
                     (double*)workspace);
  hipLaunchKernelGGL(dense_l1_final_kernel, dim3(1), dim3(256), 0, st, (const double*)workspace, (int)nb, eps, out2);
  return cp_launch_status();
}

extern "C" int cp_dense_l1_backward(const float* pred, const float* target, const float* mask, int64_t n,
                                    const float* den, const float* grad_loss, float* grad_pred, void* stream) {
  CP_CHECK_ARG(pred && target && mask && den && grad_loss && grad_pred && n > 0);
  long long nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(dense_l1_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, pred, target, mask,
                     (long long)n, den, grad_loss, grad_pred);
  return cp_launch_status();
}
