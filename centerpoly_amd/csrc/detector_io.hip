// The two host stages either side of the network in the detector, moved to the device.
//
// (1) Pre-processing: affine warp of the 8-bit HWC image + normalisation to the CHW fp32
// network input, one kernel, one pass over the output.
//
// Replaces the host cv2 stage of BaseDetector.pre_process
// (reference: src/lib/detectors/base_detector.py:66-87):
//     cv2.warpAffine(image, trans_input, (inp_w, inp_h), flags=cv2.INTER_LINEAR)
//     ((inp / 255. - mean) / std).astype(float32).transpose(2, 0, 1)   [+ the flipped copy]
// The arithmetic is OpenCV's for 8-bit images (imgwarp.cpp: cv::warpAffine + remapBilinear):
// float64 inversion of the forward matrix, source coordinates in fixed point with 10
// fractional bits rounded to 5, 15-bit integer bilinear weights, constant border 0, result
// rounded to uint8 -- then the numpy normalisation in float64, cast to fp32 once.  Integer
// work throughout, so the output is bit-identical to oracle/pre.py.
//
// HBM-bound: reads 3 B/px of the source window once (neighbouring lanes share the 2x2 taps
// through L1), writes 12 B/px (24 with the flipped copy), coalesced per channel plane.
//
// (2) Post-processing: polydet_post_process's transform_preds (reference:
// src/lib/utils/post_process.py:105-122, src/lib/utils/image.py:19-24,62-65) -- the inverse
// affine applied in float64 to the two box corners and the N polygon vertices of every decoded
// row, cast to fp32, then the `/ scale` of PolydetDetector.post_process
// (src/lib/detectors/polydet.py:52-57).  One thread per (row, point); score, class and depth
// columns are copied.  The per-class split stays on the host (it builds Python dicts).
#include "cp_common.h"

namespace {

struct PreArgs {
  const uint8_t* src;
  float* out;
  double m[6];            // inverse map dst -> src
  double mean[3], stdv[3];
  int sh, sw, dh, dw, flip;
};

__device__ __forceinline__ long long round_fix(double v) {   // cvRound(v * 2^10), saturated to int
  const double s = v * 1024.0;
  if (s >= 2147483647.0) return 2147483647ll;
  if (s <= -2147483648.0) return -2147483648ll;
  return (long long)__double2int_rn(s);                      // round half to even
}

__global__ __launch_bounds__(256) void preprocess_kernel(PreArgs a) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= a.dw) return;
  // explicit rn ops: the compiler must not contract M1*y + M2 into an fma
  const long long X0 = round_fix(__dadd_rn(__dmul_rn(a.m[1], (double)y), a.m[2])) + 16;
  const long long Y0 = round_fix(__dadd_rn(__dmul_rn(a.m[4], (double)y), a.m[5])) + 16;
  const long long X = (X0 + round_fix(__dmul_rn(a.m[0], (double)x))) >> 5;
  const long long Y = (Y0 + round_fix(__dmul_rn(a.m[3], (double)x))) >> 5;
  const int sx = (int)min(max(X >> 5, -32768ll), 32767ll);   // saturate_cast<short>
  const int sy = (int)min(max(Y >> 5, -32768ll), 32767ll);
  const int fx = (int)(X & 31), fy = (int)(Y & 31);
  const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32;
  const int w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
  const bool y0 = sy >= 0 && sy < a.sh, y1 = sy + 1 >= 0 && sy + 1 < a.sh;
  const bool x0 = sx >= 0 && sx < a.sw, x1 = sx + 1 >= 0 && sx + 1 < a.sw;
  const uint8_t* r0 = a.src + ((long long)(y0 ? sy : 0) * a.sw) * 3;
  const uint8_t* r1 = a.src + ((long long)(y1 ? sy + 1 : 0) * a.sw) * 3;
  const int c0 = (x0 ? sx : 0) * 3, c1 = (x1 ? sx + 1 : 0) * 3;
  const long long plane = (long long)a.dh * a.dw;
  float* o = a.out + (long long)y * a.dw + x;
  float* of = a.out + 3 * plane + (long long)y * a.dw + (a.dw - 1 - x);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int p00 = (y0 && x0) ? r0[c0 + c] : 0, p01 = (y0 && x1) ? r0[c1 + c] : 0;
    const int p10 = (y1 && x0) ? r1[c0 + c] : 0, p11 = (y1 && x1) ? r1[c1 + c] : 0;
    const int v = (w00 * p00 + w01 * p01 + w10 * p10 + w11 * p11 + (1 << 14)) >> 15;   // <= 255
    const float f = (float)__ddiv_rn(__dsub_rn(__ddiv_rn((double)v, 255.0), a.mean[c]), a.stdv[c]);
    o[c * plane] = f;
    if (a.flip) of[c * plane] = f;
  }
}

struct PostArgs {
  const float* dets;
  float* out;
  const double* trans;    // device [B][6]
  float scale;
  int rows_per_image, ncols, npts;   // npts = 2 box corners + N vertices
  long long total;        // B * rows_per_image * ncols
};

__global__ __launch_bounds__(256) void postprocess_kernel(PostArgs a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.total) return;
  const long long row = i / a.ncols;
  const int col = (int)(i - row * a.ncols);
  const float* d = a.dets + row * a.ncols;
  // columns: 0-3 box corners, 4 score, 5 class, 6 .. ncols-2 vertices, ncols-1 depth
  const bool is_pt = col < 4 || (col >= 6 && col < a.ncols - 1);
  if (!is_pt) {
    a.out[i] = d[col];
    return;
  }
  const int xcol = col < 4 ? (col & ~1) : 6 + ((col - 6) & ~1);
  const bool is_y = col < 4 ? (col & 1) : ((col - 6) & 1);
  const double* t = a.trans + (row / a.rows_per_image) * 6 + (is_y ? 3 : 0);
  const double v = __dadd_rn(__dadd_rn(__dmul_rn(t[0], (double)d[xcol]), __dmul_rn(t[1], (double)d[xcol + 1])), t[2]);
  a.out[i] = (float)v / a.scale;
}

}  // namespace

extern "C" int cp_polydet_post_process(const float* dets, const double* trans_dev, float scale,
                                       int32_t B, int32_t K, int32_t ncols, float* out,
                                       void* stream) {
  CP_CHECK_ARG(dets && trans_dev && out && B > 0 && K > 0);
  CP_CHECK_ARG(ncols >= 9 && ((ncols - 7) & 1) == 0 && scale > 0.f);
  PostArgs a;
  a.dets = dets; a.out = out; a.trans = trans_dev; a.scale = scale;
  a.rows_per_image = K; a.ncols = ncols; a.npts = 2 + (ncols - 7) / 2;
  a.total = (long long)B * K * ncols;
  hipLaunchKernelGGL(postprocess_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, a);
  return cp_launch_status();
}

extern "C" int cp_preprocess_warp_normalize(const uint8_t* src, int32_t src_h, int32_t src_w,
                                            const double* trans, const float* mean,
                                            const float* stdv, int32_t dst_h, int32_t dst_w,
                                            int32_t flip_copy, float* out, void* stream) {
  CP_CHECK_ARG(src && trans && mean && stdv && out);
  CP_CHECK_ARG(src_h > 0 && src_w > 0 && dst_h > 0 && dst_w > 0);
  if (src_h > 32767 || src_w > 32767 || dst_h > 65535) return CP_EUNSUPPORTED;
  PreArgs a;
  a.src = src; a.out = out;
  // cv::warpAffine inverts the forward map in place, in float64
  double M[6];
  for (int i = 0; i < 6; ++i) M[i] = trans[i];
  double D = M[0] * M[4] - M[1] * M[3];
  D = D != 0 ? 1. / D : 0;
  const double A11 = M[4] * D, A22 = M[0] * D;
  M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
  const double b1 = -M[0] * M[2] - M[1] * M[5];
  const double b2 = -M[3] * M[2] - M[4] * M[5];
  M[2] = b1; M[5] = b2;
  for (int i = 0; i < 6; ++i) a.m[i] = M[i];
  for (int c = 0; c < 3; ++c) { a.mean[c] = (double)mean[c]; a.stdv[c] = (double)stdv[c]; }
  a.sh = src_h; a.sw = src_w; a.dh = dst_h; a.dw = dst_w; a.flip = flip_copy ? 1 : 0;
  hipLaunchKernelGGL(preprocess_kernel, dim3((dst_w + 255) / 256, dst_h), dim3(256), 0,
                     (hipStream_t)stream, a);
  return cp_launch_status();
}

// ------------------------------------------------------ training-input colour augmentation ---
// color_aug + normalise of the training sampler (src/lib/datasets/sample/polydet.py:128-136,
// src/lib/utils/image.py:231-264) on the warped input, in place on BGR planes holding x / 255:
//   gs = grey(image), gs_mean = mean(gs)                       (taken ONCE, before any op)
//   three ops in a random order: brightness  x *= a
//                                contrast    x = x * a + gs_mean * (1 - a)
//                                saturation  x = x * a + gs * (1 - a)
//   lighting   x += eig_vec . (eig_val * alpha)   (a per-channel constant, added in float64)
//   normalise  x = (x - mean) / std
// float32 operations in numpy's order, no fma contraction.  Two kernels: grey-level sum (double
// partials), then one streaming pass.
namespace {
constexpr int CA_BLOCKS = 1024;

__global__ __launch_bounds__(256) void color_gs_sum_kernel(const float* __restrict__ img, long long HW,
                                                           double* __restrict__ part) {
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long long)gridDim.x * 256) {
    const float g = __fadd_rn(__fadd_rn(__fmul_rn(img[i], 0.114f), __fmul_rn(img[HW + i], 0.587f)),
                              __fmul_rn(img[2 * HW + i], 0.299f));
    s += (double)g;
  }
  s = cp_wave_sum_d(s);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

struct ColorArgs {
  float* img;
  long long HW;
  const double* part;
  int nparts;
  int order[3];
  float alpha[3];
  double light[3];
  float mean[3], stdv[3];
  int color_on;
};

__global__ __launch_bounds__(256) void color_apply_kernel(ColorArgs a) {
  __shared__ float s_mean;
  if (a.color_on) {
    double s = 0.0;
    for (int i = threadIdx.x; i < a.nparts; i += 256) s += a.part[i];
    s = cp_wave_sum_d(s);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) s_mean = (float)((red[0] + red[1] + red[2] + red[3]) / (double)a.HW);
    __syncthreads();
  }
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.HW) return;
  float v[3] = {a.img[i], a.img[a.HW + i], a.img[2 * a.HW + i]};
  if (a.color_on) {
    const float gs = __fadd_rn(__fadd_rn(__fmul_rn(v[0], 0.114f), __fmul_rn(v[1], 0.587f)), __fmul_rn(v[2], 0.299f));
    const float gmean = s_mean;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float al = a.alpha[k];
      const float other = a.order[k] == 1 ? __fmul_rn(gmean, 1.f - al) : __fmul_rn(gs, 1.f - al);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        v[c] = __fmul_rn(v[c], al);
        if (a.order[k] != 0) v[c] = __fadd_rn(v[c], other);
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = (float)((double)v[c] + a.light[c]);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) a.img[c * a.HW + i] = __fdiv_rn(__fsub_rn(v[c], a.mean[c]), a.stdv[c]);
}
}  // namespace

extern "C" size_t cp_color_aug_workspace_bytes(void) { return CA_BLOCKS * sizeof(double); }

extern "C" int cp_color_aug_normalize(float* img, int64_t HW, int32_t color_on, const int32_t* order,
                                      const float* alpha, const double* light, const float* mean,
                                      const float* stdv, void* workspace, size_t workspace_bytes, void* stream) {
  CP_CHECK_ARG(img && mean && stdv && HW > 0);
  ColorArgs a;
  a.img = img; a.HW = HW; a.color_on = color_on ? 1 : 0;
  a.part = (const double*)workspace; a.nparts = 0;
  for (int c = 0; c < 3; ++c) { a.mean[c] = mean[c]; a.stdv[c] = stdv[c]; a.order[c] = 0; a.alpha[c] = 1.f; a.light[c] = 0.0; }
  hipStream_t st = (hipStream_t)stream;
  if (color_on) {
    CP_CHECK_ARG(order && alpha && light && workspace && workspace_bytes >= cp_color_aug_workspace_bytes());
    for (int c = 0; c < 3; ++c) {
      CP_CHECK_ARG(order[c] >= 0 && order[c] <= 2);
      a.order[c] = order[c]; a.alpha[c] = alpha[c]; a.light[c] = light[c];
    }
    long long nb = (HW + 255) / 256;
    a.nparts = (int)(nb < CA_BLOCKS ? nb : CA_BLOCKS);
    hipLaunchKernelGGL(color_gs_sum_kernel, dim3(a.nparts), dim3(256), 0, st, img, (long long)HW, (double*)workspace);
  }
  hipLaunchKernelGGL(color_apply_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, st, a);
  return cp_launch_status();
}
