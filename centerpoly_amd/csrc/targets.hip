// Training-target construction of the polydet sampler on the device.
//
// Replaces the per-object Python loop of PolydetDataset.__getitem__
// (reference: src/lib/datasets/sample/polydet.py:160-405) and the helpers it calls --
// affine_transform, gaussian_radius, gaussian2D, draw_umich_gaussian
// (src/lib/utils/image.py:62-65, 95-141): from the raw annotations of a batch (COCO boxes,
// polygon vertices, class ids) and each image's output affine it writes the heat maps and the
// per-object regression targets the loss consumes (batch schema of :425-449).
//
//   targets_object_kernel   one workgroup per image, one lane per object slot: flip + vertex
//                           re-ordering, affine + clip of every vertex and of the box, Gaussian
//                           radius, mass centre, ind / reg / peak / wh / poly (cartesian or
//                           polar) / reg_mask, and the image's mean class frequency.  float64
//                           where the reference computes in Python floats, float32 where it
//                           stores into float32 arrays, no fma contraction.
//   targets_splat_kernel    one workgroup per (object, splat): the (2r+1)^2 Gaussian of the
//                           centre into hm[class] and of every vertex into border_hm,
//                           max-composited with an integer atomicMax on the fp32 bit pattern
//                           (values are >= 0, so the order of objects does not matter --
//                           exactly np.maximum's result).
// HBM traffic is the heat-map zero fill (B*(C+1)*h*w*4 bytes) plus the splats; everything else
// is a few KB per image.
#include "cp_common.h"

#pragma clang fp contract(off)

namespace {

struct TargetArgs {
  const double* bbox;       // [B][M][4] x, y, w, h
  const double* poly_in;    // [B][M][2N]
  const int* cls_id;        // [B][M]
  const float* depth_in;    // [B][M]
  const float* freq_in;     // [B][M]
  const int* num_objs;      // [B]
  const uint8_t* flipped;   // [B]
  const int* img_width;     // [B]
  const double* trans;      // [B][6]
  float* hm;                // [B][C][h][w]
  float* border_hm;         // [B][1][h][w] or null
  uint8_t* reg_mask;        // [B][M]
  long long* ind;           // [B][M]
  float* poly;              // [B][M][2N]
  float* pseudo_depth;      // [B][M][1]
  float* peak;              // [B][M][2]
  float* reg;               // [B][M][2]
  float* wh;                // [B][M][2]
  float* freq_mask;         // [B]
  int* desc;                // workspace [B][M][4 + 2N]: valid, cls, cx, cy | radius in [3]... see below
  int B, M, N, C, h, w, rep, no_reorder_flip;
};

// desc row layout (ints): [0] radius or -1 (invalid), [1] class, [2] cx, [3] cy, [4 + 2i], [5 + 2i] vertex i
__device__ __forceinline__ int desc_stride(int N) { return 4 + 2 * N; }

// index into the un-reordered (already mirrored) vertex list that lands at position j after
// the reference's two re-ordering loops (polydet.py:181-187); Python negative indices wrap
__device__ __forceinline__ int flip_source(int j, int L) {
  const int fa = L / 4;
  int src = j;
  const int e = j & ~1;
  if (e < L / 4 + 2) src = (j & 1) ? fa - e + 1 : fa - e;
  const int d = j - fa;
  if (d >= 2 && !(d & 1) && d < 3 * L / 4) src = L - d;
  if (d - 1 >= 2 && !((d - 1) & 1) && d - 1 < 3 * L / 4) src = L - (d - 1) + 1;
  if (src < 0) src += L;
  return src;
}

__device__ __forceinline__ double clipd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

__device__ double gaussian_radius_d(long long height, long long width) {
  const double mo = 0.7;
  const double b1 = (double)(height + width);
  const double c1 = (double)(width * height) * (1 - mo) / (1 + mo);
  const double r1 = (b1 + sqrt((double)((height + width) * (height + width)) - 4 * c1)) / 2;
  const double b2 = (double)(2 * (height + width));
  const double c2 = (1 - mo) * (double)width * (double)height;
  const double r2 = (b2 + sqrt((double)(4 * (height + width) * (height + width)) - 16 * c2)) / 2;
  const double a3 = 4 * mo;
  const double b3 = -2 * mo * (double)(height + width);
  const double c3 = (mo - 1) * (double)width * (double)height;
  const double r3 = (b3 + sqrt(b3 * b3 - 4 * a3 * c3)) / 2;
  return fmin(r1, fmin(r2, r3));
}

__global__ __launch_bounds__(1024) void targets_object_kernel(TargetArgs a) {
  const int b = blockIdx.x, k = threadIdx.x;
  __shared__ double s_sum[1024];
  __shared__ int s_cnt[1024];
  double my_freq = 0.0;
  int my_cnt = 0;
  if (k < a.M) {
    const long long row = (long long)b * a.M + k;
    const int L = 2 * a.N;
    float* poly_o = a.poly + row * L;
    int* desc = a.desc + row * desc_stride(a.N);
    // defaults of an empty / skipped slot
    for (int i = 0; i < L; ++i) poly_o[i] = 0.f;
    a.reg_mask[row] = 0;
    a.ind[row] = 0;
    a.pseudo_depth[row] = 0.f;
    a.peak[2 * row] = a.peak[2 * row + 1] = 0.f;
    a.reg[2 * row] = a.reg[2 * row + 1] = 0.f;
    a.wh[2 * row] = a.wh[2 * row + 1] = 0.f;
    desc[0] = -1;
    const int n = min(a.num_objs[b], a.M);
    if (k < n) {
      const double* t = a.trans + 6 * b;
      const double* pin = a.poly_in + row * L;
      const bool flip = a.flipped[b] != 0;
      const int width = a.img_width[b];
      const double wmax = (double)(a.w - 1), hmax = (double)(a.h - 1);
      a.pseudo_depth[row] = a.depth_in[row];
      // vertex j (x for even j, y for odd j) after mirror + re-ordering, before the affine
      auto vertex = [&](int j) -> double {
        int s = j;
        if (flip && !a.no_reorder_flip) s = flip_source(j, L);
        double v = pin[s];
        if (flip && !(s & 1)) v = (double)width - v - 1;
        return v;
      };
      // affine_transform casts the point to float32, multiplies in float64; then np.clip
      auto point = [&](int i, double& x, double& y) {
        const double px = (double)(float)vertex(2 * i), py = (double)(float)vertex(2 * i + 1);
        x = clipd(t[0] * px + t[1] * py + t[2], 0.0, wmax);
        y = clipd(t[3] * px + t[4] * py + t[5], 0.0, hmax);
      };
      const double* bx = a.bbox + row * 4;
      float b0 = (float)bx[0], b1 = (float)bx[1], b2 = (float)(bx[0] + bx[2]), b3 = (float)(bx[1] + bx[3]);
      if (flip) {
        const float n0 = (float)width - b2 - 1.f, n2 = (float)width - b0 - 1.f;
        b0 = n0;
        b2 = n2;
      }
      {
        const double x0 = (double)b0, y0 = (double)b1, x1 = (double)b2, y1 = (double)b3;
        b0 = (float)(t[0] * x0 + t[1] * y0 + t[2]);
        b1 = (float)(t[3] * x0 + t[4] * y0 + t[5]);
        b2 = (float)(t[0] * x1 + t[1] * y1 + t[2]);
        b3 = (float)(t[3] * x1 + t[4] * y1 + t[5]);
      }
      b0 = fminf(fmaxf(b0, 0.f), (float)(a.w - 1));
      b2 = fminf(fmaxf(b2, 0.f), (float)(a.w - 1));
      b1 = fminf(fmaxf(b1, 0.f), (float)(a.h - 1));
      b3 = fminf(fmaxf(b3, 0.f), (float)(a.h - 1));
      const float hh = b3 - b1, ww = b2 - b0;
      if (hh > 0.f && ww > 0.f) {
        const double rad_d = gaussian_radius_d((long long)ceilf(hh), (long long)ceilf(ww));
        const int radius = max(0, (int)rad_d);
        double mx = 0.0, my = 0.0;
        for (int i = 0; i < a.N; ++i) {
          double x, y;
          point(i, x, y);
          mx += x;
          my += y;
        }
        const float ctx = (float)(mx / ((double)L / 2)), cty = (float)(my / ((double)L / 2));
        const int cxi = (int)ctx, cyi = (int)cty;
        desc[0] = radius;
        desc[1] = a.cls_id[row];
        desc[2] = cxi;
        desc[3] = cyi;
        a.wh[2 * row] = ww;
        a.wh[2 * row + 1] = hh;
        for (int i = 0; i < a.N; ++i) {
          double x, y;
          point(i, x, y);
          desc[4 + 2 * i] = (int)x;
          desc[5 + 2 * i] = (int)y;
          const double dx = x - (double)ctx, dy = y - (double)cty;
          if (a.rep == CP_REP_CARTESIAN) {
            poly_o[2 * i] = (float)dx;
            poly_o[2 * i + 1] = (float)dy;
          } else {
            const double r = sqrt(dx * dx + dy * dy);
            double th = atan((dy + 1e-8) / (dx + 1e-8));
            if (dx < 0) th = th + 3.141592653589793;
            else if (dy < 0) th = th + 2 * 3.141592653589793;
            poly_o[2 * i] = (float)r;
            poly_o[2 * i + 1] = (float)th;
          }
        }
        a.peak[2 * row] = ctx;
        a.peak[2 * row + 1] = cty;
        a.ind[row] = (long long)cyi * a.w + cxi;
        a.reg[2 * row] = (float)((double)ctx - (double)cxi);
        a.reg[2 * row + 1] = (float)((double)cty - (double)cyi);
        a.reg_mask[row] = (a.rep == CP_REP_POLAR && L > 5 && poly_o[1] > poly_o[5]) ? 0 : 1;
        const float f = a.freq_in[row];
        my_freq = (double)f;
        my_cnt = f != 0.f ? 1 : 0;
      }
    }
  }
  // mean class frequency over the slots with a non-zero entry (np.sum / np.count_nonzero)
  s_sum[k] = my_freq;
  s_cnt[k] = my_cnt;
  __syncthreads();
  if (k == 0) {
    double s = 0.0;
    int c = 0;
    for (int i = 0; i < (int)blockDim.x; ++i) {
      s += s_sum[i];
      c += s_cnt[i];
    }
    a.freq_mask[b] = c == 0 ? 1.f : (float)(s / c);
  }
}

__global__ __launch_bounds__(256) void targets_splat_kernel(TargetArgs a) {
  const int row = blockIdx.x;                 // object slot b * M + k
  const int s = blockIdx.y;                   // 0: centre -> hm[class]; 1..N: vertex -> border_hm
  const int* desc = a.desc + (long long)row * desc_stride(a.N);
  const int radius = desc[0];
  if (radius < 0) return;
  const int b = row / a.M;
  float* plane;
  int x, y;
  if (s == 0) {
    const int cls = desc[1];
    if (cls < 0 || cls >= a.C) return;
    plane = a.hm + ((long long)b * a.C + cls) * a.h * a.w;
    x = desc[2];
    y = desc[3];
  } else {
    if (!a.border_hm) return;
    plane = a.border_hm + (long long)b * a.h * a.w;
    x = desc[4 + 2 * (s - 1)];
    y = desc[5 + 2 * (s - 1)];
  }
  const int left = min(x, radius), right = min(a.w - x, radius + 1);
  const int top = min(y, radius), bottom = min(a.h - y, radius + 1);
  const int nx = left + right, ny = top + bottom;
  if (nx <= 0 || ny <= 0 || x - left < 0 || y - top < 0) return;
  const double diameter = (double)(2 * radius + 1);
  const double sigma = diameter / 6;
  const double denom = 2 * sigma * sigma;
  for (int e = threadIdx.x; e < nx * ny; e += 256) {
    const int iy = e / nx, ix = e - iy * nx;
    const double dx = (double)(ix - left), dy = (double)(iy - top);
    double g = exp(-(dx * dx + dy * dy) / denom);
    if (g < 2.220446049250313e-16) g = 0.0;   // h[h < eps * h.max()] = 0, h.max() == 1
    const float v = (float)g;
    atomicMax(reinterpret_cast<unsigned*>(plane + (long long)(y - top + iy) * a.w + (x - left + ix)),
              __float_as_uint(v));
  }
}

// --dense_poly (sample/polydet.py:401-403 -> utils/image.py:176-204 draw_dense_reg, called once per object right after
// the object's splat): regmap[:, window] = poly[k] wherever gaussian_k >= hm.max(axis=0) AS IT STANDS after objects
// 0..k.  One thread per pixel replays the objects in order with a running class-maximum: float64 Gaussian (the splat's
// own expression) against the float32 map value -- so where object k itself sets the maximum the test is
// g >= float32(g), true only when the rounding to float32 went down (the reference's behaviour, reproduced).
__global__ __launch_bounds__(256) void targets_dense_kernel(const int* __restrict__ desc_all, const float* __restrict__ poly,
                                                            float* __restrict__ dense, float* __restrict__ dmask, int M,
                                                            int N, int h, int w) {
  const int b = blockIdx.y;
  const int px = blockIdx.x * 256 + threadIdx.x;
  if (px >= h * w) return;
  const int y = px / w, x = px - y * w;
  const int ds = desc_stride(N);
  float run_max = 0.f;
  int cur = -1;
  for (int k = 0; k < M; ++k) {
    const int* desc = desc_all + ((long long)b * M + k) * ds;
    const int radius = desc[0];
    if (radius < 0) continue;
    const int cx = desc[2], cy = desc[3];
    const int left = min(cx, radius), right = min(w - cx, radius + 1);
    const int top = min(cy, radius), bottom = min(h - cy, radius + 1);
    if (left + right <= 0 || top + bottom <= 0 || cx - left < 0 || cy - top < 0) continue;
    if (x < cx - left || x >= cx + right || y < cy - top || y >= cy + bottom) continue;
    const double diameter = (double)(2 * radius + 1);
    const double sigma = diameter / 6;
    const double denom = 2 * sigma * sigma;
    const double dx = (double)(x - cx), dy = (double)(y - cy);
    double g = exp(-(dx * dx + dy * dy) / denom);
    if (g < 2.220446049250313e-16) g = 0.0;
    run_max = fmaxf(run_max, (float)g);                  // hm.max(axis=0) after draw_gaussian(hm[cls], ...)
    if (g >= (double)run_max) cur = k;
  }
  const long long plane = (long long)h * w;
  float* d = dense + (long long)b * 2 * N * plane + px;
  float* m = dmask + (long long)b * 2 * N * plane + px;
  const float* row = cur >= 0 ? poly + ((long long)b * M + cur) * 2 * N : nullptr;
  for (int c = 0; c < 2 * N; ++c) {
    const float v = row ? row[c] : 0.f;
    d[c * plane] = v;
    m[c * plane] = v != 0.f ? 1.f : 0.f;
  }
}

}  // namespace

extern "C" int cp_polydet_dense_targets(const cp_target_shape* s, const float* poly, const void* workspace,
                                        size_t workspace_bytes, float* dense_poly, float* dense_mask, void* stream) {
  CP_CHECK_ARG(s && poly && workspace && dense_poly && dense_mask);
  CP_CHECK_ARG(s->B > 0 && s->max_objs > 0 && s->nbr_points >= 3 && s->out_h > 0 && s->out_w > 0);
  if (s->B > 65535) return CP_EUNSUPPORTED;
  if (workspace_bytes < cp_polydet_targets_workspace_bytes(s)) return CP_EWORKSPACE;
  const int hw = s->out_h * s->out_w;
  hipLaunchKernelGGL(targets_dense_kernel, dim3((hw + 255) / 256, s->B), dim3(256), 0, (hipStream_t)stream,
                     (const int*)workspace, poly, dense_poly, dense_mask, s->max_objs, s->nbr_points, s->out_h, s->out_w);
  return cp_launch_status();
}

extern "C" size_t cp_polydet_targets_workspace_bytes(const cp_target_shape* s) {
  if (!s || s->B <= 0 || s->max_objs <= 0 || s->nbr_points <= 0) return 0;
  return cp_align_up((size_t)s->B * s->max_objs * (4 + 2 * (size_t)s->nbr_points) * sizeof(int), 256);
}

extern "C" int cp_polydet_targets(const cp_target_shape* s, const double* bbox_xywh,
                                  const double* poly_xy, const int32_t* cls_id,
                                  const float* pseudo_depth_in, const float* class_freq,
                                  const int32_t* num_objs, const uint8_t* flipped,
                                  const int32_t* img_width, const double* trans_output, float* hm,
                                  float* border_hm, uint8_t* reg_mask, int64_t* ind, float* poly,
                                  float* pseudo_depth, float* peak, float* reg, float* wh,
                                  float* freq_mask, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  CP_CHECK_ARG(s && bbox_xywh && poly_xy && cls_id && pseudo_depth_in && class_freq && num_objs);
  CP_CHECK_ARG(flipped && img_width && trans_output && hm && reg_mask && ind && poly);
  CP_CHECK_ARG(pseudo_depth && peak && reg && wh && freq_mask);
  CP_CHECK_ARG(s->B > 0 && s->max_objs > 0 && s->nbr_points >= 3 && s->num_classes > 0);
  CP_CHECK_ARG(s->out_h > 0 && s->out_w > 0);
  CP_CHECK_ARG(s->rep == CP_REP_CARTESIAN || s->rep == CP_REP_POLAR || s->rep == CP_REP_POLAR_FIXED);
  if (s->max_objs > 1024 || s->nbr_points > 1024) return CP_EUNSUPPORTED;
  if (!workspace || workspace_bytes < cp_polydet_targets_workspace_bytes(s)) return CP_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  TargetArgs a;
  a.bbox = bbox_xywh; a.poly_in = poly_xy; a.cls_id = cls_id; a.depth_in = pseudo_depth_in;
  a.freq_in = class_freq; a.num_objs = num_objs; a.flipped = flipped; a.img_width = img_width;
  a.trans = trans_output; a.hm = hm; a.border_hm = border_hm; a.reg_mask = reg_mask;
  a.ind = (long long*)ind; a.poly = poly; a.pseudo_depth = pseudo_depth; a.peak = peak; a.reg = reg;
  a.wh = wh; a.freq_mask = freq_mask; a.desc = (int*)workspace;
  a.B = s->B; a.M = s->max_objs; a.N = s->nbr_points; a.C = s->num_classes; a.h = s->out_h;
  a.w = s->out_w; a.rep = s->rep; a.no_reorder_flip = s->no_reorder_flip;
  const size_t plane = (size_t)s->out_h * s->out_w * sizeof(float);
  if (hipMemsetAsync(hm, 0, (size_t)s->B * s->num_classes * plane, st) != hipSuccess) return CP_EHIP;
  if (border_hm && hipMemsetAsync(border_hm, 0, (size_t)s->B * plane, st) != hipSuccess) return CP_EHIP;
  const int threads = (s->max_objs + 63) / 64 * 64;
  hipLaunchKernelGGL(targets_object_kernel, dim3(s->B), dim3(threads), 0, st, a);
  hipLaunchKernelGGL(targets_splat_kernel, dim3(s->B * s->max_objs, 1 + s->nbr_points), dim3(256), 0, st, a);
  return cp_launch_status();
}
