// Instance-mask rasteriser of the Cityscapes result writer for gfx950.
//
// Replaces the per-instance PIL drawing of `CITYSCAPES.format_and_write_to_cityscapes`
// (reference: src/lib/datasets/dataset/cityscapes.py:240-272): for every detection, in ascending depth,
//   ImageDraw.polygon(points, outline=255, fill=255)  +  an ellipse of radius 2 around every pixel of the
//   closed Bresenham contour,  times (1 - to_remove_mask),  to_remove_mask |= mask when score >= 0.5.
// On 2048x1024 canvases that is one full-canvas PIL image, thousands of ellipse calls and several 2-Mpixel
// numpy passes per instance on the host.  Here:
//   contour kernel   one thread per polygon edge walks the integer Bresenham line (the `bresenham`
//                    package's walk: both end points, ties as its error term has them) and stamps the
//                    21-pixel pattern PIL 12.2 draws for that ellipse (5x5 without its corners);
//   fill kernel      one thread per bounding-box pixel: even-odd crossing test at the pixel's integer
//                    coordinates.  PIL's scan-line fill differs from it only in pixels on the polygon's
//                    boundary, all of which lie inside the contour band, so the UNION equals PIL's
//                    (mask for mask on the fixtures of tests/golden/writer_*.npz, made with PIL itself);
//   occlusion kernel one thread per canvas pixel walks the instances in depth order: the first covering
//                    instance with score >= 0.5 hides every later one; per-instance pixel counts.
#include "cp_common.h"

namespace {

struct WriterArgs {
  const int* poly;            // [n][N][2] (x, y), depth-sorted
  const unsigned char* flags; // [n]: bit 0 = draw (label has masks), bit 1 = occludes (score >= 0.5)
  unsigned char* masks;       // [n][H][W], 0 / 255
  int* counts;                // [n]
  int n, N, H, W;
};

__device__ __forceinline__ void stamp(unsigned char* m, int H, int W, int cx, int cy) {
#pragma unroll
  for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
      if ((dy == -2 || dy == 2) && (dx == -2 || dx == 2)) continue;      // PIL's radius-2 ellipse: no corners
      const int x = cx + dx, y = cy + dy;
      if (x >= 0 && x < W && y >= 0 && y < H) m[(long long)y * W + x] = 255;
    }
}

__global__ __launch_bounds__(64) void writer_contour_kernel(WriterArgs a) {
  const int e = blockIdx.x * 64 + threadIdx.x;                            // edge (instance, vertex)
  if (e >= a.n * a.N) return;
  const int i = e / a.N, v = e - i * a.N;
  if (!(a.flags[i] & 1)) return;
  const int* p = a.poly + (long long)i * a.N * 2;
  const int u = v == 0 ? a.N - 1 : v - 1;                                 // edge (v-1) -> v, closed
  int x0 = p[2 * u], y0 = p[2 * u + 1];
  const int x1 = p[2 * v], y1 = p[2 * v + 1];
  int dx = x1 - x0, dy = y1 - y0;
  const int xsign = dx > 0 ? 1 : -1, ysign = dy > 0 ? 1 : -1;
  dx = abs(dx); dy = abs(dy);
  int xx, xy, yx, yy;
  if (dx > dy) { xx = xsign; xy = 0; yx = 0; yy = ysign; }
  else { const int t = dx; dx = dy; dy = t; xx = 0; xy = ysign; yx = xsign; yy = 0; }
  int D = 2 * dy - dx, y = 0;
  unsigned char* m = a.masks + (long long)i * a.H * a.W;
  // far-away vertices (detections can leave the canvas): skip the part of the walk that cannot stamp it
  for (int x = 0; x <= dx; ++x) {
    const int px = x0 + x * xx + y * yx, py = y0 + x * xy + y * yy;
    if (px >= -2 && px < a.W + 2 && py >= -2 && py < a.H + 2) stamp(m, a.H, a.W, px, py);
    if (D >= 0) { y += 1; D -= 2 * dx; }
    D += 2 * dy;
  }
}

__global__ __launch_bounds__(256) void writer_fill_kernel(WriterArgs a) {
  const int i = blockIdx.z;
  if (!(a.flags[i] & 1)) return;
  __shared__ int sp[2 * 64];
  __shared__ int bb[4];
  const int* p = a.poly + (long long)i * a.N * 2;
  for (int k = threadIdx.x; k < 2 * a.N; k += 256) sp[k] = p[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    int x0 = sp[0], x1 = sp[0], y0 = sp[1], y1 = sp[1];
    for (int k = 1; k < a.N; ++k) {
      x0 = min(x0, sp[2 * k]); x1 = max(x1, sp[2 * k]);
      y0 = min(y0, sp[2 * k + 1]); y1 = max(y1, sp[2 * k + 1]);
    }
    bb[0] = max(x0, 0); bb[1] = min(x1, a.W - 1); bb[2] = max(y0, 0); bb[3] = min(y1, a.H - 1);
  }
  __syncthreads();
  const int bw = bb[1] - bb[0] + 1, bh = bb[3] - bb[2] + 1;
  if (bw <= 0 || bh <= 0) return;
  unsigned char* m = a.masks + (long long)i * a.H * a.W;
  for (long long q = (long long)(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; q < (long long)bw * bh;
       q += (long long)gridDim.x * gridDim.y * 256) {
    const int y = bb[2] + (int)(q / bw), x = bb[0] + (int)(q % bw);
    bool in = false;
    for (int k = 0; k < a.N; ++k) {
      const int u = k == 0 ? a.N - 1 : k - 1;
      const int ax = sp[2 * u], ay = sp[2 * u + 1], bx = sp[2 * k], by = sp[2 * k + 1];
      if (ay == by) continue;
      const bool span = (ay <= y && y < by) || (by <= y && y < ay);   // half-open in y: vertices count once
      if (span) {
        // x < ax + (y - ay) (bx - ax) / (by - ay), in exact integer arithmetic
        const long long num = (long long)(y - ay) * (bx - ax), den = by - ay;
        const long long lhs = (long long)(x - ax) * den;
        if (den > 0 ? lhs < num : lhs > num) in = !in;
      }
    }
    if (in) m[(long long)y * a.W + x] = 255;
  }
}

__global__ __launch_bounds__(256) void writer_occlude_kernel(WriterArgs a) {
  __shared__ int cnt[128];
  for (int k = threadIdx.x; k < a.n; k += 256) cnt[k] = 0;
  __syncthreads();
  const long long HW = (long long)a.H * a.W;
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p < HW) {
    bool removed = false;
    for (int i = 0; i < a.n; ++i) {
      unsigned char* m = a.masks + (long long)i * HW + p;
      if (*m) {
        if (removed) {
          *m = 0;
        } else {
          atomicAdd(&cnt[i], 1);
          if (a.flags[i] & 2) removed = true;
        }
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < a.n; k += 256)
    if (cnt[k]) atomicAdd(&a.counts[k], cnt[k]);
}

}  // namespace

extern "C" int cp_instance_masks(const int32_t* poly, const uint8_t* flags, int32_t n, int32_t N, int32_t H,
                                 int32_t W, uint8_t* masks, int32_t* counts, void* stream) {
  CP_CHECK_ARG(n >= 0 && N >= 3 && H > 0 && W > 0);
  if (n == 0) return CP_OK;
  CP_CHECK_ARG(poly && flags && masks && counts);
  if (n > 128 || N > 64) return CP_EUNSUPPORTED;
  WriterArgs a;
  a.poly = poly; a.flags = flags; a.masks = masks; a.counts = counts; a.n = n; a.N = N; a.H = H; a.W = W;
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(masks, 0, (size_t)n * H * W, st);
  (void)hipMemsetAsync(counts, 0, (size_t)n * sizeof(int), st);
  hipLaunchKernelGGL(writer_contour_kernel, dim3((n * N + 63) / 64), dim3(64), 0, st, a);
  hipLaunchKernelGGL(writer_fill_kernel, dim3(32, 8, n), dim3(256), 0, st, a);
  const long long HW = (long long)H * W;
  hipLaunchKernelGGL(writer_occlude_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, st, a);
  return cp_launch_status();
}
