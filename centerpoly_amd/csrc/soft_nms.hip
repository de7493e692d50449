// soft_nms on HOST memory (no kernel: a sequential pass over at most a few hundred rows).
//
// Replaces the Cython extension external.nms.soft_nms the detector calls when several test
// scales are merged or --nms is set (reference: src/lib/external/nms.pyx:77-170, call site
// src/lib/detectors/polydet.py:66-67).  Literal behaviour, the arithmetic of the Cython-generated C: only
// columns 0-4 of a row are swapped / overwritten, discarded rows are overwritten by the last
// live row and the array keeps its length; the return value is the live count N.
#include <math.h>

#include "cp_common.h"

#pragma clang fp contract(off)

extern "C" int cp_soft_nms(float* boxes, int32_t n, int32_t row_stride, float sigma, float Nt,
                           float threshold, int32_t method) {
  if (n < 0 || (n > 0 && !boxes) || row_stride < 5) return CP_EINVAL;
  if (method < 0 || method > 2) return CP_EINVAL;
  auto at = [&](int r, int c) -> float& { return boxes[(long long)r * row_stride + c]; };
  int N = n;
  for (int i = 0; i < n; ++i) {           // the reference's range(N) is evaluated once
    float maxscore = at(i, 4);
    int maxpos = i;
    float t[5];
    for (int c = 0; c < 5; ++c) t[c] = at(i, c);
    for (int pos = i + 1; pos < N; ++pos)
      if (maxscore < at(pos, 4)) {
        maxscore = at(pos, 4);
        maxpos = pos;
      }
    for (int c = 0; c < 5; ++c) at(i, c) = at(maxpos, c);
    for (int c = 0; c < 5; ++c) at(maxpos, c) = t[c];
    const float tx1 = at(i, 0), ty1 = at(i, 1), tx2 = at(i, 2), ty2 = at(i, 3);
    int pos = i + 1;
    while (pos < N) {
      const float x1 = at(pos, 0), y1 = at(pos, 1), x2 = at(pos, 2), y2 = at(pos, 3);
      // Cython writes the int literal of `x2 - x1 + 1` as the double constant 1.0: float difference, then double
      // (pinned by the reference's own build, tests/golden/softnms_ref.npz)
      const float area = (float)(((double)(x2 - x1) + 1.0) * ((double)(y2 - y1) + 1.0));
      const float iw = (float)((double)((tx2 <= x2 ? tx2 : x2) - (tx1 >= x1 ? tx1 : x1)) + 1.0);
      if (iw > 0) {
        const float ih = (float)((double)((ty2 <= y2 ? ty2 : y2) - (ty1 >= y1 ? ty1 : y1)) + 1.0);
        if (ih > 0) {
          const float ua = (float)(((double)(tx2 - tx1) + 1.0) * ((double)(ty2 - ty1) + 1.0) + (double)area -
                                   (double)(iw * ih));
          const float ov = (iw * ih) / ua;
          float weight;
          if (method == 1) weight = ov > Nt ? (float)(1.0 - (double)ov) : 1.f;
          else if (method == 2) weight = (float)exp((double)(-(ov * ov) / sigma));
          else weight = ov > Nt ? 0.f : 1.f;
          at(pos, 4) = weight * at(pos, 4);
          if (at(pos, 4) < threshold) {
            for (int c = 0; c < 5; ++c) at(pos, c) = at(N - 1, c);
            --N;
            --pos;
          }
        }
      }
      ++pos;
    }
  }
  return N;
}
