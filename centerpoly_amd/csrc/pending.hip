// Entry points declared in include/centerpoly_hip.h whose kernels are not written yet.
// They fail loudly (CP_EUNSUPPORTED) -- there is no fallback behind them.
#include "cp_common.h"

extern "C" size_t cp_poly_iou_order_workspace_bytes(int32_t, int32_t, int32_t) { return 0; }
extern "C" int cp_poly_iou_order_forward(const float*, const int64_t*, const uint8_t*, const float*,
                                         int32_t, int32_t, int32_t, int32_t, int32_t, int32_t,
                                         float*, float*, float*, void*, size_t, void*) {
  return CP_EUNSUPPORTED;
}
extern "C" int cp_poly_iou_order_backward(const float*, const int64_t*, const uint8_t*,
                                          const float*, int32_t, int32_t, int32_t, int32_t,
                                          int32_t, int32_t, const float*, const float*, float*,
                                          void*, size_t, void*) {
  return CP_EUNSUPPORTED;
}
