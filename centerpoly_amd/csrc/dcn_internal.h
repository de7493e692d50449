// Internal interface between the translation units of the DCNv2 forward (not part of the C ABI).
#pragma once
#include "cp_common.h"

// region-sampling split-bf16 kernel (dcn_fwd_region.hip): 3x3, stride 1, pad 1, dilation 1, Cin % 16 == 0
bool cp_dcn_region_supported(const cp_dcn_shape* s);
size_t cp_dcn_region_wperm_bytes(const cp_dcn_shape* s);
int cp_dcn_region_prepare(const cp_dcn_shape* s, const float* weight, void* wp, hipStream_t st);
size_t cp_dcn_region_om_wperm_bytes(const cp_dcn_shape* s);
int cp_dcn_region_prepare_om(const cp_dcn_shape* s, const float* om_weight, void* wp, hipStream_t st);
// om_wp != null: conv_offset_mask fused into the kernel (offset / mask unused, om_out optional [B][27][H][W]).
// ksplit > 1 (never with om_wp): the input channels are split over grid z, the slices' RAW sums go to
// partial[ksplit'][B][Cout][H][W] (ksplit' = the slice count the launch really uses, cp_dcn_region_ksplit(s) when that
// was passed) and the caller reduces them and applies the epilogue.
int cp_dcn_region_ksplit(const cp_dcn_shape* s);
int cp_dcn_region_forward(const cp_dcn_shape* s, const float* x, const float* offset, int64_t offset_bstride,
                          const float* mask, int64_t mask_bstride, int32_t mask_is_logit, const void* wp,
                          const float* bias, const float* ep_scale, const float* ep_shift, int32_t relu, float* out,
                          const void* om_wp, const float* om_bias, float* om_out, float* partial, int ksplit,
                          hipStream_t st);
