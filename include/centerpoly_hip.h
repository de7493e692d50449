/*
 * centerpoly_hip.h -- C ABI of libcenterpoly_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary of the CenterPoly v2 `polydet` hot path.  Every entry
 * point takes raw DEVICE pointers, explicit sizes and a hipStream_t (passed as
 * void*); the library never allocates, frees or keeps a pointer after the call
 * returns, holds no global mutable state, is re-entrant and asynchronous on the
 * supplied stream.  Return value: CP_OK (0) or a negative CP_E* code; no C++
 * exception crosses this boundary.
 *
 * Which reference interface each entry point replaces (paths are relative to
 * the reference tree, /root/reference):
 *
 *   cp_dcn_v2_forward / cp_dcn_v2_backward
 *       the native extension behind `from .DCNv2.dcn_v2 import DCN`
 *       (src/lib/models/networks/pose_dla_dcn.py:16, call site :354;
 *       src/lib/models/networks/resnet_dcn.py:18,221).  Upstream
 *       CharlesShang/DCNv2 (absent from the tree) exposes
 *       dcn_v2_forward(input, weight, bias, offset, mask, kh,kw, sh,sw, ph,pw,
 *       dh,dw, dg) and dcn_v2_backward(..., grad_output) -> (grad_input,
 *       grad_offset, grad_mask, grad_weight, grad_bias).
 *   cp_depthwise_up_forward
 *       IDAUp's depth-wise ConvTranspose2d `up` + skip add,
 *       src/lib/models/networks/pose_dla_dcn.py:372-375, 381-387.
 *   cp_polydet_decode
 *       _nms + _topk + polydet_decode, src/lib/models/decode.py:13-19, 117-133,
 *       512-670, and the gather helpers src/lib/models/utils.py:12-26.
 *   cp_sigmoid_focal_forward / cp_sigmoid_focal_backward
 *       _sigmoid (src/lib/models/utils.py:8-10) + _neg_loss / FocalLoss
 *       (src/lib/models/losses.py:146-171, 792-799) as used at
 *       src/lib/trains/polydet.py:46,84.
 *   cp_gather_l1_forward / cp_gather_l1_backward
 *       _transpose_and_gather_feat + RegL1Loss (src/lib/models/losses.py:817-830)
 *       and the regression part of PolyLoss (src/lib/models/losses.py:910-949).
 *   cp_poly_iou_order_forward / cp_poly_iou_order_backward
 *       the per-object loop of PolyLoss (src/lib/models/losses.py:868-909) with
 *       WeilPolygonClipper (:373-628) and area (:25-41).
 *
 * All tensors are dense fp32 NCHW unless stated.  "B" is the per-device batch.
 */
#ifndef CENTERPOLY_HIP_H
#define CENTERPOLY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CP_ABI_VERSION 3   /* 3 (round 4): + cp_polydet_decode_ex, cp_dense_l1_*, cp_polydet_dense_targets, cp_conv_direct_forward_ex,
                              cp_conv_mfma_forward_split, cp_activation_split / _unsplit, cp_dla_base_pair_*; no signature changed */

enum {
  CP_OK = 0,
  CP_EINVAL = -1,       /* null pointer, non-positive size, inconsistent arguments */
  CP_EUNSUPPORTED = -2, /* valid but not implemented shape/option */
  CP_EWORKSPACE = -3,   /* workspace smaller than cp_*_workspace_bytes() */
  CP_EHIP = -4          /* a HIP runtime call or kernel launch failed */
};

/* representation of the polygon head (src/lib/opts.py `--rep`) */
enum { CP_REP_CARTESIAN = 0, CP_REP_POLAR = 1, CP_REP_POLAR_FIXED = 2 };

/* contraction arithmetic of cp_dcn_v2_forward */
enum {
  CP_DCN_F32 = 0,    /* fp32 MFMA, exact fp32 fma chain                                        */
  CP_DCN_BF16X3 = 1, /* split-bf16: a*b ~ ah*bh + ah*bl + al*bh on bf16 MFMA, fp32 accumulate;
                        ~2^-16 relative error.  The weights are split and permuted into the
                        head of `workspace` by a prologue launch of the call                  */
  CP_DCN_BF16X3_PREPARED = 2, /* the same, and `workspace` still holds the permuted weights of an
                        earlier CP_DCN_BF16X3 call with the same weight tensor and shape
                        (inference: one workspace per layer, prologue paid once)              */
  CP_DCN_BF16X3_REGION = 3, /* CP_DCN_BF16X3 on the LDS-region kernel whatever the map size (the
                        library otherwise picks it only where its tiles fill the chip);
                        CP_EUNSUPPORTED unless 3x3, stride 1, pad 1, dilation 1, Cin % 16 == 0 */
  CP_DCN_BF16X3_REGION_PREPARED = 4 /* ... with the weights already in `workspace`            */
};

/* regression flavour for cp_gather_l1_* */
enum {
  CP_L1_PLAIN = 0,       /* RegL1Loss / PolyLoss cartesian: sum |p*m - t*m|            */
  CP_L1_POLAR = 1,       /* PolyLoss polar: L1 on even slots + sum(1-cos) on odd slots */
  CP_L1_POLAR_FIXED = 2, /* PolyLoss polar_fixed: L1 on even (radius) slots only       */
  CP_L1_RELU20 = 3,      /* PolyLoss 'relu': |p-t| kept only where >= 20               */
  CP_L1_SMOOTH = 4       /* RegLoss (--reg_loss sl1): smooth_l1, divided by sum(mask) + eps (not x D) */
};

int cp_abi_version(void);
const char* cp_strerror(int code);
/* Name of the HIP device architecture the library was built for ("gfx950"). */
const char* cp_build_arch(void);

/* ------------------------------------------------------------------ DCNv2 --
 * Modulated deformable convolution, kernel kh x kw, one deformable group.
 *   x        [B, Cin, H, W]
 *   offset   per-tap (dy, dx) interleaved: channel 2k = dy_k, 2k+1 = dx_k,
 *            k = ky*kw + kx; element (b, ch, ho, wo) at
 *            offset[b*offset_bstride + ch*Ho*Wo + ho*Wo + wo]
 *   mask     [.., kh*kw, Ho, Wo], batch stride mask_bstride; if mask_is_logit != 0
 *            the kernel applies sigmoid itself (lets the caller pass the raw
 *            27-channel conv_offset_mask output: offset = om, mask = om + 18*Ho*Wo,
 *            both batch strides 27*Ho*Wo, without chunk/cat/sigmoid passes)
 *   weight   [Cout, Cin, kh, kw], bias [Cout] or NULL
 *   out      [B, Cout, Ho, Wo]
 * Optional fused epilogue (inference): out = act(acc * ep_scale[co] + ep_shift[co])
 * with ep_scale/ep_shift NULL meaning scale 1 / shift = bias; relu != 0 clamps at 0.
 * When ep_scale/ep_shift are given, bias must already be folded into ep_shift.
 * Small-spatial layers split K over workgroups; their partial sums live in the
 * caller's workspace (cp_dcn_v2_forward_workspace_bytes), behind the permuted weights of the
 * split-bf16 contraction; with CP_DCN_F32 and no K split the workspace may be NULL.
 */
typedef struct cp_dcn_shape {
  int32_t B, Cin, H, W, Cout;
  int32_t kh, kw, stride, pad, dil;
  int32_t deformable_groups; /* only 1 is implemented */
} cp_dcn_shape;

size_t cp_dcn_v2_forward_workspace_bytes(const cp_dcn_shape* s);
/* Which kernel cp_dcn_v2_forward runs for this shape and contraction: 0 = gather kernel, exact fp32 MFMA;
 * 1 = gather kernel, split-bf16; 2 = LDS-region kernel, split-bf16 (dcn_fwd_region.hip: the maps whose 8 x 32 pixel
 * tiles fill the chip).  Lets a caller choose between CP_DCN_F32 and CP_DCN_BF16X3 per layer; negative = CP_E*. */
int cp_dcn_v2_forward_kernel(const cp_dcn_shape* s, int32_t contraction);
int cp_dcn_v2_forward(const cp_dcn_shape* s, const float* x, const float* offset,
                      int64_t offset_bstride, const float* mask, int64_t mask_bstride,
                      int32_t mask_is_logit, const float* weight, const float* bias,
                      const float* ep_scale, const float* ep_shift, int32_t relu,
                      int32_t contraction, float* out, void* workspace, size_t workspace_bytes,
                      void* stream);

/* The DCN module of the reference in ONE launch: `DCN.forward` (upstream DCNv2/dcn_v2.py: out = conv_offset_mask(x);
 * o1, o2, mask = chunk(out, 3); offset = cat(o1, o2); mask = sigmoid(mask); dcn_v2_conv(x, offset, mask, ...)), i.e.
 * cp_dcn_v2_forward with the 27-channel 3x3 / pad 1 convolution that produces its offsets and mask logits computed
 * inside the kernel for each tile (split-bf16 x3 like cp_conv3x3_mfma_forward) instead of by a launch of its own.
 *   om_weight [27][Cin][3][3], om_bias [27]: conv_offset_mask's parameters
 *   om_out    NULL, or [B][27][H][W]: receives the convolution's output (what cp_dcn_v2_backward needs as
 *             offset = om_out, mask = om_out + 18 H W, both batch strides 27 H W, mask_is_logit = 1)
 *   prepared  != 0: `workspace` still holds the permuted weights of an earlier call with the same two weight tensors
 * Only where the LDS-region kernel runs (cp_dcn_v2_forward_fused_supported: 3x3, stride 1, pad 1, dilation 1,
 * Cin % 16 == 0, enough tiles to fill the chip); CP_EUNSUPPORTED otherwise -- run the convolution and
 * cp_dcn_v2_forward then. */
int cp_dcn_v2_forward_fused_supported(const cp_dcn_shape* s);
size_t cp_dcn_v2_forward_fused_workspace_bytes(const cp_dcn_shape* s);
int cp_dcn_v2_forward_fused(const cp_dcn_shape* s, const float* x, const float* om_weight, const float* om_bias,
                            const float* weight, const float* bias, const float* ep_scale, const float* ep_shift,
                            int32_t relu, int32_t prepared, float* om_out, float* out, void* workspace,
                            size_t workspace_bytes, void* stream);

/* Backward.  grad_* outputs may be NULL to skip that gradient.  grad_x, grad_offset and grad_mask are
 * OVERWRITTEN (the library zero-fills grad_x itself before its kernels accumulate into it: uninitialised memory is
 * fine, a caller that sums two branches adds them itself); grad_weight and grad_bias are ACCUMULATED INTO (caller
 * zero-fills).  mask is the post-sigmoid mask (mask_is_logit == 0) or the logits (then grad_mask is w.r.t. the
 * logits).  flags: 0 = the default kernels (split-bf16 x3 contraction, fp32 accumulate), or a bit-or of CP_DCN_BWD_*. */
enum {
  CP_DCN_BWD_EXACT_F32 = 1,     /* contract on the exact-fp32 MFMA chain instead of split-bf16 x3                 */
  CP_DCN_BWD_NARROW_TILES = 2,  /* data gradients: 8-row tiles on every layer (A/B timing of the 12-row form)    */
  CP_DCN_BWD_ROUND1_KERNELS = 4 /* the first-generation kernels (global float atomics; A/B timing and fallback)  */
};
size_t cp_dcn_v2_backward_workspace_bytes(const cp_dcn_shape* s);
int cp_dcn_v2_backward(const cp_dcn_shape* s, const float* x, const float* offset,
                       int64_t offset_bstride, const float* mask, int64_t mask_bstride,
                       int32_t mask_is_logit, const float* weight, const float* grad_out,
                       float* grad_x, float* grad_offset, int64_t grad_offset_bstride,
                       float* grad_mask, int64_t grad_mask_bstride, float* grad_weight,
                       float* grad_bias, int32_t flags, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------- depth-wise up-sampling --
 * IDAUp's `up` (depth-wise ConvTranspose2d, kernel 2f, stride f, padding f/2,
 * groups = C, no bias; src/lib/models/networks/pose_dla_dcn.py:372-375) fused with
 * the `+ layers[i-1]` that follows it (:381-387).
 *   x [B,C,H,W], weight [C,1,2f,2f], skip [B,C,H*f,W*f] or NULL -> out [B,C,H*f,W*f]
 * f in {2,4,8} forward; backward for f in {2,4}: grad_x is overwritten, grad_weight is
 * ACCUMULATED INTO (caller zero-fills); grad_skip = grad_out is the caller's business. */
int cp_depthwise_up_forward(const float* x, const float* weight, const float* skip, float* out,
                            int32_t B, int32_t C, int32_t H, int32_t W, int32_t f, void* stream);
int cp_depthwise_up_backward(const float* x, const float* weight, const float* grad_out,
                             float* grad_x, float* grad_weight, int32_t B, int32_t C, int32_t H,
                             int32_t W, int32_t f, void* stream);

/* 2x2 / stride 2 max pooling (floor mode) of the DLA trees' `downsample` (src/lib/models/networks/pose_dla_dcn.py:
 * 186-187,203-204), torch's tie rule (first maximum in row-major order, NaN propagates).  The backward recomputes the
 * arg-max from x and OVERWRITES grad_in (every element, also a trailing odd row / column). */
int cp_maxpool2x2_forward(const float* x, float* out, int32_t B, int32_t C, int32_t H, int32_t W, void* stream);
int cp_maxpool2x2_backward(const float* x, const float* grad_out, float* grad_in, int32_t B, int32_t C, int32_t H,
                           int32_t W, void* stream);

/* out = a + nearest-neighbour x2 up-sampling of low: `up1 + nn.Upsample(scale_factor=2)(low3)` of the Hourglass'
 * kp_module (src/lib/models/networks/large_hourglass.py:334-342) in one pass.
 * a, out [B][C][2H][2W]; low [B][C][H][W]; out may alias a. */
int cp_upsample2x_add(const float* a, const float* low, float* out, int32_t B, int32_t C, int32_t H, int32_t W,
                      void* stream);

/* y <- act(y + bias[c] + residual) in place (fp32 NCHW, HW = H*W): the epilogue of a library
 * convolution whose BatchNorm was folded (inference).  bias / residual may be NULL. */
int cp_bias_act_inplace(float* y, const float* bias, const float* residual, int32_t B, int32_t C,
                        int64_t HW, int32_t relu, void* stream);
/* backward of y = relu(conv + bias[c]) (cp_bias_act_inplace with relu, no residual; the heads'
 * Conv2d(3x3, bias) -> ReLU in training, src/lib/models/networks/pose_dla_dcn.py:448-451):
 * grad_in = grad_out * [y > 0] (grad_in == grad_out allowed), grad_bias[c] is ACCUMULATED INTO
 * (caller zero-fills).  HW % 4 == 0, 16-byte aligned tensors. */
int cp_bias_relu_backward(const float* y, const float* grad_out, float* grad_in, float* grad_bias,
                          int32_t B, int32_t C, int64_t HW, void* stream);
/* out[c] += sum over images and pixels of x[b][c][p] (fp32 NCHW): the bias gradient of a library
 * convolution (conv_offset_mask of DCN, src/lib/models/networks/pose_dla_dcn.py:354 via DCNv2). */
int cp_channel_sum_accumulate(const float* x, float* out, int32_t B, int32_t C, int64_t HW, void* stream);

/* Direct convolution + bias (+ ReLU) of the full-resolution, low-channel layers of the DLA base at
 * inference (src/lib/models/networks/pose_dla_dcn.py:236-246,266-276: `base_layer` 7x7 3->16,
 * `level0` 3x3 16->16, `level1` 3x3 s2 16->32; Conv2d(bias=False) -> BatchNorm2d -> ReLU with the
 * BatchNorm folded by the caller: scale into `w`, shift into `bias`):
 *   out[b][co][y][x] = act(bias[co] + sum w[co][ci][ky][kx] * x[b][ci][y*stride - pad + ky][x*stride - pad + kx])
 * x [B][Cin][H][W], w [Cout][Cin][k][k], bias [Cout] or NULL, out [B][Cout][Ho][Wo], zero padding.
 * Implemented shapes: (k 7, Cin 3, Cout 16, stride 1, pad 3) and (k 3, Cin 16, Cout 16|32, stride 1|2,
 * pad 1); cp_conv_direct_supported tells, anything else returns CP_EUNSUPPORTED. */
int cp_conv_direct_supported(int32_t Cin, int32_t Cout, int32_t k, int32_t stride, int32_t pad);
int cp_conv_direct_forward(const float* x, const float* w, const float* bias, float* out, int32_t B, int32_t Cin,
                           int32_t H, int32_t W, int32_t Cout, int32_t k, int32_t stride, int32_t pad,
                           int32_t relu, void* stream);
/* ABI v3: the same call with the arithmetic as an argument -- split_bf16 != 0 runs the stride-1 3x3 / 16-input-channel
 * layers (level0 of the DLA base) as split-bf16 x3 on the bf16 matrix cores (fp32 in and out, ~2^-16 per product; the
 * stride-2 layers keep the exact kernel, which measured faster); 0 = cp_conv_direct_forward (exact fp32 fma chains). */
int cp_conv_direct_forward_ex(const float* x, const float* w, const float* bias, float* out, int32_t B, int32_t Cin,
                           int32_t H, int32_t W, int32_t Cout, int32_t k, int32_t stride, int32_t pad,
                           int32_t relu, int32_t split_bf16, void* stream);

/* 3x3 / stride 1 / pad 1 convolution of float32 NCHW maps on the bf16 matrix cores (split-bf16 x3: float32 in and
 * out, ~2^-16 relative error per product): the dense convolutions of BasicBlock (src/lib/models/networks/
 * pose_dla_dcn.py:38-66), of the heads' `fc` (:445-462), of DCN.conv_offset_mask (DCNv2/dcn_v2.py:137-145) and of
 * large_hourglass.py's convolution / residual (:24-37, :55-81) -- what the reference hands to cuDNN.
 *   out[b][co][y][x] = act(bias[co] + residual[b][co][y][x] + sum w[co][ci][ky][kx] * x[b][ci][y - 1 + ky][x - 1 + kx])
 * The weights are split and permuted once by cp_conv3x3_mfma_prepare into `wperm` (cp_conv3x3_mfma_weight_bytes);
 * with transposed = 1 the prologue reads a [Cin][Cout][3][3] tensor as its transposed, flipped self, so that the
 * same kernel computes the INPUT GRADIENT of the convolution whose weight that tensor is (x := grad_out).
 * The contraction runs in steps of 32 input channels (a ragged last step is zero-filled); tensors must stay within
 * 32-bit byte offsets per image: cp_conv3x3_mfma_supported tells, anything else returns CP_EUNSUPPORTED.
 * bias, residual may be NULL. */
int cp_conv3x3_mfma_supported(int32_t Cin, int32_t Cout, int32_t H, int32_t W);
size_t cp_conv3x3_mfma_weight_bytes(int32_t Cin, int32_t Cout);
int cp_conv3x3_mfma_prepare(const float* weight, int32_t Cin, int32_t Cout, int32_t transposed, void* wperm,
                            void* stream);
int cp_conv3x3_mfma_forward(const float* x, const void* wperm, const float* bias, const float* residual, float* out,
                            int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t relu, void* stream);

/* General form of the above: taps = 9 (3x3 / pad 1) or 1 (1x1), and the input given as the channel concatenation of
 * nsrc (1..4) tensors xs[i] = [B][cs[i]][H][W] read in place -- the torch.cat of `Root.forward`
 * (src/lib/models/networks/pose_dla_dcn.py:148-166) is never materialised.  With nsrc > 1 every cs[i] must be a
 * multiple of 32.  Weights prepared by cp_conv_mfma_prepare with the same taps over Cin = sum cs[i]. */
size_t cp_conv_mfma_weight_bytes(int32_t Cin, int32_t Cout, int32_t taps);
int cp_conv_mfma_prepare(const float* weight, int32_t Cin, int32_t Cout, int32_t taps, int32_t transposed, void* wperm,
                         void* stream);
/* The same for a whole table of weights in ONE launch (a training step uses ~110 forms -- every convolution's forward and
 * transposed one -- each a 4-microsecond launch of its own when prepared per use): `jobs` lives in DEVICE memory, job i's
 * workgroups are first_block .. first_block + cp_conv_mfma_prepare_blocks(Cin, Cout, taps) - 1 (first_block ascending,
 * job 0 at 0), total_blocks their sum.  Cin / Cout / transposed as for cp_conv_mfma_prepare. */
typedef struct cp_conv_prepare_job {
  const float* weight;
  void* wperm;
  int32_t Cin, Cout, taps, transposed, first_block, reserved;
} cp_conv_prepare_job;
int32_t cp_conv_mfma_prepare_blocks(int32_t Cin, int32_t Cout, int32_t taps);
int cp_conv_mfma_prepare_batch(const cp_conv_prepare_job* jobs_device, int32_t njobs, int32_t total_blocks, void* stream);
int cp_conv_mfma_forward(const float* const* xs, const int32_t* cs, int32_t nsrc, const void* wperm, const float* bias,
                         const float* residual, float* out, int32_t B, int32_t H, int32_t W, int32_t Cout,
                         int32_t taps, int32_t relu, void* stream);
/* the same with a stride of 1 or 2 (3x3: the first convolution of DLA levels 2-5, pose_dla_dcn.py:38-46, and of the
 * Hourglass' down-sampling residuals; 1x1: their skip convolutions, large_hourglass.py:55-81);
 * out is [B][Cout][(H - 1) / stride + 1][(W - 1) / stride + 1] */
int cp_conv_mfma_forward_strided(const float* const* xs, const int32_t* cs, int32_t nsrc, const void* wperm,
                                 const float* bias, const float* residual, float* out, int32_t B, int32_t H, int32_t W,
                                 int32_t Cout, int32_t taps, int32_t stride, int32_t relu, void* stream);
/* level0 + level1 of the DLA base at inference as ONE launch (pose_dla_dcn.py:236-246,266-276, BatchNorm folded):
 *   out = relu(conv3x3 stride 2 pad 1 (relu(conv3x3 pad 1 (x, w0) + b0), w1) + b1)
 * x [B][16][H][W] -> out [B][32][(H-1)/2+1][(W-1)/2+1]; w0 [16][16][3][3], w1 [32][16][3][3], b0 / b1 may be null.
 * The 16-channel full-resolution intermediate (the network's largest activation, one consumer) stays in LDS.
 * split-bf16 x3 arithmetic in both layers.  W % 4 == 0 and a 16-byte aligned x (cp_dla_base_pair_supported), else
 * CP_EUNSUPPORTED -- run the two layers through cp_conv_direct_forward_ex. */
int cp_dla_base_pair_supported(int32_t H, int32_t W);
int cp_dla_base_pair_forward(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, float* out,
                             int32_t B, int32_t H, int32_t W, void* stream);
/* SPLIT activations (round 4; inference, between the two convolutions of a BasicBlock, pose_dla_dcn.py:38-66): a
 * float32 tensor [B][C][H][W] kept as two planes [hi | lo] of [B][C / 8][H][W][8 x bf16] (hi = bf16(v), lo =
 * bf16(v - hi): the halves the convolution's staging forms anyway; same bytes as float32; C % 8 == 0).  The producer's
 * epilogue writes them (out_split), the consumer stages them with two 16-byte loads per unit and no conversion
 * arithmetic (x_split: 3x3 / stride 1, Cin % 32 == 0).  Results are bit-identical to the float32 route.
 * cp_conv_mfma_forward_split = cp_conv_mfma_forward_strided for one source; x / out are float32 tensors or split
 * planes as the flags say (out_split: no residual).  cp_activation_split / _unsplit convert (tests, probes). */
int cp_conv_mfma_forward_split(const void* x, int32_t x_split, const void* wperm, const float* bias, const float* residual,
                               void* out, int32_t out_split, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t Cout,
                               int32_t taps, int32_t stride, int32_t relu, void* stream);
int cp_activation_split(const float* x, void* out, int32_t B, int32_t C, int32_t H, int32_t W, void* stream);
int cp_activation_unsplit(const void* in, float* x, int32_t B, int32_t C, int32_t H, int32_t W, void* stream);
/* Weight gradient of a 3x3 / stride 2 / pad 1 convolution (the first convolution of DLA levels 2-5,
 * src/lib/models/networks/pose_dla_dcn.py:32-40; cuDNN's backward-filter in the reference), same arithmetic:
 *   gw[co][ci][ky][kx] += sum_{b,i,j} grad_out[b][co][i][j] * x[b][ci][2 i + ky - 1][2 j + kx - 1]
 * x [B][Cin][H][W], grad_out [B][Cout][(H - 1) / 2 + 1][W / 2]; gw [Cout][Cin][3][3] is ACCUMULATED into (float atomics:
 * zero it first).  Needs W % 8 == 0; cp_conv3x3_s2_wgrad_supported tells, else CP_EUNSUPPORTED. */
int cp_conv3x3_s2_wgrad_supported(int32_t Cin, int32_t Cout, int32_t H, int32_t W);
int cp_conv3x3_s2_wgrad(const float* x, const float* grad_out, float* gw, int32_t B, int32_t Cin, int32_t H, int32_t W,
                        int32_t Cout, void* stream);

/* 7x7 / pad 3 convolution of a 3-channel image (+ bias, + ReLU), stride 1 or 2, on the bf16 matrix cores (split-bf16 x3,
 * the arithmetic above): the Hourglass stem `pre = convolution(7, 3, 128, stride=2)`
 * (src/lib/models/networks/large_hourglass.py:287-290) and DLA's `base_layer` Conv2d(3, 16, 7, stride=1)
 * (src/lib/models/networks/pose_dla_dcn.py:236-241); cuDNN in the reference, BatchNorm folded by the caller:
 *   out[b][co][y][x] = act(bias[co] + sum w[co][ci][ky][kx] * x[b][ci][S y - 3 + ky][S x - 3 + kx])
 * x [B][3][H][W], weight [Cout][3][7][7] (prepared once into wperm), out [B][Cout][(H - 1) / S + 1][(W - 1) / S + 1]. */
int cp_conv7x7_c3_supported(int32_t Cout, int32_t H, int32_t W, int32_t stride);
size_t cp_conv7x7_c3_weight_bytes(int32_t Cout);
int cp_conv7x7_c3_prepare(const float* weight, int32_t Cout, void* wperm, void* stream);
int cp_conv7x7_c3_forward(const float* x, const void* wperm, const float* bias, float* out, int32_t B, int32_t H,
                          int32_t W, int32_t Cout, int32_t stride, int32_t relu, void* stream);

/* Input gradient of a stride-1 convolution (3x3 / pad 1 or 1x1) whose INPUT was the output y of a bias + ReLU epilogue
 * (the heads' Conv2d(3x3, bias) -> ReLU -> Conv2d(1x1), src/lib/models/networks/pose_dla_dcn.py:445-462), with that
 * ReLU's backward and its bias gradient in the kernel's epilogue (threshold_backward + the bias sum in the reference):
 *   grad_y[b][c][p] = [y[b][c][p] > 0] * sum_{co, tap} weight[co][c][tap'] * grad_out[b][co][p + tap]
 *   grad_bias[c]   += sum_{b, p} grad_y[b][c][p]     (NULL: not wanted; accumulated into: zero it first.  Every wave
 *                     leaves its channel sums in the workspace, a second small kernel adds them up)
 *   wperm_t  cp_conv_mfma_prepare(weight [Cout][Cin][k][k], Cin := Cout, Cout := Cin, taps, transposed = 1)
 *   grad_out [B][Cout][H][W], y and grad_y [B][Cin][H][W].  Cout below 32 multiplies zeros up to 32 (the launch is
 *   bound by the two [B][Cin][H][W] streams). */
size_t cp_conv_mfma_input_grad_relu_workspace_bytes(int32_t B, int32_t Cin, int32_t H, int32_t W);
int cp_conv_mfma_input_grad_relu(const float* grad_out, const void* wperm_t, const float* y, float* grad_y,
                                 float* grad_bias, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t Cout,
                                 int32_t taps, void* workspace, size_t workspace_bytes, void* stream);
/* Input gradient of a 3x3 / stride 2 / pad 1 convolution (the first convolution of DLA levels 1-5,
 * src/lib/models/networks/pose_dla_dcn.py:32-40,236-246; cuDNN's backward-data in the reference) in ONE launch: per parity
 * class (py, px) of the gradient's rows / columns it is a stride-1 convolution of grad_out with 1, 2, 2 or 4 of the nine
 * taps; grad_out is staged once per workgroup and feeds the four classes' accumulators tap by tap -- the matrix cores do
 * exactly the forward's flops (no multiplications by inserted zeros), grad_out is read once:
 *   grad_in = (residual ? residual : 0) + conv_transpose2d(grad_out, weight, stride 2, pad 1) cropped to H x W
 *   wperm_t  cp_conv_mfma_prepare(weight [Cout][Cin][3][3], Cin := Cout, Cout := Cin, taps 9, transposed = 6)
 *   grad_out [B][Cout][(H - 1) / 2 + 1][(W - 1) / 2 + 1] -> grad_in [B][Cin][H][W], every element written
 *   residual [B][Cin][H][W] or NULL (may be grad_in itself: every element is read, then written, by the same lane) */
int cp_conv3x3_s2_input_grad(const float* grad_out, const void* wperm_t, const float* residual, float* grad_in, int32_t B,
                             int32_t Cin, int32_t H, int32_t W, int32_t Cout, void* stream);

/* Weight gradient of the same convolution, same arithmetic (what the reference gets from cuDNN's backward-filter):
 *   gw[co][ci][ky][kx] += sum_{b,y,x} go[b][co][y][x] * x[b][ci][y - 1 + ky][x - 1 + kx]
 * gw [Cout][Cin][3][3] is ACCUMULATED into (float atomics: zero it first; the summation order varies from run to
 * run in the last bits).  Needs W % 4 == 0; cp_conv3x3_mfma_wgrad_supported tells, else CP_EUNSUPPORTED. */
int cp_conv3x3_mfma_wgrad_supported(int32_t Cin, int32_t Cout, int32_t H, int32_t W);
int cp_conv3x3_mfma_wgrad(const float* x, const float* go, float* gw, int32_t B, int32_t Cin, int32_t H, int32_t W,
                          int32_t Cout, void* stream);
/* the same with taps = 9 (3x3 / pad 1) or 1 (1x1: gw [Cout][Cin][1][1]) */
int cp_conv_mfma_wgrad(const float* x, const float* go, float* gw, int32_t B, int32_t Cin, int32_t H, int32_t W,
                       int32_t Cout, int32_t taps, void* stream);

/* Weight gradient of the same three full-resolution layers (cuDNN backward-filter in the reference), exact fp32 MFMA:
 *   gw[co][ci][ky][kx] += sum_{b,y,x} go[b][co][y][x] * x[b][ci][y*stride - pad + ky][x*stride - pad + kx]
 * gw [Cout][Cin][k][k] is ACCUMULATED into (float atomics: zero it first).  Shapes: (k 7, 3->16, stride 1, pad 3),
 * (k 3, 16->16, stride 1, pad 1), (k 3, 16->32, stride 2, pad 1); cp_conv_direct_wgrad_supported tells. */
int cp_conv_direct_wgrad_supported(int32_t Cin, int32_t Cout, int32_t k, int32_t stride, int32_t pad);
int cp_conv_direct_wgrad(const float* x, const float* go, float* gw, int32_t B, int32_t Cin, int32_t H, int32_t W,
                         int32_t Cout, int32_t k, int32_t stride, int32_t pad, void* stream);

/* The detection heads at inference as ONE kernel (the `fc` Sequentials of DLASeg over their shared input,
 * src/lib/models/networks/pose_dla_dcn.py:445-462,479-481):
 *   out[h][b][o][p] = b2[h][o] + sum_c w2[h][o][c] * relu(b1[h * HC + c] + conv3x3(x, w1)[b][h * HC + c][p])
 * w1 = the heads' 3x3 weights concatenated along the output channels ([nheads * HC][Cin][3][3]) and prepared by
 * cp_conv_mfma_prepare(taps 9) into wperm1; w2[h] = the head's 1x1 weight [cout[h]][HC] prepared by
 * cp_heads_fused_prepare_w2.  The HC-channel intermediate stays in accumulator registers (split-bf16 x3 on both
 * stages).  nheads <= 4, cout[h] <= 64, HC % 64 == 0, Cin % 32 == 0; otherwise CP_EUNSUPPORTED. */
size_t cp_heads_fused_w2_bytes(int32_t head_conv);
int cp_heads_fused_prepare_w2(const float* w2, int32_t cout, int32_t head_conv, void* w2perm, void* stream);
int cp_heads_fused_forward(const float* x, const void* wperm1, const float* b1, const void* const* w2perm,
                           const float* const* b2, float* const* out, const int32_t* cout, int32_t nheads, int32_t B,
                           int32_t Cin, int32_t H, int32_t W, int32_t head_conv, void* stream);

/* Output stage of a detection head at inference (the `fc` Sequential of DLASeg,
 * src/lib/models/networks/pose_dla_dcn.py:445-462: Conv2d 3x3 + bias -> ReLU -> Conv2d 1x1 + bias),
 * everything after the 3x3 convolution's matrix product, one pass:
 *   out[b][o][p] = bias[o] + sum_c w_t[c][o] * act(y[b][c][p] + in_bias[c])
 * y: raw 3x3-conv output, channel slice of a larger tensor allowed (y_bstride = elements between
 * images); in_bias [Cin] or NULL; relu_in != 0 applies ReLU; w_t is the 1x1 weight TRANSPOSED to
 * [Cin][Cout]; bias [Cout] or NULL; out [B][Cout][HW].  Cout <= 32, HW % 4 == 0, 16-byte aligned
 * y / out (otherwise CP_EUNSUPPORTED). */
int cp_conv1x1_act_forward(const float* y, int64_t y_bstride, const float* in_bias, int32_t relu_in,
                           const float* w_t, const float* bias, float* out, int32_t B, int32_t Cin,
                           int32_t Cout, int64_t HW, void* stream);

/* ------------------------------------------------ evaluation result writer --
 * cp_instance_masks: the per-instance rasterisation of CITYSCAPES.format_and_write_to_cityscapes
 * (src/lib/datasets/dataset/cityscapes.py:240-272) for the instances of ONE image, already sorted by
 * ascending depth: polygon fill + outline, radius-2 dilation of the closed Bresenham contour, removal of
 * what nearer instances with score >= 0.5 hide.
 *   poly   DEVICE int32 [n][N][2]  integer (x, y) vertices      flags  DEVICE uint8 [n]: bit 0 = the label
 *   has masks (not pole / traffic sign / traffic light), bit 1 = score >= 0.5 (hides farther instances)
 *   masks  DEVICE uint8 [n][H][W] out, 0 / 255                  counts DEVICE int32 [n] out, non-zero pixels
 * n <= 128, N <= 64.  Equal to PIL 12.2's drawing mask for mask (tests/golden/writer_*.npz). */
int cp_instance_masks(const int32_t* poly, const uint8_t* flags, int32_t n, int32_t N, int32_t H, int32_t W,
                      uint8_t* masks, int32_t* counts, void* stream);

/* ------------------------------------------------ detector pre/post-processing --
 * cp_preprocess_warp_normalize: the cv2 stage of BaseDetector.pre_process
 * (src/lib/detectors/base_detector.py:66-87): cv2.warpAffine(image, trans_input, (dst_w, dst_h),
 * flags=INTER_LINEAR) on the 8-bit image followed by ((x / 255. - mean) / std) and the HWC -> CHW
 * transpose; OpenCV's fixed-point arithmetic, bit-identical to oracle/pre.py.
 *   src   DEVICE uint8 [src_h][src_w][3]      trans  HOST float64[6], forward map src -> dst
 *   mean, std  HOST float32[3]                out    DEVICE fp32 [1 + flip_copy][3][dst_h][dst_w]
 * flip_copy != 0 also writes the horizontally flipped image behind it (--flip_test, :84-85).
 *
 * cp_polydet_post_process: transform_preds of polydet_post_process
 * (src/lib/utils/post_process.py:105-122, src/lib/utils/image.py:19-24) + the `/ scale` of
 * PolydetDetector.post_process (src/lib/detectors/polydet.py:52-57) on the decoded rows
 *   dets [B][K][ncols] (ncols = 2N + 7: x1,y1,x2,y2,score,class, N vertices, depth), DEVICE
 *   trans_dev  DEVICE float64 [B][6]: inverse affine (output map -> image) per image
 * Box corners and vertices are mapped in float64, cast to fp32, divided by `scale`; the other
 * columns are copied.  out has the layout of dets (out == dets is NOT allowed). */
int cp_preprocess_warp_normalize(const uint8_t* src, int32_t src_h, int32_t src_w,
                                 const double* trans, const float* mean, const float* stdv,
                                 int32_t dst_h, int32_t dst_w, int32_t flip_copy, float* out,
                                 void* stream);

/* cp_color_aug_normalize: the colour augmentation + normalisation of the TRAINING sampler
 * (src/lib/datasets/sample/polydet.py:128-136: `inp / 255.` -> color_aug -> `(inp - mean) / std`;
 * src/lib/utils/image.py:231-264) in place on the warped input, BGR planes [3][HW] holding x / 255
 * (cp_preprocess_warp_normalize with mean 0 / std 1).  The random draws stay with the caller
 * (reference: data_rng + random.shuffle in the loader worker): order[3] = the three ops in
 * application order (0 brightness, 1 contrast, 2 saturation), alpha[3] their blend factors,
 * light[3] = eig_vec . (eig_val * alpha_pca) per BGR channel (float64).  color_on == 0 only
 * normalises.  All HOST pointers except img / workspace (cp_color_aug_workspace_bytes). */
size_t cp_color_aug_workspace_bytes(void);
int cp_color_aug_normalize(float* img, int64_t HW, int32_t color_on, const int32_t* order,
                           const float* alpha, const double* light, const float* mean, const float* stdv,
                           void* workspace, size_t workspace_bytes, void* stream);
int cp_polydet_post_process(const float* dets, const double* trans_dev, float scale, int32_t B,
                            int32_t K, int32_t ncols, float* out, void* stream);

/* soft_nms of external/nms.pyx:77-170 (Cython in the reference) on HOST float32 rows
 * [n][row_stride] (x1,y1,x2,y2,score,...), in place, as merge_outputs uses it
 * (src/lib/detectors/polydet.py:66-67: Nt=0.5, method=2).  method 0 hard / 1 linear /
 * 2 gaussian.  Only columns 0-4 move; returns the live row count N (>= 0) or CP_EINVAL. */
int cp_soft_nms(float* boxes_host, int32_t n, int32_t row_stride, float sigma, float Nt,
                float threshold, int32_t method);

/* ------------------------------------------------- training-target construction --
 * The per-object loop of PolydetDataset.__getitem__ (src/lib/datasets/sample/polydet.py:160-405)
 * with affine_transform / gaussian_radius / draw_umich_gaussian (src/lib/utils/image.py:62-65,
 * 95-141): raw annotations of a batch -> the tensors of the batch dict (:425-449).
 * Inputs (all DEVICE): bbox_xywh f64 [B][M][4] COCO boxes, poly_xy f64 [B][M][2N] vertices in
 * image coordinates, cls_id i32 [B][M], pseudo_depth_in f32 [B][M], class_freq f32 [B][M] (the
 * object's class frequency), num_objs i32 [B] (slots >= num_objs stay zero), flipped u8 [B],
 * img_width i32 [B] (mirror axis), trans_output f64 [B][6] (get_affine_transform(c, s, 0,
 * [out_w, out_h])).  Outputs (DEVICE, fully written): hm f32 [B][C][h][w], border_hm f32
 * [B][1][h][w] or NULL, reg_mask u8 [B][M], ind i64 [B][M], poly f32 [B][M][2N], pseudo_depth
 * f32 [B][M][1], peak / reg / wh f32 [B][M][2], freq_mask f32 [B] (mean over the image's
 * objects, 1 when none).  rep: CP_REP_* (polar targets are (r, theta) pairs). */
typedef struct cp_target_shape {
  int32_t B, max_objs, nbr_points, num_classes, out_h, out_w, rep, no_reorder_flip;
} cp_target_shape;
size_t cp_polydet_targets_workspace_bytes(const cp_target_shape* s);
int cp_polydet_targets(const cp_target_shape* s, const double* bbox_xywh, const double* poly_xy,
                       const int32_t* cls_id, const float* pseudo_depth_in, const float* class_freq,
                       const int32_t* num_objs, const uint8_t* flipped, const int32_t* img_width,
                       const double* trans_output, float* hm, float* border_hm, uint8_t* reg_mask,
                       int64_t* ind, float* poly, float* pseudo_depth, float* peak, float* reg,
                       float* wh, float* freq_mask, void* workspace, size_t workspace_bytes,
                       void* stream);

/* `--dense_poly` targets (src/lib/datasets/sample/polydet.py:401-403,429-441 with draw_dense_reg,
 * src/lib/utils/image.py:176-204): after cp_polydet_targets on the SAME shape and workspace (its per-object
 * descriptors are read from it), dense_poly [B][2N][h][w] holds, per pixel, the polygon row of the LAST object k (in
 * annotation order) whose Gaussian there is >= the class-maximum heat map as it stood after drawing objects 0..k
 * (the comparison the reference makes: float64 Gaussian against the float32 map), 0 where no object qualifies;
 * dense_mask [B][2N][h][w] = (dense_poly != 0) as 0 / 1 floats.  poly = cp_polydet_targets' poly output. */
int cp_polydet_dense_targets(const cp_target_shape* s, const float* poly, const void* workspace,
                             size_t workspace_bytes, float* dense_poly, float* dense_mask, void* stream);

/* --------------------------- fused training BatchNorm2d (+residual) (+ReLU) --
 * y = act(bn(x) + residual) with batch statistics (torch.nn.BatchNorm2d training semantics:
 * biased variance for normalisation, running stats updated with `momentum`, unbiased variance).
 * Replaces the BN -> add -> ReLU chains of DeformConv / BasicBlock / Root / conv levels
 * (src/lib/models/networks/pose_dla_dcn.py:32-60,148-166,266-277,347-359) in training.
 * x, y, residual and their gradients: fp32 [B,C,H,W], HW = H x W; weight, bias, running and
 * saved statistics: [C].  backward: grad_weight / grad_bias are OVERWRITTEN; grad_residual (may be NULL) receives
 * the post-activation gradient; workspace from cp_bn_workspace_bytes.  `bias` is the forward's bias (NULL if it had none);
 * with relu and NO residual in the forward, pass y = NULL: the ReLU mask [y > 0] is then recomputed from x (the forward's
 * own expression, bit-identical) and y is not read -- a third less traffic (y = NULL with grad_residual: CP_EINVAL). */
size_t cp_bn_workspace_bytes(int32_t B, int32_t C, int64_t HW);
int cp_bn_act_forward_train(const float* x, const float* weight, const float* bias,
                            const float* residual, float* y, float* save_mean, float* save_invstd,
                            float* running_mean, float* running_var, float momentum, float eps,
                            int32_t relu, int32_t B, int32_t C, int64_t HW, void* workspace,
                            size_t workspace_bytes, void* stream);
int cp_bn_act_backward(const float* x, const float* y, const float* grad_y, const float* weight, const float* bias,
                       const float* save_mean, const float* save_invstd, int32_t relu, float* grad_x,
                       float* grad_residual, float* grad_weight, float* grad_bias, int32_t B,
                       int32_t C, int64_t HW, void* workspace, size_t workspace_bytes, void* stream);

/* ----------------------------------------------------------------- decode --
 * heat [B,C,H,W] (already activated), polys [B,2N,H,W], depth [B,1,H,W],
 * reg [B,2,H,W] or NULL (then +0.5).  K <= 256.
 * Outputs: dets [B,K,2N+7] = [x1,y1,x2,y2,score,cls,poly(2N),depth],
 * inds [B,K] int64 (flat y*W+x), clses [B,K] int32 (both may be NULL).
 * Selection order: score descending, ties by (class, flat index) ascending --
 * the order a stable sort of the reference's two-level top-k yields.
 */
size_t cp_polydet_decode_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W, int32_t K);
int cp_polydet_decode(const float* heat, const float* polys, const float* depth, const float* reg,
                      int32_t B, int32_t C, int32_t H, int32_t W, int32_t N2, int32_t K,
                      int32_t rep, float* dets, int64_t* inds, int32_t* clses, void* workspace,
                      size_t workspace_bytes, void* stream);
/* `--cat_spec_poly` (src/lib/models/decode.py:534-537): polys is [B, C * N2, H, W], one polygon of N2 numbers per
 * class, and a detection of class c takes channels c * N2 .. c * N2 + N2 - 1 (`polys.view(batch, K, cat,
 * nbr_points).gather(2, clses)`).  cat_spec_poly = 0 is cp_polydet_decode.  (The reference takes `nbr_points` from
 * `polys.shape[-1]` of the MAP, i.e. its width: its view only succeeds when C * W equals the channel count -- the
 * Python mirror keeps that precondition, the kernel does not need it.) */
int cp_polydet_decode_ex(const float* heat, const float* polys, const float* depth, const float* reg,
                         int32_t B, int32_t C, int32_t H, int32_t W, int32_t N2, int32_t K,
                         int32_t rep, int32_t cat_spec_poly, float* dets, int64_t* inds, int32_t* clses,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------- focal loss --
 * forward: hm[n] <- clamp(sigmoid(hm[n]), 1e-4, 1-1e-4) IN PLACE (the reference
 * mutates output['hm'] the same way) and loss_out[0] <- CornerNet focal loss;
 * stats_out[0..2] <- {sum pos term, sum neg term, num_pos} (device floats).
 * backward: grad_logits[n] <- grad_loss[0] * dLoss/dlogit from the ACTIVATED hm.
 */
size_t cp_sigmoid_focal_workspace_bytes(int64_t n);
int cp_sigmoid_focal_forward(float* hm_inout, const float* gt, int64_t n, float* loss_out,
                             float* stats_out, void* workspace, size_t workspace_bytes,
                             void* stream);
int cp_sigmoid_focal_backward(const float* hm_act, const float* gt, int64_t n,
                              const float* stats, const float* grad_loss, float* grad_logits,
                              void* stream);

/* ----------------------------------------------------- gathered L1 losses --
 * feat [B,D,H,W]; ind [B,M] int64; mask [B,M] uint8; target [B,M,D].
 * pred[b,m,d] = feat[b,d,ind[b,m]] (no NHWC copy).
 * loss_out[0] = S / (sum(mask)*D + eps), S by `mode` (CP_L1_*).
 * pred_add [B,M,D] or NULL: constant added to pred before the loss (the order
 * term's in-place `angles += 2*3.14`, src/lib/models/losses.py:892-899).
 * backward scatter-ADDS grad_loss[0] * dS/dfeat / denom into grad_feat (caller zero-fills).
 */
int cp_gather_l1_forward(const float* feat, const int64_t* ind, const uint8_t* mask,
                         const float* target, const float* pred_add, int32_t B, int32_t D,
                         int32_t H, int32_t W, int32_t M, int32_t mode, float eps,
                         float* loss_out, void* stream);
int cp_gather_l1_backward(const float* feat, const int64_t* ind, const uint8_t* mask,
                          const float* target, const float* pred_add, int32_t B, int32_t D,
                          int32_t H, int32_t W, int32_t M, int32_t mode, float eps,
                          const float* grad_loss, float* grad_feat, void* stream);

/* `--mse_loss` (src/lib/trains/polydet.py:23,44-46,84): torch.nn.MSELoss() between the RAW heat-map head (no
 * sigmoid) and the target, loss = mean((x - gt)^2); backward grad_x = 2 (x - gt) / n * grad_loss[0]. */
size_t cp_mse_workspace_bytes(void);
int cp_mse_forward(const float* x, const float* gt, int64_t n, float* loss_out, void* workspace,
                   size_t workspace_bytes, void* stream);
int cp_mse_backward(const float* x, const float* gt, int64_t n, const float* grad_loss, float* grad_x,
                    void* stream);

/* `--dense_poly` loss (src/lib/trains/polydet.py:107-110): torch.nn.L1Loss(reduction='sum')(pred * mask, target * mask)
 * / (mask.sum() + eps) over whole maps of n elements.  forward: out2[0] = loss, out2[1] = mask.sum() + eps (the
 * denominator the backward needs); backward: grad_pred = sign(pred * mask - target * mask) * mask * grad_loss[0] / den[0]. */
size_t cp_dense_l1_workspace_bytes(void);
int cp_dense_l1_forward(const float* pred, const float* target, const float* mask, int64_t n, float eps, float* out2,
                        void* workspace, size_t workspace_bytes, void* stream);
int cp_dense_l1_backward(const float* pred, const float* target, const float* mask, int64_t n, const float* den,
                         const float* grad_loss, float* grad_pred, void* stream);

/* ----------------------------------- polygon IoU (Weiler-Atherton) + order --
 * feat = polygon head [B,2N,H,W]; per masked object the literal reference
 * computation (every point read as (r, theta), doubled first shoelace term,
 * early-terminating traversal).  flags: bit0 = IoU term, bit1 = order term.
 *   iou_loss_out[0]   = 1 - sum(iou) / (sum(mask) + 1e-6)          (bit0)
 *   order_loss_out[0] = sum(hinge) / (10*sum(mask) + 1e-4)         (bit1)
 *   pred_add_out [B,M,2N] (bit1): the +2*3.14 edits, to feed cp_gather_l1_*.
 * backward scatter-ADDS into grad_feat (caller zero-fills).
 */
size_t cp_poly_iou_order_workspace_bytes(int32_t B, int32_t M, int32_t N);
int cp_poly_iou_order_forward(const float* feat, const int64_t* ind, const uint8_t* mask,
                              const float* target, int32_t B, int32_t N, int32_t H, int32_t W,
                              int32_t M, int32_t flags, float* iou_loss_out,
                              float* order_loss_out, float* pred_add_out, void* workspace,
                              size_t workspace_bytes, void* stream);
int cp_poly_iou_order_backward(const float* feat, const int64_t* ind, const uint8_t* mask,
                               const float* target, int32_t B, int32_t N, int32_t H, int32_t W,
                               int32_t M, int32_t flags, const float* grad_iou_loss,
                               const float* grad_order_loss, float* grad_feat, void* workspace,
                               size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CENTERPOLY_HIP_H */
