"""`DCN` -- the plugin slot the reference imports as `from .DCNv2.dcn_v2 import DCN`
(src/lib/models/networks/pose_dla_dcn.py:16, constructed at :354 and at
src/lib/models/networks/resnet_dcn.py:221).  Upstream (CharlesShang/DCNv2) is
absent from the reference tree; parameter names (`weight`, `bias`,
`conv_offset_mask.{weight,bias}`) are fixed by the published checkpoints.

forward = conv_offset_mask (dense 3x3 conv) -> ONE fused HIP kernel that reads the
raw 27-channel tensor (offsets + mask logits, sigmoid applied in-kernel), samples,
modulates and contracts on the matrix cores.  No chunk/cat/sigmoid/im2col tensors.
"""
import math
import os

import torch
import torch.nn.functional as F
import torch.nn as nn
from torch.nn.modules.utils import _pair

from .... import _C
from ..conv3x3 import conv3x3_infer, conv_bias_act, conv_raw


def _shape(x, weight, stride, pad, dil, dg):
    s = _C.DcnShape()
    s.B, s.Cin, s.H, s.W = x.shape
    s.Cout, _, s.kh, s.kw = weight.shape
    s.stride, s.pad, s.dil, s.deformable_groups = stride, pad, dil, dg
    return s


_AUTO = {}


def auto_contraction(s):
    """"auto": split-bf16 wherever the library runs its LDS-region kernel (the large maps, any channel count) and for the
    layers with more than 64 output channels (gather kernel bound by the f32 matrix pipe); exact f32 for the rest
    (small 64-channel maps: bound by their gathers, three waves per SIMD only in the f32 kernel)."""
    key = (s.B, s.Cin, s.H, s.W, s.Cout, s.kh, s.kw, s.stride, s.pad, s.dil, s.deformable_groups)
    c = _AUTO.get(key)
    if c is None:
        region = _C.lib().cp_dcn_v2_forward_kernel(s, _C.DCN_CONTRACTION["bf16x3"]) == 2
        c = _AUTO[key] = "bf16x3" if (region or s.Cout > 64) else "f32"
    return c


def dcn_v2_forward_raw(x, om, weight, bias, stride=1, pad=1, dil=1, dg=1, ep_scale=None,
                       ep_shift=None, relu=False, contraction="f32", owner=None):
    """Forward on the raw offset/mask tensor `om` [B, 3*kh*kw, Ho, Wo] (mask as logits).
    Optional fused per-channel epilogue out = act(acc*ep_scale + ep_shift).
    `owner` (inference, contraction "bf16x3"): a module on which the workspace -- whose head holds the split,
    permuted weights -- is kept for as long as `weight` is the same unmodified tensor and the shape the same, so
    that the permutation prologue runs once instead of per call."""
    L = _C.lib()
    s = _shape(x, weight, stride, pad, dil, dg)
    K = s.kh * s.kw
    Ho, Wo = om.shape[2], om.shape[3]
    out = torch.empty((s.B, s.Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    bs = 3 * K * Ho * Wo
    mask_ptr = _C.c_void_p(om.data_ptr() + 4 * 2 * K * Ho * Wo)
    nws = L.cp_dcn_v2_forward_workspace_bytes(s)
    if contraction == "auto":
        contraction = auto_contraction(s)
    mode = _C.DCN_CONTRACTION[contraction]
    ws = None
    if owner is not None and mode in (1, 3):
        # (data_ptr / device in the key: `module.to(device)` or a `.data` swap re-points the Parameter's storage without
        # a version bump; a stale workspace would then hold another tensor's weights, possibly on another device)
        key = (weight._version, weight.data_ptr(), str(weight.device), mode, tuple(x.shape), tuple(om.shape))
        cache = owner.__dict__.get("_dcn_fwd_ws")
        if cache is not None and cache[0] is weight and cache[1] == key:
            ws, mode = cache[2], mode + 1                # CP_DCN_BF16X3[_REGION]_PREPARED
        else:
            ws = _C.workspace(nws, x.device)
            owner.__dict__["_dcn_fwd_ws"] = (weight, key, ws)
    elif nws:
        ws = _C.workspace(nws, x.device)
    timer = _C.kernel_timer
    end = timer.start(("dcn_fwd", s.Cin, s.Cout, Ho, Wo, s.B)) if timer is not None else None
    rc = L.cp_dcn_v2_forward(s, _C.ptr(x), _C.ptr(om), bs, mask_ptr, bs, 1, _C.ptr(weight),
                             _C.ptr(bias), _C.ptr(ep_scale), _C.ptr(ep_shift), 1 if relu else 0,
                             mode, _C.ptr(out), _C.ptr(ws), nws, _C.stream())
    if end is not None:
        end.record()
    _C.check(rc, "cp_dcn_v2_forward")
    return out


def dcn_v2_module_forward(x, om_weight, om_bias, weight, bias, ep_scale=None, ep_shift=None, relu=False, owner=None,
                          want_om=False):
    """The whole DCN module in one launch (cp_dcn_v2_forward_fused): conv_offset_mask computed inside the DCN kernel.
    Returns (out, om or None), or None where the library does not run its region kernel for this shape (the caller
    then runs the convolution and dcn_v2_forward_raw).  `owner` keeps the workspace with both permuted weight sets."""
    L = _C.lib()
    s = _shape(x, weight, 1, 1, 1, 1)
    if tuple(weight.shape[2:]) != (3, 3) or tuple(om_weight.shape) != (27, s.Cin, 3, 3) \
            or not L.cp_dcn_v2_forward_fused_supported(s):
        return None
    nws = L.cp_dcn_v2_forward_fused_workspace_bytes(s)
    prepared, ws = 0, None
    if owner is not None:
        key = (weight._version, weight.data_ptr(), om_weight._version, om_weight.data_ptr(), str(weight.device), tuple(x.shape))
        cache = owner.__dict__.get("_dcn_fused_ws")
        if cache is not None and cache[0] is weight and cache[1] is om_weight and cache[2] == key:
            ws, prepared = cache[3], 1
        else:
            ws = _C.workspace(nws, x.device)
            owner.__dict__["_dcn_fused_ws"] = (weight, om_weight, key, ws)
    else:
        ws = _C.workspace(nws, x.device)
    out = torch.empty((s.B, s.Cout, s.H, s.W), dtype=torch.float32, device=x.device)
    om = torch.empty((s.B, 27, s.H, s.W), dtype=torch.float32, device=x.device) if want_om else None
    timer = _C.kernel_timer
    end = timer.start(("dcn_fwd", s.Cin, s.Cout, s.H, s.W, s.B)) if timer is not None else None
    rc = L.cp_dcn_v2_forward_fused(s, _C.ptr(x), _C.ptr(om_weight), _C.ptr(om_bias), _C.ptr(weight), _C.ptr(bias),
                                   _C.ptr(ep_scale), _C.ptr(ep_shift), 1 if relu else 0, prepared, _C.ptr(om), _C.ptr(out),
                                   _C.ptr(ws), nws, _C.stream())
    if end is not None:
        end.record()
    _C.check(rc, "cp_dcn_v2_forward_fused")
    return out, om


class _DCNv2Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, om, weight, bias, stride, pad, dil, dg):
        for t in (x, om, weight):
            if t.dtype != torch.float32:
                raise _C.NativeError("DCN expects float32")
        x, om, weight = x.contiguous(), om.contiguous(), weight.contiguous()
        ctx.cfg = (stride, pad, dil, dg)
        ctx.save_for_backward(x, om, weight)
        ctx.has_bias = bias is not None
        return dcn_v2_forward_raw(x, om, weight, bias, stride, pad, dil, dg, contraction=DCN.train_contraction)

    @staticmethod
    def backward(ctx, grad_out):
        x, om, weight = ctx.saved_tensors
        stride, pad, dil, dg = ctx.cfg
        L = _C.lib()
        s = _shape(x, weight, stride, pad, dil, dg)
        K = s.kh * s.kw
        Ho, Wo = om.shape[2], om.shape[3]
        grad_out = grad_out.contiguous()
        gx = torch.empty_like(x)                         # (overwritten: the library zero-fills it itself)
        gom = torch.empty_like(om)
        gw = _C.zeros(weight.shape, weight.device)
        gb = _C.zeros((s.Cout,), x.device)
        bs = 3 * K * Ho * Wo
        off_m = 4 * 2 * K * Ho * Wo
        ws = _C.workspace(L.cp_dcn_v2_backward_workspace_bytes(s), x.device)
        gmask_ptr = _C.c_void_p(gom.data_ptr() + off_m)

        def call(data, wgt, bias_):
            rc = L.cp_dcn_v2_backward(s, _C.ptr(x), _C.ptr(om), bs, _C.c_void_p(om.data_ptr() + off_m),
                                      bs, 1, _C.ptr(weight), _C.ptr(grad_out),
                                      _C.ptr(gx) if data else None, _C.ptr(gom) if data else None, bs,
                                      gmask_ptr if data else None, bs, _C.ptr(gw) if wgt else None,
                                      _C.ptr(gb) if bias_ else None, DCN.backward_flags, _C.ptr(ws), ws.numel(),
                                      _C.stream())
            _C.check(rc, "cp_dcn_v2_backward")

        timer = _C.kernel_timer
        if timer is None:
            call(True, True, True)
        else:
            # bench.py's roofline_bwd: the same entry point called once per gradient group so the
            # data and weight kernels get their own HIP-event brackets (same kernels, same inputs)
            key = (s.Cin, s.Cout, Ho, Wo, s.B)
            for tag, sel in (("dcn_bwd_data", (True, False, False)), ("dcn_bwd_weight", (False, True, False)),
                             ("dcn_bwd_bias", (False, False, True))):
                end = timer.start((tag,) + key)
                call(*sel)
                end.record()
        return gx, gom, gw, (gb if ctx.has_bias else None), None, None, None, None


def _dcn_backward(x, om, weight, grad_out, has_bias):
    """cp_dcn_v2_backward on the raw 27-channel tensor: (grad_x, grad_om, grad_weight, grad_bias)."""
    L = _C.lib()
    s = _shape(x, weight, 1, 1, 1, 1)
    K = 9
    Ho, Wo = om.shape[2], om.shape[3]
    grad_out = grad_out.contiguous()
    gx = torch.empty_like(x)                             # (overwritten: the library zero-fills it itself)
    gom = torch.empty_like(om)
    gw = _C.zeros(weight.shape, weight.device)
    gb = _C.zeros((s.Cout,), x.device)
    bs = 3 * K * Ho * Wo
    off_m = 4 * 2 * K * Ho * Wo
    ws = _C.workspace(L.cp_dcn_v2_backward_workspace_bytes(s), x.device)

    def call(data, wgt, bias_):
        rc = L.cp_dcn_v2_backward(s, _C.ptr(x), _C.ptr(om), bs, _C.c_void_p(om.data_ptr() + off_m),
                                  bs, 1, _C.ptr(weight), _C.ptr(grad_out),
                                  _C.ptr(gx) if data else None, _C.ptr(gom) if data else None, bs,
                                  _C.c_void_p(gom.data_ptr() + off_m) if data else None, bs, _C.ptr(gw) if wgt else None,
                                  _C.ptr(gb) if bias_ else None, DCN.backward_flags, _C.ptr(ws), ws.numel(),
                                  _C.stream())
        _C.check(rc, "cp_dcn_v2_backward")

    timer = _C.kernel_timer
    if timer is None:
        call(True, True, True)
    else:
        key = (s.Cin, s.Cout, Ho, Wo, s.B)
        for tag, sel in (("dcn_bwd_data", (True, False, False)), ("dcn_bwd_weight", (False, True, False)),
                         ("dcn_bwd_bias", (False, False, True))):
            end = timer.start((tag,) + key)
            call(*sel)
            end.record()
    return gx, gom, gw, (gb if has_bias else None)


class _DCNModuleFunction(torch.autograd.Function):
    """The whole DCN module (3x3, stride 1, pad 1) as ONE autograd node: x -> conv_offset_mask -> deformable convolution.
    Forward: cp_dcn_v2_forward_fused where the library fuses the offset convolution into the DCN kernel (it copies the
    27 channels out for the backward), else the convolution kernel + cp_dcn_v2_forward.  Backward: cp_dcn_v2_backward,
    then the convolution's gradients with the DCN's grad_x riding in the input-gradient kernel's epilogue -- one grad_x
    leaves the node, autograd has no `add` to run (16 full-map adds per step before)."""

    @staticmethod
    def forward(ctx, x, om_weight, om_bias, weight, bias):
        from .. import conv3x3
        x = x.contiguous()
        r = dcn_v2_module_forward(x, om_weight, om_bias, weight, bias, want_om=True) \
            if DCN.fuse_offset_conv and DCN.train_contraction in ("auto", "bf16x3") else None
        if r is not None:
            out, om = r
        else:
            cout, cin = om_weight.shape[0], om_weight.shape[1]
            om = conv3x3._launch(x, conv3x3._prepare(om_weight, cin, cout, False), om_bias, None, cout, False, 9)
            out = dcn_v2_forward_raw(x, om, weight, bias, 1, 1, 1, 1, contraction=DCN.train_contraction)
        ctx.save_for_backward(x, om, weight, om_weight)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from .. import conv3x3
        x, om, weight, om_weight = ctx.saved_tensors
        gx, gom, gw, gb = _dcn_backward(x, om, weight, grad_out, ctx.has_bias)
        B, C, H, W = gom.shape
        L = _C.lib()
        gb_om = _C.zeros((C,), gom.device)
        _C.check(L.cp_channel_sum_accumulate(_C.ptr(gom), _C.ptr(gb_om), B, C, H * W, _C.stream()), "cp_channel_sum_accumulate")
        _, gw_om = conv3x3.grads(x, om_weight, gom, want_x=False, want_w=True)
        cin = om_weight.shape[1]
        # input gradient of the offset convolution (the forward kernel over grad_om with the transposed, flipped weights)
        # + the DCN's own grad_x as the kernel's residual operand
        gx_all = conv3x3._launch(gom, conv3x3._prepare(om_weight, C, cin, True), None, gx, cin, False, 9)
        return gx_all, gw_om, gb_om, gw, gb


class _ConvBias(torch.autograd.Function):
    """y_raw + bias[c] in place on a convolution's raw output; the bias gradient is one
    segmented channel sum (cp_channel_sum_accumulate) instead of torch's generic reduction."""

    @staticmethod
    def forward(ctx, y_raw, bias):
        B, C, H, W = y_raw.shape
        _C.check(_C.lib().cp_bias_act_inplace(_C.ptr(y_raw), _C.ptr(bias), None, B, C, H * W, 0, _C.stream()),
                 "cp_bias_act_inplace")
        ctx.mark_dirty(y_raw)
        return y_raw

    @staticmethod
    def backward(ctx, go):
        go = go.contiguous()
        B, C, H, W = go.shape
        gb = _C.zeros((C,), go.device)
        _C.check(_C.lib().cp_channel_sum_accumulate(_C.ptr(go), _C.ptr(gb), B, C, H * W, _C.stream()),
                 "cp_channel_sum_accumulate")
        return go, gb


def conv_bias(conv, x):
    """conv with bias; in training on a HIP device the bias add / bias gradient are the fused
    in-place epilogue and one channel-sum kernel."""
    if x.is_cuda and conv.bias is not None and torch.is_grad_enabled() and x.dtype == torch.float32 \
            and conv.groups == 1:
        y = conv_bias_act(conv, x, False)               # bias in the convolution kernel's epilogue
        if y is not None:
            return y
        y = conv_raw(conv, x)
        if y.is_contiguous() and (y.shape[2] * y.shape[3]) % 4 == 0 and y.shape[0] * y.shape[1] <= 65535:
            return _ConvBias.apply(y, conv.bias)
        return y + conv.bias.view(1, -1, 1, 1)
    return conv(x)


class DCN(nn.Module):
    """DCN(in_channels, out_channels, kernel_size, stride, padding, dilation=1,
    deformable_groups=1): forward(x[B,Cin,H,W]) -> [B,Cout,Ho,Wo]."""

    # contraction of the TRAINING forward ("auto" | "f32" | "bf16x3"), a class-wide option (opt.dcn_contraction)
    train_contraction = "auto"
    infer_contraction = "auto"  # default of forward_fused (prepare_inference(dcn_contraction=...) sets it per module)
    # flags of cp_dcn_v2_backward (0 = split-bf16 x3; _C.DCN_BWD_EXACT_F32 = exact fp32 chain), class-wide as well
    backward_flags = 0
    # conv_offset_mask inside the DCN kernel where the library's region kernel runs (cp_dcn_v2_forward_fused)
    fuse_offset_conv = True

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1,
                 deformable_groups=1):
        super(DCN, self).__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = stride if isinstance(stride, int) else stride[0]
        self.padding = padding if isinstance(padding, int) else padding[0]
        self.dilation = dilation if isinstance(dilation, int) else dilation[0]
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        kh, kw = self.kernel_size
        self.conv_offset_mask = nn.Conv2d(in_channels, deformable_groups * 3 * kh * kw,
                                          kernel_size=self.kernel_size, stride=self.stride,
                                          padding=self.padding, bias=True)
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        stdv = 1.0 / math.sqrt(n)
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)
            self.bias.zero_()
            # zero offsets / mask logits at construction: DCN(x) == 0.5*conv(x, W) + b
            self.conv_offset_mask.weight.zero_()
            self.conv_offset_mask.bias.zero_()

    def forward(self, x):
        cm = self.conv_offset_mask
        if x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and self.stride == 1 and self.padding == 1 \
                and self.dilation == 1 and self.deformable_groups == 1 and self.kernel_size == (3, 3) and cm.bias is not None:
            from .. import conv3x3
            if conv3x3.usable(cm, x) and _C.lib().cp_conv3x3_mfma_supported(27, x.shape[1], x.shape[2], x.shape[3]):
                return _DCNModuleFunction.apply(x, cm.weight, cm.bias, self.weight, self.bias)
        om = conv_bias(cm, x)
        return _DCNv2Function.apply(x, om, self.weight, self.bias, self.stride, self.padding,
                                    self.dilation, self.deformable_groups)

    def forward_fused(self, x, ep_scale, ep_shift, relu=True):
        """Inference: DCN + per-channel affine (folded BatchNorm, bias included) + ReLU
        in the kernel's epilogue (replaces DeformConv.forward's three passes)."""
        cm = self.conv_offset_mask
        con = getattr(self, "contraction", None) or DCN.infer_contraction
        if DCN.fuse_offset_conv and con in ("auto", "bf16x3") and x.is_cuda and x.dtype == torch.float32 \
                and self.stride == 1 and self.padding == 1 and self.dilation == 1 and self.deformable_groups == 1:
            r = dcn_v2_module_forward(x.contiguous(), cm.weight, cm.bias, self.weight, None, ep_scale, ep_shift, relu,
                                      owner=self)
            if r is not None:
                return r[0]
        om = conv3x3_infer(x, cm, cm.weight, cm.bias, conv=cm)      # split-bf16 MFMA kernel, bias in its epilogue
        if om is None:
            om = cm(x)
        return dcn_v2_forward_raw(x.contiguous(), om.contiguous(), self.weight, None, self.stride,
                                  self.padding, self.dilation, self.deformable_groups, ep_scale,
                                  ep_shift, relu, getattr(self, "contraction", None) or DCN.infer_contraction, owner=self)
