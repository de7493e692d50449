"""DLA-34 backbone with DCNv2 up-sampling and the polydet heads.

Same public surface and the same state_dict key grammar as the reference's
src/lib/models/networks/pose_dla_dcn.py (DLA :225-308, BasicBlock :32-60, Root
:148-166, Tree :169-222, DeformConv :347-359, IDAUp :362-387, DLAUp :390-413,
DLASeg :427-482, get_pose_net :485-492), so published checkpoints load
unchanged.  Differences, all deliberate:
  * `pretrained` defaults to False and never touches the network (the reference
    forces an ImageNet URL fetch, :485-491);
  * in eval mode DeformConv runs DCN + BatchNorm + ReLU as ONE kernel (BN folded
    into the DCN epilogue) instead of three passes over the feature map.
"""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ... import _C
from .DCNv2.dcn_v2 import DCN, conv_bias
from . import conv3x3
from .conv3x3 import conv3x3_infer, conv_infer, conv_raw

# A/B switches of this module (tools/probe_*.py set them; nothing reads the environment)
HEADS_FUSED = True           # all heads of a stage as one kernel (cp_heads_fused_forward)
CONV_DIRECT_WGRAD = True     # weight gradient of the full-resolution base layers from cp_conv_direct_wgrad

BN_MOMENTUM = 0.1


def _bn(c):
    return nn.BatchNorm2d(c, momentum=BN_MOMENTUM)


# ---- inference-time BatchNorm folding --------------------------------------------------
# prepare_inference() caches, per (conv, bn) pair, the conv weight scaled by
# gamma/sqrt(var+eps) and the matching bias, so eval-mode forward runs conv(+bias)+ReLU with
# no BatchNorm pass; DeformConv gets the same affine in the DCN kernel's epilogue.  The cache
# is dropped by train(); checkpoints / state_dict are untouched.

def _fold_conv_bn(conv, bn):
    with torch.no_grad():
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        w = (conv.weight * scale.view(-1, 1, 1, 1)).contiguous()
        b = bn.bias - bn.running_mean * scale
        if conv.bias is not None:
            b = b + conv.bias * scale
    return w, b.contiguous()


def _use_folded(m):
    return getattr(m, "_folded", None) is not None and not m.training and not torch.is_grad_enabled()


def _conv_direct(x, conv, wb, relu):
    """The hand-written direct convolution (cp_conv_direct_forward) for the full-resolution,
    low-channel layers, folded-BN shift and ReLU in its epilogue; None when the shape is not one of its."""
    kh, kw = conv.kernel_size
    if not (x.is_cuda and x.dtype == torch.float32 and kh == kw and conv.groups == 1 and conv.dilation == (1, 1)
            and conv.stride[0] == conv.stride[1] and conv.padding[0] == conv.padding[1]):
        return None
    L = _C.lib()
    if not L.cp_conv_direct_supported(conv.in_channels, conv.out_channels, kh, conv.stride[0], conv.padding[0]):
        return None
    x = x.contiguous()
    B, _, H, W = x.shape
    Ho = (H + 2 * conv.padding[0] - kh) // conv.stride[0] + 1
    Wo = (W + 2 * conv.padding[0] - kh) // conv.stride[0] + 1
    out = torch.empty((B, conv.out_channels, Ho, Wo), dtype=torch.float32, device=x.device)
    # (round 4: under the split-bf16 arithmetic level0 / level1 contract on the bf16 matrix cores like every other layer)
    rc = L.cp_conv_direct_forward_ex(_C.ptr(x), _C.ptr(wb[0]), _C.ptr(wb[1]), _C.ptr(out), B, conv.in_channels, H, W,
                                     conv.out_channels, kh, conv.stride[0], conv.padding[0], 1 if relu else 0,
                                     1 if conv3x3.mfma_enabled() else 0, _C.stream())
    if rc == -2:                                    # CP_EUNSUPPORTED (tensor too large for 32-bit offsets)
        return None
    _C.check(rc, "cp_conv_direct_forward_ex")
    return out


class _DirectConvFn(torch.autograd.Function):
    """Training forward of a bias-free convolution of the DLA base through the direct kernel; weight gradient
    from cp_conv_direct_wgrad, level0's and level1's input gradients from the MFMA convolution."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad):
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad)
        B, cin, H, W = x.shape
        cout, _, k, _ = weight.shape
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        out = torch.empty((B, cout, Ho, Wo), dtype=torch.float32, device=x.device)
        _C.check(_C.lib().cp_conv_direct_forward_ex(_C.ptr(x), _C.ptr(weight), None, _C.ptr(out), B, cin, H, W, cout, k,
                                                    stride, pad, 0, 1 if conv3x3.mfma_enabled() else 0, _C.stream()),
                 "cp_conv_direct_forward_ex")
        return out

    @staticmethod
    def backward(ctx, go):
        x, weight = ctx.saved_tensors
        stride, pad = ctx.cfg
        go = go.contiguous()
        direct_w = CONV_DIRECT_WGRAD
        gx = gw = None
        if weight.shape[2] == 3 and stride == 1 and pad == 1 and conv3x3.mfma_enabled():
            # level0 (16 -> 16 at full resolution): the input gradient through the split-bf16 MFMA kernel (most of
            # its 32-channel tiles multiply zeros, but the layer is bandwidth-bound and the library's NHWC round
            # trip costs 5x more); the weight gradient from the direct kernel below
            gx, gw = conv3x3.grads(x, weight, go, ctx.needs_input_grad[0], ctx.needs_input_grad[1] and not direct_w)
        elif ctx.needs_input_grad[0]:
            if weight.shape[2] == 3 and stride == 2 and pad == 1 and conv3x3.mfma_enabled():
                gx = conv3x3.s2_input_grad(tuple(x.shape), weight, go)     # level1 (16 -> 32, stride 2): one MFMA launch
            if gx is None:
                gx = torch.nn.grad.conv2d_input(x.shape, weight, go, stride=stride, padding=pad)
        if ctx.needs_input_grad[1] and gw is None:
            L = _C.lib()
            B, cin, H, W = x.shape
            cout, k = weight.shape[0], weight.shape[2]
            if direct_w and L.cp_conv_direct_wgrad_supported(cin, cout, k, stride, pad):
                # pixel-contraction kernel on the exact f32 MFMA instead of the library's NHWC implicit GEMM behind
                # two full-resolution layout transposes
                gw = _C.zeros(weight.shape, weight.device)
                rc = L.cp_conv_direct_wgrad(_C.ptr(x), _C.ptr(go), _C.ptr(gw), B, cin, H, W, cout, k, stride, pad,
                                            _C.stream())
                if rc == -2:
                    gw = None
                else:
                    _C.check(rc, "cp_conv_direct_wgrad")
            if gw is None:
                gw = torch.nn.grad.conv2d_weight(x, weight.shape, go, stride=stride, padding=pad)
        return gx, gw, None, None


def conv_train(conv, x):
    """conv(x) in training: the direct kernel's forward for the shapes it has, the library otherwise."""
    kh, kw = conv.kernel_size
    if (x.is_cuda and x.dtype == torch.float32 and conv.bias is None and kh == kw and conv.groups == 1
            and conv.dilation == (1, 1) and conv.stride[0] == conv.stride[1] and conv.padding[0] == conv.padding[1]
            and x.shape[1] * x.shape[2] * x.shape[3] * 4 < 0x70000000
            and _C.lib().cp_conv_direct_supported(conv.in_channels, conv.out_channels, kh, conv.stride[0],
                                                  conv.padding[0])):
        return _DirectConvFn.apply(x.contiguous(), conv.weight, conv.stride[0], conv.padding[0])
    if conv.bias is None:
        return conv_raw(conv, x)                    # split-bf16 MFMA kernel for 3x3 / stride 1, else the library
    return conv(x)


def _base_pair(dla, x):
    """level0 + level1 (folded) as one launch of cp_dla_base_pair_forward -- the 16-channel full-resolution map between
    them stays in LDS; None when the layers / the map are not the kernel's (the caller runs them one by one)."""
    c0, c1 = dla.level0[0], dla.level1[0]
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv3x3.mfma_enabled()
            and (c0.in_channels, c0.out_channels, c0.kernel_size, c0.stride, c0.padding, c0.dilation, c0.groups)
            == (16, 16, (3, 3), (1, 1), (1, 1), (1, 1), 1)
            and (c1.in_channels, c1.out_channels, c1.kernel_size, c1.stride, c1.padding, c1.dilation, c1.groups)
            == (16, 32, (3, 3), (2, 2), (1, 1), (1, 1), 1)):
        return None
    B, _, H, W = x.shape
    L = _C.lib()
    if not L.cp_dla_base_pair_supported(H, W) or B > 65535:
        return None
    (w0, b0), (w1, b1) = dla._folded[1], dla._folded[2]
    x = x.contiguous()
    out = torch.empty((B, 32, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device)
    end = _C.kernel_timer.start(("conv_base_pair", 16, 32, H, W, B)) if _C.kernel_timer is not None else None
    rc = L.cp_dla_base_pair_forward(_C.ptr(x), _C.ptr(w0.contiguous()), _C.ptr(b0), _C.ptr(w1.contiguous()), _C.ptr(b1),
                                    _C.ptr(out), B, H, W, _C.stream())
    if rc == -2:
        return None
    _C.check(rc, "cp_dla_base_pair_forward")
    if end is not None:
        end.record()
    return out


def _conv_folded(x, conv, wb, relu=False, residual=None):
    """conv with folded-BN weights, then ONE fused in-place pass: + bias (+ residual) (+ ReLU)."""
    if residual is None:
        y = None
        if conv.kernel_size == (7, 7) and x.is_cuda:       # 3-channel 7x7: the bf16 matrix-core kernel (split_bf16 arithmetic)
            y = conv3x3.stem_infer(x, conv, wb[0], wb[1], relu)
        if y is None:
            y = _conv_direct(x, conv, wb, relu)
        if y is not None:
            return y
    if x.is_cuda:
        y = conv3x3_infer(x, conv, wb[0], wb[1], residual, relu, conv=conv)
        if y is not None:
            return y
    if not x.is_cuda:
        y = F.conv2d(x, wb[0], wb[1], conv.stride, conv.padding, conv.dilation, conv.groups)
        if residual is not None:
            y = y + residual
        return F.relu_(y) if relu else y
    y = F.conv2d(x, wb[0], None, conv.stride, conv.padding, conv.dilation, conv.groups)
    B, C, H, W = y.shape
    if not y.is_contiguous() or (H * W) % 4 != 0 or (residual is not None and not residual.is_contiguous()):
        y = y + wb[1].view(1, -1, 1, 1)
        if residual is not None:
            y = y + residual
        return F.relu_(y) if relu else y
    rc = _C.lib().cp_bias_act_inplace(_C.ptr(y), _C.ptr(wb[1]), _C.ptr(residual), B, C, H * W,
                                      1 if relu else 0, _C.stream())
    _C.check(rc, "cp_bias_act_inplace")
    return y


# ---- 2x2 / stride 2 max pooling (the trees' `downsample`) --------------------------------------------
class _MaxPool2x2Fn(torch.autograd.Function):
    """MaxPool2d(2, 2) through cp_maxpool2x2_*: no index tensor; the backward recomputes the arg-max from x (torch's
    tie rule) and writes every element of the gradient."""

    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        out = torch.empty((B, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
        _C.check(_C.lib().cp_maxpool2x2_forward(_C.ptr(x), _C.ptr(out), B, C, H, W, _C.stream()), "cp_maxpool2x2_forward")
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, go):
        (x,) = ctx.saved_tensors
        B, C, H, W = x.shape
        gx = torch.empty_like(x)
        _C.check(_C.lib().cp_maxpool2x2_backward(_C.ptr(x), _C.ptr(go.contiguous()), _C.ptr(gx), B, C, H, W,
                                                 _C.stream()), "cp_maxpool2x2_backward")
        return gx


def downsample2(pool, x):
    """pool(x) for the trees' MaxPool2d(2, 2): the HIP kernels on a device tensor, the module otherwise."""
    if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and pool.kernel_size in (2, (2, 2))
            and pool.stride in (2, (2, 2)) and pool.padding in (0, (0, 0)) and not pool.ceil_mode
            and x.shape[0] * x.shape[1] <= 65535 and x.shape[2] >= 2 and x.shape[3] >= 2):
        return _MaxPool2x2Fn.apply(x.contiguous())
    return pool(x)


# ---- training-time fused conv-bias + ReLU (the heads' Conv2d(3x3, bias) -> ReLU) -----------
class _BiasRelu(torch.autograd.Function):
    """y = relu(y_raw + bias[c]) in place on the convolution's raw output; backward is one pass
    (mask + bias reduction) instead of threshold_backward and a separate bias sum."""

    @staticmethod
    def forward(ctx, y_raw, bias):
        B, C, H, W = y_raw.shape
        _C.check(_C.lib().cp_bias_act_inplace(_C.ptr(y_raw), _C.ptr(bias), None, B, C, H * W, 1, _C.stream()),
                 "cp_bias_act_inplace")
        ctx.mark_dirty(y_raw)
        ctx.save_for_backward(y_raw)
        return y_raw

    @staticmethod
    def backward(ctx, go):
        (y,) = ctx.saved_tensors
        go = go.contiguous()
        B, C, H, W = y.shape
        g = torch.empty_like(go)
        gb = _C.zeros((C,), y.device)
        _C.check(_C.lib().cp_bias_relu_backward(_C.ptr(y), _C.ptr(go), _C.ptr(g), _C.ptr(gb), B, C, H * W,
                                                _C.stream()), "cp_bias_relu_backward")
        return g, gb


def conv_bias_relu(conv, x):
    """conv (with bias) followed by ReLU: fused epilogue on a HIP device in training."""
    if (x.is_cuda and conv.bias is not None and x.dtype == torch.float32 and torch.is_grad_enabled()
            and conv.groups == 1):
        y = conv3x3.conv_bias_act(conv, x, True)        # bias + ReLU in the convolution kernel's epilogue
        if y is not None:
            return y
        y = conv_raw(conv, x)
        if y.is_contiguous() and (y.shape[2] * y.shape[3]) % 4 == 0 and y.shape[0] * y.shape[1] <= 65535:
            return _BiasRelu.apply(y, conv.bias)
        return F.relu(y + conv.bias.view(1, -1, 1, 1))
    return F.relu(conv(x))


# ---- training-time fused BatchNorm (+ residual) (+ ReLU) -------------------------------
class _BnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, momentum, eps, relu):
        L = _C.lib()
        x = x.contiguous()
        res = residual.contiguous() if residual is not None else None
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = _C.workspace(L.cp_bn_workspace_bytes(B, C, H * W), x.device)
        rc = L.cp_bn_act_forward_train(_C.ptr(x), _C.ptr(weight), _C.ptr(bias), _C.ptr(res), _C.ptr(y),
                                       _C.ptr(mean), _C.ptr(invstd), _C.ptr(running_mean),
                                       _C.ptr(running_var), momentum, eps, 1 if relu else 0, B, C,
                                       H * W, _C.ptr(ws), ws.numel(), _C.stream())
        _C.check(rc, "cp_bn_act_forward_train")
        # ReLU without a residual: the backward recomputes the mask from x (y is neither saved here nor read there)
        recompute = relu and residual is None
        ctx.save_for_backward(x, None if recompute else y, weight, bias, mean, invstd)
        ctx.cfg = (relu, residual is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, weight, bias, mean, invstd = ctx.saved_tensors
        relu, has_res = ctx.cfg
        L = _C.lib()
        gy = gy.contiguous()
        B, C, H, W = x.shape
        gx = torch.empty_like(x)
        gres = torch.empty_like(x) if has_res else None
        gw = torch.empty(C, dtype=torch.float32, device=x.device)       # overwritten by the kernel
        gb = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = _C.workspace(L.cp_bn_workspace_bytes(B, C, H * W), x.device)
        rc = L.cp_bn_act_backward(_C.ptr(x), _C.ptr(y), _C.ptr(gy), _C.ptr(weight), _C.ptr(bias), _C.ptr(mean),
                                  _C.ptr(invstd), 1 if relu else 0, _C.ptr(gx), _C.ptr(gres), _C.ptr(gw),
                                  _C.ptr(gb), B, C, H * W, _C.ptr(ws), ws.numel(), _C.stream())
        _C.check(rc, "cp_bn_act_backward")
        return gx, gw, gb, gres, None, None, None, None, None


# BatchNorm2d.num_batches_tracked of the modules that took the fused path: incremented together by ONE multi-tensor
# launch at the end of the network's forward (flush_batch_counts) instead of ~55 four-microsecond kernels per step.
# Flushed in a `finally` of DLASeg.forward / exkp.forward (a forward that raises part-way leaves nothing pending), at 256
# entries, and before state_dict() of those networks; a BatchNorm that runs twice in one forward is counted twice.
_PENDING_BATCH_COUNTS = {}         # id(buffer) -> [buffer, increments]


def flush_batch_counts():
    if _PENDING_BATCH_COUNTS:
        once = [t for t, n in _PENDING_BATCH_COUNTS.values() if n == 1]
        more = [(t, n) for t, n in _PENDING_BATCH_COUNTS.values() if n != 1]
        _PENDING_BATCH_COUNTS.clear()
        with torch.no_grad():
            if once:
                torch._foreach_add_(once, 1)
            for t, n in more:
                t.add_(n)


def bn_act(bn, x, relu=True, residual=None):
    """relu(bn(x) + residual).  Training on a HIP device: one fused statistics pass + one fused
    apply pass (and two passes backward) instead of BatchNorm, add and ReLU kernels; otherwise
    the plain torch modules (eval mode normally takes the folded path before getting here)."""
    if bn.training and x.is_cuda and x.dtype == torch.float32 and bn.track_running_stats \
            and bn.affine and bn.momentum is not None and x.numel() // x.shape[1] > 1:
        e = _PENDING_BATCH_COUNTS.setdefault(id(bn.num_batches_tracked), [bn.num_batches_tracked, 0])
        e[1] += 1
        if len(_PENDING_BATCH_COUNTS) >= 256:
            flush_batch_counts()
        return _BnAct.apply(x, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var,
                            float(bn.momentum), float(bn.eps), relu)
    y = bn(x)
    if residual is not None:
        y = y + residual
    return F.relu_(y) if relu else y


class BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn1 = _bn(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, dilation, dilation, bias=False)
        self.bn2 = _bn(planes)
        self.stride = stride

    def fold(self):
        self._folded = (_fold_conv_bn(self.conv1, self.bn1), _fold_conv_bn(self.conv2, self.bn2))

    def forward(self, x, residual=None):
        skip = x if residual is None else residual
        if _use_folded(self):
            z = conv3x3.block_infer(x, self.conv1, self._folded[0], self.conv2, self._folded[1], skip)
            if z is not None:                       # the intermediate as split bf16 planes (bit-identical)
                return z
            y = _conv_folded(x, self.conv1, self._folded[0], relu=True)
            return _conv_folded(y, self.conv2, self._folded[1], relu=True, residual=skip)
        pair = conv3x3.conv_raw_skip(self.conv1, x) if (residual is None and x.is_cuda and self.training) else None
        if pair is not None:                        # the skip starts at conv1's input: its gradient joins in conv1's kernel
            y, skip = bn_act(self.bn1, pair[0], relu=True), pair[1]
        else:
            y = bn_act(self.bn1, conv_train(self.conv1, x), relu=True)
        return bn_act(self.bn2, conv_train(self.conv2, y), relu=True, residual=skip)


class Root(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, residual):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, 1, 1, (kernel_size - 1) // 2, bias=False)
        self.bn = _bn(out_channels)
        self.relu = nn.ReLU(inplace=True)
        self.residual = residual

    def fold(self):
        self._folded = _fold_conv_bn(self.conv, self.bn)

    def forward(self, *xs):
        if _use_folded(self):
            res = xs[0] if self.residual else None
            if xs[0].is_cuda:                       # 1x1 over the inputs in place: no concatenated copy
                y = conv_infer(list(xs), self.conv, self._folded[0], self._folded[1], res, True, conv=self.conv)
                if y is not None:
                    return y
            return _conv_folded(torch.cat(xs, 1), self.conv, self._folded, relu=True, residual=res)
        y = conv3x3.concat_conv1x1(self.conv, xs) if (self.training and xs[0].is_cuda) else None
        if y is None:
            y = conv_train(self.conv, torch.cat(xs, 1))
        return bn_act(self.bn, y, relu=True, residual=xs[0] if self.residual else None)


class Tree(nn.Module):
    """Hierarchical aggregation node: two sub-trees (or two blocks at the leaves)
    whose outputs, plus optional passed-down children, meet in a 1x1 Root."""

    def __init__(self, levels, block, in_channels, out_channels, stride=1, level_root=False,
                 root_dim=0, root_kernel_size=1, dilation=1, root_residual=False):
        super().__init__()
        root_dim = root_dim or 2 * out_channels
        if level_root:
            root_dim += in_channels
        self.levels, self.level_root, self.root_dim = levels, level_root, root_dim
        if levels == 1:
            self.tree1 = block(in_channels, out_channels, stride, dilation=dilation)
            self.tree2 = block(out_channels, out_channels, 1, dilation=dilation)
            self.root = Root(root_dim, out_channels, root_kernel_size, root_residual)
        else:
            kw = dict(root_kernel_size=root_kernel_size, dilation=dilation,
                      root_residual=root_residual)
            self.tree1 = Tree(levels - 1, block, in_channels, out_channels, stride, root_dim=0, **kw)
            self.tree2 = Tree(levels - 1, block, out_channels, out_channels,
                              root_dim=root_dim + out_channels, **kw)
        self.downsample = nn.MaxPool2d(stride, stride=stride) if stride > 1 else None
        self.project = None
        if in_channels != out_channels:
            self.project = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1, 1, bias=False),
                                         _bn(out_channels))
        # In the reference a multi-level Tree hands `residual` to a nested Tree, which overwrites
        # it (pose_dla_dcn.py:206-213): its `project` output is dead and its parameters never get
        # a gradient.  Keep the module (checkpoint keys, BatchNorm running statistics) but take it
        # out of autograd, or DistributedDataParallel would wait for gradients that never come.
        self.project_is_dead = levels > 1 and self.project is not None
        if self.project_is_dead:
            for prm in self.project.parameters():
                prm.requires_grad_(False)

    def fold(self):
        self._folded = _fold_conv_bn(self.project[0], self.project[1]) if self.project is not None \
            else None

    def forward(self, x, residual=None, children=None):
        children = [] if children is None else children
        bottom = x if self.downsample is None else downsample2(self.downsample, x)
        if self.project is None:
            residual = bottom
        elif self.project_is_dead:
            residual = None
            if self.training:                 # only its BatchNorm running stats are observable
                with torch.no_grad():
                    if bottom.is_cuda:            # (same kernels as the live projections)
                        bn_act(self.project[1], conv_train(self.project[0], bottom), relu=False)
                    else:
                        self.project(bottom)
        elif _use_folded(self):
            residual = _conv_folded(bottom, self.project[0], self._folded)
        else:                                  # conv1x1 + BatchNorm (no activation), fused BN in training
            residual = bn_act(self.project[1], conv_train(self.project[0], bottom), relu=False)
        if self.level_root:
            children.append(bottom)
        x1 = self.tree1(x, residual)
        if self.levels == 1:
            return self.root(self.tree2(x1), x1, *children)
        children.append(x1)
        return self.tree2(x1, children=children)


class DLA(nn.Module):
    def __init__(self, levels, channels, num_classes=1000, block=BasicBlock, residual_root=False,
                 linear_root=False):
        super().__init__()
        self.channels, self.num_classes = channels, num_classes
        c = channels
        self.base_layer = nn.Sequential(nn.Conv2d(3, c[0], 7, 1, 3, bias=False), _bn(c[0]),
                                        nn.ReLU(inplace=True))
        self.level0 = self._conv_level(c[0], c[0], levels[0])
        self.level1 = self._conv_level(c[0], c[1], levels[1], stride=2)
        for i in range(2, 6):
            setattr(self, "level%d" % i, Tree(levels[i], block, c[i - 1], c[i], 2,
                                              level_root=i > 2, root_residual=residual_root))

    @staticmethod
    def _conv_level(inplanes, planes, convs, stride=1, dilation=1):
        mods = []
        for i in range(convs):
            mods += [nn.Conv2d(inplanes, planes, 3, stride if i == 0 else 1, dilation, dilation,
                               bias=False), _bn(planes), nn.ReLU(inplace=True)]
            inplanes = planes
        return nn.Sequential(*mods)

    def fold(self):
        seqs = (self.base_layer, self.level0, self.level1)
        if any(len(seq) != 3 for seq in seqs):        # multi-conv levels: keep the plain path
            self._folded = None
            return
        self._folded = [_fold_conv_bn(seq[0], seq[1]) for seq in seqs]

    def forward(self, x):
        pyramid = []
        if _use_folded(self):
            x = _conv_folded(x, self.base_layer[0], self._folded[0], relu=True)
            y1 = _base_pair(self, x) if getattr(self, "skip_level0_output", False) else None
            if y1 is not None:                      # level0 + level1 in one launch; nobody reads level0's map
                pyramid += [None, y1]
                x = y1
            else:
                for seq, wb in zip((self.level0, self.level1), self._folded[1:]):
                    x = _conv_folded(x, seq[0], wb, relu=True)
                    pyramid.append(x)
            first_tree = 2
        elif self.training and x.is_cuda and all(len(q) == 3 for q in (self.base_layer, self.level0,
                                                                        self.level1)):
            for seq in (self.base_layer, self.level0, self.level1):
                x = bn_act(seq[1], conv_train(seq[0], x), relu=True)
                if seq is not self.base_layer:
                    pyramid.append(x)
            first_tree = 2
        else:
            x = self.base_layer(x)
            first_tree = 0
        for i in range(first_tree, 6):
            x = getattr(self, "level%d" % i)(x)
            pyramid.append(x)
        return pyramid

    def load_pretrained_model(self, data="imagenet", name="dla34", hash="ba72cf86"):
        """Offline only: `data + name` must be a local .pth (the reference downloads)."""
        if not name.endswith(".pth"):
            raise RuntimeError("no network: pass a local ImageNet checkpoint as name='*.pth'")
        weights = torch.load(data + name, map_location="cpu")
        self.load_state_dict(weights, strict=False)


def dla34(pretrained=False, **kwargs):
    model = DLA([1, 1, 1, 2, 2, 1], [16, 32, 64, 128, 256, 512], block=BasicBlock, **kwargs)
    if pretrained:
        model.load_pretrained_model(data="imagenet", name="dla34", hash="ba72cf86")
    return model


def fill_up_weights(up):
    """Bilinear kernel for the depth-wise transposed conv (reference :335-344)."""
    w = up.weight.data
    k = w.size(2)
    f = math.ceil(k / 2)
    c = (2 * f - 1 - f % 2) / (2.0 * f)
    tri = torch.tensor([1 - abs(i / f - c) for i in range(k)], dtype=w.dtype)
    w[:, 0] = torch.outer(tri, tri[:w.size(3)])


def fill_fc_weights(layers):
    for m in layers.modules():
        if isinstance(m, nn.Conv2d) and m.bias is not None:
            nn.init.constant_(m.bias, 0)


class DeformConv(nn.Module):
    """ReLU(BN(DCN(x))) (reference :347-359)."""

    def __init__(self, chi, cho):
        super().__init__()
        self.actf = nn.Sequential(_bn(cho), nn.ReLU(inplace=True))
        self.conv = DCN(chi, cho, kernel_size=(3, 3), stride=1, padding=1, dilation=1,
                        deformable_groups=1)

    def folded_affine(self):
        bn = self.actf[0]
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        shift = (self.conv.bias - bn.running_mean) * scale + bn.bias
        return scale.contiguous(), shift.contiguous()

    def fold(self):
        with torch.no_grad():
            self._folded = self.folded_affine()

    def forward(self, x):
        if not self.training and not torch.is_grad_enabled():
            scale, shift = self._folded if getattr(self, "_folded", None) is not None \
                else self.folded_affine()
            return self.conv.forward_fused(x, scale, shift, relu=True)
        if self.training:
            return bn_act(self.actf[0], self.conv(x), relu=True)
        return self.actf(self.conv(x))


def depthwise_up_add(x, up, skip):
    """up(x) + skip in one HIP kernel (inference): `up` is IDAUp's depth-wise ConvTranspose2d."""
    L = _C.lib()
    B, C, H, W = x.shape
    f = up.stride[0]
    out = torch.empty((B, C, H * f, W * f), dtype=torch.float32, device=x.device)
    x, skip = x.contiguous(), skip.contiguous()
    rc = L.cp_depthwise_up_forward(_C.ptr(x), _C.ptr(up.weight), _C.ptr(skip), _C.ptr(out), B, C, H,
                                   W, f, _C.stream())
    _C.check(rc, "cp_depthwise_up_forward")
    return out


class _DepthwiseUpAdd(torch.autograd.Function):
    """up(x) + skip for training: forward = the fused kernel, backward = two streaming kernels
    (grad_x, grad_weight); grad_skip is grad_out itself."""

    @staticmethod
    def forward(ctx, x, weight, skip, f):
        L = _C.lib()
        x, skip = x.contiguous(), skip.contiguous()
        B, C, H, W = x.shape
        out = torch.empty((B, C, H * f, W * f), dtype=torch.float32, device=x.device)
        rc = L.cp_depthwise_up_forward(_C.ptr(x), _C.ptr(weight), _C.ptr(skip), _C.ptr(out), B, C, H,
                                       W, f, _C.stream())
        _C.check(rc, "cp_depthwise_up_forward")
        ctx.save_for_backward(x, weight)
        ctx.f = f
        return out

    @staticmethod
    def backward(ctx, go):
        x, weight = ctx.saved_tensors
        L = _C.lib()
        go = go.contiguous()
        B, C, H, W = x.shape
        gx = torch.empty_like(x)
        gw = _C.zeros(weight.shape, weight.device)
        rc = L.cp_depthwise_up_backward(_C.ptr(x), _C.ptr(weight), _C.ptr(go), _C.ptr(gx), _C.ptr(gw),
                                        B, C, H, W, ctx.f, _C.stream())
        _C.check(rc, "cp_depthwise_up_backward")
        return gx, gw, go, None


class IDAUp(nn.Module):
    def __init__(self, o, channels, up_f):
        super().__init__()
        for i in range(1, len(channels)):
            f = int(up_f[i])
            up = nn.ConvTranspose2d(o, o, f * 2, stride=f, padding=f // 2, output_padding=0,
                                    groups=o, bias=False)
            fill_up_weights(up)
            setattr(self, "proj_%d" % i, DeformConv(channels[i], o))
            setattr(self, "up_%d" % i, up)
            setattr(self, "node_%d" % i, DeformConv(o, o))

    def forward(self, layers, startp, endp):
        fused = not self.training and not torch.is_grad_enabled()
        for i in range(startp + 1, endp):
            k = i - startp
            up, proj = getattr(self, "up_%d" % k), getattr(self, "proj_%d" % k)
            aligned = layers[i].is_cuda and (layers[i].shape[3] * up.stride[0]) % 4 == 0
            if fused and aligned and up.stride[0] in (2, 4, 8):
                summed = depthwise_up_add(proj(layers[i]), up, layers[i - 1])
            elif not fused and aligned and up.stride[0] in (2, 4):
                summed = _DepthwiseUpAdd.apply(proj(layers[i]), up.weight, layers[i - 1], up.stride[0])
            else:
                summed = up(proj(layers[i])) + layers[i - 1]
            layers[i] = getattr(self, "node_%d" % k)(summed)


class DLAUp(nn.Module):
    def __init__(self, startp, channels, scales, in_channels=None):
        super().__init__()
        self.startp = startp
        in_channels = list(channels) if in_channels is None else in_channels
        self.channels = channels
        channels = list(channels)
        scales = np.array(scales, dtype=int)
        for i in range(len(channels) - 1):
            j = -i - 2
            setattr(self, "ida_%d" % i, IDAUp(channels[j], in_channels[j:], scales[j:] // scales[j]))
            scales[j + 1:] = scales[j]
            in_channels[j + 1:] = [channels[j] for _ in channels[j + 1:]]

    def forward(self, layers):
        out = [layers[-1]]
        for i in range(len(layers) - self.startp - 1):
            getattr(self, "ida_%d" % i)(layers, len(layers) - i - 2, len(layers))
            out.insert(0, layers[-1])
        return out


def heads_fused_infer(owner, key, feat, w, b, tails, names):
    """All heads of one output stage as ONE kernel (cp_heads_fused_forward): 3x3 convolution + bias + ReLU + 1x1
    convolution + bias, the nheads x head_conv-channel intermediate never leaves the accumulator registers.
    w, b: the heads' 3x3 weights / biases concatenated along the output channels; tails: per head (1x1 weight
    transposed [hc][co], 1x1 bias or None, hc, co).  The permuted weights are cached on `owner` under `key`.
    None when the shapes are not the kernel's (more than 4 heads, a head wider than 64 outputs, head_conv not a
    multiple of 64, input channels not a multiple of 32)."""
    if not HEADS_FUSED or not conv3x3.mfma_enabled():
        return None
    hcs = {t[2] for t in tails}
    B, cin, H, W = feat.shape
    if not (feat.is_cuda and feat.dtype == torch.float32 and len(tails) <= 4 and len(hcs) == 1 and cin % 32 == 0
            and all(t[3] <= 64 for t in tails)):
        return None
    hc = hcs.pop()
    if hc % 64 != 0:
        return None
    L = _C.lib()
    cache = owner.__dict__.get(key)
    if cache is None or cache[0] is not w or cache[1] != w._version:
        wp1 = conv3x3._prepare(w, cin, w.shape[0], False)
        w2p = []
        for (w_t, b2, _, co) in tails:
            w2 = w_t.t().contiguous()                       # [co][hc]
            buf = torch.empty(L.cp_heads_fused_w2_bytes(hc), dtype=torch.uint8, device=w.device)
            _C.check(L.cp_heads_fused_prepare_w2(_C.ptr(w2), co, hc, _C.ptr(buf), _C.stream()),
                     "cp_heads_fused_prepare_w2")
            w2p.append(buf)
        cache = (w, w._version, wp1, w2p)
        owner.__dict__[key] = cache
    feat = feat.contiguous()
    outs = [torch.empty((B, t[3], H, W), dtype=torch.float32, device=feat.device) for t in tails]
    n = len(tails)
    vp = _C.c_void_p
    w2arr = (vp * n)(*[t.data_ptr() for t in cache[3]])
    b2arr = (vp * n)(*[(t[1].data_ptr() if t[1] is not None else None) for t in tails])
    oarr = (vp * n)(*[o.data_ptr() for o in outs])
    carr = (_C.c_int32 * n)(*[t[3] for t in tails])
    end = _C.kernel_timer.start(("heads_fused", cin, w.shape[0], H, W, B, hc, sum(t[3] for t in tails))) \
        if _C.kernel_timer is not None else None
    rc = L.cp_heads_fused_forward(_C.ptr(feat), _C.ptr(cache[2]), _C.ptr(b), w2arr, b2arr, oarr, carr, n, B, cin, H, W,
                                  hc, _C.stream())
    if end is not None:
        end.record()
    if rc == -2:
        return None
    _C.check(rc, "cp_heads_fused_forward")
    return dict(zip(names, outs))


class DLASeg(nn.Module):
    def __init__(self, base_name, heads, pretrained, down_ratio, final_kernel, last_level,
                 head_conv, out_channel=0):
        super().__init__()
        assert down_ratio in [2, 4, 8, 16]
        self.first_level = int(np.log2(down_ratio))
        self.last_level = last_level
        self.base = globals()[base_name](pretrained=pretrained)
        self.base.skip_level0_output = self.first_level >= 1     # dla_up reads the pyramid from first_level on
        channels = self.base.channels
        fl = self.first_level
        scales = [2 ** i for i in range(len(channels[fl:]))]
        self.dla_up = DLAUp(fl, channels[fl:], scales)
        out_channel = out_channel or channels[fl]
        self.ida_up = IDAUp(out_channel, channels[fl:last_level],
                            [2 ** i for i in range(last_level - fl)])
        self.heads = heads
        for head, classes in heads.items():
            if head_conv > 0:
                fc = nn.Sequential(
                    nn.Conv2d(channels[fl], head_conv, 3, padding=1, bias=True),
                    nn.ReLU(inplace=True),
                    nn.Conv2d(head_conv, classes, final_kernel, 1, final_kernel // 2, bias=True))
                last = fc[-1]
            else:
                fc = nn.Conv2d(channels[fl], classes, final_kernel, 1, final_kernel // 2, bias=True)
                last = fc
            if "hm" in head:
                last.bias.data.fill_(-2.19)
            else:
                fill_fc_weights(fc)
            self.__setattr__(head, fc)

    def prepare_inference(self, dcn_contraction="auto"):
        """Fold every BatchNorm into its convolution / DCN epilogue and concatenate the heads'
        first 3x3 convolutions into one (call after loading weights, in eval mode).  Undone by
        train().  dcn_contraction: "f32" (exact fp32 MFMA), "bf16x3" (split-bf16 emulation,
        ~2^-16 relative error) or "auto" (DCNv2/dcn_v2.py::auto_contraction, decided per layer and map size at
        the call: bf16x3 wherever the library runs its LDS-region kernel and for the layers with more than 64
        output channels, f32 for the small 64-channel maps)."""
        self.eval()
        for m in self.modules():
            if hasattr(m, "fold"):
                m.fold()
            if isinstance(m, DCN):
                m.contraction = None if dcn_contraction == "auto" else dcn_contraction   # None: DCN.infer_contraction
        self._heads_cat = None
        fcs = [getattr(self, h) for h in self.heads]
        if all(isinstance(fc, nn.Sequential) and len(fc) == 3 for fc in fcs):
            with torch.no_grad():
                tails = [(fc[2].weight.reshape(fc[2].out_channels, -1).t().contiguous(),
                          fc[2].bias.detach().clone() if fc[2].bias is not None else None,
                          fc[0].out_channels, fc[2].out_channels) for fc in fcs]
                fast = all(fc[2].kernel_size == (1, 1) and fc[2].out_channels <= 32 and fc[0].bias is not None
                           and fc[0].kernel_size == (3, 3) for fc in fcs)
                if fast:
                    self._heads_cat = (torch.cat([fc[0].weight for fc in fcs], 0).contiguous(),
                                       torch.cat([fc[0].bias for fc in fcs], 0).contiguous(), tails)
        return self

    def _heads_fused(self, feat):
        w, b, tails = self._heads_cat
        return heads_fused_infer(self, "_heads_fused_cache", feat, w, b, tails, list(self.heads))

    def _heads_fast(self, feat):
        """All heads' conv3x3 as ONE library convolution (they share the input); each head's
        bias + ReLU + 1x1 convolution is then one streaming kernel over its channel slice of the
        raw result (cp_conv1x1_act_forward) -- the 4x256-channel tensor is read once instead of
        going through a bias/ReLU pass and a GEMM with a handful of output rows."""
        w, b, tails = self._heads_cat
        out = self._heads_fused(feat)
        if out is not None:
            return out
        y = conv3x3_infer(feat, self, w, key="_heads_wperm")
        if y is None:
            y = F.conv2d(feat, w, None, padding=1)
        B, ctot, H, W = y.shape
        hw = H * W
        out, c0 = {}, 0
        L = _C.lib()
        for h, (w_t, b1, hc, co) in zip(self.heads, tails):
            o = torch.empty((B, co, H, W), dtype=torch.float32, device=y.device)
            _C.check(L.cp_conv1x1_act_forward(
                _C.c_void_p(y.data_ptr() + 4 * c0 * hw), ctot * hw, _C.c_void_p(b.data_ptr() + 4 * c0), 1,
                _C.ptr(w_t), _C.ptr(b1), _C.ptr(o), B, hc, co, hw, _C.stream()), "cp_conv1x1_act_forward")
            out[h] = o
            c0 += hc
        return out

    def train(self, mode=True):
        if mode:
            for m in self.modules():
                if hasattr(m, "_folded"):
                    m._folded = None
                for k in [k for k in m.__dict__ if k.startswith(("_mfma_wperm", "_heads_wperm", "_heads_fused", "_dcn_fwd_ws", "_dcn_fused_ws", "_stem_wperm"))]:
                    del m.__dict__[k]          # permuted inference weights / DCN workspaces of the folded tensors
            self._heads_cat = None
        _C.release_zero_pool()                 # (gradient accumulators of the mode being left)
        return super().train(mode)

    def state_dict(self, *args, **kwargs):
        flush_batch_counts()                   # (num_batches_tracked increments still pending from a sub-module run)
        return super().state_dict(*args, **kwargs)

    def forward(self, x):
        try:
            x = self.dla_up(self.base(x))
            # (the reference clones these tensors, pose_dla_dcn.py:476-478; IDAUp here only re-binds list
            #  entries and never writes into its inputs, so the copies are not needed)
            y = [x[i] for i in range(self.last_level - self.first_level)]
            self.ida_up(y, 0, len(y))
        finally:
            flush_batch_counts()
        if getattr(self, "_heads_cat", None) is not None and not self.training \
                and not torch.is_grad_enabled() and y[-1].is_cuda \
                and (y[-1].shape[2] * y[-1].shape[3]) % 4 == 0:
            return [self._heads_fast(y[-1])]
        if self.training and y[-1].is_cuda and all(
                isinstance(getattr(self, h), nn.Sequential) and len(getattr(self, h)) == 3 for h in self.heads):
            # training: each head's Conv3x3 + bias + ReLU with the fused epilogue, then its 1x1 conv
            fcs = [getattr(self, h) for h in self.heads]
            outs = conv3x3.heads_train(fcs, y[-1]) if torch.is_grad_enabled() and y[-1].dtype == torch.float32 else None
            if outs is None:                       # (shapes outside the MFMA kernels': the heads from their pieces)
                outs = [conv_bias(fc[2], conv_bias_relu(fc[0], y[-1])) for fc in fcs]
            return [dict(zip(self.heads, outs))]
        return [{head: getattr(self, head)(y[-1]) for head in self.heads}]


def get_pose_net(num_layers, heads, head_conv=256, down_ratio=4, pretrained=False):
    return DLASeg("dla{}".format(num_layers), heads, pretrained=pretrained, down_ratio=down_ratio,
                  final_kernel=1, last_level=5, head_conv=head_conv)
