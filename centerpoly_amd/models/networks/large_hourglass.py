"""Stacked Hourglass (Hourglass-104) with the polydet heads.

Same constructor surface and state_dict key grammar as the live classes of the
reference's src/lib/models/networks/large_hourglass.py (convolution :24-37,
residual :55-81, kp_module :283-342, exkp :345-462, HourglassNet :471-484):
`pre.*`, `kps.s.{up1,low1,low2,low3}.*`, `cnvs.s.*`, `inters.*`, `inters_.*`,
`cnvs_.*`, heads `{head}.s.0.conv.*` / `{head}.s.1.*`.  The dead U-Net / SqEx
classes of that file are not part of the hot path and are not restated.
Pooling is replaced by stride-2 residuals and up-sampling is nearest x2, as in
the reference.  No DCN here: every layer is a dense convolution.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import _C
from . import conv3x3
from .conv3x3 import conv3x3_infer
from .pose_dla_dcn import _conv_folded, _fold_conv_bn, _use_folded, bn_act, conv_train, flush_batch_counts, heads_fused_infer

# Inference (`prepare_inference()`): every BatchNorm is folded into its convolution (weights scaled,
# shift as a bias) and each conv is followed by ONE fused in-place pass -- + bias (+ residual) (+ ReLU),
# cp_bias_act_inplace -- instead of BatchNorm, add and ReLU passes; the four heads' 3x3 convolutions of a
# stack run as one convolution (they share their input) and each head's bias + ReLU + 1x1 convolution
# is one streaming kernel (cp_conv1x1_act_forward).  train() drops the folded copies.


class convolution(nn.Module):
    def __init__(self, k, inp_dim, out_dim, stride=1, with_bn=True):
        super().__init__()
        pad = (k - 1) // 2
        self.conv = nn.Conv2d(inp_dim, out_dim, (k, k), padding=(pad, pad), stride=(stride, stride),
                              bias=not with_bn)
        self.bn = nn.BatchNorm2d(out_dim) if with_bn else nn.Sequential()
        self.relu = nn.ReLU(inplace=True)

    def fold(self):
        if isinstance(self.bn, nn.BatchNorm2d):
            self._folded = _fold_conv_bn(self.conv, self.bn)
        else:
            self._folded = (self.conv.weight, self.conv.bias)

    def forward(self, x):
        if _use_folded(self):
            return _conv_folded(x, self.conv, self._folded, relu=True)
        if isinstance(self.bn, nn.BatchNorm2d):
            return bn_act(self.bn, conv_train(self.conv, x), relu=True)     # fused BN+ReLU in training
        return self.relu(self.bn(self.conv(x)))


class residual(nn.Module):
    def __init__(self, k, inp_dim, out_dim, stride=1, with_bn=True):
        super().__init__()
        self.conv1 = nn.Conv2d(inp_dim, out_dim, (3, 3), padding=(1, 1), stride=(stride, stride),
                               bias=False)
        self.bn1 = nn.BatchNorm2d(out_dim)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(out_dim, out_dim, (3, 3), padding=(1, 1), bias=False)
        self.bn2 = nn.BatchNorm2d(out_dim)
        needs_proj = stride != 1 or inp_dim != out_dim
        self.skip = nn.Sequential(
            nn.Conv2d(inp_dim, out_dim, (1, 1), stride=(stride, stride), bias=False),
            nn.BatchNorm2d(out_dim)) if needs_proj else nn.Sequential()
        self.relu = nn.ReLU(inplace=True)

    def fold(self):
        skip = _fold_conv_bn(self.skip[0], self.skip[1]) if len(self.skip) else None
        self._folded = (_fold_conv_bn(self.conv1, self.bn1), _fold_conv_bn(self.conv2, self.bn2), skip)

    def forward(self, x):
        if _use_folded(self):
            f1, f2, fs = self._folded
            skip = x if fs is None else _conv_folded(x, self.skip[0], fs, relu=False)
            z = conv3x3.block_infer(x, self.conv1, f1, self.conv2, f2, skip)
            if z is not None:                       # the intermediate as split bf16 planes (bit-identical)
                return z
            y = _conv_folded(x, self.conv1, f1, relu=True)
            return _conv_folded(y, self.conv2, f2, relu=True, residual=skip)
        y = bn_act(self.bn1, conv_train(self.conv1, x), relu=True)
        return bn_act(self.bn2, conv_train(self.conv2, y), relu=True, residual=self.skip(x))


def _stack(first, rest_dim, count, tail=None):
    """`count` residuals; `first` and optional `tail` are given, the middle ones keep rest_dim."""
    layers = [first] + [residual(3, rest_dim, rest_dim) for _ in range(count - 1)]
    if tail is not None:
        layers[-1] = tail
    return nn.Sequential(*layers)


def make_layer(k, inp_dim, out_dim, modules, layer=residual, **kw):
    return _stack(layer(k, inp_dim, out_dim, **kw), out_dim, modules)


def make_layer_revr(k, inp_dim, out_dim, modules, layer=residual, **kw):
    layers = [layer(k, inp_dim, inp_dim, **kw) for _ in range(modules - 1)]
    layers.append(layer(k, inp_dim, out_dim, **kw))
    return nn.Sequential(*layers)


def make_hg_layer(k, dim0, dim1, mod, layer=residual, **kw):
    """First residual has stride 2 (replaces max-pooling, reference :465-468)."""
    return _stack(layer(k, dim0, dim1, stride=2), dim1, mod)


class _Up2Add(torch.autograd.Function):
    """up1 + nearest x2 up-sampling of low in one kernel (cp_upsample2x_add); gradients: grad_out and its 2x2 sums."""

    @staticmethod
    def forward(ctx, up1, low):
        B, C, H, W = low.shape
        out = torch.empty_like(up1)
        _C.check(_C.lib().cp_upsample2x_add(_C.ptr(up1), _C.ptr(low), _C.ptr(out), B, C, H, W, _C.stream()),
                 "cp_upsample2x_add")
        return out

    @staticmethod
    def backward(ctx, go):
        return go, F.avg_pool2d(go, 2) * 4.0


class MergeUp(nn.Module):
    def forward(self, up1, up2):
        return up1 + up2

    def fused(self, up1, low):
        """up1 + Upsample(x2)(low) without the up-sampled tensor."""
        if up1.is_cuda and up1.dtype == torch.float32 and up1.is_contiguous() and low.is_contiguous() \
                and up1.shape[2:] == (2 * low.shape[2], 2 * low.shape[3]) and up1.shape[0] * up1.shape[1] <= 65535:
            return _Up2Add.apply(up1, low)
        return None


def make_kp_layer(cnv_dim, curr_dim, out_dim):
    return nn.Sequential(convolution(3, cnv_dim, curr_dim, with_bn=False),
                         nn.Conv2d(curr_dim, out_dim, (1, 1)))


make_poly_layer = make_kp_layer


class kp_module(nn.Module):
    def __init__(self, n, dims, modules):
        super().__init__()
        self.n = n
        curr_mod, next_mod = modules[0], modules[1]
        curr_dim, next_dim = dims[0], dims[1]
        self.up1 = make_layer(3, curr_dim, curr_dim, curr_mod)
        self.max1 = nn.Sequential()
        self.low1 = make_hg_layer(3, curr_dim, next_dim, curr_mod)
        self.low2 = kp_module(n - 1, dims[1:], modules[1:]) if n > 1 else \
            make_layer(3, next_dim, next_dim, next_mod)
        self.low3 = make_layer_revr(3, next_dim, curr_dim, curr_mod)
        self.up2 = nn.Upsample(scale_factor=2)
        self.merge = MergeUp()

    def forward(self, x):
        up1 = self.up1(x)
        low = self.low3(self.low2(self.low1(self.max1(x))))
        out = self.merge.fused(up1, low)
        return out if out is not None else self.merge(up1, self.up2(low))


class exkp(nn.Module):
    def __init__(self, n, nstack, dims, modules, heads, cnv_dim=256):
        super().__init__()
        self.nstack = nstack
        self.heads = heads
        curr_dim = dims[0]
        self.pre = nn.Sequential(convolution(7, 3, 128, stride=2), residual(3, 128, 256, stride=2))
        self.kps = nn.ModuleList([kp_module(n, dims, modules) for _ in range(nstack)])
        self.cnvs = nn.ModuleList([convolution(3, curr_dim, cnv_dim) for _ in range(nstack)])
        self.inters = nn.ModuleList([residual(3, curr_dim, curr_dim) for _ in range(nstack - 1)])
        self.inters_ = nn.ModuleList([
            nn.Sequential(nn.Conv2d(curr_dim, curr_dim, (1, 1), bias=False),
                          nn.BatchNorm2d(curr_dim)) for _ in range(nstack - 1)])
        self.cnvs_ = nn.ModuleList([
            nn.Sequential(nn.Conv2d(cnv_dim, curr_dim, (1, 1), bias=False),
                          nn.BatchNorm2d(curr_dim)) for _ in range(nstack - 1)])
        for head in heads.keys():
            module = nn.ModuleList([make_kp_layer(cnv_dim, curr_dim, heads[head])
                                    for _ in range(nstack)])
            self.__setattr__(head, module)
            if "hm" in head:
                for heat in module:
                    heat[-1].bias.data.fill_(-2.19)
        self.relu = nn.ReLU(inplace=True)

    def prepare_inference(self):
        """Fold every BatchNorm, concatenate each stack's head convolutions (call after loading weights).
        Undone by train()."""
        self.eval()
        for m in self.modules():
            if m is not self and hasattr(m, "fold"):
                m.fold()
        with torch.no_grad():
            self._inter_folded = [(_fold_conv_bn(a[0], a[1]), _fold_conv_bn(b[0], b[1]))
                                  for a, b in zip(self.inters_, self.cnvs_)]
            self._heads_cat = []
            for s in range(self.nstack):
                fcs = [getattr(self, h)[s] for h in self.heads]
                w = torch.cat([fc[0].conv.weight for fc in fcs], 0).contiguous()
                b = torch.cat([fc[0].conv.bias for fc in fcs], 0).contiguous()
                tails = [(fc[1].weight.reshape(fc[1].out_channels, -1).t().contiguous(),
                          fc[1].bias.detach().clone(), fc[0].conv.out_channels, fc[1].out_channels) for fc in fcs]
                self._heads_cat.append((w, b, tails))
        return self

    def train(self, mode=True):
        if mode:
            for m in self.modules():
                if hasattr(m, "_folded"):
                    m._folded = None
                for k in [k for k in m.__dict__ if k.startswith(("_mfma_wperm", "_heads_wperm", "_stem_wperm", "_dcn_fwd_ws", "_dcn_fused_ws"))]:
                    del m.__dict__[k]          # permuted inference weights / DCN workspaces of the folded tensors
            self._heads_cat = None
            self._inter_folded = None
        _C.release_zero_pool()                 # (gradient accumulators of the mode being left)
        return super().train(mode)

    def _heads_fast(self, s, cnv):
        w, b, tails = self._heads_cat[s]
        out = heads_fused_infer(self, "_heads_fused_cache%d" % s, cnv, w, b, tails, list(self.heads))
        if out is not None:
            return out
        y = conv3x3_infer(cnv, self, w, key="_heads_wperm%d" % s)
        if y is None:
            y = F.conv2d(cnv, w, None, padding=1)
        B, ctot, H, W = y.shape
        hw = H * W
        out, c0 = {}, 0
        L = _C.lib()
        for h, (w_t, b1, hc, co) in zip(self.heads, tails):
            o = torch.empty((B, co, H, W), dtype=torch.float32, device=y.device)
            if co <= 32:
                _C.check(L.cp_conv1x1_act_forward(
                    _C.c_void_p(y.data_ptr() + 4 * c0 * hw), ctot * hw, _C.c_void_p(b.data_ptr() + 4 * c0), 1,
                    _C.ptr(w_t), _C.ptr(b1), _C.ptr(o), B, hc, co, hw, _C.stream()), "cp_conv1x1_act_forward")
            else:                                   # wide head (48-channel polar polygons): 32-channel slices
                for a0 in range(0, co, 32):
                    a1 = min(co, a0 + 32)
                    wt = w_t[:, a0:a1].contiguous()
                    bb = b1[a0:a1].contiguous()
                    for i in range(B):
                        _C.check(L.cp_conv1x1_act_forward(
                            _C.c_void_p(y.data_ptr() + 4 * (i * ctot + c0) * hw), ctot * hw,
                            _C.c_void_p(b.data_ptr() + 4 * c0), 1, _C.ptr(wt), _C.ptr(bb),
                            _C.c_void_p(o.data_ptr() + 4 * (i * co + a0) * hw), 1, hc, a1 - a0, hw, _C.stream()),
                            "cp_conv1x1_act_forward")
            out[h] = o
            c0 += hc
        return out

    def forward(self, image):
        try:
            fused = getattr(self, "_heads_cat", None) is not None and not self.training \
                and not torch.is_grad_enabled() and image.is_cuda
            inter = self.pre(image)
            outs = []
            for s in range(self.nstack):
                cnv = self.cnvs[s](self.kps[s](inter))
                if fused and (cnv.shape[2] * cnv.shape[3]) % 4 == 0:
                    outs.append(self._heads_fast(s, cnv))
                else:
                    outs.append({head: getattr(self, head)[s](cnv) for head in self.heads})
                if s < self.nstack - 1:
                    if fused:
                        fa, fb = self._inter_folded[s]
                        t = _conv_folded(inter, self.inters_[s][0], fa, relu=False)
                        inter = _conv_folded(cnv, self.cnvs_[s][0], fb, relu=True, residual=t)
                    else:
                        inter = self.relu(self.inters_[s](inter) + self.cnvs_[s](cnv))
                    inter = self.inters[s](inter)
        finally:
            flush_batch_counts()
        return outs


class HourglassNet(exkp):
    def __init__(self, heads, num_stacks=2):
        super().__init__(5, num_stacks, [256, 256, 384, 384, 384, 512], [2, 2, 2, 2, 2, 4], heads,
                         cnv_dim=256)


def get_large_hourglass_net(num_layers, heads, head_conv):
    return HourglassNet(heads, 2)


def get_small_hourglass_net(num_layers, heads, head_conv):
    return HourglassNet(heads, 1)
