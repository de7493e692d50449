"""Mirror of src/lib/models/losses.py for polydet: FocalLoss, RegL1Loss, PolyLoss.

Same class names, constructor and forward signatures as the reference
(losses.py:792-799, 817-830, 833-959); every forward/backward is a HIP kernel
behind the C ABI (include/centerpoly_hip.h).  Losses stay device scalars: no
`.item()` / host sync anywhere (the reference syncs once per object and per
`num_pos == 0` test).
"""
import torch
import torch.nn as nn

from .. import _C


def _f32c(t):
    if t.dtype != torch.float32:
        raise _C.NativeError("expected float32, got %s" % t.dtype)
    return t.contiguous()


# ------------------------------------------------------------------ focal ---

class _SigmoidFocal(torch.autograd.Function):
    """clamp(sigmoid_(logits)) fused with the CornerNet focal loss.

    forward mutates `logits` in place into the activated heat map (what
    trains/polydet.py:46 does with `_sigmoid`) and returns (loss, activated)."""

    @staticmethod
    def forward(ctx, logits, gt):
        L = _C.lib()
        if not logits.is_contiguous():
            raise _C.NativeError("heat-map head output must be contiguous")
        gt = _f32c(gt)
        n = logits.numel()
        dev = logits.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        stats = torch.empty((3,), dtype=torch.float32, device=dev)
        ws = _C.workspace(L.cp_sigmoid_focal_workspace_bytes(n), dev)
        rc = L.cp_sigmoid_focal_forward(_C.ptr(logits), _C.ptr(gt), n, _C.ptr(loss), _C.ptr(stats),
                                        _C.ptr(ws), ws.numel(), _C.stream())
        _C.check(rc, "cp_sigmoid_focal_forward")
        ctx.mark_dirty(logits)
        ctx.save_for_backward(logits, gt, stats)
        return loss, logits

    @staticmethod
    def backward(ctx, grad_loss, grad_act):
        act, gt, stats = ctx.saved_tensors
        L = _C.lib()
        grad = torch.empty_like(act)
        gl = _f32c(grad_loss.reshape(1))
        rc = L.cp_sigmoid_focal_backward(_C.ptr(act), _C.ptr(gt), act.numel(), _C.ptr(stats),
                                         _C.ptr(gl), _C.ptr(grad), _C.stream())
        _C.check(rc, "cp_sigmoid_focal_backward")
        if grad_act is not None:
            # gradient arriving on the activated map itself (not used by PolydetLoss):
            # d act / d logit = act (1 - act) strictly inside the clamp
            inside = (act > 1e-4) & (act < 1 - 1e-4)
            grad = grad + grad_act * act * (1 - act) * inside
        return grad, None


def sigmoid_focal_loss(logits, gt):
    """(loss, activated_hm); `logits` is overwritten with the activated map."""
    return _SigmoidFocal.apply(logits, gt)


class FocalLoss(nn.Module):
    """losses.py:792-799 signature: forward(out, target) with `out` ALREADY
    activated by _sigmoid.  Differentiable w.r.t. `out` through torch ops on the
    device; PolydetLoss uses the fused `sigmoid_focal_loss` instead."""

    def forward(self, out, target):
        pos = target.eq(1).float()
        neg = target.lt(1).float()
        pos_loss = (torch.log(out) * torch.pow(1 - out, 2) * pos).sum()
        neg_loss = (torch.log(1 - out) * torch.pow(out, 2) * torch.pow(1 - target, 4) * neg).sum()
        num_pos = pos.sum()
        return -(pos_loss + neg_loss) / torch.clamp(num_pos, min=1.0)   # no host sync


# ---------------------------------------------------------- gathered L1 -----

class _GatherL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, mask, ind, target, pred_add, mode, eps):
        L = _C.lib()
        feat, target = _f32c(feat), _f32c(target)
        mask = mask.contiguous()
        if mask.dtype != torch.uint8:
            mask = mask.to(torch.uint8)
        ind = ind.contiguous()
        if ind.dtype != torch.int64:
            ind = ind.long()
        B, D, H, W = feat.shape
        M = ind.shape[1]
        loss = torch.empty((), dtype=torch.float32, device=feat.device)
        pa = _f32c(pred_add) if pred_add is not None else None
        rc = L.cp_gather_l1_forward(_C.ptr(feat), _C.ptr(ind), _C.ptr(mask), _C.ptr(target),
                                    _C.ptr(pa), B, D, H, W, M, mode, eps, _C.ptr(loss), _C.stream())
        _C.check(rc, "cp_gather_l1_forward")
        ctx.save_for_backward(feat, mask, ind, target, pa if pa is not None else torch.empty(0))
        ctx.cfg = (mode, eps, pa is not None)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        feat, mask, ind, target, pa = ctx.saved_tensors
        mode, eps, has_pa = ctx.cfg
        L = _C.lib()
        B, D, H, W = feat.shape
        grad = torch.zeros_like(feat)
        gl = _f32c(grad_loss.reshape(1))
        rc = L.cp_gather_l1_backward(_C.ptr(feat), _C.ptr(ind), _C.ptr(mask), _C.ptr(target),
                                     _C.ptr(pa) if has_pa else None, B, D, H, W, ind.shape[1], mode,
                                     eps, _C.ptr(gl), _C.ptr(grad), _C.stream())
        _C.check(rc, "cp_gather_l1_backward")
        return grad, None, None, None, None, None, None


class RegL1Loss(nn.Module):
    """losses.py:817-830: forward(output[B,D,h,w], mask[B,M], ind[B,M], target[B,M,D])."""

    def forward(self, output, mask, ind, target):
        return _GatherL1.apply(output, mask, ind, target, None, _C.L1_PLAIN, 1e-4)


class RegLoss(nn.Module):
    """losses.py:801-815 (`--reg_loss sl1`): smooth-L1 on the gathered rows, divided by the number of
    masked objects + 1e-4 (not by objects x D)."""

    def forward(self, output, mask, ind, target):
        return _GatherL1.apply(output, mask, ind, target, None, _C.L1_SMOOTH, 1e-4)


class _Mse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gt):
        L = _C.lib()
        x, gt = _f32c(x), _f32c(gt)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        nws = L.cp_mse_workspace_bytes()
        ws = _C.workspace(nws, x.device)
        _C.check(L.cp_mse_forward(_C.ptr(x), _C.ptr(gt), x.numel(), _C.ptr(loss), _C.ptr(ws), nws, _C.stream()),
                 "cp_mse_forward")
        ctx.save_for_backward(x, gt)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        x, gt = ctx.saved_tensors
        g = torch.empty_like(x)
        _C.check(_C.lib().cp_mse_backward(_C.ptr(x), _C.ptr(gt), x.numel(), _C.ptr(_f32c(grad_loss.reshape(1))),
                                          _C.ptr(g), _C.stream()), "cp_mse_backward")
        return g, None


class MSELoss(nn.Module):
    """`--mse_loss`: torch.nn.MSELoss() on the raw heat-map head (trains/polydet.py:23)."""

    def forward(self, out, target):
        return _Mse.apply(out, target)


class _DenseL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, mask, eps):
        L = _C.lib()
        pred, target, mask = _f32c(pred), _f32c(target), _f32c(mask)
        if pred.shape != target.shape or pred.shape != mask.shape:
            raise _C.NativeError("dense L1: pred, target and mask must have one shape")
        out2 = torch.empty((2,), dtype=torch.float32, device=pred.device)
        nws = L.cp_dense_l1_workspace_bytes()
        ws = _C.workspace(nws, pred.device)
        _C.check(L.cp_dense_l1_forward(_C.ptr(pred), _C.ptr(target), _C.ptr(mask), pred.numel(), float(eps), _C.ptr(out2),
                                       _C.ptr(ws), nws, _C.stream()), "cp_dense_l1_forward")
        ctx.save_for_backward(pred, target, mask, out2)
        return out2[0]

    @staticmethod
    def backward(ctx, grad_loss):
        pred, target, mask, out2 = ctx.saved_tensors
        g = torch.empty_like(pred)
        _C.check(_C.lib().cp_dense_l1_backward(_C.ptr(pred), _C.ptr(target), _C.ptr(mask), pred.numel(),
                                               _C.c_void_p(out2.data_ptr() + 4), _C.ptr(_f32c(grad_loss.reshape(1))),
                                               _C.ptr(g), _C.stream()), "cp_dense_l1_backward")
        return g, None, None, None


def dense_poly_l1_loss(pred, target, mask, eps=1e-4):
    """`--dense_poly` (trains/polydet.py:107-110): L1Loss(reduction='sum')(pred * mask, target * mask) /
    (mask.sum() + 1e-4) over the whole [B, 2N, h, w] maps, one HIP streaming pass each way."""
    return _DenseL1.apply(pred, target, mask, eps)


# ------------------------------------------------------- polygon losses -----

class _PolyIouOrder(torch.autograd.Function):
    """Per-object Weiler-Atherton IoU term and order (pairwise hinge) term."""

    @staticmethod
    def forward(ctx, feat, mask, ind, target, flags):
        L = _C.lib()
        feat, target = _f32c(feat), _f32c(target)
        mask = mask.contiguous() if mask.dtype == torch.uint8 else mask.to(torch.uint8).contiguous()
        ind = ind.contiguous() if ind.dtype == torch.int64 else ind.long().contiguous()
        B, N2, H, W = feat.shape
        M = ind.shape[1]
        dev = feat.device
        iou = torch.zeros((), dtype=torch.float32, device=dev)
        order = torch.zeros((), dtype=torch.float32, device=dev)
        pred_add = torch.zeros((B, M, N2), dtype=torch.float32, device=dev)
        ws = _C.workspace(L.cp_poly_iou_order_workspace_bytes(B, M, N2 // 2), dev)
        rc = L.cp_poly_iou_order_forward(_C.ptr(feat), _C.ptr(ind), _C.ptr(mask), _C.ptr(target), B,
                                         N2 // 2, H, W, M, flags, _C.ptr(iou), _C.ptr(order),
                                         _C.ptr(pred_add), _C.ptr(ws), ws.numel(), _C.stream())
        _C.check(rc, "cp_poly_iou_order_forward")
        ctx.save_for_backward(feat, mask, ind, target, ws)
        ctx.flags = flags
        ctx.mark_non_differentiable(pred_add)
        return iou, order, pred_add

    @staticmethod
    def backward(ctx, g_iou, g_order, _g_add):
        feat, mask, ind, target, ws = ctx.saved_tensors
        L = _C.lib()
        B, N2, H, W = feat.shape
        grad = torch.zeros_like(feat)
        gi = _f32c(g_iou.reshape(1)) if g_iou is not None else torch.zeros(1, device=feat.device)
        go = _f32c(g_order.reshape(1)) if g_order is not None else torch.zeros(1, device=feat.device)
        rc = L.cp_poly_iou_order_backward(_C.ptr(feat), _C.ptr(ind), _C.ptr(mask), _C.ptr(target), B,
                                          N2 // 2, H, W, ind.shape[1], ctx.flags, _C.ptr(gi),
                                          _C.ptr(go), _C.ptr(grad), _C.ptr(ws), ws.numel(),
                                          _C.stream())
        _C.check(rc, "cp_poly_iou_order_backward")
        return grad, None, None, None, None


class PolyLoss(nn.Module):
    """losses.py:833-959.  `opt` supplies poly_loss {l1|iou|l1+iou|relu},
    rep {cartesian|polar|polar_fixed}, poly_order (bool)."""

    def __init__(self, opt):
        super(PolyLoss, self).__init__()
        self.opt = opt

    def forward(self, output, mask, ind, target, freq_mask=None, peak=None, hm=None):
        opt = self.opt
        kind, rep, order_on = opt.poly_loss, opt.rep, bool(opt.poly_order)
        use_iou = kind in ("iou", "l1+iou", "relu")
        use_l1 = kind in ("l1", "l1+iou", "relu")
        loss = output.new_zeros(())
        loss_order = output.new_zeros(())
        pred_add = None
        flags = (1 if use_iou else 0) | (2 if order_on else 0)
        if flags:
            iou, loss_order, add = _PolyIouOrder.apply(output, mask, ind, target, flags)
            if use_iou:
                loss = loss + iou
            if order_on:
                pred_add = add          # the order term's in-place edit is seen by the L1 term
        if use_l1:
            if kind == "relu" and rep == "cartesian":
                mode = _C.L1_RELU20
            elif rep == "cartesian":
                mode = _C.L1_PLAIN
            elif rep == "polar":
                mode = _C.L1_POLAR
            elif rep == "polar_fixed":
                mode = _C.L1_POLAR_FIXED
            else:
                raise ValueError("unknown rep %r" % rep)
            loss = loss + _GatherL1.apply(output, mask, ind, target, pred_add, mode, 1e-6)
        if order_on:
            return loss, loss_order
        return loss
