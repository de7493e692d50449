"""Oracle: polydet losses, torch CPU fp32 with autograd.  TEST INFRASTRUCTURE.

Restates, with the reference's LITERAL semantics:
  * _sigmoid            src/lib/models/utils.py:8-10
  * _neg_loss/FocalLoss src/lib/models/losses.py:146-171, 792-799
  * RegL1Loss           src/lib/models/losses.py:817-830
  * PolyLoss            src/lib/models/losses.py:833-959
  * WeilPolygonClipper  src/lib/models/losses.py:373-628
  * area                src/lib/models/losses.py:25-41
  * PolydetLoss         src/lib/trains/polydet.py:38-162

Literal quirks kept on purpose: the clipper and `area` read every point as
(r, theta) whatever `rep` says; `area` counts the first shoelace term twice;
the traversal stops once len(used) >= len(remaining inbounds); the order term
adds 2*3.14 to negative angles and that edit is seen by the L1 term.

Where the reference would crash or hang the oracle DEFINES the result (the HIP
kernel follows the same definition):
  * inbounds non-empty and outbounds empty (IndexError at losses.py:602)
      -> empty clip polygon;
  * inbounds exhausted while walking the clip polygon (endless loop at
    losses.py:607) -> traversal ends, polygon accumulated so far is returned;
  * more than MAXPTS(n, m) = 8*(n+m) output vertices -> traversal ends there.
"""
import math

import torch
import torch.nn.functional as F

from .decode import gather_feat


def sigmoid_clamp(x):
    """utils.py:8-10 (out of place here; the reference's sigmoid_ is in place)."""
    return torch.clamp(torch.sigmoid(x), min=1e-4, max=1 - 1e-4)


def neg_loss(pred, gt):
    """losses.py:146-171."""
    pos_inds = gt.eq(1).float()
    neg_inds = gt.lt(1).float()
    neg_weights = torch.pow(1 - gt, 4)
    pos_loss = (torch.log(pred) * torch.pow(1 - pred, 2) * pos_inds).sum()
    neg_loss_ = (torch.log(1 - pred) * torch.pow(pred, 2) * neg_weights * neg_inds).sum()
    num_pos = pos_inds.sum()
    if num_pos == 0:
        return -neg_loss_
    return -(pos_loss + neg_loss_) / num_pos


def reg_l1_loss(output, mask, ind, target):
    """losses.py:817-830."""
    pred = gather_feat(output, ind)
    m = mask.unsqueeze(2).expand_as(pred).float()
    loss = F.l1_loss(pred * m, target * m, reduction="sum")
    return loss / (m.sum() + 1e-4)


def reg_sl1_loss(output, mask, ind, target):
    """RegLoss / _reg_loss (losses.py:201-216, 801-815), `--reg_loss sl1`."""
    pred = gather_feat(output, ind)
    num = mask.float().sum()
    m = mask.unsqueeze(2).expand_as(target).float()
    loss = F.smooth_l1_loss(pred * m, target * m, reduction="sum")
    return loss / (num + 1e-4)


# ----------------------------------------------------------------------------
# Weiler-Atherton clip, literal (losses.py:373-628)
# ----------------------------------------------------------------------------

def maxpts(n, m):
    return 8 * (n + m)


def _cart(P):
    return P[:, 0] * torch.cos(P[:, 1]), P[:, 0] * torch.sin(P[:, 1])


def _side(ax, ay, bx, by, qx, qy):
    """losses.py:378-397: 1 if R<0, 2 if R==0, 0 if R>0."""
    R = (bx - ax) * (qy - ay) - (by - ay) * (qx - ax)
    if R < 0:
        return 1
    if R == 0:
        return 2
    return 0


def _intersection(x1, y1, x2, y2, x3, y3, x4, y4):
    """losses.py:401-486: slope/intercept line intersection, returned in polar."""
    if x2 - x1 == 0:
        x = x1
        m2 = (y4 - y3) / (x4 - x3)
        b2 = y3 - m2 * x3
        y = m2 * x + b2
    elif x4 - x3 == 0:
        x = x3
        m1 = (y2 - y1) / (x2 - x1)
        b1 = y1 - m1 * x1
        y = m1 * x + b1
    else:
        m1 = (y2 - y1) / (x2 - x1)
        b1 = y1 - m1 * x1
        m2 = (y4 - y3) / (x4 - x3)
        b2 = y3 - m2 * x3
        x = (b2 - b1) / (m1 - m2)
        y = m1 * x + b1
    r = torch.sqrt(x * x + y * y)
    theta = torch.atan((y + 1e-8) / (x + 1e-8))
    if x < 0:
        theta = theta + math.pi
    elif y < 0:
        theta = theta + 2 * math.pi
    return torch.stack((r, theta))


def wa_scan(S, C):
    """Crossing scan (losses.py:506-579).  Returns inters (list of [2] tensors),
    inbounds, outbounds (lists of [j, i, k])."""
    n, m = S.shape[0], C.shape[0]
    sx, sy = _cart(S)
    cx, cy = _cart(C)
    inters, inb, outb = [], [], []
    for i in range(m):
        i0 = (i - 1) % m
        for j in range(n):
            j0 = (j - 1) % n
            te = _side(cx[i0], cy[i0], cx[i], cy[i], sx[j], sy[j])
            ts = _side(cx[i0], cy[i0], cx[i], cy[i], sx[j0], sy[j0])
            is_out = ts == 0 and te in (1, 2)
            is_in = te == 0 and ts in (1, 2)
            if not (is_out or is_in):
                continue
            a = _side(sx[j0], sy[j0], sx[j], sy[j], cx[i], cy[i])
            b = _side(sx[j0], sy[j0], sx[j], sy[j], cx[i0], cy[i0])
            if a == b:
                continue
            inters.append(_intersection(sx[j0], sy[j0], sx[j], sy[j],
                                        cx[i0], cy[i0], cx[i], cy[i]))
            (outb if is_out else inb).append([j, i, len(inters) - 1])
    return inters, inb, outb


def wa_traverse(n, m, inb, outb):
    """Traversal (losses.py:581-620) on indices only.  Returns a list of vertex
    descriptors ('S', j) / ('C', i) / ('I', k)."""
    out = []
    if len(inb) == 0:
        return out
    if len(outb) == 0:
        return out            # reference: IndexError -> defined as empty
    cap = maxpts(n, m)
    inb = [list(r) for r in inb]
    out_j = [r[0] for r in outb]
    used = 0
    while used < len(inb):
        stop_j, stop_i = inb[0][0], inb[0][1]
        j, i = stop_j, stop_i
        start = True
        while j != stop_j or i != stop_i or start:
            start = False
            while j not in out_j:
                out.append(("S", j))
                j = (j + 1) % n
                if len(out) >= cap:
                    return out[:cap]
            k = out_j.index(j)
            out.append(("I", outb[k][2]))
            i = outb[k][1]
            if len(inb) == 0:
                return out[:cap]    # reference: endless loop -> defined as stop
            in_i = [r[1] for r in inb]
            while i not in in_i:
                out.append(("C", i))
                i = (i + 1) % m
                if len(out) >= cap:
                    return out[:cap]
            k = in_i.index(i)
            j = inb[k][0]
            out.append(("I", inb[k][2]))
            del inb[k]
            used += 1
            if len(out) >= cap:
                return out[:cap]
    return out


def wa_clip(S, C):
    """Clip subject S[n,2] by clipping polygon C[m,2]; both (r, theta).  -> [K,2]."""
    inters, inb, outb = wa_scan(S, C)
    desc = wa_traverse(S.shape[0], C.shape[0], inb, outb)
    if not desc:
        return S.new_zeros((0, 2))
    rows = []
    for kind, idx in desc:
        rows.append(S[idx] if kind == "S" else C[idx] if kind == "C" else inters[idx])
    return torch.stack(rows)


def area(P):
    """losses.py:25-41: shoelace on (r cos t, r sin t) with the k=0 term doubled."""
    K = P.shape[0]
    if K == 0:
        return P.new_zeros(())
    x, y = _cart(P)
    dx = torch.cat((x, x))
    dy = torch.cat((y, y))
    left = (dx[0:K + 1] * dy[1:K + 2]).sum()
    right = (dy[0:K + 1] * dx[1:K + 2]).sum()
    return torch.abs(0.5 * (right - left))


def object_iou(pred_row, target_row):
    """losses.py:876-888 for one masked object."""
    p = pred_row.view(-1, 2)
    order = torch.sort(p[:, 1], 0)[1]
    sp = p[order]
    sp = torch.stack((torch.abs(sp[:, 0]), sp[:, 1]), 1)
    t = target_row.view(-1, 2)
    a_clip = area(wa_clip(sp, t))
    a_s, a_t = area(sp), area(t)
    inter = float(a_clip.item() == 0.0) * torch.min(a_s, a_t) + a_clip
    union = a_t + a_s - inter
    return inter / (union + 1e-6)


def order_adjust_mask(angles):
    """losses.py:892-899: which angle slots receive += 2*3.14 (depends on the
    ORIGINAL values only: the flag is raised by a positive angle, and edits
    happen only after it is raised)."""
    adj = torch.zeros_like(angles, dtype=torch.bool)
    seen = False
    for j in range(angles.shape[0]):
        if angles[j] > 0:
            seen = True
        if angles[j] < 0 and seen:
            adj[j] = True
    return adj


def poly_loss(output, mask, ind, target, poly_loss="l1", rep="cartesian", poly_order=False):
    """losses.py:833-959.  Returns loss or (loss, loss_order) like the reference."""
    pred = gather_feat(output, ind)
    B, M, D = pred.shape
    use_iou = poly_loss in ("iou", "l1+iou", "relu")
    loss = pred.new_zeros(())
    loss_order = pred.new_zeros(())
    adj = torch.zeros_like(pred)
    for b in range(B):
        for i in range(M):
            if not mask[b][i]:
                continue
            if use_iou:
                loss = loss + object_iou(pred[b, i], target[b, i])
            if poly_order:
                a = pred[b, i, 1::2]
                am = order_adjust_mask(a.detach())
                adj[b, i, 1::2] = am.float() * (2 * 3.14)
                a = a + adj[b, i, 1::2]
                Na = a.shape[0]
                for j in range(Na - 1):
                    d = a[j] - a[j:]
                    loss_order = loss_order + torch.clamp(d, min=0).sum()
    loss_order = loss_order / (10 * mask.sum() + 1e-4)
    pred = pred + adj                      # in-place edit seen by the L1 term
    if use_iou:
        loss = 1 - loss / (mask.sum() + 1e-6)
    loss_reg = pred.new_zeros(())
    if poly_loss in ("l1", "l1+iou", "relu"):
        mf = mask.unsqueeze(2).expand_as(pred).float()
        if poly_loss == "relu" and rep == "cartesian":
            d = (pred - target).abs()
            d = d * (d >= 20)
            loss_reg = (d * mf).abs().sum()
        elif rep == "cartesian":
            loss_reg = F.l1_loss(pred * mf, target * mf, reduction="sum")
        elif rep in ("polar", "polar_fixed"):
            ma = torch.tensor([1.0, 0.0] * (D // 2)).view(1, 1, D).expand_as(pred)
            loss_reg = F.l1_loss(pred * mf * ma, target * mf * ma, reduction="sum")
            if rep == "polar":
                loss_reg = loss_reg + torch.sum(
                    1 - torch.cos(pred * mf * (1 - ma) - target * mf * (1 - ma)))
        loss_reg = loss_reg / (mf.sum() + 1e-6)
    loss = loss + loss_reg
    if poly_order:
        return loss, loss_order
    return loss


def dense_poly_l1(pred, dense_poly, dense_mask):
    """trains/polydet.py:107-110 (`--dense_poly`): torch.nn.L1Loss(reduction='sum')(pred * mask, target * mask) /
    (mask.sum() + 1e-4)."""
    mask_weight = dense_mask.sum() + 1e-4
    return F.l1_loss(pred * dense_mask, dense_poly * dense_mask, reduction="sum") / mask_weight


def polydet_loss(outputs, batch, *, num_stacks=1, poly_loss_kind="l1", rep="cartesian",
                 poly_order=False, hm_weight=1.0, off_weight=1.0, poly_weight=1.0,
                 depth_weight=0.1, reg_offset=True, reg_loss="l1", mse_loss=False, dense_poly=False,
                 cat_spec_poly=False):
    """trains/polydet.py:38-162.  `outputs` = list of dicts of RAW head outputs; returns (loss, stats dict) and leaves
    outputs untouched.  cat_spec_poly (:103-106): PolyLoss gets the [B,M,C*2N] mask and its `if mask[batch][i]:`
    (models/losses.py:870) raises RuntimeError on the first object; dense_poly (:107-110): dense masked L1."""
    hm_l = off_l = poly_l = depth_l = order_l = 0
    for s in range(num_stacks):
        o = outputs[s]
        reg_crit = reg_l1_loss if reg_loss == "l1" else reg_sl1_loss
        depth_l = depth_l + reg_crit(o["pseudo_depth"], batch["reg_mask"], batch["ind"],
                                     batch["pseudo_depth"]) / num_stacks
        if mse_loss:                         # trains/polydet.py:23,44-46: MSELoss on the raw head
            hm_l = hm_l + F.mse_loss(o["hm"], batch["hm"]) / num_stacks
        else:
            hm = sigmoid_clamp(o["hm"])
            hm_l = hm_l + neg_loss(hm, batch["hm"]) / num_stacks
        if cat_spec_poly:
            if bool(batch["cat_spec_mask"][0][0]):       # raises for a row of more than one element, as the reference
                pass
            raise NotImplementedError("single-element cat_spec_mask rows: the reference's accidental 1x1 case")
        elif dense_poly:
            poly_l = poly_l + dense_poly_l1(o["poly"], batch["dense_poly"], batch["dense_poly_mask"]) / num_stacks
        else:
            r = poly_loss(o["poly"], batch["reg_mask"], batch["ind"], batch["poly"],
                          poly_loss_kind, rep, poly_order)
            if poly_order:
                poly_l = poly_l + r[0] / num_stacks
                order_l = order_l + r[1] / num_stacks
            else:
                poly_l = poly_l + r / num_stacks
        if reg_offset and off_weight > 0:
            off_l = off_l + reg_crit(o["reg"], batch["reg_mask"], batch["ind"],
                                     batch["reg"]) / num_stacks
    if poly_order:
        loss = hm_weight * hm_l + off_weight * off_l + poly_weight * (poly_l + order_l) \
            + depth_weight * depth_l
        stats = {"loss": loss, "hm_l": hm_l, "off_l": off_l, "poly_l": poly_l,
                 "order_l": order_l, "depth_l": depth_l}
    else:
        loss = hm_weight * hm_l + off_weight * off_l + poly_weight * poly_l \
            + depth_weight * depth_l
        stats = {"loss": loss, "hm_l": hm_l, "off_l": off_l, "poly_l": poly_l,
                 "depth_l": depth_l}
    return loss, stats
