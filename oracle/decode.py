"""Oracle: 3x3 max-pool NMS + top-K + polygon decode, torch CPU fp32.  TEST INFRASTRUCTURE.

Follows src/lib/models/decode.py:13-19 (_nms), :117-133 (_topk), :512-670
(polydet_decode) and src/lib/models/utils.py:12-26 (gather helpers).

Tie rule.  torch.topk's order among equal values is unspecified
("torch.topk ties").  The oracle fixes it: equal scores
are taken lowest flat index first (a stable descending sort).  On tie-free
inputs this is exactly torch.topk, which is what the golden vectors use.
"""
import math

import torch
import torch.nn.functional as F


def nms(heat, kernel=3):
    """decode.py:13-19."""
    pad = (kernel - 1) // 2
    hmax = F.max_pool2d(heat, (kernel, kernel), stride=1, padding=pad)
    keep = (hmax == heat).float()
    return heat * keep


def _topk_stable(x, K):
    """Top-K along the last dim, ties -> lowest index first."""
    order = torch.sort(x, dim=-1, descending=True, stable=True)[1][..., :K]
    return torch.gather(x, -1, order), order


def topk(scores, K):
    """decode.py:117-133 with the oracle's tie rule."""
    B, C, H, W = scores.shape
    topk_scores, topk_inds = _topk_stable(scores.view(B, C, -1), K)
    topk_inds = topk_inds % (H * W)
    topk_ys = (topk_inds / W).int().float()
    topk_xs = (topk_inds % W).int().float()
    topk_score, topk_ind = _topk_stable(topk_scores.view(B, -1), K)
    topk_clses = (topk_ind / K).int()
    g = lambda t: torch.gather(t.view(B, -1), 1, topk_ind)
    return topk_score, g(topk_inds), topk_clses, g(topk_ys), g(topk_xs)


def gather_feat(feat, ind):
    """utils.py:22-26 without the NHWC copy: feat[B,D,H,W], ind[B,M] -> [B,M,D]."""
    B, D = feat.shape[:2]
    f = feat.reshape(B, D, -1)
    idx = ind.unsqueeze(1).expand(B, D, ind.shape[1])
    return torch.gather(f, 2, idx).permute(0, 2, 1).contiguous()


def polydet_decode(heat, polys, depth, reg=None, K=100, rep="cartesian", cat_spec_poly=False):
    """decode.py:512-670.  heat is the activated heat map.  cat_spec_poly (:534-537): `nbr_points` is the MAP's
    width (`int(polys.shape[-1])` at :514), polys [B,K,Cp] is viewed as [B,K,cat,width] -- torch raises unless
    Cp == cat * width -- and the detection's class picks its block.

    Returns dets[B,K,2N+7] = [x1,y1,x2,y2,score,cls,poly(2N),depth] and the
    selected flat indices inds[B,K] (int64), classes[B,K] (int32)."""
    B = heat.shape[0]
    heat = nms(heat)
    scores, inds, clses, ys, xs = topk(heat, K)
    if reg is not None:
        r = gather_feat(reg, inds)
        xs = xs.view(B, K, 1) + r[:, :, 0:1]
        ys = ys.view(B, K, 1) + r[:, :, 1:2]
    else:
        xs = xs.view(B, K, 1) + 0.5
        ys = ys.view(B, K, 1) + 0.5
    p = gather_feat(polys, inds).clone()
    if cat_spec_poly:
        cat, nbr_points = heat.shape[1], int(polys.shape[-1])
        p = p.view(B, K, cat, nbr_points)
        ci = clses.view(B, K, 1, 1).expand(B, K, 1, nbr_points).long()
        p = p.gather(2, ci).view(B, K, nbr_points).clone()
    d = gather_feat(depth, inds).view(B, K, 1).float()
    n2 = p.shape[-1]
    if rep in ("polar", "polar_fixed"):
        # decode.py:582-614: math.cos/sin on a 0-d fp32 tensor = double precision
        # trig of the fp32 value, then an fp32 multiply by the fp32-rounded result.
        r_ = p[..., 0::2].clone()
        if rep == "polar_fixed":
            j = torch.arange(0, n2, 2, dtype=torch.float64)
            ang = (2 * 3.14 - 2 * 3.14 / n2 * j).expand(B, K, n2 // 2)
        else:
            ang = p[..., 1::2].double()
        p[..., 0::2] = r_ * torch.cos(ang).float()
        p[..., 1::2] = r_ * torch.sin(ang).float()
    p[..., 0::2] += xs
    p[..., 1::2] += ys
    px, py = p[..., 0::2], p[..., 1::2]
    bboxes = torch.cat([px.min(2, keepdim=True)[0], py.min(2, keepdim=True)[0],
                        px.max(2, keepdim=True)[0], py.max(2, keepdim=True)[0]], dim=2)
    dets = torch.cat([bboxes, scores.view(B, K, 1), clses.view(B, K, 1).float(), p, d], dim=2)
    return dets, inds, clses


def _check_polar_trig():
    """math.cos(float32 tensor) == cos in double of the fp32 value (documentation aid)."""
    t = torch.tensor(1.2345678, dtype=torch.float32)
    return math.cos(t) == math.cos(float(t))
