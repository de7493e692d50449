"""Mirror of src/lib/models/decode.py for polydet: _nms, _topk, polydet_decode.

polydet_decode is ONE call into the HIP library (fused 3x3 NMS + exact top-K +
gather + polar conversion + bbox).  Ties are ordered lowest (class, y, x) first.
"""
import torch

from .. import _C


def _decode_native(heat, polys, depth, reg, K, rep, cat_spec_poly=False):
    L = _C.lib()
    B, C, H, W = heat.shape
    N2 = polys.shape[1] // C if cat_spec_poly else polys.shape[1]
    heat, polys, depth = heat.contiguous(), polys.contiguous(), depth.contiguous()
    reg = reg.contiguous() if reg is not None else None
    for t in (heat, polys, depth) + ((reg,) if reg is not None else ()):
        if t.dtype != torch.float32:
            raise _C.NativeError("polydet_decode expects float32 tensors")
    dev = heat.device
    dets = torch.empty((B, K, N2 + 7), dtype=torch.float32, device=dev)
    inds = torch.empty((B, K), dtype=torch.int64, device=dev)
    clses = torch.empty((B, K), dtype=torch.int32, device=dev)
    nb = L.cp_polydet_decode_workspace_bytes(B, C, H, W, K)
    ws = _C.workspace(nb, dev)
    rc = L.cp_polydet_decode_ex(_C.ptr(heat), _C.ptr(polys), _C.ptr(depth), _C.ptr(reg), B, C, H, W,
                                N2, K, _C.REP[rep], 1 if cat_spec_poly else 0, _C.ptr(dets), _C.ptr(inds),
                                _C.ptr(clses), _C.ptr(ws), ws.numel(), _C.stream())
    _C.check(rc, "cp_polydet_decode_ex")
    return dets, inds, clses


def polydet_decode(heat, polys, depth, reg=None, cat_spec_poly=False, K=100, rep="cartesian",
                   return_inds=False):
    """decode.py:512-670.  heat is the ACTIVATED heat map [B,C,h,w]; returns
    dets[B,K,2N+7] = [x1,y1,x2,y2,score,cls,poly(2N),depth]."""
    if cat_spec_poly:
        # decode.py:514,534-535: `nbr_points = int(polys.shape[-1])` is read off the MAP (its width), and
        # `polys.view(batch, K, cat, nbr_points)` then only succeeds when the head has cat * width channels; the
        # reference raises torch's view error otherwise, and so does this mirror (same type, same message form).
        B, cat, _, W = heat.shape
        if polys.shape[1] != cat * W:
            raise RuntimeError("shape '[%d, %d, %d, %d]' is invalid for input of size %d"
                               % (B, K, cat, W, B * K * polys.shape[1]))
    dets, inds, clses = _decode_native(heat, polys, depth, reg, K, rep, cat_spec_poly)
    if return_inds:
        return dets, inds, clses
    return dets


def _nms(heat, kernel=3):
    """decode.py:13-19 (stand-alone form, device torch ops; the fused kernel does
    not materialise this map)."""
    pad = (kernel - 1) // 2
    hmax = torch.nn.functional.max_pool2d(heat, (kernel, kernel), stride=1, padding=pad)
    return heat * (hmax == heat).float()


def _topk(scores, K=40):
    """decode.py:117-133 on an ALREADY suppressed map.  Implemented through the
    decode kernel's exact selection so the tie order is the documented one."""
    B, C, H, W = scores.shape
    z1 = torch.zeros((B, 2, H, W), dtype=torch.float32, device=scores.device)
    # A suppressed map is a fixed point of the NMS test wherever it is non-zero
    # only if peaks are isolated; run selection on it directly via the kernel by
    # passing it as heat: non-peaks are 0 and stay 0, peaks stay peaks.
    dets, inds, clses = _decode_native(scores, z1, z1[:, :1], z1, K, "cartesian")
    ys = (inds / W).int().float()
    xs = (inds % W).int().float()
    return dets[:, :, 4], inds, clses, ys, xs
