"""polydet_post_process (reference: src/lib/utils/post_process.py:105-122): map
bbox corners and polygon vertices from output-map to image coordinates and split
by class into {1..C: rows [x1,y1,x2,y2,score,poly(2N),depth]}.  The inverse affine
is computed ONCE per image and applied to every vertex column in one product."""
import numpy as np

from .image import apply_affine, get_affine_transform


def polydet_post_process(dets, c, s, h, w, num_classes):
    ret = []
    for i in range(dets.shape[0]):
        trans = get_affine_transform(c[i], s[i], 0, (w, h), inv=1)
        d = dets[i]
        K = d.shape[0]
        cols = [0, 2] + list(range(6, d.shape[-1] - 1, 2))
        pts = np.stack([d[:, j:j + 2] for j in cols], axis=1).reshape(-1, 2)
        moved = apply_affine(pts, trans).reshape(K, len(cols), 2)
        for n, j in enumerate(cols):
            d[:, j:j + 2] = moved[:, n]
        classes = d[:, 5]
        top = {}
        for j in range(num_classes):
            inds = classes == j
            top[j + 1] = np.concatenate([d[inds, :4].astype(np.float32),
                                         d[inds, 4:5].astype(np.float32),
                                         d[inds, 6:].astype(np.float32)], axis=1).tolist()
        ret.append(top)
    return ret


_TRANS_CACHE = {}


def _inverse_transforms(c, s, h, w, device):
    """[B,6] float64 device tensor of the output-map -> image affines, cached per (c, s, h, w)."""
    import torch
    key = (tuple(tuple(np.asarray(ci, dtype=np.float64).ravel().tolist()) for ci in c),
           tuple(tuple(np.asarray(si, dtype=np.float64).ravel().tolist()) for si in s), int(h), int(w),
           str(device))
    t = _TRANS_CACHE.get(key)
    if t is None:
        m = np.stack([get_affine_transform(c[i], s[i], 0, (w, h), inv=1).reshape(6)
                      for i in range(len(c))]).astype(np.float64)
        t = torch.from_numpy(m).to(device)
        if len(_TRANS_CACHE) > 64:
            _TRANS_CACHE.clear()
        _TRANS_CACHE[key] = t
    return t


def polydet_post_process_device(dets, c, s, h, w, num_classes, scale=1.0):
    """polydet_post_process + the `/ scale` of PolydetDetector.post_process with the affine of
    every box corner and vertex done on the device (cp_polydet_post_process); one device -> host
    copy of the [B,K,2N+7] rows, then the per-class split.  dets: HIP tensor [B,K,2N+7].
    Returns [{1..C: float32 [n, 2N+6]}] per image (rows x1,y1,x2,y2,score,poly(2N),depth)."""
    import torch

    from .. import _C
    dets = dets.contiguous()
    B, K, ncols = dets.shape
    trans = _inverse_transforms(c, s, h, w, dets.device)
    out = torch.empty_like(dets)
    _C.check(_C.lib().cp_polydet_post_process(_C.ptr(dets), _C.ptr(trans), float(scale), B, K, ncols,
                                              _C.ptr(out), _C.stream()), "cp_polydet_post_process")
    rows = out.cpu().numpy()
    ret = []
    for i in range(B):
        d = rows[i]
        cls = d[:, 5]
        keep = np.concatenate([d[:, :5], d[:, 6:]], axis=1)
        ret.append({j + 1: keep[cls == j] for j in range(num_classes)})
    return ret
