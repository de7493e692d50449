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
