"""Affine helpers of the detector's host tail (reference: src/lib/utils/image.py:19-66).

`cv2.getAffineTransform` is restated as the exact 3-point affine solve (numpy
float64); `transform_preds` is vectorised over rows instead of a Python loop that
recomputes the transform per call site.
"""
import ctypes

import numpy as np


def get_3rd_point(a, b):
    d = a - b
    return b + np.array([-d[1], d[0]], dtype=np.float32)


def get_dir(src_point, rot_rad):
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return [src_point[0] * cs - src_point[1] * sn, src_point[0] * sn + src_point[1] * cs]


def _solve_affine(src, dst):
    A = np.zeros((6, 6), dtype=np.float64)
    b = np.zeros(6, dtype=np.float64)
    for k in range(3):
        A[2 * k, 0:3] = (src[k, 0], src[k, 1], 1.0)
        A[2 * k + 1, 3:6] = (src[k, 0], src[k, 1], 1.0)
        b[2 * k:2 * k + 2] = dst[k]
    return np.linalg.solve(A, b).reshape(2, 3)


def get_affine_transform(center, scale, rot, output_size,
                         shift=np.array([0, 0], dtype=np.float32), inv=0):
    if not isinstance(scale, np.ndarray) and not isinstance(scale, list):
        scale = np.array([scale, scale], dtype=np.float32)
    scale_tmp = scale
    src_w = scale_tmp[0]
    dst_w, dst_h = output_size[0], output_size[1]
    src_dir = get_dir([0, src_w * -0.5], np.pi * rot / 180)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_tmp * shift
    src[1, :] = center + src_dir + scale_tmp * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir
    src[2:, :] = get_3rd_point(src[0, :], src[1, :])
    dst[2:, :] = get_3rd_point(dst[0, :], dst[1, :])
    if inv:
        return _solve_affine(np.float32(dst), np.float32(src))
    return _solve_affine(np.float32(src), np.float32(dst))


def affine_transform(pt, t):
    new_pt = np.array([pt[0], pt[1], 1.0], dtype=np.float32).T
    return np.dot(t, new_pt)[:2]


def apply_affine(coords, trans):
    """All rows at once: [n,2] fp32 points through a 2x3 float64 matrix."""
    pts = np.concatenate([coords[:, 0:2].astype(np.float32),
                          np.ones((coords.shape[0], 1), np.float32)], axis=1)
    return pts @ trans.T


def transform_preds(coords, center, scale, output_size):
    target = np.zeros(coords.shape)
    target[:, 0:2] = apply_affine(coords, get_affine_transform(center, scale, 0, output_size, inv=1))
    return target


def warp_affine_normalize(image_u8, trans, mean, std, dst_h, dst_w, flip_copy=False):
    """cv2.warpAffine(INTER_LINEAR, constant border 0) of an 8-bit HWC image followed by
    ((x / 255. - mean) / std) and HWC -> CHW, on the device (reference:
    src/lib/detectors/base_detector.py:66-87).  image_u8: uint8 [H,W,3] HIP tensor; trans: the
    forward 2x3 map (float64); returns fp32 [1 + flip_copy, 3, dst_h, dst_w] on the same device."""
    import torch

    from .. import _C
    if image_u8.dtype != torch.uint8 or image_u8.dim() != 3 or image_u8.shape[2] != 3:
        raise TypeError("warp_affine_normalize needs a uint8 [H,W,3] image tensor")
    t = (ctypes.c_double * 6)(*[float(v) for v in np.asarray(trans, dtype=np.float64).ravel()])
    m = (ctypes.c_float * 3)(*[float(v) for v in np.asarray(mean, dtype=np.float32).ravel()])
    sd = (ctypes.c_float * 3)(*[float(v) for v in np.asarray(std, dtype=np.float32).ravel()])
    out = torch.empty((2 if flip_copy else 1, 3, int(dst_h), int(dst_w)), dtype=torch.float32,
                      device=image_u8.device)
    _C.check(_C.lib().cp_preprocess_warp_normalize(
        _C.ptr(image_u8), image_u8.shape[0], image_u8.shape[1], ctypes.cast(t, ctypes.c_void_p),
        ctypes.cast(m, ctypes.c_void_p), ctypes.cast(sd, ctypes.c_void_p), int(dst_h), int(dst_w),
        1 if flip_copy else 0, _C.ptr(out), _C.stream()), "cp_preprocess_warp_normalize")
    return out


# ---- host-side helpers of the reference's sampler, kept for API parity (src/lib/utils/image.py) ----
# The accelerated path builds targets and augments inputs on the device (cp_polydet_targets,
# cp_preprocess_warp_normalize, cp_color_aug_normalize); these numpy forms serve code that imports the
# reference's names (`from utils.image import flip, color_aug, gaussian_radius, draw_umich_gaussian`).

def flip(img):
    """utils/image.py:16-17: mirror the last (width) axis of a CHW / NCHW array."""
    return img[..., ::-1].copy()


def gaussian_radius(det_size, min_overlap=0.7):
    """utils/image.py:95-115: the smallest of the three radii for which a box displaced by r still
    overlaps the ground truth with IoU >= min_overlap (the reference's literal quadratic roots,
    `(b + sqrt(b^2 - 4ac)) / 2` without the division by a)."""
    h, w = det_size
    s, p, k = h + w, w * h, min_overlap
    abc = ((1.0, s, p * (1 - k) / (1 + k)), (4.0, 2 * s, (1 - k) * p), (4 * k, -2 * k * s, (k - 1) * p))
    return min((b + np.sqrt(b ** 2 - 4 * a * c)) / 2 for a, b, c in abc)


def gaussian2D(shape, sigma=1):
    """utils/image.py:118-124: un-normalised Gaussian window, entries below eps * max zeroed."""
    ry, rx = [(n - 1.0) / 2.0 for n in shape]
    yy, xx = np.ogrid[-ry:ry + 1, -rx:rx + 1]
    g = np.exp(-(xx * xx + yy * yy) / (2 * sigma * sigma))
    g[g < np.finfo(g.dtype).eps * g.max()] = 0
    return g


def draw_umich_gaussian(heatmap, center, radius, k=1):
    """utils/image.py:126-141: max-composite a (2r+1)^2 Gaussian (sigma = (2r+1)/6) at an integer centre."""
    d = 2 * radius + 1
    g = gaussian2D((d, d), sigma=d / 6)
    cx, cy = int(center[0]), int(center[1])
    H, W = heatmap.shape[:2]
    l, r = min(cx, radius), min(W - cx, radius + 1)
    t, b = min(cy, radius), min(H - cy, radius + 1)
    dst = heatmap[cy - t:cy + b, cx - l:cx + r]
    src = g[radius - t:radius + b, radius - l:radius + r]
    if min(src.shape) > 0 and min(dst.shape) > 0:
        np.maximum(dst, src * k, out=dst)
    return heatmap


def grayscale(image):
    """cv2.cvtColor(image, COLOR_BGR2GRAY) on a float BGR image: 0.114 B + 0.587 G + 0.299 R."""
    return (image[..., 0] * np.float32(0.114) + image[..., 1] * np.float32(0.587)
            + image[..., 2] * np.float32(0.299)).astype(image.dtype)


def color_aug_params(data_rng, shuffle_rng=None):
    """The random draws of color_aug (utils/image.py:255-264) without touching an image: the order of
    brightness (0) / contrast (1) / saturation (2), their three alphas (1 + U(-0.4, 0.4), drawn in
    application order) and the lighting alphas N(0, 0.1)^3.  The device kernel applies them."""
    import random
    order = [0, 1, 2]
    (shuffle_rng or random).shuffle(order)
    alphas = [1.0 + data_rng.uniform(low=-0.4, high=0.4) for _ in order]
    light = data_rng.normal(scale=0.1, size=(3,))
    return order, alphas, light


def color_aug(data_rng, image, eig_val, eig_vec):
    """utils/image.py:255-264 on a float BGR HWC image, in place (host form of cp_color_aug_normalize)."""
    order, alphas, light = color_aug_params(data_rng)
    gs = grayscale(image)
    gs_mean = gs.mean()
    for op, alpha in zip(order, alphas):
        if op == 0:                                   # brightness_
            image *= alpha
        elif op == 1:                                 # contrast_: blend with the mean grey level
            image *= alpha
            image += gs_mean * (1 - alpha)
        else:                                         # saturation_: blend with the grey image
            image *= alpha
            image += gs[:, :, None] * (1 - alpha)
    image += np.dot(eig_vec, eig_val * light)         # lighting_
