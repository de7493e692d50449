"""Oracle: training-target construction of the polydet sample (numpy / Python floats).
TEST INFRASTRUCTURE.

Follows PolydetDataset.__getitem__ from the point where the annotations and the output affine
are known (src/lib/datasets/sample/polydet.py:138-449) and the helpers it calls:
gaussian_radius / gaussian2D / draw_umich_gaussian (src/lib/utils/image.py:95-141) and
affine_transform (:62-65).  PINNED: the helpers by tests/golden/targets_prims.npz (made by
running the reference's own utils/image.py, tests/golden/gen_targets_golden.py), the object
loop by tests/golden/sampler_*.npz -- the reference's own PolydetDataset.__getitem__ run on its
KITTIPolyStuff/BBoxes/val16.json annotations (tests/golden/gen_sampler_golden.py: blank-image
stand-ins for the absent cv2, the targets do not depend on pixel values), seven cases incl.
random crop / shift, mirroring with and without vertex re-ordering, polar and polar_fixed;
tests/test_targets.py::test_object_loop_matches_reference_sampler holds every array bit-exact.

Round 4: --dense_poly (draw_dense_reg, utils/image.py:176-204, called at sample/polydet.py:401-403; dict edit :429-441)
and --cat_spec_poly (:245-248, 288-291, 424-425) are restated too and pinned by sampler_cart_dense.npz /
sampler_cart_catspec.npz (the reference's own __getitem__ with those flags).
Not restated (flags the accelerated path refuses): --elliptical_gt, --mse_loss; `fg` (a warped instance image) is an
input the loss never reads.
"""
import math

import numpy as np


def gaussian_radius(det_size, min_overlap=0.7):
    """utils/image.py:95-115."""
    height, width = det_size
    a1 = 1
    b1 = (height + width)
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    sq1 = np.sqrt(b1 ** 2 - 4 * a1 * c1)
    r1 = (b1 + sq1) / 2
    a2 = 4
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    sq2 = np.sqrt(b2 ** 2 - 4 * a2 * c2)
    r2 = (b2 + sq2) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    sq3 = np.sqrt(b3 ** 2 - 4 * a3 * c3)
    r3 = (b3 + sq3) / 2
    return min(r1, r2, r3)


def gaussian2D(shape, sigma=1):
    """utils/image.py:118-124."""
    m, n = [(ss - 1.) / 2. for ss in shape]
    y, x = np.ogrid[-m:m + 1, -n:n + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def draw_umich_gaussian(heatmap, center, radius, k=1):
    """utils/image.py:126-141: max-composite a (2r+1)^2 Gaussian, sigma = (2r+1)/6."""
    diameter = 2 * radius + 1
    gaussian = gaussian2D((diameter, diameter), sigma=diameter / 6)
    x, y = int(center[0]), int(center[1])
    height, width = heatmap.shape[0:2]
    left, right = min(x, radius), min(width - x, radius + 1)
    top, bottom = min(y, radius), min(height - y, radius + 1)
    masked_heatmap = heatmap[y - top:y + bottom, x - left:x + right]
    masked_gaussian = gaussian[radius - top:radius + bottom, radius - left:radius + right]
    if min(masked_gaussian.shape) > 0 and min(masked_heatmap.shape) > 0:
        np.maximum(masked_heatmap, masked_gaussian * k, out=masked_heatmap)
    return heatmap


def draw_dense_reg(regmap, heatmap, center, value, radius):
    """utils/image.py:176-204 (is_offset=False): inside the object's splat window, regmap <- value wherever the
    float64 Gaussian is >= the (float32) class-maximum heat map."""
    diameter = 2 * radius + 1
    gaussian = gaussian2D((diameter, diameter), sigma=diameter / 6)
    value = np.array(value, dtype=np.float32).reshape(-1, 1, 1)
    dim = value.shape[0]
    reg = np.ones((dim, diameter * 2 + 1, diameter * 2 + 1), dtype=np.float32) * value
    x, y = int(center[0]), int(center[1])
    height, width = heatmap.shape[0:2]
    left, right = min(x, radius), min(width - x, radius + 1)
    top, bottom = min(y, radius), min(height - y, radius + 1)
    masked_heatmap = heatmap[y - top:y + bottom, x - left:x + right]
    masked_regmap = regmap[:, y - top:y + bottom, x - left:x + right]
    masked_gaussian = gaussian[radius - top:radius + bottom, radius - left:radius + right]
    masked_reg = reg[:, radius - top:radius + bottom, radius - left:radius + right]
    if min(masked_gaussian.shape) > 0 and min(masked_heatmap.shape) > 0:
        idx = (masked_gaussian >= masked_heatmap).reshape(1, masked_gaussian.shape[0], masked_gaussian.shape[1])
        masked_regmap = (1 - idx) * masked_regmap + idx * masked_reg
    regmap[:, y - top:y + bottom, x - left:x + right] = masked_regmap
    return regmap


def affine_transform(pt, t):
    """utils/image.py:62-65: the point is cast to float32, the product runs in float64."""
    new_pt = np.array([pt[0], pt[1], 1.], dtype=np.float32).T
    new_pt = np.dot(t, new_pt)
    return new_pt[:2]


def build_targets(anns, trans_output, flipped, width, output_h, output_w, num_classes, max_objs,
                  nbr_points, rep="cartesian", no_reorder_flip=False, dense_poly=False, cat_spec_poly=False):
    """One image.  anns: list of dicts {bbox: [x,y,w,h], poly: [2N numbers], cls_id: int,
    pseudo_depth: float, freq: float (the class frequency the reference looks up by name)}.
    Returns the sample dict of sample/polydet.py:425-449 (without 'input' and 'fg')."""
    num_objs = min(len(anns), max_objs)
    hm = np.zeros((num_classes, output_h, output_w), dtype=np.float32)
    wh = np.zeros((max_objs, 2), dtype=np.float32)
    border_hm = np.zeros((1, output_h, output_w), dtype=np.float32)
    pseudo_depth = np.zeros((max_objs, 1), dtype=np.float32)
    poly = np.zeros((max_objs, nbr_points * 2), dtype=np.float32)
    reg = np.zeros((max_objs, 2), dtype=np.float32)
    ind = np.zeros((max_objs), dtype=np.int64)
    peak = np.zeros((max_objs, 2), dtype=np.float32)
    reg_mask = np.zeros((max_objs), dtype=np.uint8)
    freq_mask = np.zeros((max_objs), dtype=np.float32)
    dense = np.zeros((nbr_points * 2, output_h, output_w), dtype=np.float32)
    cs_poly = np.zeros((max_objs, num_classes * nbr_points * 2), dtype=np.float32)
    cs_mask = np.zeros((max_objs, num_classes * nbr_points * 2), dtype=np.uint8)
    for k in range(num_objs):
        ann = anns[k]
        box = ann["bbox"]
        bbox = np.array([box[0], box[1], box[0] + box[2], box[1] + box[3]], dtype=np.float32)
        pseudo_depth[k] = ann["pseudo_depth"]
        cls_id = int(ann["cls_id"])
        pts = list(ann["poly"])
        if flipped:
            bbox[[0, 2]] = width - bbox[[2, 0]] - 1
            for i in range(0, len(pts), 2):
                pts[i] = width - pts[i] - 1
            not_flipped = list(pts)
            first_angle = len(pts) // 4
            if not no_reorder_flip:
                for i in range(0, len(pts) // 4 + 2, 2):
                    pts[i] = not_flipped[first_angle - i]
                    pts[i + 1] = not_flipped[first_angle - i + 1]
                for i in range(2, 3 * len(pts) // 4, 2):
                    pts[first_angle + i] = not_flipped[len(pts) - i]
                    pts[first_angle + i + 1] = not_flipped[len(pts) - i + 1]
        for i in range(0, len(pts), 2):
            pts[i], pts[i + 1] = affine_transform([pts[i], pts[i + 1]], trans_output)
            pts[i] = np.clip(pts[i], 0, output_w - 1)
            pts[i + 1] = np.clip(pts[i + 1], 0, output_h - 1)
        bbox[:2] = affine_transform(bbox[:2], trans_output)
        bbox[2:] = affine_transform(bbox[2:], trans_output)
        bbox[[0, 2]] = np.clip(bbox[[0, 2]], 0, output_w - 1)
        bbox[[1, 3]] = np.clip(bbox[[1, 3]], 0, output_h - 1)
        h, w = bbox[3] - bbox[1], bbox[2] - bbox[0]
        if h > 0 and w > 0:
            radius = gaussian_radius((math.ceil(h), math.ceil(w)))
            radius = max(0, int(radius))
            ct = np.array([(bbox[0] + bbox[2]) / 2, (bbox[1] + bbox[3]) / 2], dtype=np.float32)
            mass_cx, mass_cy = 0, 0
            for i in range(0, len(pts), 2):
                mass_cx += pts[i]
                mass_cy += pts[i + 1]
            ct[0] = mass_cx / (len(pts) / 2)
            ct[1] = mass_cy / (len(pts) / 2)
            ct_int = ct.astype(np.int32)
            draw_umich_gaussian(hm[cls_id], ct_int, radius)
            wh[k] = 1. * w, 1. * h
            for i in range(0, len(pts), 2):
                draw_umich_gaussian(border_hm[0], (int(pts[i]), int(pts[i + 1])), radius)
                if rep == "cartesian":
                    poly[k][i] = pts[i] - ct[0]
                    poly[k][i + 1] = pts[i + 1] - ct[1]
                else:                                  # `elif rep == 'polar' or 'polar_fixed'` is always true
                    x = pts[i] - ct[0]
                    y = pts[i + 1] - ct[1]
                    r = math.sqrt(x * x + y * y)
                    theta = math.atan((y + 1e-8) / (x + 1e-8))
                    if x < 0:
                        theta = theta + math.pi
                    elif y < 0:
                        theta = theta + 2 * math.pi
                    poly[k][i] = r
                    poly[k][i + 1] = theta
                if cat_spec_poly:                      # :245-248 / :288-291: the same two numbers in the class's block
                    o = cls_id * (nbr_points * 2) + i
                    cs_poly[k][o], cs_poly[k][o + 1] = poly[k][i], poly[k][i + 1]
                    cs_mask[k][o:o + 2] = 1
            peak[k] = ct
            ind[k] = ct_int[1] * output_w + ct_int[0]
            reg[k] = ct - ct_int
            if rep == "polar" and poly[k][1] > poly[k][5]:
                reg_mask[k] = 0
            else:
                reg_mask[k] = 1
            freq_mask[k] = ann["freq"]
            if dense_poly:                             # :401-403
                draw_dense_reg(dense, hm.max(axis=0), ct_int, poly[k], radius)
    if np.count_nonzero(freq_mask) == 0:
        freq_mean = 1.0
    else:
        freq_mean = np.sum(freq_mask) / (np.count_nonzero(freq_mask))
    if cat_spec_poly:                                  # :424-425: this dict has no freq_mask / border_hm / wh
        ret = {"hm": hm, "reg_mask": reg_mask, "ind": ind, "poly": poly, "cat_spec_poly": cs_poly,
               "cat_spec_mask": cs_mask, "pseudo_depth": pseudo_depth, "peak": peak, "reg": reg}
    else:
        ret = {"hm": hm, "reg_mask": reg_mask, "ind": ind, "poly": poly, "pseudo_depth": pseudo_depth,
               "freq_mask": freq_mean, "border_hm": border_hm, "wh": wh, "peak": peak, "reg": reg}
    if dense_poly:                                     # :429-441
        dmask = dense.copy()
        dmask[dmask != 0] = 1
        ret.update({"dense_poly": dense, "dense_poly_mask": dmask})
        del ret["poly"]
    return ret
