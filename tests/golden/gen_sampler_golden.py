#!/usr/bin/env python3
"""Generate tests/golden/sampler_*.npz by running the REFERENCE's own PolydetDataset.__getitem__
(src/lib/datasets/sample/polydet.py:66-449) in this container -- the object loop that builds the training targets
(hm / ind / reg / reg_mask / poly / pseudo_depth / peak / wh / border_hm / freq_mask), including its random crop /
scale / flip draws -- on the annotations the reference ships (KITTIPolyStuff/BBoxes/val16.json).

The targets do not depend on pixel values: cv2 is absent here, and the stand-in below returns blank images of the
right shapes for imread / resize / warpAffine (its getAffineTransform is the closed-form 3-point solve the other
golden generators use).  `bresenham` (imported by the reference file, never called on this path) is an empty module,
pycocotools is replaced by a 20-line in-memory index over the same JSON.  Build container only (needs
/root/reference); the fixtures hold the draws, the transforms and the expected arrays -- no reference source.

Usage:  python tests/golden/gen_sampler_golden.py
"""
import copy
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src/lib"
ANN = "/root/reference/KITTIPolyStuff/BBoxes/val16.json"
IMG_H, IMG_W = 375, 1242                      # KITTI frames


def _stubs():
    sys.path.insert(0, ROOT)
    from oracle import post as opost
    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR = 1
    cv2.getAffineTransform = lambda s, d: opost.affine_from_3pts(np.asarray(s), np.asarray(d))
    cv2.imread = lambda path, flag=1: np.zeros((IMG_H, IMG_W, 3) if flag != 0 else (IMG_H, IMG_W), np.uint8)
    cv2.resize = lambda img, size, **kw: np.zeros((size[1], size[0]) + img.shape[2:], img.dtype)
    cv2.warpAffine = lambda img, m, size, **kw: np.zeros((size[1], size[0]) + img.shape[2:], img.dtype)
    sys.modules["cv2"] = cv2
    sys.modules["bresenham"] = types.ModuleType("bresenham")
    sys.path.insert(0, REF)


class Index:
    """The three pycocotools calls of the sampler over the reference's own annotation file."""

    def __init__(self, path):
        d = json.load(open(path))
        self.imgs = {im["id"]: im for im in d["images"]}
        self.by_img = {}
        for a in d["annotations"]:
            self.by_img.setdefault(a["image_id"], []).append(a)
        self.anns = {a["id"]: a for a in d["annotations"]}

    def loadImgs(self, ids):
        return [self.imgs[i] for i in ids]

    def getAnnIds(self, imgIds):
        return [a["id"] for i in imgIds for a in self.by_img.get(i, [])]

    def loadAnns(self, ids):
        return [copy.deepcopy(self.anns[i]) for i in ids]      # (the sampler edits the polygons in place)


CASES = [
    # name, rep, split, not_rand_crop, flip probability, no_reorder_flip, image ids, seed
    ("cart_crop", "cartesian", "train", False, 0.5, False, [0, 1, 2, 3, 4], 11),
    ("cart_flip", "cartesian", "train", False, 1.0, False, [5, 6, 7], 12),
    ("cart_shift", "cartesian", "train", True, 0.0, False, [8, 9, 0], 13),
    ("cart_noreorder", "cartesian", "train", False, 1.0, True, [1, 2], 14),
    ("polar_flip", "polar", "train", False, 1.0, False, [3, 4, 5], 15),
    ("polar_fixed", "polar_fixed", "train", False, 0.5, False, [6, 7], 16),
    ("cart_val", "cartesian", "val", False, 0.0, False, [0, 9], 17),
    # round 4: the two switches of sample/polydet.py:245-248 / 401-403 (a 9th field = extra opt flags)
    ("cart_dense", "cartesian", "train", False, 0.5, False, [0, 2, 5], 18, {"dense_poly": True}),
    ("cart_catspec", "cartesian", "train", False, 0.5, False, [1, 3], 19, {"cat_spec_poly": True}),
    ("polar_catspec", "polar", "train", False, 0.0, False, [4], 20, {"cat_spec_poly": True}),
]


def main():
    _stubs()
    from datasets.sample.polydet import PolydetDataset           # the reference's own class
    index = Index(ANN)

    class DS(PolydetDataset):
        num_classes = 8
        mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 1, 3)
        std = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 1, 3)

        def __init__(self, opt, split):
            self.opt, self.split = opt, split
            self.img_dir = ""
            self.max_objs = 128
            self.class_name = ["__background__", "person", "rider", "car", "truck", "bus", "train", "motorcycle", "bicycle"]
            self.class_frequencies = {"person": 0.15, "rider": 0.03, "car": 0.20, "truck": 0.03, "bus": 0.03,
                                      "train": 0.03, "motorcycle": 0.03, "bicycle": 0.03}
            self.cat_ids = {v: i for i, v in enumerate([1, 2, 3, 4, 5, 6, 7, 8])}
            self._data_rng = np.random.RandomState(123)
            self.coco = index
            self.images = sorted(index.imgs)

    only = set(sys.argv[1:])
    for case in CASES:
        name, rep, split, not_rand_crop, flip, no_reorder, ids, seed = case[:8]
        extra = case[8] if len(case) > 8 else {}
        if only and name not in only:
            continue
        opt = types.SimpleNamespace(
            nbr_points=16, keep_res=False, pad=31, input_h=384, input_w=1280, down_ratio=4, not_rand_crop=not_rand_crop,
            scale=0.4, shift=0.1, flip=flip, no_color_aug=True, mse_loss=False, elliptical_gt=False, hm_gauss=4,
            rep=rep, cat_spec_poly=False, dense_poly=False, no_reorder_flip=no_reorder, debug=1, reg_offset=True)
        for k, v in extra.items():
            setattr(opt, k, v)
        ds = DS(opt, split)
        out = {"rep": np.array(rep), "split": np.array(split), "no_reorder_flip": np.array(no_reorder), "seed": np.array(seed),
               "not_rand_crop": np.array(not_rand_crop), "flip_prob": np.array(flip), "scale": np.array(0.4), "shift": np.array(0.1),
               "img_ids": np.array(ids), "img_hw": np.array([IMG_H, IMG_W]), "out_hw": np.array([384 // 4, 1280 // 4]),
               "class_freq": np.array([ds.class_frequencies[n] for n in ds.class_name[1:]], np.float64),
               "dense_poly": np.array(bool(extra.get("dense_poly", False))),
               "cat_spec_poly": np.array(bool(extra.get("cat_spec_poly", False)))}
        np.random.seed(seed)
        for j, i in enumerate(ids):
            r = ds[i]
            for k in ("hm", "reg_mask", "ind", "poly", "pseudo_depth", "border_hm", "wh", "peak", "reg", "cat_spec_poly",
                      "cat_spec_mask", "dense_poly", "dense_poly_mask"):
                if k in r:                       # (the cat-spec dict has no border_hm / wh / freq_mask, the dense one no poly)
                    out["s%d_%s" % (j, k)] = np.asarray(r[k])
            out["s%d_keys" % j] = np.array(sorted(k for k in r if k not in ("input", "meta", "fg")))
            if "freq_mask" in r:
                out["s%d_freq_mask" % j] = np.float64(r["freq_mask"])
            out["s%d_c" % j] = np.asarray(r["meta"]["c"], np.float32)        # what the draws came to
            out["s%d_s" % j] = np.float64(r["meta"]["s"])
            out["s%d_gt_det" % j] = np.asarray(r["meta"]["gt_det"])
            # the sample's input annotations (data of the reference's KITTIPolyStuff/BBoxes/val16.json)
            anns = index.loadAnns(index.getAnnIds([i]))
            out["s%d_ann_bbox" % j] = np.array([a["bbox"] for a in anns], np.float64).reshape(-1, 4)
            out["s%d_ann_poly" % j] = np.array([a["poly"] for a in anns], np.float64).reshape(len(anns), -1)
            out["s%d_ann_cat" % j] = np.array([a["category_id"] for a in anns], np.int64)
            out["s%d_ann_depth" % j] = np.array([a["pseudo_depth"] for a in anns], np.float64)
        path = os.path.join(HERE, "sampler_%s.npz" % name)
        np.savez_compressed(path, **out)
        print("wrote", os.path.basename(path), {k: v.shape for k, v in out.items() if k.startswith("s0_")})


if __name__ == "__main__":
    main()
