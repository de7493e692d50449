"""Target construction of the polydet sampler on the device.

The reference builds every training target in a per-object Python loop inside
PolydetDataset.__getitem__ (src/lib/datasets/sample/polydet.py:160-405, helpers in
src/lib/utils/image.py:62-65,95-141).  Here a loader worker only PACKS the raw annotations of
an image into flat arrays (`pack_annotations`, host, no arithmetic); after the batch reached
the GPU `build_targets` turns them into the batch dict of :425-449 with two HIP kernels
(cp_polydet_targets): hm, reg_mask, ind, poly, pseudo_depth, freq_mask, border_hm, wh, peak, reg.
"""
import numpy as np
import torch
import torch.utils.data

from ... import _C

_FIELDS = ("bbox", "poly", "cls_id", "pseudo_depth", "freq", "num_objs", "flipped", "width", "trans_output")


def pack_annotations(anns, trans_output, flipped, width, max_objs, nbr_points):
    """One image: list of {bbox [x,y,w,h], poly [2N], cls_id, pseudo_depth, freq} -> dict of
    fixed-size numpy arrays (what a DataLoader worker returns and default_collate stacks)."""
    n = min(len(anns), max_objs)
    out = {"bbox": np.zeros((max_objs, 4), np.float64), "poly": np.zeros((max_objs, 2 * nbr_points), np.float64),
           "cls_id": np.zeros((max_objs,), np.int32), "pseudo_depth": np.zeros((max_objs,), np.float32),
           "freq": np.zeros((max_objs,), np.float32), "num_objs": np.int32(n),
           "flipped": np.uint8(1 if flipped else 0), "width": np.int32(width),
           "trans_output": np.asarray(trans_output, dtype=np.float64).reshape(6)}
    for k in range(n):
        a = anns[k]
        if len(a["poly"]) != 2 * nbr_points:
            raise ValueError("annotation %d has %d polygon numbers, expected %d" % (k, len(a["poly"]), 2 * nbr_points))
        out["bbox"][k] = a["bbox"]
        out["poly"][k] = a["poly"]
        out["cls_id"][k] = a["cls_id"]
        out["pseudo_depth"][k] = a["pseudo_depth"]
        out["freq"][k] = a["freq"]
    return out


def collate(packed):
    """List of pack_annotations() dicts -> dict of stacked tensors (host)."""
    return {k: torch.from_numpy(np.stack([np.asarray(p[k]) for p in packed])) for k in _FIELDS}


def build_targets(raw, output_h, output_w, num_classes, rep="cartesian", no_reorder_flip=False,
                  with_border_hm=True, dense_poly=False, cat_spec_poly=False):
    """raw: dict of DEVICE tensors with the keys of pack_annotations, batched on dim 0.
    Returns the batch dict (device tensors) the polydet loss consumes.
    dense_poly (`--dense_poly`, sample/polydet.py:401-403,429-441): adds 'dense_poly' / 'dense_poly_mask' [B,2N,h,w]
    (cp_polydet_dense_targets) and drops 'poly', as the reference's dict does.  cat_spec_poly (:245-248,288-291,424-425):
    adds 'cat_spec_poly' / 'cat_spec_mask' [B,M,C*2N] -- every object's polygon row in its class's block -- and drops
    the keys the reference's cat-spec dict lacks (freq_mask, border_hm, wh)."""
    bbox = raw["bbox"]
    if not bbox.is_cuda:
        raise _C.NativeError("build_targets needs HIP device tensors (got %s); there is no CPU "
                             "fallback" % bbox.device)
    B, M = bbox.shape[0], bbox.shape[1]
    N = raw["poly"].shape[2] // 2
    dev = bbox.device
    shape = _C.TargetShape(B, M, N, int(num_classes), int(output_h), int(output_w), _C.REP[rep],
                           1 if no_reorder_flip else 0)
    t = {k: raw[k].contiguous() for k in _FIELDS}
    expect = {"bbox": torch.float64, "poly": torch.float64, "cls_id": torch.int32,
              "pseudo_depth": torch.float32, "freq": torch.float32, "num_objs": torch.int32,
              "flipped": torch.uint8, "width": torch.int32, "trans_output": torch.float64}
    for k, dt in expect.items():
        if t[k].dtype != dt:
            raise TypeError("raw[%r] must be %s (got %s)" % (k, dt, t[k].dtype))
    f32 = dict(dtype=torch.float32, device=dev)
    out = {"hm": torch.empty((B, num_classes, output_h, output_w), **f32),
           "border_hm": torch.empty((B, 1, output_h, output_w), **f32) if with_border_hm else None,
           "reg_mask": torch.empty((B, M), dtype=torch.uint8, device=dev),
           "ind": torch.empty((B, M), dtype=torch.int64, device=dev),
           "poly": torch.empty((B, M, 2 * N), **f32), "pseudo_depth": torch.empty((B, M, 1), **f32),
           "peak": torch.empty((B, M, 2), **f32), "reg": torch.empty((B, M, 2), **f32),
           "wh": torch.empty((B, M, 2), **f32), "freq_mask": torch.empty((B,), **f32)}
    lib = _C.lib()
    nws = lib.cp_polydet_targets_workspace_bytes(shape)
    ws = _C.workspace(nws, dev)
    _C.check(lib.cp_polydet_targets(
        shape, _C.ptr(t["bbox"]), _C.ptr(t["poly"]), _C.ptr(t["cls_id"]), _C.ptr(t["pseudo_depth"]),
        _C.ptr(t["freq"]), _C.ptr(t["num_objs"]), _C.ptr(t["flipped"]), _C.ptr(t["width"]),
        _C.ptr(t["trans_output"]), _C.ptr(out["hm"]), _C.ptr(out["border_hm"]), _C.ptr(out["reg_mask"]),
        _C.ptr(out["ind"]), _C.ptr(out["poly"]), _C.ptr(out["pseudo_depth"]), _C.ptr(out["peak"]),
        _C.ptr(out["reg"]), _C.ptr(out["wh"]), _C.ptr(out["freq_mask"]), _C.ptr(ws), nws, _C.stream()),
        "cp_polydet_targets")
    if not with_border_hm:
        del out["border_hm"]
    if cat_spec_poly:
        # index bookkeeping on the device (no arithmetic): row k -> block cls_id[k] of a [C, 2N] table, for the slots
        # the object kernel filled (h > 0 and w > 0: wh is set there and nowhere else)
        L2 = 2 * N
        live = (out["wh"][..., 0] > 0) & (out["wh"][..., 1] > 0)
        cls = t["cls_id"].long().clamp(0, num_classes - 1)
        onehot = torch.zeros((B, M, num_classes), dtype=torch.float32, device=dev)
        onehot.scatter_(2, cls.unsqueeze(2), 1.0)
        onehot = onehot * live.unsqueeze(2).float()
        out["cat_spec_poly"] = (onehot.unsqueeze(3) * out["poly"].unsqueeze(2)).reshape(B, M, num_classes * L2)
        out["cat_spec_mask"] = onehot.unsqueeze(3).expand(B, M, num_classes, L2).reshape(B, M, num_classes * L2).to(torch.uint8)
        for k in ("freq_mask", "border_hm", "wh"):
            out.pop(k, None)
    if dense_poly:
        dp = torch.empty((B, 2 * N, output_h, output_w), **f32)
        dm = torch.empty((B, 2 * N, output_h, output_w), **f32)
        _C.check(lib.cp_polydet_dense_targets(shape, _C.ptr(out["poly"]), _C.ptr(ws), nws, _C.ptr(dp), _C.ptr(dm),
                                              _C.stream()), "cp_polydet_dense_targets")
        out["dense_poly"], out["dense_poly_mask"] = dp, dm
        del out["poly"]
    return out


class PolydetDataset(torch.utils.data.Dataset):
    """Sampler mixed into a dataset class by get_dataset (reference: PolydetDataset.__getitem__,
    src/lib/datasets/sample/polydet.py:66-449), split between host and device:

      loader worker (here)   read the image, draw the augmentation (scale / centre / flip / colour
                             parameters, in the reference's order of random calls), mirror the 8-bit
                             image when flipped, pack the raw annotations -- no per-pixel arithmetic
      GPU (PolydetTrainer.prepare_batch)   cv2-style warp to the network input
                             (cp_preprocess_warp_normalize), colour augmentation + normalisation
                             (cp_color_aug_normalize), all training targets (cp_polydet_targets)

    Items of one batch must have equal image sizes (true for Cityscapes / KITTI crops of one size)."""

    def _get_border(self, border, size):
        i = 1
        while size - border // i <= border // i:
            i *= 2
        return border // i

    def __getitem__(self, index):
        import random

        from ...utils.image import color_aug_params, get_affine_transform
        opt = self.opt
        img_id = self.images[index]
        info = self.coco.loadImgs(ids=[img_id])[0]
        anns = self.coco.loadAnns(ids=self.coco.getAnnIds(imgIds=[img_id]))
        img = self.read_image(info["file_name"])
        height, width = img.shape[0], img.shape[1]
        c = np.array([width / 2.0, height / 2.0], dtype=np.float32)
        if opt.keep_res:
            input_h, input_w = (height | opt.pad) + 1, (width | opt.pad) + 1
            s = np.array([input_w, input_h], dtype=np.float32)
        else:
            s = max(height, width) * 1.0
            input_h, input_w = opt.input_h, opt.input_w
        flipped = False
        if self.split == "train":
            if not opt.not_rand_crop:
                s = s * np.random.choice(np.arange(0.6, 1.4, 0.1))
                w_border = self._get_border(128, width)
                h_border = self._get_border(128, height)
                c[0] = np.random.randint(low=w_border, high=width - w_border)
                c[1] = np.random.randint(low=h_border, high=height - h_border)
            else:
                sf, cf = opt.scale, opt.shift
                c[0] += s * np.clip(np.random.randn() * cf, -2 * cf, 2 * cf)
                c[1] += s * np.clip(np.random.randn() * cf, -2 * cf, 2 * cf)
                s = s * np.clip(np.random.randn() * sf + 1, 1 - sf, 1 + sf)
            if np.random.random() < opt.flip:
                flipped = True
                img = np.ascontiguousarray(img[:, ::-1, :])
                c[0] = width - c[0] - 1
        trans_input = get_affine_transform(c, s, 0, [input_w, input_h])
        color = np.zeros(10, dtype=np.float64)                 # [on, order x3, alpha x3, light x3]
        if self.split == "train" and not opt.no_color_aug:
            order, alphas, light = color_aug_params(self._data_rng, random)
            color[0] = 1.0
            color[1:4] = order
            color[4:7] = alphas
            color[7:10] = np.dot(self._eig_vec.astype(np.float64), self._eig_val.astype(np.float64) * light)
        output_h, output_w = input_h // opt.down_ratio, input_w // opt.down_ratio
        trans_output = get_affine_transform(c, s, 0, [output_w, output_h])
        packed_anns = []
        for a in anns[:self.max_objs]:
            packed_anns.append({"bbox": [a["bbox"][0], a["bbox"][1], a["bbox"][0] + a["bbox"][2],
                                         a["bbox"][1] + a["bbox"][3]],
                                "poly": a["poly"], "cls_id": int(self.cat_ids[a["category_id"]]),
                                "pseudo_depth": a.get("pseudo_depth", 0),
                                "freq": self.class_frequencies[self.class_name[a["category_id"]]]})
        item = pack_annotations(packed_anns, trans_output, flipped, width, self.max_objs, opt.nbr_points)
        item["image_u8"] = img
        item["trans_input"] = np.asarray(trans_input, dtype=np.float64).reshape(6)
        item["color"] = color
        item["input_hw"] = np.array([input_h, input_w], dtype=np.int32)
        if self.split != "train":
            item["meta"] = {"c": c, "s": np.float32(s) if np.isscalar(s) else s, "img_id": img_id,
                            "out_width": input_w, "out_height": input_h}
        return item


def build_inputs(image_u8, trans_input, color, mean, std, input_h, input_w):
    """Device half of the sampler's image path: image_u8 [B,H,W,3] uint8 HIP tensor (already mirrored
    where flipped), trans_input [B,6] / color [B,10] host arrays -> network input fp32 [B,3,h,w]."""
    import ctypes

    from ...utils.image import warp_affine_normalize
    if not image_u8.is_cuda:
        raise _C.NativeError("build_inputs needs HIP device tensors; there is no CPU fallback")
    lib = _C.lib()
    B = image_u8.shape[0]
    out = torch.empty((B, 3, int(input_h), int(input_w)), dtype=torch.float32, device=image_u8.device)
    nws = lib.cp_color_aug_workspace_bytes()
    ws = _C.workspace(nws, image_u8.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in np.asarray(mean, np.float32).ravel()])
    sd = (ctypes.c_float * 3)(*[float(v) for v in np.asarray(std, np.float32).ravel()])
    P = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
    for b in range(B):
        out[b] = warp_affine_normalize(image_u8[b], trans_input[b], (0.0, 0.0, 0.0), (1.0, 1.0, 1.0),
                                       input_h, input_w)[0]
        col = np.asarray(color[b], dtype=np.float64)
        order = (ctypes.c_int32 * 3)(*[int(v) for v in col[1:4]])
        alpha = (ctypes.c_float * 3)(*[float(np.float32(v)) for v in col[4:7]])
        light = (ctypes.c_double * 3)(*[float(v) for v in col[7:10]])
        _C.check(lib.cp_color_aug_normalize(_C.c_void_p(out[b].data_ptr()), int(input_h) * int(input_w),
                                            1 if col[0] != 0 else 0, P(order), P(alpha), P(light), P(m), P(sd),
                                            _C.ptr(ws), nws, _C.stream()), "cp_color_aug_normalize")
    return out
