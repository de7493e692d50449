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
                  with_border_hm=True):
    """raw: dict of DEVICE tensors with the keys of pack_annotations, batched on dim 0.
    Returns the batch dict (device tensors) the polydet loss consumes."""
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
    return out
