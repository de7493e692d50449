"""the Cityscapes result writer (src/lib/datasets/dataset/cityscapes.py:196-283).
CPU: the oracle (same PIL calls as the reference) against the committed fixtures, the restated
`bresenham`, selection / ordering rules.  GPU: cp_instance_masks mask for mask against the fixtures, and the
written files of the host mirror against the oracle's."""
import os

import numpy as np
import pytest

from oracle import writer as ow

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ["star16", "mixed32", "selfcross16", "small16"]
CLASS_NAME = ["__background__", "person", "rider", "car", "truck", "bus", "train", "motorcycle", "bicycle", "pole",
              "traffic sign", "traffic light"]
LABEL_TO_ID = {"person": 24, "rider": 25, "car": 26, "truck": 27, "bus": 28, "train": 31, "motorcycle": 32,
               "bicycle": 33, "pole": -1, "traffic sign": -1, "traffic light": -1}


def _load(name):
    z = np.load(os.path.join(HERE, "golden", "writer_%s.npz" % name), allow_pickle=False)
    det = {int(k[4:]): z[k] for k in z.files if k.startswith("det_")}
    masks = np.unpackbits(z["packed"], axis=2)[:, :, :2048].astype(np.uint8) * 255
    return det, masks, z["keep"], [str(v) for v in z["lines"]], z["order_depth"]


def test_bresenham_restatement_known_answers():
    assert list(ow.bresenham(0, 0, 3, 1)) == [(0, 0), (1, 0), (2, 1), (3, 1)]
    assert list(ow.bresenham(2, 2, 2, 2)) == [(2, 2)]
    assert list(ow.bresenham(0, 0, -2, -5)) == [(0, 0), (0, -1), (-1, -2), (-1, -3), (-2, -4), (-2, -5)]
    assert list(ow.bresenham(5, 1, 1, 1)) == [(5, 1), (4, 1), (3, 1), (2, 1), (1, 1)]
    pts = list(ow.bresenham(-3, 7, 11, -2))
    assert pts[0] == (-3, 7) and pts[-1] == (11, -2) and len(pts) == 15


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_the_fixtures(name):
    det, masks, keep, lines, order = _load(name)
    params = ow.image_instances(det, CLASS_NAME, 0.05)
    assert [p[3] for p in params] == sorted(p[3] for p in params) and np.allclose([p[3] for p in params], order)
    assert all(p[1] > 0.05 for p in params)
    got = ow.instance_masks(params)
    assert len(got) == len(masks)
    for (m, k), want, wk in zip(got, masks, keep):
        assert k == wk and np.array_equal(m, want)
    l2, files = ow.format_image(det, "frankfurt_%s_leftImg8bit.png" % name, CLASS_NAME, LABEL_TO_ID, 0.05)
    assert l2 == lines and len(files) == int(keep.sum())
    # an instance with score >= 0.5 hides every farther one where they overlap
    removed = np.zeros(masks.shape[1:], bool)
    for (pts, score, label, depth), m in zip(params, masks):
        assert not (removed & (m > 0)).any()
        if score >= 0.5:
            removed |= m > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_masks_equal_pil_fixtures(name, tmp_path):
    import types

    import torch
    from centerpoly_amd.datasets.dataset.polygons import CITYSCAPES
    det, masks, keep, lines, order = _load(name)
    ds = CITYSCAPES.__new__(CITYSCAPES)                       # the writer needs opt.thresh and the image names only
    ds.opt = types.SimpleNamespace(thresh=0.05)
    ds.coco = types.SimpleNamespace(imgs={7: {"id": 7, "file_name": "/data/frankfurt_%s_leftImg8bit.png" % name}})
    params = ds.image_instances(det)
    ref = ow.image_instances(det, CLASS_NAME, 0.05)
    assert [(p[0], float(p[1]), p[2], float(p[3])) for p in params] == \
        [(p[0], float(p[1]), p[2], float(p[3])) for p in ref]
    got, counts = ds.instance_masks(params, torch.device("cuda"))
    assert got.shape == masks.shape
    for i in range(len(masks)):
        assert np.array_equal(got[i], masks[i]), "instance %d differs in %d pixels" % (i, (got[i] != masks[i]).sum())
        assert counts[i] == np.count_nonzero(masks[i])
    ds.format_and_write_to_cityscapes({7: det}, str(tmp_path))
    txt = open(os.path.join(str(tmp_path), "frankfurt_%s_leftImg8bit.txt" % name)).read()
    assert txt == "".join(lines)
    from PIL import Image
    _, files = ow.format_image(det, "frankfurt_%s_leftImg8bit.png" % name, CLASS_NAME, LABEL_TO_ID, 0.05)
    for fn, arr in files.items():
        assert np.array_equal(np.array(Image.open(os.path.join(str(tmp_path), "masks", fn))), arr)
