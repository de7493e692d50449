"""Fixtures of the Cityscapes result writer: seeded detections -> instance masks and text lines produced
by oracle/writer.py, i.e. by the SAME PIL calls the reference makes (ImageDraw.polygon / ellipse of the
installed PIL 12.2) on the reference's 2048x1024 canvas.  Masks are stored bit-packed.

    python tests/golden/gen_writer_golden.py        # rewrites tests/golden/writer_*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import writer as ow  # noqa: E402

CLASS_NAME = ["__background__", "person", "rider", "car", "truck", "bus", "train", "motorcycle", "bicycle", "pole",
              "traffic sign", "traffic light"]
LABEL_TO_ID = {"person": 24, "rider": 25, "car": 26, "truck": 27, "bus": 28, "train": 31, "motorcycle": 32,
               "bicycle": 33, "pole": -1, "traffic sign": -1, "traffic light": -1}


def detections(seed, n_pts, n_obj, kind):
    """{cls: [[x1,y1,x2,y2,score, poly(2N), depth], ...]} in the layout PolydetDetector.run returns."""
    rng = np.random.RandomState(seed)
    out = {}
    for k in range(n_obj):
        cls = int(rng.randint(1, 12 if kind == "mixed" else 9))
        if kind == "small":
            c = rng.uniform(40, 1900, 2) * [1, 0.5]
            r = rng.uniform(2, 14, n_pts)
        else:
            c = np.array([rng.uniform(-40, 2090), rng.uniform(-30, 1060)])
            r = rng.uniform(15, 260, n_pts) if kind != "selfcross" else rng.uniform(5, 300, n_pts)
        th = np.sort(rng.uniform(0, 2 * np.pi, n_pts)) if kind != "selfcross" else rng.uniform(0, 2 * np.pi, n_pts)
        pts = np.stack([c[0] + r * np.cos(th), c[1] + r * np.sin(th)], 1).astype(np.float32)
        score = float(np.float32(rng.choice([0.03, 0.2, 0.45, 0.5, 0.62, 0.9])))
        depth = float(np.float32(rng.uniform(0, 50)))
        row = np.concatenate([[pts[:, 0].min(), pts[:, 1].min(), pts[:, 0].max(), pts[:, 1].max(), score],
                              pts.reshape(-1), [depth]]).astype(np.float32)
        out.setdefault(cls, []).append(row)
    return {c: np.stack(v) for c, v in out.items()}


CASES = [("star16", 1, 16, 14, "star"), ("mixed32", 2, 32, 18, "mixed"), ("selfcross16", 3, 16, 10, "selfcross"),
         ("small16", 4, 16, 12, "small")]

if __name__ == "__main__":
    for name, seed, n_pts, n_obj, kind in CASES:
        det = detections(seed, n_pts, n_obj, kind)
        params = ow.image_instances(det, CLASS_NAME, 0.05)
        masks = ow.instance_masks(params)
        lines, files = ow.format_image(det, "frankfurt_%s_leftImg8bit.png" % name, CLASS_NAME, LABEL_TO_ID, 0.05)
        arr = np.stack([m for m, _ in masks]) if masks else np.zeros((0, 1024, 2048), np.uint8)
        np.savez_compressed(os.path.join(HERE, "writer_%s.npz" % name),
                            packed=np.packbits(arr > 0, axis=2), keep=np.array([k for _, k in masks], bool),
                            lines=np.array(lines), order_depth=np.array([p[3] for p in params], np.float64),
                            **{"det_%d" % c: v for c, v in det.items()})
        print(name, "instances", len(params), "kept", int(sum(k for _, k in masks)), "files", len(files))
