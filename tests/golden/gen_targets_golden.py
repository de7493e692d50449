#!/usr/bin/env python3
"""Generate tests/golden/targets_prims.npz by running the REFERENCE's utils/image.py helpers
(gaussian_radius, draw_umich_gaussian, affine_transform) on seeded inputs.  Build container
only (needs /root/reference); the fixture holds inputs' seeds and expected outputs, no source.

Usage:  python tests/golden/gen_targets_golden.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src/lib")
sys.modules["cv2"] = types.ModuleType("cv2")          # imported, unused by these helpers

from centerpoly_amd import synth                       # noqa: E402
from utils.image import affine_transform, draw_umich_gaussian, gaussian_radius   # noqa: E402

SIZES = [(1, 1), (2, 3), (5, 40), (17, 17), (33, 120), (200, 75), (256, 512)]


def main():
    out = {}
    out["radius_sizes"] = np.array(SIZES, dtype=np.int64)
    out["radius"] = np.array([gaussian_radius(s) for s in SIZES], dtype=np.float64)
    # Gaussian splats, max-composited, clipped at every border
    hm = np.zeros((48, 64), dtype=np.float32)
    centers = synth.integers("targets/centers", (12, 2), 0, 48)
    radii = synth.integers("targets/radii", (12,), 0, 9)
    centers[0] = (0, 0)
    centers[1] = (63, 47)
    for (cx, cy), r in zip(centers, radii):
        draw_umich_gaussian(hm, (int(min(cx + 8, 63)), int(cy)), int(r))
    out["splat_hm"] = hm
    pts = synth.uniform("targets/pts", (32, 2), -50.0, 2100.0)
    t = np.array([[0.2461, 0.0013, -3.25], [-0.0009, 0.2502, 7.5]], dtype=np.float64)
    out["affine_t"] = t
    out["affine"] = np.stack([affine_transform(p, t) for p in pts]).astype(np.float64)
    np.savez_compressed(os.path.join(HERE, "targets_prims.npz"), **out)
    print("wrote targets_prims.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
