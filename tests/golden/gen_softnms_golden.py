"""Fixtures of the reference's own soft-NMS (src/lib/external/nms.pyx:77-170, Cython): seeded box tables -> the table
after the in-place call and the returned keep list, produced by the module oracle/build_ref_nms.py compiles from the
reference's file (its `soft_nms` text unmodified; see that recipe for the two tokens patched in the unrelated `nms()`).

    python tests/golden/gen_softnms_golden.py       # builds oracle/_ref/refnms*.so if needed, rewrites softnms_ref.npz

Cases cover the three methods (0 hard, 1 linear, 2 gaussian), the detector's call (Nt=0.5, method=2:
src/lib/detectors/polydet.py:66-67), dense and sparse tables, rows discarded below the threshold, ties in the score,
a single row, identical boxes, and the 2N+6-column rows of the polydet results (columns >= 5 never move)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import build_ref_nms  # noqa: E402


def boxes(seed, n, ncols, spread, quant=None, dup=False):
    rng = np.random.RandomState(seed)
    c = rng.uniform(0.0, spread, (n, 2))
    wh = rng.uniform(5.0, 80.0, (n, 2))
    b = rng.uniform(0.0, 300.0, (n, ncols)).astype(np.float32)
    b[:, 0:2] = c - wh / 2
    b[:, 2:4] = c + wh / 2
    b[:, 4] = rng.uniform(0.0, 1.0, n)
    if quant:                                   # score ties
        b[:, 4] = np.round(b[:, 4] * quant) / quant
    if dup and n >= 4:                          # identical boxes with different scores
        b[1, :4] = b[0, :4]
        b[3, :4] = b[2, :4]
    return np.ascontiguousarray(b.astype(np.float32))


CASES = []
for method in (0, 1, 2):
    CASES += [
        dict(seed=11 + method, n=1, ncols=38, spread=50.0, sigma=0.5, Nt=0.5, threshold=0.001, method=method),
        dict(seed=21 + method, n=40, ncols=38, spread=60.0, sigma=0.5, Nt=0.5, threshold=0.001, method=method),
        dict(seed=31 + method, n=256, ncols=8, spread=400.0, sigma=0.5, Nt=0.5, threshold=0.001, method=method),
        dict(seed=41 + method, n=300, ncols=8, spread=40.0, sigma=0.5, Nt=0.5, threshold=0.2, method=method),
        dict(seed=51 + method, n=64, ncols=5, spread=30.0, sigma=0.3, Nt=0.3, threshold=0.05, method=method, quant=8),
        dict(seed=61 + method, n=24, ncols=70, spread=20.0, sigma=0.7, Nt=0.4, threshold=0.01, method=method, dup=True),
    ]


def main():
    ref = build_ref_nms.load()
    out = {"n_cases": np.int64(len(CASES))}
    for i, c in enumerate(CASES):
        b = boxes(c["seed"], c["n"], c["ncols"], c["spread"], c.get("quant"), c.get("dup", False))
        out["c%d_in" % i] = b.copy()
        keep = ref.soft_nms(b, sigma=c["sigma"], Nt=c["Nt"], threshold=c["threshold"], method=c["method"])
        out["c%d_out" % i] = b
        out["c%d_keep" % i] = np.asarray(keep, dtype=np.int64)
        out["c%d_par" % i] = np.array([c["sigma"], c["Nt"], c["threshold"], c["method"]], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "softnms_ref.npz"), **out)
    print("wrote softnms_ref.npz: %d cases" % len(CASES))


if __name__ == "__main__":
    main()
