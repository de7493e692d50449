"""End-to-end PolydetDetector.run timing on one MI355X (pre / net / dec / post / merge split)."""
import contextlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(ROOT, ".miopen_cache"))
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, ".miopen_cache"))

from centerpoly_amd import synth  # noqa: E402
from centerpoly_amd.detectors.detector_factory import detector_factory  # noqa: E402
from centerpoly_amd.opts import opts  # noqa: E402


def main():
    h, w = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    with contextlib.redirect_stdout(sys.stderr):
        opt = opts().init(["polydet", "--arch", "dla_34", "--input_h", str(h), "--input_w", str(w)])
        det = detector_factory["polydet"](opt)
    img = (synth.uniform("probe/img", (h, w, 3)) * 255).astype(np.uint8)
    keys = ["tot", "load", "pre", "net", "dec", "post", "merge"]
    for it in range(40):
        ret = det.run(img)
        print(it, " ".join("%s %.2fms" % (k, 1e3 * ret[k]) for k in keys), flush=True)


if __name__ == "__main__":
    main()
