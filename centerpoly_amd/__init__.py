"""centerpoly_amd -- MI355X-native hot path of CenterPoly v2 (`polydet`).

Sub-packages mirror the reference's `src/lib` namespace (models, trains,
detectors, utils, opts) so that the reference's drivers can import them under
the same names; every dense op behind them is a hand-written HIP kernel reached
through the C ABI in include/centerpoly_hip.h (see _C.py).  No CPU fallback.
"""
import importlib
import sys

__version__ = "0.1.0"

_MIRRORED = ("models", "trains", "detectors", "utils", "external", "opts", "logger", "datasets")


def install_as_reference_lib():
    """Expose this package under the top-level names the reference's main.py /
    test.py import (`models.model`, `trains.train_factory`, `detectors...`,
    `opts`), i.e. what src/_init_paths.py:8-12 does for src/lib."""
    for name in _MIRRORED:
        mod = importlib.import_module("centerpoly_amd." + name)
        sys.modules.setdefault(name, mod)
    for sub in ("models.model", "models.decode", "models.losses", "models.utils",
                "models.networks.pose_dla_dcn", "models.networks.large_hourglass",
                "models.networks.DCNv2.dcn_v2", "trains.train_factory", "trains.polydet",
                "trains.base_trainer", "detectors.detector_factory", "detectors.polydet",
                "detectors.base_detector", "utils.image", "utils.post_process", "utils.utils", "external.nms",
                "models.data_parallel", "datasets.dataset_factory", "datasets.sample.polydet"):
        sys.modules.setdefault(sub, importlib.import_module("centerpoly_amd." + sub))
