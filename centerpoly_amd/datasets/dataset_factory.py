"""get_dataset (reference: src/lib/datasets/dataset_factory.py:50-53).  The real datasets
(Cityscapes / KITTI-poly / IDD images and annotation JSONs) are not available offline, so the only
registered dataset is a synthetic one with the reference sampler's batch schema
(src/lib/datasets/sample/polydet.py:425-449) and the Cityscapes constants
(src/lib/datasets/dataset/cityscapes.py:41-49,87)."""
from .synthetic import SyntheticPolydet

dataset_factory = {"cityscapes": SyntheticPolydet, "synthetic": SyntheticPolydet}


def get_dataset(dataset, task):
    if task != "polydet":
        raise KeyError("only the polydet task is on the accelerated path")
    if dataset not in dataset_factory:
        raise KeyError("dataset %r is not available offline (have: %s)" % (dataset, sorted(dataset_factory)))
    return dataset_factory[dataset]
