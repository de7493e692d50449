"""get_dataset (reference: src/lib/datasets/dataset_factory.py:24-53): `dataset_factory` maps the names
the reference registers for polygon detection to dataset classes, `get_dataset(dataset, task)` mixes
the polydet sampler in.  `cityscapes`, `kitti_poly` and `IDD` read the reference's annotation JSON and
image files (a missing file is an error, never a silent substitute); `synthetic` is the offline set of
hash-generated items with the same batch schema, which the benches and tests use."""
from .dataset.polygons import CITYSCAPES, IDD, KITTIPOLY
from .sample.polydet import PolydetDataset
from .synthetic import SyntheticPolydet

dataset_factory = {"cityscapes": CITYSCAPES, "kitti_poly": KITTIPOLY, "IDD": IDD, "synthetic": SyntheticPolydet}
_sample_factory = {"polydet": PolydetDataset}


def get_dataset(dataset, task):
    if task not in _sample_factory:
        raise KeyError("only the polydet task is on the accelerated path (got %r)" % (task,))
    if dataset not in dataset_factory:
        raise KeyError("dataset %r is not registered (have: %s)" % (dataset, sorted(dataset_factory)))
    if dataset == "synthetic":
        return SyntheticPolydet

    class Dataset(dataset_factory[dataset], _sample_factory[task]):
        pass
    return Dataset
