#!/usr/bin/env python3
"""Evaluation driver with the reference's CLI and control flow (src/test.py:47-131):
`python test.py polydet --arch dla_34 --load_model model_last.pth`.  Images come from the
synthetic dataset (uint8 arrays through PolydetDetector.run, pre-process included)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

from centerpoly_amd import synth
from centerpoly_amd.datasets.dataset_factory import get_dataset
from centerpoly_amd.detectors.detector_factory import detector_factory
from centerpoly_amd.opts import opts
from centerpoly_amd.utils.utils import AverageMeter


def test(opt):
    Dataset = get_dataset(opt.dataset, opt.task)
    opt = opts().update_dataset_info_and_set_heads(opt, Dataset)
    dataset = Dataset(opt, "val")
    detector = detector_factory[opt.task](opt)
    results = {}
    time_stats = ["tot", "load", "pre", "net", "dec", "post", "merge"]
    avg = {t: AverageMeter() for t in time_stats}
    for ind in range(len(dataset)):
        img = (synth.uniform("test/img%d" % ind, (opt.input_h, opt.input_w, 3)) * 255).astype(np.uint8)
        ret = detector.run(img)
        results[ind] = ret["results"]
        for t in avg:
            avg[t].update(ret[t])
        print("[{}/{}] ".format(ind, len(dataset)) + " ".join("|{} {:.3f}s".format(t, avg[t].avg) for t in avg))
    dataset.run_eval(results, opt.save_dir)


if __name__ == "__main__":
    test(opts().parse())
