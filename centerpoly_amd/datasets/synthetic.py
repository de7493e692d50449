"""Synthetic Cityscapes-shaped polydet dataset: items are regenerated from a counter-based hash
(centerpoly_amd.synth), with the keys PolydetLoss and save_result consume."""
import numpy as np
import torch.utils.data as data

from .. import synth


class SyntheticPolydet(data.Dataset):
    num_classes = 8                                   # cityscapes.py:41-43
    default_resolution = [512, 1024]                  # cityscapes.py:45
    mean = np.array([0.284, 0.323, 0.282], dtype=np.float32).reshape(1, 1, 3)
    std = np.array([0.04, 0.04, 0.04], dtype=np.float32).reshape(1, 1, 3)
    max_objs = 128                                    # cityscapes.py:87

    def __init__(self, opt, split):
        self.opt = opt
        self.split = split
        self.num_samples = getattr(opt, "synthetic_samples", 64 if split == "train" else 8)
        print("Loaded synthetic %s split: %d samples" % (split, self.num_samples))

    def __len__(self):
        return self.num_samples

    def __getitem__(self, index):
        opt = self.opt
        h, w = opt.input_h // opt.down_ratio, opt.input_w // opt.down_ratio
        b = synth.train_batch(1, h, w, nbr_points=opt.nbr_points, num_classes=self.num_classes,
                              max_objs=self.max_objs, rep=opt.rep,
                              stream="%s/%d" % (self.split, index), in_h=opt.input_h, in_w=opt.input_w)
        item = {k: v[0] for k, v in b.items()}
        item["freq_mask"] = np.float32(1.0)
        if self.split != "train":
            item["meta"] = {"c": np.array([opt.input_w / 2.0, opt.input_h / 2.0], dtype=np.float32),
                            "s": np.float32(max(opt.input_h, opt.input_w)), "img_id": index,
                            "out_height": h, "out_width": w}
        return item

    def run_eval(self, results, save_dir):
        """No ground-truth images offline: report the detection count instead of Cityscapes AP
        (reference: cityscapes.py:400-432 runs the vendored evaluator on the real dataset)."""
        n = sum(len(v) for r in results.values() for v in r.values())
        print("synthetic eval: %d images, %d detections (AP needs the real dataset)" % (len(results), n))
        return 0.0
