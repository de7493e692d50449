"""PolydetDetector (reference: src/lib/detectors/polydet.py:21-100)."""
import time

import numpy as np
import torch

from ..external.nms import soft_nms
from ..models.decode import polydet_decode
from ..models.utils import flip_tensor
from ..utils.post_process import polydet_post_process_device
from .base_detector import BaseDetector


class PolydetDetector(BaseDetector):
    def __init__(self, opt):
        super(PolydetDetector, self).__init__(opt)

    def process(self, images, return_time=False):
        with torch.no_grad():
            output = self.model(images)[-1]
            hm = output["hm"].sigmoid_()             # inference: no clamp (reference :28)
            polys = output["poly"]
            pseudo_depth = output["pseudo_depth"]
            reg = output["reg"] if self.opt.reg_offset else None
            if self.opt.flip_test:
                hm = (hm[0:1] + flip_tensor(hm[1:2])) / 2
                reg = reg[0:1] if reg is not None else None
                polys, pseudo_depth = polys[0:1], pseudo_depth[0:1]
            torch.cuda.synchronize()
            forward_time = time.time()
            dets = polydet_decode(hm, polys, pseudo_depth, reg=reg,
                                  cat_spec_poly=self.opt.cat_spec_poly, K=self.opt.K,
                                  rep=self.opt.rep)
        if return_time:
            return output, dets, forward_time
        return output, dets

    def post_process(self, dets, meta, scale=1, fg=None):
        # transform_preds + `/ scale` on the device, one copy back, class split on the host
        dets = dets.detach().reshape(1, -1, dets.shape[2])
        return polydet_post_process_device(dets, [meta["c"]], [meta["s"]], meta["out_height"],
                                           meta["out_width"], self.opt.num_classes, scale)[0]

    def merge_outputs(self, detections):
        results = {}
        for j in range(1, self.num_classes + 1):
            results[j] = np.concatenate([d[j] for d in detections], axis=0).astype(np.float32)
            if len(self.scales) > 1 or self.opt.nms:
                soft_nms(results[j], Nt=0.5, method=2)
        scores = np.hstack([results[j][:, 4] for j in range(1, self.num_classes + 1)])
        if len(scores) > self.max_per_image:
            kth = len(scores) - self.max_per_image
            thresh = np.partition(scores, kth)[kth]
            for j in range(1, self.num_classes + 1):
                results[j] = results[j][results[j][:, 4] >= thresh]
        return results
