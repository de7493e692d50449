"""GPU probe: launch times of the detector-IO and target-construction kernels (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from centerpoly_amd import synth
from centerpoly_amd.datasets.sample.polydet import build_targets, collate, pack_annotations
from centerpoly_amd.utils.image import get_affine_transform, warp_affine_normalize
from centerpoly_amd.utils.post_process import polydet_post_process_device


def bench(fn, n=50, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


img = torch.from_numpy((synth.uniform("io/img", (1024, 2048, 3)) * 255).astype(np.uint8)).cuda()
mean, std = np.array([0.284, 0.323, 0.282], np.float32), np.array([0.04, 0.04, 0.04], np.float32)
c, s = np.array([1024., 512.], np.float32), np.array([2080., 1056.], np.float32)
t = get_affine_transform(c, s, 0, [2080, 1056])
print("preprocess 2048x1024 u8 -> 3x1056x2080 fp32: %.1f us" % bench(lambda: warp_affine_normalize(img, t, mean, std, 1056, 2080)))
print("  + flipped copy: %.1f us" % bench(lambda: warp_affine_normalize(img, t, mean, std, 1056, 2080, True)))
packed = []
for b in range(4):
    anns = synth.raw_annotations("io/t%d" % b, 1024, 2048, n_objs=30)
    packed.append(pack_annotations(anns, get_affine_transform(np.array([1000., 500.], np.float32), 2048.0, 0, [512, 256]),
                                   b % 2, 2048, 128, 16))
raw = {k: v.cuda() for k, v in collate(packed).items()}
print("targets B=4, 30 objects/img, 8x256x512 maps (memset + 2 kernels + output allocs): %.1f us" % bench(lambda: build_targets(raw, 256, 512, 8)))
print("  without border_hm: %.1f us" % bench(lambda: build_targets(raw, 256, 512, 8, with_border_hm=False)))
dets = torch.rand(1, 128, 39, device="cuda") * 500
print("post-process K=128 (kernel + D2H + class split): %.1f us" % bench(lambda: polydet_post_process_device(dets, [c], [s], 264, 520, 8)))
