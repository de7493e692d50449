"""GPU probe: the three decode launches (NMS + per-tile top-K, segment merges, final rank + gather) on heat maps of
different statistics -- run under `rocprofv3 --kernel-trace --stats` to read the per-kernel times.
  smooth   bench.py's heat (sigmoid of the DLA-34 model's output would be similar: few positive maxima per tile)
  noise    sigmoid of white noise: every tile holds more than K local maxima (no zero-valued key survives a merge)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from centerpoly_amd import synth
from centerpoly_amd.models.decode import polydet_decode

dev = "cuda"
B, C, H, W, N = 1, 8, 256, 512, 16
polys = torch.from_numpy(synth.normal("pd/poly", (B, 2 * N, H, W))).to(dev)
depth = torch.from_numpy(synth.uniform("pd/depth", (B, 1, H, W))).to(dev)
reg = torch.from_numpy(synth.uniform("pd/reg", (B, 2, H, W))).to(dev)
for name in sys.argv[1:] or ["smooth", "noise"]:
    if name == "smooth":
        heat = torch.sigmoid(torch.from_numpy(synth.heat_logits("pd/hm", B, C, H, W))).to(dev)
    else:
        heat = torch.sigmoid(torch.from_numpy(synth.normal("pd/hm2", (B, C, H, W)))).to(dev)
    for _ in range(5):
        polydet_decode(heat, polys, depth, reg=reg, K=128)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        polydet_decode(heat, polys, depth, reg=reg, K=128)
    e1.record()
    torch.cuda.synchronize()
    print("%s: %.1f us per decode (3 launches, HIP events, back to back)" % (name, e0.elapsed_time(e1) / 50 * 1e3), flush=True)
