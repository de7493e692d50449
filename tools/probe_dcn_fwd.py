"""GPU probe: DCNv2 forward launch time of the dominant layer shapes for three offset fields --
white noise (std 1 px), small noise (std 0.3 px) and a smooth field of a few pixels (what a
trained offset branch produces).  Production library, HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from centerpoly_amd import synth
from centerpoly_amd.models.networks.DCNv2.dcn_v2 import dcn_v2_forward_raw

dev = "cuda"


def fields(H, W):
    noise = synth.normal("probe/om", (1, 27, H, W))
    smooth = noise.copy()
    for _ in range(6):
        smooth[:, :18] = synth.smooth_field("probe/sm", (1, 18, H, W)) if _ == 0 else smooth[:, :18]
    k = np.ones((1, 1, 9, 9), np.float32) / 81.0
    sm = torch.nn.functional.conv2d(torch.from_numpy(noise[:, :18]).reshape(18, 1, H, W), torch.from_numpy(k), padding=4)
    smooth[:, :18] = (sm.reshape(1, 18, H, W) * 27.0).numpy()        # std ~3 px, correlated over ~9 px
    small = noise.copy()
    small[:, :18] *= 0.3
    return {"noise std 1": noise, "noise std 0.3": small, "smooth std 3": smooth}


def run(ci, co, H, W, om, n=50):
    x = torch.from_numpy(synth.normal("probe/x", (1, ci, H, W))).to(dev)
    omt = torch.from_numpy(om).to(dev)
    w = torch.from_numpy(synth.normal("probe/w", (co, ci, 3, 3), 0, 0.04)).to(dev)
    b = torch.zeros(co, device=dev)
    for _ in range(20):
        dcn_v2_forward_raw(x, omt, w, b)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        dcn_v2_forward_raw(x, omt, w, b)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for shape in [(64, 64, 256, 512), (128, 128, 128, 256), (256, 256, 64, 128)]:
    for name, om in fields(*shape[2:]).items():
        print("%s %-14s %.1f us" % (shape, name, run(*shape, om)), flush=True)
