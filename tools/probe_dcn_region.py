"""GPU probe: the LDS-region DCNv2 forward kernel (contraction "bf16x3_region") against the gather kernels -- error
against the exact-f32 kernel and launch time on the offset fields of tools/probe_dcn_fwd.py plus a 0.1-px field (the
bench model's regime).  HIP events, weights prepared once (the inference path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from centerpoly_amd import synth
from centerpoly_amd.models.networks.DCNv2.dcn_v2 import dcn_v2_forward_raw

dev = "cuda"


class Owner:
    pass


def fields(H, W):
    noise = synth.normal("probe/om", (1, 27, H, W))
    k = np.ones((1, 1, 9, 9), np.float32) / 81.0
    sm = torch.nn.functional.conv2d(torch.from_numpy(noise[:, :18]).reshape(18, 1, H, W), torch.from_numpy(k), padding=4)
    smooth = noise.copy()
    smooth[:, :18] = (sm.reshape(1, 18, H, W) * 27.0).numpy()
    out = {}
    for name, sc in (("noise std 0.1", 0.1), ("noise std 0.3", 0.3), ("noise std 1", 1.0), ("noise std 2", 2.0)):
        f = noise.copy()
        f[:, :18] *= sc
        out[name] = f
    out["smooth std 3"] = smooth
    return out


def timed(fn, n=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [(1, 64, 64, 256, 512), (1, 128, 128, 128, 256), (1, 128, 64, 128, 256), (1, 256, 256, 64, 128),
          (4, 64, 64, 256, 512)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for (B, ci, co, H, W) in shapes:
    x = torch.from_numpy(synth.normal("probe/x", (B, ci, H, W))).to(dev)
    w = torch.from_numpy(synth.normal("probe/w", (co, ci, 3, 3), 0, 0.04)).to(dev)
    b = torch.from_numpy(synth.normal("probe/b", (co,))).to(dev)
    for name, om in fields(H, W).items():
        omt = torch.from_numpy(np.repeat(om, B, axis=0)).to(dev)
        ref = dcn_v2_forward_raw(x, omt, w, b, contraction="f32")
        o1, o3 = Owner(), Owner()
        reg = dcn_v2_forward_raw(x, omt, w, b, contraction="bf16x3_region", owner=o3)
        err = float((reg - ref).abs().max() / ref.abs().max())
        t_f32 = timed(lambda: dcn_v2_forward_raw(x, omt, w, b, contraction="f32"))
        t_bf = timed(lambda: dcn_v2_forward_raw(x, omt, w, b, contraction="bf16x3", owner=o1))
        t_rg = timed(lambda: dcn_v2_forward_raw(x, omt, w, b, contraction="bf16x3_region", owner=o3))
        print("B%d %d->%d @%dx%d %-14s err %.2e | f32 %.1f us  bf16x3(auto) %.1f us  region %.1f us" %
              (B, ci, co, H, W, name, err, t_f32, t_bf, t_rg), flush=True)
