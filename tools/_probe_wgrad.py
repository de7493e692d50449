import os, sys
sys.path.insert(0, "/root/repo")
import torch
from centerpoly_amd import _C
if len(sys.argv) > 1:
    _C.LIB_PATH = sys.argv[1]
L = _C.lib(); dev = "cuda"
def timed(call, n=20):
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, ci, co, H, W) in [(4, 64, 64, 256, 512), (4, 128, 128, 128, 256), (4, 256, 256, 64, 128), (4, 512, 512, 32, 64)]:
    x = torch.randn(B, ci, H, W, device=dev); go = torch.randn(B, co, H, W, device=dev)
    gw = torch.zeros(co, ci, 3, 3, device=dev)
    t = timed(lambda: L.cp_conv3x3_mfma_wgrad(_C.ptr(x), _C.ptr(go), _C.ptr(gw), B, ci, H, W, co, _C.stream()))
    print("%dx%d->%d @%dx%d wgrad %.1f us" % (B, ci, co, H, W, t), flush=True)
