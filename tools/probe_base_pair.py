"""GPU probe: level0 + level1 of the DLA base at 1 x 16 x 1024 x 2048 as the two direct kernels (level0 split-bf16,
level1 exact) against the one-launch form cp_dla_base_pair_forward; us per image and the difference of the results."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import _C
L = _C.lib(); dev = "cuda"
torch.manual_seed(0)


def timed(call, n=30):
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


H, W = 1024, 2048
x = torch.randn(1, 16, H, W, device=dev)
w0 = torch.randn(16, 16, 3, 3, device=dev) * 0.08; b0 = torch.randn(16, device=dev) * 0.1
w1 = torch.randn(32, 16, 3, 3, device=dev) * 0.08; b1 = torch.randn(32, device=dev) * 0.1
y0 = torch.empty(1, 16, H, W, device=dev); y1 = torch.empty(1, 32, H // 2, W // 2, device=dev); o = torch.empty_like(y1)


def separate():
    L.cp_conv_direct_forward_ex(_C.ptr(x), _C.ptr(w0), _C.ptr(b0), _C.ptr(y0), 1, 16, H, W, 16, 3, 1, 1, 1, 1, _C.stream())
    L.cp_conv_direct_forward_ex(_C.ptr(y0), _C.ptr(w1), _C.ptr(b1), _C.ptr(y1), 1, 16, H, W, 32, 3, 2, 1, 1, 1, _C.stream())


def fused():
    rc = L.cp_dla_base_pair_forward(_C.ptr(x), _C.ptr(w0), _C.ptr(b0), _C.ptr(w1), _C.ptr(b1), _C.ptr(o), 1, H, W, _C.stream())
    assert rc == 0, rc


for rep in range(2):
    print("two launches %.1f us   one launch %.1f us" % (timed(separate), timed(fused)), flush=True)
print("max difference %.2e of the max-norm" % ((o - y1).abs().max().item() / y1.abs().max().item()))
