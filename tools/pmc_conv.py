"""Launch one MFMA 3x3 convolution layer (default 64->64 @256x512 x4, the training network's most frequent shape) a few
times so that rocprofv3 --pmc / --kernel-trace can read its counters.  PMC_SHAPE=B,Cin,Cout,H,W."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import _C, synth
L = _C.lib()
P = _C.ptr
dev = "cuda"
B, ci, co, H, W = [int(v) for v in os.environ.get("PMC_SHAPE", "4,64,64,256,512").split(",")]
x = torch.from_numpy(synth.normal("pmc/x", (B, ci, H, W))).to(dev)
w = torch.from_numpy(synth.normal("pmc/w", (co, ci, 3, 3), 0, 0.04)).to(dev)
out = torch.empty(B, co, H, W, device=dev)
wp = torch.empty(L.cp_conv3x3_mfma_weight_bytes(ci, co), dtype=torch.uint8, device=dev)
_C.check(L.cp_conv3x3_mfma_prepare(P(w), ci, co, 0, P(wp), _C.stream()), "prepare")
for _ in range(int(os.environ.get("PMC_LAUNCHES", "12"))):
    _C.check(L.cp_conv3x3_mfma_forward(P(x), P(wp), None, None, P(out), B, ci, H, W, co, 0, _C.stream()), "conv")
torch.cuda.synchronize()
