"""Launch the DLA base layer (7x7, 3 -> 16, stride 1, 1 x 3 x 1024 x 2048) a few times for rocprofv3 --pmc passes
(PMC_SCRIPT=tools/pmc_stem.py tools/run_pmc_fwd.sh <tag>)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import _C
L = _C.lib(); dev = "cuda"
torch.manual_seed(0)
img = torch.randn(1, 3, 1024, 2048, device=dev); w = torch.randn(16, 3, 7, 7, device=dev) * 0.05; b = torch.randn(16, device=dev)
wp = torch.empty(L.cp_conv7x7_c3_weight_bytes(16), dtype=torch.uint8, device=dev)
assert L.cp_conv7x7_c3_prepare(_C.ptr(w), 16, _C.ptr(wp), _C.stream()) == 0
o = torch.empty(1, 16, 1024, 2048, device=dev)
for _ in range(12):
    L.cp_conv7x7_c3_forward(_C.ptr(img), _C.ptr(wp), _C.ptr(b), _C.ptr(o), 1, 1024, 2048, 16, 1, 1, _C.stream())
torch.cuda.synchronize()
