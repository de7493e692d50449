"""Launch the dominant DCNv2 forward layer (64->64 @256x512, the shape bench.py reports) a few
times so rocprofv3 --pmc can read its HBM traffic.  Offsets have the magnitude the bench model
produces (about one pixel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import synth
from centerpoly_amd.models.networks.DCNv2.dcn_v2 import dcn_v2_forward_raw
dev = "cuda"
ci, co, H, W = 64, 64, 256, 512
x = torch.from_numpy(synth.normal("pmc/x", (1, ci, H, W))).to(dev)
om = torch.from_numpy(synth.normal("pmc/om", (1, 27, H, W), 0.0, float(os.environ.get("PMC_OFF_STD", "1.0")))).to(dev)
w = torch.from_numpy(synth.normal("pmc/w", (co, ci, 3, 3), 0, 0.04)).to(dev)
b = torch.zeros(co, device=dev)
for _ in range(12):
    dcn_v2_forward_raw(x, om, w, b)
torch.cuda.synchronize()
