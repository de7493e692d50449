"""Launch one DCNv2 forward layer (default 64->64 @256x512, the shape bench.py reports) a few times so that
rocprofv3 --pmc / --kernel-trace can read its counters.  PMC_SHAPE=B,Cin,Cout,H,W, PMC_OFF_STD (px, default 0.3),
PMC_CONTRACTION (f32 | bf16x3 | bf16x3_region; default bf16x3 = what inference runs), weights prepared once."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import synth
from centerpoly_amd.models.networks.DCNv2.dcn_v2 import dcn_v2_forward_raw
dev = "cuda"
B, ci, co, H, W = [int(v) for v in os.environ.get("PMC_SHAPE", "1,64,64,256,512").split(",")]
x = torch.from_numpy(synth.normal("pmc/x", (B, ci, H, W))).to(dev)
om = torch.from_numpy(synth.normal("pmc/om", (B, 27, H, W), 0.0, float(os.environ.get("PMC_OFF_STD", "0.3")))).to(dev)
w = torch.from_numpy(synth.normal("pmc/w", (co, ci, 3, 3), 0, 0.04)).to(dev)
b = torch.zeros(co, device=dev)


class Owner:
    pass


own = Owner()
for _ in range(int(os.environ.get("PMC_LAUNCHES", "12"))):
    dcn_v2_forward_raw(x, om, w, b, contraction=os.environ.get("PMC_CONTRACTION", "bf16x3"), owner=own)
torch.cuda.synchronize()
