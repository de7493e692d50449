"""Launch the DCNv2 backward kernels of one layer shape a few times so that rocprofv3 --pmc /
--kernel-trace can read their counters (default: the launch that dominates the B=4 training
step, 64->64 @256x512 x4 images).  Inputs: unit-normal x / grad_out, conv_offset_mask output
with PMC_OFF_STD px offsets (default 0.5).

  rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d <dir> -- python3 tools/pmc_dcn_bwd.py
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import _C, synth

B, ci, co, H, W = [int(v) for v in os.environ.get("PMC_SHAPE", "4,64,64,256,512").split(",")]
what = os.environ.get("PMC_WHAT", "both")
n = int(os.environ.get("PMC_LAUNCHES", "6"))
dev = "cuda"
L = _C.lib()
x = torch.from_numpy(synth.normal("pmcb/x", (B, ci, H, W))).to(dev)
om = torch.from_numpy(synth.normal("pmcb/om", (B, 27, H, W)) * float(os.environ.get("PMC_OFF_STD", "0.5"))).to(dev)
w = torch.from_numpy(synth.normal("pmcb/w", (co, ci, 3, 3), 0, 0.05)).to(dev)
go = torch.from_numpy(synth.normal("pmcb/go", (B, co, H, W))).to(dev)
gx = torch.zeros_like(x); gom = torch.empty_like(om); gw = torch.zeros_like(w)
s = _C.DcnShape(B, ci, H, W, co, 3, 3, 1, 1, 1, 1)
bs = 27 * H * W; off_m = 72 * H * W
P = lambda t: ctypes.c_void_p(t.data_ptr())
ws_bytes = L.cp_dcn_v2_backward_workspace_bytes(s)
ws = _C.workspace(ws_bytes, dev)
for i in range(n):
    for data in ((True, False) if what == "both" else ((what == "data"),)):
        rc = L.cp_dcn_v2_backward(s, P(x), P(om), bs, ctypes.c_void_p(om.data_ptr() + off_m), bs, 1, P(w), P(go),
                                  P(gx) if data else None, P(gom) if data else None, bs,
                                  ctypes.c_void_p(gom.data_ptr() + off_m) if data else None, bs,
                                  None if data else P(gw), None, int(os.environ.get("PMC_BWD_FLAGS", "0")), P(ws), ws_bytes, _C.stream())
        assert rc == 0, rc
torch.cuda.synchronize()
