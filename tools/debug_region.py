import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centerpoly_amd.models.networks.DCNv2.dcn_v2 import dcn_v2_forward_raw
dev = "cuda"
torch.manual_seed(0)
B, Ci, Co, H, W = 1, 32, 64, 8, 32
x = torch.randn(B, Ci, H, W, device=dev)
om = torch.zeros(B, 27, H, W, device=dev)
b = torch.zeros(Co, device=dev)
def run(w, om=om):
    ref = dcn_v2_forward_raw(x, om, w, b, contraction="f32")
    out = dcn_v2_forward_raw(x, om, w, b, contraction="bf16x3_region")
    return ref, out
def onehot(co, ci, t):
    w = torch.zeros(Co, Ci, 3, 3, device=dev)
    w.view(Co, Ci, 9)[co, ci, t] = 1.0
    ref, out = run(w)
    return (ref - out).abs().max().item(), out.abs().max().item(), ref, out
print("taps, ci=5 co=3:", ["%d:%.1e/%.1e" % ((t,) + onehot(3, 5, t)[:2]) for t in range(9)])
print("ci, t=4 co=3:", ["%d:%.1e" % (ci, onehot(3, ci, 4)[0]) for ci in range(32)])
print("co, t=4 ci=5:", ["%d:%.1e" % (co, onehot(co, 5, 4)[0]) for co in range(0, 64, 3)])
e, m, ref, out = onehot(3, 5, 0)
print("t=0 rows of ref/out col 5:", ref[0, 3, :, 5].tolist(), out[0, 3, :, 5].tolist())
e, m, ref, out = onehot(3, 5, 1)
print("t=1 rows of ref/out col 5:", ref[0, 3, :, 5].tolist(), out[0, 3, :, 5].tolist())
e, m, ref, out = onehot(3, 5, 3)
print("t=3 row 2 of ref/out:", ref[0, 3, 2, :8].tolist(), out[0, 3, 2, :8].tolist())
