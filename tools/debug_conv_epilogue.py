"""GPU debug: identity-weight convolution through cp_conv3x3_mfma_forward; prints which (channel, row, column) classes differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import _C
L = _C.lib(); P = _C.ptr
for (B, C, H, W) in [(2, 64, 32, 64), (1, 64, 256, 512)]:
    x = torch.randn(B, C, H, W, device="cuda")
    w = torch.zeros(C, C, 3, 3, device="cuda")
    for i in range(C):
        w[i, i, 1, 1] = 1.0
    wp = torch.empty(L.cp_conv3x3_mfma_weight_bytes(C, C), dtype=torch.uint8, device="cuda")
    assert L.cp_conv3x3_mfma_prepare(P(w), C, C, 0, P(wp), _C.stream()) == 0
    out = torch.full_like(x, float("nan"))
    assert L.cp_conv3x3_mfma_forward(P(x), P(wp), None, None, P(out), B, C, H, W, C, 0, _C.stream()) == 0
    bad = (out - x).abs() > 1e-3
    print((B, C, H, W), "bad fraction", bad.float().mean().item())
    idx = bad.nonzero()
    if len(idx):
        print(" bad channels mod 16:", sorted(set((idx[:, 1] % 16).tolist())))
        print(" bad rows mod 8:", sorted(set((idx[:, 2] % 8).tolist())))
        print(" bad cols mod 32:", sorted(set((idx[:, 3] % 32).tolist())))
        b, c, y, xx = idx[0].tolist()
        v = out[b, c, y, xx].item()
        m = (x[b] - v).abs() < 1e-6
        print(" first bad", (b, c, y, xx), "holds the value of", m.nonzero()[:3].tolist())
