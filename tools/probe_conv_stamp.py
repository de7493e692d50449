"""GPU probe: phase durations inside the MFMA convolution (diagnostic build libcp_cvstamp.so: wave 0 of every workgroup
writes s_memtime deltas past out[]).  Per chunk: stage (loads + split + LDS stores) | barrier | taps | next chunk's top barrier.
The in-kernel clock (s_memtime / s_memrealtime) shows what the chip holds under this kernel's load."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centerpoly_amd import _C
L = ctypes.CDLL(os.path.join(os.path.dirname(_C.LIB_PATH), "libcp_cvstamp.so"))
vp, i32 = ctypes.c_void_p, ctypes.c_int32
L.cp_conv3x3_mfma_weight_bytes.restype = ctypes.c_size_t
L.cp_conv3x3_mfma_weight_bytes.argtypes = [i32, i32]
L.cp_conv3x3_mfma_prepare.argtypes = [vp, i32, i32, i32, vp, vp]
L.cp_conv3x3_mfma_forward.argtypes = [vp] * 5 + [i32] * 6 + [vp]
P = _C.ptr
for (B, ci, co, H, W, nwg) in [(4, 64, 64, 256, 512, 512), (1, 64, 64, 256, 512, 1024), (1, 128, 128, 128, 256, 512)]:
    x = torch.randn(B, ci, H, W, device="cuda"); w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    out = torch.zeros(B + 1, co, H, W, device="cuda")
    wp = torch.empty(L.cp_conv3x3_mfma_weight_bytes(ci, co), dtype=torch.uint8, device="cuda")
    st = _C.stream()
    assert L.cp_conv3x3_mfma_prepare(P(w), ci, co, 0, P(wp), st) == 0
    for _ in range(4):
        assert L.cp_conv3x3_mfma_forward(P(x), P(wp), None, None, P(out), B, ci, H, W, co, 0, st) == 0
    torch.cuda.synchronize()
    s = out[B].flatten()[:B * nwg * 16].view(-1, 16).cpu().numpy()
    s = s[s[:, 13] > 0]
    n = int(s[0, 13])
    names = ["prologue"] + sum([["c%d stage" % c, "c%d barrier" % c, "c%d taps" % c, "c%d top barrier" % (c + 1)] for c in range(4)], [])
    tot = s[:, :13].sum(1)
    mhz = np.median(tot / (s[:, 14] / 100.0))
    print("B%d %d->%d @%dx%d: %d workgroups stamped, %d stamps, in-kernel clock %.0f MHz" % (B, ci, co, H, W, len(s), n, mhz))
    for i in range(n - 1):
        print("   %-16s mean %8.0f  median %8.0f  p90 %8.0f cycles" % (names[i], s[:, i].mean(), np.median(s[:, i]), np.percentile(s[:, i], 90)))
    start = (s[:, 15] - s[:, 15].min()) % (1 << 24)
    print("   total mean %.0f cycles = %.1f us; start percentiles 0/50/90/100 (us): %s; duration 10/50/90 (us): %s" % (
        tot.mean(), tot.mean() / mhz, (np.percentile(start, [0, 50, 90, 100]) / 100).round(1), (np.percentile(s[:, 14], [10, 50, 90]) / 100).round(1)))
