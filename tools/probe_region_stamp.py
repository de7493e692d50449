"""GPU probe: phase durations inside the region DCN forward kernel (diagnostic build libcp_rstamp.so: wave 0 of every
workgroup writes s_memtime deltas over out[]).  Phases: recipes | bbox + addresses | chunk-0 staging | K loop | epilogue."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import _C
L = ctypes.CDLL(os.path.join(os.path.dirname(_C.LIB_PATH), "libcp_rstamp.so"))
for n in ("cp_dcn_v2_forward", "cp_dcn_v2_forward_workspace_bytes"):
    getattr(L, n).restype, getattr(L, n).argtypes = _C._SIGNATURES[n]
dev = "cuda"
for (B, ci, co, H, W) in [(1, 64, 64, 256, 512), (4, 64, 64, 256, 512), (1, 128, 128, 128, 256)]:
    torch.manual_seed(1)
    x = torch.randn(B, ci, H, W, device=dev); om = torch.randn(B, 27, H, W, device=dev) * 0.3
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05; b = torch.randn(co, device=dev)
    out = torch.empty(B, co, H, W, device=dev)
    s = _C.DcnShape(B, ci, H, W, co, 3, 3, 1, 1, 1, 1)
    nws = L.cp_dcn_v2_forward_workspace_bytes(s); ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
    bs = 27 * H * W
    for it in range(5):
        rc = L.cp_dcn_v2_forward(s, _C.ptr(x), _C.ptr(om), bs, ctypes.c_void_p(om.data_ptr() + 72 * H * W), bs, 1,
                                 _C.ptr(w), _C.ptr(b), None, None, 0, 3 if it == 0 else 4, _C.ptr(out), _C.ptr(ws), nws, _C.stream())
        assert rc == 0
    torch.cuda.synchronize()
    nt = (H // 8) * (W // 32)
    st = out.flatten()[:B * nt * 8].view(-1, 8).cpu()
    names = ["recipes", "bbox+addr", "stage0", "K loop", "epilogue"]
    tot = st[:, :5].sum(1)
    clk = tot / (st[:, 5] / 100.0)                           # cycles per us = MHz
    print("B%d %d->%d @%dx%d: %d workgroups, in-kernel clock %.0f MHz (median)" % (B, ci, co, H, W, st.shape[0], clk.median()))
    for i, n in enumerate(names):
        print("   %-10s mean %8.0f  median %8.0f  max %8.0f cycles" % (n, st[:, i].mean(), st[:, i].median(), st[:, i].max()))
    print("   total      mean %8.0f cycles = %.1f us at the median clock" % (tot.mean(), tot.mean() / clk.median()))
    import numpy as np
    stn = st.numpy()
    start = stn[:, 6]
    start = (start - start.min()) % (1 << 24)
    dur = stn[:, 5]
    print("   start (us after the first workgroup) percentiles 0/50/90/100:", (np.percentile(start, [0, 50, 90, 100]) / 100).round(2),
          " duration us 10/50/90/100:", (np.percentile(dur, [10, 50, 90, 100]) / 100).round(1))
    hw = stn[:, 7].astype(np.int64)
    xcc, cu, sh, se = hw >> 16, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    ids, cnt = np.unique(cuid, return_counts=True)
    print("   distinct CUs %d; workgroups per CU histogram: %s" % (len(ids), dict(zip(*np.unique(cnt, return_counts=True)))))
    for x in range(8):
        m = xcc == x
        if m.any():
            print("   XCC %d: %3d workgroups, K loop median %6.0f max %6.0f, duration median %.1f us" %
                  (x, m.sum(), np.median(stn[m, 3]), stn[m, 3].max(), np.median(dur[m]) / 100))
    per_cu = {c: stn[cuid == c, 3] for c in ids}
    solo = np.array([v.mean() for v in per_cu.values() if len(v) == 1])
    duo = np.array([v.mean() for v in per_cu.values() if len(v) == 2])
    tri = np.array([v.mean() for v in per_cu.values() if len(v) >= 3])
    print("   K loop mean by workgroups sharing the CU: 1 -> %s, 2 -> %s, >=3 -> %s" % (solo.mean().round() if len(solo) else None,
          duo.mean().round() if len(duo) else None, tri.mean().round() if len(tri) else None))
