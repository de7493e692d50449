"""Diagnostic: where the waves of the DCNv2 backward data kernel spend their cycles (s_memtime stamps,
`make -C centerpoly_amd/csrc libcp_stamp.so`).  Prints per-role averages over the first 1024 tiles."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from centerpoly_amd import _C, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "centerpoly_amd", "csrc", "libcp_stamp.so"))
L.cp_dcn_v2_backward_workspace_bytes.restype = ctypes.c_size_t
L.cp_dcn_v2_backward_workspace_bytes.argtypes = [ctypes.POINTER(_C.DcnShape)]
L.cp_dcn_v2_backward.restype = ctypes.c_int32
L.cp_dcn_v2_backward.argtypes = _C._SIGNATURES["cp_dcn_v2_backward"][1]
B, ci, co, H, W = [int(v) for v in os.environ.get("PMC_SHAPE", "4,64,64,256,512").split(",")]
dev = "cuda"
x = torch.from_numpy(synth.normal("pmcb/x", (B, ci, H, W))).to(dev)
om = torch.from_numpy(synth.normal("pmcb/om", (B, 27, H, W)) * float(os.environ.get("PMC_OFF_STD", "0.5"))).to(dev)
w = torch.from_numpy(synth.normal("pmcb/w", (co, ci, 3, 3), 0, 0.05)).to(dev)
go = torch.from_numpy(synth.normal("pmcb/go", (B, co, H, W))).to(dev)
gx = torch.zeros_like(x); gom = torch.empty_like(om)
s = _C.DcnShape(B, ci, H, W, co, 3, 3, 1, 1, 1, 1)
bs = 27 * H * W; off_m = 72 * H * W
P = lambda t: ctypes.c_void_p(t.data_ptr())
nws = L.cp_dcn_v2_backward_workspace_bytes(s)
ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
for _ in range(3):
    rc = L.cp_dcn_v2_backward(s, P(x), P(om), bs, ctypes.c_void_p(om.data_ptr() + off_m), bs, 1, P(w), P(go), P(gx),
                              P(gom), bs, ctypes.c_void_p(gom.data_ptr() + off_m), bs, None, None, 0, P(ws), nws,
                              _C.stream())
    assert rc == 0
torch.cuda.synchronize()
d = ws[nws - 1024 * 16 * 8 * 8:].view(torch.int64).cpu().numpy()[:1024 * 8 * 8].reshape(1024, 8, 8).copy()
names = ["mfma", "consume", "wait@odd barrier", "wait@stage barrier", "x store + flush", "loop", "recipe", "barrier0..loop"]
print("wave0: breg+bound phase %.0f" % d[:, 1, 6].mean())
ent = np.sort(d[:, 1, 7].astype(np.float64))
print("entry-time spread of the first 1024 tiles: min %.0f  median %.0f  max %.0f (cycles, relative to the earliest)"
      % (0, np.median(ent) - ent[0], ent[-1] - ent[0]))
d[:, 1, 6:8] = d[:, 0, 6:8]
for role, sl in (("X (waves 0-3)", slice(0, 4)), ("Y (waves 4-7)", slice(4, 8))):
    v = d[:, sl].reshape(-1, 8).astype(np.float64)
    print(role, "  ".join("%s %.0f" % (n, v[:, i].mean()) for i, n in enumerate(names)))

nwg = B * ((H + 7) // 8) * ((W + 15) // 16)
e = ws[nws - 1024 * 16 * 8 * 8 + 1024 * 8 * 8 * 8:].view(torch.int64).cpu().numpy()[:nwg * 4].reshape(nwg, 4)
t0 = e[:, 0].min()
span = e[:, 1].max() - t0
life = (e[:, 1] - e[:, 0]).astype(np.float64)
print("workgroups %d  span %.0f cycles  lifetime mean %.0f median %.0f min %.0f max %.0f  sum(lifetime)/span = %.1f workgroups alive on average"
      % (nwg, span, life.mean(), np.median(life), life.min(), life.max(), life.sum() / span))
order = np.argsort(e[:, 0])
for q in (0, 300, 1000, 2000, 4000):
    i = order[min(q, nwg - 1)]
    print("  wg #%d in entry order: entry +%.0f  lifetime %.0f" % (q, e[i, 0] - t0, life[i]))
cu = (e[:, 3] & 0xf) * 1000 + ((e[:, 2] >> 8) & 0xf) + 16 * ((e[:, 2] >> 12) & 0x3) + 64 * ((e[:, 2] >> 13) & 0x7)
print("distinct (xcc, cu/sh/se) ids: %d" % len(np.unique(cu)))
# gaps between consecutive workgroups on the same CU
gaps = []
for c in np.unique(cu):
    w = e[cu == c]
    w = w[np.argsort(w[:, 0])]
    gaps += list((w[1:, 0] - w[:-1, 1]).astype(np.float64))
print("gap between a workgroup's exit and the next one's entry on the same id: mean %.0f  median %.0f" % (np.mean(gaps), np.median(gaps)))
