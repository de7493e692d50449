"""GPU probe: where does the DCN forward kernel spend its time?  Uses the timing-only
ablation build (libcp_ablate.so, results are wrong by construction)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import _C
_C.LIB_PATH = os.path.join(os.path.dirname(_C.LIB_PATH), "libcp_ablate.so")
L = ctypes.CDLL(_C.LIB_PATH)
for n in ("cp_dcn_v2_forward", "cp_dcn_v2_forward_workspace_bytes"):
    getattr(L, n).restype, getattr(L, n).argtypes = _C._SIGNATURES[n]

def run(ci, co, H, W, n=10):
    dev = "cuda"
    x = torch.randn(1, ci, H, W, device=dev); om = torch.randn(1, 27, H, W, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev); b = torch.randn(co, device=dev)
    out = torch.empty(1, co, H, W, device=dev)
    s = _C.DcnShape(1, ci, H, W, co, 3, 3, 1, 1, 1, 1)
    nws = L.cp_dcn_v2_forward_workspace_bytes(s); ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
    bs = 27 * H * W
    def call():
        rc = L.cp_dcn_v2_forward(s, _C.ptr(x), _C.ptr(om), bs, ctypes.c_void_p(om.data_ptr() + 72 * H * W), bs, 1,
                                 _C.ptr(w), _C.ptr(b), None, None, 0, 0, _C.ptr(out), _C.ptr(ws), nws, _C.stream())
        assert rc == 0, rc
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

names = {0: "full", 1: "no gathers", 2: "no colT writes", 4: "no W staging", 8: "no MFMA", 7: "MFMA only", 14: "gathers only", 15: "skeleton"}
for shape in [(64, 64, 256, 512), (256, 256, 64, 128)]:
    for flag, nm in names.items():
        os.environ["CP_DCN_ABLATE"] = str(flag)
        print("%s %-16s %.3f ms" % (shape, nm, run(*shape)), flush=True)
