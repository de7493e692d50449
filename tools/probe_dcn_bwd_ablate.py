"""GPU probe: time split of the tiled DCN backward-data kernel (timing-only ablation build)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import _C
L = ctypes.CDLL(os.path.join(os.path.dirname(_C.LIB_PATH), "libcp_ablate.so"))
L.cp_dcn_v2_backward.restype, L.cp_dcn_v2_backward.argtypes = _C._SIGNATURES["cp_dcn_v2_backward"]

def run(ci, co, H, W, what, n=5, off_scale=0.5):
    dev = "cuda"
    x = torch.randn(1, ci, H, W, device=dev); om = torch.randn(1, 27, H, W, device=dev) * off_scale
    w = torch.randn(co, ci, 3, 3, device=dev); go = torch.randn(1, co, H, W, device=dev)
    gx = torch.zeros_like(x); gom = torch.empty_like(om); gw = torch.zeros_like(w); gb = torch.zeros(co, device=dev)
    s = _C.DcnShape(1, ci, H, W, co, 3, 3, 1, 1, 1, 1)
    bs = 27 * H * W; off_m = 72 * H * W
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    def call():
        rc = L.cp_dcn_v2_backward(s, P(x), P(om), bs, ctypes.c_void_p(om.data_ptr() + off_m), bs, 1, P(w), P(go),
                                  P(gx) if what != "weight" else None, P(gom) if what != "weight" else None, bs,
                                  ctypes.c_void_p(gom.data_ptr() + off_m) if what != "weight" else None, bs,
                                  P(gw) if what != "data" else None, None, 0, None, 0, _C.stream())
        assert rc == 0, rc
    for _ in range(2): call()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

names = {0: "full", 1: "no consumption", 2: "no flush", 4: "no MFMA"}
for shape in [(64, 64, 256, 512), (256, 256, 64, 128)]:
    for flag, nm in names.items():
        os.environ["CP_DCN_ABLATE"] = str(flag)
        print("data  %s %-18s %.3f ms" % (shape, nm, run(*shape, "data")), flush=True)
    os.environ["CP_DCN_ABLATE"] = "0"
    print("weight %s %.3f ms" % (shape, run(*shape, "weight")), flush=True)
