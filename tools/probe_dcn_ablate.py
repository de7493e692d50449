"""GPU probe: where does the DCN forward kernel spend its time?  Uses the timing-only
ablation build (libcp_ablate.so, results are wrong by construction)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import _C
LIBDIR = os.path.dirname(_C.LIB_PATH)
L = None


def use(mask):
    """Compile-time ablation build libcp_abl_<mask>.so (make -C centerpoly_amd/csrc libcp_abl_<mask>.so)."""
    global L
    L = ctypes.CDLL(os.path.join(LIBDIR, "libcp_abl_%d.so" % mask))
    for n in ("cp_dcn_v2_forward", "cp_dcn_v2_forward_workspace_bytes"):
        getattr(L, n).restype, getattr(L, n).argtypes = _C._SIGNATURES[n]


def run(ci, co, H, W, n=50):
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
    for _ in range(20): call()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

names = {0: "full", 1: "no gathers", 2: "no LDS writes", 4: "no W loads", 5: "no loads", 8: "no MFMA", 32: "no ds_reads",
         40: "no MFMA/ds_reads", 39: "MFMA only", 46: "gathers only", 47: "skeleton"}
for shape in [(64, 64, 256, 512), (128, 128, 128, 256)]:
    for flag, nm in names.items():
        if not os.path.exists(os.path.join(LIBDIR, "libcp_abl_%d.so" % flag)):
            continue
        use(flag)
        print("%s %-16s %.3f ms" % (shape, nm, run(*shape)), flush=True)
