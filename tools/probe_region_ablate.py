"""GPU probe: where does the region DCN forward kernel spend its time?  Timing-only ablation builds
(make -C centerpoly_amd/csrc libcp_rabl_<mask>.so; results are wrong by construction)."""
import ctypes, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import _C
LIBDIR = os.path.dirname(_C.LIB_PATH)
BITS = {1: "no W reloads", 2: "no staging", 4: "no ds_reads", 8: "no MFMA", 16: "no sample VALU", 100: "full, SLP-vectorised (packed f32 VALU)"}


def load(path):
    L = ctypes.CDLL(path)
    for n in ("cp_dcn_v2_forward", "cp_dcn_v2_forward_workspace_bytes"):
        getattr(L, n).restype, getattr(L, n).argtypes = _C._SIGNATURES[n]
    return L


def run(L, B, ci, co, H, W, scale, n=40):
    dev = "cuda"
    torch.manual_seed(1)
    x = torch.randn(B, ci, H, W, device=dev); om = torch.randn(B, 27, H, W, device=dev) * scale
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05; b = torch.randn(co, device=dev)
    out = torch.empty(B, co, H, W, device=dev)
    s = _C.DcnShape(B, ci, H, W, co, 3, 3, 1, 1, 1, 1)
    nws = L.cp_dcn_v2_forward_workspace_bytes(s); ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
    bs = 27 * H * W
    def call(mode):
        rc = L.cp_dcn_v2_forward(s, _C.ptr(x), _C.ptr(om), bs, ctypes.c_void_p(om.data_ptr() + 72 * H * W), bs, 1,
                                 _C.ptr(w), _C.ptr(b), None, None, 0, mode, _C.ptr(out), _C.ptr(ws), nws, _C.stream())
        assert rc == 0, rc
    call(3)
    for _ in range(10): call(4)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): call(4)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


only = [int(v) for v in sys.argv[1:]]
extra = [(200, os.path.join(LIBDIR, "libcp_rprio.so"))] if os.path.exists(os.path.join(LIBDIR, "libcp_rprio.so")) else []
BITS[200] = "full, WITHOUT the per-chunk priority alternation between the workgroups of a CU"
libs = [(0, _C.LIB_PATH)] + extra + sorted((int(os.path.basename(p)[11:-3]), p) for p in glob.glob(os.path.join(LIBDIR, "libcp_rabl_*.so")))
for shape in [(1, 64, 64, 256, 512), (1, 128, 128, 128, 256)]:
    for scale in (0.3,):
        for mask, path in libs:
            if only and mask not in only:
                continue
            nm = BITS[mask] if mask >= 100 else (" + ".join(v for k, v in BITS.items() if mask & k) or "full")
            print("%s offsets %.1f  %-50s %.1f us" % (shape, scale, nm, run(load(path), *shape, scale)), flush=True)
