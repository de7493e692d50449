"""GPU probe (timing only): ablation builds of the MFMA convolution (libcp_cvabl_A.so: weight fragments never
re-loaded; libcp_cvabl_B.so: activation fragments all read from one LDS address) against the production build."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import _C

here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "centerpoly_amd", "csrc")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
vp, i32 = ctypes.c_void_p, ctypes.c_int32
for name in ["libcenterpoly_hip.so", "libcp_cvold.so", "libcenterpoly_hip.so", "libcp_cvold.so"]:
    path = os.path.join(here, name)
    if not os.path.exists(path):
        continue
    L = ctypes.CDLL(path)
    L.cp_conv3x3_mfma_weight_bytes.restype = ctypes.c_size_t
    L.cp_conv3x3_mfma_weight_bytes.argtypes = [i32, i32]
    L.cp_conv3x3_mfma_prepare.argtypes = [vp, i32, i32, i32, vp, vp]
    L.cp_conv3x3_mfma_forward.argtypes = [vp] * 5 + [i32] * 6 + [vp]
    for (B, ci, co, H, W) in [(4, 64, 256, 256, 512), (1, 64, 1024, 256, 512), (1, 64, 64, 256, 512), (1, 128, 128, 128, 256), (1, 256, 256, 64, 128), (4, 128, 128, 64, 128), (4, 256, 256, 32, 64)]:
        x = torch.randn(B, ci, H, W, device="cuda"); w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
        out = torch.empty(B, co, H, W, device="cuda")
        wp = torch.empty(L.cp_conv3x3_mfma_weight_bytes(ci, co), dtype=torch.uint8, device="cuda")
        st = _C.stream()
        assert L.cp_conv3x3_mfma_prepare(P(x) and P(w), ci, co, 0, P(wp), st) == 0
        call = lambda: L.cp_conv3x3_mfma_forward(P(x), P(wp), None, None, P(out), B, ci, H, W, co, 0, st)
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20
        print("%-22s B%d %3d->%3d %3dx%3d  %.3f ms  %.0f TF/s" % (name, B, ci, co, H, W, t, 2.0 * B * ci * co * 9 * H * W / t / 1e9), flush=True)
