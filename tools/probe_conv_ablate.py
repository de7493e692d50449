"""GPU probe (timing only): ablation builds of the MFMA convolution (make -C centerpoly_amd/csrc libcp_cvabl_<mask>.so;
results wrong by construction) against the production build, on the inference (B = 1) and training (B = 4) layer shapes."""
import ctypes, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import _C

here = os.path.dirname(_C.LIB_PATH)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
vp, i32 = ctypes.c_void_p, ctypes.c_int32
BITS = {1: "no W reloads", 2: "no staging loads", 4: "no B ds_reads", 8: "no MFMA", 16: "no staging"}
only = [int(v) for v in sys.argv[1:]]
libs = [(0, _C.LIB_PATH)] + sorted((int(os.path.basename(p)[12:-3]), p) for p in glob.glob(os.path.join(here, "libcp_cvabl_*.so")))
SHAPES = [(1, 64, 64, 256, 512), (1, 128, 128, 128, 256), (1, 256, 256, 64, 128), (1, 512, 512, 32, 64), (4, 64, 64, 256, 512),
          (4, 128, 128, 128, 256), (4, 256, 256, 64, 128), (4, 512, 512, 32, 64), (4, 64, 32, 256, 512), (1, 64, 256, 256, 512)]
for (B, ci, co, H, W) in SHAPES:
    x = torch.randn(B, ci, H, W, device="cuda"); w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    out = torch.empty(B, co, H, W, device="cuda")
    for mask, path in libs:
        if only and mask not in only:
            continue
        L = ctypes.CDLL(path)
        L.cp_conv3x3_mfma_weight_bytes.restype = ctypes.c_size_t
        L.cp_conv3x3_mfma_weight_bytes.argtypes = [i32, i32]
        L.cp_conv3x3_mfma_prepare.argtypes = [vp, i32, i32, i32, vp, vp]
        L.cp_conv3x3_mfma_forward.argtypes = [vp] * 5 + [i32] * 6 + [vp]
        wp = torch.empty(L.cp_conv3x3_mfma_weight_bytes(ci, co), dtype=torch.uint8, device="cuda")
        st = _C.stream()
        assert L.cp_conv3x3_mfma_prepare(P(w), ci, co, 0, P(wp), st) == 0
        call = lambda: L.cp_conv3x3_mfma_forward(P(x), P(wp), None, None, P(out), B, ci, H, W, co, 0, st)
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            call()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 40
        nm = " + ".join(v for k, v in BITS.items() if mask & k) or "full"
        print("B%d %3d->%3d %3dx%3d  %-34s %.1f us  (%.0f TF/s fp32-equivalent)" % (B, ci, co, H, W, nm, t * 1e3, 2.0 * B * ci * co * 9 * H * W / t / 1e9), flush=True)
