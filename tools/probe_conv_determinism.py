"""GPU probe: run every tile form of the MFMA convolution several times on the same inputs and report launches whose
output differs from run to run (the forward kernels use no atomics: any difference is a race)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import _C

L = _C.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
bad = 0
for (B, ci, co, H, W, k, s) in [(1, 128, 256, 256, 256, 3, 2), (1, 256, 256, 128, 128, 3, 2), (1, 256, 384, 64, 64, 3, 2),
                                (1, 384, 384, 32, 32, 3, 2), (1, 384, 512, 16, 16, 3, 2), (1, 256, 256, 128, 128, 3, 1),
                                (1, 256, 256, 64, 64, 3, 1), (1, 384, 384, 32, 32, 3, 1), (1, 384, 384, 16, 16, 3, 1),
                                (1, 512, 512, 8, 8, 3, 1), (1, 256, 256, 128, 128, 1, 1), (1, 256, 384, 64, 64, 1, 1),
                                (1, 384, 384, 32, 32, 1, 1), (1, 512, 384, 8, 8, 1, 1), (1, 256, 16, 128, 128, 3, 1),
                                (1, 64, 64, 256, 512, 3, 1), (1, 128, 128, 128, 256, 3, 1), (1, 512, 512, 32, 64, 3, 1)]:
    torch.manual_seed(1)
    x = torch.randn(B, ci, H, W, device="cuda"); w = torch.randn(co, ci, k, k, device="cuda") * 0.05
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    wp = torch.empty(L.cp_conv_mfma_weight_bytes(ci, co, k * k), dtype=torch.uint8, device="cuda")
    assert L.cp_conv_mfma_prepare(P(w), ci, co, k * k, 0, P(wp), _C.stream()) == 0
    ptrs, chans = (ctypes.c_void_p * 1)(x.data_ptr()), (ctypes.c_int32 * 1)(ci)
    outs = []
    for r in range(6):
        out = torch.full((B, co, Ho, Wo), float("nan"), device="cuda")
        rc = L.cp_conv_mfma_forward_strided(ptrs, chans, 1, P(wp), None, None, P(out), B, H, W, co, k * k, s, 0, _C.stream())
        assert rc == 0, rc
        outs.append(out)
    torch.cuda.synchronize()
    ndiff = [int((outs[0] != o).sum().item()) for o in outs[1:]]
    nan = int(torch.isnan(outs[0]).sum().item())
    flag = "  <-- DIFFERS" if any(ndiff) or nan else ""
    bad += bool(flag)
    print("B%d %d->%d %dx%d k%d s%d: differing elements per rerun %s, nan %d%s" % (B, ci, co, H, W, k, s, ndiff, nan, flag), flush=True)
print("nondeterministic shapes:", bad)
