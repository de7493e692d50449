"""GPU probe: DCNv2 backward launch times (production library, HIP events) for the layer shapes
of DLA-34 at the Cityscapes and KITTI input sizes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import _C, synth

L = _C.lib()
dev = "cuda"
FLAGS = int(os.environ.get("PROBE_BWD_FLAGS", "0"))      # bit-or of _C.DCN_BWD_* (1 exact f32, 2 narrow tiles, 4 round-1 kernels)


def run(B, ci, co, H, W, what, n=20, off_scale=float(os.environ.get("PROBE_OFF_STD", "0.5"))):
    x = torch.from_numpy(synth.normal("pb/x", (B, ci, H, W))).to(dev)
    om = torch.from_numpy(synth.normal("pb/om", (B, 27, H, W)) * off_scale).to(dev)
    w = torch.from_numpy(synth.normal("pb/w", (co, ci, 3, 3), 0, 0.05)).to(dev)
    go = torch.from_numpy(synth.normal("pb/go", (B, co, H, W))).to(dev)
    gx = torch.zeros_like(x); gom = torch.empty_like(om); gw = torch.zeros_like(w)
    s = _C.DcnShape(B, ci, H, W, co, 3, 3, 1, 1, 1, 1)
    bs = 27 * H * W; off_m = 72 * H * W
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    data, weight = what != "weight", what != "data"
    nws = L.cp_dcn_v2_backward_workspace_bytes(s)
    ws = _C.workspace(nws, dev)

    def call():
        rc = L.cp_dcn_v2_backward(s, P(x), P(om), bs, ctypes.c_void_p(om.data_ptr() + off_m), bs, 1, P(w), P(go),
                                  P(gx) if data else None, P(gom) if data else None, bs,
                                  ctypes.c_void_p(gom.data_ptr() + off_m) if data else None, bs,
                                  P(gw) if weight else None, None, FLAGS, P(ws), nws, _C.stream())
        assert rc == 0, rc
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for shape in [(1, 64, 64, 256, 512), (4, 64, 64, 256, 512), (4, 128, 128, 128, 256), (4, 256, 256, 64, 128),
              (8, 64, 64, 96, 320), (8, 128, 128, 48, 160)]:
    print("%-28s data %.3f ms   weight %.3f ms" % (shape, run(*shape, "data"), run(*shape, "weight")), flush=True)
