"""GPU probe: the 3x3 MFMA convolution on float32 tensors against the SPLIT-plane forms (cp_conv_mfma_forward_split:
input staged from [hi | lo] bf16 planes, output written as such planes) at the DLA-34 BasicBlock shapes of the
1 x 3 x 1024 x 2048 inference step.  Prints us per launch of the four in/out combinations and checks that every
combination gives the float32 route's values bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import _C
L = _C.lib()
dev = "cuda"
torch.manual_seed(0)


def timed(call, n=40):
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (C, H, W) in [(64, 256, 512), (128, 128, 256), (256, 64, 128), (512, 32, 64)]:
    x = torch.randn(1, C, H, W, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * (1.0 / (3 * C ** 0.5))
    b = torch.randn(C, device=dev) * 0.1
    wp = torch.empty(L.cp_conv_mfma_weight_bytes(C, C, 9), dtype=torch.uint8, device=dev)
    _C.check(L.cp_conv_mfma_prepare(_C.ptr(w), C, C, 9, 0, _C.ptr(wp), _C.stream()), "prepare")
    xs = torch.empty_like(x)                       # the split planes occupy the float32 tensor's bytes
    _C.check(L.cp_activation_split(_C.ptr(x), _C.ptr(xs), 1, C, H, W, _C.stream()), "split")
    outs = {}
    line = "%3d->%3d @%dx%d:" % (C, C, H, W)
    for (xi, oi) in [(0, 0), (1, 0), (0, 1), (1, 1)]:
        out = torch.empty_like(x)

        def call():
            _C.check(L.cp_conv_mfma_forward_split(_C.ptr(xs if xi else x), xi, _C.ptr(wp), _C.ptr(b), None, _C.ptr(out), oi,
                                                  1, C, H, W, C, 9, 1, 1, _C.stream()), "fwd")
        t = timed(call)
        if oi:
            f = torch.empty_like(x)
            _C.check(L.cp_activation_unsplit(_C.ptr(out), _C.ptr(f), 1, C, H, W, _C.stream()), "unsplit")
            ref = torch.empty_like(x)             # what the split of the float32 result reads back as
            tmp = torch.empty_like(x)
            _C.check(L.cp_activation_split(_C.ptr(outs[(0, 0)]), _C.ptr(tmp), 1, C, H, W, _C.stream()), "split")
            same = torch.equal(out.view(torch.int32), tmp.view(torch.int32))
        else:
            same = (xi, oi) == (0, 0) or torch.equal(out, outs[(0, 0)])
        outs[(xi, oi)] = out
        line += "  %s->%s %.1f us%s" % ("split" if xi else "f32", "split" if oi else "f32", t, "" if same else " (DIFFERS)")
    print(line, flush=True)
