"""GPU probe: the three full-resolution DLA base layers (stem 7x7 3->16, level0 3x3 16->16, level1 3x3 s2
16->32 at 1x3x1024x2048): hand-written direct convolution with fused bias + ReLU against the library
convolution + the separate bias/ReLU pass it needed (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import _C, synth

L = _C.lib()
dev = "cuda"


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = int(os.environ.get("PROBE_B", "1"))
x = torch.from_numpy(synth.normal("pc/x", (B, 3, 1024, 2048))).to(dev)
cin = 3
for name, cout, k, stride in (("stem 7x7 3->16", 16, 7, 1), ("level0 3x3 16->16", 16, 3, 1), ("level1 3x3 s2 16->32", 32, 3, 2)):
    w = torch.from_numpy(synth.normal("pc/w%d%d" % (cout, k), (cout, cin, k, k), 0, 0.1)).to(dev)
    b = torch.from_numpy(synth.normal("pc/b%d%d" % (cout, k), (cout,), 0, 0.1)).to(dev)
    pad = k // 2
    H, W = x.shape[2:]
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((B, cout, Ho, Wo), device=dev)

    def direct():
        rc = L.cp_conv_direct_forward(_C.ptr(x), _C.ptr(w), _C.ptr(b), _C.ptr(out), B, cin, H, W, cout, k, stride, pad, 1,
                                      _C.stream())
        assert rc == 0

    def library():
        y = torch.nn.functional.conv2d(x, w, None, stride=stride, padding=pad)
        rc = L.cp_bias_act_inplace(_C.ptr(y), _C.ptr(b), None, B, cout, Ho * Wo, 1, _C.stream())
        assert rc == 0
        return y

    flops = 2.0 * cin * k * k * cout * Ho * Wo * B
    td, tl = timeit(direct), timeit(library)
    print("%-22s direct %7.1f us (%5.1f TFLOP/s)   library conv + bias/ReLU pass %7.1f us" % (name, td, flops / td / 1e6, tl), flush=True)
    x, cin = out.clone(), cout


def wgrad(cin, cout, k, stride, pad, B=4, H=512, W=1024):
    x = torch.randn(B, cin, H, W, device="cuda")
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    go = torch.randn(B, cout, Ho, Wo, device="cuda")
    gw = torch.zeros(cout, cin, k, k, device="cuda")
    L = _C.lib()
    call = lambda: L.cp_conv_direct_wgrad(_C.ptr(x), _C.ptr(go), _C.ptr(gw), B, cin, H, W, cout, k, stride, pad, _C.stream())
    lib = lambda: torch.nn.grad.conv2d_weight(x, (cout, cin, k, k), go, stride=stride, padding=pad)
    out = []
    for fn in (call, lib):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 10)
    print("wgrad %d->%d k%d s%d @%dx%d x%d: direct %.3f ms   library %.3f ms" % (cin, cout, k, stride, H, W, B, out[0], out[1]), flush=True)


for cfg in [(3, 16, 7, 1, 3), (16, 16, 3, 1, 1), (16, 32, 3, 2, 1)]:
    wgrad(*cfg)
