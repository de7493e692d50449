"""GPU probe: the split-bf16 MFMA 3x3 convolution against the library convolution (error and launch time)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from centerpoly_amd import _C, synth

L = _C.lib()
dev = "cuda"
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(B, ci, co, H, W):
    x = torch.from_numpy(synth.normal("pc/x", (B, ci, H, W))).to(dev)
    w = torch.from_numpy(synth.normal("pc/w", (co, ci, 3, 3), 0, 0.05)).to(dev)
    bias = torch.from_numpy(synth.normal("pc/b", (co,))).to(dev)
    out = torch.empty((B, co, H, W), device=dev)
    wp = torch.empty(L.cp_conv3x3_mfma_weight_bytes(ci, co), dtype=torch.uint8, device=dev)
    assert L.cp_conv3x3_mfma_prepare(P(w), ci, co, 0, P(wp), _C.stream()) == 0
    call = lambda: _C.check(L.cp_conv3x3_mfma_forward(P(x), P(wp), P(bias), None, P(out), B, ci, H, W, co, 1, _C.stream()), "conv")
    call()
    ref = F.relu(F.conv2d(x.double(), w.double(), bias.double(), padding=1))
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    lib_out = F.relu(F.conv2d(x, w, bias, padding=1))
    err_lib = (lib_out.double() - ref).abs().max().item() / ref.abs().max().item()
    t = timeit(call)
    tl = timeit(lambda: F.conv2d(x, w, None, padding=1))
    fl = 2.0 * B * ci * co * 9 * H * W
    print("B%d %3d->%3d %3dx%3d  mfma %.3f ms (%.0f TF/s)  library %.3f ms (%.0f TF/s)  err %.2e (library %.2e)"
          % (B, ci, co, H, W, t, fl / t / 1e9, tl, fl / tl / 1e9, err, err_lib), flush=True)
    # input gradient through the transposed prologue
    go = torch.from_numpy(synth.normal("pc/go", (B, co, H, W))).to(dev)
    if co % 32 == 0:
        wpt = torch.empty(L.cp_conv3x3_mfma_weight_bytes(co, ci), dtype=torch.uint8, device=dev)
        assert L.cp_conv3x3_mfma_prepare(P(w), co, ci, 1, P(wpt), _C.stream()) == 0
        gx = torch.empty_like(x)
        _C.check(L.cp_conv3x3_mfma_forward(P(go), P(wpt), None, None, P(gx), B, co, H, W, ci, 0, _C.stream()), "conv")
        gref = torch.nn.grad.conv2d_input(x.shape, w.double(), go.double(), padding=1)
        print("      input gradient err %.2e" % ((gx.double() - gref).abs().max().item() / gref.abs().max().item()), flush=True)


def run_wgrad(B, ci, co, H, W):
    x = torch.from_numpy(synth.normal("pc/x", (B, ci, H, W))).to(dev)
    go = torch.from_numpy(synth.normal("pc/go", (B, co, H, W))).to(dev)
    gw = torch.zeros((co, ci, 3, 3), device=dev)
    call = lambda: _C.check(L.cp_conv3x3_mfma_wgrad(P(x), P(go), P(gw), B, ci, H, W, co, _C.stream()), "wgrad")
    call()
    ref = torch.nn.grad.conv2d_weight(x.double(), (co, ci, 3, 3), go.double(), padding=1)
    err = (gw.double() - ref).abs().max().item() / ref.abs().max().item()
    lib = torch.nn.grad.conv2d_weight(x, (co, ci, 3, 3), go, padding=1)
    err_lib = (lib.double() - ref).abs().max().item() / ref.abs().max().item()
    t = timeit(call)
    tl = timeit(lambda: torch.nn.grad.conv2d_weight(x, (co, ci, 3, 3), go, padding=1))
    fl = 2.0 * B * ci * co * 9 * H * W
    print("wgrad B%d %3d->%3d %3dx%3d  mfma %.3f ms (%.0f TF/s)  library %.3f ms (%.0f TF/s)  err %.2e (library %.2e)"
          % (B, ci, co, H, W, t, fl / t / 1e9, tl, fl / tl / 1e9, err, err_lib), flush=True)


if os.environ.get("PROBE_WGRAD", "1") == "1":
    for shape in [(1, 32, 16, 8, 32), (2, 64, 27, 24, 80), (4, 64, 256, 256, 512), (4, 64, 27, 256, 512),
                  (4, 64, 64, 128, 256), (4, 128, 128, 64, 128), (4, 256, 256, 32, 64), (4, 512, 512, 16, 32),
                  (8, 64, 256, 96, 320)]:
        run_wgrad(*shape)

for shape in [(1, 32, 16, 8, 32), (2, 64, 27, 24, 80), (4, 64, 256, 256, 512), (1, 64, 256, 256, 512), (4, 64, 27, 256, 512), (1, 128, 128, 128, 256), (1, 256, 256, 64, 128), (1, 512, 512, 32, 64), (1, 64, 27, 256, 512), (1, 128, 27, 128, 256), (1, 256, 27, 64, 128), (1, 512, 27, 32, 64),
              (4, 64, 64, 128, 256), (4, 128, 128, 64, 128), (4, 256, 256, 32, 64), (4, 512, 512, 16, 32),
              (1, 64, 1024, 256, 512)]:
    run(*shape)
