"""GPU probe: level0 (3x3, 16 -> 16) and level1 (3x3 stride 2, 16 -> 32) of the DLA base at 1 x 1024 x 2048 on the
exact-f32 direct kernel and on its split-bf16 form (cp_conv_direct_forward_ex; round 4: stride 1 105 vs 122 us,
stride 2 93 vs 70 us with the form since removed from the dispatch -- both modes now print the exact kernel there)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerpoly_amd import _C
L = _C.lib()
dev = "cuda"
for (ci, co, stride) in [(16, 16, 1), (16, 32, 2)]:
    x = torch.randn(1, ci, 1024, 2048, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.08
    b = torch.randn(co, device=dev)
    Ho, Wo = (1024 - 1) // stride + 1, (2048 - 1) // stride + 1
    out = torch.empty(1, co, Ho, Wo, device=dev)
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=1))
    for mode in (0, 1):
        def call():
            _C.check(L.cp_conv_direct_forward_ex(_C.ptr(x), _C.ptr(w), _C.ptr(b), _C.ptr(out), 1, ci, 1024, 2048, co, 3, stride,
                                                 1, 1, mode, _C.stream()), "fwd")
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            call()
        e1.record(); torch.cuda.synchronize()
        err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
        print("%d->%d stride %d  %s: %.1f us, error %.1e of the max-norm" % (ci, co, stride, "split-bf16" if mode else "exact f32 ",
                                                                           e0.elapsed_time(e1) / 30 * 1e3, err), flush=True)
