"""GPU probe: the heads' shared 3x3 convolution (64 -> 4x256 @256x512) as one, two or four
library convolutions, and the 1x1 output convolutions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".miopen_cache"))
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.environ["MIOPEN_CUSTOM_CACHE_DIR"])
import torch
import torch.nn.functional as F

dev = "cuda"


def bench(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


x = torch.randn(1, 64, 256, 512, device=dev)
for parts in (1, 2, 4, 8):
    co = 1024 // parts
    ws = [torch.randn(co, 64, 3, 3, device=dev) for _ in range(parts)]
    t = bench(lambda: [F.conv2d(x, w, None, 1, 1) for w in ws])
    print("3x3 64->1024 as %d conv(s) of %4d: %.3f ms" % (parts, co, t), flush=True)
y = torch.randn(1, 1024, 256, 512, device=dev)
for name, co in (("hm", 8), ("poly", 32), ("depth", 1), ("reg", 2)):
    w = torch.randn(co, 256, 1, 1, device=dev)
    t = bench(lambda: F.conv2d(y[:, :256], w))
    print("1x1 256->%2d on a channel slice: %.3f ms" % (co, t), flush=True)
wall = torch.zeros(43, 1024, 1, 1, device=dev)
t = bench(lambda: F.conv2d(y, wall))
print("1x1 1024->43 block-diagonal as one conv: %.3f ms" % t, flush=True)
