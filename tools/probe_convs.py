"""GPU probe (not a test): how fast are the library convs vs our DCN on the DLA-34 shapes?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

def bench(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

dev = "cuda"
shapes = [  # Cin, Cout, k, stride, H, W (input)
    (64, 256, 3, 1, 256, 512), (128, 128, 3, 1, 128, 256), (256, 256, 3, 1, 64, 128),
    (64, 64, 3, 1, 256, 512), (512, 512, 3, 1, 32, 64), (3, 16, 7, 1, 1024, 2048),
    (16, 16, 3, 1, 1024, 2048), (16, 32, 3, 2, 1024, 2048), (64, 27, 3, 1, 256, 512),
    (256, 8, 1, 1, 256, 512),
]
print("torch", torch.__version__, torch.cuda.get_device_name(0))
for (ci, co, k, s, H, W) in (shapes if "--convs" in sys.argv else []):
    gf = 2 * ci * co * k * k * (H // s) * (W // s) / 1e9
    row = "%4d->%4d k%d s%d @%4dx%4d %7.2f GF |" % (ci, co, k, s, H, W, gf)
    for dt, cl in ((torch.float32, False), (torch.float32, True), (torch.bfloat16, False), (torch.bfloat16, True), (torch.float16, True)):
        x = torch.randn(1, ci, H, W, device=dev, dtype=dt)
        w = torch.randn(co, ci, k, k, device=dev, dtype=dt)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last); w = w.contiguous(memory_format=torch.channels_last)
        try:
            t = bench(lambda: F.conv2d(x, w, None, s, k // 2))
            row += " %s%s %6.3f ms %6.1f TF |" % (str(dt)[6:10], "cl" if cl else "  ", t, gf / t)
        except Exception as ex:
            row += " %s fail |" % str(dt)[6:10]
    print(row, flush=True)

from centerpoly_amd.models.networks.DCNv2.dcn_v2 import dcn_v2_forward_raw
for (ci, co, H, W) in [(64, 64, 256, 512), (128, 64, 128, 256), (128, 128, 128, 256), (256, 128, 64, 128), (256, 256, 64, 128), (512, 256, 32, 64), (256, 64, 64, 128)]:
    x = torch.randn(1, ci, H, W, device=dev); om = torch.randn(1, 27, H, W, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev); b = torch.randn(co, device=dev)
    t = bench(lambda: dcn_v2_forward_raw(x, om, w, b))
    gf = 2 * 9 * ci * co * H * W / 1e9
    mb = 4 * ((ci + 27 + co) * H * W + 9 * ci * co + co) / 1e6
    print("DCN %3d->%3d @%3dx%3d  %7.3f ms  %6.1f TF  %6.1f GB/s (alg %.1f MB)" % (ci, co, H, W, t, gf / t, mb / t, mb), flush=True)
