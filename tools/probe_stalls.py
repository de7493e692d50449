"""GPU probe: which host<->device interaction produces the periodic ~80 ms stalls seen in
PolydetDetector.run?  Times 40 iterations of: kernels + sync only / + pinned H2D / + small D2H."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".miopen_cache"))
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.environ["MIOPEN_CUSTOM_CACHE_DIR"])
import numpy as np
import torch
import bench

dev = torch.device("cuda")
args = type("A", (), {"dcn_contraction": "f32"})()
model, _ = bench.build_model(dev, train=False)
x = torch.randn(1, 3, 1024, 2048, device=dev)
img = (np.random.rand(1024, 2048, 3) * 255).astype(np.uint8)
pin = torch.empty(img.shape, dtype=torch.uint8).pin_memory()
small = torch.randn(128, 39, device=dev)


def run(name, h2d, d2h, n=40):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        if h2d == "pinned":
            pin.copy_(torch.from_numpy(img))
            d = pin.to(dev, non_blocking=True)
        elif h2d == "pageable":
            d = torch.from_numpy(img).to(dev)
        with torch.no_grad():
            out = model(x)[-1]
        torch.cuda.synchronize()
        if d2h:
            small.cpu()
        ts.append(1e3 * (time.perf_counter() - t0))
    ts = np.array(ts[3:])
    print("%-28s median %.2f ms  mean %.2f ms  max %.2f ms  >20ms: %d/%d" % (name, np.median(ts), ts.mean(), ts.max(), (ts > 20).sum(), len(ts)), flush=True)


big = torch.empty((4, 3, 1024, 2048), dtype=torch.float32).pin_memory()
big_pg = torch.empty((4, 3, 1024, 2048), dtype=torch.float32)


def run_big(name, mode, n=30):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        if mode == "pinned_nb":
            d = big.to(dev, non_blocking=True)
        elif mode == "pinned_block":
            d = big.to(dev)
        elif mode == "pageable":
            d = big_pg.to(dev)
        with torch.no_grad():
            for _ in range(4):
                out = model(x)[-1]
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    ts = np.array(ts[3:])
    print("%-34s median %.2f ms  mean %.2f ms  max %.2f ms  >1.5x median: %d/%d" % (name, np.median(ts), ts.mean(), ts.max(), (ts > 1.5 * np.median(ts)).sum(), len(ts)), flush=True)


run_big("4 fwd, no copy", None)
run_big("4 fwd + 100 MB pinned non_blocking", "pinned_nb")
run_big("4 fwd + 100 MB pinned blocking", "pinned_block")
run_big("4 fwd + 100 MB pageable", "pageable")
run("kernels + sync", None, False)
run("+ small D2H", None, True)
run("+ pinned H2D", "pinned", False)
run("+ pageable H2D", "pageable", False)
run("+ pinned H2D + D2H", "pinned", True)
run("kernels + sync (again)", None, False)
