"""GPU probe: where does a training step spend its time (and does MIOpen stall on first use)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def log(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)

H, W, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda")
from centerpoly_amd import synth
from centerpoly_amd.opts import opts
from centerpoly_amd.trains.train_factory import train_factory
opt = opts().init(["polydet", "--arch", "dla_34", "--poly_loss", "l1+iou", "--batch_size", str(B)])
opt.device = dev
model, _ = bench.build_model(dev, train=True)
optim = torch.optim.Adam(model.parameters(), opt.lr)
tr = train_factory["polydet"](opt, model, optim)
tr.set_device(opt.gpus, opt.chunk_sizes, dev)
nb = synth.train_batch(B, H // 4, W // 4, in_h=H, in_w=W, stream="probe/train")
batch = {k: torch.from_numpy(v).to(dev) for k, v in nb.items()}
log("batch ready", H, W, B)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = tr.model_with_loss.model(batch["input"])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss, stats = tr.model_with_loss.loss(out, batch)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    optim.zero_grad(set_to_none=True)
    loss.backward()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    optim.step()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    log("iter %d fwd %.1f ms loss %.1f ms bwd %.1f ms opt %.1f ms | loss %.4f" % (it, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3), loss.item()))
if len(sys.argv) > 4:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        tr.step(batch, train=True); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30), flush=True)
