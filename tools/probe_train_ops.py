"""GPU probe: which Python frames launch the at::sum (reduce_kernel) / copy / add kernels of a training step."""
import contextlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".miopen_cache"))
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.environ["MIOPEN_CUSTOM_CACHE_DIR"])
import torch
import bench

dev = torch.device("cuda")
args = bench.parse()
from centerpoly_amd import synth
from centerpoly_amd.opts import opts
from centerpoly_amd.trains.train_factory import train_factory
with contextlib.redirect_stdout(sys.stderr):
    opt = opts().init(["polydet", "--arch", "dla_34", "--poly_loss", "l1+iou", "--nbr_points", "16", "--batch_size", "4"])
opt.device = dev
model, _ = bench.build_model(dev, train=True)
trainer = train_factory["polydet"](opt, model, torch.optim.Adam(model.parameters(), opt.lr))
trainer.set_device(opt.gpus, opt.chunk_sizes, dev)
nb = synth.train_batch(4, 256, 512, nbr_points=16, rep="cartesian", stream="probe/train", in_h=1024, in_w=2048)
batch = {k: torch.from_numpy(v).to(dev) for k, v in nb.items()}
for _ in range(2):
    trainer.step(batch, train=True)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    trainer.step(batch, train=True)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::sum", "aten::copy_", "aten::add_", "aten::add", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::cat", "aten::mul", "aten::threshold_backward"):
        rows.append((e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total, e.key, e.count, str(e.input_shapes)[:110]))
for r in sorted(rows, reverse=True)[:40]:
    print("%9.1f us  %-22s x%-4d %s" % r)
