"""Loss trajectory of a short DLA-34 + DCNv2 training run on synthetic data (GPU): prints one JSON line with the
per-step loss terms.  Run twice -- default arithmetic and `exact_f32` as second argument (library convolutions,
exact-f32 DCN forward / backward) -- to see that the split-bf16 kernels train the same model (tests/test_conv_mfma.py)."""
import contextlib, io, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from centerpoly_amd.datasets.dataset_factory import get_dataset
from centerpoly_amd.models.model import create_model
from centerpoly_amd.opts import opts
from centerpoly_amd.trains.train_factory import train_factory

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
with contextlib.redirect_stdout(io.StringIO()):
    opt = opts().init(["polydet", "--arch", "dla_34", "--device_targets", "--input_h", "256", "--input_w", "512",
                       "--batch_size", "2", "--num_iters", str(steps), "--poly_loss", "l1+iou", "--lr", "2.5e-4",
                       "--arithmetic", sys.argv[2] if len(sys.argv) > 2 else "split_bf16"])
    Dataset = get_dataset("synthetic", opt.task)
    opt = opts().update_dataset_info_and_set_heads(opt, Dataset)
    ds = Dataset(opt, "train")
opt.device = torch.device("cuda")
torch.manual_seed(317)
model = create_model(opt.arch, opt.heads, opt.head_conv)
trainer = train_factory["polydet"](opt, model, torch.optim.Adam(model.parameters(), opt.lr))
trainer.set_device(opt.gpus, opt.chunk_sizes, opt.device)
loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=0)
traj = []
trainer.model_with_loss.train(True)
for i, batch in enumerate(loader):
    if i >= steps:
        break
    for k in batch:
        if k != "meta":
            batch[k] = batch[k].to(device=opt.device)
    batch = trainer.prepare_batch(batch)
    _, _, stats = trainer.step(batch, True)
    traj.append({k: float(v.detach().mean()) for k, v in stats.items()})
print(json.dumps({"steps": len(traj), "trajectory": traj}))
