"""Secondary BASELINE configs on one MI355X (not bench.py lines: reported in DESIGN.md 5):
  config 4  Hourglass-104, 1x3x1024x2048, 24-vertex polar head, inference + decode
  config 5  DLA-34 training at the KITTI shape 3x384x1280, 8 img/GPU, 32-vertex cartesian,
            poly_loss l1 + order loss (one rank's share of the 4-GPU job)
Prints one JSON line per config."""
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(ROOT, ".miopen_cache"))
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, ".miopen_cache"))

import torch  # noqa: E402

from centerpoly_amd import synth  # noqa: E402
from centerpoly_amd.models.decode import polydet_decode  # noqa: E402
from centerpoly_amd.models.model import create_model  # noqa: E402
from centerpoly_amd.opts import opts  # noqa: E402
from centerpoly_amd.trains.train_factory import train_factory  # noqa: E402

dev = torch.device("cuda")


def note(m):
    print("[bench_configs] " + m, file=sys.stderr, flush=True)


def fill(model):
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_by_name(shapes).items()}
    model.load_state_dict(sd)


def timed(fn, steps, warmup):
    for i in range(warmup):
        fn()
        torch.cuda.synchronize()
        note("warmup %d/%d" % (i + 1, warmup))
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def config4():
    heads = {"hm": 8, "poly": 48, "pseudo_depth": 1, "reg": 2}
    model = create_model("hourglass", heads, 256)
    fill(model)
    model = model.to(dev).eval()
    x = torch.from_numpy(synth.normal("cfg4/input", (1, 3, 1024, 2048))).to(dev)

    def step():
        with torch.no_grad():
            out = model(x)[-1]
            return polydet_decode(out["hm"].sigmoid_(), out["poly"], out["pseudo_depth"], reg=out["reg"],
                                  K=128, rep="polar")
    t = timed(step, 10, 3)
    return {"config": "4: Hourglass-104 1x3x1024x2048, 24-vertex polar, inference + decode",
            "img_per_s": 1.0 / t, "ms_per_step": 1e3 * t}


def config5():
    with contextlib.redirect_stdout(sys.stderr):
        opt = opts().init(["polydet", "--arch", "dla_34", "--poly_loss", "l1", "--poly_order",
                           "--nbr_points", "32", "--batch_size", "8", "--input_h", "384", "--input_w", "1280"])
    opt.device = dev
    model = create_model("dla_34", opt.heads, 256)
    fill(model)
    model = model.to(dev).train()
    trainer = train_factory["polydet"](opt, model, torch.optim.Adam(model.parameters(), opt.lr))
    trainer.set_device(opt.gpus, opt.chunk_sizes, dev)
    nb = synth.train_batch(8, 96, 320, nbr_points=32, rep="cartesian", stream="cfg5", in_h=384, in_w=1280)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in nb.items()}
    t = timed(lambda: trainer.step(batch, train=True), 6, 3)
    return {"config": "5: DLA-34 training 8x3x384x1280 per GPU, 32-vertex cartesian, l1 + order loss, Adam",
            "img_per_s": 8.0 / t, "ms_per_step": 1e3 * t}


if __name__ == "__main__":
    which = sys.argv[1:] or ["4", "5"]
    for w in which:
        print(json.dumps({"4": config4, "5": config5}[w]()), flush=True)
        torch.cuda.empty_cache()
