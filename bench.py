#!/usr/bin/env python3
"""Benchmark of the polydet hot path on MI355X (contract: see the task description).

  N = 1 (default)  one step = one 2048x1024 image through DLA-34 + DCNv2 -> sigmoid -> fused
                   NMS/top-k/decode (BASELINE config 2); `value` = inference img/s.  The same
                   line carries the 1-GPU training point (`train`), the DCNv2 forward roofline
                   measured with HIP events inside the timed region (`roofline` against the
                   fp32 matrix pipe that bounds it, `roofline_hbm` for the same launch against
                   HBM) and the CPU oracle timed on a bounded sample
                   (`cpu_baseline`).
  N > 1            one step = one data-parallel training step (BASELINE config 3: DLA-34 + DCNv2,
                   4 images of 2048x1024 per GPU, 16-vertex cartesian head, l1+iou polygon loss,
                   Adam): forward, losses, backward with bucketed RCCL all-reduce, optimizer.
                   `value` = whole-job training img/s, weak scaling.

Launch for N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N
                   --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MIOpen compiles its convolution kernels on first use (minutes on a fresh box).  Keep its
# compiled-kernel cache inside the repo so it travels with the tree like our own .so.
_MIOPEN_CACHE = os.path.join(ROOT, ".miopen_cache")
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", _MIOPEN_CACHE)
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(_MIOPEN_CACHE, "db"))
os.makedirs(os.environ["MIOPEN_USER_DB_PATH"], exist_ok=True)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TF = 157.3       # dense fp32 matrix peak
HEADS = {"hm": 8, "poly": 32, "pseudo_depth": 1, "reg": 2}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--mode", default="auto", choices=["auto", "infer", "train"])
    p.add_argument("--height", type=int, default=1024)
    p.add_argument("--width", type=int, default=2048)
    p.add_argument("--train_batch", type=int, default=4, help="images per GPU in training")
    p.add_argument("--train_steps", type=int, default=4, help="steps of the N=1 training point")
    p.add_argument("--dcn_contraction", default="f32", choices=["f32", "bf16x3"],
                   help="DCNv2 forward contraction at inference: exact fp32 MFMA or split-bf16 x3")
    p.add_argument("--no_cpu_baseline", action="store_true")
    p.add_argument("--no_detector_point", action="store_true",
                   help="skip the end-to-end PolydetDetector.run point of the N=1 line")
    p.add_argument("--no_train_point", action="store_true")
    return p.parse_args()


def build_model(dev, train, dcn_contraction="f32"):
    from centerpoly_amd import synth
    from centerpoly_amd.models.model import create_model
    model = create_model("dla_34", dict(HEADS), 256)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    w = synth.fill_by_name(shapes)                      # random-init weights by name (no files)
    for k in w:
        if "conv_offset_mask" in k:
            w[k] = (w[k] * 0.5).astype(w[k].dtype)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    model.load_state_dict(sd)
    model = model.to(dev)
    model.train(train)
    if not train and dev.type == "cuda":
        model.prepare_inference(dcn_contraction=dcn_contraction)
    return model, sd


def note(msg):
    """Progress on stderr (the JSON line on stdout stays alone); MIOpen compiles kernels on
    first use, so the first step of each leg can take a minute on a fresh box."""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %s] %s" % (time.strftime("%H:%M:%S"), msg), file=sys.stderr, flush=True)


def timed(fn, steps, warmup, world, tag=""):
    for i in range(warmup):
        fn()
        torch.cuda.synchronize()
        note("%s warmup %d/%d done" % (tag, i + 1, warmup))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([t], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = float(tt.item())
    return t


def infer_leg(args, dev, world):
    from centerpoly_amd import _C, synth
    from centerpoly_amd.models.decode import polydet_decode
    model, _ = build_model(dev, train=False, dcn_contraction=args.dcn_contraction)
    x = torch.from_numpy(synth.normal("bench/input", (1, 3, args.height, args.width))).to(dev)

    def step():
        with torch.no_grad():
            out = model(x)[-1]
            hm = out["hm"].sigmoid_()
            return polydet_decode(hm, out["poly"], out["pseudo_depth"], reg=out["reg"], K=128,
                                  rep="cartesian")

    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        note("infer warmup %d/%d done" % (i + 1, args.warmup))
    _C.kernel_timer = _C.KernelTimer()                 # HIP events around every DCN launch
    t = timed(step, args.steps, 0, world, "infer")
    note("infer timed region done")
    summary = _C.kernel_timer.summary()
    _C.kernel_timer = None
    return t, summary


def dcn_roofline(summary):
    """Dominant DCN launch: the layer shape with the largest total time in the timed region."""
    if not summary:
        return None, None
    key = max(summary, key=lambda k: summary[k]["avg_ms"] * summary[k]["launches"])
    _, cin, cout, h, w, nb = key
    avg_s = summary[key]["avg_ms"] * 1e-3
    alg_bytes = 4.0 * (nb * (cin + 27 + cout) * h * w + 9 * cin * cout + cout)
    alg_flops = 2.0 * 9 * cin * cout * h * w * nb
    traffic = None
    prof = os.path.join(ROOT, "profiles", "dcn_fwd_pmc.json")
    if os.path.exists(prof):
        try:
            traffic = json.load(open(prof)).get("%dx%dx%dx%d" % (cin, cout, h, w))
        except Exception:
            traffic = None
    if nb != 1:
        traffic = None                      # the PMC passes were taken on the single-image launch
    layer = "dcn_v2_forward %d->%d @%dx%d" % (cin, cout, h, w) + (" x%d images" % nb if nb != 1 else "")
    hbm = {"bound": "hbm", "achieved": alg_bytes / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": alg_bytes / avg_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "kernel": layer,
           "avg_launch_us": avg_s * 1e6, "launches": summary[key]["launches"],
           "algorithmic_bytes_per_launch": alg_bytes}
    mfma = {"bound": "mfma", "achieved": alg_flops / avg_s / 1e12, "peak": MFMA_F32_PEAK_TF,
            "unit": "TFLOP/s", "frac": alg_flops / avg_s / 1e12 / MFMA_F32_PEAK_TF, "traffic": traffic,
            "kernel": layer, "algorithmic_flops_per_launch": alg_flops,
            "avg_launch_us": avg_s * 1e6, "launches": summary[key]["launches"]}
    return hbm, mfma


def train_leg(args, dev, world, rank, steps, warmup):
    from centerpoly_amd import synth
    from centerpoly_amd.opts import opts
    from centerpoly_amd.trains.train_factory import train_factory
    with contextlib.redirect_stdout(sys.stderr):          # stdout carries the JSON line only
        opt = opts().init(["polydet", "--arch", "dla_34", "--poly_loss", "l1+iou", "--nbr_points", "16",
                           "--batch_size", str(args.train_batch * world)])
    opt.device = dev
    model, _ = build_model(dev, train=True)
    optimizer = torch.optim.Adam(model.parameters(), opt.lr)
    trainer = train_factory["polydet"](opt, model, optimizer)
    trainer.set_device(opt.gpus, opt.chunk_sizes, dev)
    B = args.train_batch
    nb = synth.train_batch(B, args.height // 4, args.width // 4, nbr_points=16, rep="cartesian",
                           stream="bench/train/rank%d" % rank, in_h=args.height, in_w=args.width)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in nb.items()}    # resident before timing

    def step():
        trainer.step(batch, train=True)

    t = timed(step, steps, warmup, world, "train")
    note("train timed region done")
    # the DCNv2 forward launches of two more (untimed) steps, HIP-event timed on rank 0
    from centerpoly_amd import _C
    if rank == 0:
        _C.kernel_timer = _C.KernelTimer()
    for _ in range(2):
        step()
    summary = _C.kernel_timer.summary() if rank == 0 else None
    _C.kernel_timer = None
    del trainer, model, optimizer, batch
    torch.cuda.empty_cache()
    return t, summary


def detector_leg(args, dev):
    """End-to-end PolydetDetector.run on a host uint8 image (upload over PCIe, device warp +
    normalise, network, decode, device affine post-process, copy back, per-class dicts)."""
    from centerpoly_amd import synth
    from centerpoly_amd.detectors.detector_factory import detector_factory
    from centerpoly_amd.opts import opts
    import numpy as np
    with contextlib.redirect_stdout(sys.stderr):
        opt = opts().init(["polydet", "--arch", "dla_34", "--input_h", str(args.height),
                           "--input_w", str(args.width)])
        det = detector_factory["polydet"](opt)
    img = (synth.uniform("bench/detector/img", (args.height, args.width, 3)) * 255).astype(np.uint8)
    for _ in range(3):
        det.run(img)
    n = 20
    keys = ["pre", "net", "dec", "post", "merge"]
    acc = dict.fromkeys(keys, 0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ret = det.run(img)
        for k in keys:
            acc[k] += ret[k]
    t = time.perf_counter() - t0
    del det
    torch.cuda.empty_cache()
    return {"metric": "PolydetDetector.run img/s (host uint8 image in, result dicts out, PCIe included)",
            "value": n / t, "ms_per_image": 1e3 * t / n,
            "stage_ms": {k: round(1e3 * v / n, 3) for k, v in acc.items()},
            "workload": "%dx%d uint8 image, keep_res (network input %dx%d), K=%d"
                        % (args.width, args.height, (args.width | 31) + 1, (args.height | 31) + 1, opt.K)}


def cpu_baseline(args):
    """The CPU oracle (a port of the reference's path) on a bounded sample: ONE full-size
    image through DLA-34 + DCNv2 + decode (a few seconds on a 16-core host share)."""
    from centerpoly_amd import synth
    from oracle import decode as odec
    from oracle import nets as onet
    # a one-GPU box owns a 16-core share of the host: more threads only oversubscribe it
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    note("cpu baseline (oracle on %d threads) ..." % threads)
    h, w = args.height, args.width
    _, sd = build_model(torch.device("cpu"), train=False)
    x = torch.from_numpy(synth.normal("bench/cpu/input", (1, 3, h, w)))
    t0 = time.perf_counter()
    with torch.no_grad():
        out = onet.dla_seg_forward(sd, x, dict(HEADS))[0]
        odec.polydet_decode(torch.sigmoid(out["hm"]), out["poly"], out["pseudo_depth"], out["reg"], K=128)
    t = time.perf_counter() - t0
    return {"value": 1.0 / t, "unit": "img/s", "cores": threads, "kind": "port",
            "sample": "1 image at %dx%d through the oracle's DLA-34+DCNv2 forward + sigmoid + "
                      "decode, %.1f s wall" % (w, h, t)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    mode = args.mode if args.mode != "auto" else ("infer" if world == 1 else "train")
    torch.backends.cudnn.benchmark = os.environ.get("CP_MIOPEN_BENCHMARK", "0") == "1"
    line = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (counter-hash inputs, name-hashed random-init weights)"}
    if mode == "infer":
        t, summary = infer_leg(args, dev, world)
        hbm, mfma = dcn_roofline(summary)
        line.update({
            "metric": "inference img/s @2048x1024 DLA-34 (1 GPU)", "unit": "img/s",
            "value": world * args.steps / t, "ms_per_step": 1e3 * t / args.steps,
            "config": {"workload": "BASELINE config 2: DLA-34 + DCNv2, 1x3x%dx%d synthetic, 16-vertex "
                                   "cartesian head, K=128, forward + sigmoid + NMS/top-k/decode"
                                   % (args.height, args.width),
                       "images_per_step": world, "parallelism": "replicas" if world > 1 else "single"},
            # the dominant kernel's arithmetic intensity (119 FLOP/B) is 6x the fp32 ridge point
            # (157.3 TFLOP/s / 8 TB/s = 19.7 FLOP/B): the fp32 matrix pipe is the binding roofline,
            # the HBM fraction of the same launch is reported beside it
            "roofline": mfma, "roofline_hbm": hbm,
            "dcn_layers_ms": {"%d->%d@%dx%d" % k[1:5]: round(v["avg_ms"], 4) for k, v in summary.items()},
        })
        if world == 1 and not args.no_detector_point:
            line["detector_end_to_end"] = detector_leg(args, dev)
            note("detector end-to-end point done")
        if world == 1 and not args.no_train_point:
            tt, _ = train_leg(args, dev, 1, 0, args.train_steps, 2)
            # same workload per GPU as the N > 1 lines: the 1-GPU point of the training scaling curve
            line["train"] = {"metric": "train img/s 1/2/4/8 GPU @2048x1024 DLA-34",
                             "value": args.train_batch * args.train_steps / tt,
                             "ms_per_step": 1e3 * tt / args.train_steps, "n_gpus": 1,
                             "global_batch": args.train_batch, "steps": args.train_steps,
                             "workload": "BASELINE config 3 per-GPU share: DLA-34 + DCNv2, %d x 3x%dx%d, "
                                         "l1+iou polygon loss, Adam" % (args.train_batch, args.height, args.width)}
    else:
        t, summary = train_leg(args, dev, world, rank, args.steps, args.warmup)
        hbm, mfma = dcn_roofline(summary) if rank == 0 else (None, None)
        line.update({
            "metric": "train img/s 1/2/4/8 GPU @2048x1024 DLA-34", "unit": "img/s",
            "value": world * args.train_batch * args.steps / t, "ms_per_step": 1e3 * t / args.steps,
            "config": {"workload": "BASELINE config 3: DLA-34 + DCNv2 training, %d x 3x%dx%d per GPU, "
                                   "16-vertex cartesian + l1+iou polygon loss, Adam lr 4e-6"
                                   % (args.train_batch, args.height, args.width),
                       "global_batch": world * args.train_batch,
                       "parallelism": "dp%d (one process per GPU, RCCL all-reduce)" % world},
            # dominant DCNv2 FORWARD launch of the training step (fp32 matrix pipe is its bound);
            # the backward kernels are priced in DESIGN.md 4.2
            "roofline": mfma, "roofline_hbm": hbm,
            "scaling_base": "weak scaling of the training leg: compare with the N=1 line's "
                            "train.value (same %d img/GPU workload), not with its inference value"
                            % args.train_batch,
        })
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
