#!/usr/bin/env python3
"""Benchmark of the polydet hot path on MI355X (contract: see the task description).

  N = 1 (default)  one step = one 2048x1024 image through DLA-34 + DCNv2 -> sigmoid -> fused
                   NMS/top-k/decode (BASELINE config 2); `value` = inference img/s.  The same
                   line carries the 1-GPU training point (`train`, with the DCNv2 backward
                   rooflines), the DCNv2 forward roofline measured with HIP events inside the timed
                   region (`roofline`: algorithmic bytes against HBM, the north star's figure;
                   `roofline_mfma`: the same launch's algorithmic flops against the matrix pipe), the
                   same launch on other offset fields (`roofline_by_offsets`), BASELINE configs 4 and
                   5 (`other_configs`) and the CPU oracle timed on a bounded sample (`cpu_baseline`).
                   stdout carries ONE compact line (< 8 KB: `compact_line`); the full tables go to
                   stderr and gpurun_out/bench_detail.json (`write_detail`).
  N > 1            one step = one data-parallel training step (BASELINE config 3: DLA-34 + DCNv2,
                   4 images of 2048x1024 per GPU, 16-vertex cartesian head, l1+iou polygon loss,
                   Adam): forward, losses, backward with bucketed RCCL all-reduce, optimizer.
                   `value` = whole-job training img/s, weak scaling.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks
itself (a child `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 --master-port P bench.py --gpus N ...`, started BEFORE this process touches the GPU) and
relays rank 0's JSON line; it refuses (exit code 2) when fewer than N devices are visible or when
WORLD_SIZE disagrees with --gpus.  Under torch.distributed.run it is one rank of the job.
"""
import argparse
import contextlib
import hashlib
import json
import os
import socket
import statistics
import subprocess
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

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TF = 157.3       # dense fp32 matrix peak
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16 matrix peak (no sparsity)
HEADS = {"hm": 8, "poly": 32, "pseudo_depth": 1, "reg": 2}
EXIT_REFUSED = 2


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--mode", default="auto", choices=["auto", "infer", "train"])
    p.add_argument("--config", default="auto", choices=["auto", "2", "3", "4", "5"],
                   help="BASELINE config timed as `value`: auto = 2 at N=1, 3 at N>1; 4 = Hourglass-104 "
                        "inference, 24-vertex polar; 5 = DLA-34 training at the KITTI shape, 8 img/GPU")
    p.add_argument("--height", type=int, default=1024)
    p.add_argument("--width", type=int, default=2048)
    p.add_argument("--train_batch", type=int, default=4, help="images per GPU in training")
    p.add_argument("--train_steps", type=int, default=8, help="steps of the N=1 training point")
    p.add_argument("--dcn_contraction", default="auto", choices=["auto", "f32", "bf16x3"],
                   help="DCNv2 forward contraction at inference: exact fp32 MFMA, split-bf16 x3, or auto (split-bf16 "
                        "wherever the LDS-region kernel runs and for the layers with > 64 output channels)")
    p.add_argument("--arithmetic", default="split_bf16", choices=["split_bf16", "exact_f32"],
                   help="contraction arithmetic of the whole step (centerpoly_amd/arithmetic.py); the line always "
                        "carries the other one as `exact_f32` / says so in `dtype`")
    p.add_argument("--no_exact_point", action="store_true",
                   help="skip the exact-fp32 re-run of the step (`exact_f32`) and the golden-error probe")
    p.add_argument("--graph", action="store_true",
                   help="time one HIP-graph replay per inference step instead of eager launches (measured: no "
                        "faster, the eager step is GPU-bound)")
    p.add_argument("--no_cpu_baseline", action="store_true")
    p.add_argument("--no_detector_point", action="store_true",
                   help="skip the end-to-end PolydetDetector.run point of the N=1 line")
    p.add_argument("--no_train_point", action="store_true")
    p.add_argument("--no_offset_points", action="store_true",
                   help="skip the dominant DCNv2 launch on other offset fields")
    p.add_argument("--no_other_configs", action="store_true",
                   help="skip the BASELINE config 4 / 5 points of the N=1 line")
    return p.parse_args(argv)


# ---------------------------------------------------------------- N-rank launch ---
def launch_plan(args, environ, argv):
    """What this process has to do, decided before anything touches the GPU:
    ("run", None) -- be one rank (or the single process);  ("spawn", cmd) -- start the N ranks as
    a child job and relay rank 0's line;  ("refuse", message) -- exit with EXIT_REFUSED."""
    world_env = environ.get("WORLD_SIZE")
    if args.gpus < 1:
        return "refuse", "--gpus must be >= 1 (got %d)" % args.gpus
    if world_env is not None:
        if int(world_env) != args.gpus:
            return "refuse", ("--gpus %d disagrees with WORLD_SIZE=%s: launch with --nproc-per-node %d"
                              % (args.gpus, world_env, args.gpus))
        return "run", None
    if args.gpus == 1:
        return "run", None
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return "spawn", cmd


def spawn_ranks(args, cmd):
    import torch
    have = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if have < args.gpus:
        print("bench.py: --gpus %d but only %d HIP device(s) visible; refusing to report a %d-GPU line"
              % (args.gpus, have, args.gpus), file=sys.stderr, flush=True)
        return EXIT_REFUSED
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    line = None
    for raw in proc.stdout.decode("utf-8", "replace").splitlines():
        try:
            cand = json.loads(raw)
        except ValueError:
            continue
        if isinstance(cand, dict) and "metric" in cand:
            line = cand
    if proc.returncode != 0 or line is None:
        print("bench.py: the %d-rank job failed (rc %d, JSON line %s)"
              % (args.gpus, proc.returncode, "missing" if line is None else "present"), file=sys.stderr, flush=True)
        return proc.returncode or 1
    if line.get("n_gpus") != args.gpus:
        print("bench.py: child reported n_gpus=%r, expected %d" % (line.get("n_gpus"), args.gpus),
              file=sys.stderr, flush=True)
        return 1
    sys.stdout.write(json.dumps(line) + "\n")
    sys.stdout.flush()
    return 0


# ------------------------------------------------------------------ helpers ---
def note(msg):
    """Progress on stderr (the JSON line on stdout stays alone); MIOpen compiles kernels on
    first use, so the first step of each leg can take a minute on a fresh box."""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %s] %s" % (time.strftime("%H:%M:%S"), msg), file=sys.stderr, flush=True)


def build_model(dev, train, dcn_contraction="auto", arch="dla_34", heads=None, offset_weight_scale=1.0):
    import torch
    from centerpoly_amd import synth
    from centerpoly_amd.models.model import create_model
    heads = dict(heads or HEADS)
    model = create_model(arch, heads, 256)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    w = synth.fill_by_name(shapes)                      # random-init weights by name (no files)
    for k in w:
        if "conv_offset_mask" in k and offset_weight_scale != 1.0:
            w[k] = (w[k] * offset_weight_scale).astype(w[k].dtype)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    model.load_state_dict(sd)
    model = model.to(dev)
    model.train(train)
    if not train and dev.type == "cuda" and hasattr(model, "prepare_inference"):
        if arch.startswith("dla"):
            model.prepare_inference(dcn_contraction=dcn_contraction)
        else:
            model.prepare_inference()
    return model, sd


def timed(fn, steps, warmup, world, tag=""):
    import torch
    import torch.distributed as dist
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


CONV_SOURCES = ("conv_mfma.hip", "cp_common.h")
HEADS_SOURCES = ("heads_fused.hip", "cp_common.h")


DCN_FWD_SOURCES = ("dcn_fwd.hip", "dcn_fwd_region.hip", "cp_common.h")
DCN_BWD_SOURCES = ("dcn_bwd.hip", "dcn_bwd_data.hip", "dcn_bwd_weight.hip", "cp_common.h")


def kernel_revision(sources=DCN_FWD_SOURCES):
    """Content hash of a kernel's sources (default: the DCNv2 forward kernel): PMC traffic files are only
    trusted for the kernel revision they were taken on."""
    h = hashlib.sha256()
    for f in sources:
        with open(os.path.join(ROOT, "centerpoly_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(cin, cout, h, w, nb, prof_name="dcn_fwd_pmc.json", sources=DCN_FWD_SOURCES):
    """HBM bytes per launch of a kernel family's dominant launch from the committed rocprofv3 PMC
    passes -- only when those passes were taken on bench.py's own tensors at the CURRENT kernel
    revision (tools/pmc_bench_traffic.py writes the files); otherwise null."""
    prof = os.path.join(ROOT, "profiles", prof_name)
    try:
        with open(prof) as fh:
            d = json.load(fh)
    except (OSError, ValueError):
        return None, "no PMC file"
    if d.get("kernel_rev") != kernel_revision(sources):
        return None, "PMC passes are from another kernel revision (%s): not reported" % d.get("kernel_rev")
    if d.get("inputs") != "bench.py infer leg":
        return None, "PMC passes were not taken on the bench inputs"
    v = d.get("layers", {}).get("%dx%dx%dx%dx%d" % (nb, cin, cout, h, w))
    if v is None:
        return None, "launch shape not in the PMC file"
    return float(v), "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command (%s), gfx950 " \
                     "FETCH_SIZE correction applied, kernel revision %s" % (d.get("command", "?"), d["kernel_rev"])


def measured_traffic_bwd(tag, cin, cout, h, w, nb):
    """HBM bytes per launch of a DCNv2 backward family (main kernel + its reduce) from profiles/dcn_bwd_pmc.json
    (tools/pmc_bwd_traffic.py over tools/run_pmc_bwd.sh): only at the CURRENT kernel revision and launch shape."""
    try:
        with open(os.path.join(ROOT, "profiles", "dcn_bwd_pmc.json")) as fh:
            d = json.load(fh)
    except (OSError, ValueError):
        return None, "no PMC file"
    if d.get("kernel_rev") != kernel_revision(DCN_BWD_SOURCES):
        return None, "PMC passes are from another kernel revision (%s): not reported" % d.get("kernel_rev")
    v = d.get("layers", {}).get("%dx%dx%dx%dx%d" % (nb, cin, cout, h, w), {}).get(tag)
    if v is None:
        return None, "launch shape not in the PMC file"
    return float(v), "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over %s, same launch shape (not the bench's own " \
                     "tensors); FETCH_SIZE doubled: an upper bound for these kernels' narrow loads; kernel revision %s" \
                     % (d.get("inputs", "?"), d["kernel_rev"])


def dcn_roofline(summary, tag="dcn_fwd", fwd_contraction="auto"):
    """Dominant launch of one DCNv2 kernel family: the layer shape with the largest total time in the
    timed region.  Algorithmic bytes / FLOPs per launch: SURVEY.md 8(d); the backward kernels read
    the forward's tensors plus grad_out and write the gradients of the same shapes."""
    summary = {k: v for k, v in (summary or {}).items() if k[0] == tag}
    if not summary:
        return None, None
    key = max(summary, key=lambda k: summary[k]["avg_ms"] * summary[k]["launches"])
    _, cin, cout, h, w, nb = key
    avg_s = summary[key]["avg_ms"] * 1e-3
    fwd_bytes = 4.0 * (nb * (cin + 27 + cout) * h * w + 9 * cin * cout + cout)
    gemm_flops = 2.0 * 9 * cin * cout * h * w * nb
    traffic, traffic_src = (None, "not measured for this kernel")
    if tag == "dcn_fwd":
        alg_bytes, alg_flops, name = fwd_bytes, gemm_flops, "dcn_v2_forward"
        traffic, traffic_src = measured_traffic(cin, cout, h, w, nb)
    elif tag == "dcn_bwd_data":     # reads x, offset/mask, weight, grad_out; writes grad_x, grad_offset/mask
        alg_bytes = 4.0 * (nb * (2 * cin + 2 * 27 + cout) * h * w + 9 * cin * cout)
        alg_flops, name = gemm_flops, "dcn_v2_backward (data: grad_x, grad_offset, grad_mask)"
        traffic, traffic_src = measured_traffic_bwd(tag, cin, cout, h, w, nb)
    else:                           # reads x, offset/mask, grad_out; writes grad_weight
        alg_bytes = 4.0 * (nb * (cin + 27 + cout) * h * w + 9 * cin * cout)
        alg_flops, name = gemm_flops, "dcn_v2_backward (weight)"
        traffic, traffic_src = measured_traffic_bwd(tag, cin, cout, h, w, nb)
    layer = "%s %d->%d @%dx%d" % (name, cin, cout, h, w) + (" x%d images" % nb if nb != 1 else "")
    common = {"traffic": traffic, "traffic_source": traffic_src, "kernel": layer, "avg_launch_us": avg_s * 1e6,
              "launches": summary[key]["launches"]}
    hbm = dict(common, bound="hbm", achieved=alg_bytes / avg_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
               frac=alg_bytes / avg_s / 1e9 / HBM_PEAK_GBS, algorithmic_bytes_per_launch=alg_bytes)
    from centerpoly_amd import _C
    from centerpoly_amd.models.networks.DCNv2.dcn_v2 import DCN, auto_contraction
    if tag in ("dcn_bwd_data", "dcn_bwd_weight") and not (DCN.backward_flags & _C.DCN_BWD_EXACT_F32) and w % 4 == 0:
        mfma = split_bf16_roofline(common, alg_flops, avg_s)   # the backward kernels contract in split-bf16 x3
    elif tag == "dcn_fwd" and fwd_contraction != "f32":
        # which kernel the library ran for this shape: 0 gather / exact f32, 1 gather / split-bf16, 2 LDS-region / split-bf16
        shp = _C.DcnShape(nb, cin, h, w, cout, 3, 3, 1, 1, 1, 1)
        con = auto_contraction(shp) if fwd_contraction == "auto" else fwd_contraction
        kid = _C.lib().cp_dcn_v2_forward_kernel(shp, _C.DCN_CONTRACTION[con])
        if kid >= 1:
            mfma = split_bf16_roofline(common, alg_flops, avg_s)
            mfma["arithmetic"] = ("split-bf16 x3 on v_mfma_f32_32x32x16_bf16, fp32 accumulate (LDS-region kernel, "
                                  "dcn_fwd_region.hip)" if kid == 2 else mfma["arithmetic"] + " (gather kernel)")
        else:
            mfma = dict(common, bound="mfma", achieved=alg_flops / avg_s / 1e12, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s",
                        frac=alg_flops / avg_s / 1e12 / MFMA_F32_PEAK_TF, algorithmic_flops_per_launch=alg_flops,
                        arithmetic="exact fp32 MFMA (gather kernel)")
    else:
        mfma = dict(common, bound="mfma", achieved=alg_flops / avg_s / 1e12, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s",
                    frac=alg_flops / avg_s / 1e12 / MFMA_F32_PEAK_TF, algorithmic_flops_per_launch=alg_flops)
    return hbm, mfma


def split_bf16_roofline(common, alg_flops, avg_s):
    """Matrix-pipe roofline of a kernel that forms every fp32 product from three bf16 MFMA products
    (hi*hi + hi*lo + lo*hi).  `achieved` / `frac` are the ALGORITHMIC flops over the launch time against
    the dense bf16 peak; the flops the kernel ISSUES are 3x those and are reported apart as `issued_frac`."""
    issued = 3.0 * alg_flops
    return dict(common, bound="mfma", achieved=alg_flops / avg_s / 1e12, peak=MFMA_BF16_PEAK_TF, unit="TFLOP/s",
                frac=alg_flops / avg_s / 1e12 / MFMA_BF16_PEAK_TF, algorithmic_flops_per_launch=alg_flops,
                issued_bf16_flops_per_launch=issued, issued_frac=issued / avg_s / 1e12 / MFMA_BF16_PEAK_TF,
                arithmetic="split-bf16 x3 on v_mfma_f32_16x16x32_bf16, fp32 accumulate")


def conv_roofline(summary, tag="conv3x3_fwd"):
    """Dominant launch of the split-bf16 3x3 convolution (forward / input gradient, or weight gradient):
    algorithmic flops 2*9*Cin*Cout*H*W*B, bytes = input + output (+ weights) once."""
    summary = {k: v for k, v in (summary or {}).items() if k[0] == tag}
    if not summary:
        return None, None
    key = max(summary, key=lambda k: summary[k]["avg_ms"] * summary[k]["launches"])
    _, cin, cout, h, w, nb = key
    avg_s = summary[key]["avg_ms"] * 1e-3
    alg_flops = 2.0 * 9 * cin * cout * h * w * nb
    alg_bytes = 4.0 * (nb * (cin + cout) * h * w + 9 * cin * cout)
    name = "conv3x3 weight gradient" if tag == "conv3x3_wgrad" else "conv3x3 forward / input gradient"
    traffic, traffic_src = (None, "not measured for this kernel")
    if tag == "conv3x3_fwd":
        traffic, traffic_src = measured_traffic(cin, cout, h, w, nb, "conv_mfma_pmc.json", CONV_SOURCES)
    common = {"traffic": traffic, "traffic_source": traffic_src,
              "kernel": "%s %d->%d @%dx%d" % (name, cin, cout, h, w) + (" x%d images" % nb if nb != 1 else ""),
              "avg_launch_us": avg_s * 1e6, "launches": summary[key]["launches"]}
    hbm = dict(common, bound="hbm", achieved=alg_bytes / avg_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
               frac=alg_bytes / avg_s / 1e9 / HBM_PEAK_GBS, algorithmic_bytes_per_launch=alg_bytes)
    return hbm, split_bf16_roofline(common, alg_flops, avg_s)


def heads_roofline(summary):
    """The fused heads launch (3x3 convolution + bias + ReLU + 1x1 convolution of all heads, csrc/heads_fused.hip):
    algorithmic flops of both stages, bytes = input + the heads' outputs + weights (the head_conv-channel
    intermediate never reaches memory)."""
    keys = [k for k in (summary or {}) if k[0] == "heads_fused"]
    if not keys:
        return None, None
    key = max(keys, key=lambda k: summary[k]["avg_ms"] * summary[k]["launches"])
    _, cin, ctot, h, w, nb, hc, cout_sum = key
    avg_s = summary[key]["avg_ms"] * 1e-3
    alg_flops = 2.0 * (9 * cin * ctot + hc * cout_sum) * h * w * nb
    alg_bytes = 4.0 * (nb * (cin + cout_sum) * h * w + 9 * cin * ctot + hc * cout_sum)
    traffic, traffic_src = measured_traffic(cin, ctot, h, w, nb, "heads_fused_pmc.json", HEADS_SOURCES)
    common = {"traffic": traffic, "traffic_source": traffic_src,
              "kernel": "fused heads: conv3x3 %d->%d + ReLU + conv1x1 ->%d @%dx%d" % (cin, ctot, cout_sum, h, w)
                        + (" x%d images" % nb if nb != 1 else ""),
              "avg_launch_us": avg_s * 1e6, "launches": summary[key]["launches"]}
    hbm = dict(common, bound="hbm", achieved=alg_bytes / avg_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
               frac=alg_bytes / avg_s / 1e9 / HBM_PEAK_GBS, algorithmic_bytes_per_launch=alg_bytes)
    return hbm, split_bf16_roofline(common, alg_flops, avg_s)


def infer_contraction(args):
    """DCNv2 forward contraction of the inference legs: --dcn_contraction, except that the exact-fp32 arithmetic
    means the exact-fp32 kernel."""
    from centerpoly_amd import arithmetic
    return "f32" if arithmetic.current() == "exact_f32" else args.dcn_contraction


def golden_error(dev):
    """Largest error of the DLA-34 head maps against the golden outputs recorded from the reference's own modules
    (tests/golden/net_dla34.npz: the reference's DLASeg on a 64x96 input, a CPU DCN in its plugin slot), as a
    fraction of each head's max-norm -- the inference path (prepare_inference) under the CURRENT arithmetic; with the
    split-bf16 arithmetic also with every DCN layer forced onto the LDS-region kernel (at the golden's map sizes "auto"
    picks the gather kernels, at the bench's 256x512 maps the region kernel)."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import cases
    from centerpoly_amd import arithmetic
    from centerpoly_amd.models.model import create_model
    from centerpoly_amd.models.networks.DCNv2.dcn_v2 import DCN
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "net_dla34.npz"), allow_pickle=True))
    shapes = {k: tuple(v) for k, v in json.loads(str(gold["shapes"])).items()}
    m = create_model("dla_34", dict(cases.HEADS), 256)
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in cases.fill_weights(shapes).items()})
    m = m.to(dev).eval()
    x = torch.from_numpy(cases.net_input("dla")).to(dev)

    def err(contraction):
        m.prepare_inference(dcn_contraction=contraction)
        with torch.no_grad():
            out = m(x)[0]
        return max(float(np.abs(out[h].cpu().numpy() - gold["s0_" + h]).max() / np.abs(gold["s0_" + h]).max())
                   for h in dict(cases.HEADS))

    out = {"reference": "tests/golden/net_dla34.npz (reference DLASeg outputs, 64x96 input)", "bar": 1e-3}
    if arithmetic.current() == "exact_f32":
        out["exact_f32"] = err("f32")
    else:
        out["split_bf16"] = err("auto")
        out["split_bf16_dcn_on_region_kernel"] = err("bf16x3_region")
    del m
    return out


# --------------------------------------------------------------------- legs ---
def infer_leg(args, dev, world, arch="dla_34", heads=None, rep="cartesian", steps=None, warmup=None,
              offset_weight_scale=1.0, tag="infer"):
    import torch
    from centerpoly_amd import _C, synth
    from centerpoly_amd.models.decode import polydet_decode
    model, _ = build_model(dev, train=False, dcn_contraction=infer_contraction(args), arch=arch, heads=heads,
                           offset_weight_scale=offset_weight_scale)
    x = torch.from_numpy(synth.normal("bench/input", (1, 3, args.height, args.width))).to(dev)

    def step():
        with torch.no_grad():
            out = model(x)[-1]
            hm = out["hm"].sigmoid_()
            return polydet_decode(hm, out["poly"], out["pseudo_depth"], reg=out["reg"], K=128, rep=rep)

    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    for i in range(warmup):
        step()
        torch.cuda.synchronize()
        note("%s warmup %d/%d done" % (tag, i + 1, warmup))
    # Per-launch HIP events cost a few microseconds of stream time each; around all ~50 DCN / conv3x3 launches of
    # a step they lengthen it by ~8 %.  So: an untimed probe pass with events on every launch (per-layer table,
    # and which launch shape dominates each kernel family), then the timed region with events ONLY around the
    # dominant shape of each family -- those are the durations `roofline` reports.
    _C.kernel_timer = _C.KernelTimer()
    probe_steps = 3
    for _ in range(probe_steps):
        step()
    probe = _C.kernel_timer.summary()
    _C.kernel_timer = None
    watch = set()
    for fam in ("dcn_fwd", "conv3x3_fwd", "heads_fused"):
        keys = [k for k in probe if k[0] == fam]
        if keys:
            watch.add(max(keys, key=lambda k: probe[k]["avg_ms"] * probe[k]["launches"]))
    run, graphed = step, False
    if args.graph:                                         # opt-in: one HIP graph replay per step
        try:
            from centerpoly_amd.utils.hip_graph import GraphedStep
            run, graphed = GraphedStep(step), True
        except Exception as e:                             # noqa: BLE001 -- report and time the eager step
            note("%s: HIP-graph capture failed (%s: %s); timing the eager step" % (tag, type(e).__name__, e))
            torch.cuda.synchronize()
    if not graphed:
        _C.kernel_timer = _C.KernelTimer(watch=watch)
    t = timed(run, steps, 0, world, tag)
    note("%s timed region done (%s)" % (tag, "graph replay" if graphed else "eager"))
    summary = dict(probe)
    if not graphed:
        summary.update(_C.kernel_timer.summary())          # dominant shapes: the timed region's own durations
        _C.kernel_timer = None
    summary["__meta__"] = {"graph": graphed, "probe_steps": probe_steps,
                           "conv3x3_ms_per_image": sum(v["avg_ms"] * v["launches"] for k, v in probe.items()
                                                       if k[0].startswith("conv3x3")) / probe_steps}
    del model
    torch.cuda.empty_cache()
    return t, summary


def offset_field_points(dev, contraction="auto", n=40):
    """The dominant DCNv2 forward launch (64->64 @256x512, one image) on three synthetic offset fields: white noise of
    0.3 px and 1 px, and a smooth field of about 3 px (what a trained offset branch produces), with the contraction
    the inference step uses and the weights prepared once (as the inference modules do).  The region kernel's cost
    grows with the share of samples leaving its staged window (cold gathers), the gather kernels' with the spatial
    noise of the offsets: the bench model (offset weights halved) is the optimistic end."""
    import numpy as np
    import torch
    from centerpoly_amd import synth
    from centerpoly_amd.models.networks.DCNv2.dcn_v2 import dcn_v2_forward_raw
    ci = co = 64
    H, W = 256, 512
    noise = synth.normal("bench/offsets/om", (1, 27, H, W))
    small = noise.copy()
    small[:, :18] *= 0.3
    k = torch.full((1, 1, 9, 9), 1.0 / 81.0)
    sm = torch.nn.functional.conv2d(torch.from_numpy(noise[:, :18]).reshape(18, 1, H, W), k, padding=4)
    smooth = noise.copy()
    smooth[:, :18] = (sm.reshape(1, 18, H, W) * 27.0).numpy()
    x = torch.from_numpy(synth.normal("bench/offsets/x", (1, ci, H, W))).to(dev)
    w = torch.from_numpy(synth.normal("bench/offsets/w", (co, ci, 3, 3), 0, 0.04)).to(dev)
    b = torch.zeros(co, device=dev)

    class Owner:
        pass

    own = Owner()
    out = []
    for name, om in (("white noise, std 0.3 px", small), ("white noise, std 1 px", noise),
                     ("smooth field, std 3 px (9x9 box-filtered noise)", smooth)):
        omt = torch.from_numpy(np.ascontiguousarray(om)).to(dev)
        for _ in range(10):
            dcn_v2_forward_raw(x, omt, w, b, contraction=contraction, owner=own)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            dcn_v2_forward_raw(x, omt, w, b, contraction=contraction, owner=own)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        hbm, mfma = dcn_roofline({("dcn_fwd", ci, co, H, W, 1): {"avg_ms": ms, "launches": n}}, fwd_contraction=contraction)
        out.append({"offsets": name, "avg_launch_us": ms * 1e3, "bound": mfma["bound"], "achieved": mfma["achieved"],
                    "unit": mfma["unit"], "peak": mfma["peak"], "frac": mfma["frac"], "arithmetic": mfma.get("arithmetic"),
                    "hbm_frac": hbm["frac"], "hbm_achieved_gbs": hbm["achieved"]})
    return {"kernel": "dcn_v2_forward 64->64 @256x512 (stand-alone launches, %d each, contraction %s)" % (n, contraction),
            "points": out}


def train_leg(args, dev, world, rank, steps, warmup, cfg="3"):
    import torch
    from centerpoly_amd import _C, synth
    from centerpoly_amd.opts import opts
    from centerpoly_amd.trains.train_factory import train_factory
    if cfg == "5":
        B, in_h, in_w, npts = 8, 384, 1280, 32
        flags = ["--poly_loss", "l1", "--poly_order", "--nbr_points", "32", "--input_h", "384", "--input_w", "1280"]
    else:
        B, in_h, in_w, npts = args.train_batch, args.height, args.width, 16
        flags = ["--poly_loss", "l1+iou", "--nbr_points", "16"]
    with contextlib.redirect_stdout(sys.stderr):          # stdout carries the JSON line only
        opt = opts().init(["polydet", "--arch", "dla_34", "--batch_size", str(B * world)] + flags)
    opt.device = dev
    heads = dict(HEADS, poly=2 * npts)
    model, _ = build_model(dev, train=True, heads=heads)
    optimizer = torch.optim.Adam(model.parameters(), opt.lr)
    trainer = train_factory["polydet"](opt, model, optimizer)
    trainer.set_device(opt.gpus, opt.chunk_sizes, dev)
    nb = synth.train_batch(B, in_h // 4, in_w // 4, nbr_points=npts, rep="cartesian",
                           stream="bench/train%s/rank%d" % (cfg, rank), in_h=in_h, in_w=in_w)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in nb.items()}    # resident before timing

    def step():
        trainer.step(batch, train=True)

    t = timed(step, steps, warmup, world, "train(cfg %s)" % cfg)
    note("train timed region done")
    # the DCNv2 launches of two more (untimed) steps, HIP-event timed on rank 0; with the timer on,
    # the backward entry point is called once per gradient group (data / weight / bias)
    if rank == 0:
        _C.kernel_timer = _C.KernelTimer()
    for _ in range(2):
        step()
    summary = _C.kernel_timer.summary() if rank == 0 else None
    _C.kernel_timer = None
    del trainer, model, optimizer, batch
    torch.cuda.empty_cache()
    return t, summary, B, (in_h, in_w, npts)


def train_rooflines(summary):
    out = {}
    for tag, name in (("dcn_fwd", "roofline"), ("dcn_bwd_data", "roofline_bwd_data"),
                      ("dcn_bwd_weight", "roofline_bwd_weight")):
        from centerpoly_amd.models.networks.DCNv2.dcn_v2 import DCN
        hbm, mfma = dcn_roofline(summary, tag, DCN.train_contraction)
        if mfma is not None:
            out[name] = hbm                    # DCNv2: the north star's HBM fraction is the headline of the pair
            out[name + "_mfma"] = mfma
    for tag, name in (("conv3x3_fwd", "roofline_conv3x3"), ("conv3x3_wgrad", "roofline_conv3x3_wgrad")):
        hbm, mfma = conv_roofline(summary, tag)
        if mfma is not None:
            out[name] = mfma
            out[name + "_hbm"] = hbm
    if summary:
        tot = {}
        for k, v in summary.items():
            tot[k[0]] = tot.get(k[0], 0.0) + v["avg_ms"] * v["launches"] / 2.0     # two traced steps
        out["dcn_ms_per_step"] = {k: round(v, 3) for k, v in tot.items()}
    return out


def detector_leg(args, dev):
    """End-to-end PolydetDetector.run on a host uint8 image (upload over PCIe, device warp +
    normalise, network, decode, device affine post-process, copy back, per-class dicts)."""
    import numpy as np
    import torch
    from centerpoly_amd import synth
    from centerpoly_amd.detectors.detector_factory import detector_factory
    from centerpoly_amd.opts import opts
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


def physical_cores():
    """CPU share of this process: physical cores of its affinity mask (SMT siblings counted once), capped by
    the cgroup CPU quota when one is set (a one-GPU box owns a 16-core share of a much larger host; more
    threads than the quota only thrash).  Returns (threads to use, logical CPUs in the mask, quota or None)."""
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    cores = set()
    for c in cpus:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as fh:
                cores.add(fh.read().strip())
        except OSError:
            cores.add(str(c))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                parts = fh.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    quota = float(parts[0]) / float(parts[1])
            elif float(parts[0]) > 0:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                    quota = float(parts[0]) / float(fh.read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    n = max(1, len(cores))
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    n = min(n, int(os.environ.get("CP_CPU_BASELINE_THREADS", "32")))     # bound: the CPU port stops scaling past ~32 threads
    return n, len(cpus), quota


def cpu_baseline(args):
    """The CPU oracle (a port of the reference's path) on a bounded sample: full-size images through
    DLA-34 + DCNv2 + sigmoid + decode, 1 warm-up + 3 timed passes, median, one thread per physical
    core this process may use (SURVEY 8(d) protocol, pass count bounded to about 30 s)."""
    import torch
    from centerpoly_amd import synth
    from oracle import decode as odec
    from oracle import nets as onet
    cores, logical, quota = physical_cores()
    torch.set_num_threads(cores)
    note("cpu baseline (oracle on %d threads; %d logical CPUs in the affinity mask, cgroup quota %s) ..."
         % (cores, logical, quota))
    h, w = args.height, args.width
    _, sd = build_model(torch.device("cpu"), train=False)
    x = torch.from_numpy(synth.normal("bench/cpu/input", (1, 3, h, w)))

    def one():
        t0 = time.perf_counter()
        with torch.no_grad():
            out = onet.dla_seg_forward(sd, x, dict(HEADS))[0]
            odec.polydet_decode(torch.sigmoid(out["hm"]), out["poly"], out["pseudo_depth"], out["reg"], K=128)
        return time.perf_counter() - t0

    warm = one()
    ts = [one() for _ in range(3)]
    t = statistics.median(ts)
    return {"value": 1.0 / t, "unit": "img/s", "cores": cores, "kind": "port",
            "sample": "%dx%d images through the oracle's DLA-34+DCNv2 forward + sigmoid + decode: 1 warm-up "
                      "(%.1f s) + 3 timed passes (%s s), median; %d threads (physical cores of this process's "
                      "CPU share: %d logical CPUs in the affinity mask, cgroup quota %s, capped at 32)"
                      % (w, h, warm, "/".join("%.1f" % v for v in ts), cores, logical, quota)}


def other_config_points(args, dev):
    """BASELINE configs 4 and 5 on one GPU (their per-GPU share), so that they are driver-run numbers."""
    out = {}
    try:
        heads = {"hm": 8, "poly": 48, "pseudo_depth": 1, "reg": 2}
        t, _ = infer_leg(args, dev, 1, arch="hourglass", heads=heads, rep="polar", steps=8, warmup=2,
                         offset_weight_scale=1.0, tag="config4")
        out["config4"] = {"workload": "BASELINE config 4: Hourglass-104 (2 stacks), 1x3x%dx%d, 24-vertex polar "
                                      "head, K=128, forward + sigmoid + decode" % (args.height, args.width),
                          "metric": "inference img/s", "value": 8 / t, "ms_per_step": 1e3 * t / 8, "steps": 8,
                          "warmup": 2, "n_gpus": 1}
    except Exception as e:                                   # never lose the main line to a side point
        out["config4"] = {"error": "%s: %s" % (type(e).__name__, e)}
    try:
        t, summary, B, _ = train_leg(args, dev, 1, 0, 6, 2, cfg="5")
        out["config5"] = dict({"workload": "BASELINE config 5 per-GPU share: DLA-34 + DCNv2 training, 8 x 3x384x1280, "
                                           "32-vertex cartesian, l1 + order loss, Adam",
                               "metric": "train img/s", "value": B * 6 / t, "ms_per_step": 1e3 * t / 6, "steps": 6,
                               "warmup": 2, "n_gpus": 1}, **train_rooflines(summary))
    except Exception as e:
        out["config5"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out

# --------------------------------------------------------------- the JSON line ---
LINE_LIMIT = 8000      # bytes; the driver keeps the tail of stdout, a 21 KB line lost its head in round 3


def _sig(v, n=5):
    """Floats to n significant digits (recursively); everything else unchanged."""
    if isinstance(v, float):
        return float("%.*g" % (n, v))
    if isinstance(v, dict):
        return {k: _sig(x, n) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, n) for x in v]
    return v


def _pick(d, keys):
    return None if d is None else {k: d[k] for k in keys if k in d}


ROOF_KEYS = ("kernel", "bound", "avg_launch_us", "launches", "algorithmic_bytes_per_launch",
             "algorithmic_flops_per_launch", "achieved", "peak", "unit", "frac", "issued_frac", "traffic")


def write_detail(line, args):
    """Everything the compact line leaves out (per-family tables, the other configs' rooflines, sources of the PMC
    traffic figures ...) goes to stderr and to a side file; returns the file's path relative to the repo."""
    text = json.dumps(_sig(line, 7), indent=1, sort_keys=True)
    print("[bench detail]\n" + text, file=sys.stderr, flush=True)
    for rel in (os.path.join("gpurun_out", "bench_detail.json"), "bench_detail.json"):
        try:
            os.makedirs(os.path.dirname(os.path.join(ROOT, rel)) or ROOT, exist_ok=True)
            with open(os.path.join(ROOT, rel), "w") as fh:
                fh.write(text + "\n")
            return rel
        except OSError:
            continue
    return "stderr only"


def compact_line(line):
    """The ONE stdout line: the contract's keys, the rooflines as algorithmic fractions, the precision contract, the
    training point and the CPU baseline -- below LINE_LIMIT bytes whatever the legs produced."""
    out = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data", "config") if k in line}
    for k in ("roofline", "roofline_mfma", "roofline_bwd_data", "roofline_bwd_data_mfma", "roofline_bwd_weight",
              "roofline_bwd_weight_mfma", "roofline_conv3x3", "roofline_heads"):
        if line.get(k) is not None:
            out[k] = _pick(line[k], ROOF_KEYS)
    if "max_rel_err_vs_golden" in line:
        out["max_rel_err_vs_golden"] = {k: v for k, v in line["max_rel_err_vs_golden"].items() if k != "reference"}
    for k in ("exact_f32", "split_bf16"):
        if k in line:
            out[k] = _pick(line[k], ("value", "ms_per_step", "steps", "warmup"))
    if "roofline_by_offsets" in line:
        out["roofline_by_offsets"] = [{"offsets": p["offsets"].split(" (")[0], "avg_launch_us": p["avg_launch_us"],
                                       "hbm_frac": p["hbm_frac"]} for p in line["roofline_by_offsets"]["points"]]
    if "detector_end_to_end" in line:
        out["detector_end_to_end"] = _pick(line["detector_end_to_end"], ("value", "ms_per_image"))
    if "train" in line:
        tr = line["train"]
        out["train"] = _pick(tr, ("metric", "value", "ms_per_step", "n_gpus", "global_batch", "steps", "warmup"))
        for k in ("roofline", "roofline_bwd_data", "roofline_bwd_weight"):
            if tr.get(k) is not None:
                out["train"][k] = _pick(tr[k], ROOF_KEYS)
        if "dcn_ms_per_step" in tr:
            out["train"]["family_ms_per_step"] = tr["dcn_ms_per_step"]
    if "other_configs" in line:
        out["other_configs"] = {c: _pick(v, ("metric", "value", "ms_per_step", "steps", "warmup", "n_gpus", "error"))
                                for c, v in line["other_configs"].items()}
    for k in ("scaling_base", "cpu_baseline", "detail"):
        if k in line:
            out[k] = line[k]
    out = _sig(out)
    text = json.dumps(out, separators=(",", ":"))
    for drop in ("other_configs", "roofline_by_offsets", "roofline_heads", "roofline_conv3x3", "detector_end_to_end"):
        if len(text) <= LINE_LIMIT:
            break
        out.pop(drop, None)                      # never reached with today's legs; the limit is a hard promise
        text = json.dumps(out, separators=(",", ":"))
    assert len(text) <= LINE_LIMIT, len(text)
    return text


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    plan, detail = launch_plan(args, os.environ, argv)
    if plan == "refuse":
        print("bench.py: " + detail, file=sys.stderr, flush=True)
        return EXIT_REFUSED
    if plan == "spawn":
        return spawn_ranks(args, detail)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print("bench.py needs a HIP device (there is no CPU fallback)", file=sys.stderr, flush=True)
        return EXIT_REFUSED
    if torch.cuda.device_count() <= local_rank:
        print("bench.py: rank %d has no device %d" % (rank, local_rank), file=sys.stderr, flush=True)
        return EXIT_REFUSED
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    cfg = args.config
    if cfg == "auto":
        if args.mode == "auto":
            cfg = "2" if world == 1 else "3"
        else:
            cfg = "2" if args.mode == "infer" else "3"
    mode = "infer" if cfg in ("2", "4") else "train"
    torch.backends.cudnn.benchmark = os.environ.get("CP_MIOPEN_BENCHMARK", "0") == "1"
    from centerpoly_amd import arithmetic
    arithmetic.configure(args.arithmetic)
    line = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": arithmetic.describe(), "arithmetic": args.arithmetic,
            "data": "synthetic (counter-hash inputs, name-hashed random-init weights)"}
    if mode == "infer":
        if cfg == "4":
            heads = {"hm": 8, "poly": 48, "pseudo_depth": 1, "reg": 2}
            t, summary = infer_leg(args, dev, world, arch="hourglass", heads=heads, rep="polar",
                                   offset_weight_scale=1.0)
            workload = ("BASELINE config 4: Hourglass-104 (2 stacks), 1x3x%dx%d synthetic, 24-vertex polar head, "
                        "K=128, forward + sigmoid + NMS/top-k/decode" % (args.height, args.width))
            metric = "inference img/s @2048x1024 Hourglass-104 (1 GPU)"
        else:
            t, summary = infer_leg(args, dev, world)
            workload = ("BASELINE config 2: DLA-34 + DCNv2, 1x3x%dx%d synthetic, 16-vertex cartesian head, "
                        "K=128, forward + sigmoid + NMS/top-k/decode" % (args.height, args.width))
            metric = "inference img/s @2048x1024 DLA-34 (1 GPU)"
        meta = summary.pop("__meta__")
        hbm, mfma = dcn_roofline(summary, fwd_contraction=infer_contraction(args))
        chbm, cmfma = conv_roofline(summary)
        hhbm, hmfma = heads_roofline(summary)
        if mfma is None:                                   # Hourglass: no DCN, the 3x3 convolution dominates
            hbm, mfma = chbm, cmfma
        line.update({
            "metric": metric, "unit": "img/s",
            "value": world * args.steps / t, "ms_per_step": 1e3 * t / args.steps,
            "config": {"workload": workload, "images_per_step": world,
                       "parallelism": "replicas" if world > 1 else "single"},
            # `roofline` is the north star's figure -- the dominant DCNv2 forward launch against HBM (algorithmic
            # bytes of SURVEY 8(d) / HIP-event launch time / 8 TB/s); `roofline_mfma` prices the same launch's
            # algorithmic flops against the matrix pipe it issues on (the issued flops are `issued_frac`)
            "roofline": hbm, "roofline_mfma": mfma,
            "roofline_conv3x3": cmfma, "roofline_conv3x3_hbm": chbm,
            "roofline_heads": hmfma, "roofline_heads_hbm": hhbm,
            "dcn_layers_ms": {"%d->%d@%dx%d" % k[1:5]: round(v["avg_ms"], 4) for k, v in summary.items()
                              if k[0].startswith("dcn")},
            "conv3x3_ms_per_image": round(meta["conv3x3_ms_per_image"], 4),
            "execution": ("one HIP graph replay per step (captured once after warm-up)" if meta["graph"]
                          else "eager launches"),
            "kernel_timing": ("HIP events around each launch on %d eager passes of the same step before the "
                              "timed region" % meta["probe_steps"]) if meta["graph"]
                             else ("roofline / roofline_conv3x3: HIP events around the dominant launch shape inside "
                                   "the timed region; per-layer tables: %d probe passes before it"
                                   % meta["probe_steps"]),
        })
        if world == 1 and cfg == "2":
            if not args.no_offset_points:
                line["roofline_by_offsets"] = offset_field_points(dev, infer_contraction(args))
                note("offset-field points done")
            if not args.no_exact_point:
                # the precision contract beside the headline: error of the timed arithmetic against the reference's
                # golden outputs, and the same step on exact fp32 chains (library convolutions, f32-MFMA DCNv2)
                line["max_rel_err_vs_golden"] = golden_error(dev)
                other = "exact_f32" if args.arithmetic == "split_bf16" else "split_bf16"
                arithmetic.configure(other)
                try:
                    t2, _ = infer_leg(args, dev, 1, steps=10, warmup=3, tag="infer(%s)" % other)
                    line[other] = {"what": "the same step under the %s arithmetic (%s)" % (other, arithmetic.describe()),
                                   "value": 10 / t2, "ms_per_step": 1e3 * t2 / 10, "steps": 10, "warmup": 3}
                    line["max_rel_err_vs_golden"].update({k: v for k, v in golden_error(dev).items()
                                                          if k not in ("reference", "bar")})
                finally:
                    arithmetic.configure(args.arithmetic)
                note("%s point done" % other)
            if not args.no_detector_point:
                line["detector_end_to_end"] = detector_leg(args, dev)
                note("detector end-to-end point done")
            if not args.no_train_point:
                tt, ts, B, _ = train_leg(args, dev, 1, 0, args.train_steps, 2)
                # same workload per GPU as the N > 1 lines: the 1-GPU point of the training scaling curve
                line["train"] = dict({"metric": "train img/s 1/2/4/8 GPU @2048x1024 DLA-34",
                                      "value": B * args.train_steps / tt,
                                      "ms_per_step": 1e3 * tt / args.train_steps, "n_gpus": 1,
                                      "global_batch": B, "steps": args.train_steps, "warmup": 2,
                                      "workload": "BASELINE config 3 per-GPU share: DLA-34 + DCNv2, %d x 3x%dx%d, "
                                                  "l1+iou polygon loss, Adam" % (B, args.height, args.width)},
                                     **train_rooflines(ts))
            if not args.no_other_configs:
                line["other_configs"] = other_config_points(args, dev)
    else:
        t, summary, B, (in_h, in_w, npts) = train_leg(args, dev, world, rank, args.steps, args.warmup, cfg=cfg)
        line.update({
            "metric": "train img/s 1/2/4/8 GPU @%dx%d DLA-34" % (in_w, in_h), "unit": "img/s",
            "value": world * B * args.steps / t, "ms_per_step": 1e3 * t / args.steps,
            "config": {"workload": "BASELINE config %s: DLA-34 + DCNv2 training, %d x 3x%dx%d per GPU, "
                                   "%d-vertex cartesian, %s, Adam lr 4e-6"
                                   % (cfg, B, in_h, in_w, npts,
                                      "l1+iou polygon loss" if cfg == "3" else "l1 + order loss"),
                       "global_batch": world * B,
                       "parallelism": "dp%d (one process per GPU, RCCL all-reduce)" % world},
            "scaling_base": "weak scaling of the training leg: compare with the N=1 line's "
                            "train.value (same %d img/GPU workload), not with its inference value" % B,
        })
        if rank == 0:
            line.update(train_rooflines(summary))      # forward (`roofline`) and both backward kernels
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        line["detail"] = write_detail(line, args)
        print(compact_line(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
