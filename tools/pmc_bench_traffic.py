"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) taken over `bench.py --config 2` (the
inference leg on the bench's own tensors) to HBM bytes per launch of the DCNv2 forward launches, keyed
by launch shape, stamped with the kernel revision bench.py checks.

  python3 tools/pmc_bench_traffic.py <fetch_dir> <write_dir> <out.json>

Correction (MI355X_MICROARCH.md, HBM): gfx950's FETCH_SIZE reports half of the bytes of a wide
coalesced stream, WRITE_SIZE is exact; traffic = (2*FETCH_SIZE + WRITE_SIZE) KB.  This kernel's reads
are 8-byte gathers, for which the doubling is an upper bound (stated in the JSON)."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

# (B, Cin, Cout, H, W) of the DLA-34 DCN layers at 2048x1024 by launch grid (threads).  Region kernel
# (dcn_fwd_region.hip): 256 threads per 8 x 32 pixel tile and 64-channel block -> 64->64 @256x512 = 512 workgroups;
# the gather kernels' 524288-thread grid (256 threads per 64-pixel tile) is kept for the exact-f32 arithmetic
# (round 4: other region launches -- the K-split ones, grid (32768, 1, 4) -- have the same thread count as the fused module
# launch: the kernel NAME tells them apart, `<true>` = the fused form)
SHAPES = {("dcn_fwd_region_kernel<true>", 131072): (1, 64, 64, 256, 512), ("dcn_fwd_pipe", 524288): (1, 64, 64, 256, 512)}


# the dominant launch shape of the split-bf16 3x3 convolution at inference (bench.py's roofline_conv3x3):
# 128 -> 128 @128x256 = conv_mfma_kernel<2, 2, 9, ...>, 128 tiles x 4 output-channel tiles x 256 threads
# and 256 -> 256 @64x128 = conv_mfma_kernel<2, 1, 9, 2, ...> (in-workgroup K split), 64 tiles x 8 x 512 threads
CONV_LAUNCHES = [("conv_mfma_kernel<2, 2, 9", 128 * 4 * 256, (1, 128, 128, 128, 256)),
                 ("conv_mfma_kernel<2, 1, 9, 2", 64 * 8 * 512, (1, 256, 256, 64, 128))]


def avg(dirname, counter, match="dcn_fwd"):
    vals = {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if match in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.setdefault(int(row["Grid_Size"]), []).append(float(row["Counter_Value"]))
    return {g: (sum(v) / len(v), len(v)) for g, v in vals.items()}


layers, raw = {}, {}
for (kern, grid), shape in SHAPES.items():
    fetch, write = avg(sys.argv[1], "FETCH_SIZE", kern), avg(sys.argv[2], "WRITE_SIZE", kern)
    if grid in fetch and grid in write:
        f, n = fetch[grid]
        w, _ = write[grid]
        layers["%dx%dx%dx%dx%d" % shape] = (2.0 * f + w) * 1024.0
        raw["%dx%dx%dx%dx%d" % shape] = {"FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w, "launches": n}
out = {"kernel_rev": bench.kernel_revision(), "inputs": "bench.py infer leg",
       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py "
                  "--config 2 --steps 3 --warmup 2 --no_cpu_baseline --no_train_point --no_detector_point "
                  "--no_offset_points --no_other_configs --no_exact_point (separate passes)",
       "correction": "traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes (gfx950: FETCH_SIZE reports half of a wide coalesced "
                     "stream's bytes; the region kernel stages rows with dword loads, 160-byte runs: the doubling is an "
                     "upper bound for them)",
       "layers": layers, "_raw": raw}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))


# ---- the same two passes reduced for the MFMA convolution (wide coalesced streams: the doubling applies) ----
if len(sys.argv) > 4:
    layers, raw = {}, {}
    for kern, grid, shape in CONV_LAUNCHES:
        fetch, write = avg(sys.argv[1], "FETCH_SIZE", kern), avg(sys.argv[2], "WRITE_SIZE", kern)
        if grid in fetch and grid in write:
            f, n = fetch[grid]
            w, _ = write[grid]
            layers["%dx%dx%dx%dx%d" % shape] = (2.0 * f + w) * 1024.0
            raw["%dx%dx%dx%dx%d" % shape] = {"FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w, "launches": n}
    out = {"kernel_rev": bench.kernel_revision(bench.CONV_SOURCES), "inputs": "bench.py infer leg",
           "command": out["command"],
           "correction": "traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes (gfx950: FETCH_SIZE reports half of a wide "
                         "coalesced stream's bytes, WRITE_SIZE is exact)",
           "layers": layers, "_raw": raw}
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(out))


# ---- and for the fused heads launch (csrc/heads_fused.hip): 512 tiles x 4 heads x 256 threads ----
if len(sys.argv) > 5:
    fetch, write = avg(sys.argv[1], "FETCH_SIZE", "conv_heads_fused"), avg(sys.argv[2], "WRITE_SIZE", "conv_heads_fused")
    layers, raw = {}, {}
    for grid, shape in {512 * 4 * 256: (1, 64, 1024, 256, 512)}.items():
        if grid in fetch and grid in write:
            f, n = fetch[grid]
            w, _ = write[grid]
            layers["%dx%dx%dx%dx%d" % shape] = (2.0 * f + w) * 1024.0
            raw["%dx%dx%dx%dx%d" % shape] = {"FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w, "launches": n}
    out = {"kernel_rev": bench.kernel_revision(bench.HEADS_SOURCES), "inputs": "bench.py infer leg", "command": out["command"],
           "correction": "traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes (gfx950: FETCH_SIZE reports half of a wide "
                         "coalesced stream's bytes, WRITE_SIZE is exact)",
           "layers": layers, "_raw": raw}
    json.dump(out, open(sys.argv[5], "w"), indent=1)
    print(json.dumps(out))
