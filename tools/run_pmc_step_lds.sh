#!/bin/bash
# One PMC pass (LDS counters) over bench.py's inference leg, reduced per kernel: which kernels of the step wait on LDS /
# fight over banks.  Usage: tools/run_pmc_step_lds.sh <tag> -> gpurun_out/pmc_step_lds_<tag>/summary.txt
set -u
cd "$(dirname "$0")/.."
tag=${1:-r04}
out=gpurun_out/pmc_step_lds_${PMC_LEG:-infer}_$tag
mkdir -p $out
export TMPDIR=/tmp
# PMC_LEG=train: the B = 4 training leg instead (the inference leg shrunk to one step)
if [ "${PMC_LEG:-infer}" = train ]; then
  B="python3 bench.py --config 2 --steps 1 --warmup 1 --no_cpu_baseline --no_detector_point --no_offset_points --no_other_configs --no_exact_point"
else
  B="python3 bench.py --config 2 --steps 3 --warmup 2 --no_cpu_baseline --no_train_point --no_detector_point --no_offset_points --no_other_configs --no_exact_point"
fi
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $out/p1 -o p1 -- $B > $out/p1.log 2>&1
rc=$?; echo "pass rc=$rc"
python3 - "$out" <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/p1/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:90]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        n[k] += 1
rows = []
for k, c in acc.items():
    wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    rows.append((c.get("SQ_WAIT_INST_LDS", 0.0), k, n[k], wc, c))
rows.sort(reverse=True)
with open(out + "/summary.txt", "w") as o:
    o.write("kernel | launches | LDS-wait share of wave cycles | bank-conflict share of LDS-active cycles | VALU / MFMA / LDS instructions per launch\n")
    for w, k, m, wc, c in rows[:40]:
        ia = c.get("SQ_LDS_IDX_ACTIVE", 0.0) or 1.0
        o.write("%-90s %4d  %5.1f%%  %5.1f%%  %10.0f %10.0f %10.0f\n" % (k, m, 100 * w / wc, 100 * c.get("SQ_LDS_BANK_CONFLICT", 0.0) / ia,
                c.get("SQ_INSTS_VALU", 0) / max(m, 1), c.get("SQ_INSTS_MFMA", 0) / max(m, 1), c.get("SQ_INSTS_LDS", 0) / max(m, 1)))
print(open(out + "/summary.txt").read())
P
