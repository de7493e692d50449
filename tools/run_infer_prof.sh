#!/bin/bash
# rocprofv3 kernel statistics of the inference leg (bench.py --config 2 on one GPU).
set -u
cd "$(dirname "$0")/.."
tag=${1:-r03}
out=gpurun_out/prof_infer_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o infer -- python3 bench.py --config 2 --no_cpu_baseline --no_train_point --no_detector_point --no_offset_points --no_other_configs --no_exact_point > $out/bench.json 2> $out/bench.err
echo "rc=$?"
f=$(find $out -name "*kernel_stats.csv" | head -1)
head -40 $f | cut -c1-220
