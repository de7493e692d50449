#!/bin/bash
# rocprofv3 kernel statistics of the B=4 training leg (bench.py --config 3 on one GPU).
set -u
cd "$(dirname "$0")/.."
tag=${1:-r03}
out=gpurun_out/prof_train_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o train -- python3 bench.py --config 3 --steps 4 --warmup 2 --no_cpu_baseline > $out/bench.json 2> $out/bench.err
echo "rc=$?"
tail -2 $out/bench.err
f=$(find $out -name "*kernel_stats.csv" | head -1)
head -25 $f | cut -c1-200
