#!/bin/bash
# rocprofv3 kernel statistics of BASELINE config 4 (Hourglass-104 inference) or 5 (KITTI-shape training) on one GPU.
# The first pass warms MIOpen's find / kernel caches (its naive_conv_* launches are find-time); the second is the
# profile that is kept.  Usage: tools/run_cfg_prof.sh <4|5> <tag>
set -u
cd "$(dirname "$0")/.."
cfg=${1:-4}
tag=${2:-r03}
out=gpurun_out/prof_cfg${cfg}_$tag
mkdir -p $out
export TMPDIR=/tmp
args="--config $cfg --steps 6 --warmup 2 --no_cpu_baseline"
timeout -k 10 600 python3 bench.py $args > $out/warm.json 2> $out/warm.err
echo "warm rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o cfg$cfg -- python3 bench.py $args > $out/bench.json 2> $out/bench.err
echo "rc=$?"
f=$(find $out -name "*kernel_stats.csv" | head -1)
head -30 $f | cut -c1-200
