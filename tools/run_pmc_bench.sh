#!/bin/bash
# HBM-traffic PMC passes over bench.py's own inference leg (one counter per pass).
set -u
cd "$(dirname "$0")/.."
tag=${1:-r03}
out=gpurun_out/pmc_bench_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --config 2 --steps 3 --warmup 2 --no_cpu_baseline --no_train_point --no_detector_point --no_offset_points --no_other_configs --no_exact_point"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -o $c -- $B > $out/$c.log 2>&1
  rc=$?; echo "pass $c rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 $out/$c.log; exit 1; fi
done
python3 tools/pmc_bench_traffic.py $out/FETCH_SIZE $out/WRITE_SIZE $out/dcn_fwd_pmc.json $out/conv_mfma_pmc.json $out/heads_fused_pmc.json
for c in FETCH_SIZE WRITE_SIZE; do
  f=$(find $out/$c -name "*counter_collection.csv" | head -1)
  (head -1 $f; grep dcn_fwd $f) > $out/${tag}_bench_dcn_fwd_pmc_$c.csv
  (head -1 $f; grep -e "conv_mfma_kernel<2, 2, 9" -e "conv_mfma_kernel<2, 1, 9, 2" $f) > $out/${tag}_bench_conv_mfma_pmc_$c.csv
  (head -1 $f; grep "conv_heads_fused" $f) > $out/${tag}_bench_heads_fused_pmc_$c.csv
done
