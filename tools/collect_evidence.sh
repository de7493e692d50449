#!/bin/bash
# One GPU call that re-takes every measurement the round's documents cite and copies the summaries into profiles/
# (tracked).  Usage (on the GPU box, through gpurun): tools/collect_evidence.sh r03
set -u
cd "$(dirname "$0")/.."
tag=${1:-r03}
P=gpurun_out/publish_$tag          # (only gpurun_out/ travels back from the GPU box: copy P/* into profiles/ afterwards)
mkdir -p $P
log=gpurun_out/evidence_$tag.log
mkdir -p gpurun_out
step() { echo "[$(date +%H:%M:%S)] $*" | tee -a $log; }

step "bench.py (default line)"
timeout -k 10 900 python3 bench.py > gpurun_out/${tag}_bench_n1.json 2> gpurun_out/${tag}_bench_n1.err || { step "bench failed"; exit 1; }
tail -1 gpurun_out/${tag}_bench_n1.json > $P/${tag}_bench_n1.json

step "rocprofv3 kernel stats: inference leg"
tools/run_infer_prof.sh $tag >> $log 2>&1
cp gpurun_out/prof_infer_$tag/infer_kernel_stats.csv $P/${tag}_infer_kernel_stats.csv
python3 tools/by_launch_shape.py gpurun_out/prof_infer_$tag/infer_kernel_trace.csv > $P/${tag}_infer_by_launch_shape.csv

step "rocprofv3 kernel stats: training leg (B = 4)"
tools/run_train_prof.sh $tag >> $log 2>&1
cp gpurun_out/prof_train_$tag/train_kernel_stats.csv $P/${tag}_train_b4_kernel_stats.csv
python3 tools/by_launch_shape.py gpurun_out/prof_train_$tag/train_kernel_trace.csv > $P/${tag}_train_b4_by_launch_shape.csv

for cfg in 4 5; do
  step "rocprofv3 kernel stats: BASELINE config $cfg (after a warm pass)"
  tools/run_cfg_prof.sh $cfg $tag >> $log 2>&1
  cp gpurun_out/prof_cfg${cfg}_$tag/cfg${cfg}_kernel_stats.csv $P/${tag}_cfg${cfg}_kernel_stats.csv
  python3 tools/by_launch_shape.py gpurun_out/prof_cfg${cfg}_$tag/cfg${cfg}_kernel_trace.csv > $P/${tag}_cfg${cfg}_by_launch_shape.csv
done

step "PMC: HBM traffic of the dominant launches on the bench's own tensors"
tools/run_pmc_bench.sh $tag >> $log 2>&1
for f in dcn_fwd_pmc.json conv_mfma_pmc.json heads_fused_pmc.json; do
  [ -s gpurun_out/pmc_bench_$tag/$f ] && cp gpurun_out/pmc_bench_$tag/$f $P/$f
done
cp gpurun_out/pmc_bench_$tag/${tag}_bench_*_pmc_*.csv $P/ 2>/dev/null

step "PMC: counters of the DCNv2 forward region kernel (64->64 @256x512)"
tools/run_pmc_fwd.sh $tag >> $log 2>&1
cp gpurun_out/pmc_fwd_$tag/summary.txt $P/${tag}_dcn_fwd_region_pmc_summary.txt

step "PMC: counters of the DCNv2 backward kernels (64->64 @256x512 x4)"
PMC_WHAT=data tools/run_pmc_bwd.sh ${tag}_data >> $log 2>&1
cp gpurun_out/pmc_bwd_${tag}_data/summary.txt $P/${tag}_dcn_bwd_data_pmc_summary.txt
PMC_WHAT=weight tools/run_pmc_bwd.sh ${tag}_weight >> $log 2>&1
cp gpurun_out/pmc_bwd_${tag}_weight/summary.txt $P/${tag}_dcn_bwd_weight_pmc_summary.txt
python3 tools/pmc_bwd_traffic.py $P/${tag}_dcn_bwd_data_pmc_summary.txt $P/${tag}_dcn_bwd_weight_pmc_summary.txt $P/dcn_bwd_pmc.json >> $log 2>&1

step "PMC: counters of the MFMA convolution (64->64 @256x512 x4)"
PMC_SCRIPT=tools/pmc_conv.py tools/run_pmc_fwd.sh ${tag}_conv >> $log 2>&1
cp gpurun_out/pmc_fwd_${tag}_conv/summary.txt $P/${tag}_conv_mfma_pmc_summary.txt

step "probes: convolution ablations / in-kernel stamps / stride-2 input gradient"
timeout -k 10 300 python3 tools/probe_conv_ablate.py > $P/${tag}_conv_mfma_ablations.txt 2>> $log
timeout -k 10 300 python3 tools/probe_conv_stamp.py > $P/${tag}_conv_mfma_stamps.txt 2>> $log
timeout -k 10 300 python3 tools/probe_s2_igrad.py > $P/${tag}_conv_s2_igrad_probe.txt 2>> $log

step "probes: region kernel phases, ablations, offset fields; backward launch times"
timeout -k 10 300 python3 tools/probe_region_stamp.py > $P/${tag}_dcn_fwd_region_stamps.txt 2>> $log
timeout -k 10 300 python3 tools/probe_region_ablate.py > $P/${tag}_dcn_fwd_region_ablations.txt 2>> $log
timeout -k 10 300 python3 tools/probe_dcn_region.py > $P/${tag}_dcn_fwd_region_probe.txt 2>> $log
timeout -k 10 300 python3 tools/probe_dcn_bwd.py > $P/${tag}_dcn_bwd_probe.txt 2>> $log
step "done"
