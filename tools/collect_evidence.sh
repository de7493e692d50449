#!/bin/bash
# One GPU call that re-takes every measurement the round's documents cite and stages the summaries for profiles/
# (tracked).  Usage (on the GPU box, through gpurun): tools/collect_evidence.sh r04 [steps...]
#   steps (default: all): bench infer train cfg pmc_bench pmc_fwd pmc_bwd probes
# Every step's return code is checked: a failed step leaves `FAILED_<step>.txt` (with the tail of its log) under the
# publish directory and removes the artefacts it would have overwritten -- no stale or partial file is published under
# the new tag.  Only gpurun_out/ travels back from the GPU box: copy gpurun_out/publish_<tag>/* into profiles/ afterwards.
set -uo pipefail
cd "$(dirname "$0")/.."
tag=${1:-r04}
shift || true
steps=${*:-bench infer train cfg pmc_bench pmc_fwd pmc_bwd probes}
P=gpurun_out/publish_$tag
mkdir -p "$P"
log=gpurun_out/evidence_$tag.log
: > "$log"
say() { echo "[$(date +%H:%M:%S)] $*" | tee -a "$log"; }
failed=0

# run NAME TARGETS... -- CMD...: stale targets are removed first; on failure they are removed again and the step is marked
run() {
  local name=$1; shift
  local targets=()
  while [ "$1" != "--" ]; do targets+=("$1"); shift; done
  shift
  rm -f "${targets[@]}" "$P/FAILED_$name.txt"
  say "$name: $*"
  if "$@" >> "$log" 2>&1; then
    for t in "${targets[@]}"; do
      [ -s "$t" ] || { say "$name: produced no $t"; echo "missing $t" >> "$P/FAILED_$name.txt"; failed=1; }
    done
  else
    local rc=$?
    say "$name FAILED (rc $rc)"
    { echo "rc $rc: $*"; tail -20 "$log"; } > "$P/FAILED_$name.txt"
    rm -f "${targets[@]}"
    failed=1
  fi
}
want() { case " $steps " in *" $1 "*) return 0;; *) return 1;; esac; }

if want bench; then
  run bench "$P/${tag}_bench_n1.json" -- bash -c "timeout -k 10 900 python3 bench.py > gpurun_out/${tag}_bench_stdout.txt 2> gpurun_out/${tag}_bench_n1.err && tail -1 gpurun_out/${tag}_bench_stdout.txt > $P/${tag}_bench_n1.json && cp gpurun_out/bench_detail.json $P/${tag}_bench_n1_detail.json && test \$(wc -c < $P/${tag}_bench_n1.json) -lt 8000"
fi
if want infer; then
  run infer "$P/${tag}_infer_kernel_stats.csv" "$P/${tag}_infer_by_launch_shape.csv" -- bash -c "tools/run_infer_prof.sh $tag && cp gpurun_out/prof_infer_$tag/infer_kernel_stats.csv $P/${tag}_infer_kernel_stats.csv && python3 tools/by_launch_shape.py gpurun_out/prof_infer_$tag/infer_kernel_trace.csv > $P/${tag}_infer_by_launch_shape.csv"
fi
if want train; then
  run train "$P/${tag}_train_b4_kernel_stats.csv" "$P/${tag}_train_b4_by_launch_shape.csv" -- bash -c "tools/run_train_prof.sh $tag && cp gpurun_out/prof_train_$tag/train_kernel_stats.csv $P/${tag}_train_b4_kernel_stats.csv && python3 tools/by_launch_shape.py gpurun_out/prof_train_$tag/train_kernel_trace.csv > $P/${tag}_train_b4_by_launch_shape.csv"
fi
if want cfg; then
  for cfg in 4 5; do
    run cfg$cfg "$P/${tag}_cfg${cfg}_kernel_stats.csv" "$P/${tag}_cfg${cfg}_by_launch_shape.csv" -- bash -c "tools/run_cfg_prof.sh $cfg $tag && cp gpurun_out/prof_cfg${cfg}_$tag/cfg${cfg}_kernel_stats.csv $P/${tag}_cfg${cfg}_kernel_stats.csv && python3 tools/by_launch_shape.py gpurun_out/prof_cfg${cfg}_$tag/cfg${cfg}_kernel_trace.csv > $P/${tag}_cfg${cfg}_by_launch_shape.csv"
  done
fi
if want pmc_bench; then
  run pmc_bench "$P/dcn_fwd_pmc.json" "$P/conv_mfma_pmc.json" "$P/heads_fused_pmc.json" -- bash -c "tools/run_pmc_bench.sh $tag && cp gpurun_out/pmc_bench_$tag/dcn_fwd_pmc.json gpurun_out/pmc_bench_$tag/conv_mfma_pmc.json gpurun_out/pmc_bench_$tag/heads_fused_pmc.json $P/ && cp gpurun_out/pmc_bench_$tag/${tag}_bench_*_pmc_*.csv $P/"
fi
if want pmc_fwd; then
  run pmc_fwd "$P/${tag}_dcn_fwd_region_pmc_summary.txt" -- bash -c "tools/run_pmc_fwd.sh $tag && cp gpurun_out/pmc_fwd_$tag/summary.txt $P/${tag}_dcn_fwd_region_pmc_summary.txt"
fi
if want pmc_bwd; then
  run pmc_bwd_data "$P/${tag}_dcn_bwd_data_pmc_summary.txt" -- bash -c "PMC_WHAT=data tools/run_pmc_bwd.sh ${tag}_data && cp gpurun_out/pmc_bwd_${tag}_data/summary.txt $P/${tag}_dcn_bwd_data_pmc_summary.txt"
  run pmc_bwd_weight "$P/${tag}_dcn_bwd_weight_pmc_summary.txt" -- bash -c "PMC_WHAT=weight tools/run_pmc_bwd.sh ${tag}_weight && cp gpurun_out/pmc_bwd_${tag}_weight/summary.txt $P/${tag}_dcn_bwd_weight_pmc_summary.txt"
  run pmc_bwd_json "$P/dcn_bwd_pmc.json" -- python3 tools/pmc_bwd_traffic.py "$P/${tag}_dcn_bwd_data_pmc_summary.txt" "$P/${tag}_dcn_bwd_weight_pmc_summary.txt" "$P/dcn_bwd_pmc.json"
fi
if want probes; then
  run probe_decode "$P/${tag}_decode_probe.txt" -- bash -c "timeout -k 10 300 python3 tools/probe_decode.py > $P/${tag}_decode_probe.txt"
  run probe_bwd "$P/${tag}_dcn_bwd_probe.txt" -- bash -c "timeout -k 10 300 python3 tools/probe_dcn_bwd.py > $P/${tag}_dcn_bwd_probe.txt"
  run probe_conv_split "$P/${tag}_conv_split_probe.txt" -- bash -c "timeout -k 10 200 python3 tools/probe_conv_split.py > $P/${tag}_conv_split_probe.txt"
  run probe_base_pair "$P/${tag}_base_pair_probe.txt" -- bash -c "timeout -k 10 200 python3 tools/probe_base_pair.py > $P/${tag}_base_pair_probe.txt"
  run probe_conv_direct "$P/${tag}_conv_direct_bf16_probe.txt" -- bash -c "timeout -k 10 200 python3 tools/probe_conv_direct_bf16.py > $P/${tag}_conv_direct_bf16_probe.txt"
fi
say "done (failed=$failed)"
exit $failed
