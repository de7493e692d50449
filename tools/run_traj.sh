#!/bin/bash
# Loss trajectories of 8 training steps under both arithmetics (tools/train_trajectory.py): gpurun_out/traj_*.json.
# A crashed run leaves a FAILED marker and its stderr in gpurun_out/traj_*.err instead of an empty JSON.
set -euo pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for spec in "a split_bf16" "a2 split_bf16" "b exact_f32"; do
  set -- $spec
  if python3 tools/train_trajectory.py 8 "$2" > "gpurun_out/traj_$1.out" 2> "gpurun_out/traj_$1.err"; then
    tail -1 "gpurun_out/traj_$1.out" > "gpurun_out/traj_$1.json"
  else
    echo "trajectory $1 ($2) FAILED: see gpurun_out/traj_$1.err" | tee "gpurun_out/traj_$1.FAILED"
    exit 1
  fi
done
