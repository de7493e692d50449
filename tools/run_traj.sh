python tools/train_trajectory.py 8 split_bf16 2>/dev/null | tail -1 > gpurun_out/traj_a.json
python tools/train_trajectory.py 8 split_bf16 2>/dev/null | tail -1 > gpurun_out/traj_a2.json
python tools/train_trajectory.py 8 exact_f32 2>/dev/null | tail -1 > gpurun_out/traj_b.json
