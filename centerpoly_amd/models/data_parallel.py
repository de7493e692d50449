"""`DataParallel(module, device_ids, output_device, dim, chunk_sizes)` -- the name the reference's
main.py / base_trainer.py import (src/lib/models/data_parallel.py:119-128, base_trainer.py:51-57).

The reference replicates the model inside ONE process (per-step parameter broadcast, scatter with
optionally uneven `chunk_sizes`, output gather and gradient reduce onto GPU 0).  This package trains
with one process per GPU and RCCL all-reduces (trains/base_trainer.py), so the only in-process
configuration is a single device: the module is returned as is.  Asking for several devices in one
process is refused with the command that starts the per-GPU processes."""
import torch


def DataParallel(module, device_ids=None, output_device=None, dim=0, chunk_sizes=None):
    ids = list(device_ids) if device_ids is not None else list(range(max(1, torch.cuda.device_count())))
    if len(ids) <= 1:
        return module
    raise RuntimeError(
        "single-process DataParallel over %d GPUs is not provided: start one process per GPU, e.g. "
        "`python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 main.py "
        "polydet ...`; BaseTrainer.set_device then wraps the model in DistributedDataParallel (gradients "
        "averaged over ranks = the reference's mean of replica losses)" % (len(ids), len(ids)))
