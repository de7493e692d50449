#!/usr/bin/env python3
"""Training driver with the reference's CLI and control flow (src/main.py:24-198):
`python main.py polydet --arch dla_34 --batch_size 8 --gpus 0 ...`.
Multi-GPU is one process per GPU: launch with
`python -m torch.distributed.run --nproc-per-node N main.py polydet ...` (RCCL all-reduce);
`--batch_size` is then the global batch, split evenly across ranks."""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist
import torch.utils.data

from centerpoly_amd.datasets.dataset_factory import get_dataset
from centerpoly_amd.models.model import create_model, load_model, save_model
from centerpoly_amd.opts import opts
from centerpoly_amd.trains.train_factory import train_factory


def main(opt):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.manual_seed(opt.seed)
    Dataset = get_dataset(opt.dataset, opt.task)
    opt = opts().update_dataset_info_and_set_heads(opt, Dataset)
    if opt.gpus[0] < 0:
        raise SystemExit("centerpoly_amd trains on HIP devices only (--gpus -1 has no CPU path)")
    torch.cuda.set_device(local_rank)
    opt.device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=opt.device)

    print("Creating model...")
    model = create_model(opt.arch, opt.heads, opt.head_conv)
    optimizer = torch.optim.Adam(model.parameters(), opt.lr)
    start_epoch = 0
    if opt.load_model != "":
        model, optimizer, start_epoch = load_model(model, opt.load_model, optimizer, opt.resume, opt.lr,
                                                   opt.lr_step)
    trainer = train_factory[opt.task](opt, model, optimizer)
    trainer.set_device(opt.gpus, opt.chunk_sizes, opt.device)

    per_rank = max(1, opt.batch_size // world)
    if world > 1 and len(opt.chunk_sizes) != world:
        # opts derives chunk_sizes from --gpus (reference: opts.py:301-310); under torchrun the rank
        # count is the world size
        from centerpoly_amd.opts import chunk_sizes_for
        opt.chunk_sizes = chunk_sizes_for(opt.batch_size, opt.master_batch_size_arg, world)
    val_loader = torch.utils.data.DataLoader(Dataset(opt, "val"), batch_size=1, shuffle=False,
                                             num_workers=1, pin_memory=False)
    if opt.test:
        _, preds = trainer.val(0, val_loader)
        val_loader.dataset.run_eval(preds, opt.save_dir)
        return
    train_set = Dataset(opt, "train")
    sampler = None
    loader_kw = dict(num_workers=opt.num_workers,
                     # pageable batches: 100 MB reach the GPU in 1.9 ms, while host writes into pinned
                     # staging memory followed by non-blocking copies showed periodic ~90 ms stalls on
                     # MI355X (tools/probe_stalls.py); the reference pins (main.py:59)
                     pin_memory=False)
    if world > 1:
        # rank r takes chunk_sizes[r] samples of every global batch (even split unless
        # --master_batch_size says otherwise), no data collective
        from centerpoly_amd.utils.sampler import ChunkedDistributedSampler
        sampler = ChunkedDistributedSampler(len(train_set), opt.chunk_sizes, rank, shuffle=True, seed=opt.seed)
        train_loader = torch.utils.data.DataLoader(train_set, batch_sampler=sampler, **loader_kw)
    else:
        train_loader = torch.utils.data.DataLoader(train_set, batch_size=per_rank, shuffle=True, drop_last=True,
                                                   **loader_kw)
    if rank == 0:
        os.makedirs(opt.save_dir, exist_ok=True)
    print("Starting training...")
    best = 1e10
    for epoch in range(start_epoch + 1, opt.num_epochs + 1):
        mark = epoch if opt.save_all else "last"
        if sampler is not None:
            sampler.set_epoch(epoch)
        log_dict_train, _ = trainer.train(epoch, train_loader)
        print("epoch: {} |".format(epoch) + "".join("{} {:8f} | ".format(k, v) for k, v in log_dict_train.items()))
        if rank == 0:
            if opt.val_intervals > 0 and epoch % opt.val_intervals == 0:
                save_model(os.path.join(opt.save_dir, "model_{}.pth".format(mark)), epoch, model, optimizer)
                with torch.no_grad():
                    log_dict_val, preds = trainer.val(epoch, val_loader)
                val_loader.dataset.run_eval(preds, opt.save_dir)
                if log_dict_val[opt.metric] < best:
                    best = log_dict_val[opt.metric]
                    save_model(os.path.join(opt.save_dir, "model_best.pth"), epoch, model)
            else:
                save_model(os.path.join(opt.save_dir, "model_last.pth"), epoch, model, optimizer)
        if epoch in opt.lr_step:
            if rank == 0:
                save_model(os.path.join(opt.save_dir, "model_{}.pth".format(epoch)), epoch, model, optimizer)
            lr = opt.lr * (0.1 ** (opt.lr_step.index(epoch) + 1))
            print("Drop LR to", lr)
            for group in optimizer.param_groups:
                group["lr"] = lr
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(opts().parse())
