"""ModelWithLoss / BaseTrainer (reference: src/lib/trains/base_trainer.py:14-149).

Same surface: BaseTrainer(opt, model, optimizer), .set_device(gpus, chunk_sizes,
device), .train(epoch, loader) / .val(epoch, loader) -> (stats dict incl. 'time',
results).  Multi-GPU is ONE PROCESS PER GPU: when torch.distributed is
initialised (backend "nccl" = RCCL over xGMI) the model is wrapped in
DistributedDataParallel -- gradients are averaged by bucketed all-reduces that
overlap backward -- instead of the reference's single-process DataParallel
(per-step parameter broadcast, output gather and gradient reduce onto GPU 0).
Averaging gradients across ranks == the reference's mean of per-replica losses
(base_trainer.py:95); BatchNorm statistics stay per rank, rank 0's are saved.
"""
import os
import time

import torch
import torch.distributed as dist

from ..models.networks import conv3x3
from ..utils.utils import AverageMeter


class ModelWithLoss(torch.nn.Module):
    def __init__(self, model, loss):
        super(ModelWithLoss, self).__init__()
        self.model = model
        self.loss = loss

    def forward(self, batch):
        outputs = self.model(batch["input"])
        loss, loss_stats = self.loss(outputs, batch)
        return outputs[-1], loss, loss_stats


class BaseTrainer(object):
    def __init__(self, opt, model, optimizer=None):
        self.opt = opt
        self.optimizer = optimizer
        self.loss_stats, self.loss = self._get_losses(opt)
        self.model_with_loss = ModelWithLoss(model, self.loss)
        self._ddp = None

    def set_device(self, gpus, chunk_sizes, device):
        self.model_with_loss = self.model_with_loss.to(device)
        force = bool(int(os.environ.get("CP_FORCE_DDP", "0")))      # tests: wrap even at world_size 1
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force):
            ids = [device.index] if getattr(device, "type", "cpu") == "cuda" else None
            self._ddp = torch.nn.parallel.DistributedDataParallel(
                self.model_with_loss, device_ids=ids, broadcast_buffers=False,
                bucket_cap_mb=getattr(self.opt, "bucket_cap_mb", 32), gradient_as_bucket_view=True)
        if self.optimizer is not None:
            for state in self.optimizer.state.values():
                for k, v in state.items():
                    if isinstance(v, torch.Tensor):
                        state[k] = v.to(device=device, non_blocking=True)

    def step(self, batch, train=True):
        """One iteration of the hot loop (base_trainer.py:87-102) on a device batch."""
        module = self._ddp if (train and self._ddp is not None) else self.model_with_loss
        output, loss, loss_stats = module(batch)
        loss = loss.mean()
        if train:
            self.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            if getattr(self.opt, "clip", False):
                torch.nn.utils.clip_grad_norm_(self.model_with_loss.parameters(),
                                               float(self.opt.clip_value))
            self.optimizer.step()
            conv3x3.refresh_weight_bank()        # the MFMA kernels' weight forms for the next step, in one launch
        return output, loss, loss_stats

    def run_epoch(self, phase, epoch, data_loader):
        train = phase == "train"
        self.model_with_loss.train(train)
        opt = self.opt
        results = {}
        data_time, batch_time = AverageMeter(), AverageMeter()
        avg = {l: AverageMeter() for l in self.loss_stats}
        num_iters = len(data_loader) if opt.num_iters < 0 else opt.num_iters
        t0 = end = time.time()
        for iter_id, batch in enumerate(data_loader):
            if iter_id >= num_iters:
                break
            data_time.update(time.time() - end)
            for k in batch:
                if k != "meta":
                    batch[k] = batch[k].to(device=opt.device, non_blocking=True)
            batch = self.prepare_batch(batch)
            with torch.set_grad_enabled(train):
                output, loss, loss_stats = self.step(batch, train)
            batch_time.update(time.time() - end)
            end = time.time()
            for l in avg:                       # the only device->host syncs of the loop
                avg[l].update(loss_stats[l].mean().item(), batch["input"].size(0))
            if getattr(opt, "print_iter", 0) > 0 and iter_id % opt.print_iter == 0:
                msg = " ".join("|{} {:.4f}".format(l, avg[l].avg) for l in avg)
                print("{}/{}| {}: [{}][{}/{}] {} |Data {:.3f}s |Net {:.3f}s".format(
                    opt.task, opt.exp_id, phase, epoch, iter_id, num_iters, msg, data_time.avg,
                    batch_time.avg))
            if getattr(opt, "test", False) or (opt.dataset == "cityscapes" and phase == "val"):
                self.save_result(output, batch, results)
            del output, loss, loss_stats
        ret = {k: v.avg for k, v in avg.items()}
        ret["time"] = (time.time() - t0) / 60.0
        return ret, results

    def prepare_batch(self, batch):
        """Hook between the batch upload and the step (device-side target construction)."""
        return batch

    def debug(self, batch, output, iter_id):
        raise NotImplementedError

    def save_result(self, output, batch, results):
        raise NotImplementedError

    def _get_losses(self, opt):
        raise NotImplementedError

    def val(self, epoch, data_loader):
        return self.run_epoch("val", epoch, data_loader)

    def train(self, epoch, data_loader):
        return self.run_epoch("train", epoch, data_loader)
