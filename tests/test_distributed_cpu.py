"""world_size-2 gloo tests (CPU) of the multi-GPU path: one process per rank, gradients
averaged by all-reduce == the reference's mean of per-replica losses
(src/lib/trains/base_trainer.py:95 with models/data_parallel.py), BatchNorm stats per rank."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _ToyLoss(torch.nn.Module):
    def forward(self, outputs, batch):
        loss = ((outputs[-1]["y"] - batch["t"]) ** 2).mean()
        return loss, {"loss": loss}


class _ToyNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(3, 4, 3, padding=1)
        self.bn = torch.nn.BatchNorm2d(4)

    def forward(self, x):
        return [{"y": self.bn(self.conv(x))}]


def _make_trainer(opt):
    from centerpoly_amd.trains.base_trainer import BaseTrainer

    class Toy(BaseTrainer):
        def _get_losses(self, opt):
            return ["loss"], _ToyLoss()

    torch.manual_seed(0)
    net = _ToyNet()
    return Toy(opt, net, torch.optim.SGD(net.parameters(), lr=0.1)), net


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Opt:
        clip = False
        bucket_cap_mb = 1
    trainer, net = _make_trainer(Opt())
    trainer.set_device([0, 1], [2, 2], torch.device("cpu"))
    assert trainer._ddp is not None
    g = torch.Generator().manual_seed(100)
    xs = torch.randn(4, 3, 8, 8, generator=g)
    ts = torch.randn(4, 4, 8, 8, generator=g)
    sl = slice(2 * rank, 2 * rank + 2)                       # shard by image, no data collective
    trainer.model_with_loss.train()
    trainer.step({"input": xs[sl], "t": ts[sl]}, train=True)
    q.put((rank, {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_mean_of_replica_losses():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    got = {r: {k: torch.from_numpy(v) for k, v in d.items()} for r, d in got.items()}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # parameters identical on both ranks after the step
    for k in got[0]:
        if "running" in k or "num_batches" in k:
            continue
        assert torch.allclose(got[0][k], got[1][k], atol=1e-7), k
    # BatchNorm statistics are per rank (no SyncBN), like the reference's replicas
    assert not torch.allclose(got[0]["bn.running_mean"], got[1]["bn.running_mean"])

    # single-process restatement: mean of the two replica losses, each replica normalising
    # its own BatchNorm batch
    class Opt:
        clip = False
    trainer, net = _make_trainer(Opt())
    g = torch.Generator().manual_seed(100)
    xs = torch.randn(4, 3, 8, 8, generator=g)
    ts = torch.randn(4, 4, 8, 8, generator=g)
    net.train()
    losses = []
    for r in range(2):
        sl = slice(2 * r, 2 * r + 2)
        out = net(xs[sl])
        losses.append(((out[-1]["y"] - ts[sl]) ** 2).mean())
    trainer.optimizer.zero_grad()
    (sum(losses) / 2).backward()
    trainer.optimizer.step()
    for k, v in net.state_dict().items():
        if "running" in k or "num_batches" in k:
            continue
        assert torch.allclose(v, got[0][k], atol=1e-6), k


def test_bench_shards_batches_per_rank():
    """bench.py's training leg draws a different synthetic shard per rank (no data collective)."""
    from centerpoly_amd import synth
    a = synth.train_batch(1, 8, 8, stream="bench/train/rank0", in_h=32, in_w=32)
    b = synth.train_batch(1, 8, 8, stream="bench/train/rank1", in_h=32, in_w=32)
    assert not (a["input"] == b["input"]).all()
    a2 = synth.train_batch(1, 8, 8, stream="bench/train/rank0", in_h=32, in_w=32)
    assert (a["input"] == a2["input"]).all()
