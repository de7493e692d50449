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


# ---- the real trainer plumbing around a CPU-runnable stand-in of DLASeg ------------------------------
class _StandInNet(torch.nn.Module):
    """Shaped like DLASeg where it matters for data parallelism: a trunk with BatchNorm, a DEAD branch
    whose parameters never receive gradients (the reference's Tree.forward leaves
    base.level{3,4}.project unused, pose_dla_dcn.py:206-213; the real model freezes them) but whose
    BatchNorm statistics still update, and one output dict per stack with the four polydet heads."""

    def __init__(self):
        super().__init__()
        self.trunk = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8),
                                         torch.nn.ReLU())
        self.project = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 1, bias=False), torch.nn.BatchNorm2d(8))
        for p in self.project.parameters():
            p.requires_grad_(False)
        self.heads = torch.nn.ModuleDict({h: torch.nn.Conv2d(8, c, 1) for h, c in
                                          (("hm", 8), ("poly", 32), ("pseudo_depth", 1), ("reg", 2))})

    def forward(self, x):
        f = self.trunk(x)
        with torch.no_grad():
            self.project(x)                                  # dead branch: statistics only
        return [{h: m(f) for h, m in self.heads.items()}]


class _DenseStandInLoss(torch.nn.Module):
    """CPU stand-in of PolydetLoss (the real one is HIP-only): same outputs -> (loss, stats with the
    trainer's loss_stats keys), each replica normalising by its own batch."""

    def forward(self, outputs, batch):
        o = outputs[-1]
        hm_l = ((torch.sigmoid(o["hm"]) - batch["hm"]) ** 2).mean()
        poly_l = o["poly"].abs().mean()
        depth_l = o["pseudo_depth"].abs().mean()
        off_l = o["reg"].abs().mean()
        loss = hm_l + poly_l + 0.1 * depth_l + off_l
        return loss, {"loss": loss, "hm_l": hm_l, "off_l": off_l, "poly_l": poly_l, "depth_l": depth_l}


def _polydet_like_trainer(opt):
    from centerpoly_amd.trains.polydet import PolydetTrainer

    class CpuPolydetTrainer(PolydetTrainer):
        def _get_losses(self, opt):
            return ["loss", "hm_l", "off_l", "poly_l", "depth_l"], _DenseStandInLoss()

    torch.manual_seed(1)
    net = _StandInNet()
    opt_ = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-2)
    return CpuPolydetTrainer(opt, net, opt_), net


def _worker2(rank, world, port, q, chunk_sizes, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from centerpoly_amd.models.model import save_model
    from centerpoly_amd.utils.sampler import ChunkedDistributedSampler

    class Opt:
        clip = False
        bucket_cap_mb = 1
        num_stacks = 1
    trainer, net = _polydet_like_trainer(Opt())
    trainer.set_device([0, 1], chunk_sizes, torch.device("cpu"))
    assert trainer._ddp is not None, "DDP wrapper missing"
    g = torch.Generator().manual_seed(5)
    n = 4 * sum(chunk_sizes)
    xs, ts = torch.randn(n, 3, 8, 8, generator=g), torch.rand(n, 8, 8, 8, generator=g)
    sampler = ChunkedDistributedSampler(n, chunk_sizes, rank, shuffle=True, seed=3)
    trainer.model_with_loss.train()
    seen = []
    for idx in sampler:                                      # 4 steps; frozen params must not stall the reducer
        assert len(idx) == chunk_sizes[rank]
        seen += idx
        trainer.step({"input": xs[idx], "hm": ts[idx]}, train=True)
    if rank == 0:
        save_model(os.path.join(tmp, "model_last.pth"), 1, trainer.model_with_loss.model, trainer.optimizer)
    q.put((rank, seen, {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_polydet_trainer_with_frozen_branch_and_uneven_chunks(tmp_path):
    """world_size 2, chunk sizes 3 + 5 of a global batch of 8 (--master_batch_size 3): DDP neither hangs
    on the frozen branch nor lets the ranks drift; BatchNorm statistics (live AND dead branch) stay per
    rank; the shards are disjoint and complete; rank 0's buffers are what save_model writes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    chunks = [3, 5]
    procs = [ctx.Process(target=_worker2, args=(r, 2, port, q, chunks, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got = {r: {k: torch.from_numpy(v) for k, v in d.items()} for r, _, d in res}
    seen = {r: s for r, s, _ in res}
    assert not set(seen[0]) & set(seen[1]) and len(seen[0]) == 12 and len(seen[1]) == 20
    assert sorted(seen[0] + seen[1]) == list(range(32))
    for k in got[0]:
        if "running" in k or "num_batches" in k:
            continue
        assert torch.equal(got[0][k], got[1][k]), "ranks drifted apart on " + k
    assert not torch.allclose(got[0]["trunk.1.running_mean"], got[1]["trunk.1.running_mean"])
    assert not torch.allclose(got[0]["project.1.running_mean"], got[1]["project.1.running_mean"])
    ck = torch.load(os.path.join(str(tmp_path), "model_last.pth"), map_location="cpu")
    assert set(ck) == {"epoch", "state_dict", "optimizer"}
    for k, v in ck["state_dict"].items():                    # un-prefixed keys, rank 0's tensors
        assert torch.equal(v, got[0][k]), k


def test_chunked_sampler_cuts_the_reference_chunk_sizes():
    from centerpoly_amd.opts import chunk_sizes_for
    from centerpoly_amd.utils.sampler import ChunkedDistributedSampler
    sizes = chunk_sizes_for(32, 2, 4)                        # --batch_size 32 --master_batch_size 2 on 4 GPUs
    assert sizes == [2, 10, 10, 10]
    per_rank = [list(ChunkedDistributedSampler(100, sizes, r, shuffle=True, seed=1)) for r in range(4)]
    assert all(len(b) == 3 for b in per_rank)                # 100 // 32 global batches, tail dropped
    for step in range(3):
        glob = sum((per_rank[r][step] for r in range(4)), [])
        assert [len(per_rank[r][step]) for r in range(4)] == sizes and len(set(glob)) == 32
