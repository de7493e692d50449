"""Per-rank index sampler that honours the reference's uneven split of a global batch.

The reference scatters a batch of `--batch_size` over its replicas in `chunk_sizes` =
[`--master_batch_size`, rest spread over the others] (src/lib/opts.py:301-310,
src/lib/models/scatter_gather.py:6-25) and averages the replica losses with equal weight
(src/lib/trains/base_trainer.py:95).  With one process per GPU, rank r therefore has to see
chunk_sizes[r] samples of every global batch -- DistributedSampler only does equal shares.  Averaging
the gradients over ranks (DDP) then reproduces the reference's equal-weight mean of replica losses
whatever the chunk sizes."""
import torch
import torch.utils.data


class ChunkedDistributedSampler(torch.utils.data.Sampler):
    """Yields BATCHES (lists of indices) for rank `rank`: each global batch of sum(chunk_sizes) shuffled
    indices is cut at the chunk boundaries; incomplete trailing batches are dropped.  Use as
    `DataLoader(dataset, batch_sampler=ChunkedDistributedSampler(...))`."""

    def __init__(self, dataset_len, chunk_sizes, rank, shuffle=True, seed=0):
        if not 0 <= rank < len(chunk_sizes):
            raise ValueError("rank %d outside chunk_sizes %r" % (rank, chunk_sizes))
        if min(chunk_sizes) < 1:
            raise ValueError("every rank needs at least one sample per step (chunk_sizes %r)" % (chunk_sizes,))
        self.n = int(dataset_len)
        self.chunk_sizes = [int(c) for c in chunk_sizes]
        self.rank = rank
        self.shuffle = shuffle
        self.seed = seed
        self.epoch = 0
        self.global_batch = sum(self.chunk_sizes)
        self.start = sum(self.chunk_sizes[:rank])

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.n // self.global_batch

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()       # identical on every rank
        else:
            order = list(range(self.n))
        for b in range(len(self)):
            lo = b * self.global_batch + self.start
            yield order[lo:lo + self.chunk_sizes[self.rank]]
