"""Bucket farm plumbing shared by bench.py and the tests: how buckets / clouds are split over ranks and
device workers, and how per-rank timings and counts are combined.  No data-path collective exists on this
path (buckets are independent); torch.distributed only carries a barrier and two reductions.
"""


def worker_share(items, k, nworkers):
    """Buckets of device worker k of `nworkers` on one GPU: k, k + nworkers, ... (greedy in arrival order,
    like DeviceWorkerGroup's queue, src/worker_group.h)."""
    return items[k::nworkers]


def rank_share(items, rank, world):
    """Strong-scaling split of ONE bucket stream over ranks: contiguous runs, sizes differing by at most one."""
    n = len(items)
    base, extra = divmod(n, world)
    first = rank * base + min(rank, extra)
    return items[first:first + base + (1 if rank < extra else 0)]


def combine(elapsed_s, units, dist=None, device=None):
    """Whole-job figures from per-rank ones: the MAX of the elapsed time and the SUM of the units processed.
    `dist` is torch.distributed (already initialised) or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), int(units), 1
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    u = torch.tensor([int(units)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), int(u.item()), dist.get_world_size()


def throughput(units, elapsed_s):
    """Mega-units per second."""
    return units / elapsed_s / 1e6
