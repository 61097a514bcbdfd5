"""The N > 1 path of the bucket farm on CPU: two gloo ranks, no data-path collective."""
import os
import socket

import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mlsgpu_amd import farm, synth
        # weak scaling as bench.py does it: every rank generates ITS OWN cloud (seed offset = rank) ...
        cloud, grid = synth.make_cloud("cfg2", scale=0.0005, seed_offset=rank)
        allb, buckets = synth.bucketize(cloud, grid, 127)
        units = sum(b.cells for b in buckets)
        elapsed = 1.0 + rank                      # rank 1 is the slow one
        dist.barrier()
        t, u, w = farm.combine(elapsed, units, dist)
        # ... and the strong-scaling split of one stream covers every bucket exactly once
        mine = farm.rank_share(list(range(len(buckets))), rank, world)
        import torch
        cover = torch.zeros(len(buckets), dtype=torch.int64)
        cover[mine] = 1
        dist.all_reduce(cover)
        out.put((rank, t, u, w, units, float(cloud["position"][0][0]), int(cover.min()), int(cover.max())))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_farm():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, t0, u0, w0, own0, x0, cmin0, cmax0), (r1, t1, u1, w1, own1, x1, cmin1, cmax1) = res
    assert w0 == w1 == 2
    assert t0 == t1 == 2.0                       # MAX over ranks
    assert u0 == u1 == own0 + own1 == 2 * 255 ** 3   # SUM over ranks: whole-job units
    assert x0 != x1                               # different clouds per rank
    assert cmin0 == cmax0 == 1                    # rank_share is a partition


def test_shares():
    from mlsgpu_amd import farm
    items = list(range(27))
    assert sorted(sum((farm.worker_share(items, k, 2) for k in range(2)), [])) == items
    for world in (1, 2, 4, 8):
        parts = [farm.rank_share(items, r, world) for r in range(world)]
        assert sum(parts, []) == items
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert farm.combine(1.5, 10) == (1.5, 10, 1)
    assert farm.throughput(2_000_000, 2.0) == 1.0
