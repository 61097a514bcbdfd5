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


def _sink_rank_main(rank, world, port, out):
    """One rank of the distributed mesh sink: its share of ONE cloud's buckets (farm.rank_share) -> meshes (the CPU oracle
    stands in for the GPU pipeline here) -> this rank's HostMesher -> dist_sink.global_prune over gloo."""
    import sys
    import numpy as np
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        here = os.path.dirname(os.path.abspath(__file__))
        sys.path[:0] = [here, os.path.join(os.path.dirname(here), "oracle")]
        import mesher_oracle as mo
        import oracle_binding as ob
        import mlsgpu_amd as m
        from mlsgpu_amd import dist_sink, farm, synth
        cloud = synth.shells_cloud(60_000, 63.0, 12.0, 1.5, 2.5, seed=77)
        cloud = np.concatenate([cloud, synth.sphere_cloud(400, (8.0, 8.0, 8.0), 3.0, 1.0, 1.5, seed=3)])   # an island to prune
        allb, buckets = synth.bucketize(cloud, 64, 21)
        assert len(buckets) == 27
        mine = farm.rank_share(list(range(len(buckets))), rank, world)
        welder = m.HostMesher(0.05)
        ref = allb.copy()
        everything = []
        for i, b in enumerate(buckets):
            batches, _ = ob.bucket(ref, b.first, b.count, b.num_vertices, b.low, max_cells=63, max_swathe=64,
                                   mesh_memory=63 * 63 * 2 * 872)
            owner = next(r for r in range(world) if i in farm.rank_share(list(range(len(buckets))), r, world))
            for g in batches:
                ni = g["num_internal"]
                everything.append(dict(chunk=owner, vertices=g["vertices"], num_internal=ni, keys=g["keys"][ni:],
                                       triangles=g["triangles"]))
                if i in mine:
                    welder.add(rank, g["vertices"], ni, g["keys"][ni:], g["triangles"])
        n, stats = dist_sink.global_prune(welder, 0.05, dist)
        exp, exp_stats = mo.mesh_sink(everything, 0.05)           # the whole job in one process, one chunk per rank
        ok = all(stats[k] == exp_stats[k] for k in exp_stats)
        exp_mine = [(v, t) for c, v, t in exp if c == rank]
        if exp_mine:
            _, v, t = welder.chunk(0)
            ok = ok and n == 1 and mo.isomorphic(v, t, *exp_mine[0])
        else:
            ok = ok and n == 0
        out.put((rank, ok, stats["components"], stats["kept_components"], stats["total_vertices"],
                 sum(len(mm["vertices"]) for mm in everything)))
    finally:
        dist.destroy_process_group()


def test_two_rank_distributed_sink():
    """The N > 1 mesh sink on CPU: two gloo ranks, each welding its share of one cloud's 27 buckets, one all-gather of
    the boundary; components crossing the rank seam are united, the island falls under the global threshold, and every
    rank's output equals the single-process oracle's chunk for it."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sink_rank_main, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert res[0][2:] == res[1][2:]                 # both ranks computed the same whole-job statistics
    assert res[0][3] < res[0][2]                    # something was pruned
    assert res[0][4] < res[0][5]                    # shared vertices were counted once


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


def _verdict_main(rank, world, port, out, wrong_rank):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json
        from mlsgpu_amd import farm, synth
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        pins = json.load(open(os.path.join(root, "tests", "golden", "cfg4slab_uniform.json")))["slabs"]
        pin = pins[str(rank)][synth.slab_variant(world, rank)]
        digest = pin["digest"] if rank != wrong_rank else "0123456789abcdef"
        v = farm.slab_verdicts(pins, world, rank, digest, pin["vertices"], pin["triangles"], dist)
        out.put((rank, v))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("wrong_rank", [-1, 1])
def test_every_rank_checks_its_own_slab(wrong_rank):
    """bench.py --gpus N: rank r holds its digest against slab r's pin (inner slab, or the job's last), the verdicts are
    gathered: every rank sees every rank's verdict, and one wrong slab anywhere fails the run on all of them."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_verdict_main, args=(r, 2, port, out, wrong_rank)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1]                                     # the same picture on every rank
    v = res[0]
    assert v["ok"] == [True, wrong_rank != 1] and v["all_ok"] == (wrong_rank < 0)
    assert v["expected"][0] != v["expected"][1] and None not in v["expected"]      # slab 0 inner, slab 1 as the last slab
    assert v["digests"][0] == v["expected"][0]
    assert (v["digests"][1] == v["expected"][1]) == (wrong_rank != 1)


def test_slab_variants_cover_every_job_size():
    """Every (slab, variant) an N = 1 / 2 / 4 / 8 job contains is pinned."""
    import json
    from mlsgpu_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pins = json.load(open(os.path.join(root, "tests", "golden", "cfg4slab_uniform.json")))["slabs"]
    for world in (1, 2, 4, 8):
        for rank in range(world):
            pin = pins[str(rank)][synth.slab_variant(world, rank)]
            assert len(pin["digest"]) == 16 and pin["triangles"] > 0 and pin["shipouts"] >= 25
