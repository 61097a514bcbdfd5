#!/usr/bin/env python3
"""The reference partition of cfg3 through the farm (bench.py's device_partition leg) alone, for rocprofv3:
   rocprofv3 --kernel-trace --stats -d /tmp/pp -o run -- python3 tools/partition_pipeline.py [--passes 5] [--lanes 8]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--lanes", type=int, default=8)
    ap.add_argument("--workers", type=int, default=2)
    a = ap.parse_args()
    import torch
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as mb, farm as fm, synth
    dev = torch.device("cuda", 0)
    cloud, grid = synth.make_cloud_device("cfg3", dev)
    n = len(cloud)
    ctx = m.Context(0)
    raw = m.DeviceBuffer(ctx, nbytes=cloud.numel() * 4, borrow=cloud.data_ptr())
    ext = (0, grid - 1) * 3
    bp = dict(max_splats=2097152, max_cells=255, chunk_cells=0, micro_cells=63, max_split=1 << 30)
    leaves = mb.bucket_cloud(ctx, raw, n, (0.0, 0.0, 0.0), 1.0, ext, on_bucket=lambda leaf, ids: None, **bp)
    pmax = max(l["num_splats"] for l in leaves)
    pcells = max(max(l["extents"][2 * i + 1] - l["extents"][2 * i] for i in range(3)) for l in leaves)
    farm = m.BucketFarm([0], pmax, workers_per_device=a.workers, spare=a.lanes * a.workers, max_cells=pcells, mesh_memory=4096 << 20)
    farm.set_batch(a.lanes)
    fm.partition_to_farm(ctx, farm, 0, raw, n, (0.0, 0.0, 0.0), 1.0, ext, bp)
    farm.finish()
    t0 = time.perf_counter()
    for _ in range(a.passes):
        fm.partition_to_farm(ctx, farm, 0, raw, n, (0.0, 0.0, 0.0), 1.0, ext, bp)
        farm.finish()
    dt = (time.perf_counter() - t0) / a.passes
    wc = farm.worker_clock()
    print("partition pipeline: %.2f ms per pass, %d buckets, %.2f buckets per set of launches"
          % (dt * 1e3, len(leaves), wc["buckets"] / max(wc["launch_sets"], 1)))
    farm.close()


if __name__ == "__main__":
    main()
