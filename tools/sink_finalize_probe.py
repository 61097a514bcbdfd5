#!/usr/bin/env python3
"""Per-stage times of the device sink's finalize on one synthetic cloud (HIP events around every launch)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    dist = sys.argv[1] if len(sys.argv) > 1 else "shells"
    import torch

    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    device = torch.device("cuda", 0)
    cloud, g = synth.make_cloud_device("cfg3", device, scale=1.0, dist=dist)
    sb_t, buckets = synth.bucketize_device(cloud, synth.grid_buckets((g, g, g), 255))
    del cloud
    torch.cuda.synchronize()
    ctx = m.Context(0)
    smax = max(b.count for b in buckets)
    scells = max(max(b.num_vertices) for b in buckets) - 1
    w = m.Worker(ctx, smax, max_cells=scells, mesh_memory=4096 << 20)
    w.set_keep_splats(True)
    w.set_batch(4)
    buf = m.DeviceBuffer(ctx, nbytes=sb_t.numel() * 4, borrow=sb_t.data_ptr())
    sink = m.Mesher(ctx, 0.02)
    for rep in range(3):
        sink.reset()
        w.process_batch(buf, buckets, collector=sink.collector(ctx, 0))
        ctx.synchronize()
        ctx.reset_stats()
        ctx.set_timing(rep == 2)
        t0 = time.perf_counter()
        n = sink.finalize()
        ctx.synchronize()
        dt = time.perf_counter() - t0
        ctx.set_timing(False)
    st = {k: (round(v[0], 3), v[1]) for k, v in ctx.stats().items() if k.startswith("mesher")}
    print(json.dumps({"finalize_ms_instrumented": round(dt * 1e3, 2), "chunks": n, "stats": sink.stats(), "stages_ms_launches": st}))
    ctx.set_timing(False)
    for rep in range(3):
        sink.reset()
        w.process_batch(buf, buckets, collector=sink.collector(ctx, 0))
        ctx.synchronize()
        t0 = time.perf_counter()
        sink.finalize()
        ctx.synchronize()
        print("finalize alone ms", round((time.perf_counter() - t0) * 1e3, 2))


if __name__ == "__main__" and not (len(sys.argv) > 2 and sys.argv[2] == "sequence"):
    main()


def sequence():
    """finalize -> boundary -> finalize_with and add -> boundary -> finalize_with, timed, with the stage stats of each call."""
    dist = sys.argv[1] if len(sys.argv) > 1 else "shells"
    import numpy as np
    import torch

    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    device = torch.device("cuda", 0)
    cloud, g = synth.make_cloud_device("cfg3", device, scale=1.0, dist=dist)
    sb_t, buckets = synth.bucketize_device(cloud, synth.grid_buckets((g, g, g), 255))
    del cloud
    ctx = m.Context(0)
    smax = max(b.count for b in buckets)
    scells = max(max(b.num_vertices) for b in buckets) - 1
    w = m.Worker(ctx, smax, max_cells=scells, mesh_memory=4096 << 20)
    w.set_keep_splats(True)
    w.set_batch(2)
    buf = m.DeviceBuffer(ctx, nbytes=sb_t.numel() * 4, borrow=sb_t.data_ptr())
    sink = m.Mesher(ctx, 0.02)

    def timed(name, fn):
        ctx.synchronize()
        ctx.reset_stats()
        ctx.set_timing(True)
        t0 = time.perf_counter()
        out = fn()
        ctx.synchronize()
        dt = time.perf_counter() - t0
        ctx.set_timing(False)
        st = {k.replace("mesher.", "").replace(".time", ""): (round(v[0], 2), v[1]) for k, v in ctx.stats().items() if k.startswith("mesher")}
        print(name, round(dt * 1e3, 2), "ms", st)
        return out
    for rep in range(2):
        sink.reset()
        w.process_batch(buf, buckets, collector=sink.collector(ctx, 0))
        timed("finalize", sink.finalize)
        part = timed("boundary after finalize", sink.boundary)
        keep = np.ones(len(part[2]), np.uint8)
        timed("finalize_with after that", lambda: sink.finalize_with(keep))
        from mlsgpu_amd import dist_sink
        timed("boundary again (behind a finalize_with: nothing cached)", sink.boundary)
        timed("boundary a third time (behind a boundary)", sink.boundary)
        timed("boundary a fourth time", sink.boundary)
        timed("finalize again", sink.finalize)
        part = timed("boundary after finalize", sink.boundary)
        keep2, _ = dist_sink.merge_boundaries([part], 0.02)
        print("verdict keeps", int(keep2[0].sum()), "of", len(keep2[0]), keep2[0].dtype)
        timed("finalize_with (merged verdict)", lambda: sink.finalize_with(keep2[0]))
        sink.reset()
        w.process_batch(buf, buckets, collector=sink.collector(ctx, 0))
        part = timed("boundary (fresh)", sink.boundary)
        timed("finalize_with after that", lambda: sink.finalize_with(keep))


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "sequence":
    sequence()
