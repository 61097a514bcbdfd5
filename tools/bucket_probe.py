"""Runs ON THE GPU BOX: Bucket::bucket on the device (mlsgpu_hip_bucket) alone, on the cfg5 cloud generated in HBM --
wall time per pass and a digest of the leaves (extents + sizes + a checksum of every member list), so that builds can be
compared (tools/ab_env.sh-style: run it under rocprofv3 --kernel-trace --stats for the per-kernel times).
usage: python3 tools/bucket_probe.py [--scale 1.0] [--passes 3] [--cfg cfg5]"""
import argparse
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--cfg", default="cfg5")
    p.add_argument("--scale", type=float, default=1.0)
    p.add_argument("--passes", type=int, default=3)
    p.add_argument("--max-splats", type=int, default=2097152)
    a = p.parse_args()
    import torch
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as mb, synth
    dev = torch.device("cuda:0")
    cloud, g = synth.make_cloud_device(a.cfg, dev, scale=a.scale)
    torch.cuda.synchronize()
    n = cloud.shape[0]
    ctx = m.Context(0)
    raw = m.DeviceBuffer(ctx, nbytes=cloud.numel() * 4, borrow=cloud.data_ptr())
    ext = (0, g - 1, 0, g - 1, 0, g - 1)
    bp = dict(max_splats=a.max_splats, max_cells=255, chunk_cells=0, micro_cells=63, max_split=1 << 30)
    h = hashlib.sha256()
    # the checked pass: every member list copied to the host
    leaves = mb.bucket_cloud(ctx, raw, n, (0.0, 0.0, 0.0), 1.0, ext, **bp)
    pairs = 0
    for l in leaves:
        h.update(repr((tuple(l["extents"]), tuple(l["chunk"]), l["depth"], l["num_splats"])).encode())
        h.update(l["ids"].tobytes())
        pairs += l["num_splats"]
        l["ids"] = None
    t0 = time.perf_counter()
    for _ in range(a.passes):
        mb.bucket_cloud(ctx, raw, n, (0.0, 0.0, 0.0), 1.0, ext, on_bucket=lambda l, ids: None, **bp)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / a.passes
    print("splats %d leaves %d pairs %d  %.2f ms per pass  %.0f Msplats/s  leaves-digest %s"
          % (n, len(leaves), pairs, dt * 1e3, n / dt / 1e6, h.hexdigest()[:16]))


if __name__ == "__main__":
    main()
