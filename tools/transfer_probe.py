#!/usr/bin/env python3
"""SURVEY 8(d)'s region (host splats in -> every ship-out back in host memory through the ring) on ONE cloud, for A/B runs
of the copy side: where the process and its memory live, how many copy threads, with or without the host welder.

    python tools/transfer_probe.py [--dist shells] [--steps 5] [--bind node|none|other] [--copy-threads 8] [--weld]

--bind node   the whole process (every thread, every first touch) on the CPUs of the GPU's NUMA node, before anything is
              allocated (what `numactl --cpunodebind` does); other = the OTHER node (the worst case); none = wherever
              the scheduler puts it.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def node_cpus(node):
    out = []
    for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
        lo, _, hi = part.partition("-")
        out += range(int(lo), int(hi or lo) + 1)
    return out


def gpu_node(index=0):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    buf = ctypes.create_string_buffer(64)
    if hip.hipDeviceGetPCIBusId(buf, 64, index) != 0:
        return -1
    try:
        return int(open("/sys/bus/pci/devices/%s/numa_node" % buf.value.decode().lower()).read())
    except OSError:
        return -1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--dist", default="shells")
    ap.add_argument("--bind", default="none", choices=["none", "node", "other"])
    ap.add_argument("--copy-threads", type=int, default=8)
    ap.add_argument("--farm-workers", type=int, default=4)
    ap.add_argument("--weld", action="store_true")
    ap.add_argument("--weld-threads", type=int, default=0)
    ap.add_argument("--farm-batch", type=int, default=1)
    ap.add_argument("--spare", type=int, default=1)
    ap.add_argument("--staging", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--pin-gb", type=float, default=0.0, help="pinned host memory allocated (and kept) before the legs, as bench.py's landing buffers are")
    ap.add_argument("--preload-s", type=float, default=0.0, help="seconds of resident-input passes on the GPU before the legs")
    a = ap.parse_args()
    import torch  # noqa: F401  (its HIP runtime first, see tests/conftest.py)
    node = gpu_node(0)
    nodes = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
    bound = None
    if a.bind != "none" and node >= 0 and nodes > 1:
        bound = node if a.bind == "node" else (node + 1) % nodes
        os.sched_setaffinity(0, node_cpus(bound))
    sys.argv = [sys.argv[0], "--copy-threads", str(a.copy_threads), "--farm-workers", str(a.farm_workers), "--farm-spare", str(a.spare),
                "--staging-buffers", str(a.staging), "--weld-threads", str(a.weld_threads), "--farm-batch", str(a.farm_batch)]
    import bench
    args = bench.parse_args()
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    device = torch.device("cuda", 0)
    cloud, g = synth.make_cloud_device("cfg3", device, scale=1.0, dist=a.dist)
    sb_t, sbuckets = synth.bucketize_device(cloud, synth.grid_buckets((g, g, g), 255))
    del cloud
    torch.cuda.synchronize()
    smax = max(b.count for b in sbuckets)
    scells = max(max(b.num_vertices) for b in sbuckets) - 1
    svox = sum(b.cells for b in sbuckets)
    host = synth.to_host_splats(sb_t)
    del sb_t
    torch.cuda.empty_cache()
    pins = []
    if a.pin_gb > 0:
        for _ in range(3):
            pins.append(m.binding.PinnedBuffer(int(a.pin_gb * (1 << 30) / 3)))
    if a.preload_s > 0:
        import time
        ctx = m.Context(0)
        dbuf = m.DeviceBuffer(ctx, array=host)
        w = m.Worker(ctx, smax, max_cells=scells, mesh_memory=4096 << 20)
        w.set_keep_splats(True)
        col = m.binding.SizeCollector()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < a.preload_s:
            for b in sbuckets:
                w.process(dbuf, b.first, b.count, b.low, b.num_vertices, collector=col)
        ctx.synchronize()
        del w, dbuf
    res = {"gpu_node": node, "bound_to": bound, "pin_gb": a.pin_gb, "preload_s": a.preload_s, "copy_threads": a.copy_threads, "spare": a.spare, "staging": a.staging,
           "farm_workers": a.farm_workers, "runs": []}
    for _ in range(a.repeat):
        out = bench.transfer_legs(m, args, 0, host, sbuckets, smax, scells, svox, a.steps, with_sink=False)
        r = {k: v for k, v in out["shipouts"].items() if k not in ("note", "placement", "unit", "h2d_GB_per_step", "d2h_GB_per_step")}
        if a.weld:
            hw = bench.host_weld_leg(m, args, 0, host, sbuckets, smax, scells, svox, steps=a.steps)
            r["host_weld"] = {k: v for k, v in hw.items() if k != "note"}
        res["runs"].append(r)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
