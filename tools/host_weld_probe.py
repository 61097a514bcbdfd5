#!/usr/bin/env python3
"""Runs ON THE GPU BOX: the host-weld leg of bench.py alone, on the shells cloud, through both routes (the welder's own landing
memory / the farm's pinned ring).  MLSGPU_HIP_WELDER_TRACE=1 adds the welder's own breakdown per job on stderr.
    python3 tools/host_weld_probe.py [--steps 5] [--weld-threads 0]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--weld-threads", type=int, default=0)
    p.add_argument("--farm-workers", type=int, default=2)
    p.add_argument("--farm-spare", type=int, default=6)
    p.add_argument("--copy-threads", type=int, default=16)
    p.add_argument("--staging-buffers", type=int, default=0)
    p.add_argument("--mesh-memory-mb", type=int, default=4096)
    p.add_argument("--routes", default="landing,ring,landing,ring")
    args = p.parse_args()
    import torch
    import mlsgpu_amd as m
    from mlsgpu_amd import farm as _farm, synth
    from benchlegs import host_weld_leg
    _farm.bind_process_to_device_node(0, before_hip=True)
    dev = torch.device("cuda", 0)
    cloud, g = synth.make_cloud_device("cfg3", dev, dist="shells")
    sb_t, buckets = synth.bucketize_device(cloud, synth.grid_buckets((g, g, g), 255))
    del cloud
    host = synth.to_host_splats(sb_t)
    smax = max(b.count for b in buckets)
    scells = max(max(b.num_vertices) for b in buckets) - 1
    svox = sum(b.cells for b in buckets)
    del sb_t
    torch.cuda.empty_cache()
    for route in args.routes.split(","):
        r = host_weld_leg(m, args, 0, host, buckets, smax, scells, svox, steps=args.steps, landing=route == "landing")
        print(json.dumps({"route": route, "ms_per_job": r["ms_per_step"], "pass_ms": r["pass_until_last_mesh_welded_ms"],
                          "finalize_ms": r["finalize_ms"], "streamed_ms": r["streamed"]["ms_per_step"],
                          "streamed_in_ms": r["streamed"]["stream_in_ms"], "streamed_wait_ms": r["streamed"]["wait_for_previous_finalize_ms"],
                          "streamed_finalize_ms": r["streamed"]["finalize_ms"], "ring_waits": r["ring_waits"]}), flush=True)


if __name__ == "__main__":
    main()
