#!/usr/bin/env python3
"""The two transfer-inclusive routes of SURVEY 8(d) on the shells cloud alone (bench.py's transfer_legs), for A/B runs of
the device-sink route: python tools/sink_route_probe.py [--steps 3]  (environment: MLSGPU_BENCH_SINK_PIECE_MB, ...)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--dist", default="shells")
    ap.add_argument("--rotation", type=int, default=3)
    a = ap.parse_args()
    sys.argv = [sys.argv[0], "--sink-rotation", str(a.rotation)]
    args = bench.parse_args()
    import torch

    import mlsgpu_amd as m
    from mlsgpu_amd import farm as fm, synth
    fm.bind_process_to_device_node(0)
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
    out = bench.transfer_legs(m, args, 0, host, sbuckets, smax, scells, svox, a.steps)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk not in ("note", "placement")} for k, v in out.items()}))


if __name__ == "__main__":
    main()
