#!/usr/bin/env python3
"""Headline benchmark: Mvoxels/s evaluated + triangulated by the per-bucket device pipeline.

    python bench.py --gpus N --steps K --warmup W

One "step" is one pass of the hot path (octree build -> MLS corner evaluation -> marching tetrahedra with welding ->
scale/bias) over every bucket of ONE synthetic splat cloud whose splats are already resident in HBM.

N = 1   BASELINE.json configs[2]: 512^3 grid, 50 M uniform-random splats, 27 buckets -- the configuration the
        north_star target is quoted on.
N > 1   ONE sharded cloud: BASELINE.json configs[3] (1024^3 grid, 200 M uniform-random splats) cut into z-slabs of
        128 corner slices, one slab of 25 buckets (<= 205 x 205 x 128 cells) per GPU; at N = 8 the slabs are the
        whole of cfg4, at N = 2 / 4 the first N slabs of the same cloud.  One process per GPU (RANK / LOCAL_RANK /
        WORLD_SIZE from torch.distributed.run); every rank generates the cloud in its own HBM and keeps the splats of
        its slab (with halo).  No data-path collective: buckets are independent; torch.distributed
        carries the barrier and the reductions of the timing.  Per-GPU work is fixed (about 134 M voxels, as in
        cfg3): "scaling": "weak".
        `python bench.py --gpus N` without a launcher starts the N ranks itself (fresh child processes, before this
        process has touched a GPU) and exits with their status.

Per GPU, `--workers` device worker threads (the reference's --device-threads, src/mlsgpu_core.cpp:114) each own a stream,
an octree, an MLS functor and a Marching instance and take alternate buckets, so one worker's host synchronisations
overlap another's kernels.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").  Beside the headline it carries: the roofline of the
dominant kernel (durations measured live with HIP events on the worker's stream), a digest of every mesh the pipeline
produced (checked against the value pinned in tests/test_gpu_configs.py for the default workload), the
transfer-inclusive figures of SURVEY.md 8(d) (host splats in -> last mesh byte out) on both synthetic distributions, the
device mesh sink, the reference partition, and the CPU baseline (the oracle, parallel over buckets on the host cores).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector (counts packed FMA: 2 flops x 2 per lane per clock)
from mlsgpu_amd.synth import SLAB  # noqa: E402 - corner slices per GPU of the N > 1 workload
# digest of the meshes of the default N = 1 workload (cfg3 uniform): the value tests/test_gpu_configs.py pins next to
# oracle bit-parity on three of the 27 buckets (a data fixture, tests/golden/cfg3_uniform.json)
try:
    CFG3_UNIFORM_DIGEST = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg3_uniform.json")))["digest"]
except Exception:   # noqa: BLE001 - the fixture is optional for the benchmark
    CFG3_UNIFORM_DIGEST = None
try:    # EVERY slab of the N > 1 workload, as an inner and as a last slab: {slab: {"inner" | "last": totals + digest}}
    # (tests/test_gpu_configs.py::test_cfg4_slab_full_density runs all of them on one GPU next to oracle parity)
    CFG4SLAB_PINS = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg4slab_uniform.json")))["slabs"]
except Exception:   # noqa: BLE001
    CFG4SLAB_PINS = None


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200, help="timed passes (default: about 5 s of timed region at N = 1)")
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--workload", default="auto", choices=["auto", "cfg2", "cfg3", "cfg4slab", "cfg5"],
                   help="auto: cfg3 at N = 1, the cfg4 slab family at N > 1; cfg4slab runs one slab of cfg4 at N = 1; cfg5: "
                        "BASELINE configs[4], 2048^3 grid, 10^9 splats read from PLY files (written to --cfg5-dir first)")
    p.add_argument("--cfg5-dir", default="/dev/shm", help="where cfg5's PLY files are written (28 GB at scale 1)")
    p.add_argument("--cfg5-files", type=int, default=8)
    p.add_argument("--dist", default="uniform", choices=["uniform", "shells"])
    p.add_argument("--scale", type=float, default=1.0, help="splat-count scale (debug only; 1.0 = BASELINE size)")
    p.add_argument("--mesh-memory-mb", type=int, default=4096, help="Marching mesh arena per worker")
    p.add_argument("--workers", type=int, default=2,
                   help="device worker threads per GPU for the resident-input passes (round 3, cfg3 uniform: 1 / 2 / 3 / 4 / 6 workers "
                        "23.1 / 21.2-21.6 / 21.6-22.2 / 22.2-22.5 / 22.1 ms per step -- every kernel fills the GPU, a second worker "
                        "hides the host's gaps and more only interleave)")
    p.add_argument("--farm-workers", type=int, default=2,
                   help="device workers per GPU of the farm legs (host splats in, meshes out).  Round 5, shells cloud, 8d region with "
                        "six spare items: 2 workers 33.5-34 ms per job (the host-to-device copies busy 0.90-0.92 of it), 4 workers "
                        "36.5-38 (0.80), 8 workers 41 (0.74): the link is the floor, and fewer streams queue less in front of it")
    p.add_argument("--batch", type=int, default=4,
                   help="buckets a device worker takes through the path in lock-step (mlsgpu_hip_worker_process_batch: every "
                        "kernel has a bucket dimension, one set of launches and three host decisions per batch); 1 = bucket by "
                        "bucket (mlsgpu_hip_worker_process)")
    p.add_argument("--marching-group", type=int, default=2,
                   help="of a batch's buckets, how many share one set of processCorners / marching launches (the octree build "
                        "takes the whole batch); 0 = all (mlsgpu_hip_worker_set_marching_group)")
    p.add_argument("--variant", type=int, default=4, choices=[1, 4],
                   help="MLS kernel: 4 sub-block culling + cube streams (default), 1 the reference's structure")
    p.add_argument("--leg-steps", type=int, default=3, help="passes of every secondary leg")
    p.add_argument("--legs", default="all", choices=["all", "none"],
                   help="none: only the timed region, its roofline and (N > 1) the in-run per-GPU reference; all: the secondary legs "
                        "too (never `value`)")
    p.add_argument("--leg-budget-s", type=float, default=None,
                   help="wall-clock budget of ALL secondary legs together (default: 150 s at N > 1, none at N = 1): a leg that would "
                        "start after it is spent is skipped and named in leg_errors, so the line always arrives")
    p.add_argument("--restore-splats", action="store_true",
                   help="rounds 1-2's protocol: the tree build mutates the resident splats (radius -> 1/radius^2 in place, as the "
                        "reference's does) and every bucket starts with a device-to-device restore of its splats inside the timed "
                        "region.  Default since round 3: the workers keep the splats intact (mlsgpu_hip_worker_set_keep_splats), so "
                        "the resident input needs no restoring; the line reports this mode's step beside the headline")
    p.add_argument("--farm-spare", type=int, default=6,
                   help="device items per GPU beyond one per worker in the transfer legs (62 MB each here).  With the reference's one "
                        "spare item the copy side waits 4-6 ms per job for a worker to hand an item back and the link idles meanwhile")
    p.add_argument("--farm-batch", type=int, default=1,
                   help="buckets a farm worker of the transfer legs takes through one set of launches when that many are queued "
                        "(mlsgpu_hip_farm_set_batch)")
    p.add_argument("--staging-buffers", type=int, default=0, help="pinned staging buffers per copy side (0: the side's GPUs + 2)")
    p.add_argument("--copy-threads", type=int, default=16,
                   help="host threads (a persistent pool per copy side, bound to the GPU's NUMA node) copying one bucket into pinned "
                        "staging.  The copies must outrun the link: 1.68 GB per job in 11-16 ms with a source on the GPU's node, 21-31 "
                        "ms with a source on the other socket (8 threads: 30)")
    p.add_argument("--sink-rotation", type=int, default=3,
                   help="device sinks a stream of jobs rotates through in the device-sink transfer leg (a job's weld and read-back "
                        "overlap the following jobs)")
    p.add_argument("--no-bind", action="store_true",
                   help="leave the process where the scheduler puts it (default: the process -- every thread, every first touch of "
                        "host memory -- is bound to the CPUs of its GPU's NUMA node before anything is allocated)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-transfer", action="store_true", help="skip the transfer-inclusive legs (SURVEY 8d timed region)")
    p.add_argument("--no-shells", action="store_true", help="skip the D1 (shells) secondary measurement")
    p.add_argument("--no-partition", action="store_true", help="skip the device-bucketer leg (reference partition)")
    p.add_argument("--partition-max-splats", type=int, default=2097152,
                   help="bucket capacity of the device-bucketer leg (reference default 64 MiB / 32 B)")
    p.add_argument("--partition-workers", type=int, default=2, help="device workers of the device-bucketer leg when --batch > 1")
    p.add_argument("--weld-threads", type=int, default=0,
                   help="threads of the host welder (0: the library's default, min(32, hardware threads))")
    p.add_argument("--no-sink", action="store_true", help="skip the device mesh-sink leg (weld / components / prune)")
    p.add_argument("--no-timing", action="store_true", help="do not time individual kernels with HIP events")
    p.add_argument("--no-cross-check", action="store_true",
                   help="skip the bucket-by-bucket pass whose digest is held against the batched passes' (profiling runs: every "
                        "launch of the run is then a batched one)")
    p.add_argument("--headline-only", action="store_true", help="only the timed region and the roofline")
    a = p.parse_args()
    if a.legs == "none":
        a.headline_only = True
    return a


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks the way the driver does.  Nothing in this process has
    touched a GPU yet (device_count does not initialise HIP on this image), and the children are fresh processes."""
    import socket

    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("MLSGPU_BENCH_BACKEND", "nccl") == "nccl":
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible; refusing to report an %d-GPU number from fewer "
                         "devices" % (args.gpus, have, args.gpus))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


# ---------------------------------------------------------------------------------------------------- workloads

def build_workload(args, rank, world, device):
    """The rank's buckets, generated in HBM.  Returns a dict: bucketed (torch (n, 8) f32), buckets, voxels (this rank),
    n_splats (cloud splats this rank accounts for), text, cloud (the raw cloud tensor or None), grid."""
    import torch

    from mlsgpu_amd import synth
    name = args.workload
    if name == "auto":
        name = "cfg3" if world == 1 else "cfg4slab"
    if world > 1 and name != "cfg4slab":
        raise SystemExit("N > 1 runs the sharded cfg4 slab family only")
    if name in ("cfg2", "cfg3"):
        cloud, g = synth.make_cloud_device(name, device, scale=args.scale, dist=args.dist)
        boxes = synth.grid_buckets((g, g, g), 255)
        bucketed, buckets = synth.bucketize_device(cloud, boxes)
        text = ("%s: %d^3 grid, %d splats (%s), %d buckets of <= %d cells per side, octree+MLS+MC end-to-end"
                % (name, g, len(cloud), args.dist, len(buckets), max(max(b.num_vertices) for b in buckets) - 1))
        return dict(bucketed=bucketed, buckets=buckets, n_splats=len(cloud), text=text, cloud=cloud, grid=(g, g, g),
                    name=name, all_buckets=len(buckets))
    # cfg4 slab family: the cfg4 cloud, the first `world` slabs of SLAB corner slices, slab r = rank r's 25 buckets
    if args.dist != "uniform":
        raise SystemExit("the cfg4 slab family is defined on the uniform cloud")
    cloud, g = synth.make_cloud_device("cfg4", device, scale=args.scale)
    dims = (g, g, SLAB * world)
    mine_boxes = synth.slab_boxes(g, world, rank, SLAB)
    per = len(mine_boxes)
    bucketed, buckets = synth.bucketize_device(cloud, mine_boxes)
    # splats this rank accounts for in Msplats/s: centres inside its slab (the halo copies are not counted twice)
    z0 = buckets[0].low[2]
    z1 = z0 + buckets[0].num_vertices[2] - 1
    zc = cloud[:, 2]
    inside = (zc >= float(z0)) & ((zc <= float(z1)) if rank == world - 1 else (zc < float(z1)))
    mine = int(inside.sum().item())
    del inside
    text = ("cfg4 slab family: cfg4's cloud (1024^3 grid, %d uniform splats), grid %d x %d x %d = %d slab(s) of %d corner "
            "slices, %d buckets of <= %d x %d x %d cells per slab, one slab per GPU%s"
            % (len(cloud), dims[0], dims[1], dims[2], world, SLAB, per, buckets[0].num_vertices[0] - 1,
               buckets[0].num_vertices[1] - 1, max(b.num_vertices[2] for b in buckets) - 1,
               " (= BASELINE configs[3] in full)" if world * SLAB == g else ""))
    del zc
    return dict(bucketed=bucketed, buckets=buckets, n_splats=mine, text=text, cloud=None, grid=dims, name="cfg4slab",
                all_buckets=per * world)


# ---------------------------------------------------------------------------------------------------- cfg5

CFG5_PARTITION = dict(max_splats=2097152, max_cells=255, chunk_cells=0, micro_cells=63, max_split=1 << 30)   # reference defaults


def cfg5_paths(directory, nfiles, n, dist):
    return [os.path.join(directory, "mlsgpu_cfg5_%s_%d_%dof%d.ply" % (dist, n, k, nfiles)) for k in range(nfiles)]


def run_cfg5(args, rank, world, local_rank, device, dist, reduce_device):
    """BASELINE configs[4]: 2048^3 grid, 10^9 splats in PLY files -> FileSet reader threads -> HBM -> Bucket::bucket on the
    device -> the farm's device workers (leaves by device-side gathers).  The cloud (32 GB) and its partition live in HBM;
    the meshes are counted and checksummed on the device (the noise cloud's mesh is tens of G triangles).
    N > 1: every rank loads the files (page cache) and takes the leaves l with l % N == rank -- the bucket fan-out."""
    import shutil

    import torch

    import mlsgpu_amd as m
    from mlsgpu_amd import binding as mb, farm, synth
    g = synth.CONFIGS["cfg5"]["grid"]
    n = max(int(synth.CONFIGS["cfg5"]["splats"] * args.scale), 1)
    paths = cfg5_paths(args.cfg5_dir, args.cfg5_files, n, args.dist)
    need = n * 28 + 4096 * len(paths)
    wrote_s = 0.0
    # files left by an earlier run are reused only when every one has EXACTLY the size the generator gives it (header + 28
    # bytes per splat): a file truncated by a killed run is written again, not read
    sizes = synth.cloud_ply_sizes(len(paths), "cfg5", args.scale)
    if rank == 0 and not all(os.path.exists(p) and os.path.getsize(p) == sz for p, sz in zip(paths, sizes)):
        free = shutil.disk_usage(args.cfg5_dir).free
        if free < need * 1.05:
            raise SystemExit("cfg5: %s has %.1f GB free, the files need %.1f GB (use --cfg5-dir or --scale)"
                             % (args.cfg5_dir, free / 1e9, need / 1e9))
        t0 = time.time()
        synth.write_cloud_ply(paths, "cfg5", device, scale=args.scale, dist=args.dist)
        wrote_s = time.time() - t0
    if dist is not None:
        dist.barrier()
    ctx = m.Context(local_rank)
    nworkers = max(1, args.farm_workers)
    fs = mb.FileSet(paths, buffer_size=512 << 20)
    assert len(fs) == n
    raw = m.DeviceBuffer(ctx, nbytes=n * 32)
    ext = (0, g - 1, 0, g - 1, 0, g - 1)
    ref0 = (0.0, 0.0, 0.0)

    def load():
        fs.load(ctx, raw, reader_threads=32)
        ctx.synchronize()
    t0 = time.perf_counter()
    load()                                            # also the warm-up of the page cache
    first_load_s = time.perf_counter() - t0
    # sizes of the partition (untimed): worker capacity, voxels
    leaves = mb.bucket_cloud(ctx, raw, n, ref0, 1.0, ext, on_bucket=lambda leaf, ids: None, **CFG5_PARTITION)
    mine = [i for i in range(len(leaves)) if i % world == rank]
    voxels = sum(farm.leaf_cells(leaves[i]) for i in mine)
    pmax = max(l["num_splats"] for l in leaves)
    pcells = max(max(l["extents"][2 * a + 1] - l["extents"][2 * a] for a in range(3)) for l in leaves)
    bfarm = m.BucketFarm([local_rank], pmax, workers_per_device=nworkers, spare=1, max_cells=pcells,
                         mesh_memory=args.mesh_memory_mb << 20, collect="checksum")

    def resident_pass():
        farm.partition_to_farm(ctx, bfarm, local_rank, raw, n, ref0, 1.0, ext, CFG5_PARTITION,
                               keep=(lambda i: i % world == rank) if world > 1 else None)
        bfarm.finish()
    resident_pass()                                   # warm-up + the checked pass
    if bfarm.error is not None:
        raise bfarm.error
    digest = bfarm.digest()
    st0 = bfarm.stats()
    bfarm.checksums = False                           # the timed passes only count (the farm's own counters)
    for _ in range(max(0, args.warmup - 1)):
        resident_pass()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        resident_pass()
    ctx.synchronize()
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    elapsed, total_voxels, _ = farm.combine(own_elapsed, voxels * args.steps, dist, reduce_device)
    st1 = bfarm.stats()
    per_pass = {k: (st1[k] - st0[k]) // max(args.steps + max(0, args.warmup - 1), 1) for k in ("shipouts", "vertices", "triangles", "external", "buckets")}
    if per_pass["vertices"] != st0["vertices"] or per_pass["triangles"] != st0["triangles"]:
        raise SystemExit("cfg5: the timed passes produced %d vertices / %d triangles per pass, the checked pass %d / %d"
                         % (per_pass["vertices"], per_pass["triangles"], st0["vertices"], st0["triangles"]))
    # from the files: load + partition + pipeline, L passes (never `value`)
    L = max(1, args.leg_steps)
    bucket_s = 0.0
    t0 = time.perf_counter()
    for _ in range(L):
        mb.bucket_cloud(ctx, raw, n, ref0, 1.0, ext, on_bucket=lambda leaf, ids: None, **CFG5_PARTITION)
    ctx.synchronize()
    bucket_s = (time.perf_counter() - t0) / L
    load_s = 0.0
    t0 = time.perf_counter()
    for _ in range(L):
        t1 = time.perf_counter()
        load()
        load_s += time.perf_counter() - t1
        resident_pass()
    files_s = (time.perf_counter() - t0) / L
    load_s /= L
    t0 = time.perf_counter()
    bg = mb.bounding_grid(ctx, raw, n, 1.0, 63)
    ctx.synchronize()
    bound_s = time.perf_counter() - t0
    # out of core: the same set as if it did NOT fit the device -- the files streamed through a chunk buffer, once to count
    # and once per batch of top-level regions that fit the budget (mlsgpu_hip_bucket_stream); a third of the cloud at a time
    streamed = None
    if world == 1 and not args.headline_only:
        try:
            budget = max(int(n * 0.3), 1)
            chunk = max(min(64_000_000, n // 4), 1)
            bfarm.checksums = True
            bfarm.sums.clear()

            def stream_leaf(leaf, d_splats, d_ids, count=[0]):
                low, nv = farm.leaf_geometry(leaf, ext)
                bfarm.submit_device(local_rank, d_splats, d_ids, leaf["num_splats"], ref0, 1.0, ext, low, nv, count[0])
                count[0] += 1
            t0 = time.perf_counter()
            sg = mb.bounding_grid_files(ctx, fs, 1.0, 63, chunk, reader_threads=32)
            sbound_s = time.perf_counter() - t0
            t0 = time.perf_counter()
            sleaves, sstats = mb.bucket_cloud_stream(ctx, fs, ref0, 1.0, ext, budget_splats=budget, chunk_splats=chunk,
                                                     reader_threads=32, on_bucket=stream_leaf, **CFG5_PARTITION)
            bfarm.finish()
            stream_s = time.perf_counter() - t0
            streamed = {
                "ms_per_pass": round(stream_s * 1e3, 1), "msplats_per_s": round(n / stream_s / 1e6, 1),
                "mvoxels_per_s": round(voxels / stream_s / 1e6, 1), "budget_splats": budget, "chunk_splats": chunk,
                "file_passes": sstats["file_passes"], "batches": sstats["batches"], "chunks_skipped": sstats["chunks_skipped"],
                "splats_loaded_into_batches": sstats["batch_splats"], "buckets": len(sleaves),
                "same_buckets_as_resident": [l["extents"] for l in sleaves] == [l["extents"] for l in leaves]
                and [l["num_splats"] for l in sleaves] == [l["num_splats"] for l in leaves],
                "same_meshes_as_resident": bfarm.digest() == digest,
                "bounding_grid_from_files_ms": round(sbound_s * 1e3, 1), "bounding_grid_matches": list(sg[2]) == list(bg[2]),
                "note": "the set treated as larger than the device: never more than budget_splats of it resident; files -> chunk "
                        "buffer -> microblock-octree counters (pass 1) -> per batch of top-level regions: files -> filter into the "
                        "batch buffer in file order -> member lists, recursion, device gathers into the farm (mlsgpu_hip_bucket_stream)"}
        except Exception as e:      # noqa: BLE001
            streamed = {"error": "%s: %s" % (type(e).__name__, e)}
    ms_per_step = elapsed / args.steps * 1e3
    golden = None
    try:
        golden = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg5_%s.json" % args.dist))).get(str(n))
    except Exception:   # noqa: BLE001 - the fixture is optional for the benchmark
        golden = None
    result = {
        "metric": "Mvoxels/s evaluated+triangulated", "value": round(total_voxels / elapsed / 1e6, 3), "unit": "Mvoxels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "msplats_per_s": round(n * args.steps / elapsed / 1e6, 3), "timed_region_s": round(elapsed, 3),
        "config": {
            "workload": "cfg5: %d^3 grid, %d splats (%s) in %d PLY files, loaded into HBM (32 B per splat), partitioned on the "
                        "device with the reference's defaults (255 cells, 2 097 152 splats per bucket, 63-cell microblocks): %d "
                        "buckets of <= %d cells per side, <= %d splats; octree+MLS+MC on every bucket, meshes counted and "
                        "checksummed in HBM" % (g, n, args.dist, len(paths), len(leaves), pcells, pmax),
            "voxels_per_step": int(total_voxels // args.steps), "buckets": len(leaves), "buckets_this_rank": len(mine),
            "bucket_splats_total": int(sum(l["num_splats"] for l in leaves)), "device_workers": nworkers,
            "mesh_memory_mb": args.mesh_memory_mb, "timed_region": "cloud resident in HBM -> Bucket::bucket -> device gathers -> "
            "device workers (the partition is recomputed in every step)",
            "triangles_per_step": st0["triangles"], "vertices_per_step": st0["vertices"], "shipouts_per_step": st0["shipouts"],
            "sharding": "leaf l of the partition goes to rank l mod N; every rank holds the cloud" if world > 1 else "single GPU",
            "files_written_s": round(wrote_s, 1), "file_bytes": int(sum(os.path.getsize(p) for p in paths)),
        },
        "output_digest": {"rank0_digest": digest, "what": "sha256/16 over (bucket number, sizes and vertex / triangle / external-key "
                          "checksums of its ship-outs) in bucket order, computed on the device"},
        "from_files": {
            "ms_per_pass": round(files_s * 1e3, 1), "msplats_per_s": round(n / files_s / 1e6, 1),
            "mvoxels_per_s": round(voxels / files_s / 1e6, 1),
            "loader_ms": round(load_s * 1e3, 1), "loader_GBps_of_splats": round(n * 32 / load_s / 1e9, 2),
            "loader_GBps_of_file": round(n * 28 / load_s / 1e9, 2), "first_load_ms": round(first_load_s * 1e3, 1),
            "bucketing_ms": round(bucket_s * 1e3, 2), "bucketing_msplats_per_s": round(n / bucket_s / 1e6, 1),
            "bounding_grid_ms": round(bound_s * 1e3, 2), "bounding_grid_extents": [int(x) for x in bg[2]],
            "passes": L,
            "note": "PLY files in %s (page cache) -> 32 reader threads decoding into a 512 MiB pinned buffer -> H2D -> the timed "
                    "region's pipeline; never `value`" % args.cfg5_dir},
    }
    if streamed is not None:
        result["out_of_core"] = streamed
    if golden is not None and world == 1:
        # the triangle total does not depend on how the mesh memory cuts a bucket into ship-outs; the digest is pinned for
        # this bench's own mesh memory
        result["output_digest"]["triangles_expected"] = golden["triangles"]
        if st0["triangles"] != golden["triangles"]:
            raise SystemExit("cfg5: %d triangles per pass, tests/golden/cfg5_%s.json has %d" % (st0["triangles"], args.dist, golden["triangles"]))
        gb = golden.get("bench")
        if gb is not None and gb.get("mesh_memory_mb") == args.mesh_memory_mb:
            result["output_digest"]["expected"] = gb["digest"]
            result["output_digest"]["ok"] = digest == gb["digest"]
            if digest != gb["digest"]:
                raise SystemExit("cfg5 digest %s differs from the pinned %s" % (digest, gb["digest"]))
    if rank == 0:
        print(json.dumps(result))
    bfarm.close()
    fs.close()


# ---------------------------------------------------------------------------------------------------- legs

def drain_utilisation(c):
    """How the accumulation loops of processCorners use a wave's 64 lanes (mlsgpu_hip_mls_set_stats): a drain call runs as
    many iterations as the longest of its lanes' hit lists, so utilisation = hits / (64 x iterations)."""
    hits, calls, it = c[2], c[3], c[4]
    if calls == 0 or it == 0:
        return None
    hist = c[8:41]
    lanes = sum(hist)
    return {"drain_calls": calls, "iterations": it, "mean_hits_per_lane_per_call": round(hits / max(lanes, 1), 3),
            "mean_longest_list": round(it / calls, 3), "utilisation": round(hits / (64.0 * it), 4),
            "if_two_calls_were_one": round(hits / (64.0 * c[5]), 4) if c[5] else None,
            "if_a_round_were_one_call": round(hits / (64.0 * c[6]), 4) if c[6] else None,
            "if_a_block_were_one_call": round(hits / (64.0 * c[7]), 4) if c[7] else None,
            "lanes_by_hits_per_call": {str(n): hist[n] for n in range(33) if hist[n]},
            "what": "the accumulation order per corner is fixed (bit-identical sums), so a lane's hits cannot move to another "
                    "lane: utilisation is bounded by how unevenly a call's hits fall on the 64 corners; merging calls "
                    "evens them out at the price of LDS for the lists that must stay alive"}


def cpu_sample_boxes(grid, side=63):
    """Where the CPU baseline samples the cloud: cubes of `side` cells on a regular lattice through the whole grid, in an
    order that visits distant places first, so that any prefix is spread over the cloud."""
    per = (grid - 1) // side
    boxes = [((x * side, y * side, z * side), (side + 1, side + 1, side + 1))
             for z in range(per) for y in range(per) for x in range(per)]
    rng = np.random.default_rng(12345)
    return [boxes[i] for i in rng.permutation(len(boxes))]


def cpu_baseline(sample_host, sample_buckets, max_cells):
    """The CPU baseline: the oracle (oracle/, "port") on the host cores, PARALLEL OVER BUCKETS like the GPU farm -- one
    single-threaded worker process per hardware thread, each with its own box of the same cloud, all started together;
    throughput = cells of all those boxes / wall time until the last one finishes.  The oracle is rebuilt here with -O3
    -march=native for this machine's CPU.  The boxes are `side`-cell cubes rather than whole 170-cell buckets so that the
    leg takes seconds, not minutes (a whole cfg3 bucket is about 145 s of one core); the cloud is uniform, so the rate is
    the rate of whole buckets."""
    # every host core, whatever the process was bound to for the GPU legs (the children inherit this thread's mask)
    try:
        os.sched_setaffinity(0, range(os.cpu_count() or 1))
    except OSError:
        pass
    cores = len(os.sched_getaffinity(0))
    try:
        import psutil
        mem_gb = psutil.virtual_memory().available / 2 ** 30
    except Exception:
        mem_gb = 64.0
    lib = os.path.join(ROOT, "oracle", "liboracle_native.so")
    built = subprocess.call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "native"],
                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 0 and os.path.exists(lib)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("MLSGPU_ORACLE_LIB", None)
    if built:
        env["MLSGPU_ORACLE_LIB"] = lib
    nproc = int(max(1, min(cores, mem_gb / 1.0, len(sample_buckets))))
    tmp = tempfile.mkdtemp(prefix="mlsgpu_cpu_")
    go = os.path.join(tmp, "go")
    procs = []
    for w in range(nproc):
        b = sample_buckets[w]
        job = os.path.join(tmp, "job%d.npz" % w)
        np.savez(job, splats=sample_host[b.first:b.first + b.count],
                 buckets=np.array([[0, b.count] + list(b.low) + list(b.num_vertices)], np.int64), max_cells=max_cells)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "cpu_bucket_worker.py"), job, go],
                                      stdout=subprocess.PIPE, env=env))
    deadline = time.time() + 120
    while time.time() < deadline and sum(1 for f in os.listdir(tmp) if ".ready." in f) < nproc:
        time.sleep(0.05)
    t0 = time.time()
    open(go, "w").close()
    outs = []
    for p in procs:
        line = p.communicate()[0].decode().strip().splitlines()
        if p.returncode == 0 and line:
            outs.append(json.loads(line[-1]))
    wall = max(o["t_end"] for o in outs) - t0 if outs else float("nan")
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)
    if not outs:
        return None
    cells = sum(o["cells"] for o in outs)
    busy = sum(o["seconds"] for o in outs) / (wall * min(cores, len(outs)))
    flops = sum(10 * 512 * o["listed"] + 25 * o["hits"] for o in outs)
    mls_s = max(sum(o["mls_s"] for o in outs), 1e-9)
    return {
        "value": round(cells / wall / 1e6, 4), "unit": "Mvoxels/s", "cores": len(outs), "kind": "port",
        "sample": "%d cubes of %d^3 cells spread over the same cloud (%d cells, %d splats with halo), one single-threaded "
                  "oracle process per cube, all started together on a %d-thread host; %.1f s wall until the last finished, "
                  "%.1f s of work per cube on average"
                  % (len(outs), sample_buckets[0].num_vertices[0] - 1, cells, sum(o["splats"] for o in outs), cores, wall,
                     sum(o["seconds"] for o in outs) / len(outs)),
        "host_threads": cores, "cores_busy_frac": round(busy, 3),
        "build": "-O3 -march=native on this host" if built else "portable -O2 -mavx2 build (no compiler run here)",
        "stage_cpu_seconds": {"octree": round(sum(o["tree_s"] for o in outs), 2),
                              "processCorners": round(mls_s, 2),
                              "marching": round(sum(o["marching_s"] for o in outs), 2)},
        "processCorners_GFLOPs_per_core": round(flops / mls_s / 1e9, 3),
        "processCorners_GFLOPs_all_cores": round(flops / mls_s / 1e9 * len(outs), 1),
        "note": "the reference has no CPU path of its own (SURVEY 8d); this is the scalar restatement used as the parity "
                "oracle, every core busy on its own bucket.  The north_star's >= 10x target is met with a wide margin "
                "under any plausible CPU number; the kernel quality figure is roofline.frac, not this ratio.",
    }


_SINK_PINS = []


def transfer_legs(m, args, device_index, bucketed_host, buckets, max_count, max_cells, voxels, steps, with_sink=True):
    """SURVEY 8(d)'s timed region: host splats of every bucket in -> last byte of mesh back in host memory.
    route "shipouts": every ship-out read back asynchronously through the farm's pinned circular buffer, overlapped with
    the next buckets (the reference's route, src/workers.h:488-509, src/mesh.cpp:62-102);
    route "device_sink": ship-outs appended to the device mesher, weld / components / prune in HBM, ONE read-back."""
    out = {}
    views = [bucketed_host[b.first:b.first + b.count] for b in buckets]
    nworkers = max(1, min(args.farm_workers, len(buckets)))
    farm = m.BucketFarm([device_index], max_count, workers_per_device=nworkers, spare=args.farm_spare, max_cells=max_cells,
                        mesh_memory=args.mesh_memory_mb << 20, copy_threads=args.copy_threads, staging_buffers=args.staging_buffers)
    farm.set_host_output(6 << 30, None)
    if args.farm_batch > 1:
        farm.set_batch(min(args.farm_batch, m.binding.MAX_BATCH))

    drain = [0.0]

    def stream_pass():
        for i, (b, v) in enumerate(zip(buckets, views)):
            farm.submit(v, b.low, b.num_vertices, i)
        t = time.perf_counter()
        farm.finish()
        drain[0] += time.perf_counter() - t
    # the first passes of a process through this route run 20-30 % slower than the ones that follow, whatever farm they go
    # through (a fresh farm in a warm process is fast at once: the probe of tools/transfer_probe.py reads 42-47 ms per job in
    # its first call and 33.5 in the second): about half a second of untimed passes first
    t0 = time.perf_counter()
    stream_pass()
    first_s = time.perf_counter() - t0
    for _ in range(max(2, min(12, int(0.5 / max(first_s, 1e-3))))):
        stream_pass()
    before = farm.host_stats()
    c0 = farm.copy_clock()
    w0 = farm.worker_clock()
    drain[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        stream_pass()
    dt = (time.perf_counter() - t0) / steps
    hs = farm.host_stats()
    c1 = farm.copy_clock()
    w1 = farm.worker_clock()
    d2h = (hs["bytes"] - before["bytes"]) / steps
    per = {k: (c1[k] - c0[k]) / steps for k in ("fill_s", "wait_staging_s", "wait_item_s", "h2d_s", "enqueue_s")}
    # ... and the same passes as ONE stream of buckets (no drain between jobs): what the link sustains when the next job's
    # splats follow the last bucket of this one, as they do when jobs queue up
    c2 = farm.copy_clock()
    t0 = time.perf_counter()
    for _ in range(steps):
        for i, (b, v) in enumerate(zip(buckets, views)):
            farm.submit(v, b.low, b.num_vertices, i)
    farm.finish()
    dt_stream = (time.perf_counter() - t0) / steps
    c3 = farm.copy_clock()
    out["shipouts"] = {
        "value": round(voxels / dt / 1e6, 3), "unit": "Mvoxels/s", "ms_per_step": round(dt * 1e3, 3),
        "h2d_GB_per_step": round(bucketed_host.nbytes / 1e9, 3), "d2h_GB_per_step": round(d2h / 1e9, 3),
        "link_GBps": round((bucketed_host.nbytes + d2h) / dt / 1e9, 2), "ring_waits": hs["ring_waits"] - before["ring_waits"],
        # the copy side's clock: the link is the floor of this region, so the figure to watch is how busy it is
        "h2d_busy_frac": round(per["h2d_s"] / dt, 3), "h2d_GBps_while_copying": round(bucketed_host.nbytes / max(per["h2d_s"], 1e-9) / 1e9, 1),
        "copy_side_ms_per_step": {"fill_staging": round(per["fill_s"] * 1e3, 2), "wait_for_staging": round(per["wait_staging_s"] * 1e3, 2),
                                  "wait_for_device_item": round(per["wait_item_s"] * 1e3, 2), "enqueue_calls": round(per["enqueue_s"] * 1e3, 2),
                                  "h2d_copies": round(per["h2d_s"] * 1e3, 2)},
        "streamed": {"ms_per_step": round(dt_stream * 1e3, 3), "value": round(voxels / dt_stream / 1e6, 3),
                     "h2d_busy_frac": round((c3["h2d_s"] - c2["h2d_s"]) / steps / dt_stream, 3),
                     "what": "%d jobs submitted back to back, one wait at the end: no pipeline drain between jobs" % steps},
        "link_floor_ms": round(per["h2d_s"] * 1e3, 2),
        "drain_ms_per_job": round(drain[0] / steps * 1e3, 2),
        "workers_ms_per_step": {"busy": round((w1["busy_s"] - w0["busy_s"]) / steps * 1e3, 2),
                                "idle": round((w1["idle_s"] - w0["idle_s"]) / steps * 1e3, 2), "threads": nworkers},
        "placement": farm.placement(),
        "note": "per job (ms_per_step): pageable host splats -> pinned staging (%d copy threads) -> H2D -> %d device workers (+ %d "
                "spare items) -> every ship-out read back through a 6 GiB pinned circular buffer, consumed (dropped) by the farm's "
                "mesher thread -> wait for the last byte.  link_floor_ms = the job's host-to-device copies alone, at the rate the "
                "link gave them next to the read-backs: the floor of this region"
                % (args.copy_threads, nworkers, args.farm_spare)}
    farm.close()
    if not with_sink:
        return out
    # route 2: the device sink, one final D2H of the welded, pruned mesh.  Two sinks (and two farms) alternate: while job
    # k's weld, prune and read-back run, job k + 1's splats are already on their way in -- the steady state of a stream of
    # jobs, which is what the ship-out route's ring gives the reference (its read-backs overlap the next buckets too).
    import threading
    # every sink welds and reads back on a stream of its own, of HIGH priority: the weld of job k competes with the
    # kernels of job k + 1 for the GPU, and it is the weld that is on the critical path of the steady state
    import torch
    # sinks in rotation: NS - 1 welds / read-backs may be in flight while the next job streams in (two sinks: 47-54 ms per job
    # on the shells cloud, three 38-40)
    NS = max(2, args.sink_rotation)
    hi = [torch.cuda.Stream(device=device_index, priority=int(os.environ.get("MLSGPU_BENCH_SINK_PRIORITY", "-1"))) for _ in range(NS)]
    fctx = [m.Context(device_index, stream=s_.cuda_stream) for s_ in hi]
    sinks = [m.Mesher(c, 0.02) for c in fctx]
    for s_ in sinks:
        s_.set_background(True)     # their welds run while the next job's buckets are on the GPU
    # spare device items beyond one per worker: the previous job's weld shares the GPU with this job's kernels, and with
    # one spare item every delayed bucket stalls the host-to-device copies behind it (shells cloud, steady state: 47 ms per
    # job with 1 spare item, 42 with 4, 44 with 12)
    sink_spare = int(os.environ.get("MLSGPU_BENCH_FARM_SPARE", "4"))
    farms = [m.BucketFarm([device_index], max_count, workers_per_device=nworkers, spare=sink_spare, max_cells=max_cells,
                          mesh_memory=args.mesh_memory_mb << 20, sink=s_, copy_threads=args.copy_threads) for s_ in sinks]
    # the pinned landing buffers are kept from leg to leg: buffers allocated afresh after a leg that had pinned (and freed)
    # tens of GB land on slower memory (read-back 12.9 -> 17.6 ms for the shells mesh, and the job 37.6 -> 47.8 ms)
    while len(_SINK_PINS) < NS:
        _SINK_PINS.append(m.binding.PinnedBuffer(1))
    pins = _SINK_PINS[:NS]
    got_bytes = [0] * NS
    errors = []

    phase = {"submit": 0.0, "weld": 0.0, "read_back": 0.0, "jobs": 0}      # seconds, summed over the timed jobs

    def submit_job(k):
        t = time.perf_counter()
        for b, v in zip(buckets, views):
            farms[k].submit(v, b.low, b.num_vertices, 0)
        farms[k].finish()
        phase["submit"] += time.perf_counter() - t

    def finish_job(k):
        try:
            t = time.perf_counter()
            skip = os.environ.get("MLSGPU_BENCH_SINK_SKIP", "")      # diagnosis: "weld", "readback"
            n = 0 if skip == "weld" else sinks[k].finalize()
            t1 = time.perf_counter()
            nb = 0
            for i in range(0 if skip == "readback" else n):
                nb += m.binding.download_into_pinned(fctx[k], sinks[k].chunk(i, download=False), pins[k])
            fctx[k].synchronize()
            sinks[k].reset()
            got_bytes[k] = nb
            phase["weld"] += t1 - t
            phase["read_back"] += time.perf_counter() - t1
            phase["jobs"] += 1
        except Exception as e:      # noqa: BLE001 - raised by the main thread
            errors.append(e)

    def run_jobs(count):
        pending = []                           # finish threads in flight, oldest first: at most NS - 1
        for j in range(count):
            k = j % NS
            submit_job(k)
            while len(pending) >= NS - 1:      # sink (j + 1) % NS must be free before the next job is submitted into it
                pending.pop(0).join()
            t = threading.Thread(target=finish_job, args=(k,))
            t.start()
            pending.append(t)
        for t in pending:
            t.join()
        if errors:
            raise errors[0]
    run_jobs(NS)                               # warm-up; sizes the arenas and the pinned landing buffers
    t0 = time.perf_counter()
    submit_job(0)
    finish_job(0)
    single_s = time.perf_counter() - t0        # one job alone, nothing overlapped: its latency
    jobs = max(3 * steps, 9)                  # the pipeline's fill and drain (one job's weld) amortised over the jobs
    phase.update(submit=0.0, weld=0.0, read_back=0.0, jobs=0)
    t0 = time.perf_counter()
    run_jobs(jobs)
    dt = (time.perf_counter() - t0) / jobs
    out["device_sink"] = {
        "phases_ms_per_job": {k_: round(phase[k_] / max(phase["jobs"], 1) * 1e3, 2) for k_ in ("submit", "weld", "read_back")},
        "value": round(voxels / dt / 1e6, 3), "unit": "Mvoxels/s", "ms_per_step": round(dt * 1e3, 3),
        "one_job_alone_ms": round(single_s * 1e3, 3),
        "h2d_GB_per_step": round(bucketed_host.nbytes / 1e9, 3), "d2h_GB_per_step": round(got_bytes[0] / 1e9, 3),
        "note": "host splats -> farm -> ship-outs appended in HBM -> weld + components + prune (0.02) on the device -> ONE "
                "read-back of the final mesh into pinned memory; consecutive jobs rotate through %d sinks, so a job's "
                "weld and read-back overlap the following jobs' transfer and compute (ms_per_step is the steady state over %d "
                "jobs; one_job_alone_ms is a single job's latency)" % (NS, jobs)}
    for k in range(NS):
        farms[k].close()
        sinks[k].close()
        fctx[k].close()
    return out


def host_weld_leg(m, args, device_index, bucketed_host, buckets, max_count, max_cells, voxels, steps=3):
    """The reference's complete route, welder included: host splats -> farm -> every ship-out read back through the pinned ring
    -> the mesher thread hands it to the host welder (OOCMesher's weld: local components, key map, union-find;
    src/mesher.cpp:220-311 -- a task per block on the welder's pool of threads, where the reference has one thread and an
    OpenMP rewrite, src/mesher.cpp:597-600) -> finalize (components, prune, one mesh per chunk).  One job = one fresh welder;
    a warm-up job first (the welder's memory comes from a cache of mapped slabs).  Two figures: a job ALONE (its latency: pass,
    then finalize), and a STREAM of jobs in which job k's finalize runs on its own thread while job k + 1's buckets are already
    going through the farm into the next welder -- the steady state `value` is quoted on."""
    import threading
    nworkers = max(1, min(args.farm_workers, len(buckets)))
    farm = m.BucketFarm([device_index], max_count, workers_per_device=nworkers, spare=args.farm_spare, max_cells=max_cells,
                        mesh_memory=args.mesh_memory_mb << 20, copy_threads=args.copy_threads, staging_buffers=args.staging_buffers)
    views = [bucketed_host[b.first:b.first + b.count] for b in buckets]
    last = {}

    def stream_in(welder):
        farm.set_host_output(2 << 30, welder)
        for b, v in zip(buckets, views):
            farm.submit(v, b.low, b.num_vertices, 0)
        farm.finish()

    def finish(welder, times=None):
        t = time.perf_counter()
        n = welder.finalize()
        if times is not None:
            times.append(time.perf_counter() - t)
        last.update(n=n, st=welder.stats(), threads=welder.threads())
        welder.close()

    def job_alone():
        welder = m.HostMesher(0.02, threads=args.weld_threads)
        t0 = time.perf_counter()
        stream_in(welder)
        t1 = time.perf_counter()
        finish(welder)
        return t1 - t0, time.perf_counter() - t1
    job_alone()                                             # warm-up: arenas, pinned ring, the welder's slabs
    alone = [job_alone() for _ in range(max(1, steps))]
    # the stream: at most one finalize in flight behind the job that is streaming in.  TWO welders are alive at a time, so
    # the stream has its own warm-up (the second set of slabs is mapped and faulted in once)
    in_times, join_times = [], []

    def stream(count, times):
        pending = None
        for _ in range(count):
            welder = m.HostMesher(0.02, threads=args.weld_threads)
            t = time.perf_counter()
            stream_in(welder)
            in_times.append(time.perf_counter() - t)
            t = time.perf_counter()
            if pending is not None:
                pending.join()
            join_times.append(time.perf_counter() - t)
            pending = threading.Thread(target=finish, args=(welder, times))
            pending.start()
        pending.join()
    stream(3, None)
    jobs = max(2 * steps, 6)
    fin_times = []
    t0 = time.perf_counter()
    stream(jobs, fin_times)
    per_job = (time.perf_counter() - t0) / jobs
    hs = farm.host_stats()
    farm.close()
    st = last["st"]
    total = sum(a + b for a, b in alone) / len(alone)
    return {"value": round(voxels / total / 1e6, 3), "unit": "Mvoxels/s", "ms_per_step": round(total * 1e3, 1), "steps": len(alone),
            "pass_until_last_mesh_welded_ms": round(sum(a for a, _ in alone) / len(alone) * 1e3, 1),
            "finalize_ms": round(sum(b for _, b in alone) / len(alone) * 1e3, 1),
            "streamed": {"ms_per_step": round(per_job * 1e3, 1), "value": round(voxels / per_job / 1e6, 3), "jobs": jobs,
                         "finalize_ms": round(sum(fin_times) / len(fin_times) * 1e3, 1),
                         "stream_in_ms": round(sum(in_times[-jobs:]) / jobs * 1e3, 1),
                         "wait_for_previous_finalize_ms": round(sum(join_times[-jobs:]) / jobs * 1e3, 1),
                         "what": "job k's finalize on its own thread while job k + 1 streams into the next welder"},
            "vertices_welded_per_s": round(st["vertices_added"] / total), "weld_threads": last["threads"], "chunks": last["n"],
            "ring_waits": hs["ring_waits"], "welded_vertices": st["total_vertices"], "kept_triangles": st["kept_triangles"],
            "note": "per job, jobs one after the other: host splats in -> farm -> ring read-backs -> host welder (a task per block on "
                    "weld_threads threads) -> finalize; streamed = the same jobs with a job's finalize overlapping the next job's "
                    "transfer; the reference welds on one mesher thread (doc/mlsgpu-user-manual.xml:508-511)"}


def multi_gpu_legs(m, args, result, dist, park, reduce_device, rank, world, local_rank, ndev, ctx, bucketed_t, buckets, max_count,
                   max_cells, voxels, L, nworkers, deadline=None):
    """N > 1 only, never `value`.
    transfer_inclusive: SURVEY 8(d)'s region on every rank at once, with the weld in it -- host splats -> the rank's farm
        (pinned staging, H2D) -> device workers -> ship-outs appended to the rank's device sink -> dist_sink.global_prune
        (boundary export, ONE all-gather, merged verdict, output pass) -> the rank's welded, pruned mesh read back.
    single_process: the reference's own shape (src/mlsgpu_core.cpp:704-741): ONE process, one copy side, a device group
        per GPU, fed from one place -- rank 0 drives all N GPUs while the other ranks wait; N x rank 0's slab, from host
        memory (one copy thread + staging, what the manual names as the limiter) and from a cloud resident on GPU 0 (peer
        copies over xGMI)."""
    import torch

    from mlsgpu_amd import dist_sink, synth

    class LegFailed(Exception):
        pass

    def in_budget(what):
        """collective: False on every rank once ANY rank's clock is past the deadline of the secondary legs; the leg is then
        skipped everywhere (nobody waits in a collective the others never reach) and named in leg_errors"""
        if deadline is None:
            return True
        t = torch.tensor([1 if time.perf_counter() > deadline else 0], dtype=torch.int64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if int(t.item()) != 0:
            result.setdefault("leg_errors", {})[what] = ("skipped: the %.0f s budget of the secondary legs was spent (--leg-budget-s)"
                                                         % args.leg_budget_s)
            return False
        return True

    def all_ok(ok, what):
        """collective: True on every rank iff every rank is fine -- a rank that failed locally must not leave the others in
        a collective it never reaches"""
        t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) != 1:
            raise LegFailed(what)

    def wall(fn, steps):
        """max over ranks of the time of `steps` calls, bracketed by barriers"""
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        out = None
        for _ in range(steps):
            out = fn()
        torch.cuda.synchronize()
        own = time.perf_counter() - t0
        dist.barrier()
        t = torch.tensor([own], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / steps, out

    # ---- transfer-inclusive, every rank, with the cross-rank weld ----
    local_error = None
    host = views = sink = bfarm = pinned = None
    run_transfer = in_budget("transfer_inclusive")
    try:
        host = synth.to_host_splats(bucketed_t)
        views = [host[b.first:b.first + b.count] for b in buckets]
        sink = m.Mesher(ctx, 0.02)
        bfarm = m.BucketFarm([local_rank], max_count, workers_per_device=nworkers, spare=args.farm_spare, max_cells=max_cells,
                             mesh_memory=args.mesh_memory_mb << 20, sink=sink)
        pinned = m.binding.PinnedBuffer(1)
    except Exception as e:      # noqa: BLE001
        local_error = "%s: %s" % (type(e).__name__, e)
    parts = {}

    def sink_pass():
        nonlocal local_error
        t0 = time.perf_counter()
        mine = None
        try:
            for b, v in zip(buckets, views):
                bfarm.submit(v, b.low, b.num_vertices, rank)
            bfarm.finish()
            t1 = time.perf_counter()
            mine = sink.boundary()
        except Exception as e:      # noqa: BLE001
            local_error = "%s: %s" % (type(e).__name__, e)
        all_ok(local_error is None, "pass / boundary export")
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        keep, stats = dist_sink.merge_boundaries(gathered, 0.02)       # the same computation on every rank
        nbytes = 0
        try:
            nchunks = sink.finalize_with(keep[rank])
            t2 = time.perf_counter()
            for i in range(nchunks):
                nbytes += m.binding.download_into_pinned(ctx, sink.chunk(i, download=False), pinned)
            ctx.synchronize()
            t3 = time.perf_counter()
            sink.reset()
            parts.update(pass_ms=(t1 - t0) * 1e3, weld_ms=(t2 - t1) * 1e3, readback_ms=(t3 - t2) * 1e3, nbytes=nbytes, stats=stats)
        except Exception as e:      # noqa: BLE001
            local_error = "%s: %s" % (type(e).__name__, e)
        all_ok(local_error is None, "verdict pass / read-back")
        return nbytes
    try:
        all_ok(local_error is None, "set-up")
        if not run_transfer:
            raise LegFailed("budget")
        sink_pass()                                       # warm-up: arenas, pinned landing buffer
        if not in_budget("transfer_inclusive (timed passes)"):
            raise LegFailed("budget")
        dt, nbytes = wall(sink_pass, L)
        tot = torch.tensor([float(nbytes), float(host.nbytes)], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(tot)
        st = parts["stats"]
        result["transfer_inclusive"] = {
            "device_sink_global_weld": {
                "value": round(voxels * world / dt / 1e6, 3), "unit": "Mvoxels/s", "ms_per_step": round(dt * 1e3, 3),
                "h2d_GB_per_step": round(float(tot[1].item()) / 1e9, 3), "d2h_GB_per_step": round(float(tot[0].item()) / 1e9, 3),
                "rank0_ms": {k: round(parts[k], 2) for k in ("pass_ms", "weld_ms", "readback_ms")},
                "whole_job": {k: int(st[k]) for k in ("total_vertices", "components", "kept_components", "kept_vertices", "kept_triangles")},
                "note": "every rank at once: host splats -> pinned staging -> H2D -> %d device workers -> ship-outs appended in HBM -> "
                        "per-rank weld + boundary export -> ONE all-gather -> merged components and prune threshold (0.02 of the "
                        "whole job) -> output pass -> the rank's mesh read back into pinned memory; time = slowest rank" % nworkers},
            "distribution": "uniform"}
    except LegFailed as e:
        if str(e) != "budget":
            result["transfer_inclusive"] = {"error": "a rank failed in %s%s" % (e, ": " + local_error if local_error else "")}
    for obj in (bfarm, sink):
        try:
            if obj is not None:
                obj.close()
        except Exception:       # noqa: BLE001
            pass
    if pinned is not None:
        pinned.free()

    # ---- the reference's shape: one process, N device groups; rank 0 drives, the others wait ----
    devices = [d % ndev for d in range(world)]
    single = None
    torch.cuda.synchronize()
    if not in_budget("single_process"):
        return
    dist.barrier(group=park)
    try:
        if rank == 0:
            single = single_process_leg(m, args, result, ctx, local_rank, devices, bucketed_t, buckets, views, max_count, max_cells,
                                        voxels, L, nworkers, world, deadline)
    except Exception as e:      # noqa: BLE001 - rank 0 still has to reach the barrier the others wait at
        single = {"error": "%s: %s" % (type(e).__name__, e)}
    dist.barrier(group=park)
    if single is not None:
        result["single_process"] = single


def single_process_leg(m, args, result, ctx, local_rank, devices, bucketed_t, buckets, views, max_count, max_cells, voxels, L,
                       nworkers, world, deadline=None):
    """The reference's own shape (src/mlsgpu_core.cpp:704-741) on rank 0: one farm over every GPU, N x rank 0's slab."""
    sfarm = m.BucketFarm(devices, max_count, workers_per_device=nworkers, spare=args.farm_spare, max_cells=max_cells,
                         mesh_memory=args.mesh_memory_mb << 20)

    def host_fed():
        for rep in range(world):
            for i, (b, v) in enumerate(zip(buckets, views)):
                sfarm.submit(v, b.low, b.num_vertices, rep)
        sfarm.finish()
    def passes():
        # as many timed passes as the budget of the secondary legs still allows (rank 0 works alone here: its clock rules)
        return L if deadline is None else (L if time.perf_counter() + 2.0 < deadline else 1)
    t0 = time.perf_counter()
    host_fed()
    warm_s = time.perf_counter() - t0
    if deadline is not None and time.perf_counter() + warm_s > deadline:
        sfarm.close()
        return {"error": "skipped after the warm-up pass (%.1f s): the budget of the secondary legs was spent" % warm_s}
    s0 = sfarm.stats()
    Lh = passes()
    t0 = time.perf_counter()
    for _ in range(Lh):
        host_fed()
    host_s = (time.perf_counter() - t0) / Lh
    s1 = sfarm.stats()
    # the same buckets resident on GPU 0, handed out by device gathers (another GPU's group: scratch ring + peer copy)
    raw = m.DeviceBuffer(ctx, nbytes=bucketed_t.numel() * 4, borrow=bucketed_t.data_ptr())
    iota = m.DeviceBuffer(ctx, array=np.arange(max_count, dtype=np.uint32))
    gx, gy, gz = result["_grid"]
    ext = (0, gx - 1, 0, gy - 1, 0, gz - 1)

    class _Sub:
        def __init__(self, ptr):
            self.ptr = ptr

    def device_fed():
        for rep in range(world):
            for b in buckets:
                sfarm.submit_device(local_rank, _Sub(raw.ptr + 32 * b.first), iota.ptr, b.count, (0.0, 0.0, 0.0), 1.0, ext,
                                    b.low, b.num_vertices, rep)
        sfarm.finish()
    if deadline is not None and time.perf_counter() > deadline:
        sfarm.close()
        return {"devices": devices, "buckets_per_pass": world * len(buckets),
                "host_fed": {"value": round(voxels * world / host_s / 1e6, 3), "unit": "Mvoxels/s", "ms_per_pass": round(host_s * 1e3, 2),
                             "h2d_GBps": round((s1["h2d_bytes"] - s0["h2d_bytes"]) / Lh / host_s / 1e9, 2)},
                "device_fed": {"error": "skipped: the budget of the secondary legs was spent"}}
    device_fed()
    s1b = sfarm.stats()
    Ld = passes()
    t0 = time.perf_counter()
    for _ in range(Ld):
        device_fed()
    dev_s = (time.perf_counter() - t0) / Ld
    s2 = sfarm.stats()
    single = {
        "devices": devices, "buckets_per_pass": world * len(buckets),
        "host_fed": {"value": round(voxels * world / host_s / 1e6, 3), "unit": "Mvoxels/s", "ms_per_pass": round(host_s * 1e3, 2),
                     "h2d_GBps": round((s1["h2d_bytes"] - s0["h2d_bytes"]) / Lh / host_s / 1e9, 2)},
        "device_fed": {"value": round(voxels * world / dev_s / 1e6, 3), "unit": "Mvoxels/s", "ms_per_pass": round(dev_s * 1e3, 2)},
        "buckets_per_device_device_fed": [int(x) for x in (np.array(s2["per_device"][:world]) - np.array(s1b["per_device"][:world]))],
        "device_fed_passes": Ld,
        "in_flight_max": s2["in_flight_max"],
        "note": "ONE process (rank 0) with one device group per GPU, %d workers each, the other ranks idle: N x rank 0's slab "
                "from pageable host memory through ONE copy side (4 copy threads -> pinned staging -> H2D to the chosen "
                "group), and from a cloud resident on GPU 0 (device gather, peer copy to other GPUs' items); meshes counted "
                "only" % nworkers}
    sfarm.close()
    del raw, iota
    return single


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d: refusing to report a number for a different device count"
                         % (args.gpus, world))

    # one process per GPU: next to its GPU, before any host memory is touched and before the HIP runtime starts its threads
    # (rank r of an 8-GPU node lands on the socket GPU r hangs off; the pinned staging, the read-back ring, the copy threads,
    # the runtime's event threads and the welder follow).  torch.cuda.device_count() does not initialise HIP on this image.
    import torch
    from mlsgpu_amd import farm as _farm
    if args.no_bind:
        process_placement = {"bound": False, "why": "--no-bind"}
    else:
        process_placement = _farm.bind_process_to_device_node(local_rank % max(torch.cuda.device_count(), 1), before_hip=True)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one process per GPU; MLSGPU_BENCH_BACKEND=gloo lets several ranks share one GPU (a single-GPU check of the
    # N > 1 code path: the only collectives are a barrier and reductions of scalars, so RCCL is not essential)
    backend = os.environ.get("MLSGPU_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= ndev:
        raise SystemExit("LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, ndev))
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    # MLSGPU_BENCH_FORCE_DIST=1 (with `torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 --workload cfg4slab`): the
    # N > 1 code path -- RCCL process group, collectives, the N > 1 legs -- with ONE rank, for a box with one GPU.  A check
    # of that code, never a measurement; the line says so.
    force_dist = world == 1 and os.environ.get("MLSGPU_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    # Ranks that only WAIT while one rank works alone (the per-GPU reference, the one-process farm) wait on the CPU (gloo):
    # a rank parked in an RCCL barrier keeps a spinning kernel on its GPU, and the one-process leg uses those GPUs.
    park = None
    if dist is not None:
        park = dist.new_group(backend="gloo") if backend == "nccl" else dist.group.WORLD
    reduce_device = "cuda" if (dist is not None and backend == "nccl") else None

    import mlsgpu_amd as m
    from mlsgpu_amd import farm, synth

    if args.workload == "cfg5":
        run_cfg5(args, rank, world, local_rank, device, dist, reduce_device)
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- workload (generated in HBM, untimed) ----
    t0 = time.time()
    W = build_workload(args, rank, world, device)
    torch.cuda.synchronize()
    bucketed_t, buckets = W["bucketed"], W["buckets"]
    voxels = sum(b.cells for b in buckets)
    max_count = max(b.count for b in buckets)
    max_cells = max(max(b.num_vertices) for b in buckets) - 1
    setup_s = time.time() - t0

    # the CPU baseline's sample: small cubes all over the same cloud, cut while the raw cloud is in HBM
    cpu_sample = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.headline_only and W["cloud"] is not None:
        boxes = cpu_sample_boxes(W["grid"][0])[:os.cpu_count() or 1]
        st, sb = synth.bucketize_device(W["cloud"], boxes)
        cpu_sample = (st.cpu().numpy().view(m.SPLAT_DTYPE).reshape(-1), sb)
        del st

    nworkers = max(1, min(args.workers, len(buckets)))
    ctxs = [m.Context(local_rank) for _ in range(nworkers)]
    ctx = ctxs[0]
    nbytes = bucketed_t.numel() * 4
    pristine = m.DeviceBuffer(ctx, nbytes=nbytes, borrow=bucketed_t.data_ptr())
    work = m.DeviceBuffer(ctx, nbytes=nbytes)
    workers = [m.Worker(c, max_count, max_cells=max_cells, mesh_memory=args.mesh_memory_mb << 20) for c in ctxs]
    # The reference's octree build overwrites splat.w with 1/r^2 (kernels/octree.cl:193).  Its splats arrive by H2D copy for
    # every work item, so nothing is lost; RESIDENT splats would have to be restored before every bucket (rounds 1-2 did
    # that with a device-to-device copy inside the timed region).  Since round 3 the workers leave the splats intact
    # (non-mutating build, processCorners takes the reciprocal while staging: bit-identical field) and the resident
    # input is simply processed again.  --restore-splats brings the old protocol back; its step is reported either way.
    mutating = [bool(args.restore_splats)]
    batch = max(1, min(args.batch, m.binding.MAX_BATCH, -(-len(buckets) // nworkers)))
    for w in workers:
        w.set_mls_variant(args.variant)
        w.set_keep_splats(not mutating[0])
        w.set_batch(batch)
        w.set_marching_group(max(0, min(args.marching_group, m.binding.MAX_BATCH)))
    pool = ThreadPoolExecutor(nworkers)
    collectors = [m.binding.SizeCollector() for _ in range(nworkers)]

    def fresh(c, b):
        if mutating[0]:
            m.binding.check(m.lib().mlsgpu_hip_memcpy_d2d(c.h, work.ptr + 32 * b.first, pristine.ptr + 32 * b.first, 32 * b.count))

    def run_buckets(w, c, some, col):
        # a worker's buckets: `batch` at a time through ONE set of launches (the SubItems of a work item,
        # src/workers.cpp:232-286), or bucket by bucket
        if batch > 1:
            for b in some:
                fresh(c, b)
            w.process_batch(work, some, collector=col)
        else:
            for b in some:
                fresh(c, b)
                w.process(work, b.first, b.count, b.low, b.num_vertices, collector=col)

    shares = [list(farm.worker_share(buckets, k, nworkers)) for k in range(nworkers)]

    def run_share(k):
        # worker k takes buckets k, k + nworkers, ... (ctypes releases the GIL inside the library).  With a mutating build
        # every bucket starts from a fresh copy of its resident splats, ON THE WORKER'S STREAM, where the reference has the
        # host-to-device copy of the work item (src/workers.cpp:356-361), inside the timed region.
        run_buckets(workers[k], ctxs[k], shares[k], collectors[k])
        ctxs[k].synchronize()

    def step():
        list(pool.map(run_share, range(nworkers)))

    def barrier():
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # The instrumented passes (per-kernel HIP-event timing on one worker, work counters, output digest) run BEFORE the
    # warm-up and the timed region: they are needed anyway, and they leave the clocks, the caches and every arena in the
    # state of a running job, so that a short timed region (the driver's K = 20) measures the same steady state as a
    # long one.
    # ---- per-kernel durations: passes on ONE worker with HIP events around every launch.  Kept out of the headline
    # region because with several workers the streams overlap and an event pair then measures a kernel sharing the GPU,
    # not the kernel; single-worker durations are what the roofline divides by (and what `rocprofv3 --kernel-trace
    # --stats ... --workers 1` reports). ----
    kernel_stats = {}
    ksteps = max(1, min(args.steps, 10))
    if not args.no_timing:
        ctx.reset_stats()
        ctx.set_timing(True)
        for _ in range(ksteps):
            work.copy_from(pristine)
            run_buckets(workers[0], ctx, buckets, m.binding.SizeCollector())
        ctx.set_timing(False)
        kernel_stats = dict(ctx.stats())

    # ---- one worker alone, nothing instrumented: what the host decisions and launch gaps of a bucket cost when no other
    # worker fills them (the instrumented device.compute above carries two event records per launch) ----
    single_worker_ms = None
    if not args.no_timing:
        sw_steps = max(3, min(args.steps, 10))
        col1 = m.binding.SizeCollector()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(sw_steps):
            run_buckets(workers[0], ctx, buckets, col1)
        ctx.synchronize()
        single_worker_ms = (time.perf_counter() - t0) / sw_steps * 1e3

    # ---- algorithmic work + output digest (one instrumented, untimed pass on worker 0) ----
    counters = m.DeviceBuffer(ctx, array=np.zeros(m.binding.MLS_STATS_WORDS, np.uint64))
    w0 = workers[0]
    before = w0.marching_counters()
    w0.set_mls_stats(counters)
    work.copy_from(pristine)
    corners = entries = 0
    check = m.binding.ChecksumCollector(ctx)
    for g0 in range(0, len(buckets), batch):
        group = buckets[g0:g0 + batch]
        run_buckets(w0, ctx, group, check)                  # the path the timed region runs
        for lane, b in enumerate(group):
            entries += w0.tree_num_entries(lane)
            corners += int(np.prod([-(-n // 8) * 8 for n in b.num_vertices]))
    ctx.synchronize()
    mls_counters = [int(x) for x in counters.download(np.uint64)]
    listed, tests, hits = mls_counters[:3]
    w0.set_mls_stats(None)
    after = w0.marching_counters()
    mc = {k: after[k] - before[k] for k in after}
    digest = check.digest()
    if batch > 1 and not args.no_cross_check:
        # ... and bucket by bucket (mlsgpu_hip_worker_process): the same meshes, ship-out by ship-out
        check_1 = m.binding.ChecksumCollector(ctx)
        for b in buckets:
            fresh(ctx, b)
            w0.process(work, b.first, b.count, b.low, b.num_vertices, collector=check_1)
        ctx.synchronize()
        if check_1.digest() != digest:
            raise SystemExit("batched passes (batch %d) produce digest %s, bucket-by-bucket passes %s"
                             % (batch, digest, check_1.digest()))

    for _ in range(args.warmup):
        step()
    barrier()
    # N > 1: ONE rank's slab alone on its GPU while the other ranks are parked at the barrier -- the per-GPU rate the
    # N-rank value is held against (same cloud, same density, same process, measured in this run)
    ref_steps = max(3, min(args.steps, 10))
    ref_elapsed = None
    if dist is not None:
        if rank == 0:
            t0 = time.perf_counter()
            for _ in range(ref_steps):
                step()
            for c in ctxs:
                c.synchronize()
            ref_elapsed = time.perf_counter() - t0
        dist.barrier(group=park)
        barrier()
    collectors[:] = [m.binding.SizeCollector() for _ in range(nworkers)]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    for c in ctxs:
        c.synchronize()
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0
    # the other splat protocol, a few passes, never `value`: what the restore copies (or their absence) are worth
    other_mode_ms = None
    if dist is None and not args.headline_only:
        timed_collectors = list(collectors)      # the counts of the timed passes stay what they are
        collectors[:] = [m.binding.SizeCollector() for _ in range(nworkers)]
        mutating[0] = not mutating[0]
        for w in workers:
            w.set_keep_splats(not mutating[0])
        work.copy_from(pristine)
        ctx.synchronize()
        step()                                   # warm-up of the mode
        t1 = time.perf_counter()
        for _ in range(ref_steps):
            step()
        for c in ctxs:
            c.synchronize()
        other_mode_ms = (time.perf_counter() - t1) / ref_steps * 1e3
        mutating[0] = not mutating[0]
        for w in workers:
            w.set_keep_splats(not mutating[0])
        work.copy_from(pristine)
        ctx.synchronize()
        collectors[:] = timed_collectors
    if dist is not None:
        dist.barrier()
    # whole job: MAX of the elapsed time over ranks, SUM of the voxels (each rank ran `steps` passes over its buckets)
    elapsed, total_voxels, _ = farm.combine(own_elapsed, voxels * args.steps, dist, reduce_device)
    per_rank = None
    total_splats = W["n_splats"]
    if dist is not None:
        t = torch.zeros((world, 4), dtype=torch.float64, device=reduce_device)
        t[rank, 0], t[rank, 1], t[rank, 2], t[rank, 3] = own_elapsed, len(buckets), voxels, W["n_splats"]
        dist.all_reduce(t)
        per_rank = t.cpu().numpy()
        total_splats = int(per_rank[:, 3].sum())

    triangles = sum(c.triangles for c in collectors) // max(args.steps, 1)
    vertices = sum(c.vertices for c in collectors) // max(args.steps, 1)
    external = sum(c.external for c in collectors) // max(args.steps, 1)
    shipouts = sum(c.batches for c in collectors) // max(args.steps, 1)
    if check.vertices != vertices or check.triangles != triangles:
        raise SystemExit("the timed passes produced %d vertices / %d triangles per step, the checked pass %d / %d"
                         % (vertices, triangles, check.vertices, check.triangles))

    ms_per_step = elapsed / args.steps * 1e3
    value = farm.throughput(total_voxels, elapsed)
    result = {
        "metric": "Mvoxels/s evaluated+triangulated",
        "value": round(value, 3),
        "unit": "Mvoxels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "msplats_per_s": round(total_splats * args.steps / elapsed / 1e6, 3),
        # what the timed region does with the resident input (rounds 1-2 timed a per-bucket restore copy in place of the
        # reference's per-item H2D transfer; the other protocol's step is under other_splat_protocol)
        "input_protocol": "restored_by_d2d_copy_per_bucket" if args.restore_splats else "resident_in_place",
        "timed_region_s": round(elapsed, 3),
        "config": {
            "workload": W["text"],
            "voxels_per_step": int(total_voxels // args.steps),
            "bucket_splats_total": int(bucketed_t.shape[0]),
            "mesh_memory_mb": args.mesh_memory_mb,
            "device_workers": nworkers,
            "batch": batch, "marching_group": args.marching_group,
            "resident_splats": ("restored by a device-to-device copy before every bucket, inside the timed region (the tree build "
                                "mutates them, as the reference's does)" if args.restore_splats else
                                "processed in place: the workers keep them intact (non-mutating tree build, processCorners takes "
                                "1/r^2 while staging; bit-identical output), so nothing is restored"),
            "mls_variant": {1: "basic", 4: "culled+cube-streams"}[args.variant],
            "sharding": "one process per GPU, rank r owns z-slab r (25 buckets); no data-path collective" if world > 1
                        else "single GPU",
            "triangles_per_step": triangles,
            "vertices_per_step": vertices,
            "shipouts_per_step": shipouts,
            "counts_cover": "the whole cloud" if world == 1 else "rank 0's slab (bucket_splats_total, triangles, vertices, ship-outs)",
            "setup_s": round(setup_s, 1),
        },
        "output_digest": {
            "rank0_digest": digest, "batches": check.batches,
            "what": "sha256/16 over (sizes, vertex / triangle / external-key checksums) of every ship-out of one pass of rank "
                    "0's buckets, computed on the device (mlsgpu_hip_mesh_checksum)"},
    }
    if other_mode_ms is not None:
        result["other_splat_protocol"] = {
            "ms_per_step": round(other_mode_ms, 3), "steps": ref_steps,
            "what": ("keep-splats workers, no restore copies" if args.restore_splats else
                     "rounds 1-2's protocol (--restore-splats): mutating tree build + a device-to-device restore of every bucket's "
                     "splats inside the step")}
    result["_grid"] = W["grid"]
    result["placement"] = process_placement
    if dist is not None:
        pl = torch.zeros((world, 3), dtype=torch.int64, device=reduce_device)
        pl[rank, 0], pl[rank, 1], pl[rank, 2] = process_placement.get("gpu_node", -1), int(process_placement.get("bound", False)), \
            process_placement.get("cpus", 0)
        dist.all_reduce(pl)
        pl = pl.cpu().numpy()
        result["placement"]["per_rank"] = {"gpu_node": [int(x) for x in pl[:, 0]], "bound": [bool(x) for x in pl[:, 1]],
                                           "cpus": [int(x) for x in pl[:, 2]]}
    if W["name"] == "cfg3" and args.dist == "uniform" and args.scale == 1.0 and CFG3_UNIFORM_DIGEST is not None:
        result["output_digest"]["expected"] = CFG3_UNIFORM_DIGEST
        result["output_digest"]["ok"] = digest == CFG3_UNIFORM_DIGEST
        if digest != CFG3_UNIFORM_DIGEST:
            raise SystemExit("output digest %s differs from the pinned %s: the timed pipeline did not produce the meshes "
                             "the parity tests check" % (digest, CFG3_UNIFORM_DIGEST))
    # EVERY rank's slab is pinned (slab r as an inner slab of 128 cell slices, or as the job's last one of 127): each rank holds
    # its own digest against its pin, the verdicts are gathered, and one mismatch anywhere fails the whole run
    digest_failure = None
    if W["name"] == "cfg4slab" and args.scale == 1.0 and CFG4SLAB_PINS is not None:
        pin = CFG4SLAB_PINS.get(str(rank), {}).get(synth.slab_variant(world, rank))
        mine_ok = pin is not None and digest == pin["digest"] and check.vertices == pin["vertices"] \
            and check.triangles == pin["triangles"]
        verdicts = [bool(mine_ok)]
        digests = [digest]
        if dist is not None:
            d = torch.zeros((world, 3), dtype=torch.int64, device=reduce_device)
            d[rank, 0], d[rank, 1], d[rank, 2] = int(mine_ok), int(digest[:8], 16), int(digest[8:], 16)
            dist.all_reduce(d)
            d = d.cpu().numpy()
            verdicts = [bool(x) for x in d[:, 0]]
            digests = ["%08x%08x" % (int(a), int(b)) for a, b in d[:, 1:3]]
        expected = [CFG4SLAB_PINS.get(str(r), {}).get(synth.slab_variant(world, r), {}).get("digest") for r in range(world)]
        result["output_digest"].update(expected=expected[0], ok=verdicts[0], per_rank_digest=digests, per_rank_expected=expected,
                                       per_rank_ok=verdicts, all_ok=all(verdicts))
        if not all(verdicts):
            digest_failure = ("output digests %s of the ranks' slabs differ from the pinned %s (tests/golden/cfg4slab_uniform.json)"
                              % (digests, expected))
    if digest_failure is not None:
        if rank == 0:
            print("bench.py: " + digest_failure, file=sys.stderr, flush=True)
        if dist is not None:
            dist.destroy_process_group()
        raise SystemExit(3)
    if force_dist:
        result["debug_forced_dist"] = "ONE rank with the N > 1 code path (MLSGPU_BENCH_FORCE_DIST=1): a check of that code, not a measurement"
    if world > ndev:
        result["debug_shared_gpu"] = "%d ranks on %d GPU(s) (MLSGPU_BENCH_BACKEND=gloo): a check of the N > 1 code path, NOT an N-GPU measurement" % (world, ndev)
    if per_rank is not None:
        result["per_rank"] = {
            "buckets": [int(x) for x in per_rank[:, 1]],
            "ms_per_step": [round(x / args.steps * 1e3, 3) for x in per_rank[:, 0]],
            "mvoxels_per_s": [round(v * args.steps / e / 1e6, 1) for e, v in zip(per_rank[:, 0], per_rank[:, 2])],
            "digest_ok": result["output_digest"].get("per_rank_ok"),
            "note": "every rank is one GPU working on its own slab; value = sum of voxels / slowest rank's time",
            "reading_the_scaling_curve": "the N = 1 point of the driver's curve is cfg3 (0.37 splats per voxel); the N > 1 "
                                         "points are slabs of cfg4 (0.19 splats per voxel, the density BASELINE names), a lighter "
                                         "cloud per voxel: hold the N > 1 values against per_gpu_reference (rank 0's slab alone on "
                                         "its GPU, measured in this run), not against the N = 1 point"}
        ref = torch.zeros(1, dtype=torch.float64, device=reduce_device)
        if rank == 0:
            ref[0] = voxels * ref_steps / ref_elapsed / 1e6
        dist.all_reduce(ref)
        result["per_gpu_reference"] = {
            "value": round(float(ref.item()), 3), "unit": "Mvoxels/s", "steps": ref_steps,
            "what": "rank 0's slab (25 buckets) alone on GPU 0, the other ranks parked at a barrier; same process, same cloud"}
        result["scaling_efficiency"] = round(value / (world * float(ref.item())), 4)

    # ---- roofline: algorithmic bytes over HIP-event kernel time, per stage ----
    if kernel_stats and not args.no_timing:
        K = ksteps
        T, O, Vw, C = 3 * triangles, mc["occupied"], vertices, voxels
        # 16 key bits in two 8-bit passes; the first is fused into writeEntries (octree.hip entryScatterKernel) unless
        # MLSGPU_HIP_OCTREE_FUSED=0, so the sort stage proper is one pass
        sort_passes = 2 if os.environ.get("MLSGPU_HIP_OCTREE_FUSED") == "0" else 1
        we_name = "writeEntries (count+scan+write%s)" % (", first sort pass fused" if sort_passes == 1 else "")
        # round 4: on the default route the sort's last pass writes the ids straight to their command positions (whole-key
        # counts ride on its histogram kernel, a scan over the NODES replaces the scan over the entries): per entry the
        # histogram reads its key (4), the scatter reads key + id (8) and writes the id (4)
        direct = sort_passes == 1
        sort_name = ("sortHist(+key counts)+sortScatter (last pass, ids to command positions)" if direct
                     else "sortHist+sortScatter (octree entries, %d pass)" % sort_passes)
        scan_name = "scan over the nodes: start / jump slots / command bases" if direct else "countCommands+scan+writeSplatIds"
        nb_splats = int(bucketed_t.shape[0])
        num_start = 37449 * len(buckets)                   # nodes of the default six-level tree, per bucket
        models = {
            # stat name: (kernel, algorithmic bytes per step)
            "kernel.mls.processCorners.time": ("processCorners", 36 * listed + 4 * corners),
            "kernel.octree.sort.time": (sort_name, 16 * entries if direct else sort_passes * 20 * entries),
            # 16 N read + 16 N written back (1 / r^2) + 8 E' of entries
            "kernel.octree.writeEntries.time": (we_name, 32 * nb_splats + 8 * entries),
            "kernel.octree.scan.time": (scan_name, 20 * num_start if direct else 2 * 4 * entries + 8 * entries + 4 * entries),
            "kernel.marching.generateElements.time": ("latticeTriangles", 4 * T + 16 * O + O),
            "kernel.marching.compactVertices.time": ("latticeVertices", 12 * Vw + 8 * external + 8 * Vw),
            "kernel.marching.countUniqueVertices.time": ("latticeMask", C + 8 * 12 * (corners // 64)),
            "kernel.marching.genOccupied.time": ("cellCode+classify", 4 * corners + C + C),
        }
        tj = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath)).get("%s/%s" % (W["name"], args.dist), {})
            except Exception:
                tj = {}
        traffic_of = {"processCorners": ["processCorners"], "latticeTriangles": ["latticeTriangles"],
                      "latticeVertices": ["latticeVertices"], "latticeMask": ["latticeMask"],
                      "cellCode+classify": ["cellCode"], we_name: ["writeEntries"],
                      scan_name: ["writeSplatIds"] if not direct else [],
                      sort_name: ["sortHist", "sortScatter"]}
        stages = []
        for stat, (kname, nb) in models.items():
            if stat in kernel_stats and kernel_stats[stat][1] > 0:
                ms = kernel_stats[stat][0] / K
                per_launch = [tj.get(k) for k in traffic_of.get(kname, [])]
                stages.append({"stat": stat, "kernel": kname, "ms_per_step": round(ms, 3),
                               "launches_per_step": kernel_stats[stat][1] // K,
                               "algorithmic_bytes_per_step": int(nb),
                               "achieved_GBps": round(nb / (ms * 1e-3) / 1e9, 1),
                               "hbm_frac": round(nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "hbm_traffic_bytes_per_launch": (sum(per_launch) if per_launch and all(v is not None for v in per_launch)
                                                                else None)})
        stages.sort(key=lambda s: -s["ms_per_step"])
        single_ms = sum(v[0] for k, v in kernel_stats.items() if k == "device.compute") / K
        pc = "kernel.mls.processCorners.time"
        measured = ("hipEvent pairs on the worker's stream, %d single-worker passes before the timed region (%.1f ms per pass)"
                    % (K, single_ms))
        if stages and stages[0]["stat"] == pc:
            # Dominant kernel = processCorners: fp32 VALU + LDS work (>= 140 flop/B); no MFMA is issued (no dense
            # contraction).  `achieved` counts the flops the kernel EXECUTES (10 per lane-test that survives sub-block culling,
            # 25 per hit) against the fp32 vector peak, which counts packed FMAs.  The reference algorithm's count (every
            # corner tests every listed splat) is given beside it as algorithmic_equiv_*.
            total_ms, launches = kernel_stats[pc]
            ms = total_ms / K
            alg_flops = 10 * 512 * listed + 25 * hits          # SURVEY 8d per-bucket figure, summed over buckets
            done_flops = 10 * tests + 25 * hits
            achieved = done_flops / (ms * 1e-3) / 1e12
            result["roofline"] = {
                "kernel": "processCorners",
                "bound": "valu_fp32",
                "achieved": round(achieved, 3),
                "peak": FP32_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / FP32_VALU_PEAK_TFLOPS, 4),
                "traffic": tj.get("processCorners"),
                "avg_launch_ms": round(total_ms / launches, 4),
                "launches_per_step": launches // K,
                "executed_flops_per_launch": int(done_flops // max(launches // K, 1)),
                "executed": "10 flops per (corner, splat) lane-test executed + 25 per hit; the tests of lanes outside the support "
                            "are executed work, the culled ones are not counted",
                "reading_frac": "frac counts EXECUTED flops, so it falls when finer culling removes tests: rounds 1-2 executed "
                                "20.9 G lane-tests per step for these 3.47 G hits at 390 us per launch (frac 0.18), the cube "
                                "streams of round 3 execute 9.3 G at ~300 us (frac 0.14); avg_launch_ms and "
                                "algorithmic_equiv_frac (the reference algorithm's flop count over the same time) are the "
                                "figures that compare across rounds",
                "algorithmic_equiv_TFLOPs": round(alg_flops / (ms * 1e-3) / 1e12, 3),
                "algorithmic_equiv_frac": round(alg_flops / (ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS, 4),
                "algorithmic_equiv": "SURVEY 8d: 10*512*SigmaL + 25*H flops of the reference's every-corner-tests-every-listed-"
                                     "splat loop, most of which sub-block culling never executes",
                "hbm_algorithmic_GBps": round((36 * listed + 4 * corners) / (ms * 1e-3) / 1e9, 1),
                "sigma_L": listed, "tests": tests, "hits": hits, "corners": corners,
                "drain_lane_utilisation": drain_utilisation(mls_counters),
                "share_of_kernel_time": round(stages[0]["ms_per_step"] / sum(s_["ms_per_step"] for s_ in stages), 3),
                "measured": measured,
            }
        elif stages:
            top = stages[0]
            total_ms, launches = kernel_stats[top["stat"]]
            result["roofline"] = {
                "kernel": top["kernel"],
                "bound": "hbm",
                "achieved": top["achieved_GBps"],
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(top["achieved_GBps"] / HBM_PEAK_GBS, 5),
                "traffic": top["hbm_traffic_bytes_per_launch"],
                "avg_launch_ms": round(total_ms / launches, 4),
                "launches_per_step": launches // K,
                "algorithmic_bytes_per_launch": int(top["algorithmic_bytes_per_step"] // max(launches // K, 1)),
                "share_of_kernel_time": round(top["ms_per_step"] / sum(s_["ms_per_step"] for s_ in stages), 3),
                "measured": measured,
            }
        if stages:
            result["roofline"]["hbm_stages"] = stages
            result["roofline"]["traffic_source"] = ("profiles/traffic.json: rocprofv3 --pmc passes of this commit's kernels, "
                                                    "corrected as MI355X_MICROARCH.md prescribes (tools/profile_summary.py)")
        tpp = {}
        if tj:
            try:
                tpp = json.load(open(tpath)).get("%s/%s/per_pass" % (W["name"], args.dist), {})
            except Exception:
                tpp = {}
        if tpp and "roofline" in result:
            # bytes of all of a kernel's launches over one pass of the workload (the profile's own batching: a launch covers
            # several buckets), so this holds for the full-size workload the profile was taken on
            moved = sum(tpp.values()) if args.scale == 1.0 else 0
            result["roofline"]["pipeline_hbm"] = {
                "traffic_bytes_per_step": int(moved), "achieved_GBps": round(moved / (ms_per_step * 1e-3) / 1e9, 1),
                "peak_GBps": HBM_PEAK_GBS, "frac": round(moved / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "PMC traffic of the tracked kernels / step time with %d device workers" % nworkers}
        result["kernel_ms_per_step"] = {k: round(v[0] / K, 3) for k, v in sorted(kernel_stats.items())}
        result["single_worker_ms_per_step"] = {
            "value": round(single_worker_ms, 3),
            "what": "the same buckets on ONE device worker, un-instrumented (three host decisions and ~40 launches per bucket "
                    "with nothing to overlap them); kernel_ms_per_step['device.compute'] is the instrumented pass"}
        result["work_per_step"] = {"octree_entries": entries, "occupied_cells": O, "unwelded_vertices": mc["unwelded"],
                                   "welded_vertices": Vw, "external_vertices": external, "indices": T}

    secondary = dist is None and not args.headline_only
    L = max(1, args.leg_steps)

    # N > 1: the headline, the in-run per-GPU reference and the scaling efficiency leave on stderr BEFORE any secondary leg
    # starts (the ONE line on stdout comes at the end): whatever happens to a leg, the curve is on record
    if dist is not None and rank == 0:
        early = {k: result[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "per_gpu_reference",
                                        "scaling_efficiency") if k in result}
        print("bench.py headline before the secondary legs: " + json.dumps(early), file=sys.stderr, flush=True)
    if args.leg_budget_s is None and dist is not None:
        args.leg_budget_s = 150.0
    leg_deadline = None if args.leg_budget_s is None else time.perf_counter() + args.leg_budget_s
    if dist is not None and not args.headline_only:
        multi_gpu_legs(m, args, result, dist, park, reduce_device, rank, world, local_rank, ndev, ctxs[0], bucketed_t, buckets,
                       max_count, max_cells, voxels, L, max(1, args.farm_workers), leg_deadline)

    # ---- mesh-sink leg (never `value`): every ship-out of one pass appended to the device mesher (d2d), then
    # finalize = weld by key across buckets + connected components + prune (--fit-prune default 0.02) + compaction ----
    if secondary and not args.no_sink:
        try:     # a secondary leg never costs the line its headline
            sink = m.Mesher(ctx, 0.02)
            sink.reserve(mc["welded"] + 1024, mc["indices"] // 3 + 1024, mc["external"] + 1024)   # counts of the stats pass
            work.copy_from(pristine)
            ctx.synchronize()

            def sink_share(k):
                col = sink.collector(ctxs[k], 0)
                for b in farm.worker_share(buckets, k, nworkers):
                    workers[k].process(work, b.first, b.count, b.low, b.num_vertices, collector=col)
                ctxs[k].synchronize()
            t0 = time.perf_counter()
            list(pool.map(sink_share, range(nworkers)))
            add_s = time.perf_counter() - t0
            t0 = time.perf_counter()
            nchunks = sink.finalize()
            ctx.synchronize()
            fin_s = time.perf_counter() - t0
            st = sink.stats()
            result["mesh_sink"] = {
                "pass_with_appends_ms": round(add_s * 1e3, 3), "finalize_ms": round(fin_s * 1e3, 3),
                "finalize_mvertices_per_s": round(st["vertices_added"] / fin_s / 1e6, 1), "chunks": nchunks,
                "vertices_added": st["vertices_added"], "triangles_added": st["triangles_added"],
                "welded_vertices": st["total_vertices"], "components": st["components"], "kept_components": st["kept_components"],
                "kept_vertices": st["kept_vertices"], "kept_triangles": st["kept_triangles"],
                "device_workers": nworkers,
                "note": "meshes never leave HBM; finalize = key sort + union-find + sizes + two compaction scans",
            }
            # what one rank of a one-process-per-GPU job pays instead of finalize: the boundary export, the merge of all ranks'
            # exports (here: its own) and the output pass with the merged verdict (mlsgpu_amd/dist_sink.py)
            from mlsgpu_amd import dist_sink
            b_calls = []
            for _ in range(3):                             # the first export of a process maps its host vectors on the way
                t0 = time.perf_counter()
                part = sink.boundary()
                b_calls.append(time.perf_counter() - t0)
            b_s = min(b_calls)
            dist_sink.merge_boundaries([part], 0.02)        # (numpy / scipy warm up)
            t0 = time.perf_counter()
            keep, dstats = dist_sink.merge_boundaries([part], 0.02)
            m_s = time.perf_counter() - t0
            t0 = time.perf_counter()
            sink.finalize_with(keep[0])
            ctx.synchronize()
            f_s = time.perf_counter() - t0
            result["mesh_sink"]["distributed"] = {
                "boundary_ms": round(b_s * 1e3, 3), "boundary_calls_ms": [round(x * 1e3, 3) for x in b_calls],
                "merge_ms": round(m_s * 1e3, 3), "finalize_with_ms": round(f_s * 1e3, 3),
                "what": "behind a finalize(): the export numbers the roots, counts triangles per component and compacts the "
                        "distinct keys (the weld and the components are reused); then the merge of the exports and the output "
                        "pass with the merged verdict",
                "keys": int(len(part[0])), "components": int(len(part[2])), "export_bytes": int(sum(a.nbytes for a in part)),
                "same_verdict": dstats["kept_triangles"] == st["kept_triangles"] and dstats["total_vertices"] == st["total_vertices"]}
            sink.close()
        except Exception as e:      # noqa: BLE001 - reported in the line
            result.setdefault("leg_errors", {})['mesh_sink'] = "%s: %s" % (type(e).__name__, e)

    # ---- device-bucketer leg (never `value`): the RAW cloud resident in HBM, partitioned on the device exactly as
    # the reference's Bucket::bucket would with its defaults (255-cell buckets, 63-cell microblocks, 2 097 152 splats,
    # src/mlsgpu_core.cpp:112-132,655-678), each leaf gathered + transformed on the device and run through a worker ----
    if secondary and not args.no_partition and W["cloud"] is not None:
        try:     # a secondary leg never costs the line its headline
            from mlsgpu_amd import binding as mb
            grid = W["grid"][0]
            n_splats = W["n_splats"]
            raw = m.DeviceBuffer(ctx, nbytes=W["cloud"].numel() * 4, borrow=W["cloud"].data_ptr())
            ext = (0, grid - 1, 0, grid - 1, 0, grid - 1)
            bp = dict(max_splats=args.partition_max_splats, max_cells=255, chunk_cells=0, micro_cells=63, max_split=1 << 30)
            leaves = mb.bucket_cloud(ctx, raw, n_splats, (0.0, 0.0, 0.0), 1.0, ext, on_bucket=lambda leaf, ids: None, **bp)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(L):
                mb.bucket_cloud(ctx, raw, n_splats, (0.0, 0.0, 0.0), 1.0, ext, on_bucket=lambda leaf, ids: None, **bp)
            ctx.synchronize()
            part_s = (time.perf_counter() - t0) / L
            pmax = max(l["num_splats"] for l in leaves)
            pcells = max(max(l["extents"][2 * i + 1] - l["extents"][2 * i] for i in range(3)) for l in leaves)
            # the bucketer's callback hands every leaf to the bucket farm's device path (gather + transform kernel into a
            # device item, then the farm's worker threads), as CopyGroup does with host buckets
            # the leaves arrive one per device item; with lanes a worker takes as many queued items as fit a batch
            # through one set of launches (mlsgpu_hip_farm_set_batch), so fewer workers and more spare items
            # the reference partition's buckets are small (126-cell cubes where the 27-bucket split has 170-cell ones): a worker
            # takes up to MAX_BATCH of them through one set of octree launches, and as many as hold two full buckets' corners
            # through one set of processCorners / marching launches (mlsgpu_hip_worker_set_marching_group)
            pbatch = 1 if args.batch == 1 else m.binding.MAX_BATCH
            pworkers = max(1, args.farm_workers) if pbatch == 1 else max(1, args.partition_workers)
            pfarm = m.BucketFarm([local_rank], pmax, workers_per_device=pworkers, spare=1 if pbatch == 1 else pbatch * pworkers,
                                 max_cells=pcells, mesh_memory=args.mesh_memory_mb << 20)
            pfarm.set_batch(pbatch)
            leaf_no = [0]

            def leaf_work(leaf, d_ids):
                low = leaf["extents"][0::2]
                nv = [leaf["extents"][2 * i + 1] - leaf["extents"][2 * i] + 1 for i in range(3)]
                leaf_no[0] += 1
                return pfarm.submit_device(local_rank, raw, d_ids, leaf["num_splats"], (0.0, 0.0, 0.0), 1.0, ext, low, nv,
                                           leaf_no[0] - 1, wait=False)

            def partition_pass():
                mb.bucket_cloud(ctx, raw, n_splats, (0.0, 0.0, 0.0), 1.0, ext, on_bucket=leaf_work, **bp)
                pfarm.finish()
            partition_pass()                            # warm-up
            wc0 = pfarm.worker_clock()
            feed_s = 0.0
            t0 = time.perf_counter()
            for _ in range(L):
                t1 = time.perf_counter()
                mb.bucket_cloud(ctx, raw, n_splats, (0.0, 0.0, 0.0), 1.0, ext, on_bucket=leaf_work, **bp)
                feed_s += time.perf_counter() - t1
                pfarm.finish()
            pipe_s = (time.perf_counter() - t0) / L
            wc1 = pfarm.worker_clock()
            # ... and as a stream of jobs: the bucketing of job k + 1 (on this context's stream, between its host decisions)
            # shares the GPU with the workers still on job k's buckets
            t0 = time.perf_counter()
            for _ in range(L):
                mb.bucket_cloud(ctx, raw, n_splats, (0.0, 0.0, 0.0), 1.0, ext, on_bucket=leaf_work, **bp)
            pfarm.finish()
            stream_s = (time.perf_counter() - t0) / L
            pvox = sum((l["extents"][1] - l["extents"][0]) * (l["extents"][3] - l["extents"][2]) * (l["extents"][5] - l["extents"][4])
                       for l in leaves)
            result["device_partition"] = {
                "buckets": len(leaves), "bucket_splats_total": int(sum(l["num_splats"] for l in leaves)),
                "max_bucket_cells": int(pcells), "bucketing_ms": round(part_s * 1e3, 3),
                "bucketing_msplats_per_s": round(n_splats / part_s / 1e6, 1),
                "pipeline_ms_per_step": round(pipe_s * 1e3, 3), "pipeline_mvoxels_per_s": round(pvox / pipe_s / 1e6, 3),
                "streamed_ms_per_step": round(stream_s * 1e3, 3), "streamed_mvoxels_per_s": round(pvox / stream_s / 1e6, 3),
                "bucketing_and_feeding_ms": round(feed_s / L * 1e3, 3),
                "buckets_per_launch_set": round((wc1["buckets"] - wc0["buckets"]) / max(wc1["launch_sets"] - wc0["launch_sets"], 1), 2),
                "workers_idle_ms_per_step": round((wc1["idle_s"] - wc0["idle_s"]) / L * 1e3, 2),
                "workers_busy_ms_per_step": round((wc1["busy_s"] - wc0["busy_s"]) / L * 1e3, 2),
                "device_workers": pworkers, "batch": pbatch,
                "note": "raw cloud resident in HBM -> mlsgpu_hip_bucket (reference partition) -> mlsgpu_hip_farm_submit_device "
                        "(device gather + transform) -> the farm's device workers; bucketing is inside the pipeline time",
            }
            pfarm.close()
            del raw
        except Exception as e:      # noqa: BLE001 - reported in the line
            result.setdefault("leg_errors", {})['device_partition'] = "%s: %s" % (type(e).__name__, e)
    W["cloud"] = None

    # the remaining legs start from HOST memory: one copy of the bucketed splats
    bucketed_host = None
    if secondary and not args.no_transfer:
        bucketed_host = synth.to_host_splats(bucketed_t)
    del workers, work, pristine, bucketed_t, W
    torch.cuda.empty_cache()

    # the sink route's pinned landing buffers, sized once for the larger of the meshes (this workload's): allocated now, before
    # any leg has pinned and freed host memory, and kept to the end
    if secondary and not args.no_transfer and rank == 0 and world == 1:
        try:
            need = int(1.02 * 12 * (triangles + vertices)) + (1 << 20)
            while len(_SINK_PINS) < 3:
                _SINK_PINS.append(m.binding.PinnedBuffer(need))
        except Exception:      # noqa: BLE001 - the legs grow their buffers themselves
            pass

    # ---- D1 ("shells", SURVEY 8d: report both): resident rate and the same transfer-inclusive legs ----
    if secondary and not args.no_shells and args.dist == "uniform" and args.workload in ("auto", "cfg3"):
        try:     # a secondary leg never costs the line its headline
            cloud, g = synth.make_cloud_device("cfg3", device, scale=args.scale, dist="shells")
            sb_t, sbuckets = synth.bucketize_device(cloud, synth.grid_buckets((g, g, g), 255))
            del cloud
            torch.cuda.synchronize()
            smax = max(b.count for b in sbuckets)
            scells = max(max(b.num_vertices) for b in sbuckets) - 1
            svox = sum(b.cells for b in sbuckets)
            sprist = m.DeviceBuffer(ctx, nbytes=sb_t.numel() * 4, borrow=sb_t.data_ptr())
            swork = m.DeviceBuffer(ctx, nbytes=sb_t.numel() * 4)
            sworkers = [m.Worker(c, smax, max_cells=scells, mesh_memory=args.mesh_memory_mb << 20) for c in ctxs]
            scol = [m.binding.SizeCollector() for _ in range(nworkers)]

            def s_share(k):
                for b in farm.worker_share(sbuckets, k, nworkers):
                    sworkers[k].process(swork, b.first, b.count, b.low, b.num_vertices, collector=scol[k])
                ctxs[k].synchronize()

            def s_step():
                swork.copy_from(sprist)
                ctx.synchronize()
                list(pool.map(s_share, range(nworkers)))
            for _ in range(2):
                s_step()
            ssteps = max(10, min(args.steps, 50))
            scol[:] = [m.binding.SizeCollector() for _ in range(nworkers)]
            t0 = time.perf_counter()
            for _ in range(ssteps):
                s_step()
            s_dt = (time.perf_counter() - t0) / ssteps
            shells = {"value": round(svox / s_dt / 1e6, 3), "unit": "Mvoxels/s", "ms_per_step": round(s_dt * 1e3, 3), "steps": ssteps,
                      "workload": "cfg3 grid, %d splats on concentric shells (D1 of SURVEY 8d), %d buckets" % (int(50_000_000 * args.scale), len(sbuckets)),
                      "triangles_per_step": sum(c.triangles for c in scol) // ssteps,
                      "vertices_per_step": sum(c.vertices for c in scol) // ssteps}
            sb_host = synth.to_host_splats(sb_t) if not args.no_transfer else None
            del sworkers, swork, sprist, sb_t
            torch.cuda.empty_cache()
            if not args.no_transfer:
                shells["transfer_inclusive"] = transfer_legs(m, args, local_rank, sb_host, sbuckets, smax, scells, svox, L)
                shells["transfer_inclusive"]["host_weld"] = host_weld_leg(m, args, local_rank, sb_host, sbuckets, smax, scells, svox)
            result["shells"] = shells
        except Exception as e:      # noqa: BLE001 - reported in the line
            result.setdefault("leg_errors", {})['shells'] = "%s: %s" % (type(e).__name__, e)

    # (the shells legs run BEFORE the noise cloud's transfer legs: those pin and free tens of GB of host memory, after which
    # freshly allocated host arrays are slower to copy from -- profiles/NOTES_r04.md section 9.10)
    # ---- transfer-inclusive legs (never `value`): SURVEY 8(d)'s region, host splats in -> last mesh byte out ----
    if secondary and not args.no_transfer:
        try:     # a secondary leg never costs the line its headline
            result["transfer_inclusive"] = transfer_legs(m, args, local_rank, bucketed_host, buckets, max_count, max_cells, voxels, L)
            result["transfer_inclusive"]["distribution"] = args.dist
        except Exception as e:      # noqa: BLE001 - reported in the line
            result.setdefault("leg_errors", {})['transfer_inclusive'] = "%s: %s" % (type(e).__name__, e)

    # ---- CPU baseline: the oracle ("port") parallel over buckets on the host cores, rank 0 at N = 1 only ----
    # ---- SURVEY 8(d)'s own timed region at the top level of the line (never `value`: the contract wants inputs resident) ----
    if secondary and "transfer_inclusive" in result:
        v8 = {"uniform" if args.dist == "uniform" else args.dist: result["transfer_inclusive"]["shipouts"]["value"]}
        if "shells" in result and "transfer_inclusive" in result["shells"]:
            v8["shells"] = result["shells"]["transfer_inclusive"]["shipouts"]["value"]
        v8["unit"] = "Mvoxels/s"
        v8["what"] = ("host splats in -> last mesh byte back in host memory, steady state, every ship-out read back through the "
                      "pinned ring (SURVEY 8d's region; the noise cloud's 13.6 GB of mesh per step is bounded by the PCIe link)")
        result["value_8d_region"] = v8
    if rank == 0 and secondary and cpu_sample is not None:
        try:     # a secondary leg never costs the line its headline
            cb = cpu_baseline(cpu_sample[0], cpu_sample[1], 63)
            if cb is not None:
                result["cpu_baseline"] = cb
                result["gpu_over_cpu"] = round(value / cb["value"], 1)
        except Exception as e:      # noqa: BLE001 - reported in the line
            result.setdefault("leg_errors", {})['cpu_baseline'] = "%s: %s" % (type(e).__name__, e)

    result.pop("_grid", None)
    if rank == 0:
        print(json.dumps(result))
    pool.shutdown()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
