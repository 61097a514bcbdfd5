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

The secondary legs (never `value`) live in benchlegs.py; a leg that fails or no longer fits the legs' wall-clock budget
(--leg-budget-s) is named in `leg_errors`, and the headline leaves on stderr before the first of them starts.

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
CFG2_PINS = {}      # --workload cfg2: the one-bucket configuration, pinned next to whole-bucket oracle parity (test_cfg2_full_size)
for _d in ("uniform", "shells"):
    try:
        CFG2_PINS[_d] = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg2_%s.json" % _d)))["digest"]
    except Exception:   # noqa: BLE001
        pass
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
    p.add_argument("--reader-threads", type=int, default=16,
                   help="cfg5: host threads reading the PLY files (8-16 reach the link's rate; 24 and 32 were 20-35 %% slower)")
    p.add_argument("--cfg5-spare", type=int, default=10, help="device items beyond one per worker in cfg5's farm")
    p.add_argument("--cfg5-batch", type=int, default=4,
                   help="buckets per set of launches in cfg5's farm (ms per pass, same box: 1: 559-563, 2: 540-543, 3: 533-534, 4: 530-533, "
                        "6: 530-531, 8: 531-532)")
    p.add_argument("--dist", default="uniform", choices=["uniform", "shells"])
    p.add_argument("--scale", type=float, default=1.0, help="splat-count scale (debug only; 1.0 = BASELINE size)")
    p.add_argument("--mesh-memory-mb", type=int, default=4096, help="Marching mesh arena per worker")
    p.add_argument("--workers", type=int, default=3,
                   help="device worker threads per GPU for the resident-input passes (round 3, cfg3 uniform: 1 / 2 / 3 / 4 / 6 workers "
                        "23.1 / 21.2-21.6 / 21.6-22.2 / 22.2-22.5 / 22.1 ms per step -- every kernel fills the GPU, a second worker "
                        "hides the host's gaps and more only interleave.  End of round 6, processCorners 13 %% shorter: 2 workers x "
                        "batches of 6 / 8 14.05 / 14.06 ms, 3 workers 13.95 / 13.89, four rounds on one box)")
    p.add_argument("--farm-workers", type=int, default=2,
                   help="device workers per GPU of the farm legs (host splats in, meshes out).  Round 5, shells cloud, 8d region with "
                        "six spare items: 2 workers 33.5-34 ms per job (the host-to-device copies busy 0.90-0.92 of it), 4 workers "
                        "36.5-38 (0.80), 8 workers 41 (0.74): the link is the floor, and fewer streams queue less in front of it")
    p.add_argument("--batch", type=int, default=8,
                   help="buckets a device worker takes through the path in lock-step (mlsgpu_hip_worker_process_batch: every "
                        "kernel has a bucket dimension, one set of launches and three host decisions per batch); 1 = bucket by "
                        "bucket (mlsgpu_hip_worker_process).  With the octree's kernels a quarter shorter than when 4 was chosen, "
                        "6 or 8 read 1 %% better on every workload (cfg3 16.13 -> 15.9-16.0 ms, shells 11.27 -> 11.09, cfg4 slab "
                        "10.16 -> 10.0)")
    p.add_argument("--marching-group", type=int, default=2,
                   help="of a batch's buckets, how many share one set of processCorners / marching launches (the octree build "
                        "takes the whole batch); 0 = all (mlsgpu_hip_worker_set_marching_group)")
    p.add_argument("--variant", type=int, default=5, choices=[1, 4, 5],
                   help="MLS kernel: 5 sub-block culling + matrix-core prefilter (default), 4 sub-block culling + cube streams, 1 the reference's structure")
    p.add_argument("--dispatch", default="ranks", choices=["ranks", "greedy"],
                   help="ranks (default; what the driver launches): one process per GPU, rank r = z-slab r of the cfg4 cloud.  "
                        "greedy: ONE process, a device worker group per GPU, the buckets of the WHOLE cfg4 cloud handed out by the "
                        "reference's rule (the group with the most unallocated capacity, src/workers.cpp:320-351), every bucket "
                        "checked against its pin; MLSGPU_TEST_DEVICES=0,0,... runs it on a one-GPU box")
    p.add_argument("--leg-steps", type=int, default=3, help="passes of every secondary leg")
    p.add_argument("--legs", default="all", choices=["all", "none"],
                   help="none: only the timed region, its roofline and (N > 1) the in-run per-GPU reference; all: the secondary legs "
                        "too (never `value`)")
    p.add_argument("--leg-budget-s", type=float, default=None,
                   help="wall-clock budget of ALL secondary legs together (default: 150 s at N > 1, 300 s at N = 1): a leg that would "
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
    p.add_argument("--partition-spare", type=int, default=0, help="device items of the device-bucketer leg's farm beyond one per worker (0: 4 x lanes x workers -- the feeder runs far enough ahead for the workers to find full batches: 6.4 buckets per set of launches against 4.6 with lanes x workers items)")
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


from benchlegs import (CFG5_PARTITION, cfg5_paths, cpu_baseline, cpu_sample_boxes, drain_utilisation,  # noqa: E402,F401
                       host_weld_leg, multi_gpu_legs, run_cfg5, single_process_leg, transfer_legs, _SINK_PINS)


def main():
    args = parse_args()
    if args.dispatch == "greedy":
        from benchlegs import run_greedy
        if args.steps == 200:
            args.steps = 10          # a pass moves the whole cloud over the links: the default K of the resident headline is too long
        return run_greedy(args)
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
        # at FOUR buckets per set of launches, the shape every round's per-stage figures are quoted on (with more, a lone
        # instrumented worker's GPU runs dry between sets and an event pair then times the host's next submission)
        workers[0].set_batch(min(batch, 4))
        ctx.reset_stats()
        ctx.set_timing(True)
        for _ in range(ksteps):
            work.copy_from(pristine)
            run_buckets(workers[0], ctx, buckets, m.binding.SizeCollector())
        ctx.set_timing(False)
        kernel_stats = dict(ctx.stats())
        workers[0].set_batch(batch)

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
    # (MLSGPU_BENCH_MLS_STATS_WORDS: room for what an instrumented library writes behind the counters, tools/drain_sim.py)
    counters = m.DeviceBuffer(ctx, array=np.zeros(int(os.environ.get("MLSGPU_BENCH_MLS_STATS_WORDS", m.binding.MLS_STATS_WORDS)), np.uint64))
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
    if os.environ.get("MLSGPU_BENCH_MLS_STATS_FILE"):
        np.savez_compressed(os.environ["MLSGPU_BENCH_MLS_STATS_FILE"], words=np.array(mls_counters, np.uint64))
    if os.environ.get("MLSGPU_BENCH_DUMP_MLS_COUNTERS"):
        # the raw words (tools/mls_clock.sh: a library built with -DMLSGPU_MLS5_CLOCK leaves a wave's cycle sums in words 0-7)
        print("mls counters:", " ".join(str(x) for x in mls_counters[:64]), file=sys.stderr, flush=True)
    listed, tests, hits = mls_counters[:3]
    if args.variant == 5 and mls_counters[42] != 0:
        raise SystemExit("processCorners: the matrix prefilter missed %d hits of the reference's test" % mls_counters[42])
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
            "mls_variant": {1: "basic", 4: "culled+cube-streams", 5: "culled+matrix-prefilter"}[args.variant],
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
    if W["name"] == "cfg2" and args.scale == 1.0 and args.dist in CFG2_PINS:
        result["output_digest"]["expected"] = CFG2_PINS[args.dist]
        result["output_digest"]["ok"] = digest == CFG2_PINS[args.dist]
        if digest != CFG2_PINS[args.dist]:
            raise SystemExit("output digest %s differs from the pinned %s (tests/golden/cfg2_%s.json)"
                             % (digest, CFG2_PINS[args.dist], args.dist))
    # EVERY rank's slab is pinned (slab r as an inner slab of 128 cell slices, or as the job's last one of 127): each rank holds
    # its own digest against its pin, the verdicts are gathered, and one mismatch anywhere fails the whole run
    digest_failure = None
    if W["name"] == "cfg4slab" and args.scale == 1.0 and CFG4SLAB_PINS is not None:
        v = farm.slab_verdicts(CFG4SLAB_PINS, world, rank, digest, check.vertices, check.triangles, dist, reduce_device)
        result["output_digest"].update(expected=v["expected"][0], ok=v["ok"][0], per_rank_digest=v["digests"],
                                       per_rank_expected=v["expected"], per_rank_ok=v["ok"], all_ok=v["all_ok"])
        if not v["all_ok"]:
            digest_failure = ("output digests %s of the ranks' slabs differ from the pinned %s (tests/golden/cfg4slab_uniform.json)"
                              % (v["digests"], v["expected"]))
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
            # kernel 5 finds its candidates on the matrix pipe (32 flops per (corner, splat) pair of a 32 x 32 x 16 bf16 MFMA);
            # the vector side then runs the reference's test on the candidates only
            candidates = int(mls_counters[41]) if args.variant == 5 else tests
            done_flops = 10 * candidates + 25 * hits
            achieved = done_flops / (ms * 1e-3) / 1e12
            sq = {}
            try:
                sq = json.load(open(os.path.join(ROOT, "profiles", "sq.json"))).get(
                    "%s/%s/processCorners/variant%d" % (W["name"], args.dist, args.variant), {})
            except Exception:
                sq = {}
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
                "executed": "vector flops: 10 per (corner, splat) test the vector units execute (kernel 5: the matrix prefilter's "
                            "candidates; kernels 4 and 1: every lane-test that survives culling) + 25 per hit",
                "reading_frac": "frac counts EXECUTED vector flops, so it falls when finer culling or the matrix prefilter removes "
                                "tests: rounds 1-2 executed 20.9 G lane-tests per step for these 3.47 G hits (frac 0.18), the cube "
                                "streams of rounds 3-5 9.3 G (0.14), kernel 5 tests its 3.5 G candidates only; what bounds the "
                                "kernel is issuing vector instructions: valu_issue_frac (SQ counters of the timed kernel, "
                                "profiles/sq.json) is the figure to read, avg_launch_ms and algorithmic_equiv_frac the ones "
                                "that compare across rounds",
                "valu_issue_frac": sq.get("valu_issue_frac"),
                "valu_issue": ({k: sq[k] for k in ("valu_issue_frac_of_measured_rate", "insts_valu", "insts_mfma", "kernel_cycles",
                                                   "mfma_busy_frac", "lds_busy_frac", "active_lane_frac", "wave_wait_frac", "source",
                                                   "label") if k in sq} or None),
                "matrix_prefilter": ({"pairs_evaluated": tests, "candidates": candidates,
                                      "candidates_per_hit": round(candidates / max(hits, 1), 4), "missed_hits": int(mls_counters[42]),
                                      "matrix_TFLOPs": round(32 * tests / (ms * 1e-3) / 1e12, 2),
                                      "what": "bf16 MFMA (exact products of three-piece splits, f32 accumulation) marks a superset of "
                                              "the reference's hits; the accumulation runs under the reference's own d < 0.99"}
                                     if args.variant == 5 else None),
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
        result["kernel_ms_per_step"] = {k: round(v[0] / K, 3) for k, v in sorted(kernel_stats.items())
                                        if k.startswith(("kernel.", "device."))}       # (the registry also holds the reference's counters)
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
    if rank == 0 and not args.headline_only:
        early = {k: result[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "per_gpu_reference",
                                        "scaling_efficiency") if k in result}
        print("bench.py headline before the secondary legs: " + json.dumps(early), file=sys.stderr, flush=True)
    if args.leg_budget_s is None:
        args.leg_budget_s = 150.0 if dist is not None else 300.0        # the N = 1 legs take about a minute together
    leg_deadline = time.perf_counter() + args.leg_budget_s

    def leg_fits(name):
        """N = 1: a secondary leg starts only while the legs' wall-clock budget lasts; one that does not is named in leg_errors."""
        if time.perf_counter() < leg_deadline:
            return True
        result.setdefault("leg_errors", {})[name] = "skipped: the secondary legs' budget of %.0f s (--leg-budget-s) was spent" \
            % args.leg_budget_s
        return False
    if dist is not None and not args.headline_only:
        multi_gpu_legs(m, args, result, dist, park, reduce_device, rank, world, local_rank, ndev, ctxs[0], bucketed_t, buckets,
                       max_count, max_cells, voxels, L, max(1, args.farm_workers), leg_deadline)

    # ---- mesh-sink leg (never `value`): every ship-out of one pass appended to the device mesher (d2d), then
    # finalize = weld by key across buckets + connected components + prune (--fit-prune default 0.02) + compaction ----
    if secondary and not args.no_sink and leg_fits("mesh_sink"):
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
    if secondary and not args.no_partition and W["cloud"] is not None and leg_fits("device_partition"):
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
            pfarm = m.BucketFarm([local_rank], pmax, workers_per_device=pworkers, spare=args.partition_spare if args.partition_spare > 0 else 1 if pbatch == 1 else 4 * pbatch * pworkers,
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
    if secondary and not args.no_shells and args.dist == "uniform" and args.workload in ("auto", "cfg3") and leg_fits("shells"):
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
            # the HEADLINE's protocol (round 5 ran this leg on the old one -- a restore copy and a host synchronisation inside
            # the loop, bucket by bucket, mutating workers -- and printed a number 17-20 % below the one the same cloud gives as
            # `--dist shells`): resident splats processed in place by non-mutating workers, `batch` buckets per set of launches
            sworkers = [m.Worker(c, smax, max_cells=scells, mesh_memory=args.mesh_memory_mb << 20) for c in ctxs]
            sbatch = max(1, min(args.batch, m.binding.MAX_BATCH, -(-len(sbuckets) // nworkers)))
            for w in sworkers:
                w.set_mls_variant(args.variant)
                w.set_keep_splats(True)
                w.set_batch(sbatch)
                w.set_marching_group(max(0, min(args.marching_group, m.binding.MAX_BATCH)))
            scol = [m.binding.SizeCollector() for _ in range(nworkers)]
            sshares = [list(farm.worker_share(sbuckets, k, nworkers)) for k in range(nworkers)]

            def s_share(k):
                if sbatch > 1:
                    sworkers[k].process_batch(sprist, sshares[k], collector=scol[k])
                else:
                    for b in sshares[k]:
                        sworkers[k].process(sprist, b.first, b.count, b.low, b.num_vertices, collector=scol[k])
                ctxs[k].synchronize()

            def s_step():
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
                      "input_protocol": "resident_in_place (the headline's: non-mutating workers, batches of %d, no restore copy)" % sbatch,
                      "triangles_per_step": sum(c.triangles for c in scol) // ssteps,
                      "vertices_per_step": sum(c.vertices for c in scol) // ssteps}
            # the other splat protocol (mutating build, every pass starts from a restored copy: the reference's work items
            # arrive by H2D copy), a few passes, never the leg's `value`
            for w in sworkers:
                w.set_keep_splats(False)
            swork = m.DeviceBuffer(ctx, nbytes=sb_t.numel() * 4)

            def s_share_restore(k):
                for b in sshares[k]:
                    m.binding.check(m.lib().mlsgpu_hip_memcpy_d2d(ctxs[k].h, swork.ptr + 32 * b.first, sprist.ptr + 32 * b.first, 32 * b.count))
                if sbatch > 1:
                    sworkers[k].process_batch(swork, sshares[k], collector=scol[k])
                else:
                    for b in sshares[k]:
                        sworkers[k].process(swork, b.first, b.count, b.low, b.num_vertices, collector=scol[k])
                ctxs[k].synchronize()
            list(pool.map(s_share_restore, range(nworkers)))
            t0 = time.perf_counter()
            for _ in range(3):
                list(pool.map(s_share_restore, range(nworkers)))
            shells["other_splat_protocol"] = {"mode": "restore_per_bucket", "ms_per_step": round((time.perf_counter() - t0) / 3 * 1e3, 3)}
            sb_host = synth.to_host_splats(sb_t) if not args.no_transfer else None
            del sworkers, swork, sprist, sb_t
            torch.cuda.empty_cache()
            if not args.no_transfer:
                shells["transfer_inclusive"] = transfer_legs(m, args, local_rank, sb_host, sbuckets, smax, scells, svox, L)
                shells["transfer_inclusive"]["host_weld"] = host_weld_leg(m, args, local_rank, sb_host, sbuckets, smax, scells, svox)
                shells["transfer_inclusive"]["host_weld_ring"] = host_weld_leg(m, args, local_rank, sb_host, sbuckets, smax, scells, svox,
                                                                               landing=False)
            result["shells"] = shells
        except Exception as e:      # noqa: BLE001 - reported in the line
            result.setdefault("leg_errors", {})['shells'] = "%s: %s" % (type(e).__name__, e)

    # (the shells legs run BEFORE the noise cloud's transfer legs: those pin and free tens of GB of host memory, after which
    # freshly allocated host arrays are slower to copy from -- profiles/NOTES_r04.md section 9.10)
    # ---- transfer-inclusive legs (never `value`): SURVEY 8(d)'s region, host splats in -> last mesh byte out ----
    if secondary and not args.no_transfer and leg_fits("transfer_inclusive"):
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
    if rank == 0 and secondary and cpu_sample is not None and leg_fits("cpu_baseline"):
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
