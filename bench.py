#!/usr/bin/env python3
"""Headline benchmark: Mvoxels/s evaluated + triangulated by the per-bucket device pipeline.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

One "step" is one pass of the hot path (octree build -> MLS corner evaluation -> marching tetrahedra
with welding -> scale/bias) over every bucket of one synthetic splat cloud whose splats are already
resident in HBM.  At N = 1 the workload is BASELINE.json configs[2] (512^3 grid, 50 M uniform-random
splats, multi-bucket stream), the configuration the north_star target is quoted on; with N > 1 every
rank streams its own cloud of the same shape (weak scaling, no data-path collective: buckets are
independent, cross-bucket welding is host work in the reference).

Per GPU, `--workers` device worker threads (default 2, the reference's --device-threads,
src/mlsgpu_core.cpp:114) each own a stream, an octree, an MLS functor and a Marching instance and take
alternate buckets, so one worker's host synchronisations overlap the other's kernels.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector (= dense f32 MFMA rate)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3"])
    p.add_argument("--dist", default="uniform", choices=["uniform", "shells"])
    p.add_argument("--scale", type=float, default=1.0, help="splat-count scale (debug only; 1.0 = BASELINE size)")
    p.add_argument("--mesh-memory-mb", type=int, default=4096, help="Marching mesh arena per worker")
    p.add_argument("--workers", type=int, default=4, help="device worker threads per GPU (measured 2..4: +0..6 %)")
    p.add_argument("--variant", type=int, default=2, help="MLS kernel variant: 0 culled, 1 basic, 2 culled + hit lists")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-stream", action="store_true", help="skip the PCIe-inclusive bucket-farm leg")
    p.add_argument("--no-partition", action="store_true", help="skip the device-bucketer leg (reference partition)")
    p.add_argument("--partition-max-splats", type=int, default=2097152,
                   help="bucket capacity of the device-bucketer leg (reference default 64 MiB / 32 B)")
    p.add_argument("--no-sink", action="store_true", help="skip the device mesh-sink leg (weld / components / prune)")
    p.add_argument("--no-timing", action="store_true", help="do not time individual kernels with HIP events")
    return p.parse_args()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one process per GPU; MLSGPU_BENCH_BACKEND=gloo lets several ranks share one GPU (a single-GPU check of the
    # N > 1 code path: the only collectives are a barrier and two scalar reductions, so RCCL is not essential)
    backend = os.environ.get("MLSGPU_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= ndev:
        raise SystemExit("LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, ndev))
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    reduce_device = "cuda" if (dist is not None and backend == "nccl") else None

    import mlsgpu_amd as m
    from mlsgpu_amd import farm, synth

    # ---- workload (host side, untimed) ----
    t0 = time.time()
    cloud, grid = synth.make_cloud(args.workload, args.dist, scale=args.scale, seed_offset=rank)
    bucketed, buckets = synth.bucketize(cloud, grid, 255)
    n_splats = len(cloud)
    voxels = sum(b.cells for b in buckets)
    max_count = max(b.count for b in buckets)
    max_cells = max(max(b.num_vertices) for b in buckets) - 1
    setup_s = time.time() - t0

    nworkers = max(1, min(args.workers, len(buckets)))
    ctxs = [m.Context(local_rank) for _ in range(nworkers)]
    ctx = ctxs[0]
    pristine = m.DeviceBuffer(ctx, array=bucketed)
    work = m.DeviceBuffer(ctx, nbytes=bucketed.nbytes)
    workers = [m.Worker(c, max_count, max_cells=max_cells, mesh_memory=args.mesh_memory_mb << 20) for c in ctxs]
    for w in workers:
        w.set_mls_variant(args.variant)
    pool = ThreadPoolExecutor(nworkers)
    collectors = [m.binding.SizeCollector() for _ in range(nworkers)]

    def run_share(k):
        # worker k takes buckets k, k + nworkers, ... (ctypes releases the GIL inside the library)
        w, col = workers[k], collectors[k]
        for b in farm.worker_share(buckets, k, nworkers):
            w.process(work, b.first, b.count, b.low, b.num_vertices, collector=col)
        ctxs[k].synchronize()

    def step():
        # The octree build overwrites splat.w with 1/r^2 (kernels/octree.cl:193), so each pass starts from a
        # fresh copy of the resident splats: a device-to-device copy standing where the reference has its
        # host-to-device copy (src/workers.cpp:356-361).  It is inside the timed region.
        work.copy_from(pristine)
        ctx.synchronize()
        list(pool.map(run_share, range(nworkers)))

    def barrier():
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    collectors[:] = [m.binding.SizeCollector() for _ in range(nworkers)]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    for c in ctxs:
        c.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    # whole job: MAX of the elapsed time over ranks, SUM of the voxels (each rank ran `steps` passes of its cloud)
    elapsed, total_voxels, _ = farm.combine(elapsed, voxels * args.steps, dist, reduce_device)

    # ---- per-kernel durations: the same `steps` passes once more on ONE worker with HIP events around every
    # launch.  Kept out of the headline region because with several workers the streams overlap and an event
    # pair then measures a kernel sharing the GPU, not the kernel; single-worker durations are what the
    # roofline divides by (and what `rocprofv3 --kernel-trace --stats ... --workers 1` reports). ----
    kernel_stats = {}
    if not args.no_timing:
        ctx.reset_stats()
        ctx.set_timing(True)
        for _ in range(args.steps):
            work.copy_from(pristine)
            for b in buckets:
                workers[0].process(work, b.first, b.count, b.low, b.num_vertices, collector=m.binding.SizeCollector())
        ctx.set_timing(False)
        kernel_stats = dict(ctx.stats())
    triangles = sum(c.triangles for c in collectors) // max(args.steps, 1)
    vertices = sum(c.vertices for c in collectors) // max(args.steps, 1)
    external = sum(c.external for c in collectors) // max(args.steps, 1)
    shipouts = sum(c.batches for c in collectors) // max(args.steps, 1)

    # ---- algorithmic work (one instrumented, untimed pass on worker 0) ----
    counters = m.DeviceBuffer(ctx, array=np.zeros(3, np.uint64))
    w0 = workers[0]
    before = w0.marching_counters()
    w0.set_mls_stats(counters)
    work.copy_from(pristine)
    corners = entries = 0
    for b in buckets:
        w0.process(work, b.first, b.count, b.low, b.num_vertices, collector=m.binding.SizeCollector())
        entries += w0.tree_num_entries()
        corners += int(np.prod([-(-n // 8) * 8 for n in b.num_vertices]))
    ctx.synchronize()
    listed, tests, hits = (int(x) for x in counters.download(np.uint64))
    w0.set_mls_stats(None)
    after = w0.marching_counters()
    mc = {k: after[k] - before[k] for k in after}

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
        "msplats_per_s": round(world * n_splats * args.steps / elapsed / 1e6, 3),
        "config": {
            "workload": "%s: %d^3 grid, %d splats (%s), %d buckets of <= %d cells per side, octree+MLS+MC end-to-end"
                        % (args.workload, grid, n_splats, args.dist, len(buckets), max_cells),
            "voxels_per_step": voxels,
            "bucket_splats_total": len(bucketed),
            "mesh_memory_mb": args.mesh_memory_mb,
            "device_workers": nworkers,
            "mls_variant": {0: "culled", 1: "basic", 2: "culled+hit-lists"}[args.variant],
            "per_rank": "own cloud per rank (seed offset = rank)",
            "triangles_per_step": triangles,
            "vertices_per_step": vertices,
            "shipouts_per_step": shipouts,
            "host_setup_s": round(setup_s, 1),
        },
    }

    # ---- roofline: algorithmic bytes over HIP-event kernel time, per stage ----
    if kernel_stats and not args.no_timing:
        K = args.steps
        T, O, Vw, C = 3 * triangles, mc["occupied"], vertices, voxels
        sort_passes = 2       # 17 key bits at <= 10 bits per pass
        models = {
            # stat name: (kernel, algorithmic bytes per step)
            "kernel.mls.processCorners.time": ("processCorners", 36 * listed + 4 * corners),
            "kernel.octree.sort.time": ("sortHist+sortScatter (octree entries)", sort_passes * 20 * entries),
            "kernel.octree.writeEntries.time": ("writeEntries (count+scan+write)", 3 * 16 * len(bucketed) + 8 * entries),
            "kernel.octree.scan.time": ("countCommands+scan+writeSplatIds", 2 * 4 * entries + 8 * entries + 4 * entries),
            "kernel.marching.generateElements.time": ("latticeTriangles", 4 * T + 16 * O + O),
            "kernel.marching.compactVertices.time": ("latticeVertices", 12 * Vw + 8 * external + 8 * Vw),
            "kernel.marching.countUniqueVertices.time": ("latticeMask", C + 8 * 12 * (corners // 64)),
            "kernel.marching.genOccupied.time": ("cellCode+classify", 4 * corners + C + C),
        }
        stages = []
        for stat, (kname, nbytes) in models.items():
            if stat in kernel_stats and kernel_stats[stat][1] > 0:
                ms = kernel_stats[stat][0] / K
                stages.append({"stat": stat, "kernel": kname, "ms_per_step": round(ms, 3),
                               "launches_per_step": kernel_stats[stat][1] // K,
                               "algorithmic_bytes_per_step": int(nbytes),
                               "achieved_GBps": round(nbytes / (ms * 1e-3) / 1e9, 1)})
        stages.sort(key=lambda s: -s["ms_per_step"])
        single_ms = sum(v[0] for k, v in kernel_stats.items() if k == "device.compute") / K
        tj = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath)).get("%s/%s" % (args.workload, args.dist), {})
            except Exception:
                tj = {}
        for st in stages:
            st["hbm_traffic_bytes_per_launch"] = tj.get(st["kernel"])
        pc = "kernel.mls.processCorners.time"
        if stages and stages[0]["stat"] == pc:
            # Dominant kernel = processCorners: an fp32-VALU/LDS-bound kernel (>= 140 flop/B), so its
            # roof is the fp32 rate.  MI355X's dense f32 MFMA peak equals its f32 vector peak (157.3 TFLOP/s,
            # MI355X_MICROARCH.md); the kernel issues no MFMA, "mfma" here only names the compute roof.
            total_ms, launches = kernel_stats[pc]
            ms = total_ms / K
            alg_flops = 10 * 512 * listed + 25 * hits          # SURVEY 8d per-bucket figure, summed over buckets
            done_flops = 10 * tests + 25 * hits                # distance tests that survive sub-block culling
            achieved = alg_flops / (ms * 1e-3) / 1e12
            result["roofline"] = {
                "kernel": "processCorners",
                "bound": "mfma",
                "achieved": round(achieved, 3),
                "peak": FP32_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / FP32_VALU_PEAK_TFLOPS, 4),
                "traffic": tj.get("processCorners"),
                "avg_launch_ms": round(total_ms / launches, 4),
                "launches_per_step": launches // K,
                "algorithmic_flops_per_launch": int(alg_flops // max(launches // K, 1)),
                "algorithmic": "SURVEY 8d: 10*512*SigmaL + 25*H flops of the reference's every-corner-tests-every-listed-"
                               "splat loop; sub-block culling skips most of those tests, see executed_*",
                "executed_TFLOPs": round(done_flops / (ms * 1e-3) / 1e12, 3),
                "executed_frac": round(done_flops / (ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS, 4),
                "hbm_algorithmic_GBps": round((36 * listed + 4 * corners) / (ms * 1e-3) / 1e9, 1),
                "sigma_L": listed, "tests": tests, "hits": hits, "corners": corners,
                "share_of_kernel_time": round(stages[0]["ms_per_step"] / sum(s_["ms_per_step"] for s_ in stages), 3),
                "measured": "hipEvent pairs on the worker stream, %d single-worker passes after the timed region "
                            "(%.1f ms per pass)" % (K, single_ms),
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
                "traffic": tj.get(top["kernel"]),
                "avg_launch_ms": round(total_ms / launches, 4),
                "launches_per_step": launches // K,
                "algorithmic_bytes_per_launch": int(top["algorithmic_bytes_per_step"] // max(launches // K, 1)),
                "share_of_kernel_time": round(top["ms_per_step"] / sum(s_["ms_per_step"] for s_ in stages), 3),
                "measured": "hipEvent pairs on the worker stream, %d single-worker passes after the timed region "
                            "(%.1f ms per pass)" % (K, single_ms),
            }
        if stages:
            result["roofline"]["hbm_stages"] = stages
        if tj and "roofline" in result:
            # whole-pipeline HBM rate: measured PMC traffic per launch (profiles/traffic.json) x launches per
            # step of every tracked kernel, over the step time of the timed region
            per_bucket = {"processCorners": 1, "latticeTriangles": 1, "latticeVertices": 1, "latticeMask": 1, "cellCode": 1,
                          "writeEntries": 1, "writeSplatIds": 1, "sortScatter": sort_passes, "sortHist": sort_passes}
            moved = sum(tj[k] * n * len(buckets) for k, n in per_bucket.items() if k in tj)
            result["roofline"]["pipeline_hbm"] = {
                "traffic_bytes_per_step": int(moved), "achieved_GBps": round(moved / (ms_per_step * 1e-3) / 1e9, 1),
                "peak_GBps": HBM_PEAK_GBS, "frac": round(moved / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "PMC traffic of the tracked kernels (over 95 %% of kernel time) / step time with %d device workers; "
                        "the step is bound by processCorners' fp32/LDS work and by latency, not by HBM" % nworkers}
        result["kernel_ms_per_step"] = {k: round(v[0] / K, 3) for k, v in sorted(kernel_stats.items())}
        result["work_per_step"] = {"octree_entries": entries, "occupied_cells": O, "unwelded_vertices": mc["unwelded"],
                                   "welded_vertices": Vw, "external_vertices": external, "indices": T}

    # ---- mesh-sink leg (never `value`): every ship-out of one pass appended to the device mesher (d2d), then
    # finalize = weld by key across buckets + connected components + prune (--fit-prune default 0.02) + compaction ----
    if world == 1 and not args.no_sink:
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
        sink.close()
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import mesher_oracle as mo
            b = buckets[len(buckets) // 2]
            work.copy_from(pristine)
            ctx.synchronize()
            got = workers[0].process(work, b.first, b.count, b.low, b.num_vertices)          # ship-outs copied to the host
            meshes = [dict(chunk=0, vertices=g["vertices"], num_internal=g["num_internal"], keys=g["keys"][g["num_internal"]:],
                           triangles=g["triangles"]) for g in got]
            t0 = time.perf_counter()
            _, ost = mo.mesh_sink(meshes, 0.02)
            cpu_s = time.perf_counter() - t0
            nvs = sum(len(g["vertices"]) for g in got)
            result["mesh_sink"]["cpu_oracle"] = {
                "mvertices_per_s": round(nvs / cpu_s / 1e6, 2), "cores": 1, "kind": "port",
                "sample": "the ship-outs of one of the %d buckets (%d vertices), %.1f s of the numpy/scipy mesh-sink oracle"
                          % (len(buckets), nvs, cpu_s)}

    # ---- device-bucketer leg (never `value`): the RAW cloud resident in HBM, partitioned on the device exactly as
    # the reference's Bucket::bucket would with its defaults (255-cell buckets, 63-cell microblocks, 2 097 152 splats,
    # src/mlsgpu_core.cpp:112-132,655-678), each leaf gathered + transformed on the device and run through a worker ----
    if world == 1 and not args.no_partition:
        from mlsgpu_amd import binding as mb
        raw = m.DeviceBuffer(ctx, array=cloud)
        ext = (0, grid - 1, 0, grid - 1, 0, grid - 1)
        bp = dict(max_splats=args.partition_max_splats, max_cells=255, chunk_cells=0, micro_cells=63, max_split=1 << 30)
        leaves = mb.bucket_cloud(ctx, raw, n_splats, (0.0, 0.0, 0.0), 1.0, ext, on_bucket=lambda leaf, ids: None, **bp)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            mb.bucket_cloud(ctx, raw, n_splats, (0.0, 0.0, 0.0), 1.0, ext, on_bucket=lambda leaf, ids: None, **bp)
        ctx.synchronize()
        part_s = (time.perf_counter() - t0) / args.steps
        pmax = max(l["num_splats"] for l in leaves)
        pcells = max(max(l["extents"][2 * i + 1] - l["extents"][2 * i] for i in range(3)) for l in leaves)
        # the bucketer's callback hands every leaf to the bucket farm's device path (gather + transform kernel into a
        # device item, then the farm's worker threads), as CopyGroup does with host buckets
        pfarm = m.BucketFarm([local_rank], pmax, workers_per_device=nworkers, spare=1, max_cells=pcells,
                             mesh_memory=args.mesh_memory_mb << 20)
        leaf_no = [0]

        def leaf_work(leaf, d_ids):
            low = leaf["extents"][0::2]
            nv = [leaf["extents"][2 * i + 1] - leaf["extents"][2 * i] + 1 for i in range(3)]
            pfarm.submit_device(local_rank, raw, d_ids, leaf["num_splats"], (0.0, 0.0, 0.0), 1.0, ext, low, nv, leaf_no[0])
            leaf_no[0] += 1

        def partition_pass():
            mb.bucket_cloud(ctx, raw, n_splats, (0.0, 0.0, 0.0), 1.0, ext, on_bucket=leaf_work, **bp)
            pfarm.finish()
        partition_pass()                            # warm-up
        t0 = time.perf_counter()
        for _ in range(args.steps):
            partition_pass()
        pipe_s = (time.perf_counter() - t0) / args.steps
        pvox = sum((l["extents"][1] - l["extents"][0]) * (l["extents"][3] - l["extents"][2]) * (l["extents"][5] - l["extents"][4])
                   for l in leaves)
        result["device_partition"] = {
            "buckets": len(leaves), "bucket_splats_total": int(sum(l["num_splats"] for l in leaves)),
            "max_bucket_cells": int(pcells), "bucketing_ms": round(part_s * 1e3, 3),
            "bucketing_msplats_per_s": round(n_splats / part_s / 1e6, 1),
            "pipeline_ms_per_step": round(pipe_s * 1e3, 3), "pipeline_mvoxels_per_s": round(pvox / pipe_s / 1e6, 3),
            "device_workers": nworkers,
            "note": "raw cloud resident in HBM -> mlsgpu_hip_bucket (reference partition) -> mlsgpu_hip_farm_submit_device "
                    "(device gather + transform) -> the farm's device workers; bucketing is inside the pipeline time",
        }
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_binding as ob
            sample = np.ascontiguousarray(cloud[::10])
            t0 = time.perf_counter()
            cpu_leaves = ob.bucket_partition(sample, (0.0, 0.0, 0.0), 1.0, ext, bp["max_splats"] // 10, bp["max_cells"],
                                             bp["chunk_cells"], bp["micro_cells"], bp["max_split"])
            cpu_s = time.perf_counter() - t0
            result["device_partition"]["cpu_oracle"] = {
                "msplats_per_s": round(len(sample) / cpu_s / 1e6, 2), "cores": 1, "kind": "port", "buckets": len(cpu_leaves),
                "sample": "every 10th splat (%d) with a tenth of the bucket capacity, %.1f s of the single-threaded "
                          "bucketing oracle (the reference's bucketing is single-threaded too)" % (len(sample), cpu_s)}
        pfarm.close()
        del raw
    del cloud

    # ---- PCIe-inclusive leg (never `value`): the same buckets from HOST memory through the bucket farm
    # (pinned double-buffered staging, H2D on a copy stream, the same device workers), N = 1 only ----
    if world == 1 and not args.no_stream:
        del workers, work, pristine
        farm_obj = m.BucketFarm([local_rank], max_count, workers_per_device=nworkers, spare=1, max_cells=max_cells,
                                mesh_memory=args.mesh_memory_mb << 20)
        views = [bucketed[b.first:b.first + b.count] for b in buckets]

        def stream_pass():
            for i, (b, v) in enumerate(zip(buckets, views)):
                farm_obj.submit(v, b.low, b.num_vertices, i)
            farm_obj.finish()
        stream_pass()                                   # warm-up
        t0 = time.perf_counter()
        for _ in range(args.steps):
            stream_pass()
        st_s = (time.perf_counter() - t0) / args.steps
        result["pcie_inclusive"] = {
            "value": round(voxels / st_s / 1e6, 3), "unit": "Mvoxels/s", "ms_per_step": round(st_s * 1e3, 3),
            "h2d_GB_per_step": round(bucketed.nbytes / 1e9, 3),
            "h2d_GBps_sustained": round(bucketed.nbytes / st_s / 1e9, 2),
            "note": "host (pageable numpy) -> pinned staging (4 copy threads) -> H2D -> device workers; meshes stay in HBM",
        }
        farm_obj.close()

    # ---- CPU baseline: the oracle ("port") on a bounded sample, rank 0 at N = 1 only ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_binding as ob
        # Sample: buckets of the same cloud, centre outwards, until about 12 s of CPU work (at least one).
        order = sorted(range(len(buckets)), key=lambda i: abs(i - len(buckets) // 2))
        cpu_s, cells, nspl, used = 0.0, 0, 0, 0
        for i in order:
            sample = buckets[i]
            spl = bucketed[sample.first:sample.first + sample.count].copy()
            t0 = time.perf_counter()
            # the reference's own defaults: 24-slice swathes, (maxCells^2 * 2) cells of mesh memory
            ob.bucket(spl, 0, len(spl), sample.num_vertices, sample.low, max_cells=max_cells)
            cpu_s += time.perf_counter() - t0
            cells += sample.cells
            nspl += len(spl)
            used += 1
            if cpu_s > 12.0:
                break
        result["cpu_baseline"] = {
            "value": round(cells / cpu_s / 1e6, 4),
            "unit": "Mvoxels/s",
            "cores": ob.lib().orc_num_threads(),
            "kind": "port",
            "sample": "%d of the %d buckets of the same cloud (centre outwards): %d cells, %d splats, %.1f s of "
                      "the OpenMP oracle on %d threads" % (used, len(buckets), cells, nspl, cpu_s,
                                                          ob.lib().orc_num_threads()),
        }
        result["speedup_vs_cpu"] = round(value / result["cpu_baseline"]["value"], 1)

    if rank == 0:
        print(json.dumps(result))
    pool.shutdown()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
