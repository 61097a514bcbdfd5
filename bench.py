#!/usr/bin/env python3
"""Headline benchmark: Mvoxels/s evaluated + triangulated by the per-bucket device pipeline.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

One "step" is one pass of the hot path (octree build -> MLS corner evaluation -> marching tetrahedra
with welding -> scale/bias) over every bucket of one synthetic splat cloud whose splats are already
resident in HBM.  At N = 1 the workload is BASELINE.json configs[2] (512^3 grid, 50 M uniform-random
splats, multi-bucket stream), the configuration the north_star target is quoted on; with N > 1 every
rank streams its own cloud of the same shape (weak scaling, no data-path collective: buckets are
independent, cross-bucket welding is host work in the reference).

Rank 0 prints ONE JSON line (see README / DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

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
    p.add_argument("--variant", type=int, default=0, help="MLS kernel variant: 0 culled, 1 basic")
    p.add_argument("--no-cpu-baseline", action="store_true")
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
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import mlsgpu_amd as m
    from mlsgpu_amd import synth

    # ---- workload (host side, untimed) ----
    t0 = time.time()
    cloud, grid = synth.make_cloud(args.workload, args.dist, scale=args.scale, seed_offset=rank)
    bucketed, buckets = synth.bucketize(cloud, grid, 255)
    n_splats = len(cloud)
    del cloud
    voxels = sum(b.cells for b in buckets)
    max_count = max(b.count for b in buckets)
    max_cells = max(max(b.num_vertices) for b in buckets) - 1
    setup_s = time.time() - t0

    ctx = m.Context(local_rank)
    pristine = m.DeviceBuffer(ctx, array=bucketed)
    work = m.DeviceBuffer(ctx, nbytes=bucketed.nbytes)
    worker = m.Worker(ctx, max_count, max_cells=max_cells, mesh_memory=args.mesh_memory_mb << 20)
    worker.set_mls_variant(args.variant)
    sizes_box = [m.binding.SizeCollector()]

    def step():
        # The octree build overwrites splat.w with 1/r^2 (kernels/octree.cl:193), so each pass starts from a
        # fresh copy of the resident splats: a device-to-device copy standing where the reference has its
        # host-to-device copy (src/workers.cpp:356-361).  It is inside the timed region.
        work.copy_from(pristine)
        for b in buckets:
            worker.process(work, b.first, b.count, b.low, b.num_vertices, collector=sizes_box[0])

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.reset_stats()
    ctx.set_timing(not args.no_timing)
    sizes_box[0] = m.binding.SizeCollector()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ctx.set_timing(False)
    kernel_stats = ctx.stats()

    # ---- algorithmic work of the dominant kernel (one instrumented, untimed pass) ----
    counters = m.DeviceBuffer(ctx, array=np.zeros(3, np.uint64))
    worker.set_mls_stats(counters)
    work.copy_from(pristine)
    corners = 0
    for b in buckets:
        worker.process(work, b.first, b.count, b.low, b.num_vertices, collector=m.binding.SizeCollector())
        corners += int(np.prod([-(-n // 8) * 8 for n in b.num_vertices]))
    ctx.synchronize()
    listed, tests, hits = (int(x) for x in counters.download(np.uint64))
    worker.set_mls_stats(None)
    sizes = sizes_box[0]

    ms_per_step = elapsed / args.steps * 1e3
    value = world * voxels * args.steps / elapsed / 1e6
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
            "mls_variant": "culled" if args.variant == 0 else "basic",
            "per_rank": "own cloud per rank (seed offset = rank)",
            "triangles_per_step": sizes.triangles // max(args.steps, 1),
            "vertices_per_step": sizes.vertices // max(args.steps, 1),
            "shipouts_per_step": sizes.batches // max(args.steps, 1),
            "host_setup_s": round(setup_s, 1),
        },
    }

    # ---- roofline of the dominant kernel, processCorners ----
    name = "kernel.mls.processCorners.time"
    if name in kernel_stats and kernel_stats[name][1] > 0:
        total_ms, launches = kernel_stats[name]
        per_step_ms = total_ms / args.steps
        # MLS bytes = 36*SigmaL + 4*V, flops = 10*512*SigmaL + 25*H  (per step, all buckets)
        alg_bytes = 36 * listed + 4 * corners
        alg_flops = 10 * 512 * listed + 25 * hits
        done_flops = 10 * tests + 25 * hits          # distance tests actually executed after sub-block culling
        achieved = alg_bytes / (per_step_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("%s/%s" % (args.workload, args.dist), {}).get("processCorners_bytes_per_launch")
            except Exception:
                traffic = None
        result["roofline"] = {
            "kernel": "processCorners",
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic,
            "launches_per_step": launches // args.steps,
            "avg_launch_ms": round(total_ms / launches, 4),
            "share_of_step": round(per_step_ms / ms_per_step, 3),
            "algorithmic_bytes_per_step": alg_bytes,
            "note": "processCorners is fp32-VALU/LDS bound, not HBM bound (SURVEY 8d); see valu",
            "valu": {
                "unit": "TFLOP/s",
                "peak": FP32_VALU_PEAK_TFLOPS,
                "reference_algorithm": round(alg_flops / (per_step_ms * 1e-3) / 1e12, 3),
                "executed": round(done_flops / (per_step_ms * 1e-3) / 1e12, 3),
                "frac_executed": round(done_flops / (per_step_ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS, 4),
                "sigma_L": listed, "tests": tests, "hits": hits, "corners": corners,
            },
        }
        result["kernel_ms_per_step"] = {k: round(v[0] / args.steps, 3) for k, v in sorted(kernel_stats.items())}

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
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
