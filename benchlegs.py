"""The secondary legs of bench.py (never `value`): what the headline's device pipeline delivers once transfers, the mesh
sink, the reference partition, the other GPUs and the host CPU come into the picture.  bench.py imports everything here;
tools/transfer_probe.py and tools/sink_route_probe.py call single legs.

cfg5            BASELINE configs[4]: 10^9 splats from PLY files -> HBM -> device bucketer -> farm (run_cfg5)
transfer_legs   SURVEY 8(d)'s region: host splats in -> last mesh byte out, through the ring and through the device sink
host_weld_leg   the same with the host welder consuming the ring ("welding stays on host")
multi_gpu_legs  N > 1: the region on every rank with the cross-rank weld; the reference's one-process farm over all GPUs
cpu_baseline    the oracle ("port") on the host cores -- the only use of oracle/ outside tests/
"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))


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
    fs = mb.FileSet(paths, buffer_size=768 << 20)
    assert len(fs) == n
    raw = m.DeviceBuffer(ctx, nbytes=n * 32)
    ext = (0, g - 1, 0, g - 1, 0, g - 1)
    ref0 = (0.0, 0.0, 0.0)

    def load():
        fs.load(ctx, raw, reader_threads=args.reader_threads)
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
    bfarm = m.BucketFarm([local_rank], pmax, workers_per_device=nworkers, spare=max(1, args.cfg5_spare), max_cells=pcells,
                         mesh_memory=args.mesh_memory_mb << 20, collect="checksum")
    if args.cfg5_batch > 1:
        bfarm.set_batch(args.cfg5_batch)

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
            sg = mb.bounding_grid_files(ctx, fs, 1.0, 63, chunk, reader_threads=args.reader_threads)
            sbound_s = time.perf_counter() - t0
            t0 = time.perf_counter()
            sleaves, sstats = mb.bucket_cloud_stream(ctx, fs, ref0, 1.0, ext, budget_splats=budget, chunk_splats=chunk,
                                                     reader_threads=args.reader_threads, on_bucket=stream_leaf, **CFG5_PARTITION)
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
            "buckets_per_launch_set": max(1, args.cfg5_batch), "device_items": nworkers + max(1, args.cfg5_spare),
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
            "reader_threads": args.reader_threads,
            "note": "PLY files in %s (page cache) -> reader threads pread whole rows into a 768 MiB pinned buffer -> H2D of the rows (28 B a splat) -> decoded by a kernel -> the timed "
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


# ------------------------------------------------------------------------------- the reference's dispatch

def test_devices(n):
    """The GPUs of a one-process job of `n` device groups: 0 .. n - 1, or MLSGPU_TEST_DEVICES=0,0,... (a one-GPU box standing
    in for n GPUs: a check of the code path, not an n-GPU measurement)."""
    env = os.environ.get("MLSGPU_TEST_DEVICES")
    if env:
        devs = [int(x) for x in env.split(",")]
        if len(devs) != n:
            raise SystemExit("MLSGPU_TEST_DEVICES names %d devices, --gpus %d" % (len(devs), n))
        return devs, True
    return list(range(n)), False


def cfg4_bucket_pins(dist_name):
    path = os.path.join(ROOT, "tests", "golden", "cfg4_buckets_%s.json" % dist_name)
    try:
        return json.load(open(path))
    except Exception:   # noqa: BLE001
        return None


def run_greedy(args):
    """`bench.py --gpus N --dispatch greedy`: BASELINE configs[3] the way the reference runs it -- ONE process, a device
    worker group per GPU behind one copy side per socket, every bucket of the WHOLE cfg4 cloud (125 buckets of <= 255 cells
    per side) pushed in partition order to the group with the most unallocated capacity that can take an item
    (src/workers.cpp:320-351, src/mlsgpu_core.cpp:704-741).  Which GPU a bucket lands on depends on timing, so the check is
    per bucket and order-independent: every bucket's ship-outs digest against its pin (tests/golden/cfg4_buckets_<dist>.json,
    written next to oracle parity on sampled buckets by tests/test_gpu_configs.py::test_cfg4_bucket_pins)."""
    import torch
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
        raise SystemExit("--dispatch greedy is ONE process over N GPUs: run `python bench.py --gpus N --dispatch greedy` itself")
    n = args.gpus
    devs, shared = test_devices(n)
    if max(devs) >= torch.cuda.device_count():
        raise SystemExit("--gpus %d --dispatch greedy: device %d is not there (%d visible; MLSGPU_TEST_DEVICES=0,0,... runs the "
                         "code path on fewer)" % (n, max(devs), torch.cuda.device_count()))
    torch.cuda.set_device(devs[0])
    device = torch.device("cuda", devs[0])
    cloud, g = synth.make_cloud_device("cfg4", device, scale=args.scale, dist=args.dist)
    n_splats = len(cloud)
    bucketed, buckets = synth.bucketize_device(cloud, synth.grid_buckets((g, g, g), 255))
    del cloud
    max_count = max(b.count for b in buckets)
    max_cells = max(max(b.num_vertices) for b in buckets) - 1
    voxels = sum(b.cells for b in buckets)
    host = synth.to_host_splats(bucketed)
    views = [host[b.first:b.first + b.count] for b in buckets]
    farm = m.BucketFarm(devs, max_count, workers_per_device=args.farm_workers, spare=args.farm_spare, collect="checksum",
                        max_cells=max_cells, mesh_memory=args.mesh_memory_mb << 20, copy_threads=args.copy_threads,
                        staging_buffers=args.staging_buffers)
    if args.farm_batch > 1:
        farm.set_batch(args.farm_batch)

    def host_fed():
        for i, (b, v) in enumerate(zip(buckets, views)):
            farm.submit(v, b.low, b.num_vertices, i)
        farm.finish()

    # ---- the checked pass: every bucket against its pin ----
    host_fed()
    got = {i: m.binding.digest_of_sums(farm.sums.get(i, [])) for i in range(len(buckets))}
    totals = dict(triangles=sum(r[1] for v in farm.sums.values() for r in v), vertices=sum(r[0] for v in farm.sums.values() for r in v))
    pins = cfg4_bucket_pins(args.dist) if args.scale == 1.0 else None
    check = {"buckets": len(buckets), "pinned": pins is not None}
    if pins is not None:
        bad = [i for i in range(len(buckets)) if got[i] != pins["buckets"][i]["digest"]]
        check.update(ok=not bad, mismatches=bad[:8], totals_ok=totals == {k: pins["total"][k] for k in totals})
        if bad or not check["totals_ok"]:
            raise SystemExit("greedy dispatch: %d of %d buckets differ from tests/golden/cfg4_buckets_%s.json (first: %s)"
                             % (len(bad), len(buckets), args.dist, bad[:8]))
    farm.checksums = False
    for _ in range(args.warmup):
        host_fed()
    s0, c0, w0 = farm.stats(), farm.copy_clock(), farm.worker_clock()
    g0 = [farm.group_clock(k) for k in range(n)]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        host_fed()
    elapsed = time.perf_counter() - t0
    s1, c1, w1 = farm.stats(), farm.copy_clock(), farm.worker_clock()
    g1 = [farm.group_clock(k) for k in range(n)]
    per_dev = []
    for k in range(n):
        d = {key: g1[k][key] - g0[k][key] for key in g1[k]}
        workers = max(args.farm_workers, 1)
        per_dev.append({"device": devs[k], "buckets": d["buckets"], "launch_sets": d["launch_sets"],
                        "busy_s_per_worker": round(d["busy_s"] / workers, 4), "idle_s_per_worker": round(d["idle_s"] / workers, 4),
                        "idle_frac": round(d["idle_s"] / max(d["idle_s"] + d["busy_s"], 1e-12), 4)})
    value = voxels * args.steps / elapsed / 1e6

    # ---- the same buckets from the cloud resident on the first GPU: device gathers, peer copies to the other GPUs' items ----
    device_fed = None
    try:
        ctx = m.Context(devs[0])
        raw = m.DeviceBuffer(ctx, nbytes=bucketed.numel() * 4, borrow=bucketed.data_ptr())
        iota = m.DeviceBuffer(ctx, array=np.arange(max_count, dtype=np.uint32))
        ext = (0, g - 1, 0, g - 1, 0, g - 1)

        class _Sub:
            def __init__(self, ptr):
                self.ptr = ptr

        def dev_fed():
            for i, b in enumerate(buckets):
                farm.submit_device(devs[0], _Sub(raw.ptr + 32 * b.first), iota.ptr, b.count, (0.0, 0.0, 0.0), 1.0, ext, b.low,
                                   b.num_vertices, i)
            farm.finish()
        farm.checksums = True
        farm.sums.clear()
        dev_fed()
        dgot = {i: m.binding.digest_of_sums(farm.sums.get(i, [])) for i in range(len(buckets))}
        if dgot != got:
            raise SystemExit("greedy dispatch: the device-fed pass produced other meshes than the host-fed one")
        farm.checksums = False
        dev_fed()
        ds = max(2, min(args.steps, 10))
        t1 = time.perf_counter()
        for _ in range(ds):
            dev_fed()
        d_el = (time.perf_counter() - t1) / ds
        device_fed = {"value": round(voxels / d_el / 1e6, 3), "unit": "Mvoxels/s", "ms_per_pass": round(d_el * 1e3, 2), "passes": ds,
                      "digests": "equal to the host-fed pass, bucket by bucket"}
        del raw, iota
    except SystemExit:
        raise
    except Exception as e:      # noqa: BLE001 - a secondary figure
        device_fed = {"error": "%s: %s" % (type(e).__name__, e)}
    placement = farm.placement() if hasattr(farm, "placement") else None
    farm.close()
    result = {
        "metric": "Mvoxels/s evaluated+triangulated (and Msplats/s in)", "value": round(value, 3), "unit": "Mvoxels/s", "n_gpus": n,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "msplats_per_s": round(n_splats * args.steps / elapsed / 1e6, 3),
        "input_protocol": "host_fed_greedy: every pass hands the buckets' splats over from host memory (copy threads -> pinned "
                          "staging -> H2D to the chosen GPU), as the reference's loader does; meshes are counted and checksummed "
                          "on the device, not read back",
        "config": {"workload": "cfg4 (BASELINE configs[3]) WHOLE: %d^3 grid, %d splats (%s), %d buckets of <= %d cells per side, "
                               "one process, greedy dispatch over %d device group(s)" % (g, n_splats, args.dist, len(buckets), max_cells, n),
                   "dispatch": "greedy (src/workers.cpp:320-351): the group with the most unallocated splat capacity that can take an item",
                   "devices": devs, "workers_per_device": args.farm_workers, "spare_items": args.farm_spare, "farm_batch": args.farm_batch,
                   "parallelism": "%d device group(s), one process" % n},
        "output_digest": check,
        "per_device": per_dev,
        "copy_side": {"h2d_GBps": round((s1["h2d_bytes"] - s0["h2d_bytes"]) / elapsed / 1e9, 2),
                      "h2d_stream_s_over_span": round((c1["h2d_s"] - c0["h2d_s"]) / max(c1["span_s"] - c0["span_s"], 1e-12), 4),   # summed over the GPUs' copy streams: up to N
                      "wait_item_s": round(c1["wait_item_s"] - c0["wait_item_s"], 4),
                      "wait_staging_s": round(c1["wait_staging_s"] - c0["wait_staging_s"], 4)},
        "buckets_per_launch_set": round((w1["buckets"] - w0["buckets"]) / max(w1["launch_sets"] - w0["launch_sets"], 1), 2),
        "in_flight_max": s1.get("in_flight_max"),
        "device_fed": device_fed,
        "placement": placement,
    }
    if shared:
        result["debug_shared_gpu"] = ("%d device groups on GPU(s) %s (MLSGPU_TEST_DEVICES): a check of the dispatch, NOT an "
                                      "N-GPU measurement" % (n, sorted(set(devs))))
    print(json.dumps(result))


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


def host_weld_leg(m, args, device_index, bucketed_host, buckets, max_count, max_cells, voxels, steps=3, landing=True):
    """The reference's complete route, welder included: host splats -> farm -> every ship-out read back through the pinned ring
    -> the mesher thread hands it to the host welder (OOCMesher's weld: local components, key map, union-find;
    src/mesher.cpp:220-311 -- a task per block on the welder's pool of threads, where the reference has one thread and an
    OpenMP rewrite, src/mesher.cpp:597-600) -> finalize (components, prune, one mesh per chunk).  One job = one fresh welder;
    a warm-up job first (the welder's memory comes from a cache of mapped slabs).  Two figures: a job ALONE (its latency: pass,
    then finalize), and a STREAM of jobs in which job k's finalize runs on its own thread while job k + 1's buckets are already
    going through the farm into the next welder -- the steady state `value` is quoted on.
    landing=True (round 6): the read-backs land in page-locked memory of the WELDER and are adopted there
    (mlsgpu_hip_farm_set_host_landing); landing=False: through the farm's pinned ring, copied out by the welder (rounds 4-5)."""
    import threading
    nworkers = max(1, min(args.farm_workers, len(buckets)))
    farm = m.BucketFarm([device_index], max_count, workers_per_device=nworkers, spare=args.farm_spare, max_cells=max_cells,
                        mesh_memory=args.mesh_memory_mb << 20, copy_threads=args.copy_threads, staging_buffers=args.staging_buffers)
    views = [bucketed_host[b.first:b.first + b.count] for b in buckets]
    last = {}

    def stream_in(welder):
        if landing:
            farm.set_host_landing(welder)
        else:
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
            "route": ("landing: read-backs land in the welder's own page-locked memory, adopted without a copy" if landing
                      else "ring: read-backs land in the farm's pinned ring, the welder copies every block out"),
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
