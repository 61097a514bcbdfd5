#!/usr/bin/env python3
"""One CPU-baseline worker of bench.py: runs the oracle (oracle/, the CPU restatement of the reference's bucket path) on
the buckets listed in a job file and prints one JSON line with its timings.  bench.py starts one of these per host
core group, all at once -- buckets are independent, so the CPU baseline is parallel over buckets exactly as the GPU
farm is -- each pinned to OMP_NUM_THREADS threads.  Never imported by the product; never touches a GPU.

    python tools/cpu_bucket_worker.py job.npz      (job: splats [n] SPLAT_DTYPE, buckets [k, 8] int64 rows
                                                    first, count, low xyz, num_vertices xyz; max_cells)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    import oracle_binding as ob
    job = np.load(sys.argv[1])
    splats = np.ascontiguousarray(job["splats"]).view(ob.SPLAT_DTYPE).reshape(-1)
    max_cells = int(job["max_cells"])
    ob.lib()
    # all workers start computing together: the parent writes the go file once every worker has loaded its job
    go = sys.argv[2] if len(sys.argv) > 2 else None
    if go:
        open(go + ".ready.%d" % os.getpid(), "w").close()
        while not os.path.exists(go):
            time.sleep(0.005)
    out = dict(cells=0, splats=0, buckets=0, seconds=0.0, tree_s=0.0, mls_s=0.0, marching_s=0.0, listed=0, hits=0, triangles=0)
    t_start = time.time()
    for first, count, lx, ly, lz, nx, ny, nz in job["buckets"]:
        t0 = time.perf_counter()
        # the reference's own defaults: 24-slice swathes, (maxCells^2 * 2) cells of mesh memory
        batches, st = ob.bucket(splats, int(first), int(count), (int(nx), int(ny), int(nz)), (int(lx), int(ly), int(lz)),
                                max_cells=max_cells)
        out["seconds"] += time.perf_counter() - t0
        out["cells"] += int((nx - 1) * (ny - 1) * (nz - 1))
        out["splats"] += int(count)
        out["buckets"] += 1
        out["tree_s"] += st["tree_us"] * 1e-6
        out["mls_s"] += st["mls_us"] * 1e-6
        out["marching_s"] += st["marching_us"] * 1e-6
        out["listed"] += st["listed"]
        out["hits"] += st["hits"]
        out["triangles"] += sum(len(b["triangles"]) for b in batches)
    out["t_start"] = t_start
    out["t_end"] = time.time()
    out["threads"] = ob.lib().orc_num_threads()
    t = os.times()
    out["cpu_s"] = t.user + t.system
    print(json.dumps(out))


if __name__ == "__main__":
    main()
