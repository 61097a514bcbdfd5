#!/usr/bin/env python3
"""File -> HBM loader throughput (mlsgpu_hip_fileset_load): writes `files` PLY files of `millions` M splats in total to a
directory (default /dev/shm: the page cache, so the decode and the PCIe copy are what is measured, not a disk) and loads
them with several reader-thread counts.

    python tools/loader_bench.py [millions=64] [files=4] [dir=/dev/shm]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402


def main():
    import torch  # noqa: F401  (before the HIP library, tests/conftest.py)
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as mb
    millions = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    nfiles = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    directory = sys.argv[3] if len(sys.argv) > 3 else "/dev/shm"
    per = millions * 1_000_000 // nfiles
    rng = np.random.default_rng(1)
    rows = np.zeros(per, np.dtype([("p", "<f4", 3), ("n", "<f4", 3), ("r", "<f4")]))
    rows["p"] = rng.uniform(0, 1000, (per, 3)).astype(np.float32)
    rows["n"] = rng.normal(size=(per, 3)).astype(np.float32)
    rows["r"] = rng.uniform(1, 3, per).astype(np.float32)
    head = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % per \
        + "".join("property float32 %s\n" % n for n in ("x", "y", "z", "nx", "ny", "nz", "radius")) + "end_header\n"
    paths = []
    for k in range(nfiles):
        p = os.path.join(directory, "mlsgpu_loader_bench_%d.ply" % k)
        with open(p, "wb") as f:
            f.write(head.encode("ascii"))
            f.write(rows.tobytes())
        paths.append(p)
    try:
        ctx = m.Context(0)
        n = per * nfiles
        dev = m.DeviceBuffer(ctx, nbytes=n * 32)
        for threads, buf in ((4, 128 << 20), (8, 512 << 20), (12, 768 << 20), (16, 512 << 20), (32, 1 << 30)):
            fs = mb.FileSet(paths, buffer_size=buf)
            fs.load(ctx, dev, count=min(n, 4_000_000), reader_threads=threads)      # warm
            t0 = time.perf_counter()
            fs.load(ctx, dev, reader_threads=threads)
            dt = time.perf_counter() - t0
            print("reader threads %2d, pinned buffer %4d MiB: %.2f GB/s of splats into HBM (%.2f GB/s of file data), %d M splats in %.2f s"
                  % (threads, buf >> 20, n * 32 / dt / 1e9, n * 28 / dt / 1e9, n // 1_000_000, dt))
            fs.close()
        check = dev.download(m.SPLAT_DTYPE, 1000, offset=(n - 1000) * 32)
        assert np.array_equal(check["position"], rows["p"][-1000:])
    finally:
        for p in paths:
            os.remove(p)


if __name__ == "__main__":
    main()
