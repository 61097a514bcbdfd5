"""Scan and radix sort of the HIP path against numpy (clogs::Scan / clogs::Radixsort semantics)."""
import ctypes as C

import numpy as np
import pytest

from gpu_common import ctx  # noqa: F401

pytestmark = pytest.mark.gpu


# one launch up to 1024 tiles of 2048 elements, two up to 4096 tiles, three beyond
@pytest.mark.parametrize("n,seed", [(0, 5), (1, 0), (63, 1), (64, 7), (4096, 0), (4097, 3), (1_000_003, 1), (2_097_152, 2),
                                    (2_097_153, 9), (5_000_000, 4), (9_000_001, 6)])
def test_scan_u32(ctx, n, seed):
    import mlsgpu_amd as m
    rng = np.random.RandomState(n + 1)
    data = rng.randint(0, 4, size=n).astype(np.uint32)
    buf = m.DeviceBuffer(ctx, array=data if n else np.zeros(1, np.uint32))
    m.binding.check(m.lib().mlsgpu_hip_test_scan_u32(ctx.h, buf.ptr, n, seed))
    got = buf.download(np.uint32, n)
    exp = (np.concatenate([[0], np.cumsum(data[:-1], dtype=np.uint64)]) + seed).astype(np.uint32) if n else data
    np.testing.assert_array_equal(got, exp)


def test_scan_launches_back_to_back(ctx):
    """The one-launch scan's flags are never cleared (a launch's flags carry its epoch): scans of changing sizes, one
    behind the other on one stream, each see only their own."""
    import mlsgpu_amd as m
    rng = np.random.RandomState(77)
    sizes = [int(x) for x in rng.randint(1, 600_000, size=40)] + [2_097_152, 5, 2_000_000, 2049, 2048, 1]
    bufs, datas = [], []
    for n in sizes:
        data = rng.randint(0, 5, size=n).astype(np.uint32)
        datas.append(data)
        bufs.append(m.DeviceBuffer(ctx, array=data))
    for rep in range(3):
        for k, (n, buf) in enumerate(zip(sizes, bufs)):
            if rep:
                buf.upload(datas[k])
            m.binding.check(m.lib().mlsgpu_hip_test_scan_u32(ctx.h, buf.ptr, n, k))
        for k, (n, buf) in enumerate(zip(sizes, bufs)):
            exp = (np.concatenate([[0], np.cumsum(datas[k][:-1], dtype=np.uint64)]) + k).astype(np.uint32)
            np.testing.assert_array_equal(buf.download(np.uint32, n), exp)


def _scan_batch(m, ctx, ins, outs, sizes, seeds, repeats):
    k = len(ins)
    pin = (C.c_void_p * k)(*[b.ptr for b in ins])
    pout = (C.c_void_p * k)(*[b.ptr for b in outs])
    pn = (C.c_uint64 * k)(*sizes)
    ps = (C.c_uint32 * k)(*seeds)
    m.binding.check(m.lib().mlsgpu_hip_test_scan_u32_batch(ctx.h, pin, pout, pn, ps, k, repeats))


def test_scan_batch_lanes(ctx):
    """A batch's scans: one set of launches for up to eight lanes of different lengths (an empty lane among them)."""
    import mlsgpu_amd as m
    rng = np.random.RandomState(3)
    sizes = [2_097_152, 1, 0, 777_777, 2048, 2049, 1_500_000, 63]
    datas = [rng.randint(0, 4, size=max(n, 1)).astype(np.uint32) for n in sizes]
    ins = [m.DeviceBuffer(ctx, array=d) for d in datas]
    outs = [m.DeviceBuffer(ctx, array=np.full(max(n, 1), 0xDEAD, np.uint32)) for n in sizes]
    seeds = list(range(10, 18))
    _scan_batch(m, ctx, ins, outs, sizes, seeds, 3)
    for n, d, o, sd in zip(sizes, datas, outs, seeds):
        if n:
            exp = (np.concatenate([[0], np.cumsum(d[:n - 1], dtype=np.uint64)]) + sd).astype(np.uint32)
            np.testing.assert_array_equal(o.download(np.uint32, n), exp)


def test_scan_one_launch_under_contention(ctx):
    """The shape bench.py runs: eight lanes x 1024 tiles per launch (four times the workgroups the chip holds), launches
    back to back, while a second context's worker keeps the CUs busy with processCorners on a cfg2-sized bucket.  A
    workgroup's tile is a ticket it draws when it starts, so nothing it waits for can be un-dispatched (VERDICT round 5:
    the scan used blockIdx.x, which is only safe if workgroups are dispatched in order).  A watchdog ends the process with
    its own exit code if a scan call does not return in 10 s -- a hung stream cannot be recovered from inside."""
    import os
    import threading
    import time
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    lanes, n = 8, 1024 * 2048
    rng = np.random.RandomState(11)
    datas = [rng.randint(0, 3, size=n).astype(np.uint32) for _ in range(lanes)]
    ins = [m.DeviceBuffer(ctx, array=d) for d in datas]
    outs = [m.DeviceBuffer(ctx, array=np.zeros(n, np.uint32)) for _ in range(lanes)]
    seeds = [7 * k for k in range(lanes)]

    cloud, g = synth.make_cloud("cfg2", scale=0.4)
    ctx2 = m.Context(0)
    w = m.Worker(ctx2, len(cloud), max_cells=255)
    w.set_keep_splats(True)
    buf = m.DeviceBuffer(ctx2, array=cloud)
    stop = threading.Event()
    passes = [0]
    errors = []

    def hog():
        try:
            col = m.binding.ChecksumCollector(ctx2)
            while not stop.is_set():
                w.process(buf, 0, len(cloud), (0, 0, 0), (g, g, g), collector=col)
                passes[0] += 1
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    deadline = [None]

    def watchdog():
        while not stop.is_set():
            d = deadline[0]
            if d is not None and time.time() > d:
                os.write(2, b"test_scan_one_launch_under_contention: a scan did not return in 10 s (hung stream)\n")
                os._exit(97)
            time.sleep(0.05)

    th = threading.Thread(target=hog)
    wd = threading.Thread(target=watchdog, daemon=True)
    th.start()
    wd.start()
    try:
        t0 = time.time()
        calls = 0
        while calls < 30 or (passes[0] < 2 and time.time() - t0 < 60):
            deadline[0] = time.time() + 10.0
            _scan_batch(m, ctx, ins, outs, [n] * lanes, seeds, 20)
            deadline[0] = None
            calls += 1
    finally:
        stop.set()
        th.join()
    assert not errors, errors
    assert passes[0] >= 1                                     # the other context really ran beside the scans
    for d, o, sd in zip(datas, outs, seeds):
        exp = (np.concatenate([[0], np.cumsum(d[:-1], dtype=np.uint64)]) + sd).astype(np.uint32)
        np.testing.assert_array_equal(o.download(np.uint32, n), exp)
    del w, buf


@pytest.mark.parametrize("n,bits", [(1, 17), (1000, 1), (4096, 8), (5000, 10), (70_001, 17), (300_000, 28), (300_000, 32)])
def test_sort_u32_stable(ctx, n, bits):
    import mlsgpu_amd as m
    rng = np.random.RandomState(bits * 7 + n % 97)
    keys = rng.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    if bits == 17:
        keys[rng.rand(n) < 0.3] = 0xFFFFFFFF       # the octree's "no entry" keys must sort last
        keys[keys != 0xFFFFFFFF] &= 0x7FFF
    vals = np.arange(n, dtype=np.uint32)
    kb = m.DeviceBuffer(ctx, array=keys)
    vb = m.DeviceBuffer(ctx, array=vals)
    m.binding.check(m.lib().mlsgpu_hip_test_sort_u32(ctx.h, kb.ptr, vb.ptr, n, bits))
    mask = np.uint32((1 << bits) - 1) if bits < 32 else np.uint32(0xFFFFFFFF)
    order = np.argsort(keys & mask, kind="stable")
    np.testing.assert_array_equal(kb.download(np.uint32, n), keys[order])
    np.testing.assert_array_equal(vb.download(np.uint32, n), vals[order])


@pytest.mark.parametrize("n,bits", [(10, 43), (123_457, 43), (50_000, 64)])
def test_sort_u64_stable(ctx, n, bits):
    import mlsgpu_amd as m
    rng = np.random.RandomState(bits + n % 11)
    keys = rng.randint(0, 2 ** 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.randint(0, 2, size=n).astype(np.uint64)
    keys[: n // 3] = keys[n // 3: 2 * (n // 3)]      # duplicates exercise stability
    vals = np.arange(n, dtype=np.uint32)
    kb = m.DeviceBuffer(ctx, array=keys)
    vb = m.DeviceBuffer(ctx, array=vals)
    m.binding.check(m.lib().mlsgpu_hip_test_sort_u64(ctx.h, kb.ptr, vb.ptr, n, bits))
    mask = np.uint64((1 << bits) - 1) if bits < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    order = np.argsort(keys & mask, kind="stable")
    np.testing.assert_array_equal(kb.download(np.uint64, n), keys[order])
    np.testing.assert_array_equal(vb.download(np.uint32, n), vals[order])
