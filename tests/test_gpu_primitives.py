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
