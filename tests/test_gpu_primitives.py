"""Scan and radix sort of the HIP path against numpy (clogs::Scan / clogs::Radixsort semantics)."""
import ctypes as C

import numpy as np
import pytest

from gpu_common import ctx  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,seed", [(0, 5), (1, 0), (63, 1), (64, 7), (4096, 0), (4097, 3), (1_000_003, 1)])
def test_scan_u32(ctx, n, seed):
    import mlsgpu_amd as m
    rng = np.random.RandomState(n + 1)
    data = rng.randint(0, 4, size=n).astype(np.uint32)
    buf = m.DeviceBuffer(ctx, array=data if n else np.zeros(1, np.uint32))
    m.binding.check(m.lib().mlsgpu_hip_test_scan_u32(ctx.h, buf.ptr, n, seed))
    got = buf.download(np.uint32, n)
    exp = (np.concatenate([[0], np.cumsum(data[:-1], dtype=np.uint64)]) + seed).astype(np.uint32) if n else data
    np.testing.assert_array_equal(got, exp)


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
