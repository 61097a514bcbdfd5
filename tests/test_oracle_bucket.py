"""Pins the bucketing oracle (oracle/bucket_oracle.cpp) with the reference's own tests for Bucket::bucket:
test/test_bucket.cpp (Node, forEachNode, TestBucket incl. the `validate` properties) and the splatToBuckets vectors of
test/test_splat_set.cpp.  The numbers are the reference's."""
import numpy as np
import pytest

import oracle_binding as ob
from bucket_checks import create_splats, make_splat, random_case, validate_partition


def test_node_child():
    # test/test_bucket.cpp:119-135
    parent = (1, 2, 3, 4)
    exp = [(2, 4, 6, 3), (3, 4, 6, 3), (2, 5, 6, 3), (3, 5, 6, 3), (2, 4, 7, 3), (3, 4, 7, 3), (2, 5, 7, 3), (3, 5, 7, 3)]
    assert [ob.node_child(parent, i) for i in range(8)] == exp


def test_for_each_node_order():
    # test/test_bucket.cpp:203-238: dims (4,4,6), 4 levels, recurse into nodes containing microblock (2,1,4)
    nodes = ob.for_each_node((4, 4, 6), 4, (2, 1, 4))
    assert nodes == [(0, 0, 0, 3), (0, 0, 0, 2), (0, 0, 1, 2), (0, 0, 2, 1), (1, 0, 2, 1), (2, 0, 4, 0), (3, 0, 4, 0),
                     (2, 1, 4, 0), (3, 1, 4, 0), (2, 0, 5, 0), (3, 0, 5, 0), (2, 1, 5, 0), (3, 1, 5, 0), (0, 1, 2, 1),
                     (1, 1, 2, 1)]


def test_splat_to_buckets():
    # test/test_splat_set.cpp:132-156
    ref, spacing, ext = (10.0, -50.0, 40.0), 20.0, (-1, 5, 1, 100, 2, 50)
    lo, hi = ob.splat_to_buckets(make_splat(115.0, -31.0, 1090.0, 7.0), ref, spacing, ext, 3)
    assert list(lo) == [1, -1, 16] and list(hi) == [2, 0, 16]
    lo, hi = ob.splat_to_buckets(make_splat(-1000.0, -1000.0, -1000.0, 100.0), ref, spacing, ext, 3)
    assert list(lo) == [-19, -18, -20] and list(hi) == [-15, -15, -17]


def test_splat_to_buckets_class_vectors():
    # test/test_splat_set.cpp:182-215 (SplatToBuckets functor = a grid with reference 0 and extents from 0)
    zero = ((0.0, 0.0, 0.0), (0, 1, 0, 1, 0, 1))
    lo, hi = ob.splat_to_buckets(make_splat(-9.0, 100.0, -125.0, 5.0), zero[0], 4.0, zero[1], 10)
    assert list(lo) == [-1, 2, -4] and list(hi) == [-1, 2, -3]
    lo, hi = ob.splat_to_buckets(make_splat(-1.0, -13.0, -12.0, 32.0), zero[0], 8.0, zero[1], 1)
    assert list(lo) == [-5, -6, -6] and list(hi) == [3, 2, 2]


GRID = dict(reference=(-10.0, 0.0, 10.0), spacing=2.5, extents=(4, 20, 0, 20, -4, 4))


def run(splats, max_splats, max_cells, chunk_cells, micro_cells, max_split, grid=GRID):
    return ob.bucket_partition(splats, grid["reference"], grid["spacing"], grid["extents"], max_splats, max_cells,
                               chunk_cells, micro_cells, max_split)


def test_simple_gives_11_buckets():
    # test/test_bucket.cpp:459-476
    splats = create_splats()
    leaves = run(splats, 5, 8, 0, 8, 1000000)
    validate_partition(splats, GRID, leaves, 5, 8, 0)
    assert len(leaves) == 11


def test_density_error():
    # test/test_bucket.cpp:478-492
    with pytest.raises(ob.DensityError):
        run(create_splats(), 1, 8, 0, 8, 1000000)


def test_flat_is_one_bucket():
    # test/test_bucket.cpp:494-509
    splats = create_splats()
    leaves = run(splats, 15, 32, 0, 32, 1000000)
    validate_partition(splats, GRID, leaves, 15, 32, 0)
    assert len(leaves) == 1


def test_empty():
    # test/test_bucket.cpp:511-529
    assert run(np.zeros(0, ob.SPLAT_DTYPE), 5, 8, 0, 8, 1000000) == []


def test_multi_level_gives_11_buckets():
    # test/test_bucket.cpp:531-547: maxSplit = 8 forces several levels of recursion
    splats = create_splats()
    leaves = run(splats, 5, 8, 0, 8, 8)
    validate_partition(splats, GRID, leaves, 5, 8, 0)
    assert len(leaves) == 11
    assert max(l["depth"] for l in leaves) > 1


def test_chunk_cells():
    # test/test_bucket.cpp:549-565: chunkCells 14 is rounded up to 16
    splats = create_splats()
    leaves = run(splats, 20, 2 ** 31 - 1, 14, 8, 1000000)
    validate_partition(splats, GRID, leaves, 20, 2 ** 31 - 1, 16)
    assert len({l["chunk"] for l in leaves}) > 1


def test_non_finite_splats_are_skipped():
    splats = create_splats()
    bad = splats.copy()
    bad["position"][3, 1] = np.nan
    bad["radius"][7] = np.inf
    leaves = run(bad, 5, 8, 0, 8, 1000000)
    ids = np.concatenate([l["ids"] for l in leaves])
    assert 3 not in ids and 7 not in ids
    validate_partition(bad, GRID, leaves, 5, 8, 0)


@pytest.mark.parametrize("seed", range(30))
def test_random(seed):
    # test/test_bucket.cpp:591-664 (same parameter distributions; the reference validates properties only)
    rng = np.random.default_rng(seed)
    num_scans = int(rng.integers(0, 21))
    max_scan = int(rng.integers(1, 2001))
    max_split = int(rng.integers(64, 1001))
    max_cells = int(rng.integers(40, 101))
    chunk_cells = int(rng.integers(80, 514))
    if rng.random() < 0.5:
        chunk_cells = 0
    max_splats = int(rng.integers(20, 10001))
    lo = [rng.uniform(-100, 10), rng.uniform(-100, 1), rng.uniform(-100, 1)]
    hi = [rng.uniform(20, 100), rng.uniform(20, 100), rng.uniform(20, 100)]
    spacing = float(np.float32(rng.uniform(0.25, 2.5)))
    max_radius = rng.uniform(0.25, 10.0)
    n = int(sum(rng.integers(0, max_scan + 1) for _ in range(num_scans)))
    if n == 0:
        return
    splats = np.zeros(n, ob.SPLAT_DTYPE)
    for a in range(3):
        splats["position"][:, a] = rng.uniform(lo[a], hi[a], n).astype(np.float32)
    splats["radius"] = rng.uniform(0.01, max_radius, n).astype(np.float32)
    splats["normal"] = 1.0
    # bounding grid as FastBlobSet::getBoundingGrid: reference 0, extents = floor / floor + 1 of the splat boxes
    inv = np.float32(1.0) / np.float32(spacing)
    lows = np.floor((splats["position"] - splats["radius"][:, None]) * inv).min(axis=0).astype(np.int64)
    highs = np.floor((splats["position"] + splats["radius"][:, None]) * inv).max(axis=0).astype(np.int64)
    grid = dict(reference=(0.0, 0.0, 0.0), spacing=spacing,
                extents=(lows[0], highs[0] + 1, lows[1], highs[1] + 1, lows[2], highs[2] + 1))
    try:
        leaves = run(splats, max_splats, max_cells, chunk_cells, max_cells, max_split, grid)
    except ob.DensityError:
        return
    validate_partition(splats, grid, leaves, max_splats, max_cells, 0)


@pytest.mark.parametrize("seed", range(30))
def test_random_general(seed):
    """The cases the HIP bucketer is compared on (non-zero reference, other microblock requests, clipped grids)."""
    splats, grid, p = random_case(seed)
    try:
        leaves = run(splats, p["max_splats"], p["max_cells"], p["chunk_cells"], p["micro_cells"], p["max_split"], grid)
    except ob.DensityError:
        return
    validate_partition(splats, grid, leaves, p["max_splats"], p["max_cells"], 0, strict=seed % 4 != 3)


def test_rejects_an_empty_region():
    with pytest.raises(ValueError):
        run(create_splats(), 5, 8, 0, 8, 1000, dict(GRID, extents=(0, 10, 5, 5, 0, 10)))
