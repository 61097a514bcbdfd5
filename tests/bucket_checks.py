"""Fixture and property checker of the reference's bucketing tests, shared by the oracle and the HIP tests."""
import numpy as np

import oracle_binding as ob


def make_splat(x, y, z, radius):
    # test/test_splat_set.cpp:66-78
    s = np.zeros((), ob.SPLAT_DTYPE)
    s["position"] = (x, y, z)
    s["radius"] = radius
    s["normal"] = (1.0, 0.0, 0.0)
    s["quality"] = 1.0
    return s


def create_splats():
    """createSplats, test/test_splat_set.cpp:80-107 (five scans, two of them empty, flattened)."""
    z = 10.0
    rows = [(10, 20, z, 2), (30, 17, z, 1), (32, 12, z, 1), (32, 18, z, 1), (37, 18, z, 1), (35, 16, z, 3),
            (12, 37, z, 1), (13, 37, z, 1), (12, 38, z, 1), (13, 38, z, 1), (17, 32, z, 1),
            (18, 33, z, 1), (25, 45, z, 4)]
    out = np.zeros(len(rows), ob.SPLAT_DTYPE)
    for i, r in enumerate(rows):
        out[i] = make_splat(*[float(v) for v in r])
    return out


def _div_down(a, b):
    return a // b


def cell_box(splats, grid):
    """splatToBuckets with bucket size 1 for every splat: (lower, upper) cell coordinates relative to the grid"""
    inv = np.float32(1.0) / np.float32(grid["spacing"])
    ref = np.asarray(grid["reference"], np.float32)
    first = np.asarray(grid["extents"], np.int64).reshape(3, 2)[:, 0]
    lo = np.floor(((splats["position"] - splats["radius"][:, None]) - ref) * inv).astype(np.int64) - first
    hi = np.floor(((splats["position"] + splats["radius"][:, None]) - ref) * inv).astype(np.int64) - first
    return lo, hi


def validate_partition(splats, grid, leaves, max_splats, max_cells, chunk_cells, strict=True):
    """TestBucket::validate, test/test_bucket.cpp:345-451.  strict=False: the region cuts splats off, so a bucket may
    list splats of its last, partial microblock that lie beyond the border ("the intersection test is conservative so
    there may be extras", src/bucket.h:101-103); the reference only validates bounding grids."""
    full = np.asarray(grid["extents"], np.int64).reshape(3, 2)
    finite = np.isfinite(splats["position"]).all(axis=1) & np.isfinite(splats["radius"]) \
        & np.isfinite(splats["normal"]).all(axis=1) & np.isfinite(splats["quality"])
    areas = np.zeros(len(splats), np.int64)
    boxes = []
    for leaf in leaves:
        ext = np.asarray(leaf["extents"], np.int64).reshape(3, 2)
        ids = leaf["ids"].astype(np.int64)
        cells = ext[:, 1] - ext[:, 0]
        assert 0 < len(ids) <= max_splats
        assert (cells <= max_cells).all() and (cells > 0).all()
        assert (full[:, 0] <= ext[:, 0]).all() and (ext[:, 1] <= full[:, 1]).all()
        if chunk_cells:
            assert ((ext[:, 0] - full[:, 0]) // chunk_cells == (ext[:, 1] - full[:, 0] - 1) // chunk_cells).all()
        assert (np.diff(ids) > 0).all()
        assert finite[ids].all()
        sub = dict(grid, extents=tuple(int(v) for v in ext.reshape(6)))
        lo, hi = cell_box(splats[ids], sub)
        lo = np.maximum(lo, 0)
        hi = np.minimum(hi, cells - 1)
        meets = (lo <= hi).all(axis=1)
        assert meets.all() or not strict              # every listed splat meets the block
        areas[ids] += np.where(meets, np.prod(np.maximum(hi - lo + 1, 0), axis=1), 0)
        boxes.append(ext)
    for i in range(len(boxes)):
        for j in range(i + 1, len(boxes)):
            a, b = boxes[i], boxes[j]
            assert ((a[:, 1] <= b[:, 0]) | (b[:, 1] <= a[:, 0])).any(), "blocks overlap"
    # every splat is fully covered: the blocks' intersections add up to its box clipped to the full grid
    ok = np.flatnonzero(finite)
    lo, hi = cell_box(splats[ok], grid)
    lo = np.maximum(lo, 0)
    hi = np.minimum(hi, (full[:, 1] - full[:, 0]) - 1)
    want = np.where((lo <= hi).all(axis=1), np.prod(np.maximum(hi - lo + 1, 0), axis=1), 0)
    np.testing.assert_array_equal(areas[ok], want)
    assert (areas[~finite] == 0).all()


def random_case(seed):
    """Random cloud, grid and parameters in the ranges of test/test_bucket.cpp:591-640, plus references that are not
    zero, requested microblock sizes other than maxCells and grids that cut splats off at the border."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 20000))
    max_split = int(rng.integers(64, 1001))
    max_cells = int(rng.integers(40, 101))
    chunk_cells = int(rng.integers(80, 514)) if rng.random() < 0.5 else 0
    max_splats = int(rng.integers(20, 10001))
    micro = [max_cells, 0, int(rng.integers(1, max_cells + 1))][seed % 3]
    spacing = float(np.float32(rng.uniform(0.25, 2.5)))
    splats = np.zeros(n, ob.SPLAT_DTYPE)
    lo = rng.uniform(-100, 1, 3)
    hi = rng.uniform(20, 100, 3)
    splats["position"] = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    splats["radius"] = rng.uniform(0.01, rng.uniform(0.25, 10.0), n).astype(np.float32)
    splats["normal"] = 1.0
    inv = np.float32(1.0) / np.float32(spacing)
    ref = (float(np.float32(rng.uniform(-5, 5))), 0.0, float(np.float32(rng.uniform(-5, 5))))
    lows = np.floor((splats["position"] - splats["radius"][:, None] - np.float32(ref)) * inv).min(axis=0).astype(np.int64)
    highs = np.floor((splats["position"] + splats["radius"][:, None] - np.float32(ref)) * inv).max(axis=0).astype(np.int64)
    shrink = 7 if (seed % 4 == 3 and (highs - lows).min() > 20) else 0
    grid = dict(reference=ref, spacing=spacing,
                extents=(lows[0] + shrink, highs[0] + 1 - shrink, lows[1], highs[1] + 1, lows[2] + shrink, highs[2] + 1))
    return splats, grid, dict(max_splats=max_splats, max_cells=max_cells, chunk_cells=chunk_cells, micro_cells=micro,
                              max_split=max_split)
