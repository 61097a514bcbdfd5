"""Pins the oracle's marching-tetrahedra restatement (test/test_marching.cpp, test/test_mesh_filter.cpp)."""
import numpy as np
import pytest

import oracle_binding as ob
from refdata import is_manifold, weld_batches

EXT = 1 << 63


def test_tables():
    # test/test_marching.cpp:270-305 + byte sizes src/marching.h:152-171, asserts src/marching.cpp:248-251
    count, start, data, key, data_size, key_entries = ob.make_tables()
    assert data_size == 8192 and key_entries == 2432
    assert int(start[256][1]) == data_size
    assert int(start[0][0]) == 0
    assert int(start[256][0]) == 2432
    for i in range(256):
        sv, si = int(start[i][0]), int(start[i][1])
        ev, ei = int(start[i + 1][0]), int(start[i + 1][1])
        assert int(count[i][0]) == ev - sv
        assert int(count[i][1]) == ei - si
        assert count[i][1] % 3 == 0
        assert count[i][0] <= 13 and count[i][1] <= 36
        for j in range(sv, ev):
            if j > sv:
                assert data[j - 1] < data[j]
            assert data[j] < 19
        for j in range(si, ei):
            assert data[j] < ev - sv
    assert count[0].tolist() == [0, 0] and count[255].tolist() == [0, 0]
    assert key.max() == 2


def make_key(x, y, z, external):
    return (z << 42) | (y << 21) | x | (EXT if external else 0)


def test_compute_key():
    # test/test_marching.cpp:333-351
    cases = [((0, 0, 0, True), (0, 0, 0), (32, 32, 32)), ((1, 2, 3, False), (1, 2, 3), (32, 32, 32)),
             ((0, 4, 5, True), (0, 4, 5), (32, 32, 32)), ((6, 0, 7, True), (6, 0, 7), (32, 32, 32)),
             ((9, 5, 0, False), (9, 5, 0), (32, 32, 32)), ((30, 1, 2, True), (30, 1, 2), (30, 40, 50)),
             ((5, 40, 3, True), (5, 40, 3), (30, 40, 50)), ((1, 2, 50, True), (1, 2, 50), (30, 40, 50)),
             ((1, 2, 40, False), (1, 2, 40), (30, 40, 50)), ((1, 2, 30, False), (1, 2, 30), (30, 40, 50)),
             ((30, 40, 50, True), (30, 40, 50), (30, 40, 50))]
    for exp, c, top in cases:
        got = ob.lib().orc_compute_key(ob._p(np.array(c, np.uint32)), ob._p(np.array(top, np.uint32)))
        assert got == make_key(*exp)


def compact_fixture():
    """Inputs of test/test_marching.cpp:401-425."""
    in_keys = np.array([100, 100, 200, EXT | 50, EXT | 50, 0xFFFFFFFFFFFFFFFF], np.uint64)
    unique = np.array([0, 0, 1, 2, 2, 3], np.uint32)
    ids = np.array([4, 1, 2, 3, 0], np.uint32)
    in_verts = np.zeros((5, 4), np.float32)
    for i in range(5):
        in_verts[i, :3] = (i, i + 1, i + 2)
    in_verts[:, 3] = ids.view(np.float32)
    return in_keys, unique, in_verts


COMPACT_EXPECT = [
    # (minExternalKey, outKeys expectations, firstExternal) test/test_marching.cpp:426-478
    (200, [None, 200, 50], 1),
    (100, [100, 200, 50], 0),
    (EXT | 60, [None, None, None], 3),
]


@pytest.mark.parametrize("min_ext,exp_keys,exp_first", COMPACT_EXPECT)
def test_compact_vertices(min_ext, exp_keys, exp_first):
    in_keys, unique, in_verts = compact_fixture()
    dead = 0xDEADBEEFDEADBEEF
    out_v = np.zeros(9, np.float32)
    out_k = np.full(3, dead, np.uint64)
    remap = np.full(5, 0xDEADBEEF, np.uint32)
    first = np.full(1, 0xDEADBEEF, np.uint32)
    ob.lib().orc_compact_vertices(ob._p(out_v), ob._p(out_k), ob._p(remap), ob._p(first), ob._p(unique),
                                  ob._p(in_verts), ob._p(in_keys), min_ext, 0, 5)
    assert out_v[0] == 1.0 and out_v[3] == 2.0 and out_v[6] == 4.0
    for i, k in enumerate(exp_keys):
        assert int(out_k[i]) == (dead if k is None else k)
    assert remap.tolist() == [2, 0, 1, 2, 0]
    assert int(first[0]) == exp_first


def test_copy_slice():
    # test/test_marching.cpp:481-548 (the kernel call then Marching::copySlice(image, 2, 0, params))
    values = np.array([[0.1, 0.1]] * 4 + [[1.5, -0.5], [2.0, -3.0]] + [[0.1, 0.1]] * 2, np.float32)
    expected = np.array([[1.5, -0.5], [2.0, -3.0]] * 3 + [[0.1, 0.1]] * 2, np.float32)
    m = ob.MarchingOracle(2, 2, 4, 11, 4096, (7, 5, 11))
    # first the raw kernel launch: rows 4..5 -> rows 2..3 (offset (0,-2))
    values[2:4] = values[4:6]
    # then copySlice(src=2, trg=0) with zStride 2, 2x2 payload
    ob.lib().orc_marching_copy_slice(m.h, ob._p(values), 2, 2, 0, 2, 2, 2)
    np.testing.assert_array_equal(values, expected)


def host_generator(fn):
    """HostGenerator::enqueue of test/test_marching.cpp:88-130 for a numpy field."""
    def gen(field, sw):
        ys, xs = np.meshgrid(np.arange(sw.height, dtype=np.uint32), np.arange(sw.width, dtype=np.uint32),
                             indexing="ij")
        for z in range(sw.zFirst, sw.zLast + 1):
            row0 = z * sw.zStride + sw.zBias
            field[row0:row0 + sw.height, :sw.width] = fn(xs, ys, z)
    return gen


def sphere_fn(cx, cy, cz, radius):
    """SphereGenerator::generate including its (y-cx)*(y-cy) quirk, test/test_marching.cpp:141-147."""
    cx, cy, cz, radius = map(np.float32, (cx, cy, cz, radius))

    def fn(x, y, z):
        x = x.astype(np.float32)
        y = y.astype(np.float32)
        z = np.float32(z)
        with np.errstate(invalid="ignore"):
            d = np.sqrt((x - cx) * (x - cx) + (y - cx) * (y - cy) + (z - cz) * (z - cz), dtype=np.float32)
        return d - radius
    return fn


def alternating_fn(x, y, z):
    return np.where(((x ^ y ^ np.uint32(z)) & 1) != 0, np.float32(1.0), np.float32(-1.0))


GENERATE_CASES = {
    # name: (max dims, dims, generator)  test/test_marching.cpp:594-632
    "sphere": ((83, 78, 66), (71, 75, 60), sphere_fn(30.0, 41.5, 27.75, 25.3)),
    "tsphere": ((83, 78, 66), (71, 75, 60), sphere_fn(0.5 * 71, 0.5 * 75, 0.5 * 60, 42.0)),
    "alternating": ((32, 32, 32), (32, 32, 32), alternating_fn),
}


@pytest.mark.parametrize("name", sorted(GENERATE_CASES))
def test_generate_manifold(name):
    """TestMarching::testGenerate: generate -> weld external keys -> Manifold::isManifold == ''."""
    (mw, mh, md), size, fn = GENERATE_CASES[name]
    alignment = (7, 5, 11)
    mesh_memory = (mw - 1) * (mh - 1) * 872
    m = ob.MarchingOracle(mw, mh, md, alignment[2], mesh_memory, alignment)
    batches = m.generate(host_generator(fn), size)
    st = m.stats()
    assert st["shipouts"] == len(batches) >= 1
    if name == "alternating":
        assert st["overflows"] > 0 or st["shipouts"] > 1   # "lots of geometry" exercises the flush logic
    v, t, _ = weld_batches(batches)
    assert len(t) > 0
    assert np.all(np.isfinite(v))
    assert is_manifold(len(v), t) == ""
    # per batch: internal vertices precede external ones and triangle indices are in range
    for b in batches:
        assert b["num_internal"] <= len(b["vertices"])
        if len(b["triangles"]):
            assert b["triangles"].max() < len(b["vertices"])
        ext = b["keys"][b["num_internal"]:]
        assert len(np.unique(ext)) == len(ext)


def test_generate_invariant_to_swathe_and_memory():
    """The welded mesh must not depend on how a bucket is cut into swathes / ship-outs."""
    from refdata import canonical_mesh
    (mw, mh, md), size, fn = GENERATE_CASES["sphere"]
    ref = None
    for swathe, mem_slices, alignment in [(11, 1, (7, 5, 11)), (66, 200, (7, 5, 11)), (8, 3, (8, 8, 8))]:
        m = ob.MarchingOracle(mw + 5, mh + 5, md + 6, swathe, (mw + 4) * (mh + 4) * 872 * mem_slices, alignment)
        got = canonical_mesh(m.generate(host_generator(fn), size, key_offset=(3, 4, 5)))
        if ref is None:
            ref = got
        else:
            np.testing.assert_array_equal(ref[0], got[0])
            np.testing.assert_array_equal(ref[1], got[1])


def test_marching_constructor_checks():
    # src/marching.cpp:356-363
    with pytest.raises(ValueError):
        ob.MarchingOracle(1, 2, 2, 11, 4096, (7, 5, 11))
    with pytest.raises(ValueError):
        ob.MarchingOracle(2, 2, 8193, 11, 4096, (7, 5, 11))
    with pytest.raises(ValueError):
        ob.MarchingOracle(2, 2, 2, 5, 4096, (7, 5, 11))      # alignment[2] > maxSwathe
    with pytest.raises(ValueError):
        ob.MarchingOracle(10, 10, 10, 11, 9 * 9 * 872 - 1, (7, 5, 11))


def test_scale_bias():
    # test/test_mesh_filter.cpp:284-361: 5 vertices, scale 3, bias (10,-20,30), tol 1e-2; empty mesh is fine
    v = np.array([[1, 2, 3], [-1, 0.5, 4], [0, 0, 0], [100, -50, 25], [1e-3, 2e-3, 3e-3]], np.float32)
    exp = v.astype(np.float64) * 3 + np.array([10, -20, 30])
    w = v.copy()
    ob.lib().orc_scale_bias(ob._p(w), 5, 3.0, 10.0, -20.0, 30.0)
    assert np.abs(w - exp).max() < 1e-2
    ob.lib().orc_scale_bias(ob._p(w), 0, 3.0, 10.0, -20.0, 30.0)


def test_compute_max_swathe():
    # src/workers.cpp:169-182; 24 for 256-corner buckets under the 8192-row limit
    assert ob.lib().orc_compute_max_swathe(8192, 256, 8, 8) == 24
    assert ob.lib().orc_compute_max_swathe(100, 256, 8, 8) == 8
