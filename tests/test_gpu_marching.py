"""Marching on the GPU: the reference's known answers (test/test_marching.cpp) and bit parity with the oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from gpu_common import assert_batches_equal, ctx  # noqa: F401
from refdata import is_manifold, weld_batches
from test_oracle_marching import (COMPACT_EXPECT, GENERATE_CASES, compact_fixture, host_generator, make_key)

pytestmark = pytest.mark.gpu


def test_tables(ctx):
    """testConstructor invariants on the DEVICE copies of the tables, which must equal the oracle's."""
    import mlsgpu_amd as m
    mc = m.Marching(ctx, 2, 2, 2, 11, 4096, (7, 5, 11))
    count, start, data, key = mc.tables()
    ocount, ostart, odata, okey, dsize, kentries = ob.make_tables()
    np.testing.assert_array_equal(count, ocount)
    np.testing.assert_array_equal(start, ostart)
    np.testing.assert_array_equal(data, odata)
    np.testing.assert_array_equal(key, okey)
    for i in range(256):
        sv, si = int(start[i][0]), int(start[i][1])
        ev, ei = int(start[i + 1][0]), int(start[i + 1][1])
        assert int(count[i][0]) == ev - sv and int(count[i][1]) == ei - si and count[i][1] % 3 == 0
        assert all(data[j - 1] < data[j] for j in range(sv + 1, ev)) and all(data[j] < 19 for j in range(sv, ev))
        assert all(data[j] < ev - sv for j in range(si, ei))


def test_compute_key(ctx):
    import mlsgpu_amd as m
    cases = [((0, 0, 0, True), (0, 0, 0), (32, 32, 32)), ((1, 2, 3, False), (1, 2, 3), (32, 32, 32)),
             ((0, 4, 5, True), (0, 4, 5), (32, 32, 32)), ((6, 0, 7, True), (6, 0, 7), (32, 32, 32)),
             ((9, 5, 0, False), (9, 5, 0), (32, 32, 32)), ((30, 1, 2, True), (30, 1, 2), (30, 40, 50)),
             ((5, 40, 3, True), (5, 40, 3), (30, 40, 50)), ((1, 2, 50, True), (1, 2, 50), (30, 40, 50)),
             ((1, 2, 40, False), (1, 2, 40), (30, 40, 50)), ((1, 2, 30, False), (1, 2, 30), (30, 40, 50)),
             ((30, 40, 50, True), (30, 40, 50), (30, 40, 50))]
    out = C.c_uint64()
    for exp, c, top in cases:
        m.binding.check(m.lib().mlsgpu_hip_test_compute_key(ctx.h, ob._p(np.array(c, np.uint32)),
                                                             ob._p(np.array(top, np.uint32)), C.byref(out)))
        assert out.value == make_key(*exp)


@pytest.mark.parametrize("min_ext,exp_keys,exp_first", COMPACT_EXPECT)
def test_compact_vertices(ctx, min_ext, exp_keys, exp_first):
    import mlsgpu_amd as m
    in_keys, unique, in_verts = compact_fixture()
    dead = 0xDEADBEEF
    out_v = m.DeviceBuffer(ctx, nbytes=36, fill=dead)
    out_k = m.DeviceBuffer(ctx, nbytes=24, fill=dead)
    remap = m.DeviceBuffer(ctx, nbytes=20, fill=dead)
    first = m.DeviceBuffer(ctx, nbytes=4, fill=dead)
    bufs = [m.DeviceBuffer(ctx, array=a) for a in (unique, in_verts, in_keys)]
    m.binding.check(m.lib().mlsgpu_hip_compact_vertices(ctx.h, out_v.ptr, out_k.ptr, remap.ptr, first.ptr,
                                                         bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, min_ext, 0, 5))
    v = out_v.download(np.float32)
    assert v[0] == 1.0 and v[3] == 2.0 and v[6] == 4.0
    k = out_k.download(np.uint64)
    for i, e in enumerate(exp_keys):
        assert int(k[i]) == (0xDEADBEEFDEADBEEF if e is None else e)
    assert remap.download(np.uint32).tolist() == [2, 0, 1, 2, 0]
    assert int(first.download(np.uint32)[0]) == exp_first


def test_copy_slice(ctx):
    import mlsgpu_amd as m
    values = np.array([[0.1, 0.1]] * 4 + [[1.5, -0.5], [2.0, -3.0]] + [[0.1, 0.1]] * 2, np.float32)
    expected = np.array([[0.1, 0.1]] * 2 + [[1.5, -0.5], [2.0, -3.0]] * 2 + [[0.1, 0.1]] * 2, np.float32)
    mc = m.Marching(ctx, 2, 2, 4, 11, 4096, (7, 5, 11))
    img = m.DeviceBuffer(ctx, array=values)
    mc.copy_slice(img, 2, 2, 1, 2, 2, 2)            # slice 2 (rows 4-5) -> slice 1 (rows 2-3)
    np.testing.assert_array_equal(img.download(np.float32).reshape(8, 2), expected)
    mc.copy_slice(img, 2, 2, 0, 2, 2, 2)            # then Marching::copySlice(image, 2, 0, params)
    expected[0:2] = expected[4:6]
    np.testing.assert_array_equal(img.download(np.float32).reshape(8, 2), expected)


@pytest.mark.parametrize("name", sorted(GENERATE_CASES))
def test_generate_manifold_and_parity(ctx, name):
    """TestMarching::testGenerate: manifold after welding, and every batch bit-identical to the oracle's."""
    import mlsgpu_amd as m
    (mw, mh, md), size, fn = GENERATE_CASES[name]
    alignment = (7, 5, 11)
    mesh_memory = (mw - 1) * (mh - 1) * 872
    mc = m.Marching(ctx, mw, mh, md, alignment[2], mesh_memory, alignment)
    got = mc.generate(m.binding.HostGenerator(ctx, fn, alignment), size)
    v, t, _ = weld_batches(got)
    assert len(t) > 0
    assert is_manifold(len(v), t) == ""
    oracle = ob.MarchingOracle(mw, mh, md, alignment[2], mesh_memory, alignment)
    exp = oracle.generate(host_generator(fn), size)
    assert_batches_equal(got, exp)
    st, cnt = oracle.stats(), mc.counters()
    for k in ("shipouts", "overflows", "occupied", "unwelded", "indices", "welded", "external"):
        assert st[k] == cnt[k], k
    # calling generate again on the same object gives the same result (no stale device state)
    again = mc.generate(m.binding.HostGenerator(ctx, fn, alignment), size)
    assert_batches_equal(again, exp)


@pytest.mark.parametrize("swathe,mem_slices,alignment,key_offset", [
    (8, 1, (8, 8, 8), (0, 0, 0)), (64, 300, (8, 8, 8), (100, 200, 300)), (24, 3, (8, 8, 8), (5, 6, 7)),
    # one swathe for the whole volume: the sort-free lattice weld, incl. overflow splitting and mid-bucket ship-outs
    (64, 1, (8, 8, 8), (0, 0, 0)), (72, 2, (8, 8, 8), (9, 8, 7)), (66, 7, (7, 5, 11), (1, 0, 2))])
def test_generate_swathe_and_memory_variants(ctx, swathe, mem_slices, alignment, key_offset):
    """Same field through different swathe sizes / mesh memories: every variant equals the oracle run with the
    same parameters (overflow splitting, mid-bucket ship-outs, external flags at ship-out boundaries)."""
    import mlsgpu_amd as m
    (mw, mh, md), size, fn = GENERATE_CASES["tsphere"]
    mw, mh, md = 88, 80, 72
    mesh_memory = (mw - 1) * (mh - 1) * 872 * mem_slices
    mc = m.Marching(ctx, mw, mh, md, swathe, mesh_memory, alignment)
    ctx.reset_stats()
    got = mc.generate(m.binding.HostGenerator(ctx, fn, alignment), size, key_offset)
    oracle = ob.MarchingOracle(mw, mh, md, swathe, mesh_memory, alignment)
    exp = oracle.generate(host_generator(fn), size, key_offset)
    assert_batches_equal(got, exp)
    # the reference's own statistics of Marching (src/marching.cpp:350-352, 655, 739, 822), under its names, in the context's
    # registry: sum and number of samples
    st, cnt, ost = ctx.stats(), mc.counters(), oracle.stats()
    assert st["marching.shipouts"] == (float(len(got)), 1)
    assert st["marching.slices.nonempty"][0] == cnt["nonempty"] and st["marching.slices.nonempty"][1] >= cnt["nonempty"] >= 1
    assert st.get("marching.overflow", (0.0, 0))[0] == cnt["overflows"] == ost["overflows"]


@pytest.mark.parametrize("name", sorted(GENERATE_CASES))
def test_lattice_and_sort_welds_agree(ctx, name, monkeypatch):
    """The two weld mechanisms (lattice ranks vs. key sort) must produce identical batches."""
    import mlsgpu_amd as m
    (mw, mh, md), size, fn = GENERATE_CASES[name]
    alignment = (8, 8, 8)
    mw, mh, md = mw + 5, mh + 2, 72
    mesh_memory = (mw - 1) * (mh - 1) * 872 * 5
    # maxSwathe 64 < maxDepth 72 keeps the sort buffers allocated; depth 60/32 <= 64 selects the lattice weld
    mc = m.Marching(ctx, mw, mh, md, 64, mesh_memory, alignment)
    monkeypatch.delenv("MLSGPU_HIP_WELD", raising=False)
    lattice = mc.generate(m.binding.HostGenerator(ctx, fn, alignment), size, (3, 1, 4))
    monkeypatch.setenv("MLSGPU_HIP_WELD", "sort")
    sort = mc.generate(m.binding.HostGenerator(ctx, fn, alignment), size, (3, 1, 4))
    monkeypatch.delenv("MLSGPU_HIP_WELD")
    assert len(lattice) >= 1
    assert_batches_equal(lattice, sort)
    exp = ob.MarchingOracle(mw, mh, md, 64, mesh_memory, alignment).generate(host_generator(fn), size, (3, 1, 4))
    assert_batches_equal(lattice, exp)


@pytest.mark.parametrize("route", ["0", "1"])
@pytest.mark.parametrize("case", ["tsphere", "noise"])
def test_triangle_routes_agree(ctx, case, route, monkeypatch):
    """The lattice weld emits its index list by rows of cells (dense data) or from a compacted cell list (surface-like
    data), chosen per ship-out by the share of occupied cells: forced either way, on a surface and on a noise field
    (every code, NaN holes), every batch equals the oracle's."""
    import mlsgpu_amd as m
    monkeypatch.setenv("MLSGPU_HIP_TRIANGLES_BY_CELLS", route)
    alignment = (8, 8, 8)
    if case == "noise":
        size, fn = (97, 67, 41), noise_fn(205, 0.03)
    else:
        (_, _, _), size, fn = GENERATE_CASES["tsphere"]
    mw, mh, md = size[0] + 3, size[1] + 2, size[2] + 5
    for mesh_memory in ((mw - 1) * (mh - 1) * 872 * 400, (mw - 1) * (mh - 1) * 872 * 2):
        mc = m.Marching(ctx, mw, mh, md, 64, mesh_memory, alignment)
        got = mc.generate(m.binding.HostGenerator(ctx, fn, alignment), size, (7, 0, 3))
        exp = ob.MarchingOracle(mw, mh, md, 64, mesh_memory, alignment).generate(host_generator(fn), size, (7, 0, 3))
        assert len(got) >= 1
        assert_batches_equal(got, exp)


def test_empty_and_degenerate(ctx):
    import mlsgpu_amd as m
    mc = m.Marching(ctx, 16, 16, 16, 8, 15 * 15 * 872, (8, 8, 8))
    all_out = m.binding.HostGenerator(ctx, lambda x, y, z: np.ones(x.shape, np.float32), (8, 8, 8))
    assert mc.generate(all_out, (16, 16, 16)) == []                       # no surface: no ship-out at all
    nan = m.binding.HostGenerator(ctx, lambda x, y, z: np.full(x.shape, np.nan, np.float32), (8, 8, 8))
    assert mc.generate(nan, (16, 16, 16)) == []
    assert mc.generate(all_out, (1, 16, 16)) == []                        # width 1: no cells
    assert mc.generate(all_out, (16, 16, 1)) == []                        # depth 1: no cells


def test_constructor_and_generate_checks(ctx):
    import mlsgpu_amd as m
    with pytest.raises(m.InvalidArgument):
        m.Marching(ctx, 1, 2, 2, 11, 4096, (7, 5, 11))
    with pytest.raises(m.InvalidArgument):
        m.Marching(ctx, 2, 2, 8193, 11, 4096, (7, 5, 11))
    with pytest.raises(m.InvalidArgument):
        m.Marching(ctx, 2, 2, 2, 5, 4096, (7, 5, 11))
    with pytest.raises(m.InvalidArgument):
        m.Marching(ctx, 10, 10, 10, 11, 9 * 9 * 872 - 1, (7, 5, 11))
    mc = m.Marching(ctx, 16, 16, 16, 8, 15 * 15 * 872, (8, 8, 8))
    gen = m.binding.HostGenerator(ctx, lambda x, y, z: np.ones(x.shape, np.float32), (8, 8, 8))
    with pytest.raises(m.LengthError):
        mc.generate(gen, (17, 16, 16))                  # src/marching.cpp:764-766


def test_scale_bias(ctx):
    """test/test_mesh_filter.cpp:284-361: 5 vertices, scale 3, bias (10,-20,30); empty mesh is legal."""
    import mlsgpu_amd as m
    v = np.array([[1, 2, 3], [-1, 0.5, 4], [0, 0, 0], [100, -50, 25], [1e-3, 2e-3, 3e-3]], np.float32)
    dv = m.DeviceBuffer(ctx, array=v)
    mesh = m.binding.Mesh(dv.ptr, None, None, 5, 0, 5)
    m.binding.check(m.lib().mlsgpu_hip_scale_bias(ctx.h, C.byref(mesh), 3.0, 10.0, -20.0, 30.0))
    ctx.synchronize()
    got = dv.download(np.float32).reshape(5, 3)
    assert np.abs(got - (v.astype(np.float64) * 3 + np.array([10, -20, 30]))).max() < 1e-2
    exp = v.copy()
    ob.lib().orc_scale_bias(ob._p(exp), 5, 3.0, 10.0, -20.0, 30.0)
    np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32))
    empty = m.binding.Mesh(None, None, None, 0, 0, 0)
    m.binding.check(m.lib().mlsgpu_hip_scale_bias(ctx.h, C.byref(empty), 3.0, 10.0, -20.0, 30.0))


def noise_fn(seed, hole_rate):
    """A hash-noise field with NaN holes: every cube code, every lattice word boundary, cells dropped by isValid."""
    def fn(x, y, z):
        h = (x.astype(np.uint64) * np.uint64(73856093)) ^ (y.astype(np.uint64) * np.uint64(19349663)) \
            ^ (np.uint64(z) * np.uint64(83492791)) ^ np.uint64(seed * 2654435761)
        h = (h * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)
        v = (h.astype(np.float64) / float(1 << 24) - 0.5).astype(np.float32)
        v = np.where(v == 0, np.float32(0.25), v)
        holes = ((h >> np.uint64(7)) % np.uint64(1000)) < np.uint64(int(hole_rate * 1000))
        return np.where(holes, np.float32(np.nan), v)
    return fn


@pytest.mark.parametrize("size,swathe,weld", [((97, 35, 41), 48, "lattice"), ((131, 67, 9), 16, "lattice"), ((33, 129, 20), 24, "lattice"),
                                               ((97, 35, 41), 16, "sort"), ((9, 10, 11), 16, "lattice"), ((65, 9, 10), 16, "lattice")])
def test_noise_fields_with_holes(ctx, size, swathe, weld, monkeypatch):
    """Random fields (every code, NaN holes) on ragged sizes around the 32-cell word boundaries of the lattice weld."""
    import mlsgpu_amd as m
    if weld == "sort":
        monkeypatch.setenv("MLSGPU_HIP_WELD", "sort")
    fn = noise_fn(sum(size), 0.03)
    alignment = (8, 8, 8)
    mw, mh, md = size[0] + 3, size[1] + 2, size[2] + 5
    for mesh_memory in ((mw - 1) * (mh - 1) * 872 * 400, (mw - 1) * (mh - 1) * 872 * 2):
        mc = m.Marching(ctx, mw, mh, md, swathe, mesh_memory, alignment)
        got = mc.generate(m.binding.HostGenerator(ctx, fn, alignment), size, (7, 0, 3))
        oracle = ob.MarchingOracle(mw, mh, md, swathe, mesh_memory, alignment)
        exp = oracle.generate(host_generator(fn), size, (7, 0, 3))
        assert_batches_equal(got, exp)
        st, cnt = oracle.stats(), mc.counters()
        for k in ("shipouts", "overflows", "occupied", "unwelded", "indices", "welded", "external"):
            assert st[k] == cnt[k], k


@pytest.mark.parametrize("mem_slices", [300, 2])
def test_generate_batch_with_host_generators(ctx, mem_slices):
    """mlsgpu_hip_marching_generate_batch with generators that are NOT MlsFunctors (host-filled fields): the three
    reference cases of TestMarching::testGenerate as ONE batch over three Marching objects -- one set of launches with a
    bucket dimension, the swathe totals and welded counts of all three read back together -- at different sizes and key
    offsets.  Every bucket's ship-outs equal the oracle's run for that bucket alone, bit for bit; with two slices' worth of
    mesh memory the buckets overflow and take the one-bucket path behind the shared launches (same batch structure)."""
    import mlsgpu_amd as m
    alignment = (8, 8, 8)
    names = sorted(GENERATE_CASES)
    dims = (88, 80, 72)
    mesh_memory = (dims[0] - 1) * (dims[1] - 1) * 872 * mem_slices
    marchings = [m.Marching(ctx, dims[0], dims[1], dims[2], 72, mesh_memory, alignment) for _ in names]
    gens = [m.binding.HostGenerator(ctx, GENERATE_CASES[n][2], alignment) for n in names]
    sizes = [GENERATE_CASES[n][1] for n in names]
    offsets = [(3 * k, 100 + k, 7 * k) for k in range(len(names))]
    got = m.Marching.generate_batch(marchings, gens, sizes, offsets)
    splits = 0
    for k, n in enumerate(names):
        oracle = ob.MarchingOracle(dims[0], dims[1], dims[2], 72, mesh_memory, alignment)
        exp = oracle.generate(host_generator(GENERATE_CASES[n][2]), sizes[k], offsets[k])
        assert_batches_equal(got[k], exp)
        st, cnt = oracle.stats(), marchings[k].counters()
        for key in ("shipouts", "overflows", "occupied", "unwelded", "indices", "welded", "external"):
            assert st[key] == cnt[key], (n, key)
        splits += st["shipouts"] > 1
    assert (splits > 0) == (mem_slices == 2)
    # a batch of one is the one-bucket entry point
    one = m.Marching.generate_batch(marchings[:1], gens[:1], sizes[:1], offsets[:1])
    assert_batches_equal(one[0], got[0])
