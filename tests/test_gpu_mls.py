"""MlsFunctor on the GPU: the reference's known answers (test/test_mls.cpp) and bit parity with the oracle."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_binding as ob
from gpu_common import ctx  # noqa: F401
from refdata import sphere_splats
from test_oracle_mls import (EPS, SOLVE_CASES, check_process_corners, close, f32, process_corners_fixture)

pytestmark = pytest.mark.gpu


def test_solve_quadratic(ctx):
    import mlsgpu_amd as m
    out = C.c_float()
    for expected, a, b, c in SOLVE_CASES:
        m.binding.check(m.lib().mlsgpu_hip_test_solve_quadratic(ctx.h, a, b, c, C.byref(out)))
        assert close(expected, out.value, 4 * EPS), (expected, out.value, a, b, c)
        # and bit-identical to the oracle
        exp = np.float32(ob.lib().orc_solve_quadratic(a, b, c))
        assert np.float32(out.value).view(np.uint32) == exp.view(np.uint32) or (math.isnan(out.value) and math.isnan(exp))


def test_fit_sphere(ctx):
    import mlsgpu_amd as m
    n = 20
    splats = sphere_splats(n, (1.0, 2.0, 3.5), 6.5)
    params = np.zeros(5, np.float32)
    m.binding.check(m.lib().mlsgpu_hip_test_fit_sphere(ctx.h, ob._p(splats), n, ob._p(params)))
    eps = 16 * EPS
    p = [float(x) for x in params]
    for i in range(n):
        x, y, z = (float(v) for v in splats["position"][i])
        v = p[0] * x + p[1] * y + p[2] * z + p[3] * (x * x + y * y + z * z) + p[4]
        assert close(0.0, v, eps)
        g = [f32(2 * p[3] * x + p[0]), f32(2 * p[3] * y + p[1]), f32(2 * p[3] * z + p[2])]
        for k in range(3):
            assert close(float(splats["normal"][i][k]), g[k], eps)
    exp = np.zeros(5, np.float32)
    ob.lib().orc_fit_sphere(ob._p(splats), n, ob._p(exp))
    np.testing.assert_array_equal(params.view(np.uint32), exp.view(np.uint32))


@pytest.mark.parametrize("variant", [1, 4, 5])
@pytest.mark.parametrize("shape", [0, 1])
def test_process_corners(ctx, variant, shape):
    """TestMls::testProcessCorners (hand-built command list: >= 4 hits / < 4 hits / no hits)."""
    import mlsgpu_amd as m
    fx = process_corners_fixture()
    rows, pitch = fx["rows"], fx["image_w"]
    field = m.DeviceBuffer(ctx, array=np.full((rows, pitch), -12345.0, np.float32))
    dsplats = m.DeviceBuffer(ctx, array=fx["splats"])
    dcommands = m.DeviceBuffer(ctx, array=fx["commands"])
    dstart = m.DeviceBuffer(ctx, array=fx["start"])
    gen = m.MlsFunctor(ctx, shape)
    gen.set_variant(variant)
    gen.set_buffers(fx["offset"], dsplats, dcommands, dstart, fx["subsampling"])
    sw = m.Swathe(fx["size"][0], fx["size"][1], fx["z_stride"], fx["z_bias"], fx["z_first"], fx["z_last"])
    gen.enqueue(field, pitch, rows, sw)
    ctx.synchronize()
    got = field.download(np.float32).reshape(rows, pitch)
    if shape == 0:
        assert check_process_corners(fx, got) == 0
    exp = np.full((rows, pitch), -12345.0, np.float32)
    ob.process_corners(exp, fx["splats"], fx["commands"], fx["start"], fx["subsampling"], fx["offset"],
                       fx["size"][0], fx["size"][1], fx["z_stride"], fx["z_bias"], fx["z_first"], fx["z_last"],
                       ob.lib().orc_boundary_factor(1.0), shape)
    np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32))
    first_row = fx["z_first"] * fx["z_stride"] + fx["z_bias"]
    assert np.all(got[:first_row] == -12345.0)          # lower slices untouched


def _run_process_corners(ctx, fx, variant, shape, splats):
    import mlsgpu_amd as m
    rows, pitch = fx["rows"], fx["image_w"]
    field = m.DeviceBuffer(ctx, array=np.full((rows, pitch), -12345.0, np.float32))
    dsplats = m.DeviceBuffer(ctx, array=splats)
    dcommands = m.DeviceBuffer(ctx, array=fx["commands"])
    dstart = m.DeviceBuffer(ctx, array=fx["start"])
    gen = m.MlsFunctor(ctx, shape)
    gen.set_variant(variant)
    gen.set_buffers(fx["offset"], dsplats, dcommands, dstart, fx["subsampling"])
    sw = m.Swathe(fx["size"][0], fx["size"][1], fx["z_stride"], fx["z_bias"], fx["z_first"], fx["z_last"])
    gen.enqueue(field, pitch, rows, sw)
    ctx.synchronize()
    got = field.download(np.float32).reshape(rows, pitch)
    exp = np.full((rows, pitch), -12345.0, np.float32)
    ob.process_corners(exp, splats, fx["commands"], fx["start"], fx["subsampling"], fx["offset"],
                       fx["size"][0], fx["size"][1], fx["z_stride"], fx["z_bias"], fx["z_first"], fx["z_last"],
                       ob.lib().orc_boundary_factor(1.0), shape)
    return got, exp


NEG_NAN = np.array([0xFFC00000], np.uint32).view(np.float32)[0]
POS_NAN = np.array([0x7FC00000], np.uint32).view(np.float32)[0]


@pytest.mark.parametrize("variant", [1, 4, 5])
@pytest.mark.parametrize("shape", [0, 1])
def test_process_corners_non_finite_splats(ctx, variant, shape):
    """The hand-built list of testProcessCorners (test/test_mls.cpp:416-514; the splat slot that holds the radius is taken
    as 1/r^2 there) with splats no valid octree would list: position / 1/r^2 / quality / normal that are NaN of either sign,
    +-inf, zero or negative.  The reference rejects a splat with `d < 0.99` per corner, which a NaN of any sign fails; the
    default kernel takes the hit from a sign bit and must reject the same splats.  Bit-equal to the oracle,
    NaN for NaN."""
    fx = process_corners_fixture()
    base = fx["splats"]
    n = len(base)
    poisons = [
        ("position", 0, NEG_NAN), ("position", 1, POS_NAN), ("position", 2, NEG_NAN), ("position", 0, np.float32(np.inf)),
        ("position", 2, np.float32(-np.inf)), ("radius", None, NEG_NAN), ("radius", None, POS_NAN),
        ("radius", None, np.float32(np.inf)), ("radius", None, np.float32(-np.inf)), ("radius", None, np.float32(0.0)),
        ("radius", None, np.float32(-0.0004)), ("quality", None, NEG_NAN), ("quality", None, np.float32(np.inf)),
        ("normal", 1, NEG_NAN), ("quality", None, np.float32(0.0)),
    ]
    # one poisoned splat at a time (a wrong hit is then visible as a changed corner), then several at once
    cases = [[(i * 3 + 1) % n] for i in range(len(poisons))] + [[(i * 3 + 1) % n for i in range(len(poisons))]]
    for ci, where in enumerate(cases):
        splats = base.copy()
        for j, idx in enumerate(where):
            field, comp, value = poisons[ci if len(where) == 1 else j]
            if comp is None:
                splats[field][idx] = value
            else:
                splats[field][idx][comp] = value
        got, exp = _run_process_corners(ctx, fx, variant, shape, splats)
        np.testing.assert_array_equal(np.isnan(got), np.isnan(exp), err_msg=str(ci))
        ok = ~np.isnan(exp)
        np.testing.assert_array_equal(got[ok].view(np.uint32), exp[ok].view(np.uint32), err_msg=str(ci))
        if len(where) == 1 and poisons[ci][0] in ("position", "radius") and np.isnan(poisons[ci][2]):
            # a NaN position or 1/r^2 removes that splat and nothing else: the field is the one with the splat far away
            far = base.copy()
            far["position"][where[0]] = (1e6, 1e6, 1e6)
            _, exp_far = _run_process_corners(ctx, fx, variant, shape, far)
            assert np.array_equal(np.isnan(exp), np.isnan(exp_far))
            same = ~np.isnan(exp_far)
            np.testing.assert_array_equal(got[same].view(np.uint32), exp_far[same].view(np.uint32))


def cutoff_fixture(seed=5, n=600):
    """One command list shared by the eight 8^3 blocks of a 16^3 grid whose splats sit ON the reference's `d < 0.99`
    (kernels/mls.cl:371) for one corner each: d = |p - c|^2 / r^2 within a few float steps of the cutoff on either side."""
    rng = np.random.default_rng(seed)
    offset = (100, 200, 300)
    splats = np.zeros(n, ob.SPLAT_DTYPE)
    cut = np.float32(0.99)
    for i in range(n):
        c = (rng.integers(0, 16, 3) + np.array(offset)).astype(np.float32)
        u = rng.normal(size=3)
        u /= np.linalg.norm(u)
        r = rng.uniform(1.2, 3.5)
        invr2 = np.float32(1.0 / (r * r))
        dist = math.sqrt(0.99 / float(invr2))
        p = (c.astype(np.float64) + u * dist).astype(np.float32)

        def dof(p):
            q = p - c
            return np.float32(np.float32(np.float32(q[0] * q[0]) + np.float32(q[1] * q[1])) + np.float32(q[2] * q[2])) * invr2
        # walk the largest component of p in single float steps until d crosses the cutoff, then move -2 .. 2 steps on
        k = int(np.argmax(np.abs(u)))
        away = np.float32(np.inf) if u[k] > 0 else np.float32(-np.inf)
        for _ in range(64):
            if dof(p) >= cut:
                break
            p[k] = np.nextafter(p[k], away)
        for _ in range(64):
            if dof(p) < cut:
                break
            p[k] = np.nextafter(p[k], -away)
        for _ in range(int(rng.integers(0, 5))):
            p[k] = np.nextafter(p[k], away)
        splats["position"][i] = p
        splats["radius"][i] = invr2
        nrm = rng.normal(size=3)
        splats["normal"][i] = (nrm / np.linalg.norm(nrm)).astype(np.float32)
        splats["quality"][i] = np.float32(rng.uniform(0.5, 2.0))
    commands = np.array([n + 1] + list(range(n)) + [-1], np.int32)
    start = np.zeros(8, np.int32)
    return dict(offset=offset, splats=splats, commands=commands, start=start, subsampling=3, size=(16, 16, 16),
                image_w=16, rows=16 * 16, z_stride=16, z_bias=0, z_first=0, z_last=15)


@pytest.mark.parametrize("variant", [1, 4, 5])
def test_process_corners_on_the_cutoff(ctx, variant):
    """Splats within a few float steps of `d < 0.99` on both sides (the matrix prefilter's margin must keep every hit of the
    reference's test and the drain must drop what the margin lets through): bit-equal to the oracle, and the instrumented
    kernel's own superset check reads zero misses."""
    import mlsgpu_amd as m
    fx = cutoff_fixture()
    got, exp = _run_process_corners(ctx, fx, variant, 0, fx["splats"])
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    ok = ~np.isnan(exp)
    np.testing.assert_array_equal(got[ok].view(np.uint32), exp[ok].view(np.uint32))
    # near-cutoff hits exist on both sides: count the reference's test on the fixture itself
    counters = m.DeviceBuffer(ctx, array=np.zeros(m.binding.MLS_STATS_WORDS, np.uint64))
    field = m.DeviceBuffer(ctx, array=np.zeros((fx["rows"], fx["image_w"]), np.float32))
    gen = m.MlsFunctor(ctx, 0)
    gen.set_variant(variant)
    ds, dc, dst = (m.DeviceBuffer(ctx, array=fx[k]) for k in ("splats", "commands", "start"))
    gen.set_buffers(fx["offset"], ds, dc, dst, fx["subsampling"])
    gen.set_stats(counters)
    gen.enqueue(field, fx["image_w"], fx["rows"], m.Swathe(16, 16, 16, 0, 0, 15))
    ctx.synchronize()
    c = counters.download(np.uint64)
    got2 = field.download(np.float32).reshape(fx["rows"], fx["image_w"])
    np.testing.assert_array_equal(got2[ok].view(np.uint32), exp[ok].view(np.uint32))      # the instrumented kernel too
    assert c[0] == 8 * len(fx["splats"]) and c[2] > 0
    if variant == 5:
        assert c[42] == 0, "the prefilter missed %d hits" % c[42]
        assert c[41] >= c[2]                                            # candidates >= hits
        assert c[41] <= 1.5 * c[2] + 64                                 # and not many more


def dense_fixture(seed=9, n=1300):
    """A list whose every splat reaches every sub-block of its block (radii of 10-14 cells around the 16^3 grid): all 512 staged
    splats of a round are relevant to each wave, more than its slot table holds (440), so the table is worked off and refilled
    inside a round; the list has two segments joined by a jump (kernels/mls.cl:354-358) and three rounds in the first."""
    rng = np.random.default_rng(seed)
    offset = (40, 50, 60)
    splats = np.zeros(n, ob.SPLAT_DTYPE)
    splats["position"] = (rng.uniform(-2.0, 18.0, (n, 3)) + np.array(offset)).astype(np.float32)
    r = rng.uniform(10.0, 14.0, n)
    splats["radius"] = (1.0 / (r * r)).astype(np.float32)
    nrm = rng.normal(size=(n, 3))
    splats["normal"] = (nrm / np.linalg.norm(nrm, axis=1)[:, None]).astype(np.float32)
    splats["quality"] = rng.uniform(0.5, 2.0, n).astype(np.float32)
    first = 1100                                            # ids 0 .. 1099, then a jump to the second segment
    seg2 = first + 2
    commands = np.array([first + 1] + list(range(first)) + [seg2]
                        + [seg2 + 1 + (n - first)] + list(range(first, n)) + [-1], np.int32)
    start = np.zeros(8, np.int32)
    return dict(offset=offset, splats=splats, commands=commands, start=start, subsampling=3, size=(16, 16, 16),
                image_w=16, rows=16 * 16, z_stride=16, z_bias=0, z_first=0, z_last=15)


@pytest.mark.parametrize("variant", [1, 4, 5])
@pytest.mark.parametrize("shape", [0, 1])
def test_process_corners_dense_list(ctx, variant, shape):
    """Every listed splat is a hit for (nearly) every corner: full tiles, a slot table that overflows within a round (variant 5
    flushes it), several rounds and a jump -- bit-equal to the oracle."""
    fx = dense_fixture()
    got, exp = _run_process_corners(ctx, fx, variant, shape, fx["splats"])
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))     # (a fit without a surface nearby is NaN in both)
    ok = ~np.isnan(exp)
    assert ok.sum() > 3000
    np.testing.assert_array_equal(got[ok].view(np.uint32), exp[ok].view(np.uint32))


@pytest.mark.parametrize("size", [(8, 24, 16), (24, 8, 16), (24, 16, 8), (40, 24, 16)])
def test_process_corners_grid_shapes(ctx, size):
    """Grids one block wide in x or in y (the kernel then finds a block's place by division, not by its exact multiplications)
    and a grid of 5 x 3 x 2 blocks: every block reads the same list; bit-equal to the oracle, variants 5 and 4."""
    rng = np.random.default_rng(13)
    n = 300
    offset = (7, -5, 11)
    splats = np.zeros(n, ob.SPLAT_DTYPE)
    splats["position"] = (rng.uniform(-2.0, 2.0 + max(size), (n, 3)) * (np.array(size) / max(size)) + np.array(offset)).astype(np.float32)
    r = rng.uniform(1.5, 4.0, n)
    splats["radius"] = (1.0 / (r * r)).astype(np.float32)
    nrm = rng.normal(size=(n, 3))
    splats["normal"] = (nrm / np.linalg.norm(nrm, axis=1)[:, None]).astype(np.float32)
    splats["quality"] = rng.uniform(0.5, 2.0, n).astype(np.float32)
    commands = np.array([n + 1] + list(range(n)) + [-1], np.int32)
    start = np.zeros(512, np.int32)                          # every code of an 8 x 8 x 8 grid of blocks: the one list
    fx = dict(offset=offset, splats=splats, commands=commands, start=start, subsampling=3, size=size,
              image_w=size[0], rows=size[1] * size[2], z_stride=size[1], z_bias=0, z_first=0, z_last=size[2] - 1)
    for variant in (5, 4):
        got, exp = _run_process_corners(ctx, fx, variant, 0, splats)
        np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
        ok = ~np.isnan(exp)
        assert ok.sum() > 100
        np.testing.assert_array_equal(got[ok].view(np.uint32), exp[ok].view(np.uint32))


def test_enqueue_checks(ctx):
    import mlsgpu_amd as m
    fx = process_corners_fixture()
    gen = m.MlsFunctor(ctx)
    field = m.DeviceBuffer(ctx, nbytes=fx["rows"] * fx["image_w"] * 4)
    sw = m.Swathe(fx["size"][0], fx["size"][1], fx["z_stride"], fx["z_bias"], fx["z_first"], fx["z_last"])
    with pytest.raises(m.InvalidArgument):
        gen.enqueue(field, fx["image_w"], fx["rows"], sw)        # set() not called
    d = m.DeviceBuffer(ctx, array=fx["splats"])
    c = m.DeviceBuffer(ctx, array=fx["commands"])
    s = m.DeviceBuffer(ctx, array=fx["start"])
    gen.set_buffers(fx["offset"], d, c, s, fx["subsampling"])
    bad = m.Swathe(fx["size"][0], fx["size"][1], fx["z_stride"], fx["z_bias"], 4, fx["z_last"])
    with pytest.raises(m.InvalidArgument):
        gen.enqueue(field, fx["image_w"], fx["rows"], bad)       # zFirst % wgs[2] != 0, src/mls.cpp:112
    with pytest.raises(m.LengthError):
        gen.enqueue(field, 16, fx["rows"], sw)                    # image narrower than roundUp(width, 8)
    with pytest.raises(m.LengthError):
        gen.enqueue(field, fx["image_w"], 10, sw)                 # image too short


def test_prefilter_counters_on_a_bucket(ctx):
    """The instrumented kernels on a whole bucket (cfg1: 50 k splats on a sphere, 64^3 grid): the matrix prefilter hands the
    drain every hit of the reference's test (word 42 = 0) and few candidates more; listed splats and hits are the same numbers
    whichever kernel counts them; the field is the same bits."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud, g = synth.make_cloud("cfg1")
    seen = {}
    for variant in (5, 4, 1):
        w = m.Worker(ctx, len(cloud), max_cells=63)
        w.set_mls_variant(variant)
        counters = m.DeviceBuffer(ctx, array=np.zeros(m.binding.MLS_STATS_WORDS, np.uint64))
        w.set_mls_stats(counters)
        buf = m.DeviceBuffer(ctx, array=cloud)
        col = w.process(buf, 0, len(cloud), (0, 0, 0), (g, g, g), collector=m.binding.ChecksumCollector(ctx))
        ctx.synchronize()
        c = [int(x) for x in counters.download(np.uint64)]
        w.set_mls_stats(None)
        seen[variant] = (c, col.digest())
        del w, buf
    c5, c4, c1 = seen[5][0], seen[4][0], seen[1][0]
    assert c5[0] == c4[0] == c1[0] > 0 and c5[2] == c4[2] == c1[2] > 0            # listed, hits
    assert c5[42] == 0 and c5[2] <= c5[41] <= 1.05 * c5[2]                         # nothing missed, < 5 % false candidates
    assert c5[1] % 2048 == 0 and c1[1] >= c4[1] >= c5[2]                           # 32 x 64 pairs per tile; tests executed
    assert seen[5][1] == seen[4][1] == seen[1][1]
