"""Pins the oracle's MLS restatement with the reference's known answers (test/test_mls.cpp)."""
import math

import numpy as np
import pytest

import oracle_binding as ob
from refdata import sphere_splats

EPS = float(np.finfo(np.float32).eps)


def f32(x):
    return float(np.float32(x))


def close(expected, actual, eps):
    """MLSGPU_ASSERT_DOUBLES_EQUAL (test/testutil.h): NaN == NaN, otherwise abs or relative."""
    if math.isnan(expected):
        return math.isnan(actual)
    if math.isnan(actual):
        return False
    return abs(expected - actual) <= eps or abs(expected - actual) <= eps * abs(expected)


def test_make_code():
    L = ob.lib()
    # test/test_mls.cpp:255-261, test/test_splat_tree_cl.cpp:198-204
    assert L.orc_make_code(0, 0, 0) == 0
    assert L.orc_make_code(1, 1, 1) == 7
    assert L.orc_make_code(2, 5, 3) == 174
    assert L.orc_make_code(7, 7, 7) == 511
    # test/test_splat_tree.cpp:46-54
    assert L.orc_make_code(123, 456, 789) == 642569997


def test_decode():
    # test/test_mls.cpp:263-285
    out = np.zeros(3, np.int32)
    for code, exp in [(0xB1, (1, 6, 2)), (0xAAAAAAAA, (0x2AA, 0x555, 0x2AA)),
                      (0x24924924, (0, 0, 0x3FF)), (0, (0, 0, 0))]:
        ob.lib().orc_decode(code, ob._p(out))
        assert tuple(out) == exp


SOLVE_CASES = [
    # test/test_mls.cpp:287-329 (expected, a, b, c)
    (math.nan, -1, 2, -2), (math.nan, -1e20, 2e10, -1.0001), (math.nan, 1, 0, 1), (math.nan, -1, 0, -1),
    (math.nan, 0, 0, 0), (math.nan, 0, 0, 4), (math.nan, 0, 0, -3), (math.nan, 0, 0, -1e20), (math.nan, 0, 0, 1e20),
    (-1.5, 0, 2, 3), (0.0, 0, 5, 0), (0.0, 0, 1e20, 0), (0.0, 0, 1e-20, 0), (1e-20, 0, 1e10, 1e-10),
    (-1e20, 0, 1e-10, 1e10),
    (1.0, -1, 2, -1), (1.0, -10, 20, -10), (1e4, -1, 2e4, -1e8), (0.0, 1, 0, 0), (0.0, 1e30, 0, 0), (0.0, 1e-20, 0, 0),
    (2.0, -1, 5, -6), (2.0, -2, 10, -12), (2.0, 1, 1, -6), (2.0, f32(0.1), f32(0.1), f32(-0.6)),
    (2.0, -1e-12, 5e-12, -6e-12), (-2e-12, 1, 5e-12, 6e-24),
    (1e-6, -1, 1 + 1e-6, -1e-6), (1.0, -1, 1 + 1e6, -1e6), (1e20, -1e-20, 2, -1e20), (-1e-6, 1e-6, 1, 1e-6),
]


@pytest.mark.parametrize("expected,a,b,c", SOLVE_CASES)
def test_solve_quadratic(expected, a, b, c):
    got = ob.lib().orc_solve_quadratic(a, b, c)
    # the reference's expectations are doubles compared at 4 eps; mixed abs/rel as its macro does
    assert close(expected, got, 4 * EPS), (expected, got)


def make_sphere(xc, yc, zc, r, grad):
    # test/test_mls.cpp:151-166 (float arithmetic)
    xc, yc, zc, r, grad = map(np.float32, (xc, yc, zc, r, grad))
    scale = grad * np.float32(0.5) / r
    return [np.float32(-2.0) * xc * scale, np.float32(-2.0) * yc * scale, np.float32(-2.0) * zc * scale, scale,
            (xc * xc + yc * yc + zc * zc - r * r) * scale]


def make_plane(px, py, pz, dx, dy, dz):
    # test/test_mls.cpp:168-176
    px, py, pz, dx, dy, dz = map(np.float32, (px, py, pz, dx, dy, dz))
    return [dx, dy, dz, np.float32(0.0), -(dx * px + dy * py + dz * pz)]


def test_project_dist_origin_sphere():
    # test/test_mls.cpp:331-347
    cases = [
        (7.0, make_sphere(3, 4, 12, 6, 1)), (7.0, make_sphere(3, 4, 12, 6, 2.5)),
        (-7.0, make_sphere(3, 4, 12, 6, -2.5)), (0.0, make_sphere(3, 4, 12, 13, 2.5)),
        (-5.0, make_sphere(3, 4, 12, 18, 2.5)), (-6.0, make_sphere(0, 0, 0, 6, 2.5)),
        (5.0, make_sphere(0, 0, 0, 5, -1.5)),
        (f32(-5.0 / 1.5), make_plane(1, 2, 3, 1, 0.5, 1)), (f32(5.0 / 1.5), make_plane(-1, -2, -3, 1, 0.5, 1)),
    ]
    for expected, p in cases:
        got = ob.lib().orc_project_dist_origin_sphere(*[float(x) for x in p])
        assert close(expected, got, 4 * EPS), (expected, got)


def test_fit_sphere():
    # test/test_mls.cpp:382-409
    n = 20
    splats = sphere_splats(n, (1.0, 2.0, 3.5), 6.5)
    params = np.zeros(5, np.float32)
    ob.lib().orc_fit_sphere(ob._p(splats), n, ob._p(params))
    eps = 16 * EPS
    p = [float(x) for x in params]
    for i in range(n):
        x, y, z = (float(v) for v in splats["position"][i])
        v = p[0] * x + p[1] * y + p[2] * z + p[3] * (x * x + y * y + z * z) + p[4]
        assert close(0.0, v, eps)
        g = [f32(2 * p[3] * x + p[0]), f32(2 * p[3] * y + p[1]), f32(2 * p[3] * z + p[2])]
        for k in range(3):
            assert close(float(splats["normal"][i][k]), g[k], eps)


def process_corners_fixture():
    """Inputs of TestMls::testProcessCorners, test/test_mls.cpp:416-473."""
    n = 50
    center = (10.0, 20.0, 35.0)
    radius = 65.0
    size = (19, 24, 28)
    offset = (20, 15, 33)
    wgs = 8
    image_w = (size[0] + wgs - 1) // wgs * wgs
    image_h = (size[1] + wgs - 1) // wgs * wgs
    image_d = (size[2] + wgs - 1) // wgs * wgs
    z_first, z_last = wgs, 26
    z_stride = image_h + 10
    z_bias = (2 - z_first) * z_stride
    subsampling = 3
    while (2 << subsampling) < max(size):
        subsampling += 1
    splats = sphere_splats(n, center, radius)
    splats["radius"] = np.float32(1.0) / (splats["radius"] * splats["radius"])
    start = [0] * 8
    commands = [n - 1] + list(range(n - 2))
    commands.append(-2 - len(commands))
    start[6] = -1
    start[7] = len(commands)
    commands.append(len(commands) + 3)
    commands += [n - 2, n - 1, -1]
    rows = image_d * z_stride + z_bias
    return dict(n=n, center=center, radius=radius, size=size, offset=offset, image_w=image_w, rows=rows,
                z_first=z_first, z_last=z_last, z_stride=z_stride, z_bias=z_bias, subsampling=subsampling,
                splats=splats, start=np.array(start, np.int32), commands=np.array(commands, np.int32))


def check_process_corners(fx, field):
    """Expectations of test/test_mls.cpp:486-513."""
    sub = fx["subsampling"]
    bad = 0
    for z in range(fx["z_first"], fx["z_last"] + 1):
        for y in range(fx["size"][1]):
            for x in range(fx["size"][0]):
                cx, cy, cz = (f32(x + fx["offset"][0]), f32(y + fx["offset"][1]), f32(z + fx["offset"][2]))
                c = fx["center"]
                expected = f32(math.sqrt((cx - c[0]) ** 2 + (cy - c[1]) ** 2 + (cz - c[2]) ** 2) - fx["radius"])
                if (z >> sub) == 1 and (y >> sub) == 1:
                    expected = math.nan
                if abs(expected) > f32(math.sqrt(3.0)):
                    expected = math.nan
                actual = float(field[y + z * fx["z_stride"] + fx["z_bias"], x])
                if not close(expected, actual, 1e-5):
                    bad += 1
    return bad


def test_process_corners():
    fx = process_corners_fixture()
    assert fx["subsampling"] == 4
    field = np.full((fx["rows"], fx["image_w"]), -12345.0, np.float32)
    bf = ob.lib().orc_boundary_factor(1.0)
    ob.process_corners(field, fx["splats"], fx["commands"], fx["start"], fx["subsampling"], fx["offset"],
                       fx["size"][0], fx["size"][1], fx["z_stride"], fx["z_bias"], fx["z_first"], fx["z_last"], bf)
    assert check_process_corners(fx, field) == 0
    # slices below zFirst are unaffected (Generator::enqueue post-condition, src/marching.h:246-249)
    first_row = fx["z_first"] * fx["z_stride"] + fx["z_bias"]
    assert np.all(field[:first_row] == -12345.0)


def test_boundary_factor():
    # src/mls.cpp:137-144
    bs = math.sqrt(6.0) * 512 / (693 * math.pi)
    assert abs(ob.lib().orc_boundary_factor(1.0) - (1 - bs * bs)) < 1e-6
    assert ob.lib().orc_boundary_factor(0.0) == 1.0


def test_half_rsqrt_substitution_moves_vertices_not_topology():
    """The one arithmetic substitution whose size the reference leaves open: `half_rsqrt` (kernels/mls.cl:406, error
    implementation-defined) is an exactly rounded 1/sqrt in the oracle and the HIP path.  It scales |f| and never its
    sign, so which cells are occupied, every triangle and every vertex key are unchanged; vertices move along their grid
    edges.  Measured here with the result cut to 11 mantissa bits (a half-precision reciprocal square root): identical
    topology, vertices within 1e-3 cells -- the "stated float tolerance + identical topology" of the north star."""
    from mlsgpu_amd import synth
    cloud, g = synth.make_cloud("cfg1")
    kw = dict(max_cells=63, max_swathe=64, mesh_memory=63 * 63 * 2 * 872)
    exact, _ = ob.bucket(cloud.copy(), 0, len(cloud), (g, g, g), (0, 0, 0), **kw)
    try:
        ob.lib().orc_set_rsqrt_bits(11)
        rough, _ = ob.bucket(cloud.copy(), 0, len(cloud), (g, g, g), (0, 0, 0), **kw)
    finally:
        ob.lib().orc_set_rsqrt_bits(0)
    assert len(exact) == len(rough) >= 1
    worst = 0.0
    for a, b in zip(exact, rough):
        assert a["num_internal"] == b["num_internal"]
        np.testing.assert_array_equal(a["triangles"], b["triangles"])
        np.testing.assert_array_equal(a["keys"], b["keys"])
        worst = max(worst, float(np.abs(a["vertices"] - b["vertices"]).max()))
    assert 0.0 < worst < 1e-3, worst
