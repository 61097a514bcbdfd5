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


@pytest.mark.parametrize("variant", [1, 4])
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


@pytest.mark.parametrize("variant", [1, 4])
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
