"""SplatTreeCL on the GPU: the reference's own assertions (test/test_splat_tree*.cpp) and bit parity with the oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from gpu_common import ctx  # noqa: F401
from refdata import make_splats
from test_oracle_tree import BUILD_SPLATS, check_build, check_random, random_splats

pytestmark = pytest.mark.gpu


def test_make_code(ctx):
    import mlsgpu_amd as m
    out = C.c_uint32()
    for xyz, exp in [((0, 0, 0), 0), ((1, 1, 1), 7), ((2, 5, 3), 174), ((7, 7, 7), 511), ((123, 456, 789), 642569997)]:
        m.binding.check(m.lib().mlsgpu_hip_test_make_code(ctx.h, *xyz, C.byref(out)))
        assert out.value == exp


def test_level_shift(ctx):
    import mlsgpu_amd as m
    cases = [(0, (0, 0, 0), (0, 0, 0)), (0, (1, 1, 1), (0, 0, 0)), (0, (0, 1, 2), (1, 2, 3)),
             (1, (0, 1, 2), (2, 2, 3)), (1, (0, 1, 2), (1, 3, 3)), (1, (0, 1, 2), (1, 2, 4)),
             (2, (31, 0, 0), (35, 0, 0)), (3, (31, 0, 0), (36, 0, 0)), (3, (27, 0, 0), (32, 0, 0)),
             (5, (48, 0, 0), (79, 0, 0))]
    out = C.c_int32()
    for exp, lo, hi in cases:
        m.binding.check(m.lib().mlsgpu_hip_test_level_shift(ctx.h, ob._p(np.array(lo, np.int32)),
                                                             ob._p(np.array(hi, np.int32)), C.byref(out)))
        assert out.value == exp


def test_point_box_dist2(ctx):
    import mlsgpu_amd as m
    cases = [(0.0, (0.5, 0.5, 0.5), (0, 0, 0), (1, 1, 1)),
             (4.0, (0.25, 0.5, 3.0), (-1.5, 0.0, 0.5), (1.5, 0.75, 1.0)),
             (14.0, (9.0, 11.0, -10.0), (-1.0, 0.0, -7.0), (8.0, 9.0, 8.0))]
    out = C.c_float()
    for exp, p, lo, hi in cases:
        arrs = [np.array(v, np.float32) for v in (p, lo, hi)]
        m.binding.check(m.lib().mlsgpu_hip_test_point_box_dist2(ctx.h, *[ob._p(a) for a in arrs], C.byref(out)))
        assert abs(out.value - exp) < 1e-4


def gpu_build(ctx, splats, first, num, size, offset, subsampling, levels, max_splats=None):
    import mlsgpu_amd as m
    tree = m.SplatTree(ctx, levels, max_splats or max(len(splats), 1))
    buf = m.DeviceBuffer(ctx, array=splats)
    tree.enqueue_build(buf, first, num, size, offset, subsampling)
    ctx.synchronize()
    return tree.commands(), tree.start(), tree.num_levels, buf.download(m.SPLAT_DTYPE, len(splats))


def compare_with_oracle(commands, start, mutated, splats, first, num, size, offset, subsampling, levels):
    s2 = splats.copy()
    t = ob.Tree(s2, first, num, size, offset, subsampling, levels)
    np.testing.assert_array_equal(start[:t.num_start], t.start[:t.num_start])
    np.testing.assert_array_equal(commands[:t.num_commands], t.commands[:t.num_commands])
    np.testing.assert_array_equal(mutated.view(np.uint32), s2.view(np.uint32))


def test_build_without_mutation(ctx):
    """mlsgpu_hip_tree_set_mutate(0): the same commands / start, and the splats are left as they came."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud, g = synth.make_cloud("cfg1", scale=0.5)
    tree = m.SplatTree(ctx, 6, len(cloud))
    tree.set_mutate(False)
    buf = m.DeviceBuffer(ctx, array=cloud)
    tree.enqueue_build(buf, 0, len(cloud), (64, 64, 64), (0, 0, 0), 3)
    ctx.synchronize()
    t = ob.Tree(cloud.copy(), 0, len(cloud), (64, 64, 64), (0, 0, 0), 3, 6)
    np.testing.assert_array_equal(tree.start()[:t.num_start], t.start[:t.num_start])
    np.testing.assert_array_equal(tree.commands()[:t.num_commands], t.commands[:t.num_commands])
    np.testing.assert_array_equal(buf.download(m.SPLAT_DTYPE, len(cloud)).view(np.uint32), cloud.view(np.uint32))


def test_build(ctx):
    """TestSplatTree::testBuild on the device result."""
    splats = make_splats(BUILD_SPLATS)
    commands, start, num_levels, mutated = gpu_build(ctx, splats, 0, len(splats), (16, 16, 12), (3, 0, 1), 0, 9,
                                                     max_splats=1001)
    check_build(splats, commands, start, num_levels)
    compare_with_oracle(commands, start, mutated, splats, 0, len(splats), (16, 16, 12), (3, 0, 1), 0, 9)


def test_random(ctx):
    """TestSplatTree::testRandom on the device result."""
    splats, cells = random_splats()
    commands, start, _, mutated = gpu_build(ctx, splats, 0, len(splats), cells, (1, 2, -1), 2, 8, max_splats=1000)
    check_random(commands, start, len(splats), cells, 2)
    compare_with_oracle(commands, start, mutated, splats, 0, len(splats), cells, (1, 2, -1), 2, 8)


@pytest.mark.parametrize("n,first", [(0, 0), (1, 0), (5000, 0), (200_000, 1234)])
def test_parity_default_geometry(ctx, n, first):
    """levels 6 / subsampling 3 (the reference defaults) on a 256-corner bucket, ragged sizes, firstSplat > 0."""
    from mlsgpu_amd import synth
    cloud = synth.uniform_cloud(n + first, 255.0, 0.5, 9.0, seed=77)   # radii span several octree levels
    cloud["position"] += np.float32(40.0)                              # bucket offset below
    commands, start, _, mutated = gpu_build(ctx, cloud, first, n, (256, 256, 256), (40, 40, 40), 3, 6)
    compare_with_oracle(commands, start, mutated, cloud, first, n, (256, 256, 256), (40, 40, 40), 3, 6)


def test_argument_checks(ctx):
    import mlsgpu_amd as m
    with pytest.raises(m.LengthError):
        m.SplatTree(ctx, 11, 100)                   # MAX_LEVELS
    with pytest.raises(m.LengthError):
        m.SplatTree(ctx, 6, 0)
    tree = m.SplatTree(ctx, 6, 100)
    buf = m.DeviceBuffer(ctx, array=make_splats(BUILD_SPLATS))
    with pytest.raises(m.LengthError):
        tree.enqueue_build(buf, 0, 101, (8, 8, 8), (0, 0, 0), 3)        # numSplats > maxSplats
    with pytest.raises(m.LengthError):
        tree.enqueue_build(buf, 0, 8, (257, 8, 8), (0, 0, 0), 3)        # size > 2^(levels+sub-1)


@pytest.mark.parametrize("levels", [1, 2, 3, 4, 5, 6, 7, 8])
def test_every_tree_depth(ctx, levels):
    """Every route of the build: one-digit keys whose only pass is the fused one (levels <= 3: the last pass has nothing left
    to sort, only the command positions to hand out), two digits (the default: whole-key counts in the last pass's histogram
    kernel, a scan over the nodes, ids scattered to their command positions), and deeper trees that keep the scan over the
    entries.  `start` / `commands` equal the oracle's word for word."""
    from mlsgpu_amd import synth
    side = min(1 << (levels + 2), 256)
    cloud = synth.uniform_cloud(30_000, float(side - 1), 0.5, 6.0, seed=1000 + levels)
    size = (side, side - 5, side - 8) if side > 8 else (side, side, side)
    commands, start, num_levels, mutated = gpu_build(ctx, cloud, 0, len(cloud), size, (2, 0, 1), 3, levels)
    compare_with_oracle(commands, start, mutated, cloud, 0, len(cloud), size, (2, 0, 1), 3, levels)


@pytest.mark.parametrize("levels", [2, 6])
def test_two_words_per_entry_route(ctx, levels, monkeypatch):
    """The default route keeps ONE word per entry between the two passes of the entry sort (the low digit is the entry's
    position, the rest of the key sits above the splat's number inside the bucket); MLSGPU_HIP_OCTREE_PACKED=0 is the
    two-word form it replaced -- and what a bucket whose key and splat number do not fit a word together falls back to."""
    from mlsgpu_amd import synth
    monkeypatch.setenv("MLSGPU_HIP_OCTREE_PACKED", "0")
    side = min(1 << (levels + 2), 256)
    cloud = synth.uniform_cloud(40_000 + 77, float(side - 1), 0.5, 6.0, seed=2000 + levels)
    size = (side, side - 5, side - 8) if side > 8 else (side, side, side)
    commands, start, _, mutated = gpu_build(ctx, cloud, 77, 40_000, size, (2, 0, 1), 3, levels)
    compare_with_oracle(commands, start, mutated, cloud, 77, 40_000, size, (2, 0, 1), 3, levels)
