"""Pins the oracle's octree restatement (test/test_splat_tree.cpp, test/test_splat_tree_cl.cpp)."""
import numpy as np

import oracle_binding as ob
from refdata import MT19937, make_code, make_splats, walk


def test_level_shift():
    # test/test_splat_tree_cl.cpp:171-183
    cases = [(0, (0, 0, 0), (0, 0, 0)), (0, (1, 1, 1), (0, 0, 0)), (0, (0, 1, 2), (1, 2, 3)),
             (1, (0, 1, 2), (2, 2, 3)), (1, (0, 1, 2), (1, 3, 3)), (1, (0, 1, 2), (1, 2, 4)),
             (2, (31, 0, 0), (35, 0, 0)), (3, (31, 0, 0), (36, 0, 0)), (3, (27, 0, 0), (32, 0, 0)),
             (5, (48, 0, 0), (79, 0, 0))]
    for exp, lo, hi in cases:
        assert ob.lib().orc_level_shift(ob._p(np.array(lo, np.int32)), ob._p(np.array(hi, np.int32))) == exp


def test_point_box_dist2():
    # test/test_splat_tree_cl.cpp:185-196
    cases = [(0.0, (0.5, 0.5, 0.5), (0, 0, 0), (1, 1, 1)),
             (4.0, (0.25, 0.5, 3.0), (-1.5, 0.0, 0.5), (1.5, 0.75, 1.0)),
             (14.0, (9.0, 11.0, -10.0), (-1.0, 0.0, -7.0), (8.0, 9.0, 8.0))]
    for exp, p, lo, hi in cases:
        got = ob.lib().orc_point_box_dist2(*[ob._p(np.array(v, np.float32)) for v in (p, lo, hi)])
        assert abs(got - exp) < 1e-4


BUILD_SPLATS = [
    # test/test_splat_tree.cpp:78-87
    (10.5, 7.5, 8.5, 1.0), (11.5, 9.5, 5.5, 2.0), (10.0, 10.0, 6.0, 4.5), (3.0, 0.5, 1.0, 0.75),
    (3.0, 0.0, 1.0, 0.5), (19.0, 8.0, 5.0, 0.6), (0.0, 1.0, 1.0, 2.5), (5.0, 1000.0, 5.0, 900.0),
]


def check_build(splats_in, commands, start, num_levels):
    """Assertions of TestSplatTree::testBuild, test/test_splat_tree.cpp:89-170."""
    offset = (3, 0, 1)
    assert num_levels >= 5
    assert len(start) >= 16 * 16 * 16
    pos_ = splats_in["position"].astype(np.float32)
    rad = splats_in["radius"].astype(np.float32)
    max_pos = -1
    for z in range(12):
        for y in range(16):
            for x in range(16):
                idx = make_code(x, y, z)
                pos = int(start[idx])
                max_pos = max(max_pos, pos)
                found = []
                if pos != -1:
                    found = walk(commands, pos, limit=1000)
                    assert all(0 <= c < len(splats_in) for c in found)
                    assert len(set(found)) == len(found), "splat visited twice in one walk"
                    # track the highest command used, as the reference does
                    p = pos
                    while p >= 0:
                        end = int(commands[p])
                        max_pos = max(max_pos, end)
                        p = int(commands[end])
                corner = np.array([x + offset[0], y + offset[1], z + offset[2]], np.float32)
                n = np.maximum(np.minimum(pos_, corner + np.float32(1.0)), corner) - pos_
                dist2 = (n * n).sum(axis=1, dtype=np.float32)
                must = np.nonzero(dist2 <= rad * rad)[0]
                for i in must:
                    assert int(i) in found, (x, y, z, int(i))
    repeats = {}
    i = 0
    while i <= max_pos:
        end = int(commands[i])
        i += 1
        while i < end:
            cmd = int(commands[i])
            i += 1
            repeats[cmd] = repeats.get(cmd, 0) + 1
            assert repeats[cmd] <= 8
        i += 1   # the reference's for-loop increment steps over the jump slot


def test_build():
    splats = make_splats(BUILD_SPLATS)
    orig = splats.copy()
    t = ob.Tree(splats, 0, len(splats), (16, 16, 12), (3, 0, 1), 0, 9)
    check_build(orig, t.commands, t.start, t.num_levels)
    # writeEntries replaces the radius by 1/r^2, kernels/octree.cl:193
    np.testing.assert_array_equal(splats["radius"], np.float32(1.0) / (orig["radius"] * orig["radius"]))


def random_splats():
    """TestSplatTree::testRandom inputs, test/test_splat_tree.cpp:172-199."""
    eng = MT19937()
    cells = (31, 31, 16)
    rows = []
    for _ in range(207):
        rows.append((eng.uniform_real(-2.0, cells[0] + 2.0, single=True),
                     eng.uniform_real(-2.0, cells[1] + 2.0, single=True),
                     eng.uniform_real(-2.0, cells[2] + 2.0, single=True),
                     eng.uniform_real(0.25, 8.0, single=True)))
    return make_splats(rows), cells


def check_random(commands, start, nsplats, cells, subsampling):
    """Assertions of test/test_splat_tree.cpp:206-243."""
    for z in range(0, cells[2] + 1, 1 << subsampling):
        for y in range(0, cells[1] + 1, 1 << subsampling):
            for x in range(0, cells[0] + 1, 1 << subsampling):
                idx = make_code(x >> subsampling, y >> subsampling, z >> subsampling)
                assert idx < len(start)
                if start[idx] != -1:
                    assert 0 <= start[idx] < len(commands)
                    ids = walk(commands, start[idx], limit=len(commands) + 1)
                    assert all(0 <= c < nsplats for c in ids)


def test_random():
    splats, cells = random_splats()
    t = ob.Tree(splats, 0, len(splats), cells, (1, 2, -1), 2, 8)
    check_random(t.commands, t.start, len(splats), cells, 2)


def test_first_splat_and_order():
    """firstSplat offsets the ids (kernels/octree.cl:185,210) and ids ascend within a node (stable sort)."""
    splats, cells = random_splats()
    padded = np.concatenate([make_splats([(1e6, 1e6, 1e6, 1.0)] * 5), splats])
    t0 = ob.Tree(splats.copy(), 0, len(splats), cells, (1, 2, -1), 2, 8)
    t1 = ob.Tree(padded, 5, len(splats), cells, (1, 2, -1), 2, 8)
    np.testing.assert_array_equal(t0.start, t1.start)
    n = t0.num_commands
    assert t1.num_commands == n
    pos = 0
    while pos < n:
        end = int(t0.commands[pos])
        ids0 = t0.commands[pos + 1:end]
        ids1 = t1.commands[pos + 1:end]
        np.testing.assert_array_equal(ids0 + 5, ids1)
        assert np.all(np.diff(ids0) > 0)
        assert t0.commands[end] == t1.commands[end]
        pos = end + 1
    # untouched splats keep their radius
    assert np.all(padded["radius"][:5] == 1.0)


def test_size_check():
    import pytest
    splats = make_splats(BUILD_SPLATS)
    with pytest.raises(ValueError):
        ob.Tree(splats, 0, len(splats), (600, 16, 16), (0, 0, 0), 0, 9)   # > 2^(levels+sub-1)
