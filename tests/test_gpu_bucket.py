"""Whole buckets through DeviceWorkerGroupBase::Worker's restatement: tree -> MLS -> marching -> scale/bias.

Small sizes: bit parity with the oracle.  BASELINE sizes: size-independent properties.
"""
import numpy as np
import pytest

import oracle_binding as ob
from gpu_common import assert_batches_equal, ctx  # noqa: F401
from refdata import is_manifold, weld_batches

pytestmark = pytest.mark.gpu


def run_gpu_bucket(ctx, cloud, first, count, low, nv, variant=5, **kw):
    import mlsgpu_amd as m
    w = m.Worker(ctx, max(count, 1), **kw)
    w.set_mls_variant(variant)
    buf = m.DeviceBuffer(ctx, array=cloud)
    batches = w.process(buf, first, count, low, nv)
    return batches, w, buf


@pytest.mark.parametrize("variant", [1, 4, 5])
def test_cfg1_parity(ctx, variant):
    """BASELINE config 0: 64^3 grid, 50k splats on a sphere, one bucket: bit-identical to the oracle."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud, g = synth.make_cloud("cfg1")
    got, w, buf = run_gpu_bucket(ctx, cloud, 0, len(cloud), (0, 0, 0), (g, g, g), variant=variant, max_cells=63)
    s2 = cloud.copy()
    exp, st = ob.bucket(s2, 0, len(cloud), (g, g, g), (0, 0, 0), max_cells=63, max_swathe=64,
                        mesh_memory=63 * 63 * 2 * 872)
    assert_batches_equal(got, exp)
    cnt = w.marching_counters()
    for k in ("shipouts", "occupied", "unwelded", "indices", "welded", "external"):
        assert cnt[k] == st[k], k
    # the octree the worker built equals the oracle's
    commands, start = w.tree_arrays()
    t = ob.Tree(cloud.copy(), 0, len(cloud), (64, 64, 64), (0, 0, 0), 3, 6)
    np.testing.assert_array_equal(start[:t.num_start], t.start[:t.num_start])
    np.testing.assert_array_equal(commands[:t.num_commands], t.commands[:t.num_commands])
    # splat.w was replaced by 1/r^2 on the device exactly as on the host
    np.testing.assert_array_equal(buf.download(m.SPLAT_DTYPE, len(cloud)).view(np.uint32), s2.view(np.uint32))
    v, tr, _ = weld_batches(got)
    assert is_manifold(len(v), tr) == ""


@pytest.mark.parametrize("variant", [1, 4, 5])
def test_keep_splats(ctx, variant):
    """mlsgpu_hip_worker_set_keep_splats (non-mutating tree build + processCorners taking 1/r^2 while it stages a splat,
    in every kernel variant): the bucket's mesh is bit-identical to the oracle's (whose tree mutates its splats), the
    device splats are untouched, and a second pass over the SAME buffer gives the same batches."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud, g = synth.make_cloud("cfg1")
    w = m.Worker(ctx, len(cloud), max_cells=63)
    w.set_mls_variant(variant)
    w.set_keep_splats(True)
    buf = m.DeviceBuffer(ctx, array=cloud)
    got = w.process(buf, 0, len(cloud), (0, 0, 0), (g, g, g))
    exp, _ = ob.bucket(cloud.copy(), 0, len(cloud), (g, g, g), (0, 0, 0), max_cells=63, max_swathe=64,
                       mesh_memory=63 * 63 * 2 * 872)
    assert_batches_equal(got, exp)
    np.testing.assert_array_equal(buf.download(m.SPLAT_DTYPE, len(cloud)).view(np.uint32), cloud.view(np.uint32))
    again = w.process(buf, 0, len(cloud), (0, 0, 0), (g, g, g))
    assert_batches_equal(again, exp)
    # and back: the reference's behaviour
    w.set_keep_splats(False)
    third = w.process(buf, 0, len(cloud), (0, 0, 0), (g, g, g))
    assert_batches_equal(third, exp)
    assert not np.array_equal(buf.download(m.SPLAT_DTYPE, len(cloud))["radius"], cloud["radius"])


def test_offset_bucket_scale_bias_and_small_mesh_memory(ctx):
    """A bucket away from the origin, ragged size, tiny mesh memory (several ship-outs), scale/bias applied."""
    from mlsgpu_amd import synth
    cloud = synth.sphere_cloud(30_000, (70.0, 61.0, 52.0), 17.0, 1.0, 2.5, seed=99)
    low, nv = (45, 37, 30), (51, 46, 44)
    mm = 50 * 50 * 872
    got, w, _ = run_gpu_bucket(ctx, cloud, 0, len(cloud), low, nv, max_cells=50, mesh_memory=mm, max_swathe=8,
                               grid_spacing=0.25, grid_origin=(-3.0, 4.0, 0.5))
    s2 = cloud.copy()
    exp, st = ob.bucket(s2, 0, len(cloud), nv, low, max_cells=50, max_swathe=8, mesh_memory=mm)
    assert st["shipouts"] > 1
    for b in exp:
        ob.lib().orc_scale_bias(ob._p(b["vertices"]), len(b["vertices"]), 0.25, -3.0, 4.0, 0.5)
    assert_batches_equal(got, exp)


def test_plane_shape_and_boundary_limit(ctx):
    from mlsgpu_amd import synth
    cloud = synth.sphere_cloud(20_000, (30.0, 30.0, 30.0), 40.0, 1.5, 2.5, seed=5)   # an open cap: boundary test matters
    got, _, _ = run_gpu_bucket(ctx, cloud, 0, len(cloud), (0, 0, 0), (64, 64, 64), max_cells=63, shape=1,
                               boundary_limit=1.5)
    exp, _ = ob.bucket(cloud.copy(), 0, len(cloud), (64, 64, 64), (0, 0, 0), max_cells=63, max_swathe=64,
                       mesh_memory=63 * 63 * 2 * 872, shape=1, boundary_limit=1.5)
    assert_batches_equal(got, exp)
    assert sum(len(b["triangles"]) for b in got) > 0


@pytest.mark.parametrize("variant", [1, 4, 5])
def test_dense_hits(ctx, variant):
    """Large, dense splats: hundreds of hits per corner.  For the default kernel (4) that means cube lists longer than one
    32-iteration chunk (stale list entries masked by the lane's own count), full windows carried over through the
    wave's slot table, and several 512-splat staging rounds per block; the accumulation order, hence every bit of the
    result, must not change."""
    from mlsgpu_amd import synth
    cloud = synth.sphere_cloud(60_000, (32.0, 32.0, 32.0), 20.0, 5.0, 7.0, seed=4242)
    got, _, _ = run_gpu_bucket(ctx, cloud, 0, len(cloud), (0, 0, 0), (64, 64, 64), variant=variant, max_cells=63)
    exp, st = ob.bucket(cloud.copy(), 0, len(cloud), (64, 64, 64), (0, 0, 0), max_cells=63, max_swathe=64,
                        mesh_memory=63 * 63 * 2 * 872)
    assert st["hits"] > 100 * 64 ** 3
    assert_batches_equal(got, exp)


def test_empty_bucket(ctx):
    from mlsgpu_amd import synth
    cloud = synth.sphere_cloud(10, (500.0, 500.0, 500.0), 3.0, 1.0, 2.0, seed=1)    # all far away
    got, _, _ = run_gpu_bucket(ctx, cloud, 0, len(cloud), (0, 0, 0), (64, 64, 64), max_cells=63)
    assert got == []
    got, _, _ = run_gpu_bucket(ctx, cloud, 0, 0, (0, 0, 0), (64, 64, 64), max_cells=63)   # zero splats
    assert got == []


def test_multi_bucket_weld_across_faces(ctx):
    """Config-2-shaped data at reduced count, cut into 27 buckets: every bucket equals the oracle, vertices
    on shared faces are bit-identical from both sides (the host key weld asserts it) and the union is manifold."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud = synth.shells_cloud(120_000, 95.0, 16.0, 1.5, 2.5, seed=321)
    allb, buckets = synth.bucketize(cloud, 96, 32)
    assert len(buckets) == 27
    w = m.Worker(ctx, max(b.count for b in buckets), max_cells=63)
    buf = m.DeviceBuffer(ctx, array=allb)
    all_batches = []
    ref = allb.copy()
    for b in buckets:
        got = w.process(buf, b.first, b.count, b.low, b.num_vertices)
        exp, _ = ob.bucket(ref, b.first, b.count, b.num_vertices, b.low, max_cells=63, max_swathe=64,
                           mesh_memory=63 * 63 * 2 * 872)
        assert_batches_equal(got, exp)
        all_batches += got
    v, t, key_map = weld_batches(all_batches)
    assert len(key_map) > 0
    assert is_manifold(len(v), t) == ""


def mesh_digest(batches):
    import hashlib
    h = hashlib.sha256()
    for b in batches:
        h.update(b["vertices"].tobytes())
        h.update(b["triangles"].tobytes())
        h.update(b["keys"][b["num_internal"]:].tobytes())
        h.update(np.uint64(b["num_internal"]).tobytes())
    return h.hexdigest()


def test_cfg2_full_size_properties(ctx):
    """BASELINE config 1 at full size (256^3, 5M uniform-random splats, one bucket): size-independent properties --
    structural validity of every batch, idempotence, and that the three MLS kernels give the same mesh bit for bit.
    (The whole bucket against the oracle: test_gpu_configs.py::test_cfg2_full_size.)"""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud, g = synth.make_cloud("cfg2")
    digests = []
    for variant in (5, 4, 1):
        w = m.Worker(ctx, len(cloud), max_cells=255)
        w.set_mls_variant(variant)
        buf = m.DeviceBuffer(ctx, array=cloud)
        batches = w.process(buf, 0, len(cloud), (0, 0, 0), (g, g, g))
        cnt = w.marching_counters()
        assert cnt["shipouts"] == len(batches) >= 1
        assert sum(len(b["triangles"]) for b in batches) * 3 == cnt["indices"]
        assert sum(len(b["vertices"]) for b in batches) == cnt["welded"] <= cnt["unwelded"]
        for b in batches:
            nv, ni = len(b["vertices"]), b["num_internal"]
            assert ni <= nv and np.all(np.isfinite(b["vertices"]))
            assert b["triangles"].max() < nv
            used = np.zeros(nv, bool)
            used[b["triangles"].ravel()] = True
            assert used.all()                                  # no isolated vertices
            ext = b["keys"][ni:]
            assert len(np.unique(ext)) == len(ext)               # external keys are unique within a batch
            t = b["triangles"]
            assert np.all(t[:, 0] != t[:, 1]) and np.all(t[:, 1] != t[:, 2]) and np.all(t[:, 0] != t[:, 2])
            assert b["vertices"].min() >= 0 and b["vertices"].max() <= g - 1
        digests.append(mesh_digest(batches))
        del w, buf
    assert len(set(digests)) == 1


def test_cfg3_shape_scaled_cross_bucket_properties(ctx):
    """BASELINE config 2's shape (512^3 grid, 27 buckets; "shells" cloud at 20 % of the splat count), with per-kernel
    timing enabled on a context that outlives several workers.  Size-independent properties: the buckets tile
    the grid, every external vertex that two buckets share has one key and bit-identical coordinates from both
    sides, no key occurs more than 8 times, and the union of the meshes has no degenerate triangle."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud, g = synth.make_cloud("cfg3", dist="shells", scale=0.2)
    allb, buckets = synth.bucketize(cloud, g, 255)
    assert len(buckets) == 27 and sum(b.cells for b in buckets) == (g - 1) ** 3
    ctx.set_timing(True)
    w = m.Worker(ctx, max(b.count for b in buckets), max_cells=max(max(b.num_vertices) for b in buckets) - 1,
                 mesh_memory=1 << 30)
    buf = m.DeviceBuffer(ctx, array=allb)
    keys, verts = [], []
    tris = 0
    for b in buckets:
        for batch in w.process(buf, b.first, b.count, b.low, b.num_vertices):
            ni = batch["num_internal"]
            keys.append(batch["keys"][ni:])
            verts.append(batch["vertices"][ni:])
            t = batch["triangles"]
            tris += len(t)
            assert np.all(t[:, 0] != t[:, 1]) and np.all(t[:, 1] != t[:, 2]) and np.all(t[:, 0] != t[:, 2])
            lo = np.array(b.low, np.float32)
            hi = lo + np.array(b.num_vertices, np.float32) - 1
            assert np.all(batch["vertices"] >= lo) and np.all(batch["vertices"] <= hi)
    ctx.set_timing(False)
    stats = ctx.stats()
    assert stats["kernel.mls.processCorners.time"][1] == 27 and stats["kernel.mls.processCorners.time"][0] > 0
    assert tris > 0
    keys = np.concatenate(keys)
    verts = np.concatenate(verts)
    order = np.argsort(keys, kind="stable")
    keys, verts = keys[order], verts[order].view(np.uint32)
    same = keys[1:] == keys[:-1]
    assert same.any()                                                   # buckets do share vertices
    assert np.all(verts[1:][same] == verts[:-1][same])                  # ... and agree on them bit for bit
    _, counts = np.unique(keys, return_counts=True)
    assert counts.max() <= 8


# ---- batches: several buckets through the path in lock-step (mlsgpu_hip_worker_process_batch) ----

def _oracle_bucket(ref, b, **kw):
    args = dict(max_cells=63, max_swathe=64, mesh_memory=63 * 63 * 2 * 872)
    args.update(kw)
    return ob.bucket(ref, b.first, b.count, b.num_vertices, b.low, **args)


@pytest.mark.parametrize("variant,lanes,group", [(5, 4, 0), (5, 8, 2), (1, 3, 2), (4, 2, 0), (5, 8, 3), (4, 4, 1), (5, 2, 1)])
def test_process_batch_equals_oracle_per_bucket(ctx, variant, lanes, group):
    """27 buckets of a shells cloud as ONE call: groups of `lanes` buckets share every launch of the octree build, and
    `group` of them (0: all) every launch of processCorners and marching (each kernel has a bucket dimension).  Every
    bucket's ship-outs equal the oracle's bit for bit -- and therefore mlsgpu_hip_worker_process's -- and the tree of every
    lane of the last group equals the oracle's."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud = synth.shells_cloud(120_000, 95.0, 16.0, 1.5, 2.5, seed=321)
    allb, buckets = synth.bucketize(cloud, 96, 32)
    assert len(buckets) == 27
    w = m.Worker(ctx, max(b.count for b in buckets), max_cells=63)
    w.set_mls_variant(variant)
    w.set_batch(lanes)
    w.set_marching_group(group)
    buf = m.DeviceBuffer(ctx, array=allb)
    before = w.marching_counters()
    got = w.process_batch(buf, buckets)
    after = w.marching_counters()
    assert len(got) == 27
    ref = allb.copy()
    exp_counts = dict(shipouts=0, occupied=0, unwelded=0, indices=0, welded=0, external=0)
    all_batches = []
    for b, g in zip(buckets, got):
        exp, st = _oracle_bucket(ref, b)
        assert_batches_equal(g, exp)
        all_batches += g
        for k in exp_counts:
            exp_counts[k] += st[k]
    for k, v in exp_counts.items():
        assert after[k] - before[k] == v, k
    # the mutated splats: every bucket's radius slot holds 1/r^2, exactly as the oracle left its copy
    np.testing.assert_array_equal(buf.download(m.SPLAT_DTYPE, len(allb)).view(np.uint32), ref.view(np.uint32))
    # trees of the last group (27 = 6 * 4 + 3, 3 * 8 + 3, 9 * 3, 13 * 2 + 1)
    last = len(buckets) - ((len(buckets) - 1) // lanes) * lanes
    pristine = allb.copy()
    for lane in range(last):
        b = buckets[len(buckets) - last + lane]
        commands, start = w.tree_arrays(lane)
        size = tuple(-(-n // 8) * 8 for n in b.num_vertices)
        t = ob.Tree(pristine.copy(), b.first, b.count, size, b.low, 3, 6)
        np.testing.assert_array_equal(start[:t.num_start], t.start[:t.num_start])
        np.testing.assert_array_equal(commands[:t.num_commands], t.commands[:t.num_commands])
    v, t, key_map = weld_batches(all_batches)
    assert len(key_map) > 0 and is_manifold(len(v), t) == ""


def test_process_batch_ragged_empty_and_keep_splats(ctx):
    """A batch whose buckets differ: ragged sizes away from the origin, a bucket without splats, one whose splats are all
    far away (an empty mesh), one sphere bucket -- with the non-mutating build, twice over the same resident buffer."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    a = synth.sphere_cloud(30_000, (70.0, 61.0, 52.0), 17.0, 1.0, 2.5, seed=99)
    far = synth.sphere_cloud(10, (500.0, 500.0, 500.0), 3.0, 1.0, 2.0, seed=1)
    c, g = synth.make_cloud("cfg1", scale=0.3)
    allb = np.concatenate([a, far, c])
    items = [(0, len(a), (45, 37, 30), (51, 46, 44)),
             (len(a), 0, (0, 0, 0), (64, 64, 64)),                      # no splats at all
             (len(a), len(far), (0, 0, 0), (64, 64, 64)),               # nothing reaches the bucket
             (len(a) + len(far), len(c), (0, 0, 0), (g, g, g)),
             (0, len(a), (50, 40, 35), (40, 33, 27))]                   # the first cloud again, another box
    w = m.Worker(ctx, len(allb), max_cells=63)
    w.set_batch(8)
    w.set_keep_splats(True)
    buf = m.DeviceBuffer(ctx, array=allb)
    for _ in range(2):
        got = w.process_batch(buf, items)
        for (first, count, low, nv), gb in zip(items, got):
            exp, _ = ob.bucket(allb.copy(), first, count, nv, low, max_cells=63, max_swathe=64, mesh_memory=63 * 63 * 2 * 872)
            assert_batches_equal(gb, exp)
        assert got[1] == [] and got[2] == []
        np.testing.assert_array_equal(buf.download(m.SPLAT_DTYPE, len(allb)).view(np.uint32), allb.view(np.uint32))


def test_process_batch_overflow_and_multi_swathe_fallbacks(ctx):
    """Buckets that cannot take the shared launches: with a tiny mesh memory a bucket's single swathe overflows and is
    split by the sequential path behind the batch (the reference's slice splitting, src/marching.cpp:652-701) while its
    neighbours ship from the batch; with 8-slice swathes the whole batch runs bucket by bucket.  Equal to the oracle
    either way, batch structure included."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud = synth.shells_cloud(120_000, 95.0, 16.0, 1.5, 2.5, seed=321)
    allb, buckets = synth.bucketize(cloud, 96, 32)
    some = [buckets[i] for i in (12, 13, 14, 4, 22)]      # 13 is the one in the middle of the shells: it overflows
    mm = 33 * 33 * 872               # the minimum: one slice's worst case
    for kw in (dict(mesh_memory=mm), dict(mesh_memory=mm, max_swathe=8)):
        w = m.Worker(ctx, max(b.count for b in buckets), max_cells=33, **kw)
        w.set_batch(4)
        buf = m.DeviceBuffer(ctx, array=allb)
        got = w.process_batch(buf, some)
        ref = allb.copy()
        splits = 0
        for b, g in zip(some, got):
            exp, st = ob.bucket(ref, b.first, b.count, b.num_vertices, b.low, max_cells=33,
                                max_swathe=kw.get("max_swathe", 40), mesh_memory=mm)
            assert_batches_equal(g, exp)
            splits += st["shipouts"] > 1
        assert splits > 0           # the case is exercised
        del w, buf


@pytest.mark.parametrize("group", [2, 0])
def test_process_batch_thin_and_odd_shapes(ctx, group):
    """Buckets as a partition's slabs make them: one block thick along each axis in turn, widths that are not multiples of
    the 64-cell chunks or of the 2 x 2 row groups, a single cell, a row of cells -- the edges of the row-pair / row-quad
    kernels (a layer's last row and the last layer alone, one word per lattice row, a chunk with one corner) -- in one batch,
    against the oracle bit for bit."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud = synth.shells_cloud(60_000, 70.0, 12.0, 1.5, 2.5, seed=77)
    shapes = [((20, 20, 20), (8, 8, 120)), ((20, 20, 20), (120, 8, 8)), ((20, 20, 20), (8, 120, 8)),
              ((30, 30, 30), (2, 2, 2)), ((25, 31, 28), (130, 2, 2)), ((10, 12, 14), (66, 3, 65)),
              ((33, 35, 30), (9, 65, 66)), ((28, 28, 28), (129, 5, 3)), ((16, 24, 20), (67, 66, 2))]
    items = [(0, len(cloud), low, nv) for low, nv in shapes]
    w = m.Worker(ctx, len(cloud), max_cells=135)
    w.set_batch(4)
    w.set_marching_group(group)
    w.set_keep_splats(True)
    buf = m.DeviceBuffer(ctx, array=cloud)
    got = w.process_batch(buf, items)
    nonempty = 0
    for (first, count, low, nv), gb in zip(items, got):
        exp, _ = ob.bucket(cloud.copy(), first, count, nv, low, max_cells=135, max_swathe=136, mesh_memory=135 * 135 * 2 * 872)
        assert_batches_equal(gb, exp)
        nonempty += len(gb) > 0
    assert nonempty >= 6            # the shapes do cut the shells


def test_process_batch_failure_in_the_middle(ctx):
    """An output functor that fails at bucket 5 of 9 (lanes 4, groups of 2): the call reports the callback error, says that
    the four buckets of the groups before it delivered everything (mlsgpu_hip_worker_batch_completed), has released every
    lane's splats, and the worker goes on to process the same buckets correctly."""
    import ctypes as C
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as mb, synth
    cloud = synth.shells_cloud(60_000, 95.0, 16.0, 1.5, 2.5, seed=77)
    allb, buckets = synth.bucketize(cloud, 96, 32)
    some = [b for b in buckets if b.count > 0][:9]
    assert len(some) == 9
    w = m.Worker(ctx, max(b.count for b in buckets), max_cells=63)
    w.set_batch(4)
    w.set_marching_group(2)
    buf = m.DeviceBuffer(ctx, array=allb)
    arr = (mb.SubItem * len(some))()
    for i, b in enumerate(some):
        arr[i].firstSplat, arr[i].numSplats = b.first, b.count
        for a in range(3):
            arr[i].lowExtent[a] = int(b.low[a])
            arr[i].numVertices[a] = int(b.num_vertices[a])
    seen = []

    def cb(user, index, stream, meshp):
        seen.append(int(index))
        return 1 if index == 5 else 0
    fn = mb.BATCH_OUTPUT_FN(cb)
    ctx.set_timing(True)
    rc = m.lib().mlsgpu_hip_worker_process_batch(w.h, buf.ptr, arr, len(some), fn, None)
    ctx.set_timing(False)
    assert rc != 0
    assert 5 in seen and max(seen) == 5
    done = m.lib().mlsgpu_hip_worker_batch_completed(w.h)
    assert done == 4                         # lanes 0-3 finished as groups (0,1) and (2,3); group (4,5) failed at 5
    assert "device.compute" in ctx.stats()   # the timing region of the failed group was closed
    # the worker is usable: the same buckets again, against the oracle
    buf.upload(allb)
    got = w.process_batch(buf, some)
    ref = allb.copy()
    for b, batches in zip(some, got):
        exp, _ = _oracle_bucket(ref, b)
        assert_batches_equal(batches, exp)
