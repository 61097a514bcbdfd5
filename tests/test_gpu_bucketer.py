"""HIP bucketer (mlsgpu_hip_bucket / mlsgpu_hip_bucket_load) against the bucketing oracle: identical leaves --
extents, chunk, depth and member ids -- in the same order, on the reference's fixtures and random cases."""
import numpy as np
import pytest

import oracle_binding as ob
from bucket_checks import create_splats, random_case, validate_partition
from test_oracle_bucket import GRID

pytestmark = pytest.mark.gpu


def both(splats, grid, max_splats, max_cells, chunk_cells, micro_cells, max_split):
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as b
    ctx = m.Context(0)
    dev = m.DeviceBuffer(ctx, array=splats) if len(splats) else None
    try:
        try:
            exp = ob.bucket_partition(splats, grid["reference"], grid["spacing"], grid["extents"], max_splats, max_cells,
                                      chunk_cells, micro_cells, max_split)
        except ob.DensityError as e:
            with pytest.raises(m.DensityError) as info:
                b.bucket_cloud(ctx, dev, len(splats), grid["reference"], grid["spacing"], grid["extents"], max_splats,
                               max_cells, chunk_cells, micro_cells, max_split)
            assert info.value.cell_splats == e.cell_splats
            return None, None
        got = b.bucket_cloud(ctx, dev, len(splats), grid["reference"], grid["spacing"], grid["extents"], max_splats,
                             max_cells, chunk_cells, micro_cells, max_split)
    finally:
        if dev is not None:
            dev.free()
        ctx.close()
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        assert g["extents"] == e["extents"] and g["chunk"] == e["chunk"] and g["depth"] == e["depth"]
        np.testing.assert_array_equal(g["ids"].astype(np.uint64), e["ids"])
    return got, exp


@pytest.mark.parametrize("args", [(5, 8, 0, 8, 1000000), (5, 8, 0, 8, 8), (15, 32, 0, 32, 1000000),
                                  (20, 2 ** 31 - 1, 14, 8, 1000000), (1, 8, 0, 8, 1000000), (5, 8, 0, 0, 64)])
def test_reference_fixture(args):
    got, _ = both(create_splats(), GRID, *args)
    if args == (5, 8, 0, 8, 1000000) or args == (5, 8, 0, 8, 8):
        assert len(got) == 11                 # test/test_bucket.cpp:475,546
    if args == (15, 32, 0, 32, 1000000):
        assert len(got) == 1                  # test/test_bucket.cpp:508


def test_empty_and_non_finite():
    got, _ = both(np.zeros(0, ob.SPLAT_DTYPE), GRID, 5, 8, 0, 8, 1000000)
    assert got == []
    bad = create_splats()
    bad["position"][3, 1] = np.nan
    bad["quality"][7] = np.inf
    got, _ = both(bad, GRID, 5, 8, 0, 8, 1000000)
    validate_partition(bad, GRID, got, 5, 8, 0)


@pytest.mark.parametrize("seed", range(30))
def test_random(seed):
    splats, grid, p = random_case(seed)
    got, exp = both(splats, grid, p["max_splats"], p["max_cells"], p["chunk_cells"], p["micro_cells"], p["max_split"])
    if got is not None:
        validate_partition(splats, grid, got, p["max_splats"], p["max_cells"], 0, strict=seed % 4 != 3)


@pytest.mark.parametrize("seed", range(0, 30, 3))
@pytest.mark.parametrize("private", ["0", "off"])
def test_random_with_kept_ranges_and_private_counters(seed, private, monkeypatch):
    """The paths big levels take -- the count keeps every element's microblock range for the member-list passes, and counts
    in per-workgroup LDS counters with corrections for the coarser levels -- forced onto small cases: same leaves."""
    monkeypatch.setenv("MLSGPU_HIP_BUCKET_NOTES_FROM", "0")
    monkeypatch.setenv("MLSGPU_HIP_BUCKET_PRIVATE_FROM", private)
    splats, grid, p = random_case(seed)
    both(splats, grid, p["max_splats"], p["max_cells"], p["chunk_cells"], p["micro_cells"], p["max_split"])
    both(create_splats(), GRID, 5, 8, 0, 8, 1000000)
    both(create_splats(), GRID, 5, 8, 0, 0, 64)
    both(create_splats(), GRID, 20, 2 ** 31 - 1, 14, 8, 1000000)


def test_big_level_takes_the_private_counters():
    """5 * 10^6 splats whose finest counters (24^3 microblocks) miss the shared LDS table: the level is counted in
    per-workgroup counters (several spans per node, every coarser level corrected) -- leaves identical to the oracle's."""
    from conftest import record_size
    from mlsgpu_amd import synth
    n, g = 5_000_000, 1500
    splats = synth.uniform_cloud(n, float(g - 1), 2.0, 3.0, 77)
    grid = {"reference": (0.0, 0.0, 0.0), "spacing": 1.0, "extents": (0, g - 1, 0, g - 1, 0, g - 1)}
    got, exp = both(splats, grid, 200_000, 255, 0, 63, 1 << 30)
    record_size("private-counter bucketing", "%d splats, %d leaves" % (n, len(got)))
    assert len(got) > 100


def test_private_counters_near_the_lds_limit(monkeypatch):
    """39^3 finest counters (16 bits each) + the corrections of the coarser levels = 155 KB of a CU's 160 KB of LDS in ONE
    workgroup: the launch works and counts right; one microblock more per axis (40^3: 165 KB) takes the plain kernel."""
    from mlsgpu_amd import synth
    monkeypatch.setenv("MLSGPU_HIP_BUCKET_PRIVATE_FROM", "0")
    for blocks in (39, 40):
        g = blocks * 63 + 1
        splats = synth.uniform_cloud(300_000, float(g - 1), 2.0, 3.0, 78 + blocks)
        grid = {"reference": (0.0, 0.0, 0.0), "spacing": 1.0, "extents": (0, g - 1, 0, g - 1, 0, g - 1)}
        got, exp = both(splats, grid, 2_000, 255, 0, 63, 1 << 30)
        assert len(got) > 100


def test_scratch_can_be_released_between_calls():
    """The bucketer keeps its lists and counters with the context; release_scratch hands them back (device memory in use
    drops), and the next call allocates afresh and finds the same leaves."""
    import torch
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as b, synth
    n, g = 2_000_000, 700
    splats = synth.uniform_cloud(n, float(g - 1), 2.0, 3.0, 5)
    ctx = m.Context(0)
    dev = m.DeviceBuffer(ctx, array=splats)
    ext = (0, g - 1, 0, g - 1, 0, g - 1)
    first = b.bucket_cloud(ctx, dev, n, (0.0, 0.0, 0.0), 1.0, ext, 100_000, 255, 0, 63, 1 << 30)
    held = torch.cuda.mem_get_info(0)[0]
    ctx.release_scratch()
    assert torch.cuda.mem_get_info(0)[0] >= held + n * 8            # at least the kept ranges came back
    again = b.bucket_cloud(ctx, dev, n, (0.0, 0.0, 0.0), 1.0, ext, 100_000, 255, 0, 63, 1 << 30)
    assert len(again) == len(first) > 10
    for x, y in zip(first, again):
        assert x["extents"] == y["extents"]
        np.testing.assert_array_equal(x["ids"], y["ids"])
    ctx.release_scratch()
    dev.free()
    ctx.close()


def test_rejects_an_empty_region():
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as b
    ctx = m.Context(0)
    dev = m.DeviceBuffer(ctx, array=create_splats())
    with pytest.raises(m.InvalidArgument):
        b.bucket_cloud(ctx, dev, 13, (0, 0, 0), 1.0, (0, 10, 5, 5, 0, 10), 5, 8)
    ctx.close()


def test_cloud_to_meshes_without_leaving_the_device():
    """bucket -> load -> worker for every leaf equals the oracle run on the oracle's partition."""
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as b, synth
    cloud = synth.shells_cloud(150_000, 95.0, 16.0, 1.5, 2.5, seed=77)       # grid coordinates 0..191
    spacing, ref = np.float32(0.02), np.array([-1.5, 0.25, 3.0], np.float32)
    world = cloud.copy()
    world["position"] = world["position"] * spacing + ref                      # a world-space cloud
    world["radius"] = world["radius"] * spacing
    extents = (-3, 188, 2, 193, 0, 191)                                        # the full grid does not start at vertex 0
    world["position"] += (np.array(extents[0::2], np.float32) * spacing)
    max_cells, max_splats = 63, 40000
    exp = ob.bucket_partition(world, ref, spacing, extents, max_splats, max_cells, 0, 16, 1 << 30)
    ctx = m.Context(0)
    dev = m.DeviceBuffer(ctx, array=world)
    worker = m.Worker(ctx, max_splats, max_cells=max_cells)
    staged = m.DeviceBuffer(ctx, nbytes=max_splats * 32)
    results = []

    def on_bucket(leaf, d_ids):
        low = [leaf["extents"][2 * i] - extents[2 * i] for i in range(3)]     # subGrid, src/bucket_loader.cpp:91-102
        nv = [leaf["extents"][2 * i + 1] - leaf["extents"][2 * i] + 1 for i in range(3)]
        b.bucket_load(ctx, dev, d_ids, leaf["num_splats"], ref, spacing, extents, staged)
        loaded = staged.download(m.SPLAT_DTYPE, leaf["num_splats"])
        results.append((loaded, worker.process(staged, 0, leaf["num_splats"], low, nv)))
    got = b.bucket_cloud(ctx, dev, len(world), ref, spacing, extents, max_splats, max_cells, 0, 16, 1 << 30, on_bucket=on_bucket)
    assert len(got) == len(exp) > 8
    total = 0
    for leaf, e, (loaded, batches) in zip(got, exp, results):
        assert leaf["extents"] == e["extents"] and leaf["num_splats"] == len(e["ids"])
        low = [e["extents"][2 * i] - extents[2 * i] for i in range(3)]
        nv = [e["extents"][2 * i + 1] - e["extents"][2 * i] + 1 for i in range(3)]
        host = world[e["ids"].astype(np.int64)].copy()
        b.transform_splats(host, ref, float(spacing), extents[0::2])           # the host loader's transform
        np.testing.assert_array_equal(loaded.view(np.uint32), host.view(np.uint32))
        ref_batches, _ = ob.bucket(host, 0, len(host), nv, low, max_cells=max_cells)
        assert len(batches) == len(ref_batches)
        for g, r in zip(batches, ref_batches):
            np.testing.assert_array_equal(g["vertices"].view(np.uint32), r["vertices"].view(np.uint32))
            np.testing.assert_array_equal(g["triangles"], r["triangles"])
            total += len(g["triangles"])
    assert total > 100000
    del worker
    ctx.close()


def test_bucketer_feeds_the_farm_on_the_device():
    """mlsgpu_hip_bucket -> mlsgpu_hip_farm_submit_device -> two worker threads: every leaf's meshes equal the oracle's
    for that leaf (chunk id = leaf number), with no host copy of splat data."""
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as b, synth
    cloud = synth.shells_cloud(150_000, 95.0, 16.0, 1.5, 2.5, seed=78)
    spacing, ref = np.float32(0.05), np.array([2.0, -1.0, 0.5], np.float32)
    world = cloud.copy()
    world["position"] = world["position"] * spacing + ref
    world["radius"] = world["radius"] * spacing
    extents = (0, 191, 0, 191, 0, 191)
    max_cells, max_splats = 63, 40000
    exp = ob.bucket_partition(world, ref, spacing, extents, max_splats, max_cells, 0, 16, 1 << 30)
    ctx = m.Context(0)
    dev = m.DeviceBuffer(ctx, array=world)
    farm = m.BucketFarm([0], max_splats, workers_per_device=2, collect=True, max_cells=max_cells)
    count = [0]

    def on_bucket(leaf, d_ids):
        low = [leaf["extents"][2 * i] - extents[2 * i] for i in range(3)]
        nv = [leaf["extents"][2 * i + 1] - leaf["extents"][2 * i] + 1 for i in range(3)]
        farm.submit_device(0, dev, d_ids, leaf["num_splats"], ref, spacing, extents, low, nv, count[0])
        count[0] += 1
    got = b.bucket_cloud(ctx, dev, len(world), ref, spacing, extents, max_splats, max_cells, 0, 16, 1 << 30, on_bucket=on_bucket)
    farm.finish()
    assert len(got) == len(exp) == farm.stats()["buckets"]
    nonempty = 0
    for i, e in enumerate(exp):
        low = [e["extents"][2 * k] - extents[2 * k] for k in range(3)]
        nv = [e["extents"][2 * k + 1] - e["extents"][2 * k] + 1 for k in range(3)]
        host = world[e["ids"].astype(np.int64)].copy()
        b.transform_splats(host, ref, float(spacing), extents[0::2])
        ref_batches, _ = ob.bucket(host, 0, len(host), nv, low, max_cells=max_cells)
        batches = farm.meshes.get(i, [])
        assert len(batches) == len(ref_batches)
        for g, r in zip(batches, ref_batches):
            np.testing.assert_array_equal(g["vertices"].view(np.uint32), r["vertices"].view(np.uint32))
            np.testing.assert_array_equal(g["triangles"], r["triangles"])
            nonempty += len(g["triangles"]) > 0
    assert nonempty > 8
    farm.close()
    ctx.close()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_bounding_grid(seed):
    """FastBlobSet::makeBoundingGrid (src/splat_set_impl.h:770-811): floor(min(p - r) / spacing) rounded down to a
    multiple of the bucket size, ceil(max(p + r) / spacing); non-finite splats do not count."""
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as b
    rng = np.random.default_rng(seed)
    n = 100_003
    s = np.zeros(n, ob.SPLAT_DTYPE)
    s["position"] = rng.uniform(-37.0, 91.0, (n, 3)).astype(np.float32)
    s["radius"] = rng.uniform(0.01, 3.0, n).astype(np.float32)
    s["normal"] = 1.0
    s["position"][5] = (1e9, np.nan, 0.0)              # ignored
    s["radius"][77] = np.inf                           # ignored
    spacing, bucket = np.float32(0.37), 7 + seed
    ok = np.isfinite(s["position"]).all(axis=1) & np.isfinite(s["radius"])
    lo = np.floor((s["position"][ok] - s["radius"][ok, None]).min(axis=0) / spacing).astype(np.int64) // bucket * bucket
    hi = np.ceil((s["position"][ok] + s["radius"][ok, None]).max(axis=0) / spacing).astype(np.int64)
    ctx = m.Context(0)
    dev = m.DeviceBuffer(ctx, array=s)
    ref, sp, ext = b.bounding_grid(ctx, dev, n, float(spacing), bucket)
    assert ref == (0.0, 0.0, 0.0) and sp == float(spacing)
    assert ext == (lo[0], hi[0], lo[1], hi[1], lo[2], hi[2])
    with pytest.raises(m.InvalidArgument):
        b.bounding_grid(ctx, m.DeviceBuffer(ctx, array=s[5:6]), 1, float(spacing), bucket)     # "Must be at least one splat"
    ctx.close()
