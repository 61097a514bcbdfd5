"""SplatSet::FileSet restated (mlsgpu_hip_fileset_*): several PLY files as one splat sequence, and the bounded-memory
multi-file loader into HBM -- row f4 of SURVEY.md section 8, the input side of BASELINE configs[4] (out-of-core PLY
splats) at reduced count."""
import os
import sys

import numpy as np
import pytest

HEAD = "ply\nformat binary_little_endian 1.0\n"


def write_splats(path, splats, extra=False):
    """A PLY file of splats; `extra` interleaves properties the reader must skip (test/test_fast_ply.cpp:403-432)."""
    n = len(splats)
    if extra:
        rows = np.zeros(n, np.dtype([("x", "<f4"), ("pad", "u1"), ("y", "<f4"), ("z", "<f4"), ("n", "<f4", 3),
                                     ("conf", "<f8"), ("radius", "<f4")]))
        head = HEAD + "element vertex %d\nproperty float32 x\nproperty uint8 pad\nproperty float32 y\nproperty float32 z\n" % n \
            + "property float32 nx\nproperty float32 ny\nproperty float32 nz\nproperty float64 conf\nproperty float32 radius\nend_header\n"
    else:
        rows = np.zeros(n, np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("n", "<f4", 3), ("radius", "<f4")]))
        head = HEAD + "element vertex %d\nproperty float32 x\nproperty float32 y\nproperty float32 z\n" % n \
            + "property float32 nx\nproperty float32 ny\nproperty float32 nz\nproperty float32 radius\nend_header\n"
    rows["x"], rows["y"], rows["z"] = splats["position"][:, 0], splats["position"][:, 1], splats["position"][:, 2]
    rows["n"] = splats["normal"]
    rows["radius"] = splats["radius"]
    with open(path, "wb") as f:
        f.write(head.encode("ascii"))
        f.write(rows.tobytes())


def make_files(tmp_path, cloud, cuts):
    paths, first = [], 0
    for k, n in enumerate(cuts):
        p = tmp_path / ("scan%d.ply" % k)
        write_splats(p, cloud[first:first + n], extra=(k % 2 == 1))
        paths.append(p)
        first += n
    assert first == len(cloud)
    return paths


def decoded(cloud, smooth=1.0, max_radius=float("inf")):
    """What Reader::decode makes of the stored splats (src/fast_ply.cpp:334-350)."""
    out = cloud.copy()
    r = np.minimum(out["radius"], np.float32(max_radius)) * np.float32(smooth)
    out["radius"] = r
    out["quality"] = (1.0 / (r * r).astype(np.float64)).astype(np.float32)
    return out


def test_fileset_reads_files_as_one_sequence(tmp_path):
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud = synth.sphere_cloud(10_000, (3.0, -2.0, 7.5), 4.0, 0.1, 0.3, seed=8)
    paths = make_files(tmp_path, cloud, [1, 4999, 0, 3000, 2000])          # an empty file and a one-splat file too
    fs = m.binding.FileSet(paths, smooth=1.5, max_radius=0.25)
    assert len(fs) == 10_000
    want = decoded(cloud, 1.5, 0.25)
    np.testing.assert_array_equal(fs.read().view(np.uint32), want.view(np.uint32))
    for first, count in ((0, 1), (0, 2), (4999, 2), (5000, 3000), (7999, 2001), (10_000, 0), (1234, 7000)):
        np.testing.assert_array_equal(fs.read(first, count).view(np.uint32), want[first:first + count].view(np.uint32))
    with pytest.raises(m.LengthError):
        fs.read(9000, 1001)
    with pytest.raises(m.FormatError):
        bad = tmp_path / "bad.ply"
        bad.write_bytes(b"ply no not really")
        fs.add_file(bad)
    with pytest.raises(m.InvalidArgument):
        fs.add_file(tmp_path / "missing.ply")
    assert len(fs) == 10_000                                               # a rejected file leaves the set unchanged


def test_a_set_of_many_files_needs_few_descriptors(tmp_path):
    """The reference's sets are hundreds of scan files: a reader holds ONE of them open at a time (120 files under a limit
    of 40 descriptors, in a child process so that the limit is its own)."""
    import subprocess
    from mlsgpu_amd import synth
    cloud = synth.sphere_cloud(1200, (3.0, -2.0, 7.5), 4.0, 0.1, 0.3, seed=9)
    paths = make_files(tmp_path, cloud, [10] * 120)
    np.save(tmp_path / "want.npy", decoded(cloud).view(np.uint32))
    code = ("import resource, sys, numpy as np\n"
            "sys.path.insert(0, %r)\n"
            "import mlsgpu_amd as m\n"
            "resource.setrlimit(resource.RLIMIT_NOFILE, (40, 40))\n"
            "fs = m.binding.FileSet(%r)\n"
            "got = fs.read().view(np.uint32)\n"
            "assert np.array_equal(got, np.load(%r)), 'differs'\n"
            "print('ok', len(fs))\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), [str(p) for p in paths],
                                       str(tmp_path / "want.npy"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok 1200" in out.stdout, out.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 4])
def test_fileset_load_with_bounded_buffer(tmp_path, threads):
    """Files far larger than the pinned buffer (64 KiB = four 512-splat quarters against 20 k - 120 k splat files):
    every chunk boundary and every file boundary inside a chunk is exercised; the device receives exactly the sequence."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud = synth.sphere_cloud(300_000, (30.0, 20.0, 70.5), 40.0, 0.5, 1.5, seed=11)
    paths = make_files(tmp_path, cloud, [120_000, 20_001, 59_999, 100_000])
    fs = m.binding.FileSet(paths, buffer_size=64 << 10)
    want = decoded(cloud)
    ctx = m.Context(0)
    dev = m.DeviceBuffer(ctx, nbytes=len(cloud) * 32)
    assert fs.load(ctx, dev, reader_threads=threads) == len(cloud)
    np.testing.assert_array_equal(dev.download(m.SPLAT_DTYPE, len(cloud)).view(np.uint32), want.view(np.uint32))
    fs.load(ctx, dev, first=119_990, count=20_030, reader_threads=threads)          # spans three files
    np.testing.assert_array_equal(dev.download(m.SPLAT_DTYPE, 20_030).view(np.uint32), want[119_990:140_020].view(np.uint32))
    # a file that disappears while it is being read is an error, not a hang
    os.remove(paths[2])
    with pytest.raises(m.MlsError):
        fs.load(ctx, dev, reader_threads=threads)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("raw", ["1", "0"])
def test_fileset_load_rows_decoded_on_the_device(tmp_path, raw, monkeypatch):
    """Files whose rows are whole words (the plain 28-byte row; a row with a float32 the reader skips, properties in another
    order) cross the link as they are and are decoded by a kernel; MLSGPU_HIP_FILESET_RAW=0 decodes on the host.  Same
    splats bit for bit: clamped and smoothed radii, 1 / r^2 through a double, a NaN radius, file boundaries inside a chunk,
    a range that starts and ends inside files."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    monkeypatch.setenv("MLSGPU_HIP_FILESET_RAW", raw)
    cloud = synth.sphere_cloud(200_000, (30.0, 20.0, 70.5), 40.0, 0.5, 1.5, seed=12)
    cloud["radius"][77] = np.nan
    cloud["radius"][78] = 0.0
    cloud["radius"][150_000] = np.inf
    cuts = [70_000, 1, 0, 69_999, 60_000]
    paths, first = [], 0
    for k, n in enumerate(cuts):
        part = cloud[first:first + n]
        p = tmp_path / ("rows%d.ply" % k)
        if k % 2 == 0:
            write_splats(p, part)
        else:       # radius first, a float32 to skip, normals before positions: 32-byte rows of words
            rows = np.zeros(n, np.dtype([("radius", "<f4"), ("conf", "<f4"), ("n", "<f4", 3), ("x", "<f4"), ("y", "<f4"), ("z", "<f4")]))
            rows["radius"], rows["n"], rows["conf"] = part["radius"], part["normal"], 0.5
            rows["x"], rows["y"], rows["z"] = part["position"][:, 0], part["position"][:, 1], part["position"][:, 2]
            head = HEAD + "element vertex %d\nproperty float32 radius\nproperty float32 conf\nproperty float32 nx\nproperty float32 ny\n" % n \
                + "property float32 nz\nproperty float32 x\nproperty float32 y\nproperty float32 z\nend_header\n"
            with open(p, "wb") as f:
                f.write(head.encode("ascii"))
                f.write(rows.tobytes())
        paths.append(p)
        first += n
    fs = m.binding.FileSet(paths, smooth=1.5, max_radius=1.25, buffer_size=64 << 10)
    with np.errstate(divide="ignore", invalid="ignore"):
        want = decoded(cloud, 1.5, 1.25)
    ctx = m.Context(0)
    dev = m.DeviceBuffer(ctx, nbytes=len(cloud) * 32)

    def same(got, exp):
        """bit for bit, a NaN for a NaN (which NaN a division makes of a NaN is the machine's choice)"""
        got, exp = got.copy(), exp.copy()
        for field in ("radius", "quality"):
            nan = np.isnan(exp[field])
            assert np.array_equal(np.isnan(got[field]), nan)
            got[field][nan] = exp[field][nan] = 0
        np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32))
    for threads in (1, 4):
        assert fs.load(ctx, dev, reader_threads=threads) == len(cloud)
        same(dev.download(m.SPLAT_DTYPE, len(cloud)), want)
    fs.load(ctx, dev, first=69_990, count=70_030, reader_threads=3)
    same(dev.download(m.SPLAT_DTYPE, 70_030), want[69_990:140_020])
    same(fs.read(), want)               # the host route agrees
    os.remove(paths[3])
    with pytest.raises(m.MlsError):
        fs.load(ctx, dev, reader_threads=2)
    ctx.close()


@pytest.mark.gpu
def test_cfg5_shape_files_to_welded_mesh(tmp_path):
    """BASELINE configs[4]'s route at reduced count: splats in several PLY files -> HBM through the bounded buffer ->
    bounding grid -> Bucket::bucket on the device -> the farm's device groups (eight, on GPU 0) by device-side loads ->
    ship-outs into the device mesh sink -> one welded, pruned mesh; equal to the oracle chain (bucketing oracle ->
    bucket oracle -> mesh-sink oracle) up to vertex / triangle order."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import mesher_oracle as mo
    import oracle_binding as ob
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as mb, synth
    cloud = synth.shells_cloud(150_000, 127.0, 24.0, 1.5, 2.5, seed=55)
    cloud["position"] *= np.float32(0.5)                      # world units: spacing 0.5 puts the shells on a 128^3 grid
    cloud["radius"] *= np.float32(0.5)
    paths = make_files(tmp_path, cloud, [50_000, 30_000, 70_000])
    spacing, bucket_size = 0.5, 63
    fs = mb.FileSet(paths, buffer_size=256 << 10)
    ctx = m.Context(0)
    raw = m.DeviceBuffer(ctx, nbytes=len(fs) * 32)
    fs.load(ctx, raw)
    host = fs.read()
    reference, _, ext = mb.bounding_grid(ctx, raw, len(fs), spacing, bucket_size)
    grid = dict(reference=reference)
    bp = dict(max_splats=40_000, max_cells=bucket_size, chunk_cells=0, micro_cells=0, max_split=1 << 20)
    sink = m.Mesher(ctx, 0.02)
    farm = m.BucketFarm([0] * 8, bp["max_splats"], workers_per_device=1, max_cells=bucket_size, sink=sink,
                        grid_spacing=spacing, grid_origin=[grid["reference"][a] + spacing * ext[2 * a] for a in range(3)])
    leaves = []

    def leaf_work(leaf, d_ids):
        low = [leaf["extents"][2 * a] - ext[2 * a] for a in range(3)]
        nv = [leaf["extents"][2 * a + 1] - leaf["extents"][2 * a] + 1 for a in range(3)]
        farm.submit_device(0, raw, d_ids, leaf["num_splats"], grid["reference"], spacing, ext, low, nv, 0)
        leaves.append(leaf)
    mb.bucket_cloud(ctx, raw, len(fs), grid["reference"], spacing, ext, on_bucket=leaf_work, **bp)
    farm.finish()
    assert len(leaves) > 8 and sum(1 for x in farm.stats()["per_device"][:8] if x > 0) >= 4
    farm.close()
    assert sink.finalize() == 1
    got = sink.chunk(0)
    stats = sink.stats()
    # the oracle chain on the host copy of what the files decode to
    oleaves = ob.bucket_partition(host.copy(), grid["reference"], spacing, ext, bp["max_splats"], bp["max_cells"],
                                  bp["chunk_cells"], bp["micro_cells"], bp["max_split"])
    assert [l["extents"] for l in oleaves] == [tuple(l["extents"]) for l in leaves]
    meshes = []
    for leaf in oleaves:
        s = host[leaf["ids"].astype(np.int64)].copy()
        mb.transform_splats(s, grid["reference"], spacing, [ext[0], ext[2], ext[4]])
        low = [leaf["extents"][2 * a] - ext[2 * a] for a in range(3)]
        nv = [leaf["extents"][2 * a + 1] - leaf["extents"][2 * a] + 1 for a in range(3)]
        batches, _ = ob.bucket(s, 0, len(s), nv, low, max_cells=bucket_size, max_swathe=bucket_size + 1,
                               mesh_memory=bucket_size * bucket_size * 2 * 872)
        for g in batches:
            ob.lib().orc_scale_bias(ob._p(g["vertices"]), len(g["vertices"]), spacing,
                                    *[grid["reference"][a] + spacing * ext[2 * a] for a in range(3)])
            meshes.append(dict(chunk=0, vertices=g["vertices"], num_internal=g["num_internal"],
                               keys=g["keys"][g["num_internal"]:], triangles=g["triangles"]))
    exp, exp_stats = mo.mesh_sink(meshes, 0.02)
    for k in ("total_vertices", "threshold", "components", "kept_components", "kept_vertices", "kept_triangles"):
        assert stats[k] == exp_stats[k], k
    assert mo.isomorphic(got["vertices"], got["triangles"], exp[0][1], exp[0][2])
    sink.close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("budget,chunk,coherent", [(60_000, 25_000, False), (400_000, 64, False), (5_000, 300_000, False),
                                                  (60_000, 10_000, True)])
def test_streamed_bucketing_equals_resident(tmp_path, budget, chunk, coherent):
    """mlsgpu_hip_bucket_stream (a splat set that does not fit the device: the files streamed through a chunk buffer, once
    to count and once per batch of top-level regions) makes the buckets mlsgpu_hip_bucket makes of the same set resident --
    extents, order and member splats in file order -- whatever the budget and the chunk size; a budget below the largest
    top-level region is a length error, not a wrong partition."""
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as mb, synth
    cloud = synth.shells_cloud(150_000, 127.0, 24.0, 1.5, 2.5, seed=77)
    cloud["position"] *= np.float32(0.5)
    cloud["radius"] *= np.float32(0.5)
    if coherent:        # files in scan order (here: sorted along z): a batch's pass skips the chunks that stay clear of it
        cloud = cloud[np.argsort(cloud["position"][:, 2], kind="stable")]
    paths = make_files(tmp_path, cloud, [50_000, 30_000, 70_000])
    spacing, bucket_size = 0.5, 63
    fs = mb.FileSet(paths, buffer_size=256 << 10)
    ctx = m.Context(0)
    raw = m.DeviceBuffer(ctx, nbytes=len(fs) * 32)
    fs.load(ctx, raw)
    reference, _, ext = mb.bounding_grid(ctx, raw, len(fs), spacing, bucket_size)
    assert mb.bounding_grid_files(ctx, fs, spacing, bucket_size, chunk)[2] == ext
    bp = dict(max_splats=12_000, max_cells=bucket_size, chunk_cells=0, micro_cells=0, max_split=1 << 20)

    def gather(store):
        def take(leaf, d_splats, d_ids):
            staged = m.DeviceBuffer(ctx, nbytes=max(leaf["num_splats"], 1) * 32)
            mb.bucket_load(ctx, d_splats, d_ids, leaf["num_splats"], reference, spacing, ext, staged)
            ctx.synchronize()
            store.append(staged.download(m.SPLAT_DTYPE, leaf["num_splats"]))
        return take
    want = []
    resident = mb.bucket_cloud(ctx, raw, len(fs), reference, spacing, ext,
                               on_bucket=lambda leaf, d_ids: gather(want)(leaf, raw, d_ids), **bp)
    assert len(resident) > 20 and max(l["depth"] for l in resident) >= 1
    got = []
    if budget < 6_000:          # below the largest top-level region (150 k splats in at most 27 microblocks)
        with pytest.raises(mb.LengthError):
            mb.bucket_cloud_stream(ctx, fs, reference, spacing, ext, budget_splats=budget, chunk_splats=chunk,
                                   on_bucket=gather(got), **bp)
        return
    leaves, stats = mb.bucket_cloud_stream(ctx, fs, reference, spacing, ext, budget_splats=budget, chunk_splats=chunk,
                                           on_bucket=gather(got), **bp)
    assert [(l["extents"], l["chunk"], l["depth"], l["num_splats"]) for l in leaves] == \
        [(l["extents"], l["chunk"], l["depth"], l["num_splats"]) for l in resident]
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    assert stats["file_passes"] == 1 + stats["batches"] and stats["batch_splats"] >= len(fs)
    if budget < len(fs):
        assert stats["batches"] >= 2
    assert (stats["chunks_skipped"] > 0) == coherent
    ctx.close()


@pytest.mark.gpu
def test_streamed_set_to_welded_mesh(tmp_path):
    """The out-of-core route end to end: files -> streamed bucketing (three batches) -> the farm's device path (every bucket
    gathered out of the resident batch during the callback) -> device sink; the same welded, pruned mesh as the set resident."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import mesher_oracle as mo
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as mb, farm as fm, synth
    cloud = synth.shells_cloud(150_000, 127.0, 24.0, 1.5, 2.5, seed=55)
    cloud["position"] *= np.float32(0.5)
    cloud["radius"] *= np.float32(0.5)
    paths = make_files(tmp_path, cloud, [50_000, 30_000, 70_000])
    spacing, bucket_size = 0.5, 63
    fs = mb.FileSet(paths, buffer_size=256 << 10)
    ctx = m.Context(0)
    raw = m.DeviceBuffer(ctx, nbytes=len(fs) * 32)
    fs.load(ctx, raw)
    reference, _, ext = mb.bounding_grid(ctx, raw, len(fs), spacing, bucket_size)
    bp = dict(max_splats=40_000, max_cells=bucket_size, chunk_cells=0, micro_cells=0, max_split=1 << 20)
    origin = [reference[a] + spacing * ext[2 * a] for a in range(3)]
    results = []
    for streamed in (False, True):
        sink = m.Mesher(ctx, 0.02)
        farm = m.BucketFarm([0, 0], bp["max_splats"], workers_per_device=2, max_cells=bucket_size, sink=sink,
                            grid_spacing=spacing, grid_origin=origin)
        if streamed:
            def work(leaf, d_splats, d_ids):
                low, nv = fm.leaf_geometry(leaf, ext)
                farm.submit_device(0, d_splats, d_ids, leaf["num_splats"], reference, spacing, ext, low, nv, 0)
            leaves, stats = mb.bucket_cloud_stream(ctx, fs, reference, spacing, ext, budget_splats=70_000, chunk_splats=40_000,
                                                   on_bucket=work, **bp)
            assert stats["batches"] >= 3
        else:
            fm.partition_to_farm(ctx, farm, 0, raw, len(fs), reference, spacing, ext, bp, chunk_of=lambda i: 0)
        farm.finish()
        farm.close()
        assert sink.finalize() == 1
        results.append((sink.stats(), sink.chunk(0)))
        sink.close()
    (s0, c0), (s1, c1) = results
    for k in ("total_vertices", "threshold", "components", "kept_components", "kept_vertices", "kept_triangles"):
        assert s0[k] == s1[k], k
    assert mo.isomorphic(c0["vertices"], c0["triangles"], c1["vertices"], c1["triangles"])
    ctx.close()
