"""The BASELINE.json configurations that only fit a GPU, at full size (cfg3) or full shape (cfg4), under -m gpu.

cfg2  256^3 grid, 5 M splats, ONE bucket of 255 cells per side (the only configuration with a full-size single bucket and
      the only one whose bucket exceeds 2 M splats): the WHOLE bucket against the oracle bit for bit with every MLS kernel,
      on the uniform cloud and on its shells companion of SURVEY 8(d); totals + digest pinned in
      tests/golden/cfg2_{uniform,shells}.json (bench.py --workload cfg2 checks the same pins).
cfg3  512^3 grid, 50 M uniform splats, 27 buckets: the cloud bench.py times.  Oracle bit-parity on three buckets (corner,
      face, centre), size-independent properties on all 27 (bucket tiling, cross-bucket agreement of shared vertices bit
      for bit, key multiplicity), and the totals + digest that bench.py prints for its timed passes, pinned in
      tests/golden/cfg3_uniform.json.
cfg4s ALL EIGHT slabs of the sharded cfg4 cloud at FULL density (what bench.py --gpus N runs, one slab per GPU), one after
      the other on this GPU, as an inner and as a last slab: totals + digest of every rank of an N = 2 / 4 / 8 job pinned in
      tests/golden/cfg4slab_uniform.json (bench.py compares every rank's digest with its pin), oracle bit-parity on a
      bucket of slabs 0, 3 and 7, and the seven slab seams bit-identical from both sides.
cfg4w the WHOLE cfg4 cloud as the reference's partition would cut it on a regular grid (125 buckets): per-bucket digests for
      bench.py --dispatch greedy (one process, device groups, the reference's dispatch rule), uniform and shells, next to
      oracle bit-parity on three buckets.
cfg5  2048^3 grid, 10^9 splats (a smaller count only if the box lacks the RAM) written as 8 PLY files by the device
      generator -> FileSet reader threads -> HBM -> Bucket::bucket on the device -> eight device groups by device-side
      gathers: partition properties on all ~730 buckets, oracle bit-parity on three leaves, cross-bucket agreement on a
      neighbourhood, totals + digest pinned in tests/golden/cfg5_uniform.json.
cfg4  1024^3 grid cut 8 ways: the multi-device bucket farm with eight device groups (all on GPU 0 here: a one-GPU box
      stands in for eight GPUs; on an 8-GPU node pass MLSGPU_TEST_DEVICES=0,1,...,7), ship-outs read back through the
      pinned circular buffer into the host welder; splat count reduced, oracle parity on sampled buckets, and the Euler
      characteristic of the welded shells as the whole-job check.
"""
import json
import os

import numpy as np
import pytest

import oracle_binding as ob
from conftest import record_size
from gpu_common import assert_batches_equal, ctx  # noqa: F401

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg3_uniform.json")


def test_device_generators_match_numpy():
    """The clouds bench.py and the tests below generate in HBM are the numpy clouds of mlsgpu_amd.synth, bit for bit."""
    import torch
    from mlsgpu_amd import synth
    dev = torch.device("cuda", 0)
    ref, g = synth.make_cloud("cfg3", scale=0.01)
    got, _ = synth.make_cloud_device("cfg3", dev, scale=0.01)
    np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32).ravel(), ref.view(np.uint32).ravel())
    allb, buckets = synth.bucketize(ref, g, 255)
    gb, gbuckets = synth.bucketize_device(got, synth.grid_buckets((g, g, g), 255))
    np.testing.assert_array_equal(gb.cpu().numpy().view(np.uint32).ravel(), allb.view(np.uint32).ravel())
    assert [(b.low, b.num_vertices, b.first, b.count) for b in gbuckets] == \
        [(b.low, b.num_vertices, b.first, b.count) for b in buckets]
    # the tail of the full-size cloud is the same stream
    tail = synth.uniform_cloud_device(1000, 511.0, 2.0, 3.0, synth.cloud_seed("cfg3"), dev, first=49_999_000)
    exp = synth.uniform_cloud(1000, 511.0, 2.0, 3.0, synth.cloud_seed("cfg3"), first=49_999_000)
    np.testing.assert_array_equal(tail.cpu().numpy().view(np.uint32).ravel(), exp.view(np.uint32).ravel())


def test_mesh_checksum_matches_host(ctx):
    """mlsgpu_hip_mesh_checksum (the digest bench.py prints) against the same sums computed from the copied-back mesh."""
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as mb, synth
    cloud, g = synth.make_cloud("cfg1")
    w = m.Worker(ctx, len(cloud), max_cells=63)
    buf = m.DeviceBuffer(ctx, array=cloud)
    batches = w.process(buf, 0, len(cloud), (0, 0, 0), (g, g, g))
    buf.upload(cloud)
    col = w.process(buf, 0, len(cloud), (0, 0, 0), (g, g, g), collector=mb.ChecksumCollector(ctx))
    assert col.batches == len(batches) >= 1
    for b, rec in zip(batches, col.sums):
        assert rec[:3] == (len(b["vertices"]), len(b["triangles"]), b["num_internal"])
        assert rec[3:] == mb.batch_checksum(b)
    exp, _ = ob.bucket(cloud.copy(), 0, len(cloud), (g, g, g), (0, 0, 0), max_cells=63, max_swathe=64,
                       mesh_memory=63 * 63 * 2 * 872)
    sums = [(len(b["vertices"]), len(b["triangles"]), b["num_internal"]) + mb.batch_checksum(b) for b in exp]
    assert mb.digest_of_sums(sums) == col.digest()          # the oracle's meshes have the digest the device computed


def test_cfg3_full_size(ctx):
    import torch
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as mb, synth
    dev = torch.device("cuda", 0)
    cloud, g = synth.make_cloud_device("cfg3", dev)
    assert len(cloud) == 50_000_000 and g == 512
    record_size("cfg3", "%d splats, 27 buckets" % len(cloud))
    bucketed, buckets = synth.bucketize_device(cloud, synth.grid_buckets((g, g, g), 255))
    del cloud
    torch.cuda.synchronize()
    assert len(buckets) == 27 and sum(b.cells for b in buckets) == (g - 1) ** 3
    max_cells = max(max(b.num_vertices) for b in buckets) - 1
    max_count = max(b.count for b in buckets)
    nbytes = bucketed.numel() * 4
    pristine = m.DeviceBuffer(ctx, nbytes=nbytes, borrow=bucketed.data_ptr())
    work = m.DeviceBuffer(ctx, nbytes=nbytes)

    # ---- all 27 buckets with bench.py's settings: totals, digest, cross-bucket agreement ----
    w = m.Worker(ctx, max_count, max_cells=max_cells, mesh_memory=4096 << 20)
    work.copy_from(pristine)
    col = mb.ExternalCollector(ctx)
    for b in buckets:
        w.process(work, b.first, b.count, b.low, b.num_vertices, collector=col)
    cnt = w.marching_counters()
    assert cnt["welded"] == col.vertices and cnt["indices"] == 3 * col.triangles
    keys = np.concatenate(col.ext_keys)
    verts = np.concatenate(col.ext_vertices).view(np.uint32)
    order = np.argsort(keys, kind="stable")
    keys, verts = keys[order], verts[order]
    same = keys[1:] == keys[:-1]
    assert same.sum() > 1_000_000                                       # buckets share millions of face vertices
    assert np.all(verts[1:][same] == verts[:-1][same])                  # ... and agree on every one, bit for bit
    _, counts = np.unique(keys, return_counts=True)
    assert counts.max() <= 8
    got = dict(triangles=int(col.triangles), vertices=int(col.vertices), external=int(col.external), shipouts=col.batches,
               digest=col.digest())
    if os.environ.get("MLSGPU_WRITE_GOLDEN"):
        out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "cfg3_uniform.json")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        json.dump(got, open(out, "w"), indent=1)
    exp = json.load(open(GOLDEN))
    assert got == {k: exp[k] for k in got}, (got, exp)

    # ---- oracle bit-parity on a corner, a face and the centre bucket (2 M splats, 170^3 cells each) ----
    del w
    mm = 1 << 30
    w = m.Worker(ctx, max_count, max_cells=max_cells, mesh_memory=mm)
    for i in (0, 4, 13):
        b = buckets[i]
        host = bucketed[b.first:b.first + b.count].cpu().numpy().view(m.SPLAT_DTYPE).reshape(-1)
        work.copy_from(pristine)
        batches = w.process(work, b.first, b.count, b.low, b.num_vertices)
        exp_b, st = ob.bucket(host, 0, b.count, b.num_vertices, b.low, max_cells=max_cells, max_swathe=max_cells + 1,
                              mesh_memory=mm)
        assert st["shipouts"] == len(batches) >= 1
        assert_batches_equal(batches, exp_b)
        del batches, exp_b


@pytest.mark.parametrize("dist", ["uniform", "shells"])
def test_cfg2_full_size(ctx, dist):
    """BASELINE configs[1] at full size against the oracle: one 255-cell bucket of 5 M splats (the oracle's processCorners
    is OpenMP-parallel; ~3.4 x one of the buckets test_cfg3_full_size runs).  The reference's own expectations at this
    shape are analytic (test/test_mls.cpp:416-514, test/test_marching.cpp:594-632); here every ship-out -- vertex bits,
    triangles, keys, the internal / external split, batch boundaries -- equals the oracle's, with each of the three MLS
    kernels; the totals and the device-side digest that bench.py --workload cfg2 prints are pinned."""
    import torch
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as mb, synth
    dev = torch.device("cuda", 0)
    cloud, g = synth.make_cloud_device("cfg2", dev, dist=dist)
    assert len(cloud) == 5_000_000 and g == 256
    record_size("cfg2 %s" % dist, "%d splats, one bucket of 255^3 cells, whole bucket against the oracle" % len(cloud))
    bucketed, buckets = synth.bucketize_device(cloud, synth.grid_buckets((g, g, g), 255))
    del cloud
    assert len(buckets) == 1
    b = buckets[0]
    assert b.low == (0, 0, 0) and tuple(b.num_vertices) == (g, g, g)
    nbytes = bucketed.numel() * 4
    pristine = m.DeviceBuffer(ctx, nbytes=nbytes, borrow=bucketed.data_ptr())
    work = m.DeviceBuffer(ctx, nbytes=nbytes)

    # ---- bench.py's settings: totals + digest, the same for every MLS kernel ----
    got = None
    for variant in (5, 4, 1):
        w = m.Worker(ctx, b.count, max_cells=255, mesh_memory=4096 << 20)
        w.set_mls_variant(variant)
        work.copy_from(pristine)
        col = w.process(work, b.first, b.count, b.low, b.num_vertices, collector=mb.ChecksumCollector(ctx))
        cnt = w.marching_counters()
        assert cnt["welded"] == col.vertices and cnt["indices"] == 3 * col.triangles
        this = dict(triangles=int(col.triangles), vertices=int(col.vertices), external=int(col.external), shipouts=col.batches,
                    digest=col.digest())
        assert got is None or this == got, (variant, this, got)
        got = this
        del w
    golden = os.path.join(os.path.dirname(GOLDEN), "cfg2_%s.json" % dist)
    if os.environ.get("MLSGPU_WRITE_GOLDEN"):
        out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "cfg2_%s.json" % dist)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        json.dump(got, open(out, "w"), indent=1)         # (the pin is only worth what the oracle parity below says)
    else:
        exp = json.load(open(golden))
        assert got == {k: exp[k] for k in got}, (got, exp)

    # ---- the whole bucket against the oracle, 1 GB of mesh memory (several ship-outs: the overflow splitting at full size) ----
    mm = 1 << 30
    host = bucketed[b.first:b.first + b.count].cpu().numpy().view(m.SPLAT_DTYPE).reshape(-1)
    exp_b, st = ob.bucket(host, 0, b.count, b.num_vertices, b.low, max_cells=255, max_swathe=256, mesh_memory=mm)
    for variant in (5, 4):
        w = m.Worker(ctx, b.count, max_cells=255, mesh_memory=mm)
        w.set_mls_variant(variant)
        work.copy_from(pristine)
        batches = w.process(work, b.first, b.count, b.low, b.num_vertices)
        assert st["shipouts"] == len(batches) >= 1
        assert_batches_equal(batches, exp_b)
        del batches, w


@pytest.mark.parametrize("dist", ["uniform", "shells"])
def test_cfg4_bucket_pins(ctx, dist):
    """BASELINE configs[3] WHOLE (1024^3 grid, 200 M splats, 125 buckets of <= 255 cells per side), bucket by bucket on one
    worker: the per-bucket digests `bench.py --gpus N --dispatch greedy` holds every bucket against, whichever GPU the
    reference's greedy rule (src/workers.cpp:320-351) sends it to -- pinned in tests/golden/cfg4_buckets_<dist>.json next to
    oracle bit-parity on a corner, the centre and the far-corner bucket."""
    import torch
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as mb, synth
    dev = torch.device("cuda", 0)
    cloud, g = synth.make_cloud_device("cfg4", dev, dist=dist)
    assert len(cloud) == 200_000_000 and g == 1024
    record_size("cfg4 whole, %s" % dist, "%d splats, 125 buckets, per-bucket digests" % len(cloud))
    bucketed, buckets = synth.bucketize_device(cloud, synth.grid_buckets((g, g, g), 255))
    del cloud
    torch.cuda.synchronize()
    assert len(buckets) == 125 and sum(b.cells for b in buckets) == (g - 1) ** 3
    max_cells = max(max(b.num_vertices) for b in buckets) - 1
    max_count = max(b.count for b in buckets)
    pristine = m.DeviceBuffer(ctx, nbytes=bucketed.numel() * 4, borrow=bucketed.data_ptr())
    w = m.Worker(ctx, max_count, max_cells=max_cells, mesh_memory=4096 << 20)
    w.set_keep_splats(True)
    per, tri, ver = [], 0, 0
    for b in buckets:
        col = w.process(pristine, b.first, b.count, b.low, b.num_vertices, collector=mb.ChecksumCollector(ctx))
        per.append(dict(low=list(b.low), num_vertices=list(b.num_vertices), splats=int(b.count), triangles=int(col.triangles),
                        digest=col.digest()))
        tri += int(col.triangles)
        ver += int(col.vertices)
    got = dict(buckets=per, total=dict(triangles=tri, vertices=ver))
    golden = os.path.join(os.path.dirname(GOLDEN), "cfg4_buckets_%s.json" % dist)
    if os.environ.get("MLSGPU_WRITE_GOLDEN"):
        out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "cfg4_buckets_%s.json" % dist)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        json.dump(got, open(out, "w"), indent=0)
    else:
        assert got == json.load(open(golden))
    del w
    # ---- oracle bit-parity on three buckets ----
    mm = 1 << 30
    w = m.Worker(ctx, max_count, max_cells=max_cells, mesh_memory=mm)
    w.set_keep_splats(True)
    # (a corner, the centre and the far corner where they hold a surface; on the shells cloud the buckets with the most, the
    # median and the fewest triangles among those that have any)
    nonempty = sorted((i for i in range(len(per)) if per[i]["triangles"] > 0), key=lambda i: per[i]["triangles"])
    assert len(nonempty) >= 3
    sample = (0, 62, 124) if all(per[i]["triangles"] > 0 for i in (0, 62, 124)) else (nonempty[0], nonempty[len(nonempty) // 2], nonempty[-1])
    for i in sample:
        b = buckets[i]
        host = bucketed[b.first:b.first + b.count].cpu().numpy().view(m.SPLAT_DTYPE).reshape(-1)
        batches = w.process(pristine, b.first, b.count, b.low, b.num_vertices)
        exp_b, st = ob.bucket(host, 0, b.count, b.num_vertices, b.low, max_cells=max_cells, max_swathe=max_cells + 1, mesh_memory=mm)
        assert st["shipouts"] == len(batches) >= 1
        assert_batches_equal(batches, exp_b)
        del batches, exp_b


def farm_devices(n):
    env = os.environ.get("MLSGPU_TEST_DEVICES")
    if env:
        devs = [int(x) for x in env.split(",")]
        assert len(devs) == n
        return devs
    return [0] * n


def test_cfg4_shape_eight_device_groups():
    """cfg4's shape: 1024^3 grid, 125 buckets, EIGHT device groups behind one copy thread, every ship-out read back into
    the host welder.  Splats: 10 M on three concentric shells (a surface instead of cfg4's noise, so that the result has
    a topology to check; the oracle finishes sampled buckets in seconds)."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    g = 1024
    cloud = synth.shells_cloud(10_000_000, float(g - 1), 128.0, 2.5, 3.5, seed=4444)
    allb, buckets = synth.bucketize(cloud, g, 255)
    assert len(buckets) == 125
    max_cells = max(max(b.num_vertices) for b in buckets) - 1
    cap = max(b.count for b in buckets)
    devices = farm_devices(8)
    welder = m.HostMesher(0.001)
    farm = m.BucketFarm(devices, cap, workers_per_device=1, spare=1, max_cells=max_cells, mesh_memory=256 << 20)
    farm.set_host_output(64 << 20, welder)        # a small ring: workers wait for the mesher thread now and then
    for i, b in enumerate(buckets):
        room = farm.acquire(b.count)               # the loader's route: straight into pinned staging
        room[:] = allb[b.first:b.first + b.count]
        farm.push(b.count, b.low, b.num_vertices, 0)
    farm.finish()
    st = farm.stats()
    hs = farm.host_stats()
    assert st["buckets"] == 125 and sum(st["per_device"][:8]) == 125
    # greedy dispatch (src/workers.cpp:320-351: the group with the most unallocated capacity that can take an item): how many
    # groups see work depends on how fast items come back; with the welder off the mesher thread's critical path a group is
    # often free again before the loader has the next bucket, so only "more than one" is certain
    assert sum(1 for x in st["per_device"][:8] if x > 0) >= 2
    assert hs["meshes"] == st["shipouts"] > 0
    farm.close()
    assert welder.finalize() == 1
    ws = welder.stats()
    _, v, t = welder.chunk(0)
    assert ws["kept_components"] == 3 and ws["components"] >= 3              # the three shells, each in one piece
    assert ws["total_vertices"] < ws["vertices_added"]
    assert len(v) == ws["kept_vertices"] and len(t) == ws["kept_triangles"]
    # closed surfaces up to the few pin-holes random sampling leaves: every edge has at most two triangles and all but a
    # handful exactly two (a missing weld along ONE bucket face would leave thousands of open edges)
    edges = np.sort(np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]).astype(np.int64), axis=1)
    _, ecount = np.unique(edges[:, 0] * len(v) + edges[:, 1], return_counts=True)
    assert ecount.max() == 2
    open_edges = int((ecount == 1).sum())
    assert open_edges < 2000
    # the same buckets through ONE worker into the DEVICE sink: another weld (radix sort + union-find in HBM instead of the
    # host's hash map), another schedule -- the same mesh
    ctx = m.Context(devices[0])
    w = m.Worker(ctx, cap, max_cells=max_cells, mesh_memory=256 << 20)
    buf = m.DeviceBuffer(ctx, array=allb)
    sink = m.Mesher(ctx, 0.001)
    for b in buckets:
        w.process(buf, b.first, b.count, b.low, b.num_vertices, collector=sink.collector(ctx, 0))
    assert sink.finalize() == 1
    ds = sink.stats()
    for k in ("total_vertices", "threshold", "components", "kept_components", "kept_vertices", "kept_triangles",
              "vertices_added", "triangles_added"):
        assert ds[k] == ws[k], k
    dv = sink.chunk(0)["vertices"]

    def position_digest(a):
        a = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
        code = np.sort((a[:, 0] << np.uint64(42)) ^ (a[:, 1] << np.uint64(21)) ^ a[:, 2])
        return code
    assert np.array_equal(position_digest(dv), position_digest(v))            # the same welded vertices
    sink.close()
    buf.upload(allb)
    # oracle parity where the oracle is quick: the buckets with the fewest splats that still produce a mesh
    ref = allb.copy()
    done = 0
    for b in sorted(buckets, key=lambda b: b.count):
        if b.count < 2000:
            continue
        batches = w.process(buf, b.first, b.count, b.low, b.num_vertices)
        exp, _ = ob.bucket(ref, b.first, b.count, b.num_vertices, b.low, max_cells=max_cells, max_swathe=max_cells + 1,
                           mesh_memory=256 << 20)
        assert_batches_equal(batches, exp)
        done += 1 if batches else 0
        if done == 3:
            break
    assert done == 3
    del w, buf
    ctx.close()
    welder.close()


def test_cfg4_slab_full_density(ctx):
    """The per-GPU units of bench.py --gpus N: ALL EIGHT slabs (128 corner slices, 25 buckets each) of BASELINE
    configs[3]'s 200 M-splat uniform cloud at full density, one after the other on this GPU, generated in HBM exactly as
    bench.py does.  A slab's geometry depends on the job only through being its last slab (127 cell slices) or not
    (128): both variants of every slab are run, so every rank of an N = 2 / 4 / 8 job (and the one-slab N = 1 workload)
    has its totals and digest pinned in tests/golden/cfg4slab_uniform.json -- bench.py compares every rank's digest with
    them.  Oracle bit-parity on a bucket of slabs 0, 3 and 7; every one of the seven seams of the eight-GPU job checked:
    the external vertices two neighbouring slabs share (equal keys) are bit-identical from both sides."""
    import torch
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as mb, synth
    dev = torch.device("cuda", 0)
    cloud, g = synth.make_cloud_device("cfg4", dev)
    assert len(cloud) == 200_000_000 and g == 1024
    record_size("cfg4 slabs", "8 slabs x 2 variants of the %d-splat cloud at full density" % len(cloud))
    path = os.path.join(os.path.dirname(GOLDEN), "cfg4slab_uniform.json")
    writing = bool(os.environ.get("MLSGPU_WRITE_GOLDEN"))
    pinned = {} if writing else json.load(open(path))["slabs"]
    got_all = {}
    seam = {}                    # slab -> (keys, vertices) of the externals on its bottom / top corner slices (8-GPU job)
    parity = {0: 0, 3: 12, 7: 24}     # slab -> bucket for the oracle (a corner, the centre, the far corner)
    work = None
    for r in range(8):
        for variant, world in (("inner", 8), ("last", r + 1)):
            if synth.slab_variant(world, r) != variant:
                continue                                           # slab 7 of 8 is a last slab only
            boxes = synth.slab_boxes(g, world, r)
            bucketed, buckets = synth.bucketize_device(cloud, boxes)
            torch.cuda.synchronize()
            cz = buckets[0].num_vertices[2] - 1
            assert len(buckets) == 25 and cz == (127 if variant == "last" else 128) and buckets[0].low[2] == 128 * r
            assert sum(b.cells for b in buckets) == 1023 * 1023 * cz
            max_cells = max(max(b.num_vertices) for b in buckets) - 1
            max_count = max(b.count for b in buckets)
            nbytes = bucketed.numel() * 4
            pristine = m.DeviceBuffer(ctx, nbytes=nbytes, borrow=bucketed.data_ptr())
            if work is None or work.nbytes < nbytes:
                work = m.DeviceBuffer(ctx, nbytes=nbytes + (64 << 20))
            w = m.Worker(ctx, max_count, max_cells=max_cells, mesh_memory=4096 << 20)
            work.copy_from(pristine, nbytes)
            in_job = world == 8                                    # the eight-GPU job's slabs feed the seam check
            col = mb.ExternalCollector(ctx) if in_job else mb.ChecksumCollector(ctx)
            for b in buckets:
                w.process(work, b.first, b.count, b.low, b.num_vertices, collector=col)
            assert col.error is None
            got = dict(splats=int(bucketed.shape[0]), triangles=int(col.triangles), vertices=int(col.vertices),
                       external=int(col.external), shipouts=col.batches, digest=col.digest())
            got_all.setdefault(str(r), {})[variant] = got
            if not writing:
                assert got == pinned[str(r)][variant], (r, variant, got, pinned[str(r)][variant])
            if in_job:
                keys = np.concatenate(col.ext_keys)
                verts = np.concatenate(col.ext_vertices).view(np.uint32)
                z2 = (keys >> np.uint64(42)) & np.uint64((1 << 21) - 1)      # half-lattice z of the welding key
                edge = (z2 == np.uint64(256 * r)) | (z2 == np.uint64(256 * (r + 1)))
                seam[r] = (keys[edge], verts[edge])
                del keys, verts, col
            if in_job and r in parity:
                b = buckets[parity[r]]
                host = bucketed[b.first:b.first + b.count].cpu().numpy().view(m.SPLAT_DTYPE).reshape(-1)
                del w
                mm = 1 << 30
                w = m.Worker(ctx, max_count, max_cells=max_cells, mesh_memory=mm)
                work.copy_from(pristine, nbytes)
                batches = w.process(work, b.first, b.count, b.low, b.num_vertices)
                exp_b, st = ob.bucket(host, 0, b.count, b.num_vertices, b.low, max_cells=max_cells, max_swathe=max_cells + 1,
                                      mesh_memory=mm)
                assert st["shipouts"] == len(batches) >= 1
                assert_batches_equal(batches, exp_b)
                del batches, exp_b
            del w, pristine, bucketed
    # ---- the seven seams: what slab r emits on its top corner slice and slab r + 1 on its bottom one ----
    for r in range(7):
        z = np.uint64(256 * (r + 1))
        lo_k, lo_v = seam[r]
        hi_k, hi_v = seam[r + 1]
        sel = ((lo_k >> np.uint64(42)) & np.uint64((1 << 21) - 1)) == z
        a_k, a_v = lo_k[sel], lo_v[sel]
        sel = ((hi_k >> np.uint64(42)) & np.uint64((1 << 21) - 1)) == z
        b_k, b_v = hi_k[sel], hi_v[sel]
        # a slab's own buckets share edges of the seam plane: one copy of every key per side
        a_k, ia = np.unique(a_k, return_index=True)
        b_k, ib = np.unique(b_k, return_index=True)
        # (a cell with a non-finite corner emits nothing, and on the noise cloud a fifth of the seam's vertices have valid cells
        # on one side only: the shared ones are compared)
        both, ja, jb = np.intersect1d(a_k, b_k, assume_unique=True, return_indices=True)
        assert len(both) > 100_000 and len(both) > 0.6 * max(len(a_k), len(b_k)), (r, len(a_k), len(b_k), len(both))
        assert np.array_equal(a_v[ia][ja], b_v[ib][jb]), r                                   # the same vertices, bit for bit
    if writing:
        out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "cfg4slab_uniform.json")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        doc = dict(got_all["0"]["inner"])
        doc["slabs"] = got_all
        json.dump(doc, open(out, "w"), indent=1)


def cfg5_count():
    """10^9 splats need 28 GB of files in /dev/shm and a few GB of host memory on top; an eighth of the cloud otherwise (same
    grid, same route).  MLSGPU_CFG5_SPLATS overrides."""
    import shutil
    if os.environ.get("MLSGPU_CFG5_SPLATS"):
        return int(os.environ["MLSGPU_CFG5_SPLATS"])
    try:
        avail = int(next(l for l in open("/proc/meminfo") if l.startswith("MemAvailable")).split()[1]) * 1024
        free = shutil.disk_usage("/dev/shm").free
    except Exception:   # noqa: BLE001
        avail = free = 0
    return 1_000_000_000 if (avail > (96 << 30) and free > (40 << 30)) else 125_000_000


def test_cfg5_full_shape_from_files():
    import shutil
    import tempfile
    import torch
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as mb, farm as fm, synth
    n = cfg5_count()
    record_size("cfg5", "%d splats%s" % (n, "" if n == 1_000_000_000 else " (NOT the full 10^9: the box lacks the RAM / "
                                                                          "/dev/shm for 28 GB of files, or MLSGPU_CFG5_SPLATS is set)"))
    g = synth.CONFIGS["cfg5"]["grid"]
    assert g == 2048
    dev = torch.device("cuda", 0)
    tmp = tempfile.mkdtemp(prefix="mlsgpu_cfg5_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        paths = [os.path.join(tmp, "part%d.ply" % k) for k in range(8)]
        assert synth.write_cloud_ply(paths, "cfg5", dev, scale=n / synth.CONFIGS["cfg5"]["splats"]) == n
        assert sum(os.path.getsize(p) for p in paths) > 28 * n
        ctx = m.Context(0)
        fs = mb.FileSet(paths, buffer_size=768 << 20)
        assert len(fs) == n
        raw = m.DeviceBuffer(ctx, nbytes=n * 32)
        fs.load(ctx, raw, reader_threads=16)
        ctx.synchronize()
        # what arrived in HBM is the generator's cloud: positions, radii and normals bit for bit, quality = 1 / r^2 as the
        # reader computes it (src/fast_ply.cpp:334-350)
        for first in (0, n // 2 - 500, n - 1000):
            got = raw.download(m.SPLAT_DTYPE, 1000, offset=first * 32)
            exp = synth.uniform_cloud(1000, float(g - 1), 2.0, 3.0, synth.cloud_seed("cfg5"), first=first)
            for f in ("position", "radius", "normal"):
                np.testing.assert_array_equal(got[f].view(np.uint32), exp[f].view(np.uint32))
            np.testing.assert_allclose(got["quality"], 1.0 / (exp["radius"].astype(np.float64) ** 2), rtol=3e-7)
        ext = (0, g - 1, 0, g - 1, 0, g - 1)
        ref0 = (0.0, 0.0, 0.0)
        bp = dict(max_splats=2097152, max_cells=255, chunk_cells=0, micro_cells=63, max_split=1 << 30)
        devices = farm_devices(8)
        # ---- the partition: every leaf within the caps, the leaves tile the grid ----
        picked = {}

        def note(leaf, d_ids):
            picked[len(picked)] = None
        leaves = mb.bucket_cloud(ctx, raw, n, ref0, 1.0, ext, on_bucket=note, **bp)
        assert len(leaves) >= 512
        cover = np.zeros((33, 33, 33), np.int32)          # 63-cell microblocks: 2047 = 32 * 63 + 31
        for l in leaves:
            e = l["extents"]
            assert l["num_splats"] <= bp["max_splats"] and max(e[1] - e[0], e[3] - e[2], e[5] - e[4]) <= 255
            assert all(e[2 * a] % 63 == 0 and (e[2 * a + 1] % 63 == 0 or e[2 * a + 1] == g - 1) for a in range(3))
            cover[e[0] // 63:-(-e[1] // 63), e[2] // 63:-(-e[3] // 63), e[4] // 63:-(-e[5] // 63)] += 1
        assert cover.min() == 1 and cover.max() == 1
        assert sum(fm.leaf_cells(l) for l in leaves) == (g - 1) ** 3
        pmax = max(l["num_splats"] for l in leaves)
        pcells = max(max(l["extents"][2 * a + 1] - l["extents"][2 * a] for a in range(3)) for l in leaves)
        # ---- files -> ... -> eight device groups: every bucket's ship-outs counted and checksummed on the device ----
        bfarm = m.BucketFarm(devices, pmax, workers_per_device=1, spare=1, max_cells=pcells, mesh_memory=1 << 30,
                             collect="checksum")
        leaves2 = fm.partition_to_farm(ctx, bfarm, devices[0], raw, n, ref0, 1.0, ext, bp)
        bfarm.finish()
        assert bfarm.error is None
        assert [l["extents"] for l in leaves2] == [l["extents"] for l in leaves]
        st = bfarm.stats()
        assert st["buckets"] == len(leaves) and sum(st["per_device"][:8]) == len(leaves)
        assert min(st["per_device"][:8]) > 0 and st["in_flight_max"] >= 2
        assert sum(len(v) for v in bfarm.sums.values()) == st["shipouts"]
        assert sum(r[0] for v in bfarm.sums.values() for r in v) == st["vertices"]
        assert sum(r[1] for v in bfarm.sums.values() for r in v) == st["triangles"]
        got = dict(splats=n, buckets=len(leaves), bucket_splats=int(sum(l["num_splats"] for l in leaves)),
                   triangles=st["triangles"], vertices=st["vertices"], external=st["external"], shipouts=st["shipouts"],
                   digest=bfarm.digest())
        per_leaf = {k: list(v) for k, v in bfarm.sums.items()}
        bfarm.close()
        path = os.path.join(os.path.dirname(GOLDEN), "cfg5_uniform.json")
        if os.environ.get("MLSGPU_WRITE_GOLDEN"):
            out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "cfg5_uniform_%d.json" % n)
            os.makedirs(os.path.dirname(out), exist_ok=True)
            json.dump(got, open(out, "w"), indent=1)
        else:
            exp = json.load(open(path))[str(n)]
            assert got == {k: exp[k] for k in got}, (got, exp)
        # ---- three leaves against the oracle, bit for bit (a corner leaf, an edge leaf, a face leaf: the thin ones, so that
        # the CPU finishes in seconds), and their per-leaf records against what the farm saw ----
        order = sorted(range(len(leaves)), key=lambda i: (fm.leaf_cells(leaves[i]), i))
        sample = [order[0], order[len(order) // 16], order[len(order) // 4]]
        # ... plus the neighbours of the middle leaf for the cross-bucket check
        mid = min(range(len(leaves)), key=lambda i: sum(abs(leaves[i]["extents"][2 * a] - 1008) for a in range(3)))
        em = leaves[mid]["extents"]
        hood = [i for i, l in enumerate(leaves)
                if all(l["extents"][2 * a] <= em[2 * a + 1] and l["extents"][2 * a + 1] >= em[2 * a] for a in range(3))
                and i not in sample][:8]
        wanted = set(sample) | set(hood)
        ids = {}
        counter = [0]

        def grab(leaf, d_ids):
            i = counter[0]
            counter[0] += 1
            if i in wanted:
                staged = m.DeviceBuffer(ctx, nbytes=max(leaf["num_splats"], 1) * 32)
                mb.bucket_load(ctx, raw, d_ids, leaf["num_splats"], ref0, 1.0, ext, staged)
                ctx.synchronize()
                ids[i] = staged
        mb.bucket_cloud(ctx, raw, n, ref0, 1.0, ext, on_bucket=grab, **bp)
        mm = 1 << 30
        w = m.Worker(ctx, pmax, max_cells=pcells, mesh_memory=mm)
        for i in sample:
            low, nv = fm.leaf_geometry(leaves[i], ext)
            cnt = leaves[i]["num_splats"]
            host = ids[i].download(m.SPLAT_DTYPE, cnt)
            col = mb.ChecksumCollector(ctx)
            w.process(ids[i], 0, cnt, low, nv, collector=col)
            assert col.sums == per_leaf.get(i, []), i
            ids[i].upload(host)
            batches = w.process(ids[i], 0, cnt, low, nv)
            exp_b, _ = ob.bucket(host.copy(), 0, cnt, nv, low, max_cells=pcells, max_swathe=pcells + 1, mesh_memory=mm)
            assert_batches_equal(batches, exp_b)
            del batches, exp_b
        # ---- cross-bucket agreement: vertices that neighbouring buckets share (same key) are bit-identical ----
        col = mb.ExternalCollector(ctx)
        for i in hood:
            low, nv = fm.leaf_geometry(leaves[i], ext)
            w.process(ids[i], 0, leaves[i]["num_splats"], low, nv, collector=col)
        keys = np.concatenate(col.ext_keys)
        verts = np.concatenate(col.ext_vertices).view(np.uint32)
        o = np.argsort(keys, kind="stable")
        keys, verts = keys[o], verts[o]
        same = keys[1:] == keys[:-1]
        assert same.sum() > (1000 * (len(hood) - 1) if n >= 1_000_000_000 else 0)    # a sparse cloud leaves little surface
        assert np.all(verts[1:][same] == verts[:-1][same])
        del w, ids
        fs.close()
        ctx.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
