"""The bucket farm (CopyGroup + DeviceWorkerGroup restatement): batching, pinned staging, worker threads."""
import numpy as np
import pytest

import oracle_binding as ob


def test_transform_splats_matches_grid_world_to_vertex():
    """Grid::worldToVertex + radius scaling (src/grid.cpp:99-106, src/bucket_loader.cpp:77-85), bit for bit."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    s = synth.sphere_cloud(1000, (3.0, -2.0, 7.5), 4.0, 0.1, 0.3, seed=8)
    ref = np.array([0.25, -1.5, 2.0], np.float32)
    spacing = np.float32(0.05)
    low = (-10, 7, 3)
    exp = s.copy()
    inv = np.float32(1.0) / spacing
    for a in range(3):
        exp["position"][:, a] = (s["position"][:, a] - ref[a]) * inv - np.float32(low[a])
    exp["radius"] = s["radius"] * inv
    m.binding.transform_splats(s, ref, float(spacing), low)
    np.testing.assert_array_equal(s.view(np.uint32), exp.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("workers,lanes", [(1, 1), (2, 1), (1, 4), (2, 3)])
def test_farm_matches_oracle(workers, lanes):
    """27 buckets submitted one by one; several share a device item (batching); every ship-out of every chunk
    equals the oracle's, whatever worker thread processed it -- bucket by bucket, or with the buckets of an item taken
    through the path `lanes` at a time (mlsgpu_hip_farm_set_batch)."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud = synth.shells_cloud(120_000, 95.0, 16.0, 1.5, 2.5, seed=321)
    allb, buckets = synth.bucketize(cloud, 96, 32)
    cap = (3 if lanes == 1 else 9) * max(b.count for b in buckets) // 2          # room for more than one bucket per item
    farm = m.BucketFarm([0], cap, workers_per_device=workers, collect=True, max_cells=63)
    farm.set_batch(lanes)
    for i, b in enumerate(buckets):
        if i % 2 == 0:
            farm.submit(allb[b.first:b.first + b.count], b.low, b.num_vertices, i)
        else:
            # the loader's route (CopyGroup::get / push): write the bucket straight into pinned staging
            room = farm.acquire(b.count)
            room[:] = allb[b.first:b.first + b.count]
            farm.push(b.count, b.low, b.num_vertices, i)
    farm.finish()
    st = farm.stats()
    assert st["buckets"] == 27 and st["splats"] == len(allb) and st["h2d_bytes"] == 32 * len(allb)
    assert st["items"] < 27                                 # batching happened
    assert st["per_device"][0] == 27
    ref = allb.copy()
    total_tris = 0
    for i, b in enumerate(buckets):
        exp, _ = ob.bucket(ref, b.first, b.count, b.num_vertices, b.low, max_cells=63, max_swathe=64,
                           mesh_memory=63 * 63 * 2 * 872)
        got = farm.meshes.get(i, [])
        assert len(got) == len(exp), i
        for g, e in zip(got, exp):
            assert g["num_internal"] == e["num_internal"]
            np.testing.assert_array_equal(g["vertices"].view(np.uint32), e["vertices"].view(np.uint32))
            np.testing.assert_array_equal(g["triangles"], e["triangles"])
            np.testing.assert_array_equal(g["keys"][g["num_internal"]:], e["keys"][e["num_internal"]:])
            total_tris += len(g["triangles"])
    assert total_tris == st["triangles"] > 0
    farm.close()


@pytest.mark.gpu
def test_farm_rejects_oversized_bucket_and_reports_worker_errors():
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud = synth.sphere_cloud(1000, (30.0, 30.0, 30.0), 10.0, 1.0, 2.0, seed=3)
    farm = m.BucketFarm([0], 500, max_cells=63)
    with pytest.raises(m.LengthError):
        farm.submit(cloud, (0, 0, 0), (64, 64, 64), 0)      # more splats than a device item holds
    farm.submit(cloud[:400], (0, 0, 0), (300, 64, 64), 1)   # bucket larger than the octree: fails in the worker
    with pytest.raises(m.LengthError):
        farm.finish()
    farm.close()


@pytest.mark.gpu
def test_farm_dispatches_over_device_groups():
    """Two DeviceWorkerGroups (here both on GPU 0, as a one-GPU stand-in for two GPUs): the copy thread sends each item to
    the group with the most unallocated capacity (src/workers.cpp:320-351), both groups get work, and every chunk's
    meshes are the oracle's whichever group produced them."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud = synth.shells_cloud(120_000, 95.0, 16.0, 1.5, 2.5, seed=99)
    allb, buckets = synth.bucketize(cloud, 96, 32)
    cap = max(b.count for b in buckets)
    farm = m.BucketFarm([0, 0], cap, workers_per_device=1, collect=True, max_cells=63)
    for i, b in enumerate(buckets):
        farm.submit(allb[b.first:b.first + b.count], b.low, b.num_vertices, i)
    farm.finish()
    st = farm.stats()
    assert st["buckets"] == 27 and st["per_device"][0] + st["per_device"][1] == 27
    assert st["per_device"][0] > 0 and st["per_device"][1] > 0
    ref = allb.copy()
    for i, b in enumerate(buckets):
        exp, _ = ob.bucket(ref, b.first, b.count, b.num_vertices, b.low, max_cells=63, max_swathe=64,
                           mesh_memory=63 * 63 * 2 * 872)
        got = farm.meshes.get(i, [])
        assert len(got) == len(exp), i
        for g, e in zip(got, exp):
            np.testing.assert_array_equal(g["vertices"].view(np.uint32), e["vertices"].view(np.uint32))
            np.testing.assert_array_equal(g["triangles"], e["triangles"])
    farm.close()


def _oracle_meshes(allb, buckets, chunk_of=lambda i: 0):
    ref = allb.copy()
    meshes = []
    for i, b in enumerate(buckets):
        batches, _ = ob.bucket(ref, b.first, b.count, b.num_vertices, b.low, max_cells=63, max_swathe=64,
                               mesh_memory=63 * 63 * 2 * 872)
        for g in batches:
            meshes.append(dict(chunk=chunk_of(i), vertices=g["vertices"], num_internal=g["num_internal"],
                               keys=g["keys"][g["num_internal"]:], triangles=g["triangles"]))
    return meshes


@pytest.mark.gpu
@pytest.mark.parametrize("ring_kb", [512, 65536, 0])
def test_farm_host_output_welds_across_device_groups(ring_kb):
    """The reference's route for several GPUs: every device group's ship-outs are read back through the pinned circular
    buffer (OutputGeneratorBuilder::Functor, src/workers.h:488-509) and welded by ONE host mesher
    (src/mesher.cpp:220-469).  Two groups (both on GPU 0 here), two chunks whose blocks arrive interleaved; a 512 KB ring
    makes the workers wait for the mesher thread and wraps many times.  ring_kb = 0: no ring -- the read-backs land in
    page-locked memory of the welder and are adopted without a copy (mlsgpu_hip_farm_set_host_landing), over two jobs
    (a second welder takes the farm's next job).  The result equals the oracle sink fed with the oracle's bucket meshes."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import mesher_oracle as mo
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud = synth.shells_cloud(120_000, 95.0, 16.0, 1.5, 2.5, seed=321)
    allb, buckets = synth.bucketize(cloud, 96, 32)
    cap = max(b.count for b in buckets)
    welder = m.HostMesher(0.02)
    farm = m.BucketFarm([0, 0], cap, workers_per_device=2, max_cells=63)
    chunk_of = lambda i: i % 2                                      # noqa: E731
    if ring_kb == 0:
        first = m.HostMesher(0.02)
        farm.set_host_landing(first)                                # job 1 lands in another welder's memory ...
        for i, b in enumerate(buckets):
            farm.submit(allb[b.first:b.first + b.count], b.low, b.num_vertices, chunk_of(i))
        farm.finish()
        assert first.landing_pinned()
        farm.set_host_landing(welder)                               # ... job 2 in this one's
    else:
        farm.set_host_output(ring_kb << 10, welder)
    before = farm.stats()
    for i, b in enumerate(buckets):
        farm.submit(allb[b.first:b.first + b.count], b.low, b.num_vertices, chunk_of(i))
    farm.finish()
    st, hs = farm.stats(), farm.host_stats()
    assert st["per_device"][0] > 0 and st["per_device"][1] > 0
    assert hs["meshes"] == st["shipouts"] and hs["bytes"] > 0
    if ring_kb == 512:
        assert hs["bytes"] > 4 * (ring_kb << 10)                    # the ring wrapped
    if ring_kb == 0:
        assert hs["ring_waits"] == 0 and st["shipouts"] == 2 * before["shipouts"]
        n1 = first.finalize()
        s1 = first.stats()
    farm.close()
    n = welder.finalize()
    stats = welder.stats()
    exp, exp_stats = mo.mesh_sink(_oracle_meshes(allb, buckets, chunk_of), 0.02)
    for k in ("total_vertices", "threshold", "components", "kept_components", "kept_vertices", "kept_triangles"):
        assert stats[k] == exp_stats[k], k
    got = sorted((welder.chunk(i) for i in range(n)), key=lambda c: c[0])
    assert [c for c, _, _ in got] == sorted(c for c, _, _ in exp)
    for (c, v, t), (ec, ev, et) in zip(got, sorted(exp, key=lambda c: c[0])):
        assert mo.isomorphic(v, t, ev, et)
    if ring_kb == 0:
        assert n1 == n and all(s1[k] == stats[k] for k in exp_stats)         # both jobs welded the same cloud
        first.close()
    welder.close()


@pytest.mark.gpu
def test_farm_host_output_python_sink_and_oversized_mesh():
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud = synth.shells_cloud(120_000, 95.0, 16.0, 1.5, 2.5, seed=321)
    allb, buckets = synth.bucketize(cloud, 96, 32)
    cap = max(b.count for b in buckets)
    got = {}
    farm = m.BucketFarm([0], cap, workers_per_device=2, max_cells=63)
    farm.set_host_output(8 << 20, lambda device, chunk, batch: got.setdefault(chunk, []).append(batch))
    for i, b in enumerate(buckets):
        farm.submit(allb[b.first:b.first + b.count], b.low, b.num_vertices, i)
    farm.finish()
    farm.close()
    ref = allb.copy()
    for i, b in enumerate(buckets):
        exp, _ = ob.bucket(ref, b.first, b.count, b.num_vertices, b.low, max_cells=63, max_swathe=64,
                           mesh_memory=63 * 63 * 2 * 872)
        assert len(got.get(i, [])) == len(exp), i
        for g, e in zip(got.get(i, []), exp):
            assert g["num_internal"] == e["num_internal"]
            np.testing.assert_array_equal(g["vertices"].view(np.uint32), e["vertices"].view(np.uint32))
            np.testing.assert_array_equal(g["triangles"], e["triangles"])
            np.testing.assert_array_equal(g["keys"][g["num_internal"]:], e["keys"][e["num_internal"]:])
    # a ship-out that does not fit the ring is an error, reported by finish, not a hang
    farm = m.BucketFarm([0], cap, max_cells=63)
    farm.set_host_output(4096, None)
    b = max(buckets, key=lambda b: b.count)
    farm.submit(allb[b.first:b.first + b.count], b.low, b.num_vertices, 0)
    with pytest.raises(m.LengthError):
        farm.finish()
    farm.close()


@pytest.mark.gpu
def test_device_sink_behind_the_farm_and_peer_routes():
    """Ship-outs appended to the device mesher by the farm's workers in C (mlsgpu_hip_mesher_farm_output), two chunks
    interleaved over two device groups; then the same with the cross-GPU routes forced on one GPU: buckets gathered on the
    cloud's device and peer-copied into the item (mlsgpu_hip_farm_submit_device), ship-outs peer-copied into the mesher."""
    import os
    import subprocess
    import sys
    code = r"""
import os, sys
sys.path[:0] = [%r, %r, %r]
import numpy as np
import mesher_oracle as mo
import oracle_binding as ob
import mlsgpu_amd as m
from mlsgpu_amd import binding as mb, synth
from test_farm_gpu import _oracle_meshes
cloud = synth.shells_cloud(120_000, 95.0, 16.0, 1.5, 2.5, seed=321)
allb, buckets = synth.bucketize(cloud, 96, 32)
cap = max(b.count for b in buckets)
ctx = m.Context(0)
sink = m.Mesher(ctx, 0.02)
farm = m.BucketFarm([0, 0], cap, workers_per_device=2, max_cells=63, sink=sink)
raw = m.DeviceBuffer(ctx, array=allb)
ext = (0, 95, 0, 95, 0, 95)
for i, b in enumerate(buckets):
    if i %% 3 == 0:
        farm.submit(allb[b.first:b.first + b.count], b.low, b.num_vertices, i %% 2)
    else:
        ids = m.DeviceBuffer(ctx, array=np.arange(b.first, b.first + b.count, dtype=np.uint32))
        farm.submit_device(0, raw, ids.ptr, b.count, (0.0, 0.0, 0.0), 1.0, ext, b.low, b.num_vertices, i %% 2)
farm.finish()
assert min(farm.stats()["per_device"][:2]) > 0
farm.close()
n = sink.finalize()
stats = sink.stats()
exp, exp_stats = mo.mesh_sink(_oracle_meshes(allb, buckets, lambda i: i %% 2), 0.02)
for k in ("total_vertices", "threshold", "components", "kept_components", "kept_vertices", "kept_triangles"):
    assert stats[k] == exp_stats[k], k
got = sorted((sink.chunk(i) for i in range(n)), key=lambda c: c["chunk"])
for c, (ec, ev, et) in zip(got, sorted(exp, key=lambda c: c[0])):
    assert c["chunk"] == ec and mo.isomorphic(c["vertices"], c["triangles"], ev, et)
sink.reset()
assert sink.finalize() == 0
print("ok")
""" % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
       os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    for env in ({}, {"MLSGPU_HIP_FARM_FORCE_PEER": "1", "MLSGPU_HIP_MESHER_FORCE_PEER": "1"}):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


@pytest.mark.gpu
def test_submit_device_keeps_many_leaves_in_flight_and_sink_plus_host_output():
    """Round 3 (ADVICE): (1) eight device groups fed from ONE resident cloud through the peer route --
    every leaf gathered into a slot of the scratch ring and peer-copied on its target's copy stream, ordered by events on
    the GPUs, so the copy side runs ahead of the workers: at least as many device items in flight as there are groups;
    (2) a device sink AND host output on the same farm with the peer routes forced: after the peer append the worker's
    read-back event must still be created on the worker's device (mlsgpu_hip_mesher_add restores it).  Both welds equal
    the oracle's."""
    import os
    import subprocess
    import sys
    code = r"""
import os, sys
sys.path[:0] = [%r, %r, %r]
import numpy as np
import mesher_oracle as mo
import mlsgpu_amd as m
from mlsgpu_amd import synth
from test_farm_gpu import _oracle_meshes
cloud = synth.shells_cloud(120_000, 95.0, 16.0, 1.5, 2.5, seed=321)
allb, buckets = synth.bucketize(cloud, 96, 32)
cap = max(b.count for b in buckets)
exp, exp_stats = mo.mesh_sink(_oracle_meshes(allb, buckets), 0.02)
ctx = m.Context(0)
raw = m.DeviceBuffer(ctx, array=allb)
ids = [m.DeviceBuffer(ctx, array=np.arange(b.first, b.first + b.count, dtype=np.uint32)) for b in buckets]
ext = (0, 95, 0, 95, 0, 95)
sink = m.Mesher(ctx, 0.02)
welder = m.HostMesher(0.02)
farm = m.BucketFarm([0] * 8, cap, workers_per_device=1, max_cells=63, sink=sink)
farm.set_host_output(16 << 20, welder)
for i, b in enumerate(buckets):
    farm.submit_device(0, raw, ids[i].ptr, b.count, (0.0, 0.0, 0.0), 1.0, ext, b.low, b.num_vertices, 0)
farm.finish()
st = farm.stats()
assert st["in_flight_max"] >= 8, st
assert sum(1 for x in st["per_device"][:8] if x > 0) >= 4, st
for w in (sink, welder):
    assert w.finalize() == 1
    stats = w.stats()
    for k in ("total_vertices", "threshold", "components", "kept_components", "kept_vertices", "kept_triangles"):
        assert stats[k] == exp_stats[k], (k, stats[k], exp_stats[k])
got = sink.chunk(0)
assert mo.isomorphic(got["vertices"], got["triangles"], exp[0][1], exp[0][2])
hv = welder.chunk(0)
assert mo.isomorphic(hv[1], hv[2], exp[0][1], exp[0][2])
farm.close()
print("ok")
""" % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
       os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    env = dict(os.environ, MLSGPU_HIP_FARM_FORCE_PEER="1", MLSGPU_HIP_MESHER_FORCE_PEER="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


@pytest.mark.gpu
def test_farm_on_two_real_devices():
    """the farm with its device groups on DIFFERENT GPUs (staging events belong to the item's device; a
    bucket resident on GPU 0 reaches GPU 1's group by a peer copy; ship-outs of both GPUs are welded by the host welder and,
    separately, appended by peer copy to GPU 0's device sink).  Skipped on a one-GPU box; the same routes run there with
    both groups on GPU 0 (tests above)."""
    import os
    import sys
    import mlsgpu_amd as m
    count = m.binding.C.c_int(0)
    m.binding.check(m.lib().mlsgpu_hip_device_count(m.binding.C.byref(count)))
    if count.value < 2:
        pytest.skip("needs two GPUs")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import mesher_oracle as mo
    from mlsgpu_amd import synth
    cloud = synth.shells_cloud(120_000, 95.0, 16.0, 1.5, 2.5, seed=321)
    allb, buckets = synth.bucketize(cloud, 96, 32)
    cap = max(b.count for b in buckets)
    exp, exp_stats = mo.mesh_sink(_oracle_meshes(allb, buckets), 0.02)
    ctx = m.Context(0)
    raw = m.DeviceBuffer(ctx, array=allb)
    for route in ("host", "device", "both"):
        welder = m.HostMesher(0.02) if route == "host" else m.Mesher(ctx, 0.02)
        farm = m.BucketFarm([0, 1], cap, workers_per_device=2, max_cells=63, sink=None if route == "host" else welder)
        if route == "host":
            farm.set_host_output(16 << 20, welder)
        if route == "both":
            # after a peer append into GPU 0's sink, GPU 1's worker creates its read-back event on GPU 1
            farm.set_host_output(16 << 20, None)
        for i, b in enumerate(buckets):
            if i % 2 == 0:
                farm.submit(allb[b.first:b.first + b.count], b.low, b.num_vertices, 0)
            else:
                ids = m.DeviceBuffer(ctx, array=np.arange(b.first, b.first + b.count, dtype=np.uint32))
                farm.submit_device(0, raw, ids.ptr, b.count, (0.0, 0.0, 0.0), 1.0, (0, 95, 0, 95, 0, 95), b.low, b.num_vertices, 0)
        farm.finish()
        st = farm.stats()
        assert st["per_device"][0] > 0 and st["per_device"][1] > 0
        farm.close()
        assert welder.finalize() == 1
        stats = welder.stats()
        for k in ("total_vertices", "threshold", "components", "kept_components", "kept_vertices", "kept_triangles"):
            assert stats[k] == exp_stats[k], (route, k)
        got = welder.chunk(0)
        v, t = (got[1], got[2]) if route == "host" else (got["vertices"], got["triangles"])
        assert mo.isomorphic(v, t, exp[0][1], exp[0][2]), route
        welder.close()
    ctx.close()


@pytest.mark.gpu
def test_farm_reports_placement_and_clocks():
    """mlsgpu_hip_farm_placement / _copy_clock / _worker_clock on a real device: one copy side for one GPU, its staging where
    the GPU's node is (when sysfs says), and clocks that add up to what was submitted."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud = synth.shells_cloud(200_000, 95.0, 16.0, 1.5, 2.5, seed=5)
    allb, buckets = synth.bucketize(cloud, 96, 32)
    cap = max(b.count for b in buckets)
    farm = m.BucketFarm([0], cap, workers_per_device=2, spare=3, max_cells=63, copy_threads=4, staging_buffers=3)
    farm.set_host_output(64 << 20, None)
    pl = farm.placement()
    assert len(pl["sides"]) == 1 and pl["devices"] == [dict(device=0, node=pl["devices"][0]["node"], side=0)]
    side = pl["sides"][0]
    assert side["staging_buffers"] == 3 and side["copy_threads"] == 4
    node = m.lib().mlsgpu_hip_device_node(0)
    assert pl["devices"][0]["node"] == node
    if pl["nodes"] > 1 and node >= 0:
        assert side["node"] == node and side["staging_node"] == node and pl["ring_node"] == node
    n = 0
    for b in buckets:
        if b.count:
            farm.submit(allb[b.first:b.first + b.count], b.low, b.num_vertices, n)
            n += 1
    farm.finish()
    cc, wc = farm.copy_clock(), farm.worker_clock()
    assert cc["copies"] == farm.stats()["items"] >= 1 and cc["h2d_s"] > 0 and cc["fill_s"] > 0 and cc["span_s"] >= cc["h2d_s"] * 0.5
    assert wc["buckets"] == n and 1 <= wc["launch_sets"] <= n and wc["busy_s"] > 0
    farm.close()


@pytest.mark.gpu
def test_farm_keep_splats_digest_equals_mutating_worker():
    """The farm's workers never write 1/r^2 back into their items (keep_splats: processCorners takes the reciprocal while
    staging); a plain Worker with the reference's MUTATING build (kernels/octree.cl:193) on the same buckets must produce the
    same ship-outs -- compared through the device-side digest, bucket by bucket."""
    import mlsgpu_amd as m
    from mlsgpu_amd import binding as mb, synth
    cloud = synth.shells_cloud(90_000, 95.0, 16.0, 1.5, 2.5, seed=99)
    allb, buckets = synth.bucketize(cloud, 96, 48)
    cap = 2 * max(b.count for b in buckets)
    farm = m.BucketFarm([0], cap, workers_per_device=2, collect="checksum", max_cells=63)
    farm.set_batch(3)
    for i, b in enumerate(buckets):
        farm.submit(allb[b.first:b.first + b.count], b.low, b.num_vertices, i)
    farm.finish()
    farm_digests = {i: mb.digest_of_sums(farm.sums.get(i, [])) for i in range(len(buckets))}
    farm.close()
    ctx = m.Context(0)
    w = m.Worker(ctx, max(b.count for b in buckets), max_cells=63)
    w.set_keep_splats(False)
    buf = m.DeviceBuffer(ctx, array=allb)
    nonempty = 0
    for i, b in enumerate(buckets):
        col = w.process(buf, b.first, b.count, b.low, b.num_vertices, collector=mb.ChecksumCollector(ctx))
        assert col.digest() == farm_digests[i], i
        nonempty += col.triangles > 0
    assert nonempty >= 4
    # ... and the mutating worker did write 1/r^2 where the farm's items keep the radius
    after = buf.download(np.float32).reshape(-1, 8)[:, 3]
    assert not np.array_equal(after, allb["radius"])
