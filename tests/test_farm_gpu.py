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
@pytest.mark.parametrize("workers", [1, 2])
def test_farm_matches_oracle(workers):
    """27 buckets submitted one by one; several share a device item (batching); every ship-out of every chunk
    equals the oracle's, whatever worker thread processed it."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud = synth.shells_cloud(120_000, 95.0, 16.0, 1.5, 2.5, seed=321)
    allb, buckets = synth.bucketize(cloud, 96, 32)
    cap = 3 * max(b.count for b in buckets) // 2          # room for more than one bucket per item
    farm = m.BucketFarm([0], cap, workers_per_device=workers, collect=True, max_cells=63)
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
