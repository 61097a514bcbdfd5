"""HIP mesh sink (mlsgpu_hip_mesher_*) against the reference's mesher vectors and the oracle, up to isomorphism --
the comparison the reference's own tests make (test/test_mesher.cpp:401-460)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import mesher_oracle as mo  # noqa: E402
from mesher_cases import CASES, random_meshes  # noqa: E402

pytestmark = pytest.mark.gpu


def chunk_number(chunk, seen):
    return seen.setdefault(chunk, len(seen))


def run_hip(meshes, prune=0.0, background=False):
    import mlsgpu_amd as m
    ctx = m.Context(0)
    mesher = m.Mesher(ctx, prune)
    if background:
        mesher.set_background(True)
    seen = {}
    for mesh in meshes:
        mesher.add(chunk_number(mesh["chunk"], seen), mesh["vertices"], mesh["num_internal"], mesh["keys"], mesh["triangles"])
    n = mesher.finalize()
    out = [mesher.chunk(i) for i in range(n)]
    stats = mesher.stats()
    mesher.close()
    ctx.close()
    back = {v: k for k, v in seen.items()}
    return [(back[c["chunk"]], c["vertices"], c["triangles"]) for c in out], stats


@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_case(name):
    case = CASES[name]
    out, stats = run_hip(case["meshes"], case.get("prune", 0.0))
    assert [c for c, _, _ in out] == [c for c, _, _ in case["expected"]]
    for (_, v, t), (_, ev, et) in zip(out, case["expected"]):
        assert mo.isomorphic(v, t, ev, et), name
    for k, val in case.get("stats", {}).items():
        assert stats[k] == val


def test_block_order_does_not_matter():
    case = CASES["weld"]
    out, _ = run_hip(case["meshes"][::-1])
    (_, v, t), (_, ev, et) = out[0], case["expected"][0]
    assert mo.isomorphic(v, t, ev, et)


def test_chunks_may_arrive_interleaved():
    """Two workers and several chunks deliver blocks of different chunks interleaved; OOCMesher::add accepts any order
    (it indexes chunks[chunkId.gen], src/mesher.cpp:380-384).  Output chunks are in order of first arrival."""
    meshes = random_meshes(7, blocks=12, chunks=3)
    order = [0, 4, 8, 1, 5, 9, 2, 10, 6, 3, 7, 11]            # chunks 0,1,2,0,1,2,...
    mixed = [meshes[i] for i in order]
    exp, exp_stats = mo.mesh_sink(mixed, 0.02)
    out, stats = run_hip(mixed, 0.02)
    for k in ("total_vertices", "threshold", "components", "kept_components", "kept_vertices", "kept_triangles"):
        assert stats[k] == exp_stats[k], k
    assert [c for c, _, _ in out] == [c for c, _, _ in exp] == [0, 1, 2]
    for (_, v, t), (_, ev, et) in zip(out, exp):
        assert mo.isomorphic(v, t, ev, et)


def test_reserve_between_adds_keeps_pending_appends():
    """mlsgpu_hip_mesher_reserve in the middle of a job: growing an arena moves it, so the same-device appends still
    running on their producers' streams must land first.  Worker ship-outs are appended without a host
    wait, then the arenas are grown several times with nothing finalized in between."""
    import mlsgpu_amd as m
    from mlsgpu_amd import synth
    cloud = synth.shells_cloud(120_000, 95.0, 16.0, 1.5, 2.5, seed=321)
    allb, buckets = synth.bucketize(cloud, 96, 32)

    def run(grow):
        ctx = m.Context(0)
        dev = m.DeviceBuffer(ctx, array=allb)
        worker = m.Worker(ctx, max(bk.count for bk in buckets), max_cells=63)
        mesher = m.Mesher(ctx, 0.02)
        room = 1 << 12
        for bk in buckets:
            worker.process(dev, bk.first, bk.count, bk.low, bk.num_vertices, collector=mesher.collector(ctx, 0))
            if grow:
                room = min(room * 2, 1 << 22)                  # 4 K -> 4 M vertices: every arena moves several times
                mesher.reserve(room, 2 * room, room // 4)      # no finalize, no synchronize in between
        assert mesher.finalize() == 1
        got, stats = mesher.chunk(0), mesher.stats()
        mesher.close()
        del worker, dev
        ctx.close()
        return got, stats

    a, sa = run(False)
    b, sb = run(True)
    assert sa == sb
    assert np.array_equal(a["vertices"].view(np.uint32), b["vertices"].view(np.uint32))
    assert np.array_equal(a["triangles"], b["triangles"])


def test_peer_route_appends(monkeypatch):
    """The cross-GPU append (peer copies on the producer's stream, index fix-ups on the mesher's device) forced on one
    GPU: same result as the local route."""
    import subprocess
    code = ("import os,sys; sys.path[:0]=[%r,%r,%r]; os.environ['MLSGPU_HIP_MESHER_FORCE_PEER']='1';"
            "import test_gpu_mesher as t; t.test_random_sheets_match_oracle(2, 0.01); t.test_chunks_may_arrive_interleaved()"
            % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
               os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")))
    subprocess.check_call([sys.executable, "-c", code])


@pytest.mark.parametrize("seed,prune", [(1, 0.0), (2, 0.01), (3, 0.05), (4, 0.3)])
def test_random_sheets_match_oracle(seed, prune):
    meshes = random_meshes(seed)
    exp, exp_stats = mo.mesh_sink(meshes, prune)
    out, stats = run_hip(meshes, prune, background=seed % 2 == 0)     # the background mode (held-back union-find) too
    for k in ("total_vertices", "threshold", "components", "kept_components", "kept_vertices", "kept_triangles"):
        assert stats[k] == exp_stats[k], k
    assert [c for c, _, _ in out] == [c for c, _, _ in exp]
    for (_, v, t), (_, ev, et) in zip(out, exp):
        assert mo.isomorphic(v, t, ev, et)


def test_worker_meshes_weld_into_a_closed_surface(tmp_path):
    """27 buckets of a shells cloud through workers into the mesher: the welded result equals the oracle sink fed with
    the oracle's bucket meshes, external vertices are shared (fewer vertices than the sum), and the PLY round-trips."""
    import mlsgpu_amd as m
    import oracle_binding as ob
    from mlsgpu_amd import binding as b, synth
    cloud = synth.shells_cloud(120_000, 95.0, 16.0, 1.5, 2.5, seed=321)
    allb, buckets = synth.bucketize(cloud, 96, 32)
    ctx = m.Context(0)
    dev = m.DeviceBuffer(ctx, array=allb)
    worker = m.Worker(ctx, max(bk.count for bk in buckets), max_cells=63)
    mesher = m.Mesher(ctx, 0.02)
    for bk in buckets:
        worker.process(dev, bk.first, bk.count, bk.low, bk.num_vertices, collector=mesher.collector(ctx, 0))
    assert mesher.finalize() == 1
    got = mesher.chunk(0)
    stats = mesher.stats()
    ref = allb.copy()
    meshes = []
    for bk in buckets:
        batches, _ = ob.bucket(ref, bk.first, bk.count, bk.num_vertices, bk.low, max_cells=63)
        for g in batches:
            meshes.append(dict(chunk=0, vertices=g["vertices"], num_internal=g["num_internal"],
                               keys=g["keys"][g["num_internal"]:], triangles=g["triangles"]))
    exp, exp_stats = mo.mesh_sink(meshes, 0.02)
    for k in ("total_vertices", "threshold", "components", "kept_components", "kept_vertices", "kept_triangles"):
        assert stats[k] == exp_stats[k], k
    assert stats["total_vertices"] < stats["vertices_added"]            # shared vertices were welded
    assert mo.isomorphic(got["vertices"], got["triangles"], exp[0][1], exp[0][2])
    path = tmp_path / "out.ply"
    b.write_ply(path, got["vertices"], got["triangles"], ["mlsgpu version: test"])
    want = mo.ply_bytes(got["vertices"], got["triangles"], ["mlsgpu version: test"])
    assert open(path, "rb").read() == want
    # the same file straight from HBM through a bounded pinned buffer (mlsgpu_hip_mesher_write_ply), whatever its size:
    # pieces of 156 bytes (one vertex-and-face quantum), of a few KB, and one piece for everything
    for buffer_bytes in (1, 4096, 1 << 16, 0):
        streamed = tmp_path / ("streamed_%d.ply" % buffer_bytes)
        mesher.write_ply(0, streamed, ["mlsgpu version: test"], buffer_bytes)
        assert open(streamed, "rb").read() == want, buffer_bytes
    with pytest.raises(b.InvalidArgument):
        mesher.write_ply(1, tmp_path / "none.ply")
    mesher.close()
    del worker
    ctx.close()


@pytest.mark.parametrize("seed,prune,ranks", [(1, 0.0, 2), (2, 0.01, 3), (3, 0.05, 4), (4, 0.3, 2)])
def test_several_device_sinks_one_job(seed, prune, ranks):
    """Several meshers, one job (one process per GPU): every sink welds and labels its own blocks in HBM and exports its
    boundary; dist_sink.merge_boundaries unites components across sinks and applies the prune rule to the whole job; every
    sink finalizes with that verdict.  Blocks are dealt to the sinks round-robin (a seam between every two blocks); each
    sink's output equals the oracle's chunk for it and the whole-job statistics equal the single-sink oracle's."""
    import mlsgpu_amd as m
    from mlsgpu_amd import dist_sink
    meshes = random_meshes(seed, blocks=12, chunks=1)
    owner = [b % ranks for b in range(12)]
    exp, exp_stats = mo.mesh_sink([dict(mm, chunk=owner[i]) for i, mm in enumerate(meshes)], prune)
    ctx = m.Context(0)
    sinks = [m.Mesher(ctx, prune) for _ in range(ranks)]
    for i, mm in enumerate(meshes):
        sinks[owner[i]].add(0, mm["vertices"], mm["num_internal"], mm["keys"], mm["triangles"])
    parts = [s.boundary() for s in sinks]
    keep, stats = dist_sink.merge_boundaries(parts, prune)
    for k in exp_stats:
        assert stats[k] == exp_stats[k], k
    exp_by = {c: (v, t) for c, v, t in exp}
    kept_local = 0
    for r, s in enumerate(sinks):
        n = s.finalize_with(keep[r])
        if r in exp_by:
            assert n == 1
            c = s.chunk(0)
            assert mo.isomorphic(c["vertices"], c["triangles"], *exp_by[r])
        else:
            assert n == 0
        kept_local += s.stats()["kept_triangles"]
        s.close()
    assert kept_local == exp_stats["kept_triangles"]
    ctx.close()


def test_boundary_api_edge_cases():
    import mlsgpu_amd as m
    ctx = m.Context(0)
    sink = m.Mesher(ctx)
    with pytest.raises(m.InvalidArgument):
        sink.finalize_with(np.zeros(0, np.uint8))                  # no boundary() yet
    keys, kr, rv, rt = sink.boundary()                             # an empty sink has an empty boundary
    assert len(keys) == len(rv) == 0
    assert sink.finalize_with(np.zeros(0, np.uint8)) == 0
    sink.reset()                                                   # a finalized sink takes no more meshes until it is reset
    a = CASES["weld"]["meshes"]
    for mesh in a:
        sink.add(0, mesh["vertices"], mesh["num_internal"], mesh["keys"], mesh["triangles"])
    with pytest.raises(m.InvalidArgument):
        sink.finalize_with(np.zeros(0, np.uint8))                  # the boundary is stale after add
    keys, kr, rv, rt = sink.boundary()
    assert len(np.unique(keys)) == len(keys) and np.all(keys[1:] > keys[:-1]) and kr.max() < len(rv)
    assert rt.sum() == sum(len(mesh["triangles"]) for mesh in a)
    with pytest.raises(m.LengthError):
        sink.finalize_with(np.ones(len(rv) + 1, np.uint8))
    assert sink.finalize_with(np.ones(len(rv), np.uint8)) == 1     # keep everything == finalize with threshold 0
    kept = sink.chunk(0)
    assert sink.finalize() == 1
    same = sink.chunk(0)
    np.testing.assert_array_equal(kept["vertices"].view(np.uint32), same["vertices"].view(np.uint32))
    np.testing.assert_array_equal(kept["triangles"], same["triangles"])
    sink.close()
    ctx.close()


@pytest.mark.parametrize("seed,prune", [(5, 0.0), (6, 0.05)])
def test_boundary_after_finalize_reuses_the_analysis(seed, prune):
    """finalize() leaves the weld and the components in the sink's scratch; a boundary() behind it only numbers the roots and
    exports (no key sort, no union-find), and the verdict pass behind THAT starts from the same state.  Same export, same
    meshes as a sink that went add -> boundary -> finalize_with; an add in between invalidates everything."""
    import mlsgpu_amd as m
    meshes = random_meshes(seed, blocks=10, chunks=2)
    ctx = m.Context(0)

    def fill(sink, upto=None):
        seen = {}
        for mesh in meshes[:upto]:
            sink.add(chunk_number(mesh["chunk"], seen), mesh["vertices"], mesh["num_internal"], mesh["keys"], mesh["triangles"])

    fresh = m.Mesher(ctx, prune)
    fill(fresh)
    exp_boundary = fresh.boundary()
    keep = (exp_boundary[2] >= 3).astype(np.uint8)              # some verdict: components of at least three vertices
    n_exp = fresh.finalize_with(keep)
    exp_chunks = [fresh.chunk(i) for i in range(n_exp)]

    sink = m.Mesher(ctx, prune)
    fill(sink)
    n_plain = sink.finalize()
    plain = [sink.chunk(i) for i in range(n_plain)]
    got_boundary = sink.boundary()                               # reuses finalize's weld and components
    for a, b in zip(exp_boundary, got_boundary):
        np.testing.assert_array_equal(a, b)
    assert sink.finalize_with(keep) == n_exp
    for e, g in zip(exp_chunks, [sink.chunk(i) for i in range(n_exp)]):
        np.testing.assert_array_equal(e["vertices"].view(np.uint32), g["vertices"].view(np.uint32))
        np.testing.assert_array_equal(e["triangles"], g["triangles"])
    # ... and a plain finalize again gives what it gave before
    assert sink.finalize() == n_plain
    for e, g in zip(plain, [sink.chunk(i) for i in range(n_plain)]):
        np.testing.assert_array_equal(e["vertices"].view(np.uint32), g["vertices"].view(np.uint32))
        np.testing.assert_array_equal(e["triangles"], g["triangles"])
    # two boundaries in a row, then the verdict
    b1 = sink.boundary()
    b2 = sink.boundary()
    for a, b in zip(b1, b2):
        np.testing.assert_array_equal(a, b)
    assert sink.finalize_with(keep) == n_exp
    fresh.close()
    sink.close()
    ctx.close()
