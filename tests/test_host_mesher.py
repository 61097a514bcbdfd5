"""Host mesh sink (mlsgpu_hip_host_mesher_*: OOCMesher's weld on the host, the cross-GPU welder behind the farm's
host output) against the reference's mesher vectors and the oracle, up to isomorphism -- the comparison the reference's
own tests make (test/test_mesher.cpp:401-460).  Host code: runs without a GPU."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import mesher_oracle as mo  # noqa: E402
from mesher_cases import CASES, random_meshes  # noqa: E402


def run_host(meshes, prune=0.0):
    import mlsgpu_amd as m
    mesher = m.HostMesher(prune)
    seen = {}
    for mesh in meshes:
        mesher.add(seen.setdefault(mesh["chunk"], len(seen)), mesh["vertices"], mesh["num_internal"], mesh["keys"],
                   mesh["triangles"])
    n = mesher.finalize()
    out = [mesher.chunk(i) for i in range(n)]
    stats = mesher.stats()
    mesher.close()
    back = {v: k for k, v in seen.items()}
    return [(back[c], v, t) for c, v, t in out], stats


@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_case(name):
    case = CASES[name]
    out, stats = run_host(case["meshes"], case.get("prune", 0.0))
    assert [c for c, _, _ in out] == [c for c, _, _ in case["expected"]]
    for (_, v, t), (_, ev, et) in zip(out, case["expected"]):
        assert mo.isomorphic(v, t, ev, et), name
    for k, val in case.get("stats", {}).items():
        assert stats[k] == val


def test_block_order_does_not_matter():
    # test/test_mesher.cpp:497-525 adds the blocks in reverse on the second pass
    case = CASES["weld"]
    out, _ = run_host(case["meshes"][::-1])
    (_, v, t), (_, ev, et) = out[0], case["expected"][0]
    assert mo.isomorphic(v, t, ev, et)


@pytest.mark.parametrize("seed,prune", [(1, 0.0), (2, 0.01), (3, 0.05), (4, 0.3)])
def test_random_sheets_match_oracle(seed, prune):
    meshes = random_meshes(seed)
    exp, exp_stats = mo.mesh_sink(meshes, prune)
    out, stats = run_host(meshes, prune)
    for k in ("total_vertices", "threshold", "components", "kept_components", "kept_vertices", "kept_triangles"):
        assert stats[k] == exp_stats[k], k
    assert [c for c, _, _ in out] == [c for c, _, _ in exp]
    for (_, v, t), (_, ev, et) in zip(out, exp):
        assert mo.isomorphic(v, t, ev, et)


def test_interleaved_chunks_and_empty_blocks():
    meshes = random_meshes(7, blocks=12, chunks=3)
    mixed = [meshes[i] for i in [0, 4, 8, 1, 5, 9, 2, 10, 6, 3, 7, 11]]
    empty = dict(chunk=1, vertices=np.zeros((0, 3), np.float32), num_internal=0, keys=np.zeros(0, np.uint64),
                 triangles=np.zeros((0, 3), np.uint32))
    mixed.insert(5, empty)
    exp, exp_stats = mo.mesh_sink(mixed, 0.02)
    out, stats = run_host(mixed, 0.02)
    for k in ("total_vertices", "threshold", "components", "kept_components", "kept_vertices", "kept_triangles"):
        assert stats[k] == exp_stats[k], k
    assert [c for c, _, _ in out] == [c for c, _, _ in exp] == [0, 1, 2]
    for (_, v, t), (_, ev, et) in zip(out, exp):
        assert mo.isomorphic(v, t, ev, et)


def test_argument_checks():
    import mlsgpu_amd as m
    mesher = m.HostMesher()
    with pytest.raises(m.InvalidArgument):
        mesher.add(0, np.zeros((2, 3), np.float32), 2, np.zeros(0, np.uint64), np.array([[0, 1, 2]], np.uint32))   # index out of range
    with pytest.raises(m.InvalidArgument):
        mesher.stats()                       # not finalized
    with pytest.raises(m.InvalidArgument):
        m.HostMesher(1.5)


def test_bad_mesh_leaves_the_sink_unchanged_and_empty_sinks_finalize():
    import mlsgpu_amd as m
    mesher = m.HostMesher()
    assert mesher.finalize() == 0 and mesher.stats()["total_vertices"] == 0         # nothing added
    a = CASES["simple"]["meshes"][0]
    mesher.add(0, a["vertices"], a["num_internal"], a["keys"], a["triangles"])
    with pytest.raises(m.InvalidArgument):
        mesher.add(0, a["vertices"], a["num_internal"], a["keys"], np.array([[0, 1, 99]], np.uint32))
    assert mesher.finalize() == 1
    st = mesher.stats()
    assert st["vertices_added"] == len(a["vertices"]) and st["triangles_added"] == len(a["triangles"])
    keys, kc, cv, ct = mesher.boundary()
    assert cv.sum() == st["total_vertices"] and ct.sum() == st["triangles_added"]
    with pytest.raises(m.LengthError):
        mesher.finalize_with(np.ones(len(cv) + 1, np.uint8))
    assert mesher.finalize_with(np.zeros(len(cv), np.uint8)) == 0                   # everything pruned by verdict


@pytest.mark.parametrize("prune", [0.0, 0.05])
def test_thread_count_does_not_change_the_output(prune):
    """add() queues a block's work on the welder's pool and finalize builds the output on it; clump numbers, union
    records and the vertex a chunk keeps for a key are expressed in arrival order, so 1, 3 or 16 threads give the same
    arrays element for element (and the same as the oracle up to isomorphism)."""
    import mlsgpu_amd as m
    meshes = random_meshes(11, blocks=24, chunks=3)
    order = [3, 0, 9, 1, 17, 4, 2, 10, 23, 5, 6, 11, 7, 8, 12, 13, 20, 14, 15, 16, 18, 19, 21, 22]   # chunks interleaved
    mixed = [meshes[i] for i in order]
    outs = []
    for threads in (1, 3, 16):
        mesher = m.HostMesher(prune, threads=threads)
        assert mesher.threads() == threads
        seen = {}
        for mesh in mixed:
            mesher.add(seen.setdefault(mesh["chunk"], len(seen)), mesh["vertices"], mesh["num_internal"], mesh["keys"],
                       mesh["triangles"])
        n = mesher.finalize()
        outs.append(([mesher.chunk(i) for i in range(n)], mesher.stats(), mesher.boundary()))
        mesher.close()
    exp, exp_stats = mo.mesh_sink(mixed, prune)
    first, stats0, b0 = outs[0]
    for k in ("total_vertices", "threshold", "components", "kept_components", "kept_vertices", "kept_triangles"):
        assert stats0[k] == exp_stats[k], k
    for (c, v, t), (_, ev, et) in zip(first, exp):
        assert mo.isomorphic(v, t, ev, et)
    for other, stats, b in outs[1:]:
        assert stats == stats0 and len(other) == len(first)
        for (c0, v0, t0), (c1, v1, t1) in zip(first, other):
            assert c0 == c1
            np.testing.assert_array_equal(v0.view(np.uint32), v1.view(np.uint32))
            np.testing.assert_array_equal(t0, t1)
        for x, y in zip(b0, b):
            np.testing.assert_array_equal(x, y)


def test_slab_cache_can_be_trimmed():
    """The process keeps the welders' mapped slabs between jobs; mlsgpu_hip_host_mesher_trim_cache gives them back
    (a second job after the trim still welds the same mesh: reused or fresh slabs, the result does not depend on it)."""
    import mlsgpu_amd as m
    L = m.lib()
    rng = np.random.default_rng(11)
    v = rng.random((3000, 3)).astype(np.float32)
    t = rng.integers(0, 3000, (5000, 3)).astype(np.uint32)
    k = np.arange(1000, dtype=np.uint64)                      # the keys of the 1000 external vertices

    def job():
        w = m.HostMesher(0.0, threads=2)
        w.add(0, v, 2000, k, t)
        w.finalize()
        out = w.chunk(0)
        w.close()
        return out
    first = job()
    released = L.mlsgpu_hip_host_mesher_trim_cache(0)
    assert released > 0                                       # the job's slabs were being kept
    assert L.mlsgpu_hip_host_mesher_trim_cache(0) == 0        # ... and are gone
    job()
    assert L.mlsgpu_hip_host_mesher_trim_cache(0) == 0        # limit 0: nothing is kept any more
    L.mlsgpu_hip_host_mesher_trim_cache(16 << 30)
    second = job()
    assert L.mlsgpu_hip_host_mesher_trim_cache(16 << 30) == 0
    for a, b in zip(first[1:], second[1:]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("prune", [0.0, 0.05])
def test_landed_meshes_weld_like_copied_ones(prune):
    """The in-place route (mlsgpu_hip_host_mesher_landing + _add_landed: what a farm's read-back lands in is adopted, not
    copied) gives the arrays the copying route gives, element for element; blocks of both kinds mix."""
    import mlsgpu_amd as m
    meshes = random_meshes(21, blocks=18, chunks=3)
    outs = []
    for mode in ("copied", "landed", "mixed"):
        mesher = m.HostMesher(prune, threads=4)
        seen = {}
        for i, mesh in enumerate(meshes):
            landed = mode == "landed" or (mode == "mixed" and i % 2 == 1)
            (mesher.add_landed if landed else mesher.add)(seen.setdefault(mesh["chunk"], len(seen)), mesh["vertices"],
                                                          mesh["num_internal"], mesh["keys"], mesh["triangles"])
        n = mesher.finalize()
        outs.append(([mesher.chunk(i) for i in range(n)], mesher.stats()))
        mesher.close()
    first, stats0 = outs[0]
    assert len(first) >= 1
    for other, stats in outs[1:]:
        assert stats == stats0 and len(other) == len(first)
        for (c0, v0, t0), (c1, v1, t1) in zip(first, other):
            assert c0 == c1
            np.testing.assert_array_equal(v0.view(np.uint32), v1.view(np.uint32))
            np.testing.assert_array_equal(t0, t1)


def test_landed_argument_checks():
    """add_landed only adopts memory the welder handed out; a landed mesh with a bad triangle index fails the job at
    finalize (the check that rides on add()'s copy runs in the block's task)."""
    import ctypes as C
    import mlsgpu_amd as m
    from mlsgpu_amd.binding import HostMesh, lib
    a = CASES["simple"]["meshes"][0]
    mesher = m.HostMesher()
    v = np.ascontiguousarray(a["vertices"], np.float32)
    t = np.ascontiguousarray(a["triangles"], np.uint32)
    k = np.ascontiguousarray(a["keys"], np.uint64)
    hm = HostMesh(k.ctypes.data, v.ctypes.data, t.ctypes.data, len(v), len(t), a["num_internal"])      # the caller's own memory
    assert lib().mlsgpu_hip_host_mesher_add_landed(mesher.h, 0, C.byref(hm)) != 0
    assert mesher.finalize() == 0
    mesher.close()
    mesher = m.HostMesher()
    mesher.add_landed(0, a["vertices"], a["num_internal"], a["keys"], a["triangles"])
    mesher.add_landed(0, a["vertices"], a["num_internal"], a["keys"], np.array([[0, 1, 99]], np.uint32))
    with pytest.raises(m.InvalidArgument):
        mesher.finalize()
    mesher.close()


@pytest.mark.parametrize("prune", [0.0, 0.05])
def test_temporary_files_give_the_same_mesh(prune, tmp_path):
    """Bounded-memory mode (mlsgpu_hip_host_mesher_set_tmp_dir; OOCMesher's temporary files, src/mesher.cpp:404-419, 763-852):
    the welder's blocks, scratch and outputs live in mappings of nameless temporary files, every block is handed to the
    kernel to write out once it is welded (resident budget 0) -- and the meshes are the in-memory mode's, element for element,
    whichever route the blocks arrive by."""
    import mlsgpu_amd as m
    meshes = random_meshes(31, blocks=20, chunks=3)
    outs = []
    for mode in ("memory", "files", "files-landed"):
        mesher = m.HostMesher(prune, threads=4)
        if mode != "memory":
            mesher.set_tmp_dir(tmp_path, 0)
        seen = {}
        for i, mesh in enumerate(meshes):
            landed = mode == "files-landed" and i % 2 == 0
            (mesher.add_landed if landed else mesher.add)(seen.setdefault(mesh["chunk"], len(seen)), mesh["vertices"],
                                                          mesh["num_internal"], mesh["keys"], mesh["triangles"])
        n = mesher.finalize()
        outs.append(([mesher.chunk(i) for i in range(n)], mesher.stats()))
        usage = mesher.tmp_usage()
        if mode == "memory":
            assert usage["mapped"] == 0
        else:
            assert usage["mapped"] >= 256 << 20 and usage["resident"] <= usage["mapped"]
            assert not mesher.landing_pinned() or mode == "files"      # file-backed landing memory is not page-locked
            assert list(tmp_path.iterdir()) == []                   # the files have no names
        mesher.close()
    first, stats0 = outs[0]
    assert len(first) >= 1
    for other, stats in outs[1:]:
        assert stats == stats0 and len(other) == len(first)
        for (c0, v0, t0), (c1, v1, t1) in zip(first, other):
            assert c0 == c1
            np.testing.assert_array_equal(v0.view(np.uint32), v1.view(np.uint32))
            np.testing.assert_array_equal(t0, t1)


def test_temporary_files_argument_checks(tmp_path):
    import mlsgpu_amd as m
    mesher = m.HostMesher()
    with pytest.raises(m.InvalidArgument):
        mesher.set_tmp_dir(tmp_path / "not-there", 0)
    a = CASES["simple"]["meshes"][0]
    mesher.set_tmp_dir(tmp_path, 1 << 30)
    mesher.set_tmp_dir(None)                                        # back to memory
    mesher.add(0, a["vertices"], a["num_internal"], a["keys"], a["triangles"])
    with pytest.raises(m.InvalidArgument):
        mesher.set_tmp_dir(tmp_path, 0)                             # not once blocks have arrived
    assert mesher.finalize() == 1 and mesher.tmp_usage()["mapped"] == 0
    mesher.close()


def test_temporary_files_leave_memory(tmp_path):
    """What bounded means: 96 MB of blocks through a welder with a resident budget of 8 MB -- once they are welded all but the
    budget has been handed to the kernel to write out and drop.  Whether the pages really leave is the kernel's business
    (MADV_PAGEOUT needs Linux 5.4 and a file system on a disk): reported, and checked only where the kernel did it."""
    import mlsgpu_amd as m
    rng = np.random.default_rng(5)
    mesher = m.HostMesher(0.0, threads=4)
    mesher.set_tmp_dir(tmp_path, 8 << 20)
    nv, nt = 400_000, 800_000
    total = 0
    for b in range(6):
        v = rng.random((nv, 3)).astype(np.float32)
        t = rng.integers(0, nv, (nt, 3)).astype(np.uint32)
        k = (np.arange(1000, dtype=np.uint64) + np.uint64(b * 1000))
        mesher.add(b % 2, v, nv - 1000, k, t)
        total += v.nbytes + t.nbytes + k.nbytes
    n = mesher.finalize()
    assert n == 2
    usage = mesher.tmp_usage()
    assert usage["mapped"] >= total
    if usage["paged_out"]:
        assert usage["paged_out"] >= total - (8 << 20) - 6 * 3 * 8192   # whole pages inside the arrays beyond the budget
    st = mesher.stats()
    assert st["vertices_added"] == 6 * nv and st["triangles_added"] == 6 * nt
    mesher.close()
