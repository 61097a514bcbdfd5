"""mlsgpu CPU ORACLE, mesh sink -- TEST INFRASTRUCTURE ONLY.

A numpy restatement of what OOCMesher makes of the meshes it is given (src/mesher.cpp): weld of external vertices by
64-bit key (updateClumpKeyMap :286-311), connected components across blocks (computeLocalComponents :220-236 +
the clump union-find), the prune rule (getStatistics :491-536: a component is kept iff its vertex count -- every welded
vertex once -- is >= uint64(total * threshold)), and per-chunk output in which a key appears once per chunk
(writeChunkPrepare / externalRemap, :538-567, 626-700).  Vertex and triangle ORDER in the output is an implementation
detail of the reference (clump order, reorder buffer); its own tests compare up to isomorphism
(test/test_mesher.cpp:401-460) and so do ours.

Nothing in the product path may import this file.  Pinned by tests/test_oracle_mesher.py with the reference's vectors
(test/test_mesher.cpp:250-1008).  Also here: the PLY layout of FastPly::Writer (src/fast_ply.cpp:443-481).
"""
import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components


def mesh_sink(meshes, prune_threshold=0.0):
    """meshes: dicts with chunk (hashable, arrival order = chunk order), vertices (n,3) float32 with the internal
    vertices first, num_internal, keys (n - num_internal,) uint64, triangles (t,3) uint32.
    Returns (chunks, stats): chunks = list of (chunk, vertices, triangles) for chunks with triangles, arrival order."""
    bases, n = [], 0
    for m in meshes:
        bases.append(n)
        n += len(m["vertices"])
    comp_rep = np.arange(n, dtype=np.int64)          # welded identity for components: first vertex with the key
    out_rep = np.arange(n, dtype=np.int64)           # welded identity inside a chunk's file
    first_by_key, first_by_chunk_key = {}, {}
    for m, base in zip(meshes, bases):
        ni = int(m["num_internal"])
        for j, key in enumerate(np.asarray(m["keys"], np.uint64)):
            g = base + ni + j
            comp_rep[g] = first_by_key.setdefault(int(key), g)
            out_rep[g] = first_by_chunk_key.setdefault((m["chunk"], int(key)), g)
    tris = [np.asarray(m["triangles"], np.int64).reshape(-1, 3) + b for m, b in zip(meshes, bases)]
    all_t = np.concatenate(tris) if tris else np.zeros((0, 3), np.int64)
    ct = comp_rep[all_t]
    rows = np.concatenate([ct[:, 0], ct[:, 1], np.arange(n)])
    cols = np.concatenate([ct[:, 1], ct[:, 2], comp_rep])
    graph = coo_matrix((np.ones(len(rows), np.int8), (rows, cols)), shape=(n, n))
    _, label = connected_components(graph, directed=False)
    is_rep = comp_rep == np.arange(n)
    size = np.bincount(label[is_rep], minlength=label.max() + 1 if n else 0)
    total = int(is_rep.sum())
    threshold = int(np.uint64(total * prune_threshold))
    keep = size >= threshold
    stats = dict(total_vertices=total, threshold=threshold, components=int(len(size)), kept_components=int(keep.sum()),
                 kept_vertices=int(size[keep].sum()), kept_triangles=int(keep[label[all_t[:, 0]]].sum()) if len(all_t) else 0)
    chunks, order = {}, []
    for m, base, t in zip(meshes, bases, tris):
        if m["chunk"] not in chunks:
            chunks[m["chunk"]] = ([], [])
            order.append(m["chunk"])
        nv = len(m["vertices"])
        g = np.arange(base, base + nv)
        chunks[m["chunk"]][0].append(g[(out_rep[g] == g) & keep[label[g]]])
        chunks[m["chunk"]][1].append(t[keep[label[t[:, 0]]]] if len(t) else t)
    all_v = np.concatenate([np.asarray(m["vertices"], np.float32).reshape(-1, 3) for m in meshes]) if meshes \
        else np.zeros((0, 3), np.float32)
    out = []
    for c in order:
        vg = np.concatenate(chunks[c][0])
        tg = np.concatenate(chunks[c][1])
        if len(tg) == 0:
            continue                                  # "Output should not be produced for empty chunks"
        index = np.full(n, -1, np.int64)
        index[vg] = np.arange(len(vg))
        t_out = index[out_rep[tg]]
        assert (t_out >= 0).all()
        out.append((c, all_v[vg], t_out.astype(np.uint32)))
    return out, stats


def canonical(vertices, triangles):
    """Order-independent form of a mesh with unique vertex positions: sorted rotations-normalised position triples."""
    v = np.asarray(vertices, np.float32).view(np.uint32).reshape(-1, 3).astype(np.uint64)
    code = (v[:, 0] << np.uint64(42)) ^ (v[:, 1] << np.uint64(21)) ^ v[:, 2]      # a label per vertex (tests: small coords)
    assert len(np.unique(v, axis=0)) == len(v), "vertices must be unique"
    order = np.argsort(np.lexsort((v[:, 2], v[:, 1], v[:, 0])))               # rank of every vertex by position
    del code
    t = order[np.asarray(triangles, np.int64).reshape(-1, 3)]
    rot = np.stack([t, t[:, [1, 2, 0]], t[:, [2, 0, 1]]], axis=1)             # (n, 3 rotations, 3)
    best = np.lexsort((rot[:, :, 2], rot[:, :, 1], rot[:, :, 0]), axis=1)[:, 0]
    t = rot[np.arange(len(t)), best]
    t = t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]
    vs = np.asarray(vertices, np.float32)[np.argsort(order)]
    return vs, t


def isomorphic(vertices_a, triangles_a, vertices_b, triangles_b):
    """TestMesherBase::checkIsomorphic, test/test_mesher.cpp:401-460"""
    if len(vertices_a) != len(vertices_b) or len(triangles_a) != len(triangles_b):
        return False
    va, ta = canonical(vertices_a, triangles_a)
    vb, tb = canonical(vertices_b, triangles_b)
    return np.array_equal(va.view(np.uint32), vb.view(np.uint32)) and np.array_equal(ta, tb)


def ply_bytes(vertices, triangles, comments=()):
    """FastPly::Writer's file, src/fast_ply.cpp:443-481 (little endian)."""
    head = "ply\nformat binary_little_endian 1.0\n"
    for c in comments:
        head += "comment %s\n" % c
    head += "element vertex %d\nproperty float32 x\nproperty float32 y\nproperty float32 z\n" % len(vertices)
    head += "element face %d\nproperty list uint8 uint32 vertex_indices\ncomment padding:" % len(triangles)
    size = len(head) + 12
    while size % 4:
        head += "X"
        size += 1
    head += "\nend_header\n"
    faces = np.zeros(len(triangles), np.dtype([("n", np.uint8), ("i", "<u4", 3)]))
    faces["n"] = 3
    faces["i"] = np.asarray(triangles, np.uint32).reshape(-1, 3)
    return head.encode("ascii") + np.asarray(vertices, "<f4").tobytes() + faces.tobytes()
