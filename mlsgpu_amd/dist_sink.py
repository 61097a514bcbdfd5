"""Cross-rank finish of the mesh sink for one-process-per-GPU runs.

Every rank welds the ship-outs of ITS buckets with a welder of its own -- a HostMesher (OOCMesher's weld on the host,
src/mesher.cpp:220-469) or a device Mesher (the same weld in HBM; the meshes never leave the GPU) -- both export the same
boundary (keys with the component holding the vertex, component sizes) and accept the same verdict.  What is left is
global: a connected component may cross rank boundaries, and the prune threshold is a fraction of the TOTAL
number of welded vertices (getStatistics, src/mesher.cpp:491-536).  The reference's MPI build sends every ship-out to
one rank's mesher instead (src/mlsgpu_mpi.cpp); here only the boundary travels -- each rank's external keys with the
clump that holds the vertex, and the clump sizes -- in ONE all-gather (the path's only exchange step; a few tens of MB
for cfg3-sized ranks), after which every rank computes the same verdict and finalizes its own part.  Each rank's output
is its own chunk(s): a vertex shared with another rank appears in both, exactly as a vertex shared by two chunks does in
the reference (externalRemap is per chunk, src/mesher.cpp:538-567).
"""
import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components


def merge_boundaries(parts, prune_threshold):
    """parts[r] = (keys, key_clump, clump_vertices, clump_triangles) of rank r's mesher (HostMesher.boundary()).
    Returns (keep, stats): keep[r] = uint8 verdict per clump of rank r; stats as the mesher's (whole job)."""
    offsets = np.concatenate([[0], np.cumsum([len(p[2]) for p in parts])]).astype(np.int64)
    n = int(offsets[-1])
    keys = np.concatenate([np.asarray(p[0], np.uint64) for p in parts]) if parts else np.zeros(0, np.uint64)
    nodes = np.concatenate([np.asarray(p[1], np.int64) + offsets[r] for r, p in enumerate(parts)]) if parts \
        else np.zeros(0, np.int64)
    verts = np.concatenate([np.asarray(p[2], np.int64) for p in parts]) if parts else np.zeros(0, np.int64)
    tris = np.concatenate([np.asarray(p[3], np.int64) for p in parts]) if parts else np.zeros(0, np.int64)
    is_root = verts > 0
    order = np.argsort(keys, kind="stable")
    keys, nodes = keys[order], nodes[order]
    same = keys[1:] == keys[:-1]                       # a key two ranks have seen: one vertex, two clumps to unite
    rows, cols = nodes[:-1][same], nodes[1:][same]
    graph = coo_matrix((np.ones(len(rows), np.int8), (rows, cols)), shape=(n, n))
    _, label = connected_components(graph, directed=False)
    ncomp = int(label.max()) + 1 if n else 0
    size = np.bincount(label, weights=verts, minlength=ncomp).astype(np.int64)
    size -= np.bincount(label[cols], minlength=ncomp)  # counted once per rank that has it: r - 1 too many
    tcount = np.bincount(label, weights=tris, minlength=ncomp).astype(np.int64)
    live = np.bincount(label[is_root], minlength=ncomp) > 0
    total = int(size[live].sum())
    threshold = int(np.uint64(total * prune_threshold))
    keep_comp = live & (size >= threshold)
    keep_node = keep_comp[label] & is_root
    keep = [keep_node[offsets[r]:offsets[r + 1]].astype(np.uint8) for r in range(len(parts))]
    stats = dict(total_vertices=total, threshold=threshold, components=int(live.sum()), kept_components=int(keep_comp.sum()),
                 kept_vertices=int(size[keep_comp].sum()), kept_triangles=int(tcount[keep_comp].sum()))
    return keep, stats


def global_prune(mesher, prune_threshold, dist=None):
    """Finalizes `mesher` (this rank's HostMesher or device Mesher) with the whole job's components and threshold.  `dist`:
    torch.distributed, initialised, or None for a single process.  Returns (number of output chunks, whole-job stats)."""
    mine = mesher.boundary()
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        parts, rank = [mine], 0
    else:
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, mine)
        rank = dist.get_rank()
    keep, stats = merge_boundaries(parts, prune_threshold)
    return mesher.finalize_with(keep[rank]), stats
