"""The fixtures of test/test_mesher.cpp (:250-344 blocks 0-3, :462-1008 expected outputs), shared by the oracle and
the HIP mesher tests."""
import numpy as np

E = 0x8000000000000000


def mesh(chunk, internal, external, keys, indices):
    internal = np.asarray(internal, np.float32).reshape(-1, 3)
    external = np.asarray(external, np.float32).reshape(-1, 3)
    return dict(chunk=chunk, vertices=np.concatenate([internal, external]), num_internal=len(internal),
                keys=np.asarray(keys, np.uint64), triangles=np.asarray(indices, np.uint32).reshape(-1, 3))


I0 = [(0, 0, 1), (0, 0, 2), (0, 0, 3), (0, 0, 4), (0, 0, 5)]
T0 = [0, 1, 3, 1, 2, 3, 3, 4, 0]
X1 = [(1, 0, 1), (1, 0, 2), (1, 0, 3), (1, 0, 4)]
K1 = [0, E, 1, E + 1]
T1 = [0, 1, 3, 1, 2, 3, 2, 0, 3]
I2 = [(0, 1, 0), (0, 2, 0), (0, 3, 0)]
X2 = [(2, 0, 1), (2, 0, 2)]
K2 = [0x1234567812345678, 0x12345678]
T2 = [0, 1, 3, 1, 4, 3, 2, 3, 4, 0, 2, 4, 0, 3, 2]
I3 = [(3, 3, 3)]
X3 = [(4, 5, 6), (1, 0, 2), (1, 0, 3), (2, 0, 2)]
K3 = [100, E, 1, 0x12345678]
T3 = [0, 2, 1, 1, 2, 4, 4, 2, 3]


def arr(v):
    return np.asarray(v, np.float32).reshape(-1, 3)


def tri(t):
    return np.asarray(t, np.uint32).reshape(-1, 3)


CASES = {}

CASES["simple"] = dict(                                                   # testSimple, :462-544
    meshes=[mesh(0, I0, [], [], T0), mesh(0, [], X1, K1, T1), mesh(0, I2, X2, K2, T2)],
    expected=[(0, arr(I0 + X1 + I2 + X2),
               tri([0, 1, 3, 1, 2, 3, 3, 4, 0, 5, 6, 8, 6, 7, 8, 7, 5, 8, 9, 10, 12, 10, 13, 12, 11, 12, 13, 9, 11, 13, 9, 12, 11]))])

CASES["no_internal"] = dict(                                              # testNoInternal, :546-596
    meshes=[mesh(0, [], X1, K1, T1), mesh(0, [], X2, K2, [0, 1, 1, 0, 0, 1])],
    expected=[(0, arr(X1 + X2), tri([0, 1, 3, 1, 2, 3, 2, 0, 3, 4, 5, 5, 4, 4, 5]))])

CASES["no_external"] = dict(                                              # testNoExternal, :598-650
    meshes=[mesh(0, I0, [], [], T0), mesh(0, I2, [], [], [0, 1, 2, 2, 1, 0])],
    expected=[(0, arr(I0 + I2), tri([0, 1, 3, 1, 2, 3, 3, 4, 0, 5, 6, 7, 7, 6, 5]))])

CASES["empty"] = dict(meshes=[mesh(0, [], [], [], [])], expected=[])      # testEmpty, :652-669

CASES["weld"] = dict(                                                     # testWeld, :671-743
    meshes=[mesh(0, I0, [], [], T0), mesh(0, [], X1, K1, T1), mesh(0, I2, X2, K2, T2), mesh(0, I3, X3, K3, T3)],
    expected=[(0, arr(I0 + X1 + I2 + X2 + [(3, 3, 3), (4, 5, 6)]),
               tri([0, 1, 3, 1, 2, 3, 3, 4, 0, 5, 6, 8, 6, 7, 8, 7, 5, 8, 9, 10, 12, 10, 13, 12, 11, 12, 13, 9, 11, 13, 9, 12, 11,
                    14, 6, 15, 15, 6, 13, 13, 6, 7]))])

# testPrune, :745-922: components A (5 vertices, block 0), B (6, block 1), C (5, blocks 1 and 3), D (6, blocks 0-3);
# 22 vertices, threshold 6.5 / 22 -> 6: A and C go
P_I0 = [(0, 0, 0), (1, 0, 0), (2, 0, 0), (3, 0, 0), (4, 0, 0)]
P_X0 = [(0, 3, 0), (1, 3, 0), (2, 3, 0)]
P_I1 = [(0, 1, 0), (1, 1, 0), (2, 1, 0), (3, 1, 0), (4, 1, 0), (5, 1, 0), (0, 2, 0), (3, 2, 0)]
P_X1 = [(2, 2, 0), (4, 2, 0), (0, 3, 0), (2, 3, 0), (4, 3, 0)]
P_X2 = [(1, 3, 0), (2, 3, 0), (3, 3, 0)]
P_I3 = [(1, 2, 0), (5, 3, 0)]
P_X3 = [(2, 2, 0), (3, 3, 0), (4, 2, 0), (4, 3, 0), (2, 3, 0)]
CASES["prune"] = dict(
    meshes=[mesh(0, P_I0, P_X0, [0x30, 0x31, 0x32], [0, 4, 1, 1, 4, 2, 2, 4, 3, 5, 7, 6]),
            mesh(0, P_I1, P_X1, [0x22, 0x24, 0x30, 0x32, 0x34], [0, 5, 1, 1, 5, 2, 2, 5, 3, 3, 5, 4, 6, 7, 9, 9, 7, 8, 10, 12, 11]),
            mesh(0, [], P_X2, [0x31, 0x32, 0x33], [0, 1, 2]),
            mesh(0, P_I3, P_X3, [0x22, 0x33, 0x24, 0x34, 0x32], [6, 5, 3, 4, 2, 0, 3, 5, 1])],
    prune=6.5 / 22.0,
    stats=dict(total_vertices=22, threshold=6, components=4, kept_components=2, kept_vertices=12),
    expected=[(0, arr([(0, 1, 0), (1, 1, 0), (2, 1, 0), (3, 1, 0), (4, 1, 0), (5, 1, 0),
                       (0, 3, 0), (1, 3, 0), (2, 3, 0), (3, 3, 0), (4, 3, 0), (5, 3, 0)]),
               tri([0, 5, 1, 1, 5, 2, 2, 5, 3, 3, 5, 4, 6, 8, 7, 7, 8, 9, 9, 8, 10, 9, 10, 11, 6, 10, 8]))])

CASES["chunk"] = dict(                                                    # testChunk, :924-1008: one file per chunk,
    meshes=[mesh((0, 0, 1), I0, [], [], T0), mesh((1, 1, 1), [], X1, K1, T1),   # shared vertices repeated per file
            mesh((2, 4, 1), I2, X2, K2, T2), mesh((3, 9, 1), I3, X3, K3, T3)],
    expected=[((0, 0, 1), arr(I0), tri(T0)), ((1, 1, 1), arr(X1), tri(T1)),
              ((2, 4, 1), arr(I2 + X2), tri(T2)), ((3, 9, 1), arr(I3 + X3), tri(T3))])


def random_meshes(seed, blocks=12, chunks=3):
    """Blocks of a triangulated grid sheet cut into strips: vertices on the cuts are external with a shared key;
    a few small islands test pruning.  Positions are unique per welded vertex."""
    rng = np.random.default_rng(seed)
    width, height = 40, 6 * blocks
    meshes = []
    for b in range(blocks):
        y0, y1 = 6 * b, 6 * (b + 1)                       # rows y0..y1 inclusive; rows y0 and y1 are shared
        gaps = rng.random((y1 - y0, width - 1)) < 0.15    # missing quads break the sheet into components
        ids = -np.ones((y1 - y0 + 1, width), np.int64)
        tris = []
        for y in range(y0, y1):
            for x in range(width - 1):
                if gaps[y - y0, x]:
                    continue
                quad = [(y, x), (y, x + 1), (y + 1, x + 1), (y + 1, x)]
                tris.append([quad[0], quad[1], quad[2]])
                tris.append([quad[0], quad[2], quad[3]])
        used = sorted({p for t in tris for p in t})
        internal = [p for p in used if p[0] not in (y0, y1)]
        external = [p for p in used if p[0] in (y0, y1)]
        order = {p: i for i, p in enumerate(internal + external)}
        verts = np.array([[p[1], p[0], (p[0] * 7 + p[1] * 3) % 5] for p in internal + external], np.float32).reshape(-1, 3)
        keys = np.array([(p[0] << 21) | p[1] | (1 << 63) for p in external], np.uint64)
        t = np.array([[order[p] for p in tri] for tri in tris], np.uint32).reshape(-1, 3)
        meshes.append(dict(chunk=b * chunks // blocks, vertices=verts, num_internal=len(internal), keys=keys, triangles=t))
    return meshes
