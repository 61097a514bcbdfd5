"""Test-side helpers that restate the *fixtures* of the reference's own tests.

Nothing here is product code.  Citations are file:line under the reference tree.
"""
import math

import numpy as np

from oracle_binding import SPLAT_DTYPE


class MT19937:
    """std::tr1::mt19937 with the default seed 5489 (the reference's tests default-construct it)."""

    def __init__(self, seed=5489):
        self.mt = [0] * 624
        self.mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            self.mt[i] = (1812433253 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.idx = 624

    def _twist(self):
        mt = self.mt
        for i in range(624):
            y = (mt[i] & 0x80000000) | (mt[(i + 1) % 624] & 0x7FFFFFFF)
            mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
        self.idx = 0

    def next(self):
        if self.idx >= 624:
            self._twist()
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def uniform_real(self, lo, hi, single=False):
        """tr1::variate_generator<mt19937&, uniform_real<T>>: u / 2^32 scaled to [lo, hi)."""
        u = self.next()
        if single:
            f = np.float32(u) / np.float32(4294967296.0)
            return np.float32(f * np.float32(np.float32(hi) - np.float32(lo)) + np.float32(lo))
        return (u / 4294967296.0) * (hi - lo) + lo


def sphere_splats(n, center, radius, engine=None):
    """TestMls::sphereSplats, test/test_mls.cpp:349-380."""
    eng = engine or MT19937()
    s = np.zeros(n, SPLAT_DTYPE)
    for i in range(n):
        z = eng.uniform_real(-1.0, 1.0)
        t = eng.uniform_real(-math.pi, math.pi)
        xy = math.sqrt(1.0 - z * z)
        x = math.cos(t) * xy
        y = math.sin(t) * xy
        s["normal"][i] = (x, y, z)
        s["radius"][i] = eng.uniform_real(radius, 2.0 * radius)
        s["position"][i] = (np.float32(center[0]) + x * radius, np.float32(center[1]) + y * radius,
                            np.float32(center[2]) + z * radius)
        s["quality"][i] = eng.uniform_real(0.0, 1.0)
    return s


def make_splats(rows):
    """addSplat of test/test_splat_tree.cpp:56-69: rows of (x, y, z, r)."""
    s = np.zeros(len(rows), SPLAT_DTYPE)
    for i, (x, y, z, r) in enumerate(rows):
        s["position"][i] = (x, y, z)
        s["radius"][i] = r
        s["normal"][i] = (1.0, 0.0, 0.0)
        s["quality"][i] = 1.0
    return s


def make_code(x, y, z):
    """SplatTree::makeCode (src/splat_tree.cpp:67-82): Morton, z major."""
    code = 0
    for b in range(21):
        code |= ((x >> b) & 1) << (3 * b)
        code |= ((y >> b) & 1) << (3 * b + 1)
        code |= ((z >> b) & 1) << (3 * b + 2)
    return code


def walk(commands, start_pos, limit=100000):
    """Octree walk of src/splat_tree.h:40-74; returns the ids visited, in order."""
    ids = []
    pos = int(start_pos)
    steps = 0
    while pos >= 0:
        assert 0 <= pos < len(commands)
        end = int(commands[pos])
        pos += 1
        assert pos < end < len(commands), "bad end pointer"
        ids.extend(int(c) for c in commands[pos:end])
        pos = int(commands[end])
        assert pos >= -1
        steps += 1
        assert steps < limit, "infinite loop in command list"
    return ids


def is_manifold(num_vertices, triangles):
    """Manifold::isManifold, test/manifold.h:98-232. Returns '' or a reason."""
    edges = [[] for _ in range(num_vertices)]
    for t, tri in enumerate(triangles):
        idx = [int(tri[0]), int(tri[1]), int(tri[2])]
        for _ in range(3):
            if idx[0] >= num_vertices:
                return "Triangle %d contains out-of-range index %d" % (t, idx[0])
            if idx[0] == idx[1]:
                return "Triangle %d contains vertex %d twice" % (t, idx[0])
            edges[idx[0]].append((idx[1], idx[2]))
            idx = idx[1:] + idx[:1]
    for i in range(num_vertices):
        neigh = edges[i]
        if not neigh:
            return "Vertex %d is isolated" % i
        arrow = {}
        seen = set()
        for x, y in neigh:
            if x in arrow:
                return "Edge %d - %d occurs twice with same winding" % (i, x)
            arrow[x] = y
            if y in seen:
                return "Edge %d - %d occurs twice with same winding" % (y, i)
            seen.add(y)
        length = 0
        for x, _ in neigh:
            if x not in seen:
                cur = x
                while cur in arrow:
                    cur = arrow[cur]
                    length += 1
        if length != 0 and length != len(neigh):
            return "Vertex %d is both in the interior and on the boundary" % i
        if length == 0:
            start = neigh[0][0]
            cur = start
            while True:
                cur = arrow[cur]
                length += 1
                if cur == start:
                    break
            if length != len(neigh):
                return "Vertex %d tunnels between interior regions" % i
    return ""


def weld_batches(batches):
    """Host-side weld of the batches a Marching::generate call emits.

    Minimal stand-in for what OOCMesher does with external vertex keys
    (src/mesher.cpp:280-306): external vertices with equal keys are one vertex.
    Returns (vertices [n,3] float32, triangles [m,3] int64, key->index map).
    """
    verts = []
    tris = []
    key_map = {}
    for b in batches:
        nv = len(b["vertices"])
        ni = b["num_internal"]
        remap = np.zeros(nv, np.int64)
        for i in range(ni):
            remap[i] = len(verts)
            verts.append(b["vertices"][i])
        for i in range(ni, nv):
            k = int(b["keys"][i])
            if k not in key_map:
                key_map[k] = len(verts)
                verts.append(b["vertices"][i])
            else:
                # both sides must have produced bit-identical positions
                assert np.array_equal(verts[key_map[k]], b["vertices"][i])
            remap[i] = key_map[k]
        if len(b["triangles"]):
            tris.append(remap[b["triangles"].astype(np.int64)])
    v = np.array(verts, np.float32).reshape(-1, 3)
    t = np.concatenate(tris) if tris else np.zeros((0, 3), np.int64)
    return v, t, key_map


def canonical_mesh(batches):
    """Order-independent form of a welded mesh: sorted vertex rows + sorted triangles
    expressed in sorted-vertex indices (rotated so the smallest index is first)."""
    v, t, _ = weld_batches(batches)
    if len(v) == 0:
        return v, t
    order = np.lexsort((v[:, 2], v[:, 1], v[:, 0]))
    inv = np.empty(len(v), np.int64)
    inv[order] = np.arange(len(v))
    vs = v[order]
    ts = inv[t]
    if len(ts):
        m = np.argmin(ts, axis=1)
        ts = np.stack([np.roll(r, -k) for r, k in zip(ts, m)]) if len(ts) < 200000 else _roll_rows(ts, m)
        ts = ts[np.lexsort((ts[:, 2], ts[:, 1], ts[:, 0]))]
    return vs, ts


def _roll_rows(ts, m):
    out = np.empty_like(ts)
    for k in range(3):
        sel = m == k
        out[sel] = np.roll(ts[sel], -k, axis=1)
    return out
