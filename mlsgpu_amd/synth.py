"""Synthetic splat clouds and a simple host bucketer for the BASELINE.json configurations.

Host-side input preparation only (numpy): the reference's equivalents are the PLY reader, the
bucketer and BucketLoader (src/bucket.h, src/bucket_loader.cpp), all outside the device path.
Generators are counter-based splitmix64 with seed 0x6D6C736770750000 + cfg and stream index =
splat id, so any slice of a cloud can be regenerated anywhere.
"""
import numpy as np

SPLAT_DTYPE = np.dtype([("position", np.float32, 3), ("radius", np.float32),
                        ("normal", np.float32, 3), ("quality", np.float32)])
SEED_BASE = 0x6D6C736770750000
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniforms(seed, first, count, k):
    """float64 uniforms in [0,1): value j of splat i is mix(seed + (8*i + j + 1) * golden) >> 11 / 2^53."""
    with np.errstate(over="ignore"):
        ids = np.arange(first, first + count, dtype=np.uint64)
        out = np.empty((k, count), np.float64)
        for j in range(k):
            ctr = ids * np.uint64(8) + np.uint64(j + 1)
            z = _mix(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + ctr * _GOLDEN)
            out[j] = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return out


def sphere_cloud(n, center, big_radius, r_lo, r_hi, seed, first=0):
    """Area-uniform splats on a sphere (z ~ U(-1,1), theta ~ U(-pi,pi), as test/test_mls.cpp:349-380),
    radial normals, radius ~ U(r_lo, r_hi), quality ~ U(0,1]."""
    u = uniforms(seed, first, n, 4)
    z = 2.0 * u[0] - 1.0
    t = (2.0 * u[1] - 1.0) * np.pi
    xy = np.sqrt(1.0 - z * z)
    nrm = np.stack([np.cos(t) * xy, np.sin(t) * xy, z], axis=1)
    s = np.zeros(n, SPLAT_DTYPE)
    s["normal"] = nrm.astype(np.float32)
    s["position"] = (np.asarray(center, np.float64)[None, :] + nrm * big_radius).astype(np.float32)
    s["radius"] = (r_lo + (r_hi - r_lo) * u[2]).astype(np.float32)
    s["quality"] = (1.0 - u[3]).astype(np.float32)
    return s


def uniform_cloud(n, extent, r_lo, r_hi, seed, first=0, chunk=4_000_000):
    """D2 of SURVEY 8d: position ~ U[0, extent)^3, radius ~ U(r_lo, r_hi), normal = normalize(p - centre),
    quality = 1 / r^2."""
    s = np.zeros(n, SPLAT_DTYPE)
    c = 0.5 * extent
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        u = uniforms(seed, first + lo, m, 4)
        p = (u[:3].T * extent)
        d = p - c
        ln = np.sqrt((d * d).sum(axis=1))
        ln[ln == 0] = 1.0
        r = r_lo + (r_hi - r_lo) * u[3]
        v = s[lo:lo + m]
        v["position"] = p.astype(np.float32)
        v["normal"] = (d / ln[:, None]).astype(np.float32)
        v["radius"] = r.astype(np.float32)
        v["quality"] = (1.0 / (r * r)).astype(np.float32)
    return s


def shells_cloud(n, extent, spacing, r_lo, r_hi, seed, first=0):
    """D1 of SURVEY 8d: area-uniform on concentric spheres R = spacing * k around the centre."""
    kmax = max(int((0.5 * extent - 4) // spacing), 1)
    u = uniforms(seed, first, n, 5)
    # pick shell k with probability proportional to its area k^2
    w = np.arange(1, kmax + 1, dtype=np.float64) ** 2
    cdf = np.cumsum(w) / w.sum()
    k = np.searchsorted(cdf, u[4], side="right") + 1
    z = 2.0 * u[0] - 1.0
    t = (2.0 * u[1] - 1.0) * np.pi
    xy = np.sqrt(1.0 - z * z)
    nrm = np.stack([np.cos(t) * xy, np.sin(t) * xy, z], axis=1)
    s = np.zeros(n, SPLAT_DTYPE)
    s["normal"] = nrm.astype(np.float32)
    s["position"] = (0.5 * extent + nrm * (spacing * k)[:, None]).astype(np.float32)
    s["radius"] = (r_lo + (r_hi - r_lo) * u[2]).astype(np.float32)
    s["quality"] = (1.0 - u[3]).astype(np.float32)
    return s


CONFIGS = {
    # name: grid corners per side, splats   (BASELINE.json configs[0..3])
    "cfg1": dict(grid=64, splats=50_000),
    "cfg2": dict(grid=256, splats=5_000_000),
    "cfg3": dict(grid=512, splats=50_000_000),
    "cfg4": dict(grid=1024, splats=200_000_000),
}


def make_cloud(cfg, dist="uniform", scale=1.0, seed_offset=0):
    """Cloud of a BASELINE config.  `scale` < 1 shrinks the splat count (tests); the grid stays."""
    c = CONFIGS[cfg]
    g = c["grid"]
    n = max(int(c["splats"] * scale), 1)
    seed = SEED_BASE + int(cfg[3:]) + (seed_offset << 8)
    if cfg == "cfg1":
        return sphere_cloud(n, (32.0, 32.0, 32.0), 24.0, 1.0, 2.0, seed), g
    if dist == "uniform":
        return uniform_cloud(n, float(g - 1), 2.0, 3.0, seed), g
    return shells_cloud(n, float(g - 1), 16.0, 1.0, 2.0, seed), g


def split_axis(cells, max_cells):
    """Even split of `cells` grid cells into the fewest runs of at most max_cells; returns [(lo, hi)] in
    vertex coordinates (bucket covers vertices lo..hi inclusive, i.e. cells lo..hi-1)."""
    parts = -(-cells // max_cells)
    base, extra = divmod(cells, parts)
    out = []
    lo = 0
    for i in range(parts):
        n = base + (1 if i < extra else 0)
        out.append((lo, lo + n))
        lo += n
    return out


class Bucket:
    __slots__ = ("low", "num_vertices", "first", "count")

    def __init__(self, low, num_vertices, first, count):
        self.low = low
        self.num_vertices = num_vertices
        self.first = first
        self.count = count

    @property
    def cells(self):
        return int(np.prod([n - 1 for n in self.num_vertices]))


def bucketize(splats, grid, max_cells=255):
    """Fixed spatial split with splat halo.  Every bucket receives, in global order, all splats whose
    bounding box [p - r, p + r] meets the bucket's vertex range (src/bucket.h:96-98: "all splats that
    intersect the bucket will be passed"); splats near faces are duplicated.  Keeping global order inside
    a bucket makes the per-corner summation order, hence f at shared corners, independent of the bucket.
    Returns (concatenated splat array, [Bucket])."""
    runs = split_axis(grid - 1, max_cells)
    pos = splats["position"]
    rad = splats["radius"]
    lo = pos - rad[:, None]
    hi = pos + rad[:, None]
    masks = []
    for a in range(3):
        masks.append([(hi[:, a] >= r0) & (lo[:, a] <= r1) for (r0, r1) in runs])
    pieces = []
    buckets = []
    first = 0
    for (z0, z1), mz in zip(runs, masks[2]):
        for (y0, y1), my in zip(runs, masks[1]):
            myz = my & mz
            for (x0, x1), mx in zip(runs, masks[0]):
                idx = np.nonzero(mx & myz)[0]
                pieces.append(splats[idx])
                buckets.append(Bucket((x0, y0, z0), (x1 - x0 + 1, y1 - y0 + 1, z1 - z0 + 1), first, len(idx)))
                first += len(idx)
    return np.concatenate(pieces), buckets
