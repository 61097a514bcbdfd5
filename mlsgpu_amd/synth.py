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
    quality = 1 / r^2.  (uniform_cloud_device generates the same bits on a GPU.)"""
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
    "cfg5": dict(grid=2048, splats=1_000_000_000),   # configs[4]: read from PLY files (write_cloud_ply)
}


def cloud_seed(cfg, seed_offset=0):
    return SEED_BASE + int(cfg[3:]) + (seed_offset << 8)


def _i64(v):
    """A 64-bit pattern as the signed value torch's int64 holds."""
    v &= 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


def uniform_cloud_device(n, extent, r_lo, r_hi, seed, device, first=0, chunk=16_000_000):
    """uniform_cloud on a torch device (bench.py and the full-size tests generate 50 M - 200 M splats in HBM instead
    of on the host): the same counter-based splitmix64 stream and the same float64 -> float32 arithmetic, operation
    for operation, so the result is bit-identical to the numpy generator (tests/test_synth.py checks it).  Returns an
    (n, 8) float32 tensor laid out as SPLAT_DTYPE.  torch is plumbing here -- it only fills input buffers."""
    import torch
    out = torch.empty((n, 8), dtype=torch.float32, device=device)
    golden, m1, m2 = _i64(0x9E3779B97F4A7C15), _i64(0xBF58476D1CE4E5B9), _i64(0x94D049BB133111EB)

    def lsr(z, k):                                   # logical shift right of the 64-bit pattern
        return (z >> k) & ((1 << (64 - k)) - 1)

    def mix(z):
        z = (z ^ lsr(z, 30)) * m1
        z = (z ^ lsr(z, 27)) * m2
        return z ^ lsr(z, 31)
    c = 0.5 * extent
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        ids = torch.arange(first + lo, first + lo + m, dtype=torch.int64, device=device)
        u = []
        for j in range(4):
            ctr = ids * 8 + (j + 1)
            z = mix(ctr * golden + _i64(seed))
            u.append(lsr(z, 11).to(torch.float64) * (1.0 / 9007199254740992.0))
        p = [u[a] * extent for a in range(3)]
        d = [pa - c for pa in p]
        ln = torch.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
        ln = torch.where(ln == 0, torch.ones_like(ln), ln)
        r = r_lo + (r_hi - r_lo) * u[3]
        v = out[lo:lo + m]
        for a in range(3):
            v[:, a] = p[a].to(torch.float32)
            v[:, 4 + a] = (d[a] / ln).to(torch.float32)
        v[:, 3] = r.to(torch.float32)
        v[:, 7] = (1.0 / (r * r)).to(torch.float32)
    return out


def shells_cloud_device(n, extent, spacing, r_lo, r_hi, seed, device, first=0, chunk=16_000_000):
    """shells_cloud on a torch device.  Same stream and formulas; cos / sin of the device's float64 library may differ
    from numpy's in the last place, so this is the same distribution, not guaranteed the same bits (bench.py only)."""
    import math

    import torch
    out = torch.empty((n, 8), dtype=torch.float32, device=device)
    golden, m1, m2 = _i64(0x9E3779B97F4A7C15), _i64(0xBF58476D1CE4E5B9), _i64(0x94D049BB133111EB)

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)

    def mix(z):
        z = (z ^ lsr(z, 30)) * m1
        z = (z ^ lsr(z, 27)) * m2
        return z ^ lsr(z, 31)
    kmax = max(int((0.5 * extent - 4) // spacing), 1)
    w = np.arange(1, kmax + 1, dtype=np.float64) ** 2
    cdf = torch.tensor(np.cumsum(w) / w.sum(), dtype=torch.float64, device=device)
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        ids = torch.arange(first + lo, first + lo + m, dtype=torch.int64, device=device)
        u = []
        for j in range(5):
            z = mix((ids * 8 + (j + 1)) * golden + _i64(seed))
            u.append(lsr(z, 11).to(torch.float64) * (1.0 / 9007199254740992.0))
        k = (torch.searchsorted(cdf, u[4], right=True) + 1).to(torch.float64)
        zz = 2.0 * u[0] - 1.0
        t = (2.0 * u[1] - 1.0) * math.pi
        xy = torch.sqrt(1.0 - zz * zz)
        nrm = [torch.cos(t) * xy, torch.sin(t) * xy, zz]
        v = out[lo:lo + m]
        for a in range(3):
            v[:, 4 + a] = nrm[a].to(torch.float32)
            v[:, a] = (0.5 * extent + nrm[a] * (spacing * k)).to(torch.float32)
        v[:, 3] = (r_lo + (r_hi - r_lo) * u[2]).to(torch.float32)
        v[:, 7] = (1.0 - u[3]).to(torch.float32)
    return out


def make_cloud_device(cfg, device, scale=1.0, seed_offset=0, dist="uniform"):
    """make_cloud(cfg, "uniform") generated on `device`; returns ((n, 8) float32 tensor, grid corners per side)."""
    c = CONFIGS[cfg]
    g = c["grid"]
    n = max(int(c["splats"] * scale), 1)
    if dist == "shells":
        return shells_cloud_device(n, float(g - 1), 16.0, 1.0, 2.0, cloud_seed(cfg, seed_offset), device), g
    return uniform_cloud_device(n, float(g - 1), 2.0, 3.0, cloud_seed(cfg, seed_offset), device), g


PLY_ROW = np.dtype([("p", "<f4", 3), ("n", "<f4", 3), ("r", "<f4")])     # x y z nx ny nz radius, doc/mlsgpu-user-manual.xml:178-197


def ply_header(count):
    return ("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % count
            + "".join("property float32 %s\n" % n for n in ("x", "y", "z", "nx", "ny", "nz", "radius")) + "end_header\n").encode("ascii")


def cloud_ply_sizes(num_files, cfg, scale=1.0):
    """The exact size in bytes of each file write_cloud_ply makes: header + 28 bytes per splat of the file's id range."""
    n = max(int(CONFIGS[cfg]["splats"] * scale), 1)
    base, extra = divmod(n, num_files)
    return [len(ply_header(base + (1 if k < extra else 0))) + PLY_ROW.itemsize * (base + (1 if k < extra else 0))
            for k in range(num_files)]


def write_cloud_ply(paths, cfg, device, scale=1.0, dist="uniform", chunk=16_000_000):
    """Writes the cloud of a BASELINE config as len(paths) binary PLY files (equal, consecutive id ranges), generating it on
    `device` chunk by chunk -- the 10^9 splats of cfg5 never exist in one piece outside the files (28 GB).  One writer
    thread per file (file writes and device-to-host copies release the GIL).  Returns the number of splats written."""
    import threading

    import torch
    c = CONFIGS[cfg]
    g = c["grid"]
    n = max(int(c["splats"] * scale), 1)
    seed = cloud_seed(cfg)
    base, extra = divmod(n, len(paths))
    ranges, lo = [], 0
    for k in range(len(paths)):
        cnt = base + (1 if k < extra else 0)
        ranges.append((lo, cnt))
        lo += cnt
    gen_lock = threading.Lock()        # one chunk on the device at a time: generation is not the slow part
    errors = []

    def write_one(path, first, count):
        try:
            with open(path, "wb") as f:
                f.write(ply_header(count))
                for lo in range(0, count, chunk):
                    m = min(chunk, count - lo)
                    with gen_lock:
                        if dist == "shells":
                            t = shells_cloud_device(m, float(g - 1), 16.0, 1.0, 2.0, seed, device, first=first + lo)
                        else:
                            t = uniform_cloud_device(m, float(g - 1), 2.0, 3.0, seed, device, first=first + lo)
                        rows = t[:, [0, 1, 2, 4, 5, 6, 3]].contiguous().cpu()
                        del t
                    f.write(rows.numpy().tobytes())
        except Exception as e:      # noqa: BLE001 - reported by the caller's thread
            errors.append(e)
    threads = [threading.Thread(target=write_one, args=(p, r[0], r[1])) for p, r in zip(paths, ranges)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return n


def grid_buckets(dims, max_cells=255, runs=None):
    """The fixed spatial split of bucketize() for a grid of `dims` corners per axis: [(low, num_vertices)] in z-major,
    x-fastest order.  runs[a] overrides the number of runs along axis a."""
    per_axis = []
    for a in range(3):
        cells = dims[a] - 1
        if runs is not None and runs[a]:
            base, extra = divmod(cells, runs[a])
            lo, r = 0, []
            for i in range(runs[a]):
                k = base + (1 if i < extra else 0)
                r.append((lo, lo + k))
                lo += k
            assert max(h - l for l, h in r) <= max_cells
            per_axis.append(r)
        else:
            per_axis.append(split_axis(cells, max_cells))
    out = []
    for (z0, z1) in per_axis[2]:
        for (y0, y1) in per_axis[1]:
            for (x0, x1) in per_axis[0]:
                out.append(((x0, y0, z0), (x1 - x0 + 1, y1 - y0 + 1, z1 - z0 + 1)))
    return out


SLAB = 128     # corner slices per GPU of the sharded cfg4 workload (bench.py --gpus N)


def slab_boxes(grid, world, rank, slab=SLAB):
    """Rank `rank`'s buckets when the first `world` slabs of a grid x grid x ... cloud are dealt one per GPU: the grid
    is grid x grid x (slab * world) corners, cut `world` ways along z (all slabs but the last have `slab` cell slices,
    the last slab - 1), each slab cut into 255-cell runs along x and y."""
    boxes = grid_buckets((grid, grid, slab * world), 255, runs=(0, 0, world))
    per = len(boxes) // world
    assert per * world == len(boxes)
    return boxes[rank * per:(rank + 1) * per]


def slab_variant(world, rank):
    """A slab's geometry depends on the job only through being its last slab or not: "last" (slab - 1 cell slices: the
    grid ends there) or "inner" (slab cell slices, the next slab starts on its top corner slice)."""
    return "last" if rank == world - 1 else "inner"


def bucketize_device(cloud, boxes):
    """bucketize() for a cloud resident on a torch device and an explicit list of boxes [(low, num_vertices)] (a
    rank's share of grid_buckets): every box receives, in global order, all splats whose bounding box [p - r, p + r]
    meets its vertex range.  Returns ((total, 8) float32 tensor, [Bucket])."""
    import torch
    pos, rad = cloud[:, 0:3], cloud[:, 3:4]
    lo, hi = pos - rad, pos + rad
    cache = {}

    def axis_mask(a, r0, r1):
        key = (a, r0, r1)
        if key not in cache:
            cache[key] = (hi[:, a] >= r0) & (lo[:, a] <= r1)
        return cache[key]
    pieces, buckets, first = [], [], 0
    for low, nv in boxes:
        m = axis_mask(2, low[2], low[2] + nv[2] - 1) & axis_mask(1, low[1], low[1] + nv[1] - 1) \
            & axis_mask(0, low[0], low[0] + nv[0] - 1)
        idx = torch.nonzero(m).squeeze(1)
        pieces.append(cloud.index_select(0, idx))
        buckets.append(Bucket(tuple(low), tuple(nv), first, int(idx.numel())))
        first += int(idx.numel())
    return (torch.cat(pieces) if pieces else cloud[:0]), buckets


def make_cloud(cfg, dist="uniform", scale=1.0, seed_offset=0):
    """Cloud of a BASELINE config.  `scale` < 1 shrinks the splat count (tests); the grid stays."""
    c = CONFIGS[cfg]
    g = c["grid"]
    n = max(int(c["splats"] * scale), 1)
    seed = cloud_seed(cfg, seed_offset)
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


def to_host_splats(bucketed_t):
    """A (n, 8) float32 splat tensor on a GPU -> a host array of SPLAT_DTYPE in memory of its own mapping, advised to use
    huge pages and touched here (so on the caller's NUMA node): the stand-in for what a loader has in memory when it
    hands buckets to the farm.  An ordinary allocation made late in a long-lived process sits on whatever 4 KB pages the
    allocator has left, and copying out of it runs at half the rate (profiles/NOTES_r04.md 9.10)."""
    import mmap
    import torch
    n = int(bucketed_t.shape[0])
    nbytes = max(n * 32, 1)
    mem = mmap.mmap(-1, (nbytes + (2 << 20) - 1) & ~((2 << 20) - 1))
    try:
        mem.madvise(mmap.MADV_HUGEPAGE)
    except (AttributeError, OSError, ValueError):
        pass
    flat = np.frombuffer(mem, dtype=np.float32, count=n * 8).reshape(n, 8)
    step = 8_000_000
    for lo in range(0, n, step):
        torch.from_numpy(flat[lo:lo + step]).copy_(bucketed_t[lo:lo + step])
    return flat.view(SPLAT_DTYPE).reshape(-1)
