/*
 * mlsgpu CPU ORACLE, host bucketing -- TEST INFRASTRUCTURE ONLY.
 *
 * A from-scratch CPU restatement of Bucket::bucket (src/bucket.h:116-180) for an in-memory splat array:
 * the recursion of src/bucket_impl.h:439-560, the per-region state of src/bucket.cpp:133-331 and the
 * splat -> microblock mapping of src/splat_set.cpp:52-72 / src/grid.cpp:108-129.  The reference keeps its
 * node counts delta-encoded in hash maps (bucket.cpp:207-246) and its subsets as id ranges; this
 * restatement keeps plain per-level dense counts and explicit id lists, which hold the same values
 * (a node's count = number of splats whose microblock range meets it; a subset = those ids, ascending).
 *
 * Nothing in the product path may include, link or call this file (see mlsgpu_oracle.cpp).
 *
 * PARITY PINNING: tests/test_oracle_bucket.py restates the reference's known answers for this path --
 * test/test_bucket.cpp:106-171 (Node), :214-238 (forEachNode order), :453-560 (11 / 11 / 1 buckets,
 * DensityError, chunk alignment) and the `validate` properties (:345-451) on its 30 random cases -- and
 * test/test_splat_set.cpp's splatToBuckets vectors.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#define ORC_API extern "C" __attribute__((visibility("default")))

namespace
{

struct Splat
{
    float position[3];
    float radius;
    float normal[3];
    float quality;
};

/* src/grid.h: reference point, spacing and per-axis [first, second) extents in units of spacing */
struct Grid
{
    float reference[3];
    float spacing;
    int32_t lo[3], hi[3];
    uint32_t numCells(int i) const { return (uint32_t) (hi[i] - lo[i]); }
};

typedef int (*LeafFn)(void *user, const int32_t extents[6], const uint64_t chunk[3], uint32_t depth,
                      uint64_t numSplats, const uint64_t *ids);

int64_t divDown(int64_t a, int64_t b) { int64_t q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
uint64_t divUp(uint64_t a, uint64_t b) { return (a + b - 1) / b; }
uint64_t mulSat(uint64_t a, uint64_t b)
{
    if (a == 0 || b == 0) return 0;
    return a > std::numeric_limits<uint64_t>::max() / b ? std::numeric_limits<uint64_t>::max() : a * b;
}

bool isFinite(const Splat &s)
{
    /* Splat::isFinite, src/splat.h:49-59: splat sets never enumerate the others (src/splat_set.h:191) */
    return std::isfinite(s.position[0]) && std::isfinite(s.position[1]) && std::isfinite(s.position[2])
        && std::isfinite(s.radius) && std::isfinite(s.normal[0]) && std::isfinite(s.normal[1])
        && std::isfinite(s.normal[2]) && std::isfinite(s.quality);
}

/* splatToBuckets, src/splat_set.cpp:52-72 with Grid::worldToCell, src/grid.cpp:108-129 */
void splatToBuckets(const Splat &s, const Grid &g, uint32_t bucketSize, int64_t lower[3], int64_t upper[3])
{
    const float inv = 1.0f / g.spacing;
    for (int i = 0; i < 3; i++)
    {
        const float loWorld = s.position[i] - s.radius, hiWorld = s.position[i] + s.radius;
        const int64_t lo = (int64_t) std::floor((loWorld - g.reference[i]) * inv) - g.lo[i];
        const int64_t hi = (int64_t) std::floor((hiWorld - g.reference[i]) * inv) - g.lo[i];
        lower[i] = divDown(lo, bucketSize);
        upper[i] = divDown(hi, bucketSize);
    }
}

/* Bucket::detail::Node, src/bucket_internal.h + src/bucket.cpp:61-131 */
struct Node
{
    uint32_t c[3];
    uint32_t level;
    uint32_t size() const { return 1u << level; }
    Node child(unsigned idx) const { return Node{{c[0] * 2 + (idx & 1), c[1] * 2 + ((idx >> 1) & 1), c[2] * 2 + (idx >> 2)}, level - 1}; }
};

template<typename F>
void forEachNodeR(const uint32_t dims[3], const Node &node, const F &f)      /* bucket_impl.h:77-101 */
{
    if (!f(node) || node.level == 0)
        return;
    for (unsigned i = 0; i < 8; i++)
    {
        const Node ch = node.child(i);
        bool inside = true;
        for (int j = 0; j < 3; j++)
            if (((uint64_t) ch.c[j] << ch.level) >= dims[j])
                inside = false;
        if (inside)
            forEachNodeR(dims, ch, f);
    }
}

/* chooseMicroSize, src/bucket.cpp:354-377 */
uint32_t chooseMicroSize(const uint32_t dims[3], uint64_t maxSplit, uint64_t numSplats, uint64_t maxSplats, uint32_t maxCells)
{
    uint32_t microSize = 1;
    uint64_t microBlocks = 1;
    for (int i = 0; i < 3; i++)
        microBlocks = mulSat(microBlocks, divUp(dims[i], microSize));
    const double target = 0.5 * *std::min_element(dims, dims + 3) * std::sqrt((double) maxSplats / (double) numSplats);
    while (microBlocks > maxSplit || (microBlocks > 8 && microSize < target * 0.5 && (uint64_t) microSize * 2 <= maxCells))
    {
        microSize *= 2;
        microBlocks = 1;
        for (int i = 0; i < 3; i++)
            microBlocks = mulSat(microBlocks, divUp(dims[i], microSize));
    }
    return microSize;
}

struct Params
{
    uint64_t maxSplats;
    uint32_t maxCells;
    uint64_t maxSplit;
};

struct Run
{
    const Splat *splats;
    LeafFn leaf;
    void *user;
    int error;              /* 0, 1 = DensityError, 2 = callback failed */
    uint64_t cellSplats;
};

/* One BucketState (src/bucket.cpp:133-331): the region of one chunk */
struct State
{
    Grid grid;
    uint32_t microSize;
    int macroLevels;
    uint32_t dims[3];
    std::vector<std::vector<int64_t> > counts;     /* [level][(z*dy + y)*dx + x] */
    std::vector<uint32_t> levelDims;               /* 3 per level */
    std::vector<Node> regions;
    std::vector<int32_t> regionOf;                 /* level-0 microblock -> region id, -1 none */
    std::vector<std::vector<uint64_t> > members;

    int64_t &at(int level, uint32_t x, uint32_t y, uint32_t z)
    {
        const uint32_t *d = &levelDims[3 * level];
        return counts[level][((uint64_t) z * d[1] + y) * d[0] + x];
    }
};

bool clampRange(const State &st, const int64_t lower[3], const int64_t upper[3], uint32_t lo[3], uint32_t hi[3])
{
    for (int i = 0; i < 3; i++)         /* BucketState::clamp, src/bucket.cpp:176-194 */
    {
        int64_t l = lower[i], h = upper[i];
        if (l < 0) l = 0;
        if (h >= (int64_t) st.dims[i]) h = (int64_t) st.dims[i] - 1;
        if (l > h) return false;
        lo[i] = (uint32_t) l;
        hi[i] = (uint32_t) h;
    }
    return true;
}

void recurse(Run &run, const std::vector<uint64_t> *subset, uint64_t numAll, const Grid &grid, const Params &P,
             uint32_t chunkCells, uint32_t microCells, uint32_t depth, const uint64_t chunkIn[3]);

/* bucketRecurse, src/bucket_impl.h:439-560.  subset == nullptr: the whole array (not a subset type, so the
 * callback branch is not taken at the top level, bucket_impl.h:410-418). */
void recurse(Run &run, const std::vector<uint64_t> *subset, uint64_t numAll, const Grid &grid, const Params &P,
             uint32_t chunkCells, uint32_t microCells, uint32_t depth, const uint64_t chunkIn[3])
{
    if (run.error)
        return;
    uint32_t cellDims[3];
    for (int i = 0; i < 3; i++)
        cellDims[i] = grid.numCells(i);
    const uint32_t maxCellDim = std::max(std::max(cellDims[0], cellDims[1]), cellDims[2]);
    const uint64_t maxSplatsHere = subset ? subset->size() : numAll;

    if (subset && maxSplatsHere <= P.maxSplats && maxCellDim <= P.maxCells && (chunkCells == 0 || chunkCells >= maxCellDim))
    {
        const int32_t ext[6] = {grid.lo[0], grid.hi[0], grid.lo[1], grid.hi[1], grid.lo[2], grid.hi[2]};
        if (run.leaf(run.user, ext, chunkIn, depth, subset->size(), subset->data()) != 0)
            run.error = 2;
        return;
    }
    if (maxCellDim == 1)
    {
        run.error = 1;
        run.cellSplats = maxSplatsHere;
        return;
    }
    uint32_t microSize = microCells;
    if (microSize == 0 || microSize > maxCellDim)
        microSize = chooseMicroSize(cellDims, P.maxSplit, maxSplatsHere, P.maxSplats, P.maxCells);
    while (true)        /* coarsen until few enough microblocks */
    {
        uint64_t microBlocks = 1;
        for (int i = 0; i < 3; i++)
            microBlocks = mulSat(microBlocks, divUp(cellDims[i], microSize));
        if (microBlocks <= P.maxSplit)
            break;
        microSize *= 2;
    }
    if (chunkCells == 0)
        chunkCells = maxCellDim;
    else
        chunkCells = std::min(maxCellDim, chunkCells);
    if (chunkCells > P.maxCells)
    {
        uint64_t grain = (uint64_t) P.maxCells / microSize * microSize;
        if (grain == 0)
            grain = microSize;
        chunkCells = (uint32_t) (divUp(chunkCells, grain) * grain);
    }
    else
        chunkCells = (uint32_t) (divUp(chunkCells, microSize) * microSize);
    uint32_t chunks[3];
    for (int i = 0; i < 3; i++)
        chunks[i] = (uint32_t) divUp(cellDims[i], chunkCells);
    int macroLevels = 1;
    while (((uint64_t) microSize << (macroLevels - 1)) < chunkCells)
        macroLevels++;
    const uint32_t chunkRatio = chunkCells / microSize;

    /* BucketStateSet, src/bucket.cpp:304-331 */
    std::vector<State> states((size_t) chunks[0] * chunks[1] * chunks[2]);
    auto stateAt = [&](uint32_t cx, uint32_t cy, uint32_t cz) -> State & { return states[((size_t) cz * chunks[1] + cy) * chunks[0] + cx]; };
    for (uint32_t cz = 0; cz < chunks[2]; cz++)
        for (uint32_t cy = 0; cy < chunks[1]; cy++)
            for (uint32_t cx = 0; cx < chunks[0]; cx++)
            {
                State &st = stateAt(cx, cy, cz);
                st.grid = grid;
                const uint32_t cc[3] = {cx, cy, cz};
                for (int i = 0; i < 3; i++)
                {
                    const int64_t off = (int64_t) cc[i] * chunkCells;
                    st.grid.lo[i] = (int32_t) (grid.lo[i] + off);
                    st.grid.hi[i] = (int32_t) std::min<int64_t>(grid.lo[i] + off + chunkCells, grid.hi[i]);
                }
                st.microSize = microSize;
                st.macroLevels = macroLevels;
                for (int i = 0; i < 3; i++)
                    st.dims[i] = (uint32_t) divUp(st.grid.numCells(i), microSize);
                st.counts.resize(macroLevels);
                st.levelDims.resize(3 * macroLevels);
                for (int l = 0; l < macroLevels; l++)
                {
                    uint64_t total = 1;
                    for (int i = 0; i < 3; i++)
                    {
                        st.levelDims[3 * l + i] = (uint32_t) divUp(st.dims[i], (uint64_t) 1 << l);
                        total *= st.levelDims[3 * l + i];
                    }
                    st.counts[l].assign(total, 0);
                }
            }

    /* every (chunk, clamped microblock range) a splat falls into: BucketStateSet::processBlob, bucket_impl.h:302-327 */
    auto forEachChunk = [&](uint64_t id, auto &&fn)
    {
        const Splat &s = run.splats[id];
        if (!isFinite(s))
            return;
        int64_t lower[3], upper[3];
        splatToBuckets(s, grid, microSize, lower, upper);
        int64_t cl[3], cu[3];
        for (int i = 0; i < 3; i++)
        {
            cl[i] = std::max<int64_t>(divDown(lower[i], chunkRatio), 0);
            cu[i] = std::min<int64_t>(divDown(upper[i], chunkRatio), (int64_t) chunks[i] - 1);
        }
        for (int64_t cx = cl[0]; cx <= cu[0]; cx++)
            for (int64_t cy = cl[1]; cy <= cu[1]; cy++)
                for (int64_t cz = cl[2]; cz <= cu[2]; cz++)
                {
                    int64_t sl[3], su[3];
                    const int64_t cc[3] = {cx, cy, cz};
                    for (int i = 0; i < 3; i++)
                    {
                        sl[i] = lower[i] - cc[i] * chunkRatio;
                        su[i] = upper[i] - cc[i] * chunkRatio;
                    }
                    State &st = stateAt((uint32_t) cx, (uint32_t) cy, (uint32_t) cz);
                    uint32_t lo[3], hi[3];
                    if (clampRange(st, sl, su, lo, hi))
                        fn(st, lo, hi);
                }
    };
    auto forEachSplat = [&](auto &&fn)
    {
        if (subset)
            for (uint64_t id : *subset)
                fn(id);
        else
            for (uint64_t id = 0; id < numAll; id++)
                fn(id);
    };

    /* countSplats + upsweepCounts (src/bucket.cpp:207-246, 161-174): a node's count is the number of splats whose
     * range meets it */
    forEachSplat([&](uint64_t id)
    {
        forEachChunk(id, [&](State &st, const uint32_t lo[3], const uint32_t hi[3])
        {
            for (int l = 0; l < st.macroLevels; l++)
                for (uint32_t z = lo[2] >> l; z <= (hi[2] >> l); z++)
                    for (uint32_t y = lo[1] >> l; y <= (hi[1] >> l); y++)
                        for (uint32_t x = lo[0] >> l; x <= (hi[0] >> l); x++)
                            st.at(l, x, y, z) += 1;
        });
    });

    /* pickNodes, src/bucket.cpp:248-269 + PickNodes :333-352 */
    for (State &st : states)
    {
        const uint64_t n0 = (uint64_t) st.dims[0] * st.dims[1] * st.dims[2];
        st.regionOf.assign(n0, -1);
        auto pick = [&](const Node &node) -> bool
        {
            const int64_t count = st.at((int) node.level, node.c[0], node.c[1], node.c[2]);
            if (count == 0)
                return false;
            if (node.level == 0 || ((uint64_t) st.microSize * node.size() <= P.maxCells && (uint64_t) count <= P.maxSplats))
            {
                const int32_t id = (int32_t) st.regions.size();
                st.regions.push_back(node);
                for (uint32_t z = node.c[2] << node.level; z < std::min(st.dims[2], (node.c[2] + 1) << node.level); z++)
                    for (uint32_t y = node.c[1] << node.level; y < std::min(st.dims[1], (node.c[1] + 1) << node.level); y++)
                        for (uint32_t x = node.c[0] << node.level; x < std::min(st.dims[0], (node.c[0] + 1) << node.level); x++)
                            st.regionOf[((uint64_t) z * st.dims[1] + y) * st.dims[0] + x] = id;
                return false;
            }
            return true;
        };
        forEachNodeR(st.dims, Node{{0, 0, 0}, (uint32_t) (st.macroLevels - 1)}, pick);
        st.members.resize(st.regions.size());
    }

    /* bucketSplats, src/bucket.cpp:271-302: once per region a splat touches */
    forEachSplat([&](uint64_t id)
    {
        forEachChunk(id, [&](State &st, const uint32_t lo[3], const uint32_t hi[3])
        {
            for (uint32_t x = lo[0]; x <= hi[0]; x++)
                for (uint32_t y = lo[1]; y <= hi[1]; y++)
                    for (uint32_t z = lo[2]; z <= hi[2]; z++)
                    {
                        const int32_t r = st.regionOf[((uint64_t) z * st.dims[1] + y) * st.dims[0] + x];
                        const uint32_t mask = st.regions[r].size() - 1;
                        if ((x == lo[0] || (x & mask) == 0) && (y == lo[1] || (y & mask) == 0) && (z == lo[2] || (z & mask) == 0))
                            st.members[r].push_back(id);
                    }
        });
    });

    /* doCallbacks, src/bucket_impl.h:258-294, chunks x-major (:548-553) */
    for (uint32_t cx = 0; cx < chunks[0]; cx++)
        for (uint32_t cy = 0; cy < chunks[1]; cy++)
            for (uint32_t cz = 0; cz < chunks[2]; cz++)
            {
                State &st = stateAt(cx, cy, cz);
                const uint64_t chunk[3] = {chunkIn[0] + cx, chunkIn[1] + cy, chunkIn[2] + cz};
                for (size_t r = 0; r < st.regions.size(); r++)
                {
                    const Node &node = st.regions[r];
                    Grid child = st.grid;
                    for (int i = 0; i < 3; i++)     /* Node::toCells clipped to the grid, src/bucket.cpp:113-122 */
                    {
                        const uint64_t lower = std::min<uint64_t>(((uint64_t) st.microSize * node.c[i]) << node.level, st.grid.numCells(i));
                        const uint64_t upper = std::min<uint64_t>((((uint64_t) st.microSize * node.c[i]) << node.level)
                                                                  + ((uint64_t) st.microSize << node.level), st.grid.numCells(i));
                        child.lo[i] = (int32_t) (st.grid.lo[i] + (int64_t) lower);
                        child.hi[i] = (int32_t) (st.grid.lo[i] + (int64_t) upper);
                    }
                    recurse(run, &st.members[r], numAll, child, P, 0, 0, depth + 1, chunk);
                    std::vector<uint64_t>().swap(st.members[r]);
                }
            }
}

} // namespace

/* Bucket::bucket over splats[0..n).  grid: reference[3], spacing; extents[6] = lo/hi per axis.
 * Returns 0, 1 (DensityError, *cellSplats set) or 2 (callback failed). */
ORC_API int orc_bucket_partition(const void *splats, uint64_t n, const float reference[3], float spacing, const int32_t extents[6],
                                 uint64_t maxSplats, uint32_t maxCells, uint32_t chunkCells, uint32_t microCells, uint64_t maxSplit,
                                 LeafFn leaf, void *user, uint64_t *cellSplats)
{
    Run run{static_cast<const Splat *>(splats), leaf, user, 0, 0};
    for (int i = 0; i < 3; i++)
        if (extents[2 * i] >= extents[2 * i + 1])
            return 3;       /* the reference divides by zero on a region without cells */
    Grid g;
    for (int i = 0; i < 3; i++)
    {
        g.reference[i] = reference[i];
        g.lo[i] = extents[2 * i];
        g.hi[i] = extents[2 * i + 1];
    }
    g.spacing = spacing;
    const Params P{maxSplats, maxCells, maxSplit};
    const uint64_t chunk[3] = {0, 0, 0};
    recurse(run, nullptr, n, g, P, chunkCells, microCells, 0, chunk);
    if (cellSplats)
        *cellSplats = run.cellSplats;
    return run.error;
}

ORC_API void orc_splat_to_buckets(const void *splat, const float reference[3], float spacing, const int32_t extents[6],
                                  uint32_t bucketSize, int64_t lower[3], int64_t upper[3])
{
    Grid g;
    for (int i = 0; i < 3; i++)
    {
        g.reference[i] = reference[i];
        g.lo[i] = extents[2 * i];
        g.hi[i] = extents[2 * i + 1];
    }
    g.spacing = spacing;
    splatToBuckets(*static_cast<const Splat *>(splat), g, bucketSize, lower, upper);
}

/* forEachNode with the predicate of test/test_bucket.cpp:203-212 generalised: recurse into a node iff it contains
 * microblock `inside`.  Writes (x, y, z, level) per visited node; returns the number of nodes. */
ORC_API int orc_for_each_node(const uint32_t dims[3], uint32_t levels, const uint32_t inside[3], uint32_t *out, int maxOut)
{
    int n = 0;
    auto f = [&](const Node &node) -> bool
    {
        if (n < maxOut)
        {
            out[4 * n] = node.c[0]; out[4 * n + 1] = node.c[1]; out[4 * n + 2] = node.c[2]; out[4 * n + 3] = node.level;
        }
        n++;
        for (int i = 0; i < 3; i++)
        {
            const uint64_t lo = (uint64_t) node.c[i] << node.level, hi = lo + ((uint64_t) 1 << node.level);
            if (!(lo <= inside[i] && inside[i] < hi))
                return false;
        }
        return true;
    };
    forEachNodeR(dims, Node{{0, 0, 0}, levels - 1}, f);
    return n;
}

ORC_API void orc_node_child(const uint32_t node[4], uint32_t idx, uint32_t out[4])
{
    const Node c = Node{{node[0], node[1], node[2]}, node[3]}.child(idx);
    out[0] = c.c[0]; out[1] = c.c[1]; out[2] = c.c[2]; out[3] = c.level;
}

ORC_API uint32_t orc_choose_micro_size(const uint32_t dims[3], uint64_t maxSplit, uint64_t numSplats, uint64_t maxSplats, uint32_t maxCells)
{
    return chooseMicroSize(dims, maxSplit, numSplats, maxSplats, maxCells);
}
