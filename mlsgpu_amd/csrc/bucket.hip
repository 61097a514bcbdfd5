/*
 * Device-side bucketing: Bucket::bucket (src/bucket.h:116-180, src/bucket_impl.h:439-560, src/bucket.cpp:133-377)
 * for a splat cloud that is resident in HBM -- row f2 of SURVEY.md section 8.
 *
 * The reference streams blobs of an out-of-core splat set twice per recursion level through one host thread
 * (count into hash-mapped octree counters, then append id ranges to the chosen regions).  With 288 GB of HBM
 * the whole cloud (32 B per splat) stays on the device, so each level is:
 *   1. bucketCountKernel   every splat adds 1 to each octree node (all levels) its microblock range meets --
 *                          the values the reference's delta-encoded counters hold after upsweepCounts;
 *                          LDS-privatised when the dense octree fits (it does for the default 63-cell leaves);
 *      bucketCountPrivateKernel (a big level whose finest counters do not fit: 33^3 microblocks for 10^9 splats in 2048^3)
 *                          per-workgroup LDS counters, 16 bits wide at the finest level, corrections at the coarser ones,
 *                          summed and swept up by two small kernels -- no scattered global atomics;
 *      either way the count is the only pass that reads the splats: it leaves every element's microblock range as an
 *      8-byte note (RegionView::packNote) for step 3;
 *   2. host: pickNodes on the few hundred counters (same traversal order, so regions are numbered alike); the regions'
 *      counters are also the lengths of their member lists, so the lists' starts are known here;
 *   3. a scan whose producer counts the regions a splat joins ("once per node", bucket.cpp:291-301) and whose
 *      consumer writes (region, splat id) pairs, and a STABLE radix sort by region (one digit for up to 1024 regions,
 *      sorted keys not written): every region's member list, ids ascending -- exactly the reference's subset;
 *   4. recursion into each region (depth first, in region order) with its id list in place of all splats.
 * Leaves are handed to the callback with their id list still on the device; mlsgpu_hip_bucket_load gathers
 * them and transforms them into the full grid's vertex coordinates (BucketLoader, src/bucket_loader.cpp:77-85)
 * ready for mlsgpu_hip_worker_process -- no host round trip of splat data at all.
 */
#include "common.hpp"
#include "primitives.hpp"

#include <algorithm>
#include <cmath>
#include <functional>
#include <memory>

using namespace mlsgpu;

/* detail::Bbox of device-resident splats folded into lo / hi (defined with the bounding-grid entry points below) */
static int foldBbox(mlsgpu_ctx *ctx, const mlsgpu_splat *dSplats, uint64_t numSplats, float lo[3], float hi[3]);

namespace
{

enum { MAX_LEVELS = 32, LDS_NODES = 8192 };

/* what a kernel needs to map a splat to the microblocks of one chunk's region */
struct RegionView
{
    const mlsgpu_splat *splats;
    const uint32_t *ids;        /* nullptr: splat i is splat i */
    float ref[3];
    float invSpacing;
    int32_t first[3];           /* grid extents' lower ends (the grid being split, not the chunk) */
    uint32_t microSize;
    uint32_t divMagic, divShift;    /* x / microSize == (uint64_t) x * divMagic >> divShift for every x < 2^31 (setMicroSize) */
    int32_t bias[3];            /* chunk coordinate * chunkRatio, in microblocks */
    uint32_t dims[3];           /* microblocks of this chunk's region */
    const uint2 *notes = nullptr;   /* element i's microblock range as the counting pass packed it (packNote), or nullptr */

    __device__ __forceinline__ uint32_t splatId(uint64_t i) const { return ids ? ids[i] : (uint32_t) i; }

    /* Division by an invariant (Granlund & Montgomery): with L = ceil(log2 m) and M = ceil(2^(31 + L) / m) < 2^32,
     * M * m - 2^(31 + L) < m <= 2^L, which makes floor(x * M / 2^(31 + L)) the exact quotient for x < 2^31. */
    void setMicroSize(uint32_t m)
    {
        microSize = m;
        uint32_t L = 0;
        while (((uint64_t) 1 << L) < m)
            L++;
        divShift = 31 + L;
        divMagic = (uint32_t) ((((uint64_t) 1 << divShift) + m - 1) / m);
    }
    __device__ __forceinline__ uint32_t divMicro(uint32_t x) const { return (uint32_t) (((uint64_t) x * divMagic) >> divShift); }

    /* The counting pass reads every splat (32 B) and works its range out with six floor divisions; the two passes behind it
     * (how many regions does element i join; which) need the range only, so it is kept: 8 B per element, valid in bit 63,
     * lo in 16 bits and hi - lo in 5 bits per axis.  A range that does not fit is marked and worked out again. */
    static __device__ __forceinline__ uint2 packNote(bool ok, const uint32_t lo[3], const uint32_t hi[3])
    {
        if (!ok)
            return make_uint2(0u, 0u);
        const uint32_t s0 = hi[0] - lo[0], s1 = hi[1] - lo[1], s2 = hi[2] - lo[2];
        if ((lo[0] | lo[1] | lo[2]) > 0xFFFFu || (s0 | s1 | s2) > 31u)
            return make_uint2(0u, 0xFFFFFFFFu);
        return make_uint2(lo[0] | lo[1] << 16, lo[2] | s0 << 16 | s1 << 21 | s2 << 26 | 0x80000000u);
    }
    __device__ __forceinline__ bool rangeAt(uint64_t i, uint32_t lo[3], uint32_t hi[3]) const
    {
        if (notes != nullptr)
        {
            const uint2 w = notes[i];
            if (w.y != 0xFFFFFFFFu)
            {
                if (!(w.y >> 31))
                    return false;
                lo[0] = w.x & 0xFFFFu;
                lo[1] = w.x >> 16;
                lo[2] = w.y & 0xFFFFu;
                hi[0] = lo[0] + ((w.y >> 16) & 31u);
                hi[1] = lo[1] + ((w.y >> 21) & 31u);
                hi[2] = lo[2] + ((w.y >> 26) & 31u);
                return true;
            }
        }
        return range(splatId(i), lo, hi);
    }

    /* splatToBuckets (src/splat_set.cpp:52-72, Grid::worldToCell src/grid.cpp:108-129) + BucketStateSet's chunk
     * bias (bucket_impl.h:318-323) + BucketState::clamp (src/bucket.cpp:176-194) */
    __device__ __forceinline__ bool range(uint32_t id, uint32_t lo[3], uint32_t hi[3]) const
    {
        return rangeOf(splats[id], lo, hi);
    }
    __device__ __forceinline__ bool rangeOf(const mlsgpu_splat s, uint32_t lo[3], uint32_t hi[3]) const
    {
        /* 32-bit cell arithmetic is exact when the grid's lower ends and the microblock size leave room */
        const bool narrow = microSize <= (1u << 27) && abs(first[0]) < (1 << 29) && abs(first[1]) < (1 << 29) && abs(first[2]) < (1 << 29);
        if (!(isfinite(s.position[0]) && isfinite(s.position[1]) && isfinite(s.position[2]) && isfinite(s.radius)
              && isfinite(s.normal[0]) && isfinite(s.normal[1]) && isfinite(s.normal[2]) && isfinite(s.quality)))
            return false;       /* never enumerated by a splat set, src/splat_set.h:191 */
#pragma unroll
        for (int a = 0; a < 3; a++)
        {
            const float loWorld = s.position[a] - s.radius, hiWorld = s.position[a] + s.radius;
            const float fl = floorf((loWorld - ref[a]) * invSpacing), fh = floorf((hiWorld - ref[a]) * invSpacing);
            long long l, h;
            if (fabsf(fl) < 1.0e9f && fabsf(fh) < 1.0e9f && narrow)
            {
                /* the same floor division in 32 bits (|cell - first| < 2^31): a 64-bit division is ~150 instructions, six
                 * of them per splat made the three passes over the cloud compute-bound */
                const int32_t cl = (int32_t) fl - first[a], ch = (int32_t) fh - first[a];
                const uint32_t m = microSize;       /* |cl|, |ch| < 2^30 + 2^29 and m <= 2^27: every dividend below 2^31 */
                l = (long long) (cl >= 0 ? (int32_t) divMicro((uint32_t) cl) : -(int32_t) divMicro((uint32_t) -cl + m - 1)) - bias[a];
                h = (long long) (ch >= 0 ? (int32_t) divMicro((uint32_t) ch) : -(int32_t) divMicro((uint32_t) -ch + m - 1)) - bias[a];
            }
            else
            {
                const long long cl = (long long) fl - first[a];
                const long long ch = (long long) fh - first[a];
                const long long m = (long long) microSize;
                l = (cl >= 0 ? cl / m : -((-cl + m - 1) / m)) - bias[a];      /* divDown */
                h = (ch >= 0 ? ch / m : -((-ch + m - 1) / m)) - bias[a];
            }
            if (l < 0) l = 0;
            if (h >= (long long) dims[a]) h = (long long) dims[a] - 1;
            if (l > h)
                return false;
            lo[a] = (uint32_t) l;
            hi[a] = (uint32_t) h;
        }
        return true;
    }
};

struct LevelLayout
{
    uint32_t levels;
    uint32_t dims[MAX_LEVELS][3];
    uint32_t offset[MAX_LEVELS + 1];
};

/* countSplats + upsweepCounts (src/bucket.cpp:161-174, 207-246) without the delta encoding.
 * Levels >= ldsFrom (the coarse ones: few nodes, every splat of the cloud lands on them -- 10^9 atomic adds on ONE word for
 * the root of BASELINE configs[4]) are counted in LDS and flushed once per workgroup; the finer levels below, whose nodes
 * do not fit, take one global atomic per (splat, node), spread over tens of thousands of words. */
__global__ __launch_bounds__(256) void bucketCountKernel(RegionView V, LevelLayout L, uint32_t *counts, uint64_t n, uint32_t ldsFrom,
                                                         uint2 *notesOut)
{
    __shared__ uint32_t local[LDS_NODES];
    const uint32_t total = L.offset[L.levels];
    const uint32_t ldsBase = ldsFrom < L.levels ? L.offset[ldsFrom] : total;
    for (uint32_t i = threadIdx.x; i < total - ldsBase; i += blockDim.x)
        local[i] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
    {
        uint32_t lo[3], hi[3];
        const bool ok = V.range(V.splatId(i), lo, hi);
        if (notesOut != nullptr)
            notesOut[i] = RegionView::packNote(ok, lo, hi);
        if (!ok)
            continue;
        for (uint32_t l = 0; l < L.levels; l++)
        {
            const uint32_t dx = L.dims[l][0], dy = L.dims[l][1];
            for (uint32_t z = lo[2] >> l; z <= (hi[2] >> l); z++)
                for (uint32_t y = lo[1] >> l; y <= (hi[1] >> l); y++)
                    for (uint32_t x = lo[0] >> l; x <= (hi[0] >> l); x++)
                    {
                        const uint32_t node = L.offset[l] + (z * dy + y) * dx + x;
                        if (l >= ldsFrom)
                            atomicAdd(&local[node - ldsBase], 1u);
                        else
                            atomicAdd(&counts[node], 1u);
                    }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < total - ldsBase; i += blockDim.x)
        if (local[i] != 0)
            atomicAdd(&counts[ldsBase + i], local[i]);
}

/*
 * The same counters for a big cloud whose finest level does not fit the LDS table above: 10^9 splats in random order make
 * 1.25 * 10^9 scattered global atomic adds on the finest level alone, and scattered atomics execute at the memory side, one
 * 64-B request each (~18 * 10^9 / s: 56 ms for BASELINE configs[4]).  Here no counter leaves the workgroup before the end:
 *   - a workgroup takes a SPAN of at most 65535 consecutive elements, so a finest-level counter (at most one add per
 *     element) fits 16 bits: two counters per LDS word, 35 937 microblocks in 72 KB;
 *   - the coarser levels are not counted at all but CORRECTED, as the reference's delta-encoded counters are
 *     (src/bucket.cpp:207-246): the sum of a node's children counts a splat once per child it touches, so the node keeps
 *     sum(children touched - 1) over its splats, and count = sum of the children's counts - that.  Only a splat that
 *     straddles a boundary of the level below adds anything (2-3 cell radii against 63-cell microblocks: one in five at
 *     the first coarser level, almost none above), where the plain count takes one LDS add per level, all lanes on the
 *     same few words at the top;
 *   - the workgroup's LDS image goes out as one slice (coalesced stores, 1.5 B per element); bucketSliceSumKernel adds the
 *     slices up and bucketUpsweepKernel turns corrections into counts, level by level.
 * The counters are the same integers whichever way they are added up.
 */
enum { PRIV_THREADS = 1024, PRIV_SPAN = 63 * PRIV_THREADS, PRIV_LDS_BYTES = 160 * 1024 };

__global__ __launch_bounds__(PRIV_THREADS) void bucketCountPrivateKernel(RegionView V, LevelLayout L, uint32_t *slices, uint64_t n,
                                                                         uint2 *notesOut, uint32_t words0, uint32_t imageWords)
{
    extern __shared__ uint32_t image[];     /* [words0] finest level, 16 bits a counter; then the corrections of levels 1.. */
    for (uint32_t i = threadIdx.x; i < imageWords; i += PRIV_THREADS)
        image[i] = 0;
    __syncthreads();
    const uint32_t n0 = L.offset[1];        /* levels >= 2 on this path */
    const uint64_t first = (uint64_t) blockIdx.x * PRIV_SPAN;
    const uint64_t last = first + PRIV_SPAN < n ? first + PRIV_SPAN : n;
    /* the next round's splat is requested before this round's is looked at: 16 waves a CU, two 32-B loads a lane in flight */
    mlsgpu_splat cur = {}, next = {};
    if (first + threadIdx.x < last)
        cur = V.splats[V.splatId(first + threadIdx.x)];
    for (uint64_t i = first + threadIdx.x; i < last; i += PRIV_THREADS, cur = next)
    {
        if (i + PRIV_THREADS < last)
            next = V.splats[V.splatId(i + PRIV_THREADS)];
        uint32_t lo[3], hi[3];
        const bool ok = V.rangeOf(cur, lo, hi);
        if (notesOut != nullptr)
            notesOut[i] = RegionView::packNote(ok, lo, hi);
        if (!ok)
            continue;
        const uint32_t sx = hi[0] - lo[0], sy = hi[1] - lo[1], sz = hi[2] - lo[2];
        if ((sx | sy | sz) <= 1u)
        {
            /* at most two nodes per axis, at every level: no loops over nodes, whose trip counts differ from lane to lane
             * (one lane in five straddles a microblock boundary -- some lane of nearly every wave does, at most levels) */
            uint32_t dx = L.dims[0][0], dxy = dx * L.dims[0][1];
            uint32_t base = (lo[2] * L.dims[0][1] + lo[1]) * dx + lo[0];
#pragma unroll
            for (uint32_t c = 0; c < 8; c++)
                if ((c & 1u) <= sx && ((c >> 1) & 1u) <= sy && (c >> 2) <= sz)
                {
                    const uint32_t node = base + (c & 1u) + ((c >> 1) & 1u) * dx + (c >> 2) * dxy;
                    atomicAdd(&image[node >> 1], 1u << ((node & 1u) * 16u));
                }
            /* two children along an axis either share their parent (it keeps a correction, and the splat is one node wide
             * from there up) or not (nothing to correct along this axis yet: two nodes wide at the next level too); every
             * parent in the range sees the same number of children, so they all take the same correction */
            uint32_t cx = lo[0], cy = lo[1], cz = lo[2];
            uint32_t tx = sx, ty = sy, tz = sz;
            for (uint32_t l = 1; l < L.levels && (tx | ty | tz) != 0; l++)
            {
                const uint32_t px = tx & cx, py = ty & cy, pz = tz & cz;        /* 1: the two children have different parents */
                const uint32_t extra = (1u + (tx & ~px)) * (1u + (ty & ~py)) * (1u + (tz & ~pz)) - 1u;
                cx >>= 1;
                cy >>= 1;
                cz >>= 1;
                if (extra != 0)
                {
                    dx = L.dims[l][0];
                    dxy = dx * L.dims[l][1];
                    base = words0 + (L.offset[l] - n0) + (cz * L.dims[l][1] + cy) * dx + cx;
#pragma unroll
                    for (uint32_t c = 0; c < 8; c++)
                        if ((c & 1u) <= px && ((c >> 1) & 1u) <= py && (c >> 2) <= pz)
                            atomicAdd(&image[base + (c & 1u) + ((c >> 1) & 1u) * dx + (c >> 2) * dxy], extra);
                }
                tx = px;
                ty = py;
                tz = pz;
            }
            continue;
        }
        {
            const uint32_t dx = L.dims[0][0], dy = L.dims[0][1];
            for (uint32_t z = lo[2]; z <= hi[2]; z++)
                for (uint32_t y = lo[1]; y <= hi[1]; y++)
                    for (uint32_t x = lo[0]; x <= hi[0]; x++)
                    {
                        const uint32_t node = (z * dy + y) * dx + x;
                        atomicAdd(&image[node >> 1], 1u << ((node & 1u) * 16u));
                    }
        }
        for (uint32_t l = 1; l < L.levels; l++)
        {
            const uint32_t s = l - 1;
            const uint32_t cl[3] = {lo[0] >> s, lo[1] >> s, lo[2] >> s}, ch[3] = {hi[0] >> s, hi[1] >> s, hi[2] >> s};
            if (cl[0] == ch[0] && cl[1] == ch[1] && cl[2] == ch[2])
                break;          /* one node at the level below: one node from here up, nothing to correct */
            const uint32_t dx = L.dims[l][0], dy = L.dims[l][1];
            for (uint32_t z = cl[2] >> 1; z <= (ch[2] >> 1); z++)
                for (uint32_t y = cl[1] >> 1; y <= (ch[1] >> 1); y++)
                    for (uint32_t x = cl[0] >> 1; x <= (ch[0] >> 1); x++)
                    {
                        const uint32_t kx = min(ch[0], 2 * x + 1) - max(cl[0], 2 * x) + 1;
                        const uint32_t ky = min(ch[1], 2 * y + 1) - max(cl[1], 2 * y) + 1;
                        const uint32_t kz = min(ch[2], 2 * z + 1) - max(cl[2], 2 * z) + 1;
                        const uint32_t extra = kx * ky * kz - 1;
                        if (extra != 0)
                            atomicAdd(&image[words0 + (L.offset[l] - n0) + (z * dy + y) * dx + x], extra);
                    }
        }
    }
    __syncthreads();
    uint32_t *const out = slices + (uint64_t) blockIdx.x * imageWords;
    for (uint32_t i = threadIdx.x; i < imageWords; i += PRIV_THREADS)
        out[i] = image[i];
}

/* counts[finest level] and the corrections of the levels above it, summed over the slices: a workgroup takes 256 words of
 * SLICE_GROUP slices and adds its sums to the counters (consecutive words: coalesced atomics) */
enum { SLICE_GROUP = 64 };
__global__ __launch_bounds__(256) void bucketSliceSumKernel(const uint32_t *slices, uint32_t numSlices, uint32_t words0, uint32_t imageWords,
                                                            uint32_t n0, uint32_t *counts)
{
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    if (w >= imageWords)
        return;
    const uint32_t s0 = blockIdx.y * SLICE_GROUP, s1 = min(numSlices, s0 + SLICE_GROUP);
    uint32_t a = 0, b = 0;
    if (w < words0)
    {
        for (uint32_t s = s0; s < s1; s++)
        {
            const uint32_t v = slices[(uint64_t) s * imageWords + w];
            a += v & 0xFFFFu;
            b += v >> 16;
        }
        if (a != 0)
            atomicAdd(&counts[2 * w], a);
        if (b != 0 && 2 * w + 1 < n0)
            atomicAdd(&counts[2 * w + 1], b);
    }
    else
    {
        for (uint32_t s = s0; s < s1; s++)
            a += slices[(uint64_t) s * imageWords + w];
        if (a != 0)
            atomicAdd(&counts[n0 + (w - words0)], a);
    }
}

/* one workgroup: count = sum of the children's counts - the node's correction, coarser level by coarser level */
__global__ __launch_bounds__(1024) void bucketUpsweepKernel(LevelLayout L, uint32_t *counts)
{
    for (uint32_t l = 1; l < L.levels; l++)
    {
        const uint32_t dx = L.dims[l][0], dy = L.dims[l][1], dz = L.dims[l][2];
        const uint32_t cx = L.dims[l - 1][0], cy = L.dims[l - 1][1], cz = L.dims[l - 1][2];
        const uint32_t *const below = counts + L.offset[l - 1];
        uint32_t *const here = counts + L.offset[l];
        for (uint32_t node = threadIdx.x; node < dx * dy * dz; node += blockDim.x)
        {
            const uint32_t x = node % dx, y = node / dx % dy, z = node / (dx * dy);
            uint32_t sum = 0;
            for (uint32_t k = 0; k < 8; k++)
            {
                const uint32_t X = 2 * x + (k & 1u), Y = 2 * y + ((k >> 1) & 1u), Z = 2 * z + (k >> 2);
                if (X < cx && Y < cy && Z < cz)
                    sum += below[(Z * cy + Y) * cx + X];
            }
            here[node] = sum - here[node];
        }
        __threadfence();
        __syncthreads();
    }
}

/* the table entries of the (at most) 2 x 2 x 2 microblocks of a range: candidate c = bit 0 x, bit 1 y, bit 2 z */
__device__ __forceinline__ void corner8(const RegionView &V, const uint32_t *table, const uint32_t lo[3], const uint32_t hi[3],
                                        uint32_t t[8])
{
    const uint32_t base = (lo[2] * V.dims[1] + lo[1]) * V.dims[0] + lo[0];
    const uint32_t sx = hi[0] - lo[0], sy = hi[1] - lo[1], sz = hi[2] - lo[2];
#pragma unroll
    for (uint32_t c = 0; c < 8; c++)
    {
        const bool need = ((c & 1u) <= sx) && (((c >> 1) & 1u) <= sy) && ((c >> 2) <= sz);
        const uint32_t at = base + (c & 1u) + ((c >> 1) & 1u) * V.dims[0] + (c >> 2) * V.dims[0] * V.dims[1];
        t[c] = table[need ? at : 0u];
    }
}
/* does the splat join candidate c's region THROUGH candidate c (a region is joined once: through its first microblock in
 * the range along every axis, src/bucket.cpp:291-301)? */
__device__ __forceinline__ bool joins8(uint32_t c, uint32_t t, const uint32_t lo[3], const uint32_t hi[3], uint32_t rFirst, uint32_t rEnd)
{
    const uint32_t cx = c & 1u, cy = (c >> 1) & 1u, cz = c >> 2;
    if (cx > hi[0] - lo[0] || cy > hi[1] - lo[1] || cz > hi[2] - lo[2])
        return false;
    const uint32_t mask = (1u << (t & 31u)) - 1;
    return (t >> 5) - rFirst < rEnd - rFirst
        && (cx == 0 || ((lo[0] + 1) & mask) == 0) && (cy == 0 || ((lo[1] + 1) & mask) == 0) && (cz == 0 || ((lo[2] + 1) & mask) == 0);
}

/* bucketSplats, src/bucket.cpp:271-302: table[microblock] = region id << 5 | node level */
struct RegionCountIn
{
    RegionView V;
    const uint32_t *table;
    uint32_t rFirst = 0, rEnd = 0xFFFFFFFFu;    /* only regions [rFirst, rEnd) count (a batch of the streamed top level) */
    __device__ __forceinline__ uint32_t operator()(uint64_t i) const
    {
        uint32_t lo[3], hi[3];
        if (!V.rangeAt(i, lo, hi))
            return 0;
        uint32_t k = 0;
        if (((hi[0] - lo[0]) | (hi[1] - lo[1]) | (hi[2] - lo[2])) <= 1u)
        {
            /* at most two microblocks per axis (a splat is small against a microblock): the eight candidates' entries are
             * requested together -- one memory latency, where the loop below waits for each entry in turn.  A candidate
             * outside the range reads entry 0 instead (one cache line for all such lanes) and is not counted. */
            uint32_t t[8];
            corner8(V, table, lo, hi, t);
#pragma unroll
            for (uint32_t c = 0; c < 8; c++)
                k += joins8(c, t[c], lo, hi, rFirst, rEnd) ? 1u : 0u;
            return k;
        }
        for (uint32_t x = lo[0]; x <= hi[0]; x++)
            for (uint32_t y = lo[1]; y <= hi[1]; y++)
                for (uint32_t z = lo[2]; z <= hi[2]; z++)
                {
                    const uint32_t t = table[(z * V.dims[1] + y) * V.dims[0] + x];
                    const uint32_t mask = (1u << (t & 31u)) - 1;
                    if ((t >> 5) - rFirst < rEnd - rFirst
                        && (x == lo[0] || (x & mask) == 0) && (y == lo[1] || (y & mask) == 0) && (z == lo[2] || (z & mask) == 0))
                        k++;
                }
        return k;
    }
};

struct RegionEmitOut
{
    RegionView V;
    const uint32_t *table;
    uint32_t *keys, *vals;
    uint32_t rFirst = 0, rEnd = 0xFFFFFFFFu;    /* as RegionCountIn; keys are relative to rFirst */
    __device__ __forceinline__ void operator()(uint64_t i, uint32_t excl, uint32_t count) const
    {
        if (count == 0)
            return;
        const uint32_t id = V.splatId(i);
        uint32_t lo[3], hi[3];
        V.rangeAt(i, lo, hi);
        if (((hi[0] - lo[0]) | (hi[1] - lo[1]) | (hi[2] - lo[2])) <= 1u)
        {
            uint32_t t[8];
            corner8(V, table, lo, hi, t);
#pragma unroll
            for (uint32_t j = 0; j < 8; j++)
            {
                const uint32_t c = (j & 1u) << 2 | (j & 2u) | (j & 4u) >> 2;       /* x outermost, z innermost, as the loop below */
                if (joins8(c, t[c], lo, hi, rFirst, rEnd))
                {
                    keys[excl] = (t[c] >> 5) - rFirst;
                    vals[excl] = id;
                    excl++;
                }
            }
            return;
        }
        for (uint32_t x = lo[0]; x <= hi[0]; x++)
            for (uint32_t y = lo[1]; y <= hi[1]; y++)
                for (uint32_t z = lo[2]; z <= hi[2]; z++)
                {
                    const uint32_t t = table[(z * V.dims[1] + y) * V.dims[0] + x];
                    const uint32_t mask = (1u << (t & 31u)) - 1;
                    if ((t >> 5) - rFirst < rEnd - rFirst
                        && (x == lo[0] || (x & mask) == 0) && (y == lo[1] || (y & mask) == 0) && (z == lo[2] || (z & mask) == 0))
                    {
                        keys[excl] = (t >> 5) - rFirst;
                        vals[excl] = id;
                        excl++;
                    }
                }
    }
};

/* The streamed top level (mlsgpu_hip_bucket_stream): does splat i of the chunk in flight join a region of the batch being
 * assembled? ... */
struct BatchJoinIn
{
    RegionCountIn C;
    uint8_t *flags;             /* remembered for the scan's second phase (BatchFlagIn) */
    __device__ __forceinline__ uint32_t operator()(uint64_t i) const
    {
        const uint32_t j = C(i) != 0 ? 1u : 0u;
        flags[i] = (uint8_t) j;
        return j;
    }
};
struct BatchFlagIn
{
    const uint8_t *flags;
    __device__ __forceinline__ uint32_t operator()(uint64_t i) const { return flags[i]; }
};
/* ... and if so it is appended to the batch, in file order */
struct BatchCopyOut
{
    const mlsgpu_splat *chunk;
    mlsgpu_splat *batch;
    const uint32_t *base;       /* device: splats of the batch so far */
    __device__ __forceinline__ void operator()(uint64_t i, uint32_t excl, uint32_t joins) const
    {
        if (joins)
            batch[(uint64_t) *base + excl] = chunk[i];
    }
};
__global__ void addCountKernel(uint32_t *base, const uint32_t *add) { *base += *add; }

/* first pair of every region in the region-sorted list (every region has at least one member) */
__global__ void regionStartsKernel(const uint32_t *keys, uint64_t n, uint32_t *starts)
{
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    if (i == 0 || keys[i] != keys[i - 1])
        starts[keys[i]] = (uint32_t) i;
}

/* BucketLoader, src/bucket_loader.cpp:77-85 with Grid::worldToVertex (src/grid.cpp:99-106) */
__global__ void bucketLoadKernel(const mlsgpu_splat *splats, const uint32_t *ids, uint64_t n, float rx, float ry, float rz,
                                 float invSpacing, float lx, float ly, float lz, mlsgpu_splat *out)
{
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    mlsgpu_splat s = splats[ids ? ids[i] : (uint32_t) i];
    s.position[0] = (s.position[0] - rx) * invSpacing - lx;
    s.position[1] = (s.position[1] - ry) * invSpacing - ly;
    s.position[2] = (s.position[2] - rz) * invSpacing - lz;
    s.radius *= invSpacing;
    out[i] = s;
}

/* detail::Bbox of the finite splats (src/splat_set_impl.h:495-512): per-workgroup min of position - radius and max of
 * position + radius; the host folds the partial results (min / max are exact in any order) */
__global__ __launch_bounds__(256) void bboxKernel(const mlsgpu_splat *splats, uint64_t n, float *partial)
{
    __shared__ float sMin[3][4], sMax[3][4];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
    {
        const mlsgpu_splat s = splats[i];
        if (!(isfinite(s.position[0]) && isfinite(s.position[1]) && isfinite(s.position[2]) && isfinite(s.radius)
              && isfinite(s.normal[0]) && isfinite(s.normal[1]) && isfinite(s.normal[2]) && isfinite(s.quality)))
            continue;
#pragma unroll
        for (int a = 0; a < 3; a++)
        {
            lo[a] = fminf(lo[a], s.position[a] - s.radius);
            hi[a] = fmaxf(hi[a], s.position[a] + s.radius);
        }
    }
    const uint32_t wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; a++)
    {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
        {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, 64));
        }
        if (laneId() == 0)
        {
            sMin[a][wave] = lo[a];
            sMax[a][wave] = hi[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3)
    {
        const int a = threadIdx.x;
        partial[blockIdx.x * 6 + a] = fminf(fminf(sMin[a][0], sMin[a][1]), fminf(sMin[a][2], sMin[a][3]));
        partial[blockIdx.x * 6 + 3 + a] = fmaxf(fmaxf(sMax[a][0], sMax[a][1]), fmaxf(sMax[a][2], sMax[a][3]));
    }
}

/* levels with fewer elements than this go the plain way (tuning / test aids: MLSGPU_HIP_BUCKET_NOTES_FROM,
 * MLSGPU_HIP_BUCKET_PRIVATE_FROM; 0 switches a path on for every level, "off" switches it off) */
uint64_t thresholdFromEnv(const char *name, uint64_t deflt)
{
    const char *e = getenv(name);
    if (e == nullptr || *e == '\0')
        return deflt;
    if (strcmp(e, "off") == 0)
        return UINT64_MAX;
    return strtoull(e, nullptr, 10);
}
uint64_t notesFrom() { return thresholdFromEnv("MLSGPU_HIP_BUCKET_NOTES_FROM", 1u << 20); }
uint64_t privateFrom() { return thresholdFromEnv("MLSGPU_HIP_BUCKET_PRIVATE_FROM", 4u << 20); }

uint32_t bitsForCount(uint32_t count)
{
    uint32_t b = 1;
    while (b < 32 && (count - 1) >> b)
        b++;
    return b;
}

uint64_t mulSat(uint64_t a, uint64_t b)
{
    if (a == 0 || b == 0)
        return 0;
    return a > UINT64_MAX / b ? UINT64_MAX : a * b;
}

/* chooseMicroSize, src/bucket.cpp:354-377 */
uint32_t chooseMicroSize(const uint32_t dims[3], uint64_t maxSplit, uint64_t numSplats, uint64_t maxSplats, uint32_t maxCells)
{
    uint32_t microSize = 1;
    auto blocks = [&]()
    {
        uint64_t b = 1;
        for (int i = 0; i < 3; i++)
            b = mulSat(b, divUp(dims[i], microSize));
        return b;
    };
    uint64_t microBlocks = blocks();
    const double target = 0.5 * std::min(std::min(dims[0], dims[1]), dims[2]) * std::sqrt((double) maxSplats / (double) numSplats);
    while (microBlocks > maxSplit || (microBlocks > 8 && microSize < target * 0.5 && (uint64_t) microSize * 2 <= maxCells))
    {
        microSize *= 2;
        microBlocks = blocks();
    }
    return microSize;
}

struct GridBox
{
    int32_t lo[3], hi[3];
    uint32_t cells(int i) const { return (uint32_t) (hi[i] - lo[i]); }
};

/* device scratch of one recursion depth; a level's sorted member lists must outlive the recursion below it */
struct DepthBuffers
{
    uint32_t *keysA = nullptr, *valsA = nullptr, *keysB = nullptr, *valsB = nullptr;
    uint32_t *hist = nullptr, *tileSums = nullptr;
    uint64_t pairCap = 0, keysBCap = 0;
    uint32_t *counts = nullptr, *table = nullptr, *starts = nullptr;
    uint32_t nodeCap = 0, regionCap = 0;
    uint32_t *total = nullptr;
    uint32_t *scanSums = nullptr;
    uint64_t scanCap = 0;
    uint2 *notes = nullptr;             /* RegionView::packNote of every element of the level being split */
    uint64_t noteCap = 0;
    uint32_t *slices = nullptr;         /* bucketCountPrivateKernel's per-workgroup counters */
    uint64_t sliceCap = 0;
    /* events of callbacks that still read member lists of this level (mlsgpu_bucket::consumed): whatever overwrites the
     * level's buffers is ordered behind them on the GPU */
    std::vector<hipEvent_t> readers;
    ~DepthBuffers()
    {
        hipFree(scanSums);
        hipFree(notes);
        hipFree(slices);
        hipFree(keysA); hipFree(valsA); hipFree(keysB); hipFree(valsB); hipFree(hist); hipFree(tileSums);
        hipFree(counts); hipFree(table); hipFree(starts); hipFree(total);
    }
};

struct Bucketer
{
    mlsgpu_ctx *ctx;
    const mlsgpu_splat *dSplats;
    uint64_t numSplats;
    mlsgpu_grid full;
    mlsgpu_bucket_params P;
    mlsgpu_bucket_fn fn;
    void *user;
    uint64_t cellSplats = 0;
    typedef std::vector<std::unique_ptr<DepthBuffers> > DepthList;
    DepthList *depthList = nullptr;     /* lives in the context's scratch cache: allocating ~1 GB per call costs more than the kernels */

    /* the callbacks' read events (mlsgpu_bucket::consumed) of every level: this context's stream goes on behind them, and
     * the handles are forgotten -- the callers' events need not outlive the call */
    void settleReaders()
    {
        for (auto &level : *depthList)
        {
            for (hipEvent_t ev : level->readers)
                (void) hipStreamWaitEvent(ctx->stream, ev, 0);
            level->readers.clear();
        }
    }
    int ensure(uint32_t **p, size_t elems)
    {
        hipFree(*p);
        *p = nullptr;
        HIP_CHECK(hipMalloc((void **) p, std::max<size_t>(elems, 1) * sizeof(uint32_t)));
        return MLSGPU_OK;
    }

    int recurse(const uint32_t *dIds, uint64_t n, bool isSubset, const GridBox &grid, uint32_t chunkCells, uint32_t microCells,
                uint32_t depth, const uint64_t chunkIn[3]);
    /* the top level over a splat set that is NOT resident: see mlsgpu_hip_bucket_stream */
    int recurseStream(mlsgpu_fileset *files, uint64_t n, const GridBox &grid, uint64_t budget, uint64_t chunkSplats,
                      uint32_t readerThreads, uint64_t stats[4]);
};

/* bucketRecurse, src/bucket_impl.h:439-560 */
int Bucketer::recurse(const uint32_t *dIds, uint64_t n, bool isSubset, const GridBox &grid, uint32_t chunkCells,
                      uint32_t microCells, uint32_t depth, const uint64_t chunkIn[3])
{
    uint32_t cellDims[3];
    for (int i = 0; i < 3; i++)
        cellDims[i] = grid.cells(i);
    const uint32_t maxCellDim = std::max(std::max(cellDims[0], cellDims[1]), cellDims[2]);
    if (isSubset && n <= P.maxSplats && maxCellDim <= P.maxCells && (chunkCells == 0 || chunkCells >= maxCellDim))
    {
        mlsgpu_bucket b;
        for (int i = 0; i < 3; i++)
        {
            b.extents[2 * i] = grid.lo[i];
            b.extents[2 * i + 1] = grid.hi[i];
            b.chunk[i] = chunkIn[i];
        }
        b.depth = depth;
        b.numSplats = n;
        b.dIds = dIds;
        b.dSplats = dSplats;
        void *consumed = nullptr;
        b.consumed = &consumed;
        const int rc = fn(user, ctx, &b);
        if (rc != 0)
            return setError(MLSGPU_ERR_CALLBACK, "bucket callback failed with %d", rc);
        /* the list lives in the buffers of the level above (a leaf at depth 0 is the caller's own list) */
        if (consumed != nullptr && depth > 0 && depth - 1 < depthList->size())
            (*depthList)[depth - 1]->readers.push_back(static_cast<hipEvent_t>(consumed));
        return MLSGPU_OK;
    }
    if (maxCellDim == 1)
    {
        cellSplats = n;
        return setError(MLSGPU_ERR_DENSITY, "Too many splats covering one cell (%llu)", (unsigned long long) n);
    }
    uint32_t microSize = microCells;
    if (microSize == 0 || microSize > maxCellDim)
        microSize = chooseMicroSize(cellDims, P.maxSplit, n, P.maxSplats, P.maxCells);
    while (true)
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
        chunkCells = (uint32_t) ((chunkCells + grain - 1) / grain * grain);
    }
    else
        chunkCells = roundUp(chunkCells, microSize);
    uint32_t chunks[3];
    for (int i = 0; i < 3; i++)
        chunks[i] = divUp(cellDims[i], chunkCells);
    uint32_t macroLevels = 1;
    while (((uint64_t) microSize << (macroLevels - 1)) < chunkCells)
        macroLevels++;
    REQUIRE(macroLevels <= MAX_LEVELS, MLSGPU_ERR_LENGTH);
    const uint32_t chunkRatio = chunkCells / microSize;

    while (depthList->size() <= depth)
        depthList->emplace_back(new DepthBuffers);
    DepthBuffers &B = *(*depthList)[depth];
    if (B.total == nullptr)
        PROPAGATE(ensure(&B.total, 2));

    /* chunks x-major as the callbacks of bucket_impl.h:548-553; the chunks' states are independent */
    for (uint32_t cx = 0; cx < chunks[0]; cx++)
        for (uint32_t cy = 0; cy < chunks[1]; cy++)
            for (uint32_t cz = 0; cz < chunks[2]; cz++)
            {
                const uint32_t cc[3] = {cx, cy, cz};
                GridBox sub;            /* BucketStateSet, src/bucket.cpp:304-331 */
                for (int i = 0; i < 3; i++)
                {
                    const int64_t off = (int64_t) cc[i] * chunkCells;
                    sub.lo[i] = (int32_t) (grid.lo[i] + off);
                    sub.hi[i] = (int32_t) std::min<int64_t>(grid.lo[i] + off + chunkCells, grid.hi[i]);
                }
                RegionView V;
                V.splats = dSplats;
                V.ids = dIds;
                V.invSpacing = 1.0f / full.spacing;
                V.setMicroSize(microSize);
                LevelLayout L;
                L.levels = macroLevels;
                for (int i = 0; i < 3; i++)
                {
                    V.ref[i] = full.reference[i];
                    V.first[i] = grid.lo[i];
                    V.bias[i] = (int32_t) (cc[i] * chunkRatio);
                    V.dims[i] = divUp(sub.cells(i), microSize);
                }
                uint64_t totalNodes = 0;
                for (uint32_t l = 0; l < macroLevels; l++)
                {
                    L.offset[l] = (uint32_t) totalNodes;
                    uint64_t t = 1;
                    for (int i = 0; i < 3; i++)
                    {
                        L.dims[l][i] = divUp(V.dims[i], (uint64_t) 1 << l);
                        t *= L.dims[l][i];
                    }
                    totalNodes += t;
                    REQUIRE(totalNodes <= (1u << 24), MLSGPU_ERR_LENGTH);      /* dense counters; the reference hashes */
                }
                L.offset[macroLevels] = (uint32_t) totalNodes;
                const uint32_t n0 = V.dims[0] * V.dims[1] * V.dims[2];
                if (B.nodeCap < totalNodes)
                {
                    PROPAGATE(ensure(&B.counts, totalNodes));
                    PROPAGATE(ensure(&B.table, totalNodes));
                    B.nodeCap = (uint32_t) totalNodes;
                }
                /* the level's buffers are about to be overwritten: behind the callbacks that still read its lists */
                for (hipEvent_t ev : B.readers)
                    HIP_CHECK(hipStreamWaitEvent(ctx->stream, ev, 0));
                B.readers.clear();
                /* 1. counts */
                HIP_CHECK(hipMemsetAsync(B.counts, 0, totalNodes * 4, ctx->stream));
                if (n > 0)
                {
                    const uint32_t blocks = (uint32_t) std::min<uint64_t>(divUp(n, 256), 4096);
                    /* the coarsest levels that fit the LDS table together (levels are stored finest first) */
                    uint32_t ldsFrom = L.levels;
                    while (ldsFrom > 0 && totalNodes - L.offset[ldsFrom - 1] <= LDS_NODES)
                        ldsFrom--;
                    /* what the passes behind the count need of an element, kept by the count (RegionView::packNote) */
                    uint2 *notes = nullptr;
                    if (n >= notesFrom() && std::max(std::max(V.dims[0], V.dims[1]), V.dims[2]) <= 65536u)
                    {
                        if (B.noteCap < n)
                        {
                            hipFree(B.notes);
                            B.notes = nullptr;
                            B.noteCap = 0;
                            if (hipMalloc((void **) &B.notes, n * sizeof(uint2)) == hipSuccess)
                                B.noteCap = n;
                            else
                                (void) hipGetLastError();       /* without them the passes read the splats again */
                        }
                        notes = B.notes;
                    }
                    /* private counters (bucketCountPrivateKernel): a big level whose finest counters miss the LDS table */
                    const uint32_t words0 = (n0 + 1) / 2, imageWords = words0 + (uint32_t) (totalNodes - n0);
                    const uint64_t numSlices = divUp(n, (uint64_t) PRIV_SPAN);
                    bool privateCounters = (ldsFrom > 0 || privateFrom() == 0) && macroLevels >= 2 && n >= privateFrom()
                        && (uint64_t) imageWords * 4 <= PRIV_LDS_BYTES && numSlices < (1u << 24);
                    if (privateCounters && B.sliceCap < numSlices * imageWords)
                    {
                        hipFree(B.slices);
                        B.slices = nullptr;
                        B.sliceCap = 0;
                        if (hipMalloc((void **) &B.slices, numSlices * imageWords * 4) == hipSuccess)
                            B.sliceCap = numSlices * imageWords;
                        else
                        {
                            (void) hipGetLastError();
                            privateCounters = false;
                        }
                    }
                    if (privateCounters)
                    {
                        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(bucketCountPrivateKernel),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, PRIV_LDS_BYTES));
                        LAUNCH_LDS(ctx, "bucket.count.time", bucketCountPrivateKernel, dim3((uint32_t) numSlices), dim3(PRIV_THREADS),
                                   imageWords * 4, V, L, B.slices, n, notes, words0, imageWords);
                        LAUNCH(ctx, "bucket.count.time", bucketSliceSumKernel, dim3(divUp(imageWords, 256), (uint32_t) divUp(numSlices, (uint64_t) SLICE_GROUP)),
                               dim3(256), (const uint32_t *) B.slices, (uint32_t) numSlices, words0, imageWords, n0, B.counts);
                        LAUNCH(ctx, "bucket.count.time", bucketUpsweepKernel, dim3(1), dim3(1024), L, B.counts);
                    }
                    else
                        LAUNCH(ctx, "bucket.count.time", bucketCountKernel, dim3(blocks), dim3(256), V, L, B.counts, n, ldsFrom, notes);
                    V.notes = notes;
                }
                std::vector<uint32_t> counts(totalNodes);
                HIP_CHECK(hipMemcpyAsync(counts.data(), B.counts, totalNodes * 4, hipMemcpyDeviceToHost, ctx->stream));
                HIP_CHECK(hipStreamSynchronize(ctx->stream));

                /* 2. pickNodes (src/bucket.cpp:248-269, PickNodes :333-352): depth-first, children x fastest */
                struct Region { uint32_t c[3]; uint32_t level; };
                std::vector<Region> regions;
                std::vector<uint32_t> table(n0, 0);
                struct Frame { uint32_t c[3]; uint32_t level; };
                std::vector<Frame> stack;
                stack.push_back(Frame{{0, 0, 0}, macroLevels - 1});
                while (!stack.empty())
                {
                    const Frame f = stack.back();
                    stack.pop_back();
                    const uint32_t count = counts[L.offset[f.level] + (f.c[2] * L.dims[f.level][1] + f.c[1]) * L.dims[f.level][0] + f.c[0]];
                    if (count == 0)
                        continue;
                    if (f.level == 0 || (((uint64_t) microSize << f.level) <= P.maxCells && count <= P.maxSplats))
                    {
                        const uint32_t id = (uint32_t) regions.size();
                        REQUIRE(id < (1u << 27), MLSGPU_ERR_LENGTH);
                        regions.push_back(Region{{f.c[0], f.c[1], f.c[2]}, f.level});
                        for (uint32_t z = f.c[2] << f.level; z < std::min(V.dims[2], (f.c[2] + 1) << f.level); z++)
                            for (uint32_t y = f.c[1] << f.level; y < std::min(V.dims[1], (f.c[1] + 1) << f.level); y++)
                                for (uint32_t x = f.c[0] << f.level; x < std::min(V.dims[0], (f.c[0] + 1) << f.level); x++)
                                    table[(z * V.dims[1] + y) * V.dims[0] + x] = (id << 5) | f.level;
                        continue;
                    }
                    for (int idx = 7; idx >= 0; idx--)     /* pushed in reverse so that child 0 is visited first */
                    {
                        const Frame ch{{f.c[0] * 2 + (idx & 1), f.c[1] * 2 + ((idx >> 1) & 1), f.c[2] * 2 + (uint32_t) (idx >> 2)}, f.level - 1};
                        bool inside = true;
                        for (int j = 0; j < 3; j++)
                            if (((uint64_t) ch.c[j] << ch.level) >= V.dims[j])
                                inside = false;
                        if (inside)
                            stack.push_back(ch);
                    }
                }
                if (regions.empty())
                    continue;
                const uint32_t numRegions = (uint32_t) regions.size();
                /* a region's counter is the number of splats that join it -- the length of its member list: the lists' starts
                 * and the level's pair total are known here, without a word from the device.  The (region, id) pairs are
                 * counted by a 32-bit scan and addressed by 32-bit positions (10^9 uniform splats in 252-cell regions:
                 * 1.06 * 10^9) */
                std::vector<uint32_t> starts(numRegions + 1, 0);
                {
                    uint64_t pairs = 0;
                    for (uint32_t r = 0; r < numRegions; r++)
                    {
                        const Region &g = regions[r];
                        starts[r] = (uint32_t) pairs;
                        pairs += counts[L.offset[g.level] + (g.c[2] * L.dims[g.level][1] + g.c[1]) * L.dims[g.level][0] + g.c[0]];
                        REQUIRE(pairs < 0xFFFFFFFFull, MLSGPU_ERR_LENGTH);
                    }
                    starts[numRegions] = (uint32_t) pairs;
                }
                const uint32_t totalPairs = starts[numRegions];
                HIP_CHECK(hipMemcpyAsync(B.table, table.data(), (size_t) n0 * 4, hipMemcpyHostToDevice, ctx->stream));

                /* 3. member lists */
                const RegionCountIn in{V, B.table};
                if (B.scanCap < scanTiles(n))
                {
                    PROPAGATE(ensure(&B.scanSums, scanTiles(n)));
                    B.scanCap = scanTiles(n);
                }
                uint32_t *tileSums = B.scanSums;
                if (B.pairCap < totalPairs)
                {
                    const uint64_t cap = (uint64_t) totalPairs + totalPairs / 8 + 1024;
                    B.pairCap = 0;
                    PROPAGATE(ensure(&B.keysA, cap));
                    PROPAGATE(ensure(&B.valsA, cap));
                    PROPAGATE(ensure(&B.valsB, cap));
                    PROPAGATE(ensure(&B.hist, sortHistElems(cap)));
                    PROPAGATE(ensure(&B.tileSums, scanTiles(std::max<uint64_t>(sortHistElems(cap), cap))));
                    B.pairCap = cap;
                }
                PROPAGATE((scanPhase1<uint32_t, RegionCountIn>(ctx, "bucket.members.time", in, n, 0u, tileSums, B.total)));
                PROPAGATE((scanPhase2<uint32_t, RegionCountIn, RegionEmitOut>(ctx, "bucket.members.time", in,
                                                                            RegionEmitOut{V, B.table, B.keysA, B.valsA}, n,
                                                                            (const uint32_t *) tileSums)));
                SortResult<uint32_t> sorted{B.keysA, B.valsA};
                /* one digit (region numbers of up to 10 bits) leaves the values in valsB and needs no sorted keys at all;
                 * more digits ping-pong the keys, so both key buffers exist then */
                const uint32_t regionBits = bitsForCount(numRegions);
                const bool oneDigit = regionBits <= SortCaps<uint32_t>::MAX_DIGIT_BITS && getenv("MLSGPU_HIP_SORT_DIGIT_BITS") == nullptr;
                if (!oneDigit && B.keysBCap < B.pairCap)
                {
                    B.keysBCap = 0;
                    PROPAGATE(ensure(&B.keysB, B.pairCap));
                    B.keysBCap = B.pairCap;
                }
                PROPAGATE(radixSort<uint32_t>(ctx, "bucket.members.time", B.keysA, B.valsA, B.keysB, B.valsB, totalPairs,
                                              regionBits, false, B.hist, B.tileSums, &sorted, nullptr, 0, false));
                /* the callbacks may read the lists from any stream */
                HIP_CHECK(hipStreamSynchronize(ctx->stream));

                /* 4. doCallbacks, src/bucket_impl.h:258-294 */
                const uint64_t chunk[3] = {chunkIn[0] + cx, chunkIn[1] + cy, chunkIn[2] + cz};
                const uint32_t *members = sorted.vals;
                for (uint32_t r = 0; r < numRegions; r++)
                {
                    GridBox child;
                    for (int i = 0; i < 3; i++)     /* Node::toCells clipped to the grid, src/bucket.cpp:113-122 */
                    {
                        const uint64_t lower = std::min<uint64_t>(((uint64_t) microSize * regions[r].c[i]) << regions[r].level, sub.cells(i));
                        const uint64_t upper = std::min<uint64_t>((((uint64_t) microSize * regions[r].c[i]) << regions[r].level)
                                                                  + ((uint64_t) microSize << regions[r].level), sub.cells(i));
                        child.lo[i] = (int32_t) (sub.lo[i] + (int64_t) lower);
                        child.hi[i] = (int32_t) (sub.lo[i] + (int64_t) upper);
                    }
                    PROPAGATE(recurse(members + starts[r], starts[r + 1] - starts[r], true, child, 0, 0, depth + 1, chunk));
                }
            }
    return MLSGPU_OK;
}

/*
 * The top level of bucketRecurse (src/bucket_impl.h:439-560) over a splat set that does not fit the device: where the
 * reference streams its blobs twice per level through one host thread (count, then append ranges), the files are streamed
 * through a chunk buffer in HBM --
 *   pass 1   every chunk adds to the SAME microblock-octree counters (bucketCountKernel, unchanged); the host picks the
 *            level's regions from them exactly as for a resident cloud (same counters, same traversal);
 *   pass 2.. the regions are taken in order, as many at a time as fit `budget` splats by their counters; the files are
 *            streamed again, a scan keeps the splats that join a region of the batch and appends them to the batch buffer
 *            IN FILE ORDER, and from there on the batch is a resident cloud: member lists, recursion, callbacks.
 * Region numbering, member order and therefore every bucket are those of the resident path (ids are positions in the
 * batch; mlsgpu_bucket::dSplats says in which array).  Passes over the files: 1 + number of batches.
 */
int Bucketer::recurseStream(mlsgpu_fileset *files, uint64_t n, const GridBox &grid, uint64_t budget, uint64_t chunkSplats,
                            uint32_t readerThreads, uint64_t stats[4])
{
    uint32_t cellDims[3];
    for (int i = 0; i < 3; i++)
        cellDims[i] = grid.cells(i);
    const uint32_t maxCellDim = std::max(std::max(cellDims[0], cellDims[1]), cellDims[2]);
    if (maxCellDim == 1)
    {
        cellSplats = n;
        return setError(MLSGPU_ERR_DENSITY, "Too many splats covering one cell (%llu)", (unsigned long long) n);
    }
    /* the level's geometry: as recurse() for the whole set (not a subset: the top level always splits) */
    uint32_t chunkCells = P.chunkCells;
    uint32_t microSize = P.microCells;
    if (microSize == 0 || microSize > maxCellDim)
        microSize = chooseMicroSize(cellDims, P.maxSplit, n, P.maxSplats, P.maxCells);
    while (true)
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
        chunkCells = (uint32_t) ((chunkCells + grain - 1) / grain * grain);
    }
    else
        chunkCells = roundUp(chunkCells, microSize);
    uint32_t chunks[3];
    for (int i = 0; i < 3; i++)
        chunks[i] = divUp(cellDims[i], chunkCells);
    uint32_t macroLevels = 1;
    while (((uint64_t) microSize << (macroLevels - 1)) < chunkCells)
        macroLevels++;
    REQUIRE(macroLevels <= MAX_LEVELS, MLSGPU_ERR_LENGTH);
    const uint32_t chunkRatio = chunkCells / microSize;

    while (depthList->size() <= 0)
        depthList->emplace_back(new DepthBuffers);
    DepthBuffers &B = *(*depthList)[0];
    if (B.total == nullptr)
        PROPAGATE(ensure(&B.total, 2));

    /* the chunk in flight, the batch, and two words: splats of the batch so far, splats the last chunk added */
    mlsgpu_splat *dChunk = nullptr, *dBatch = nullptr;
    uint32_t *dWords = nullptr, *dScanSums = nullptr;
    uint8_t *dFlags = nullptr;
    struct Free
    {
        mlsgpu_splat *&a, *&b;
        uint32_t *&c, *&d;
        uint8_t *&e;
        ~Free() { hipFree(a); hipFree(b); hipFree(c); hipFree(d); hipFree(e); }
    } release{dChunk, dBatch, dWords, dScanSums, dFlags};
    HIP_CHECK(hipMalloc((void **) &dChunk, chunkSplats * sizeof(mlsgpu_splat)));
    HIP_CHECK(hipMalloc((void **) &dBatch, budget * sizeof(mlsgpu_splat)));
    HIP_CHECK(hipMalloc((void **) &dWords, 2 * sizeof(uint32_t)));
    HIP_CHECK(hipMalloc((void **) &dScanSums, ((size_t) scanTiles(chunkSplats) + 1) * sizeof(uint32_t)));
    HIP_CHECK(hipMalloc((void **) &dFlags, chunkSplats));

    /* The bounding box of every file chunk's splats (with their radii), noted in the first pass: a later pass skips the chunks
     * that cannot reach the regions it is collecting -- what the reference's blob index buys on inputs whose files are
     * spatially coherent (scans); a shuffled cloud skips nothing. */
    const uint64_t numChunks = (n + chunkSplats - 1) / chunkSplats;
    std::vector<float> chunkBox;        /* 6 per chunk: lo xyz, hi xyz; empty until the first pass has run */
    auto forEachFileChunk = [&](const float *want, const std::function<int(uint64_t)> &body) -> int
    {
        const bool note = chunkBox.empty();
        if (note)
            chunkBox.assign(numChunks * 6, 0.0f);
        for (uint64_t c = 0; c < numChunks; c++)
        {
            const uint64_t first = c * chunkSplats;
            const uint64_t cnt = std::min<uint64_t>(chunkSplats, n - first);
            float *box = &chunkBox[c * 6];
            if (!note && want != nullptr)
            {
                bool reach = true;
                for (int a = 0; a < 3; a++)
                    reach = reach && box[a] <= want[3 + a] && box[3 + a] >= want[a];
                if (!reach)
                {
                    stats[3]++;
                    continue;
                }
            }
            PROPAGATE(mlsgpu_hip_fileset_load(files, ctx, first, cnt, dChunk, readerThreads));
            if (note)
            {
                float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                PROPAGATE(foldBbox(ctx, dChunk, cnt, lo, hi));
                for (int a = 0; a < 3; a++)
                {
                    box[a] = lo[a];
                    box[3 + a] = hi[a];
                }
            }
            PROPAGATE(body(cnt));
        }
        stats[0]++;
        return MLSGPU_OK;
    };

    for (uint32_t cx = 0; cx < chunks[0]; cx++)
        for (uint32_t cy = 0; cy < chunks[1]; cy++)
            for (uint32_t cz = 0; cz < chunks[2]; cz++)
            {
                const uint32_t cc[3] = {cx, cy, cz};
                GridBox sub;
                for (int i = 0; i < 3; i++)
                {
                    const int64_t off = (int64_t) cc[i] * chunkCells;
                    sub.lo[i] = (int32_t) (grid.lo[i] + off);
                    sub.hi[i] = (int32_t) std::min<int64_t>(grid.lo[i] + off + chunkCells, grid.hi[i]);
                }
                RegionView V;
                V.splats = dChunk;
                V.ids = nullptr;
                V.invSpacing = 1.0f / full.spacing;
                V.setMicroSize(microSize);
                LevelLayout L;
                L.levels = macroLevels;
                for (int i = 0; i < 3; i++)
                {
                    V.ref[i] = full.reference[i];
                    V.first[i] = grid.lo[i];
                    V.bias[i] = (int32_t) (cc[i] * chunkRatio);
                    V.dims[i] = divUp(sub.cells(i), microSize);
                }
                uint64_t totalNodes = 0;
                for (uint32_t l = 0; l < macroLevels; l++)
                {
                    L.offset[l] = (uint32_t) totalNodes;
                    uint64_t t = 1;
                    for (int i = 0; i < 3; i++)
                    {
                        L.dims[l][i] = divUp(V.dims[i], (uint64_t) 1 << l);
                        t *= L.dims[l][i];
                    }
                    totalNodes += t;
                    REQUIRE(totalNodes <= (1u << 24), MLSGPU_ERR_LENGTH);
                }
                L.offset[macroLevels] = (uint32_t) totalNodes;
                const uint32_t n0 = V.dims[0] * V.dims[1] * V.dims[2];
                if (B.nodeCap < totalNodes)
                {
                    PROPAGATE(ensure(&B.counts, totalNodes));
                    PROPAGATE(ensure(&B.table, totalNodes));
                    B.nodeCap = (uint32_t) totalNodes;
                }
                /* pass 1: the counters of the whole set, chunk after chunk */
                HIP_CHECK(hipMemsetAsync(B.counts, 0, totalNodes * 4, ctx->stream));
                uint32_t ldsFrom = L.levels;
                while (ldsFrom > 0 && totalNodes - L.offset[ldsFrom - 1] <= LDS_NODES)
                    ldsFrom--;
                PROPAGATE(forEachFileChunk(nullptr, [&](uint64_t cnt) -> int
                {
                    const uint32_t blocks = (uint32_t) std::min<uint64_t>(divUp(cnt, 256), 4096);
                    LAUNCH(ctx, "bucket.count.time", bucketCountKernel, dim3(blocks), dim3(256), V, L, B.counts, cnt, ldsFrom, (uint2 *) nullptr);
                    /* the next load overwrites the chunk buffer on another stream-ordered path: finish first */
                    HIP_CHECK(hipStreamSynchronize(ctx->stream));
                    return MLSGPU_OK;
                }));
                std::vector<uint32_t> counts(totalNodes);
                HIP_CHECK(hipMemcpyAsync(counts.data(), B.counts, totalNodes * 4, hipMemcpyDeviceToHost, ctx->stream));
                HIP_CHECK(hipStreamSynchronize(ctx->stream));

                /* pickNodes (src/bucket.cpp:248-269, PickNodes :333-352): depth-first, children x fastest -- as recurse() */
                struct Region { uint32_t c[3]; uint32_t level; uint32_t count; };
                std::vector<Region> regions;
                std::vector<uint32_t> table(n0, 0);
                struct Frame { uint32_t c[3]; uint32_t level; };
                std::vector<Frame> stack;
                stack.push_back(Frame{{0, 0, 0}, macroLevels - 1});
                while (!stack.empty())
                {
                    const Frame f = stack.back();
                    stack.pop_back();
                    const uint32_t count = counts[L.offset[f.level] + (f.c[2] * L.dims[f.level][1] + f.c[1]) * L.dims[f.level][0] + f.c[0]];
                    if (count == 0)
                        continue;
                    if (f.level == 0 || (((uint64_t) microSize << f.level) <= P.maxCells && count <= P.maxSplats))
                    {
                        const uint32_t id = (uint32_t) regions.size();
                        REQUIRE(id < (1u << 27), MLSGPU_ERR_LENGTH);
                        regions.push_back(Region{{f.c[0], f.c[1], f.c[2]}, f.level, count});
                        for (uint32_t z = f.c[2] << f.level; z < std::min(V.dims[2], (f.c[2] + 1) << f.level); z++)
                            for (uint32_t y = f.c[1] << f.level; y < std::min(V.dims[1], (f.c[1] + 1) << f.level); y++)
                                for (uint32_t x = f.c[0] << f.level; x < std::min(V.dims[0], (f.c[0] + 1) << f.level); x++)
                                    table[(z * V.dims[1] + y) * V.dims[0] + x] = (id << 5) | f.level;
                        continue;
                    }
                    for (int idx = 7; idx >= 0; idx--)
                    {
                        const Frame ch{{f.c[0] * 2 + (idx & 1), f.c[1] * 2 + ((idx >> 1) & 1), f.c[2] * 2 + (uint32_t) (idx >> 2)}, f.level - 1};
                        bool inside = true;
                        for (int j = 0; j < 3; j++)
                            if (((uint64_t) ch.c[j] << ch.level) >= V.dims[j])
                                inside = false;
                        if (inside)
                            stack.push_back(ch);
                    }
                }
                if (regions.empty())
                    continue;
                const uint32_t numRegions = (uint32_t) regions.size();
                HIP_CHECK(hipMemcpyAsync(B.table, table.data(), (size_t) n0 * 4, hipMemcpyHostToDevice, ctx->stream));
                HIP_CHECK(hipStreamSynchronize(ctx->stream));          /* `table` is a local */

                const uint64_t chunk[3] = {cx, cy, cz};
                for (uint32_t r0 = 0; r0 < numRegions;)
                {
                    /* as many consecutive regions as fit the batch buffer by their counters (a splat that joins two of them
                     * is counted twice and stored once) */
                    uint32_t r1 = r0;
                    uint64_t sum = 0;
                    while (r1 < numRegions && sum + regions[r1].count <= budget)
                        sum += regions[r1++].count;
                    if (r1 == r0)
                        return setError(MLSGPU_ERR_LENGTH, "bucket stream: a region of %u splats does not fit the device budget of "
                                        "%llu splats", regions[r0].count, (unsigned long long) budget);
                    REQUIRE(sum < 0xFFFFFFFFull, MLSGPU_ERR_LENGTH);
                    stats[1]++;
                    /* pass 2: the batch's splats, in file order */
                    HIP_CHECK(hipMemsetAsync(dWords, 0, 8, ctx->stream));
                    RegionCountIn joins{V, B.table, r0, r1};
                    /* the batch's regions in world coordinates, a cell wider on every side: a splat joins a region by the cells
                     * its box [p - r, p + r] floors into, and a chunk whose box stays clear of all of them has no member */
                    float want[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    for (uint32_t r = r0; r < r1; r++)
                        for (int i = 0; i < 3; i++)
                        {
                            const int64_t lowerCell = (int64_t) sub.lo[i] + (((int64_t) microSize * regions[r].c[i]) << regions[r].level);
                            const int64_t upperCell = lowerCell + ((int64_t) microSize << regions[r].level);
                            want[i] = std::min(want[i], full.reference[i] + full.spacing * (float) (lowerCell - 1));
                            want[3 + i] = std::max(want[3 + i], full.reference[i] + full.spacing * (float) (upperCell + 1));
                        }
                    PROPAGATE(forEachFileChunk(want, [&](uint64_t cnt) -> int
                    {
                        PROPAGATE((exclusiveScan2<uint32_t, BatchJoinIn, BatchFlagIn, BatchCopyOut>(
                            ctx, "bucket.members.time", BatchJoinIn{joins, dFlags}, BatchFlagIn{dFlags},
                            BatchCopyOut{dChunk, dBatch, dWords}, cnt, 0u, dScanSums, dWords + 1)));
                        hipLaunchKernelGGL(addCountKernel, dim3(1), dim3(1), 0, ctx->stream, dWords, (const uint32_t *) (dWords + 1));
                        HIP_CHECK(hipGetLastError());
                        HIP_CHECK(hipStreamSynchronize(ctx->stream));
                        return MLSGPU_OK;
                    }));
                    uint32_t nb = 0;
                    HIP_CHECK(hipMemcpyAsync(&nb, dWords, 4, hipMemcpyDeviceToHost, ctx->stream));
                    HIP_CHECK(hipStreamSynchronize(ctx->stream));
                    REQUIRE(nb <= budget, MLSGPU_ERR_LENGTH);
                    stats[2] += nb;

                    /* from here the batch is a resident cloud: member lists of regions [r0, r1) and the recursion, as recurse() */
                    RegionView VB = V;
                    VB.splats = dBatch;
                    const uint32_t nr = r1 - r0;
                    if (B.regionCap < nr + 1)
                    {
                        PROPAGATE(ensure(&B.starts, nr + 1));
                        B.regionCap = nr + 1;
                    }
                    const RegionCountIn in{VB, B.table, r0, r1};
                    if (B.scanCap < scanTiles(nb))
                    {
                        PROPAGATE(ensure(&B.scanSums, scanTiles(nb)));
                        B.scanCap = scanTiles(nb);
                    }
                    PROPAGATE((scanPhase1<uint32_t, RegionCountIn>(ctx, "bucket.members.time", in, nb, 0u, B.scanSums, B.total)));
                    uint32_t totalPairs = 0;
                    HIP_CHECK(hipMemcpyAsync(&totalPairs, B.total, 4, hipMemcpyDeviceToHost, ctx->stream));
                    HIP_CHECK(hipStreamSynchronize(ctx->stream));
                    if (B.pairCap < totalPairs)
                    {
                        const uint64_t cap = (uint64_t) totalPairs + totalPairs / 8 + 1024;
                        PROPAGATE(ensure(&B.keysA, cap));
                        PROPAGATE(ensure(&B.valsA, cap));
                        PROPAGATE(ensure(&B.valsB, cap));
                        PROPAGATE(ensure(&B.hist, sortHistElems(cap)));
                        PROPAGATE(ensure(&B.tileSums, scanTiles(std::max<uint64_t>(sortHistElems(cap), cap))));
                        B.pairCap = cap;
                    }
                    if (B.keysBCap < B.pairCap)
                    {
                        B.keysBCap = 0;
                        PROPAGATE(ensure(&B.keysB, B.pairCap));
                        B.keysBCap = B.pairCap;
                    }
                    PROPAGATE((scanPhase2<uint32_t, RegionCountIn, RegionEmitOut>(ctx, "bucket.members.time", in,
                                                                                   RegionEmitOut{VB, B.table, B.keysA, B.valsA, r0, r1}, nb,
                                                                                   (const uint32_t *) B.scanSums)));
                    SortResult<uint32_t> sorted{B.keysA, B.valsA};
                    PROPAGATE(radixSort<uint32_t>(ctx, "bucket.members.time", B.keysA, B.valsA, B.keysB, B.valsB, totalPairs,
                                                  bitsForCount(nr), false, B.hist, B.tileSums, &sorted));
                    std::vector<uint32_t> starts(nr + 1, 0);
                    if (totalPairs > 0)
                    {
                        hipLaunchKernelGGL(regionStartsKernel, dim3(divUp(totalPairs, 256)), dim3(256), 0, ctx->stream,
                                           (const uint32_t *) sorted.keys, (uint64_t) totalPairs, B.starts);
                        HIP_CHECK(hipMemcpyAsync(starts.data(), B.starts, (size_t) nr * 4, hipMemcpyDeviceToHost, ctx->stream));
                        HIP_CHECK(hipStreamSynchronize(ctx->stream));
                    }
                    starts[nr] = totalPairs;
                    const mlsgpu_splat *const saved = dSplats;
                    dSplats = dBatch;
                    int rc = MLSGPU_OK;
                    const uint32_t *members = sorted.vals;
                    for (uint32_t r = r0; r < r1 && rc == MLSGPU_OK; r++)
                    {
                        GridBox child;
                        for (int i = 0; i < 3; i++)     /* Node::toCells clipped to the grid, src/bucket.cpp:113-122 */
                        {
                            const uint64_t lower = std::min<uint64_t>(((uint64_t) microSize * regions[r].c[i]) << regions[r].level, sub.cells(i));
                            const uint64_t upper = std::min<uint64_t>((((uint64_t) microSize * regions[r].c[i]) << regions[r].level)
                                                                      + ((uint64_t) microSize << regions[r].level), sub.cells(i));
                            child.lo[i] = (int32_t) (sub.lo[i] + (int64_t) lower);
                            child.hi[i] = (int32_t) (sub.lo[i] + (int64_t) upper);
                        }
                        rc = recurse(members + starts[r - r0], starts[r - r0 + 1] - starts[r - r0], true, child, 0, 0, 1, chunk);
                    }
                    dSplats = saved;
                    PROPAGATE(rc);
                    /* the batch buffer and this level's member lists are refilled next: behind the callbacks' own reads
                     * (mlsgpu_bucket::consumed -- a callback may only have ENQUEUED its gather, on any stream), not just behind
                     * this stream's work */
                    settleReaders();
                    HIP_CHECK(hipStreamSynchronize(ctx->stream));
                    r0 = r1;
                }
            }
    return MLSGPU_OK;
}

} // namespace

MLSGPU_API int mlsgpu_hip_bucket(mlsgpu_ctx *ctx, const mlsgpu_splat *dSplats, uint64_t numSplats, const mlsgpu_grid *region,
                                 const mlsgpu_bucket_params *params, mlsgpu_bucket_fn fn, void *user, uint64_t *cellSplats)
{
    REQUIRE(ctx != nullptr && region != nullptr && params != nullptr && fn != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(numSplats == 0 || dSplats != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(numSplats < 0xFFFFFFFFull, MLSGPU_ERR_LENGTH);           /* u32 ids; the (region, id) pair count is checked per level */
    REQUIRE(region->spacing > 0.0f && params->maxCells >= 1 && params->maxSplats >= 1 && params->maxSplit >= 8, MLSGPU_ERR_INVALID);
    for (int i = 0; i < 3; i++)
        REQUIRE(region->extents[2 * i] < region->extents[2 * i + 1], MLSGPU_ERR_INVALID);   /* at least one cell per axis */
    HIP_CHECK(hipSetDevice(ctx->device));
    std::shared_ptr<void> &cached = ctx->scratchCache["bucket"];
    if (!cached)
        cached = std::shared_ptr<void>(new Bucketer::DepthList, [](void *p) { delete static_cast<Bucketer::DepthList *>(p); });
    Bucketer b;
    b.depthList = static_cast<Bucketer::DepthList *>(cached.get());
    b.ctx = ctx;
    b.dSplats = dSplats;
    b.numSplats = numSplats;
    b.full = *region;
    b.P = *params;
    b.fn = fn;
    b.user = user;
    GridBox g;
    for (int i = 0; i < 3; i++)
    {
        g.lo[i] = region->extents[2 * i];
        g.hi[i] = region->extents[2 * i + 1];
    }
    const uint64_t chunk[3] = {0, 0, 0};
    /* the whole set is not a subset type: the top level always splits (bucket_impl.h:400-418) */
    const int rc = b.recurse(nullptr, numSplats, false, g, params->chunkCells, params->microCells, 0, chunk);
    if (cellSplats != nullptr)
        *cellSplats = b.cellSplats;
    b.settleReaders();
    hipStreamSynchronize(ctx->stream);
    return rc;
}

/* detail::Bbox of device-resident splats folded into lo / hi (min and max are exact in any order) */
static int foldBbox(mlsgpu_ctx *ctx, const mlsgpu_splat *dSplats, uint64_t numSplats, float lo[3], float hi[3])
{
    const uint32_t blocks = (uint32_t) std::max<uint64_t>(1, std::min<uint64_t>(divUp(numSplats, 256), 2048));
    float *dPartial = nullptr;
    HIP_CHECK(hipMalloc((void **) &dPartial, (size_t) blocks * 6 * sizeof(float)));
    std::vector<float> partial((size_t) blocks * 6);
    hipLaunchKernelGGL(bboxKernel, dim3(blocks), dim3(256), 0, ctx->stream, dSplats, numSplats, dPartial);
    hipError_t e = hipMemcpyAsync(partial.data(), dPartial, partial.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    hipFree(dPartial);
    if (e != hipSuccess)
        return setError(MLSGPU_ERR_HIP, "bounding grid: %s", hipGetErrorString(e));
    for (uint32_t b = 0; b < blocks; b++)
        for (int a = 0; a < 3; a++)
        {
            lo[a] = std::min(lo[a], partial[(size_t) b * 6 + a]);
            hi[a] = std::max(hi[a], partial[(size_t) b * 6 + 3 + a]);
        }
    return MLSGPU_OK;
}

static int gridFromBbox(const float lo[3], const float hi[3], float spacing, uint32_t bucketSize, mlsgpu_grid *out);

/* FastBlobSet::makeBoundingGrid, src/splat_set_impl.h:770-811 */
MLSGPU_API int mlsgpu_hip_bounding_grid(mlsgpu_ctx *ctx, const mlsgpu_splat *dSplats, uint64_t numSplats, float spacing,
                                        uint32_t bucketSize, mlsgpu_grid *out)
{
    REQUIRE(ctx != nullptr && out != nullptr && (numSplats == 0 || dSplats != nullptr), MLSGPU_ERR_INVALID);
    REQUIRE(spacing > 0.0f && bucketSize >= 1, MLSGPU_ERR_INVALID);
    HIP_CHECK(hipSetDevice(ctx->device));
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    PROPAGATE(foldBbox(ctx, dSplats, numSplats, lo, hi));
    return gridFromBbox(lo, hi, spacing, bucketSize, out);
}

/* ... for a FileSet that need not fit the device: one pass over the files through a chunk buffer (the bounding box is what
 * FastBlobSet::computeBlobs accumulates while it makes its blobs, src/splat_set_impl.h:814-880) */
MLSGPU_API int mlsgpu_hip_fileset_bounding_grid(mlsgpu_fileset *files, mlsgpu_ctx *ctx, float spacing, uint32_t bucketSize,
                                                uint64_t chunkSplats, uint32_t readerThreads, mlsgpu_grid *out)
{
    REQUIRE(files != nullptr && ctx != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(spacing > 0.0f && bucketSize >= 1 && chunkSplats >= 1, MLSGPU_ERR_INVALID);
    HIP_CHECK(hipSetDevice(ctx->device));
    const uint64_t n = mlsgpu_hip_fileset_num_splats(files);
    mlsgpu_splat *dChunk = nullptr;
    HIP_CHECK(hipMalloc((void **) &dChunk, std::min<uint64_t>(chunkSplats, std::max<uint64_t>(n, 1)) * sizeof(mlsgpu_splat)));
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    int rc = MLSGPU_OK;
    for (uint64_t first = 0; first < n && rc == MLSGPU_OK; first += chunkSplats)
    {
        const uint64_t cnt = std::min<uint64_t>(chunkSplats, n - first);
        rc = mlsgpu_hip_fileset_load(files, ctx, first, cnt, dChunk, readerThreads);
        if (rc == MLSGPU_OK)
            rc = foldBbox(ctx, dChunk, cnt, lo, hi);
    }
    hipFree(dChunk);
    PROPAGATE(rc);
    return gridFromBbox(lo, hi, spacing, bucketSize, out);
}

static int gridFromBbox(const float lo[3], const float hi[3], float spacing, uint32_t bucketSize, mlsgpu_grid *out)
{
    if (lo[0] > hi[0])
        return setError(MLSGPU_ERR_INVALID, "Must be at least one splat");        /* std::runtime_error, :773-774 */
    for (int a = 0; a < 3; a++)
    {
        out->reference[a] = 0.0f;
        int64_t l = (int64_t) std::floor(lo[a] / spacing);
        const int64_t h = (int64_t) std::ceil(hi[a] / spacing);
        const int64_t b = (int64_t) bucketSize;
        l = (l >= 0 ? l / b : -((-l + b - 1) / b)) * b;       /* the lower extent is a multiple of the bucket size */
        out->extents[2 * a] = (int32_t) l;
        out->extents[2 * a + 1] = (int32_t) h;
    }
    out->spacing = spacing;
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_bucket_load(mlsgpu_ctx *ctx, const mlsgpu_splat *dSplats, const uint32_t *dIds, uint64_t numSplats,
                                      const mlsgpu_grid *fullGrid, mlsgpu_splat *dOut)
{
    REQUIRE(ctx != nullptr && fullGrid != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(numSplats == 0 || (dSplats != nullptr && dOut != nullptr), MLSGPU_ERR_INVALID);
    REQUIRE(fullGrid->spacing > 0.0f, MLSGPU_ERR_INVALID);
    if (numSplats == 0)
        return MLSGPU_OK;
    HIP_CHECK(hipSetDevice(ctx->device));
    LAUNCH(ctx, "bucket.load.time", bucketLoadKernel, dim3(divUp(numSplats, 256)), dim3(256), dSplats, dIds, numSplats,
           fullGrid->reference[0], fullGrid->reference[1], fullGrid->reference[2], 1.0f / fullGrid->spacing,
           (float) fullGrid->extents[0], (float) fullGrid->extents[2], (float) fullGrid->extents[4], dOut);
    return MLSGPU_OK;
}

/* Bucket::bucket over a FileSet that need not fit the device (the role of FastBlobSet + the host bucketing of
 * src/bucket_impl.h:439-560 for data beyond HBM; see Bucketer::recurseStream). */
MLSGPU_API int mlsgpu_hip_bucket_stream(mlsgpu_ctx *ctx, mlsgpu_fileset *files, const mlsgpu_grid *region,
                                        const mlsgpu_bucket_params *params, uint64_t budgetSplats, uint64_t chunkSplats,
                                        uint32_t readerThreads, mlsgpu_bucket_fn fn, void *user, uint64_t *cellSplats,
                                        uint64_t stats[4])
{
    REQUIRE(ctx != nullptr && files != nullptr && region != nullptr && params != nullptr && fn != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(region->spacing > 0.0f && params->maxCells >= 1 && params->maxSplats >= 1 && params->maxSplit >= 8, MLSGPU_ERR_INVALID);
    REQUIRE(budgetSplats >= 1 && budgetSplats < 0xFFFFFFFFull && chunkSplats >= 1 && chunkSplats < 0xFFFFFFFFull, MLSGPU_ERR_INVALID);
    for (int i = 0; i < 3; i++)
        REQUIRE(region->extents[2 * i] < region->extents[2 * i + 1], MLSGPU_ERR_INVALID);
    HIP_CHECK(hipSetDevice(ctx->device));
    std::shared_ptr<void> &cached = ctx->scratchCache["bucket"];
    if (!cached)
        cached = std::shared_ptr<void>(new Bucketer::DepthList, [](void *p) { delete static_cast<Bucketer::DepthList *>(p); });
    Bucketer b;
    b.depthList = static_cast<Bucketer::DepthList *>(cached.get());
    b.ctx = ctx;
    b.dSplats = nullptr;
    b.numSplats = mlsgpu_hip_fileset_num_splats(files);
    b.full = *region;
    b.P = *params;
    b.fn = fn;
    b.user = user;
    GridBox g;
    for (int i = 0; i < 3; i++)
    {
        g.lo[i] = region->extents[2 * i];
        g.hi[i] = region->extents[2 * i + 1];
    }
    uint64_t local[4] = {0, 0, 0, 0};
    const int rc = b.numSplats == 0 ? MLSGPU_OK
        : b.recurseStream(files, b.numSplats, g, budgetSplats, chunkSplats, readerThreads, stats != nullptr ? stats : local);
    if (cellSplats != nullptr)
        *cellSplats = b.cellSplats;
    b.settleReaders();
    hipStreamSynchronize(ctx->stream);
    return rc;
}
