/*
 * mlsgpu CPU ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * A from-scratch CPU restatement of the per-bucket device pipeline of
 * bmerry/mlsgpu (octree build -> MLS corner evaluation -> marching tetrahedra
 * with on-device welding -> scale/bias).  Nothing in the product path
 * (mlsgpu_amd/, include/) may include, link, import or call this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * only as the checker / the reported CPU baseline.
 *
 * PARITY PINNING.  The reference cannot be built here (needs boost, cppunit,
 * clogs 1.1, an OpenCL device with image support; none are present), so there
 * is no oracle/_ref.  The oracle is pinned instead by every known-answer test
 * the reference holds for this path (tests/test_oracle_*.py restate them with
 * the reference's numbers): test/test_mls.cpp:255-514,
 * test/test_splat_tree.cpp:46-244, test/test_splat_tree_cl.cpp:171-204,
 * test/test_marching.cpp:270-632, test/test_mesh_filter.cpp:284-361.
 *
 * Third-party arithmetic restated here: clogs 1.1 (not vendored by the
 * reference, doc/mlsgpu-user-manual.xml:81): exclusive prefix sum with
 * optional seed, and a stable LSD radix sort on the low `maxBits` key bits.
 * Both are integer and exact; restated as a sequential scan and a stable sort
 * on the masked key.
 *
 * FLOATING-POINT CONTRACT (shared with the HIP path):
 *   - build with -ffp-contract=off; every fused multiply-add is an explicit
 *     fmaf(), placed where the reference has an explicit fma() (dot3,
 *     interp, scaleBias) and on the four accumulations of sphereFitAdd /
 *     planeFitAdd that OpenCL's default FP_CONTRACT ON contracts;
 *   - division and sqrtf are IEEE correctly rounded;
 *   - half_rsqrt(b2) of kernels/mls.cl:406 is replaced by 1.0f/sqrtf(b2)
 *     (it only scales |f|, never its sign);
 *   - normalize(v) = v * (1.0f/sqrtf(dot3(v,v))).
 *
 * All citations are file:line under /root/reference.
 */
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <utility>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API extern "C" __attribute__((visibility("default")))

namespace
{

/* src/splat.h:40-46 == kernels/octree.cl:32-36 */
struct Splat
{
    float position[3];
    float radius;      /* after tree build: 1/radius^2 (kernels/octree.cl:193) */
    float normal[3];
    float quality;
};
static_assert(sizeof(Splat) == 32, "Splat must be 32 bytes");

typedef int32_t command_type;

struct int3_ { int x, y, z; };

/* ------------------------------------------------------------------ */
/* helpers shared by octree.cl and mls.cl                               */
/* ------------------------------------------------------------------ */

/* kernels/octree.cl:121-136, kernels/mls.cl:159-174 */
static uint32_t makeCode(int x, int y, int z)
{
    uint32_t ans = 0;
    uint32_t scale = 1;
    y <<= 1;
    z <<= 2;
    while (x != 0 || y != 0 || z != 0)
    {
        uint32_t bits = (x & 1) | (y & 2) | (z & 4);
        ans += bits * scale;
        scale <<= 3;
        x >>= 1; y >>= 1; z >>= 1;
    }
    return ans;
}

/* kernels/mls.cl:183-196 */
static void decode(uint32_t code, int out[3])
{
    int x = 0, y = 0, z = 0;
    uint32_t scale = 1;
    while (code >= scale)
    {
        x += code & scale;
        y += (code >> 1) & scale;
        z += (code >> 2) & scale;
        code >>= 2;
        scale <<= 1;
    }
    out[0] = x; out[1] = y; out[2] = z;
}

/* kernels/mls.cl:105-108 */
static inline float dot3(const float a[3], const float b[3])
{
    return fmaf(a[0], b[0], fmaf(a[1], b[1], a[2] * b[2]));
}
static inline float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return fmaf(ax, bx, fmaf(ay, by, az * bz));
}

/* ------------------------------------------------------------------ */
/* octree (kernels/octree.cl, src/splat_tree_cl.cpp)                    */
/* ------------------------------------------------------------------ */

static inline int clz32(uint32_t v) { return v == 0 ? 32 : __builtin_clz(v); }

/* kernels/octree.cl:50-55 */
static int levelShift(const int lo[3], const int hi[3])
{
    int big = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
    return big > 1 ? 32 - clz32((uint32_t) (big - 1)) : 0;
}

/* kernels/octree.cl:60-66. OpenCL dot(): plain sum of products, no fma. */
static float pointBoxDist2(const float pos[3], const float lo[3], const float hi[3])
{
    float d[3];
    for (int i = 0; i < 3; i++)
    {
        float nearest = std::max(lo[i], std::min(hi[i], pos[i]));
        d[i] = nearest - pos[i];
    }
    return d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
}

/* convert_int_rtn (kernels/octree.cl:85-86): floor, saturating */
static inline int floorToInt(float v)
{
    float f = floorf(v);
    if (!(f > -2147483648.0f)) return INT32_MIN;
    if (!(f < 2147483648.0f)) return INT32_MAX;
    return (int) f;
}

/* Stable LSD radix sort of (key,value) on the low `bits` key bits.
 * Semantics of clogs::Radixsort::enqueue(..., maxBits)
 * (call sites src/splat_tree_cl.cpp:308, src/marching.cpp:572). */
template<typename K, typename V>
static void radixSortPairs(std::vector<K> &keys, std::vector<V> &values, size_t n, unsigned int bits)
{
    const unsigned int RADIX_BITS = 11;
    const size_t RADIX = size_t(1) << RADIX_BITS;
    std::vector<K> tk(n);
    std::vector<V> tv(n);
    K *ka = keys.data(), *kb = tk.data();
    V *va = values.data(), *vb = tv.data();
    std::vector<size_t> hist(RADIX);
    for (unsigned int shift = 0; shift < bits; shift += RADIX_BITS)
    {
        unsigned int w = std::min(RADIX_BITS, bits - shift);
        const K mask = (K(1) << w) - 1;
        std::fill(hist.begin(), hist.end(), 0);
        for (size_t i = 0; i < n; i++)
            hist[(ka[i] >> shift) & mask]++;
        size_t sum = 0;
        for (size_t d = 0; d < RADIX; d++)
        {
            size_t c = hist[d];
            hist[d] = sum;
            sum += c;
        }
        for (size_t i = 0; i < n; i++)
        {
            size_t p = hist[(ka[i] >> shift) & mask]++;
            kb[p] = ka[i];
            vb[p] = va[i];
        }
        std::swap(ka, kb);
        std::swap(va, vb);
    }
    if (ka != keys.data())
    {
        std::copy(ka, ka + n, keys.data());
        std::copy(va, va + n, values.data());
    }
}

struct TreeResult
{
    std::vector<command_type> commands;
    std::vector<command_type> start;
    size_t numStart;
    size_t numCommands;  /* one past the highest command slot written */
    unsigned int numLevels;
};

/* src/splat_tree_cl.cpp:269-335 */
static int treeBuild(Splat *splats, size_t firstSplat, size_t numSplats,
                     const uint32_t size[3], const int32_t offset[3],
                     unsigned int subsamplingShift, unsigned int maxLevels,
                     TreeResult &out)
{
    if (maxLevels < 1 || maxLevels > 10) return 1;
    const uint32_t maxSize = uint32_t(1) << (maxLevels + subsamplingShift - 1);
    if (size[0] > maxSize || size[1] > maxSize || size[2] > maxSize) return 1;
    const unsigned int maxShift = maxLevels + subsamplingShift - 1;
    const unsigned int minShift = std::min(subsamplingShift, maxShift);

    std::vector<uint32_t> levelOffsets(maxShift + 1, 0);
    size_t pos = 0;
    for (unsigned int i = minShift; i <= maxShift; i++)
    {
        levelOffsets[i] = (uint32_t) pos;
        pos += size_t(1) << (3 * (maxShift - i));
    }
    const size_t numStart = pos;
    const size_t numEntries = numSplats * 8;

    std::vector<uint32_t> keys(numEntries);
    std::vector<uint32_t> values(numEntries);

    /* writeEntries, kernels/octree.cl:159-214 (+prepare :79-90, goodEntry :100-110) */
    for (size_t i = 0; i < numSplats; i++)
    {
        size_t gid = i + firstSplat;
        Splat &s = splats[gid];
        const float r = s.radius;
        int lo[3], hi[3], ilo[3];
        for (int a = 0; a < 3; a++)
        {
            lo[a] = floorToInt(s.position[a] - r);
            hi[a] = floorToInt(s.position[a] + r);
        }
        int shift = levelShift(lo, hi);
        shift = std::min(std::max(shift, (int) minShift), (int) maxShift);
        for (int a = 0; a < 3; a++)
            ilo[a] = std::max(lo[a] - offset[a], 0) >> shift;

        float radius2 = r * r;
        s.radius = 1.0f / radius2;
        radius2 *= 1.00001f;
        const uint32_t levelOffset = levelOffsets[shift];
        const int bound = 1 << (maxShift - shift);
        size_t p = i * 8;
        for (int oz = 0; oz < 2; oz++)
            for (int oy = 0; oy < 2; oy++)
                for (int ox = 0; ox < 2; ox++)
                {
                    int addr[3] = {ilo[0] + ox, ilo[1] + oy, ilo[2] + oz};
                    uint32_t key = makeCode(addr[0], addr[1], addr[2]) + levelOffset;
                    float vblo[3], vbhi[3];
                    for (int a = 0; a < 3; a++)
                    {
                        /* int arithmetic wraps like OpenCL's */
                        int blo = (int) ((uint32_t) addr[a] << shift) + offset[a];
                        int bhi = (int) ((uint32_t) (addr[a] + 1) << shift) + offset[a];
                        vblo[a] = (float) blo;
                        vbhi[a] = (float) bhi;
                    }
                    bool isect = pointBoxDist2(s.position, vblo, vbhi) < radius2;
                    isect = isect && addr[0] < bound && addr[1] < bound && addr[2] < bound;
                    values[p] = (uint32_t) gid;
                    keys[p] = isect ? key : UINT32_MAX;
                    p++;
                }
    }

    /* sort on 3*(maxShift-minShift)+1 bits, src/splat_tree_cl.cpp:308 */
    radixSortPairs(keys, values, numEntries, 3 * (maxShift - minShift) + 1);

    /* countCommands (kernels/octree.cl:230-239) + scan seeded with 1
     * (src/splat_tree_cl.cpp:310-313).  indicator[numEntries-1] is never
     * written by the reference; the exclusive scan does not read it. */
    std::vector<uint32_t> commandMap(numEntries);
    {
        uint32_t sum = 1;
        for (size_t i = 0; i < numEntries; i++)
        {
            commandMap[i] = sum;
            if (i + 1 < numEntries)
                sum += (keys[i] != keys[i + 1]) ? 3 : 1;
        }
    }

    const uint64_t maxStart = (uint64_t(1) << (3 * maxLevels)) / 7;
    out.start.assign(std::max<uint64_t>(maxStart, numStart), 0);
    std::vector<command_type> jumpPos(numStart, -1);          /* fill, octree.cl:346 */
    const size_t maxRanges = std::min<uint64_t>(maxStart, 8 * (uint64_t) std::max<size_t>(numSplats, 1));
    out.commands.assign(numEntries + 2 * maxRanges + 2, 0);
    size_t highest = 0;

    /* writeSplatIds, kernels/octree.cl:256-279 */
    for (size_t p = 0; p < numEntries; p++)
    {
        uint32_t curKey = keys[p];
        if (curKey != UINT32_MAX)
        {
            uint32_t cpos = commandMap[p];
            out.commands[cpos] = (command_type) values[p];
            uint32_t prevKey = p > 0 ? keys[p - 1] : UINT32_MAX;
            uint32_t nextKey = (p < numEntries - 1) ? keys[p + 1] : UINT32_MAX;
            if (prevKey != curKey)
                out.start[curKey] = (command_type) (cpos - 1);
            if (curKey != nextKey)
            {
                jumpPos[curKey] = (command_type) (cpos + 1);
                highest = std::max<size_t>(highest, cpos + 2);
            }
        }
    }

    /* writeStartTop / writeStart, kernels/octree.cl:292-341, loop src/splat_tree_cl.cpp:320-331 */
    for (int i = (int) maxShift; i >= (int) minShift; i--)
    {
        size_t levelSize = size_t(1) << (3 * (maxShift - i));
        bool havePrev = i != (int) maxShift;
        uint32_t curOffset = levelOffsets[i];
        uint32_t prevOffset = havePrev ? levelOffsets[i + 1] : 0;
        for (size_t code = 0; code < levelSize; code++)
        {
            size_t p = code + curOffset;
            command_type jp = jumpPos[p];
            command_type prev = havePrev ? out.start[prevOffset + (code >> 3)] : -1;
            if (jp >= 0)
            {
                out.commands[jp] = prev;
                out.commands[out.start[p]] = jp;
            }
            else
                out.start[p] = prev;
        }
    }
    out.numStart = numStart;
    out.numCommands = highest;
    out.numLevels = maxShift - minShift + 1;
    return 0;
}

/* ------------------------------------------------------------------ */
/* MLS (kernels/mls.cl)                                                 */
/* ------------------------------------------------------------------ */

#define RADIUS_CUTOFF 0.99f
#define HITS_CUTOFF 4

struct SphereFit
{
    float sumWpp, sumWpn;
    float sumWp[3], sumWn[3];
    float sumW;
    uint32_t hits;
};

struct Sphere
{
    float a;
    float b[3];
    float c;
    float qDen;
    float b2;
};

static inline void sphereFitInit(SphereFit &sf)
{
    std::memset(&sf, 0, sizeof(sf));
}

/* kernels/mls.cl:129-139 with the contraction choice of the header. The plane
 * variant (:141-148) accumulates the same sums minus sumWpn. */
static inline void sphereFitAdd(SphereFit &sf, float w, const float p[3], float pp, const float n[3])
{
    float wn[3] = {w * n[0], w * n[1], w * n[2]};
    sf.sumW = sf.sumW + w;
    for (int i = 0; i < 3; i++)
        sf.sumWp[i] = fmaf(w, p[i], sf.sumWp[i]);
    for (int i = 0; i < 3; i++)
        sf.sumWn[i] = fmaf(w, n[i], sf.sumWn[i]);
    sf.sumWpp = fmaf(w, pp, sf.sumWpp);
    sf.sumWpn = sf.sumWpn + dot3(wn, p);
    sf.hits++;
}

/* kernels/mls.cl:210-229 */
static inline void fitSphere(const SphereFit &sf, Sphere &out)
{
    float invSumW = 1.0f / sf.sumW;
    float m[3] = {sf.sumWp[0] * invSumW, sf.sumWp[1] * invSumW, sf.sumWp[2] * invSumW};
    float qNum = sf.sumWpn - dot3(m, sf.sumWn);
    float qDen = sf.sumWpp - dot3(m, sf.sumWp);
    float q = qNum / qDen;
    if (fabsf(qDen) < (4 * std::numeric_limits<float>::epsilon()) * (float) sf.hits * fabsf(sf.sumWpp)
        || !std::isfinite(q))
    {
        q = 0.0f;
    }
    float a = 0.5f * q;
    float b[3];
    for (int i = 0; i < 3; i++)
        b[i] = (sf.sumWn[i] - q * sf.sumWp[i]) * invSumW;
    out.a = a;
    for (int i = 0; i < 3; i++) out.b[i] = b[i];
    out.c = (-a * sf.sumWpp - dot3(b, sf.sumWp)) * invSumW;
    out.qDen = qDen;
    out.b2 = dot3(b, b);
}

/* half_rsqrt(b2), kernels/mls.cl:406.  OpenCL leaves its error implementation-defined (at most 8192 ulp); oracle and HIP path
 * use the exactly rounded 1/sqrt.  To MEASURE what that substitution can do to a mesh, orc_set_rsqrt_bits(n) makes this
 * function keep only n mantissa bits of the result (n = 11: what a half-precision reciprocal square root delivers;
 * 0 = exact, the default) -- tests/test_oracle_mls.py compares the two meshes. */
static int g_rsqrtBits = 0;
static inline float rsqrtModel(float x)
{
    float r = 1.0f / sqrtf(x);
    if (g_rsqrtBits > 0 && g_rsqrtBits < 23 && std::isfinite(r))
    {
        uint32_t u;
        std::memcpy(&u, &r, 4);
        u &= ~((1u << (23 - g_rsqrtBits)) - 1u);
        std::memcpy(&r, &u, 4);
    }
    return r;
}

/* kernels/mls.cl:237-248 */
static inline float solveQuadratic(float a, float b, float c)
{
    float bdet = b + sqrtf(b * b - 4.0f * a * c);
    float x = -2.0f * c / bdet;
    if (!std::isfinite(x))
        x = bdet / (-2.0f * a);
    return std::isfinite(x) ? x : std::numeric_limits<float>::quiet_NaN();
}

/* kernels/mls.cl:254-258 (test kernel only, not used by processCorners) */
static inline float projectDistOriginSphere(const Sphere &s)
{
    float len = sqrtf(s.b[0] * s.b[0] + s.b[1] * s.b[1] + s.b[2] * s.b[2]);
    return -solveQuadratic(s.a, len, s.c);
}

/* Final evaluation for one corner. kernels/mls.cl:394-424 */
static inline float finishCorner(const SphereFit &fit, int shape, float boundaryFactor)
{
    float f = std::numeric_limits<float>::quiet_NaN();
    if (fit.hits >= HITS_CUTOFF)
    {
        if (shape == 0)
        {
            Sphere sphere;
            fitSphere(fit, sphere);
            /* projectOriginSphere, kernels/mls.cl:263-267 */
            float l = solveQuadratic(sphere.a * sphere.b2, sphere.b2, sphere.c);
            float a[3] = {l * sphere.b[0], l * sphere.b[1], l * sphere.b[2]};
            float aa = dot3(a, a);
            if (aa < 3.0f)
            {
                float rhs = (fit.sumWpp - 2 * dot3(fit.sumWp, a) + fit.sumW * aa);
                if (sphere.qDen > boundaryFactor * rhs)
                    f = -dot3(sphere.b, a) * rsqrtModel(sphere.b2);
            }
        }
        else
        {
            /* fitPlane kernels/mls.cl:198-203, projectOriginPlane :277-280 */
            float mean[3], normal[3];
            for (int i = 0; i < 3; i++) mean[i] = fit.sumWp[i] / fit.sumW;
            float inv = 1.0f / sqrtf(dot3(fit.sumWn, fit.sumWn));
            for (int i = 0; i < 3; i++) normal[i] = fit.sumWn[i] * inv;
            float dist = -dot3(normal, mean);
            float a[3] = {normal[0] * -dist, normal[1] * -dist, normal[2] * -dist};
            float aa = dot3(a, a);
            if (aa < 3.0f)
            {
                float qDen = fit.sumWpp - dot3(mean, fit.sumWp);
                float rhs = (fit.sumWpp - 2 * dot3(fit.sumWp, a) + fit.sumW * aa);
                if (qDen > boundaryFactor * rhs)
                    f = dist;
            }
        }
    }
    return f;
}

struct Swathe
{
    uint32_t width, height;
    uint32_t zStride;
    int32_t zBias;
    uint32_t zFirst, zLast;
};

struct MlsStats
{
    uint64_t listed;  /* sum over blocks of list length (Sigma L) */
    uint64_t hits;    /* (corner, splat) pairs with d < 0.99 (H) */
};

/* processCorners, kernels/mls.cl:299-433, launched as src/mls.cpp:101-135.
 * `field` is the linear stand-in for the 2-D image: pixel (x, row) at
 * field[row * pitch + x]. */
static void processCorners(float *field, size_t pitch,
                           const Splat *splats, const command_type *commands, const command_type *start,
                           uint32_t startShift, const int32_t offset[3],
                           const Swathe &sw, float boundaryFactor, int shape, MlsStats *stats)
{
    const uint32_t W = 8;
    const uint32_t bx = (sw.width + W - 1) / W;
    const uint32_t by = (sw.height + W - 1) / W;
    const uint32_t bz = (sw.zLast - sw.zFirst + 1 + W - 1) / W;
    int lut[512][3];
    for (uint32_t lid = 0; lid < 512; lid++)
        decode(lid, lut[lid]);
    uint64_t totListed = 0, totHits = 0;

    const int64_t nblocks = (int64_t) bx * by * bz;
#pragma omp parallel for schedule(dynamic, 8) reduction(+:totListed, totHits)
    for (int64_t blk = 0; blk < nblocks; blk++)
    {
        const uint32_t gx = blk % bx, gy = (blk / bx) % by, gz = blk / ((int64_t) bx * by);
        int wid[3] = {(int) (gx * W), (int) (gy * W), (int) (gz * W + sw.zFirst)};
        uint32_t code = makeCode(wid[0], wid[1], wid[2]) >> startShift;
        command_type pos = start[code];

        float fx[512], fy[512], fz[512];
        SphereFit fit[512];
        float out[512];
        for (int i = 0; i < 512; i++)
            out[i] = std::numeric_limits<float>::quiet_NaN();

        if (pos >= 0)
        {
            for (int i = 0; i < 512; i++)
            {
                fx[i] = (float) (wid[0] + lut[i][0] + offset[0]);
                fy[i] = (float) (wid[1] + lut[i][1] + offset[1]);
                fz[i] = (float) (wid[2] + lut[i][2] + offset[2]);
                sphereFitInit(fit[i]);
            }
            command_type end = commands[pos++];
            /* Walk exactly like the kernel (kernels/mls.cl:340-392): ids are
             * staged MAX_BUCKET = 256 at a time, the jump is followed as soon
             * as the staged batch reaches the end of the range, and a negative
             * id ends the batch it was staged in. */
            while (pos < end)
            {
                const command_type chunkFirst = pos;
                const command_type chunkLast = std::min<int64_t>((int64_t) pos + 256, end);
                pos += 256;
                if (pos >= end)
                {
                    pos = commands[end];
                    end = (pos >= 0) ? commands[pos++] : INT32_MIN;
                }
                for (command_type lpos = chunkFirst; lpos < chunkLast; lpos++)
                {
                    command_type id = commands[lpos];
                    if (id < 0)
                        break;
                    totListed++;
                    const Splat &s = splats[id];
                    const float px = s.position[0], py = s.position[1], pz = s.position[2];
                    const float invr2 = s.radius;
                    for (int i = 0; i < 512; i++)
                    {
                        float p[3] = {px - fx[i], py - fy[i], pz - fz[i]};
                        float pp = dot3(p, p);
                        float d = pp * invr2;
                        if (d < RADIUS_CUTOFF)
                        {
                            float w = 1.0f - d;
                            w *= w;
                            w *= w;
                            w *= s.quality;
                            sphereFitAdd(fit[i], w, p, pp, s.normal);
                            totHits++;
                        }
                    }
                }
            }
            for (int i = 0; i < 512; i++)
                out[i] = finishCorner(fit[i], shape, boundaryFactor);
        }
        for (int i = 0; i < 512; i++)
        {
            int x = wid[0] + lut[i][0];
            int y = wid[1] + lut[i][1];
            int z = wid[2] + lut[i][2];
            int64_t row = (int64_t) y + (int64_t) z * sw.zStride + sw.zBias;
            field[row * (int64_t) pitch + x] = out[i];
        }
    }
    if (stats)
    {
        stats->listed += totListed;
        stats->hits += totHits;
    }
}

/* ------------------------------------------------------------------ */
/* Marching tetrahedra (src/marching.cpp, kernels/marching.cl)          */
/* ------------------------------------------------------------------ */

enum
{
    NUM_EDGES = 19,
    NUM_TETRAHEDRA = 6,
    NUM_CUBES = 256,
    MAX_CELL_VERTICES = 13,
    MAX_CELL_INDICES = 36,
    MAX_CELL_BYTES = 872,        /* src/marching.h:96-100 */
    KEY_AXIS_BITS = 21,
    MAX_DIMENSION = 8192
};
static const uint64_t KEY_EXTERNAL_FLAG = uint64_t(1) << 63;

/* src/marching.cpp:50-81 */
static const unsigned char edgeIndices[NUM_EDGES][2] =
{
    {0, 1}, {0, 2}, {0, 3}, {1, 3}, {2, 3}, {0, 4}, {0, 5}, {1, 5}, {4, 5}, {0, 6},
    {2, 6}, {4, 6}, {0, 7}, {1, 7}, {2, 7}, {3, 7}, {4, 7}, {5, 7}, {6, 7}
};
static const unsigned char tetrahedronIndices[NUM_TETRAHEDRA][4] =
{
    {0, 7, 1, 3}, {0, 7, 3, 2}, {0, 7, 2, 6}, {0, 7, 6, 4}, {0, 7, 4, 5}, {0, 7, 5, 1}
};

struct Tables
{
    uint8_t count[NUM_CUBES][2];
    uint16_t start[NUM_CUBES + 1][2];
    std::vector<uint8_t> data;
    std::vector<uint32_t> key;   /* 3 per entry */
};

static unsigned int findEdgeByVertexIds(unsigned int v0, unsigned int v1)
{
    if (v0 > v1) std::swap(v0, v1);
    for (unsigned int i = 0; i < NUM_EDGES; i++)
        if (edgeIndices[i][0] == v0 && edgeIndices[i][1] == v1)
            return i;
    assert(false);
    return ~0u;
}

typedef std::pair<unsigned char, bool> tvtx;

static unsigned int permutationParity(const tvtx *first, const tvtx *last)
{
    unsigned int parity = 0;
    for (const tvtx *i = first; i != last; ++i)
        for (const tvtx *j = i + 1; j != last; ++j)
            if (*i > *j)
                parity ^= 1;
    return parity;
}

/* src/marching.cpp:109-252 */
static void makeTables(Tables &t)
{
    std::vector<uint8_t> vertexTable, indexTable;
    t.key.clear();
    for (unsigned int i = 0; i < NUM_CUBES; i++)
    {
        t.start[i][0] = (uint16_t) vertexTable.size();
        t.start[i][1] = (uint16_t) indexTable.size();
        std::vector<uint8_t> triangles;
        for (unsigned int j = 0; j < NUM_TETRAHEDRA; j++)
        {
            tvtx tvtxs[4];
            unsigned int outside = 0;
            for (unsigned int k = 0; k < 4; k++)
            {
                unsigned int v = tetrahedronIndices[j][k];
                bool o = (i & (1u << v)) != 0;
                outside += o;
                tvtxs[k] = tvtx((unsigned char) v, o);
            }
            unsigned int baseParity = permutationParity(tvtxs, tvtxs + 4);
            if (outside > 2)
            {
                baseParity ^= 1;
                for (unsigned int k = 0; k < 4; k++)
                    tvtxs[k].second = !tvtxs[k].second;
            }
            std::sort(tvtxs, tvtxs + 4);
            do
            {
                if (permutationParity(tvtxs, tvtxs + 4) == baseParity)
                {
                    const unsigned int t0 = tvtxs[0].first, t1 = tvtxs[1].first;
                    const unsigned int t2 = tvtxs[2].first, t3 = tvtxs[3].first;
                    unsigned int mask = 0;
                    for (unsigned int k = 0; k < 4; k++)
                        mask |= (unsigned int) tvtxs[k].second << k;
                    if (mask == 0)
                        break;
                    else if (mask == 1)
                    {
                        triangles.push_back(findEdgeByVertexIds(t0, t1));
                        triangles.push_back(findEdgeByVertexIds(t0, t3));
                        triangles.push_back(findEdgeByVertexIds(t0, t2));
                        break;
                    }
                    else if (mask == 3)
                    {
                        triangles.push_back(findEdgeByVertexIds(t0, t2));
                        triangles.push_back(findEdgeByVertexIds(t1, t2));
                        triangles.push_back(findEdgeByVertexIds(t1, t3));

                        triangles.push_back(findEdgeByVertexIds(t1, t3));
                        triangles.push_back(findEdgeByVertexIds(t0, t3));
                        triangles.push_back(findEdgeByVertexIds(t0, t2));
                        break;
                    }
                }
            } while (std::next_permutation(tvtxs, tvtxs + 4));
        }

        int edgeCompact[NUM_EDGES];
        int pool = 0;
        for (unsigned int j = 0; j < NUM_EDGES; j++)
        {
            if (std::count(triangles.begin(), triangles.end(), j))
            {
                edgeCompact[j] = pool++;
                vertexTable.push_back((uint8_t) j);
                for (unsigned int axis = 0; axis < 3; axis++)
                    t.key.push_back(((edgeIndices[j][0] >> axis) & 1) + ((edgeIndices[j][1] >> axis) & 1));
            }
        }
        for (size_t j = 0; j < triangles.size(); j++)
            indexTable.push_back((uint8_t) edgeCompact[triangles[j]]);
        t.count[i][0] = (uint8_t) (vertexTable.size() - t.start[i][0]);
        t.count[i][1] = (uint8_t) (indexTable.size() - t.start[i][1]);
    }
    t.start[NUM_CUBES][0] = (uint16_t) vertexTable.size();
    t.start[NUM_CUBES][1] = (uint16_t) indexTable.size();
    for (unsigned int i = 0; i <= NUM_CUBES; i++)
        t.start[i][1] = (uint16_t) (t.start[i][1] + vertexTable.size());
    t.data = vertexTable;
    t.data.insert(t.data.end(), indexTable.begin(), indexTable.end());
}

/* kernels/marching.cl:148-154 */
static inline uint64_t computeKey(const uint32_t c[3], const uint32_t top[3])
{
    uint64_t key = ((uint64_t) c[2] << (2 * KEY_AXIS_BITS)) | ((uint64_t) c[1] << KEY_AXIS_BITS) | (uint64_t) c[0];
    if (c[0] == 0 || c[1] == 0 || c[0] == top[0] || c[1] == top[1] || c[2] == top[2])
        key |= KEY_EXTERNAL_FLAG;
    return key;
}

/* kernels/marching.cl:130-138 */
static inline void interp(float iso0, float iso1, const uint32_t cell[3], unsigned int c0, unsigned int c1, float out[3])
{
    float inv = 1.0f / (iso0 - iso1);
    float t = iso0 * inv;
    for (int a = 0; a < 3; a++)
    {
        uint32_t o0 = (c0 >> a) & 1, o1 = (c1 >> a) & 1;
        uint32_t delta = o1 - o0;   /* uint3 arithmetic, then convert_float3 */
        out[a] = fmaf(t, (float) delta, (float) (cell[a] + o0));
    }
}

/* kernels/marching.cl:295-326 */
static void compactVertices(float *outVertices, uint64_t *outKeys, uint32_t *indexRemap, uint32_t *firstExternal,
                            const uint32_t *vertexUnique, const float *inVertices /* float4 */, const uint64_t *inKeys,
                            uint64_t minExternalKey, uint64_t keyOffset, size_t n)
{
    for (size_t gid = 0; gid < n; gid++)
    {
        const uint32_t u = vertexUnique[gid];
        const float *v = inVertices + 4 * gid;
        const uint64_t key = inKeys[gid];
        const uint64_t nextKey = inKeys[gid + 1];
        bool ext = key >= minExternalKey;
        if (key != nextKey)
        {
            outVertices[3 * (size_t) u + 0] = v[0];
            outVertices[3 * (size_t) u + 1] = v[1];
            outVertices[3 * (size_t) u + 2] = v[2];
            if (ext)
            {
                outKeys[u] = (key & (KEY_EXTERNAL_FLAG - 1)) + keyOffset;
                if (u == 0)
                    *firstExternal = 0;
            }
            else if (nextKey >= minExternalKey)
                *firstExternal = u + 1;
        }
        uint32_t originalIndex;
        std::memcpy(&originalIndex, &v[3], 4);
        indexRemap[originalIndex] = u;
    }
}

static inline uint32_t roundUp(uint32_t a, uint32_t b) { return (a + b - 1) / b * b; }

typedef void (*GeneratorFn)(void *user, float *field, size_t pitch, const Swathe *swathe);
typedef void (*OutputFn)(void *user, const float *vertices, const uint64_t *keys, const uint32_t *triangles,
                         uint64_t numVertices, uint64_t numTriangles, uint64_t numInternal);

struct MarchingStats
{
    uint64_t occupied, unweldedVertices, indices, weldedVertices, externalVertices, shipOuts, overflows;
};

class Marching
{
public:
    uint32_t maxWidth, maxHeight, maxDepth, maxSwathe;
    uint32_t imageWidth, imageHeight, zStride;
    size_t vertexSpace, indexSpace;
    Tables tables;
    std::vector<float> image;
    std::vector<uint32_t> cells;      /* uint3 per occupied cell */
    std::vector<uint32_t> viCount;    /* uint2 per occupied cell */
    std::vector<uint32_t> viHistogram;/* uint2 per slice */
    std::vector<float> unweldedVertices;   /* float4 */
    std::vector<uint64_t> unweldedKeys;
    std::vector<uint32_t> indices;
    MarchingStats stats;

    Marching() { std::memset(&stats, 0, sizeof(stats)); }

    /* src/marching.cpp:336-445 */
    int init(uint32_t mw, uint32_t mh, uint32_t md, uint32_t ms, size_t meshMemory, const uint32_t alignment[3])
    {
        if (!(2 <= mw && mw <= MAX_DIMENSION)) return 1;
        if (!(2 <= mh && mh <= MAX_DIMENSION)) return 1;
        if (!(2 <= md && md <= MAX_DIMENSION)) return 1;
        if (!(alignment[2] <= ms)) return 1;
        if (!(meshMemory >= (size_t) (mw - 1) * (mh - 1) * MAX_CELL_BYTES)) return 1;
        maxWidth = mw; maxHeight = mh; maxDepth = md;
        imageWidth = roundUp(mw, alignment[0]);
        imageHeight = roundUp(mh, alignment[1]);
        maxSwathe = std::min(ms, md) / alignment[2] * alignment[2];
        makeTables(tables);
        image.assign((size_t) imageWidth * imageHeight * (maxSwathe + 1), 0.0f);
        zStride = imageHeight;
        const size_t meshCells = meshMemory / MAX_CELL_BYTES;
        vertexSpace = meshCells * MAX_CELL_VERTICES;
        indexSpace = meshCells * MAX_CELL_INDICES;
        viHistogram.assign((size_t) md * 2, 0);
        unweldedVertices.resize(vertexSpace * 4);
        unweldedKeys.resize(vertexSpace + 1);
        indices.resize(indexSpace);
        return 0;
    }

    inline float px(uint32_t x, uint32_t row) const { return image[(size_t) row * imageWidth + x]; }

    /* generateCells (src/marching.cpp:500-551) + genOccupied (kernels/marching.cl:84-120).
     * The reference appends cells with atomic_inc (nondeterministic order); the
     * oracle and the HIP path both use cell-linear (z, y, x) order. */
    size_t generateCells(const Swathe &sw)
    {
        for (uint32_t z = sw.zFirst; z < sw.zLast; z++)
            viHistogram[2 * z] = viHistogram[2 * z + 1] = 0;
        cells.clear();
        viCount.clear();
        for (uint32_t z = sw.zFirst; z < sw.zLast; z++)
            for (uint32_t y = 0; y + 1 < sw.height; y++)
                for (uint32_t x = 0; x + 1 < sw.width; x++)
                {
                    uint32_t y0 = y + sw.zStride * z + sw.zBias;
                    uint32_t y1 = y0 + sw.zStride;
                    float iso[8] = {px(x, y0), px(x + 1, y0), px(x, y0 + 1), px(x + 1, y0 + 1),
                                    px(x, y1), px(x + 1, y1), px(x, y1 + 1), px(x + 1, y1 + 1)};
                    uint32_t code = 0;
                    bool valid = true;
                    for (int i = 0; i < 8; i++)
                    {
                        code |= (iso[i] >= 0.0f ? 1u : 0u) << i;
                        valid = valid && std::isfinite(iso[i]);
                    }
                    if (valid && code != 0 && code != 255)
                    {
                        cells.push_back(x); cells.push_back(y); cells.push_back(z);
                        uint32_t nv = tables.count[code][0], ni = tables.count[code][1];
                        viCount.push_back(nv); viCount.push_back(ni);
                        viHistogram[2 * z] += nv;
                        viHistogram[2 * z + 1] += ni;
                    }
                }
        return cells.size() / 3;
    }

    /* kernels/marching.cl:184-258 over the compacted cells, after the seeded scan */
    void generateElements(const Swathe &sw, size_t compacted, const uint32_t offsets[2],
                          const uint32_t gridOffset[3], const uint32_t top[3])
    {
        uint32_t vNext = offsets[0], iNext = offsets[1];
        for (size_t gid = 0; gid < compacted; gid++)
        {
            const uint32_t cell[3] = {cells[3 * gid], cells[3 * gid + 1], cells[3 * gid + 2]};
            const uint32_t y0 = cell[2] * sw.zStride + sw.zBias + cell[1];
            const uint32_t y1 = y0 + sw.zStride;
            const uint32_t globalCell[3] = {cell[0] + gridOffset[0], cell[1] + gridOffset[1], cell[2] + gridOffset[2]};
            const uint32_t x = cell[0];
            float iso[8] = {px(x, y0), px(x + 1, y0), px(x, y0 + 1), px(x + 1, y0 + 1),
                            px(x, y1), px(x + 1, y1), px(x, y1 + 1), px(x + 1, y1 + 1)};
            float lverts[NUM_EDGES][3];
            for (int e = 0; e < NUM_EDGES; e++)
                interp(iso[edgeIndices[e][0]], iso[edgeIndices[e][1]], globalCell,
                       edgeIndices[e][0], edgeIndices[e][1], lverts[e]);
            uint32_t code = 0;
            for (int i = 0; i < 8; i++)
                code |= (iso[i] >= 0.0f ? 1u : 0u) << i;
            const uint16_t *st = tables.start[code];
            const uint16_t *en = tables.start[code + 1];
            const uint32_t nv = en[0] - st[0], ni = en[1] - st[1];
            for (uint32_t i = 0; i < nv; i++)
            {
                const float *lv = lverts[tables.data[st[0] + i]];
                float *v = &unweldedVertices[4 * (size_t) (vNext + i)];
                v[0] = lv[0]; v[1] = lv[1]; v[2] = lv[2];
                uint32_t idx = vNext + i;
                std::memcpy(&v[3], &idx, 4);
                uint32_t c[3];
                for (int a = 0; a < 3; a++)
                    c[a] = 2 * cell[a] + tables.key[3 * (st[0] + i) + a];
                unweldedKeys[vNext + i] = computeKey(c, top);
            }
            for (uint32_t i = 0; i < ni; i++)
                indices[iNext + i] = vNext + tables.data[st[1] + i];
            vNext += nv;
            iNext += ni;
        }
    }

    /* src/marching.cpp:553-625 */
    void shipOut(const uint32_t keyOffset[3], const uint32_t sizes[2], uint32_t zMax, OutputFn output, void *user)
    {
        const size_t nv = sizes[0], ni = sizes[1];
        unweldedKeys[nv] = UINT64_MAX;
        /* sortVertices: stable, all 64 bits (maxBits = 0), src/marching.cpp:572 */
        std::vector<uint32_t> order(nv);
        for (size_t i = 0; i < nv; i++) order[i] = (uint32_t) i;
        std::stable_sort(order.begin(), order.end(),
                         [&](uint32_t a, uint32_t b) { return unweldedKeys[a] < unweldedKeys[b]; });
        std::vector<uint64_t> sk(nv + 1);
        std::vector<float> sv(nv * 4);
        for (size_t i = 0; i < nv; i++)
        {
            sk[i] = unweldedKeys[order[i]];
            std::memcpy(&sv[4 * i], &unweldedVertices[4 * (size_t) order[i]], 16);
        }
        sk[nv] = UINT64_MAX;
        /* countUniqueVertices (kernels/marching.cl:271-279) + scan over nv+1 */
        std::vector<uint32_t> vertexUnique(nv + 1);
        uint32_t sum = 0;
        for (size_t i = 0; i <= nv; i++)
        {
            vertexUnique[i] = sum;
            if (i < nv)
                sum += (sk[i] != sk[i + 1]) ? 1 : 0;
        }
        const uint32_t numWelded = vertexUnique[nv];
        uint64_t minExternalKey = (uint64_t) zMax << (2 * KEY_AXIS_BITS + 1);
        uint64_t keyOffsetL = ((uint64_t) keyOffset[2] << (2 * KEY_AXIS_BITS + 1))
            | ((uint64_t) keyOffset[1] << (KEY_AXIS_BITS + 1))
            | ((uint64_t) keyOffset[0] << 1);
        std::vector<float> welded((size_t) numWelded * 3 + 3);
        std::vector<uint64_t> weldedKeys((size_t) numWelded + 1, 0);
        std::vector<uint32_t> indexRemap(nv + 1);
        uint32_t firstExternal = 0;
        compactVertices(welded.data(), weldedKeys.data(), indexRemap.data(), &firstExternal,
                        vertexUnique.data(), sv.data(), sk.data(), minExternalKey, keyOffsetL, nv);
        /* reindex, kernels/marching.cl:334-340 */
        std::vector<uint32_t> tri(ni + 1);
        for (size_t i = 0; i < ni; i++)
            tri[i] = indexRemap[indices[i]];
        stats.shipOuts++;
        stats.weldedVertices += numWelded;
        stats.externalVertices += numWelded - firstExternal;
        stats.unweldedVertices += nv;
        stats.indices += ni;
        output(user, welded.data(), weldedKeys.data(), tri.data(), numWelded, ni / 3, firstExternal);
    }

    /* src/marching.cpp:627-743 */
    void addSlices(OutputFn output, void *user, const Swathe &swathe, const uint32_t keyOffset[3],
                   uint32_t offsets[2], uint32_t &zTop)
    {
        uint32_t top[3] = {2 * (swathe.width - 1), 2 * (swathe.height - 1), 2 * zTop};
        size_t compacted = generateCells(swathe);
        if (compacted > 0)
        {
            uint32_t counts[2] = {0, 0};
            for (uint32_t i = swathe.zFirst; i < swathe.zLast; i++)
                for (int j = 0; j < 2; j++)
                    counts[j] += viHistogram[2 * i + j];
            if (counts[0] > vertexSpace || counts[1] > indexSpace)
            {
                stats.overflows++;
                /* Copy: the recursive calls overwrite viHistogram entries of
                 * the sub-swathe with identical values, so reading it live
                 * (as the reference does with viReadback) is equivalent. */
                uint32_t subFirst = swathe.zFirst;
                while (subFirst < swathe.zLast)
                {
                    uint32_t subLast = subFirst;
                    counts[0] = counts[1] = 0;
                    while (subLast < swathe.zLast
                           && (uint64_t) offsets[0] + counts[0] + viHistogram[2 * subLast] <= vertexSpace
                           && (uint64_t) offsets[1] + counts[1] + viHistogram[2 * subLast + 1] <= indexSpace)
                    {
                        counts[0] += viHistogram[2 * subLast];
                        counts[1] += viHistogram[2 * subLast + 1];
                        subLast++;
                    }
                    if (subFirst == subLast)
                    {
                        while (subLast < swathe.zLast
                               && (uint64_t) counts[0] + viHistogram[2 * subLast] <= vertexSpace
                               && (uint64_t) counts[1] + viHistogram[2 * subLast + 1] <= indexSpace)
                        {
                            counts[0] += viHistogram[2 * subLast];
                            counts[1] += viHistogram[2 * subLast + 1];
                            subLast++;
                        }
                    }
                    assert(subLast > subFirst);
                    Swathe sub = swathe;
                    sub.zFirst = subFirst;
                    sub.zLast = subLast;
                    addSlices(output, user, sub, keyOffset, offsets, zTop);
                    subFirst = subLast;
                }
            }
            else
            {
                if ((uint64_t) offsets[0] + counts[0] > vertexSpace
                    || (uint64_t) offsets[1] + counts[1] > indexSpace)
                {
                    shipOut(keyOffset, offsets, swathe.zFirst, output, user);
                    offsets[0] = offsets[1] = 0;
                    zTop = swathe.zFirst;
                    top[2] = 2 * swathe.zFirst;
                }
                stats.occupied += compacted;
                generateElements(swathe, compacted, offsets, keyOffset, top);
                offsets[0] += counts[0];
                offsets[1] += counts[1];
            }
        }
    }

    /* copySlice: row-block src -> row-block trg (src/marching.cpp:447-498) */
    void copySlice(uint32_t src, uint32_t trg, const Swathe &sw)
    {
        for (uint32_t y = 0; y < sw.height; y++)
            std::memcpy(&image[((size_t) trg * sw.zStride + y) * imageWidth],
                        &image[((size_t) src * sw.zStride + y) * imageWidth],
                        sw.width * sizeof(float));
    }

    /* src/marching.cpp:745-824 */
    int generate(GeneratorFn generator, void *genUser, OutputFn output, void *outUser,
                 const uint32_t size[3], const uint32_t keyOffset[3])
    {
        Swathe swathe;
        swathe.width = size[0];
        swathe.height = size[1];
        swathe.zStride = zStride;
        const uint32_t depth = size[2];
        if (!(1 <= swathe.width && swathe.width <= maxWidth)) return 1;
        if (!(1 <= swathe.height && swathe.height <= maxHeight)) return 1;
        if (!(1 <= depth && depth <= maxDepth)) return 1;
        uint32_t offsets[2] = {0, 0};
        uint32_t zTop = 0;
        for (uint32_t z = 0; z < depth; z += maxSwathe)
        {
            swathe.zFirst = z;
            swathe.zLast = std::min(depth, z + maxSwathe) - 1;
            swathe.zBias = (1 - (int32_t) z) * (int32_t) swathe.zStride;
            if (z != 0)
                copySlice(maxSwathe, 0, swathe);
            generator(genUser, image.data(), imageWidth, &swathe);
            if (z > 0)
                swathe.zFirst--;
            addSlices(output, outUser, swathe, keyOffset, offsets, zTop);
        }
        if (offsets[0] > 0)
            shipOut(keyOffset, offsets, depth - 1, output, outUser);
        return 0;
    }
};

/* src/workers.cpp:169-182 */
static uint32_t computeMaxSwathe(uint32_t yMax, uint32_t y, uint32_t yAlign, uint32_t zAlign)
{
    y = roundUp(y, yAlign);
    if (yMax < y)
        return zAlign;
    uint32_t chunks = (yMax - y) / (y * zAlign);
    if (chunks == 0)
        chunks = 1;
    return chunks * zAlign;
}

struct MlsGenerator
{
    const Splat *splats;
    const command_type *commands;
    const command_type *start;
    uint32_t startShift;
    int32_t offset[3];
    float boundaryFactor;
    int shape;
    MlsStats stats;
    double seconds = 0.0;       /* wall time spent in processCorners (bench.py's cpu_baseline reports it per stage) */
};

static void mlsGeneratorFn(void *user, float *field, size_t pitch, const Swathe *sw)
{
    MlsGenerator *g = (MlsGenerator *) user;
    const double t0 = omp_get_wtime();
    processCorners(field, pitch, g->splats, g->commands, g->start, g->startShift, g->offset,
                   *sw, g->boundaryFactor, g->shape, &g->stats);
    g->seconds += omp_get_wtime() - t0;
}

/* src/mls.cpp:137-144 */
static float boundaryFactorFromLimit(float limit)
{
    const float pi = 3.14159265358979323846f;
    const float boundaryScale = (sqrtf(6.0f) * 512) / (693 * pi);
    const float gamma = boundaryScale * limit;
    return 1.0f - gamma * gamma;
}

} // namespace

/* ================================================================== */
/* C entry points for ctypes (tests / bench cpu_baseline only)          */
/* ================================================================== */

ORC_API uint32_t orc_make_code(int x, int y, int z) { return makeCode(x, y, z); }
ORC_API void orc_decode(uint32_t code, int out[3]) { decode(code, out); }
ORC_API int orc_level_shift(const int lo[3], const int hi[3]) { return levelShift(lo, hi); }
ORC_API float orc_point_box_dist2(const float p[3], const float lo[3], const float hi[3])
{
    return pointBoxDist2(p, lo, hi);
}
ORC_API float orc_solve_quadratic(float a, float b, float c) { return solveQuadratic(a, b, c); }
ORC_API float orc_boundary_factor(float limit) { return boundaryFactorFromLimit(limit); }

/* testProjectDistOriginSphere, kernels/mls.cl:445-449: Sphere = {a=p3, b=(p0,p1,p2), c=p4} */
ORC_API float orc_project_dist_origin_sphere(float p0, float p1, float p2, float p3, float p4)
{
    Sphere s;
    s.a = p3; s.b[0] = p0; s.b[1] = p1; s.b[2] = p2; s.c = p4;
    s.qDen = 0; s.b2 = 0;
    return projectDistOriginSphere(s);
}

/* testFitSphere, kernels/mls.cl:451-469 */
ORC_API void orc_fit_sphere(const void *splats_, uint32_t n, float out[5])
{
    const Splat *in = (const Splat *) splats_;
    SphereFit sf;
    sphereFitInit(sf);
    for (uint32_t i = 0; i < n; i++)
        sphereFitAdd(sf, in[i].quality, in[i].position, dot3(in[i].position, in[i].position), in[i].normal);
    Sphere s;
    fitSphere(sf, s);
    out[0] = s.b[0]; out[1] = s.b[1]; out[2] = s.b[2]; out[3] = s.a; out[4] = s.c;
}

ORC_API uint64_t orc_compute_key(const uint32_t c[3], const uint32_t top[3]) { return computeKey(c, top); }

ORC_API void orc_compact_vertices(float *outVertices, uint64_t *outKeys, uint32_t *indexRemap, uint32_t *firstExternal,
                                  const uint32_t *vertexUnique, const float *inVertices, const uint64_t *inKeys,
                                  uint64_t minExternalKey, uint64_t keyOffset, uint64_t n)
{
    compactVertices(outVertices, outKeys, indexRemap, firstExternal, vertexUnique, inVertices, inKeys,
                    minExternalKey, keyOffset, n);
}

/* count[256][2] u8, start[257][2] u16, data[8192] u8, key[2432][3] u32 */
ORC_API void orc_make_tables(uint8_t *count, uint16_t *start, uint8_t *data, uint32_t *key,
                             uint32_t *dataSize, uint32_t *keyEntries)
{
    Tables t;
    makeTables(t);
    std::memcpy(count, t.count, sizeof(t.count));
    std::memcpy(start, t.start, sizeof(t.start));
    std::memcpy(data, t.data.data(), std::min<size_t>(t.data.size(), 8192));
    std::memcpy(key, t.key.data(), std::min<size_t>(t.key.size(), 2432 * 3) * 4);
    *dataSize = (uint32_t) t.data.size();
    *keyEntries = (uint32_t) (t.key.size() / 3);
}

ORC_API uint32_t orc_compute_max_swathe(uint32_t yMax, uint32_t y, uint32_t yAlign, uint32_t zAlign)
{
    return computeMaxSwathe(yMax, y, yAlign, zAlign);
}

ORC_API void orc_scale_bias(float *vertices, uint64_t n, float scale, float bx, float by, float bz)
{
    /* kernels/scale_bias.cl:33-41 */
    for (uint64_t i = 0; i < n; i++)
    {
        vertices[3 * i + 0] = fmaf(vertices[3 * i + 0], scale, bx);
        vertices[3 * i + 1] = fmaf(vertices[3 * i + 1], scale, by);
        vertices[3 * i + 2] = fmaf(vertices[3 * i + 2], scale, bz);
    }
}

/* ---- octree ---- */
struct orc_tree { TreeResult r; };

ORC_API orc_tree *orc_tree_build(void *splats, uint64_t firstSplat, uint64_t numSplats,
                                 const uint32_t size[3], const int32_t offset[3],
                                 uint32_t subsamplingShift, uint32_t maxLevels)
{
    orc_tree *t = new orc_tree;
    if (treeBuild((Splat *) splats, firstSplat, numSplats, size, offset, subsamplingShift, maxLevels, t->r) != 0)
    {
        delete t;
        return NULL;
    }
    return t;
}
ORC_API void orc_tree_free(orc_tree *t) { delete t; }
ORC_API const int32_t *orc_tree_commands(const orc_tree *t) { return t->r.commands.data(); }
ORC_API const int32_t *orc_tree_start(const orc_tree *t) { return t->r.start.data(); }
ORC_API uint64_t orc_tree_num_commands(const orc_tree *t) { return t->r.numCommands; }
ORC_API uint64_t orc_tree_num_start(const orc_tree *t) { return t->r.numStart; }
ORC_API uint64_t orc_tree_commands_size(const orc_tree *t) { return t->r.commands.size(); }
ORC_API uint64_t orc_tree_start_size(const orc_tree *t) { return t->r.start.size(); }
ORC_API uint32_t orc_tree_num_levels(const orc_tree *t) { return t->r.numLevels; }

/* ---- MLS ---- */
ORC_API void orc_process_corners(float *field, uint64_t pitch, const void *splats,
                                 const int32_t *commands, const int32_t *start,
                                 uint32_t subsamplingShift, const int32_t offset[3],
                                 uint32_t width, uint32_t height, uint32_t zStride, int32_t zBias,
                                 uint32_t zFirst, uint32_t zLast, float boundaryFactor, int shape,
                                 uint64_t *stats /* [2] or NULL */)
{
    Swathe sw = {width, height, zStride, zBias, zFirst, zLast};
    MlsStats st = {0, 0};
    processCorners(field, pitch, (const Splat *) splats, commands, start, 3 * subsamplingShift, offset,
                   sw, boundaryFactor, shape, &st);
    if (stats) { stats[0] += st.listed; stats[1] += st.hits; }
}

/* ---- Marching ---- */
struct orc_marching { Marching m; };

ORC_API orc_marching *orc_marching_create(uint32_t maxWidth, uint32_t maxHeight, uint32_t maxDepth,
                                          uint32_t maxSwathe, uint64_t meshMemory, const uint32_t alignment[3])
{
    orc_marching *m = new orc_marching;
    if (m->m.init(maxWidth, maxHeight, maxDepth, maxSwathe, meshMemory, alignment) != 0)
    {
        delete m;
        return NULL;
    }
    return m;
}
ORC_API void orc_marching_free(orc_marching *m) { delete m; }
ORC_API int orc_marching_generate(orc_marching *m, GeneratorFn gen, void *genUser, OutputFn out, void *outUser,
                                  const uint32_t size[3], const uint32_t keyOffset[3])
{
    return m->m.generate(gen, genUser, out, outUser, size, keyOffset);
}
ORC_API void orc_marching_stats(const orc_marching *m, uint64_t out[7])
{
    std::memcpy(out, &m->m.stats, sizeof(MarchingStats));
}
ORC_API void orc_marching_copy_slice(orc_marching *m, float *image, uint64_t pitch, uint32_t src, uint32_t trg,
                                     uint32_t width, uint32_t height, uint32_t zStride)
{
    /* stand-alone form of Marching::copySlice for test/test_marching.cpp:481-548 */
    (void) m;
    for (uint32_t y = 0; y < height; y++)
        std::memmove(&image[((size_t) trg * zStride + y) * pitch],
                     &image[((size_t) src * zStride + y) * pitch], width * sizeof(float));
}

/* ---- whole bucket: DeviceWorkerGroupBase::Worker::operator(), src/workers.cpp:232-286 ---- */
/*
 * splats: mutated (radius -> 1/r^2) exactly like the device buffer.
 * size = numVertices of the bucket grid; offset = keyOffset = low extents.
 * stats (optional, 16 u64): [0]=Sigma L, [1]=H, [2..8]=MarchingStats, [9]=numCommands,
 * [10..12] = microseconds of wall time in the octree build, in processCorners, in marching (generate minus MLS)
 */
ORC_API int orc_bucket(void *splats, uint64_t firstSplat, uint64_t numSplats,
                       const uint32_t size[3], const int32_t offset[3],
                       uint32_t levels, uint32_t subsampling, float boundaryLimit, int shape,
                       uint32_t maxCells, uint32_t maxSwathe, uint64_t meshMemory,
                       OutputFn out, void *outUser, uint64_t *stats)
{
    const uint32_t wgs[3] = {8, 8, 8};
    uint32_t expanded[3];
    for (int i = 0; i < 3; i++)
        expanded[i] = roundUp(size[i], wgs[i]);
    TreeResult tree;
    const double tTree = omp_get_wtime();
    if (treeBuild((Splat *) splats, firstSplat, numSplats, expanded, offset, subsampling, levels, tree) != 0)
        return 1;
    const double treeSeconds = omp_get_wtime() - tTree;
    MlsGenerator gen;
    gen.splats = (const Splat *) splats;
    gen.commands = tree.commands.data();
    gen.start = tree.start.data();
    gen.startShift = 3 * subsampling;
    for (int i = 0; i < 3; i++) gen.offset[i] = offset[i];
    gen.boundaryFactor = boundaryFactorFromLimit(boundaryLimit);
    gen.shape = shape;
    gen.stats.listed = gen.stats.hits = 0;
    Marching m;
    if (m.init(maxCells + 1, maxCells + 1, maxCells + 1, maxSwathe, meshMemory, wgs) != 0)
        return 2;
    uint32_t keyOffset[3] = {(uint32_t) offset[0], (uint32_t) offset[1], (uint32_t) offset[2]};
    const double tGen = omp_get_wtime();
    if (m.generate(mlsGeneratorFn, &gen, out, outUser, size, keyOffset) != 0)
        return 3;
    const double genSeconds = omp_get_wtime() - tGen;
    if (stats)
    {
        stats[10] += (uint64_t) (treeSeconds * 1e6);
        stats[11] += (uint64_t) (gen.seconds * 1e6);
        stats[12] += (uint64_t) ((genSeconds - gen.seconds) * 1e6);
        stats[0] += gen.stats.listed;
        stats[1] += gen.stats.hits;
        const uint64_t *ms = (const uint64_t *) &m.stats;
        for (int i = 0; i < 7; i++) stats[2 + i] += ms[i];
        stats[9] += tree.numCommands;
    }
    return 0;
}

ORC_API void orc_set_rsqrt_bits(int bits) { g_rsqrtBits = bits; }

ORC_API int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
