/*
 * Moving-least-squares corner evaluation on gfx950 -- the device half of MlsFunctor
 * (reference: src/mls.{h,cpp}, kernels/mls.cl:299-433 processCorners).
 *
 * One 512-thread workgroup (8 waves) evaluates one 8x8x8 block of grid corners, all of which share
 * one octree leaf and therefore one splat list (src/mls.cpp:53-54: wgs = {8,8,8}, subsamplingMin 3).
 * Per corner the splats are accumulated in LIST ORDER (leaf range first, then each ancestor's),
 * which fixes the floating-point summation order; the kernels below keep that order, so their
 * results are bit-identical to each other and to the oracle.
 *
 * Variant 5 (default, "matrix prefilter", processCornersMatrixKernel): 512 listed splats per round are staged in LDS and
 * tagged with the 4x4x4 sub-blocks their support can reach; each wave owns a sub-block, finds the candidate (corner, splat)
 * pairs of 32 relevant splats at a time with two bf16 MFMAs and accumulates them, in list order, under the reference's own
 * test.  See the kernel.
 * Variant 4 ("cube streams", processCornersCubeKernel), round 5's default: the same staging; each wave gives each of its
 * eight 2x2x2 cubes of corners its own stream of candidate splats and tests them on the vector units.  Kept for A/B.
 * Variant 1 ("basic", processCornersKernel): the reference's structure (every thread walks every staged splat), kept as
 * the A/B partner of the parity tests.
 * (Rounds 1-3 carried three intermediate designs -- sub-block culling alone, per-lane hit lists, hit masks -- as variants
 * 0, 2 and 3; they are gone from the product, their measurements are in profiles/NOTES_r0*.md.)
 *
 * All kernels have a bucket dimension (common.hpp, Lanes): blockIdx.y selects the bucket of a batch.
 *
 * Floating-point contract: compiled with -ffp-contract=off; fmaf only where written.
 */
#include "common.hpp"

#include <algorithm>
#include <cstdlib>

using namespace mlsgpu;

struct mlsgpu_tree;
extern "C" const mlsgpu_splat *mlsgpu_hip_tree_splats(const mlsgpu_tree *);
extern "C" const int32_t *mlsgpu_hip_tree_commands(const mlsgpu_tree *);
extern "C" const int32_t *mlsgpu_hip_tree_start(const mlsgpu_tree *);
extern "C" int mlsgpu_hip_tree_mutates(const mlsgpu_tree *);

struct mlsgpu_mls
{
    mlsgpu_ctx *ctx = nullptr;
    int shape = MLSGPU_SHAPE_SPHERE;
    int variant = 5;             /* sub-block culling + matrix-core prefilter; 4 = culled + cube streams (round 5's default); 1 = the reference's loop structure (variants 0, 2, 3 were removed in round 4) */
    const mlsgpu_splat *dSplats = nullptr;
    const int32_t *dCommands = nullptr;
    const int32_t *dStart = nullptr;
    uint32_t startShift = 0;
    int32_t offset[3] = {0, 0, 0};
    float boundaryFactor = 0.0f;
    bool isSet = false;
    bool rawRadius = false;      /* the splats still hold the radius (a tree built without mutation): 1/r^2 is taken at staging */
    unsigned long long *dStats = nullptr;   /* optional work counters, see mlsgpu_hip_mls_set_stats */
};

namespace
{

#define RADIUS_CUTOFF 0.99f
#define HITS_CUTOFF 4u
#define STAGE 512

/* kernels/mls.cl:105-108 */
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return fmaf(ax, bx, fmaf(ay, by, az * bz));
}

struct Fit
{
    float sumWpp, sumWpn;
    float sumWpx, sumWpy, sumWpz;
    float sumWnx, sumWny, sumWnz;
    float sumW;
    uint32_t hits;
};

__device__ __forceinline__ void fitInit(Fit &f)
{
    f.sumWpp = f.sumWpn = 0.0f;
    f.sumWpx = f.sumWpy = f.sumWpz = 0.0f;
    f.sumWnx = f.sumWny = f.sumWnz = 0.0f;
    f.sumW = 0.0f;
    f.hits = 0;
}

/* 1 iff d < RADIUS_CUTOFF, as the sign bit of the difference instead of a compare: a compare writes VCC and the
 * add-with-carry behind it waits for that (6.8 issue cycles per instruction of the pair at 8 waves per SIMD against 2.6 for
 * plain vector instructions, profiles/r03_valu_lds_issue_microbench.txt).  The subtraction is exact for d within a factor
 * of two of the cutoff and has the sign of the exact difference otherwise; d == cutoff gives +0, -inf gives -inf (a hit for
 * the compare as well) and the NaN the hardware MAKES (0 * inf, inf - inf) is the positive one, which a compare rejects
 * too.  A NaN that comes IN with a splat keeps its sign through the arithmetic, and a negative one would read as a hit:
 * such a splat never reaches a test -- staging gives a record with a NaN position or 1/r^2 an empty sub-block mask
 * (splatIsNan), which is what `d < cutoff` does with it at every corner. */
__device__ __forceinline__ uint32_t hitBit(float d)
{
    return __float_as_uint(d - RADIUS_CUTOFF) >> 31;
}

/* a staged record no corner can accept: d = |p - c|^2 * invr2 is a NaN for every corner */
__device__ __forceinline__ bool splatIsNan(const float4 pr)
{
    return (pr.x != pr.x) | (pr.y != pr.y) | (pr.z != pr.z) | (pr.w != pr.w);
}

/* The box tests that cull a splat for a sub-block or a cube are conservative because d grows with the distance -- for
 * 1/r^2 >= 0, which is all a tree can hold, and for finite negative values too (every d is <= 0: hits everywhere, boxes
 * included).  The one value of a hand-built list (mlsgpu_hip_mls_set_buffers) that breaks it is -inf: a corner at any
 * distance > 0 has d = -inf, a hit for `d < cutoff`, while a box that CONTAINS the splat has distance 0 and 0 * -inf is a
 * NaN.  Such a record is not culled. */
__device__ __forceinline__ bool splatNeverCulled(const float4 pr)
{
    return __float_as_uint(pr.w) == 0xFF800000u;
}

/* sphereFitAdd, kernels/mls.cl:129-139 (planeFitAdd :141-148 keeps a subset of the same sums). */
__device__ __forceinline__ void fitAdd(Fit &f, float w, float px, float py, float pz, float pp,
                                       float nx, float ny, float nz)
{
    const float wnx = w * nx, wny = w * ny, wnz = w * nz;
    f.sumW = f.sumW + w;
    f.sumWpx = fmaf(w, px, f.sumWpx);
    f.sumWpy = fmaf(w, py, f.sumWpy);
    f.sumWpz = fmaf(w, pz, f.sumWpz);
    f.sumWnx = fmaf(w, nx, f.sumWnx);
    f.sumWny = fmaf(w, ny, f.sumWny);
    f.sumWnz = fmaf(w, nz, f.sumWnz);
    f.sumWpp = fmaf(w, pp, f.sumWpp);
    f.sumWpn = f.sumWpn + dot3(wnx, wny, wnz, px, py, pz);
    f.hits++;
}

/* kernels/mls.cl:237-248 */
__device__ __forceinline__ float solveQuadratic(float a, float b, float c)
{
    float bdet = b + sqrtf(b * b - 4.0f * a * c);
    float x = -2.0f * c / bdet;
    if (!isfinite(x))
        x = bdet / (-2.0f * a);
    return isfinite(x) ? x : __int_as_float(0x7FC00000);
}

/* fitSphere + projectOriginSphere + acceptance tests, kernels/mls.cl:210-229,263-267,394-408;
 * plane variant :198-203,277-280,409-422. */
template<int SHAPE>
__device__ __forceinline__ float finishCorner(const Fit &fit, float boundaryFactor)
{
    float f = __int_as_float(0x7FC00000);
    if (fit.hits >= HITS_CUTOFF)
    {
        if (SHAPE == MLSGPU_SHAPE_SPHERE)
        {
            const float invSumW = 1.0f / fit.sumW;
            const float mx = fit.sumWpx * invSumW, my = fit.sumWpy * invSumW, mz = fit.sumWpz * invSumW;
            const float qNum = fit.sumWpn - dot3(mx, my, mz, fit.sumWnx, fit.sumWny, fit.sumWnz);
            const float qDen = fit.sumWpp - dot3(mx, my, mz, fit.sumWpx, fit.sumWpy, fit.sumWpz);
            float q = qNum / qDen;
            if (fabsf(qDen) < (4 * 1.1920928955078125e-07f) * (float) fit.hits * fabsf(fit.sumWpp) || !isfinite(q))
                q = 0.0f;
            const float a = 0.5f * q;
            const float bx = (fit.sumWnx - q * fit.sumWpx) * invSumW;
            const float by = (fit.sumWny - q * fit.sumWpy) * invSumW;
            const float bz = (fit.sumWnz - q * fit.sumWpz) * invSumW;
            const float c = (-a * fit.sumWpp - dot3(bx, by, bz, fit.sumWpx, fit.sumWpy, fit.sumWpz)) * invSumW;
            const float b2 = dot3(bx, by, bz, bx, by, bz);
            const float l = solveQuadratic(a * b2, b2, c);
            const float ax = l * bx, ay = l * by, az = l * bz;
            const float aa = dot3(ax, ay, az, ax, ay, az);
            if (aa < 3.0f)
            {
                const float rhs = (fit.sumWpp - 2 * dot3(fit.sumWpx, fit.sumWpy, fit.sumWpz, ax, ay, az) + fit.sumW * aa);
                if (qDen > boundaryFactor * rhs)
                    f = -dot3(bx, by, bz, ax, ay, az) * (1.0f / sqrtf(b2));   /* half_rsqrt -> exact, DESIGN.md */
            }
        }
        else
        {
            const float mx = fit.sumWpx / fit.sumW, my = fit.sumWpy / fit.sumW, mz = fit.sumWpz / fit.sumW;
            const float inv = 1.0f / sqrtf(dot3(fit.sumWnx, fit.sumWny, fit.sumWnz, fit.sumWnx, fit.sumWny, fit.sumWnz));
            const float nx = fit.sumWnx * inv, ny = fit.sumWny * inv, nz = fit.sumWnz * inv;
            const float dist = -dot3(nx, ny, nz, mx, my, mz);
            const float ax = nx * -dist, ay = ny * -dist, az = nz * -dist;
            const float aa = dot3(ax, ay, az, ax, ay, az);
            if (aa < 3.0f)
            {
                const float qDen = fit.sumWpp - dot3(mx, my, mz, fit.sumWpx, fit.sumWpy, fit.sumWpz);
                const float rhs = (fit.sumWpp - 2 * dot3(fit.sumWpx, fit.sumWpy, fit.sumWpz, ax, ay, az) + fit.sumW * aa);
                if (qDen > boundaryFactor * rhs)
                    f = dist;
            }
        }
    }
    return f;
}

/* z-major Morton code of block-aligned coordinates, kernels/mls.cl:159-174 */
__device__ __forceinline__ uint32_t spread3(uint32_t v)
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

struct MlsArgs
{
    float *field;
    uint64_t pitch;
    const float4 *splats;        /* two float4 per splat */
    const int32_t *commands;
    const int32_t *start;
    uint32_t startShift;
    int32_t ox, oy, oz;
    uint32_t zStride;
    int32_t zBias;
    uint32_t zFirst;
    uint32_t blocksX, blocksY, blocksZ;
    uint32_t numBlocks;          /* blocksX * blocksY * blocksZ: workgroups of this lane; the grid covers the largest lane */
    float boundaryFactor;
    uint32_t xcdChunk;           /* see xcdRemap */
    /* the block's place without a division (processCornersMatrixKernel): a workgroup's first ~100 instructions were four
     * 32-bit divisions.  superShift = log2(8 * xcdChunk) when xcdChunk is a power of two, full = numBlocks rounded down to
     * a multiple of 8 * xcdChunk, magicX / magicY = ceil(2^32 / blocksX), ceil(2^32 / blocksY), exact for every dividend the
     * lane has (checked where they are made: mlsLaneArgs); superShift = 0: none of this holds, the kernel divides. */
    uint32_t superShift, full, magicX, magicY;
    uint32_t rawRadius;          /* splat.w is the radius, not 1/radius^2 */
    unsigned long long *stats;   /* MLSGPU_MLS_STATS_WORDS counters, see mlsgpu_hip_mls_set_stats */
};

/* position and 1/radius^2 of a listed splat.  A tree built without mutation leaves the radius in the splat; the same
 * expression the build would have stored (kernels/octree.cl:193) is evaluated here, once per staged record. */
__device__ __forceinline__ float4 stagedPosRad(const MlsArgs &A, int32_t id)
{
    float4 pr = A.splats[2 * (int64_t) id];
    if (A.rawRadius)
        pr.w = 1.0f / (pr.w * pr.w);
    return pr;
}

/*
 * Workgroup -> block mapping.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so consecutive
 * ids land on different XCDs.  Neighbouring blocks share most of their splat lists, so runs of `chunk` consecutive
 * blocks are kept on one XCD -- but the runs are dealt round-robin: giving each XCD one contiguous eighth of the bucket
 * (chunk = 0, the first version) leaves XCDs idle on surface-like data, where the eighths hold very different amounts
 * of surface.  (Speed only; any mapping gives the same result.)
 */
__device__ __forceinline__ uint32_t xcdRemap(uint32_t id, uint32_t n, uint32_t chunk)
{
    if (chunk == 0)
    {
        /* one contiguous eighth of the blocks per XCD */
        const uint32_t q = n / 8, r = n % 8;
        const uint32_t xcd = id % 8, k = id / 8;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    /* `chunk` consecutive blocks per XCD, chunks dealt round-robin: neighbours still share an L2, and an XCD no longer
     * owns one slab of the bucket (on surface-like data the slabs hold very different amounts of work) */
    const uint32_t super = 8 * chunk;
    const uint32_t full = n / super * super;
    if (id >= full)
        return id;
    const uint32_t g = id / super, j = id % super;
    return g * super + (j % 8) * chunk + j / 8;
}

/* Variant 1: the reference's loop (kernels/mls.cl:342-390) -- every corner tests every listed splat, 512 staged per round */
template<int SHAPE, bool STATS>
__global__ __launch_bounds__(512) void processCornersKernel(Lanes<MlsArgs> lanes)
{
    __shared__ float4 sPosRad[STAGE];
    __shared__ float4 sNormQ[STAGE];

    const MlsArgs A = lanes.a[blockIdx.y];
    if (blockIdx.x >= A.numBlocks)
        return;
    const uint32_t bid = xcdRemap(blockIdx.x, A.numBlocks, A.xcdChunk);
    const uint32_t gx = bid % A.blocksX, gy = (bid / A.blocksX) % A.blocksY, gz = bid / (A.blocksX * A.blocksY);
    const int wx = (int) (gx * 8), wy = (int) (gy * 8), wz = (int) (gz * 8 + A.zFirst);
    /* makeCode(wid) >> startShift (kernels/mls.cl:318) == makeCode(wid >> subsampling): Morton digits are independent */
    const uint32_t sub = A.startShift / 3;
    const uint32_t code = spread3((uint32_t) wx >> sub) | (spread3((uint32_t) wy >> sub) << 1) | (spread3((uint32_t) wz >> sub) << 2);
    int32_t pos = A.start[code];

    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6, lane = tid & 63;
    /* wave = 4x4x4 sub-block (sx,sy,sz); lane = x + 4y + 16z inside it */
    const int lx = (int) ((wave & 1) * 4 + (lane & 3));
    const int ly = (int) (((wave >> 1) & 1) * 4 + ((lane >> 2) & 3));
    const int lz = (int) ((wave >> 2) * 4 + (lane >> 4));

    float f = __int_as_float(0x7FC00000);
    if (pos >= 0)       /* uniform over the workgroup */
    {
        const float cx = (float) (wx + lx + A.ox), cy = (float) (wy + ly + A.oy), cz = (float) (wz + lz + A.oz);
        Fit fit;
        fitInit(fit);
        unsigned long long nListed = 0, nTests = 0;
        int32_t end = A.commands[pos++];
        while (pos < end)
        {
            /* stage up to STAGE listed splats (kernels/mls.cl:342-352 stages 256) */
            const int32_t lpos = pos + (int32_t) tid;
            const int32_t mine = lpos < end ? A.commands[lpos] : -1;
            if (mine >= 0)
            {
                sPosRad[tid] = stagedPosRad(A, mine);
                sNormQ[tid] = A.splats[2 * (int64_t) mine + 1];
            }
            if (STATS)
                nListed += __popcll(__ballot(mine >= 0));
            const int32_t staged = min(end - pos, (int32_t) STAGE);

            pos += STAGE;
            if (pos >= end)
            {
                /* follow the jump (kernels/mls.cl:354-358): negative terminates */
                pos = A.commands[end];
                end = (pos >= 0) ? A.commands[pos++] : INT32_MIN;
            }
            __syncthreads();

            for (int32_t i = 0; i < staged; i++)
            {
                if (STATS)
                    nTests += 64;
                const float4 pr = sPosRad[i];
                const float px = pr.x - cx, py = pr.y - cy, pz = pr.z - cz;
                const float pp = dot3(px, py, pz, px, py, pz);
                const float d = pp * pr.w;
                if (d < RADIUS_CUTOFF)
                {
                    const float4 nq = sNormQ[i];
                    float w = 1.0f - d;
                    w *= w;
                    w *= w;
                    w *= nq.w;
                    fitAdd(fit, w, px, py, pz, pp, nq.x, nq.y, nq.z);
                }
            }
            __syncthreads();
        }
        f = finishCorner<SHAPE>(fit, A.boundaryFactor);
        if (STATS)
        {
            const unsigned long long hits = waveSum(fit.hits);
            if (lane == 0)
            {
                atomicAdd(&A.stats[0], nListed);
                atomicAdd(&A.stats[1], nTests);
                atomicAdd(&A.stats[2], hits);
            }
        }
    }

    const int64_t row = (int64_t) (wy + ly) + (int64_t) (wz + lz) * A.zStride + A.zBias;
    A.field[row * (int64_t) A.pitch + (wx + lx)] = f;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

/*
 * Variant 4 ("cube streams").  What is left of variant 3's time is vector work, and more than half of its vector
 * instructions are distance tests of which one lane in six is a hit: a wave tests all 64 corners of its 4x4x4 sub-block
 * against every splat whose support reaches the sub-block.  Here the wave's eight 2x2x2 CUBES of corners (8 lanes each) get
 * their own streams: the relevant splats of a window (up to 64, compacted as in variant 3) are culled once more, one lane
 * per splat, against the eight cubes (the same conservative box test as the sub-block masks), and each cube's survivors are
 * appended, in window order, to the cube's list in LDS.  A test iteration then serves EIGHT different splats -- lane l
 * tests the k-th splat of its own cube's list -- and the number of iterations is the longest of the eight lists instead of
 * the size of the window: by the Minkowski volumes of box + support sphere (radius 2.5: 140 against 404 cells) a cube sees
 * about a third of what the sub-block sees.  Hits are bits of one 32-bit mask per lane (iteration k of the current chunk of
 * 32), the drain walks them from the top and finds the splat in the lane's own list, so every corner still accumulates its
 * hits in list order: bit-identical results.
 */
#define CUBE_STAGE 512      /* measured per step on cfg3: 512-splat rounds 8.25 ms, 768: 8.36; 25.5 KB of LDS per workgroup */

template<int SHAPE, bool STATS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8))) void processCornersCubeKernel(Lanes<MlsArgs> lanes)
{
    __shared__ float4 sPosRad[CUBE_STAGE];
    __shared__ float4 sNormQ[CUBE_STAGE];
    __shared__ uint8_t sMask[CUBE_STAGE];
    __shared__ uint16_t sSlot[8][128];         /* per wave: byte offsets of the window's relevant splats (64) + what a group
                                                * of 64 staged splats adds beyond a full window */
    __shared__ uint16_t sList[8][8][64];       /* per wave and cube: byte offsets of the splats that can reach the cube */
    __shared__ uint32_t sHist[STATS ? 33 : 1];  /* instrumented build: lanes per number of hits in a drain call */

    const MlsArgs A = lanes.a[blockIdx.y];
    if (blockIdx.x >= A.numBlocks)
        return;
    if (STATS)
    {
        if (threadIdx.x < 33)
            sHist[threadIdx.x] = 0;
        __syncthreads();
    }
    const uint32_t bid = xcdRemap(blockIdx.x, A.numBlocks, A.xcdChunk);
    const uint32_t gx = bid % A.blocksX, gy = (bid / A.blocksX) % A.blocksY, gz = bid / (A.blocksX * A.blocksY);
    const int wx = (int) (gx * 8), wy = (int) (gy * 8), wz = (int) (gz * 8 + A.zFirst);
    const uint32_t sub = A.startShift / 3;
    const uint32_t code = spread3((uint32_t) wx >> sub) | (spread3((uint32_t) wy >> sub) << 1) | (spread3((uint32_t) wz >> sub) << 2);
    int32_t pos = A.start[code];

    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6, lane = tid & 63;
    const int lx = (int) ((wave & 1) * 4 + (lane & 3));
    const int ly = (int) (((wave >> 1) & 1) * 4 + ((lane >> 2) & 3));
    const int lz = (int) ((wave >> 2) * 4 + (lane >> 4));

    float f = __int_as_float(0x7FC00000);
    if (pos >= 0)       /* uniform over the workgroup */
    {
        const float cx = (float) (wx + lx + A.ox), cy = (float) (wy + ly + A.oy), cz = (float) (wz + lz + A.oz);
        const float bx0 = (float) (wx + A.ox), by0 = (float) (wy + A.oy), bz0 = (float) (wz + A.oz);
        /* corner coordinates of the wave's sub-block origin, for the cube tests */
        const float sbx = bx0 + (float) ((wave & 1) * 4), sby = by0 + (float) (((wave >> 1) & 1) * 4), sbz = bz0 + (float) ((wave >> 2) * 4);
        Fit fit;
        fitInit(fit);
        unsigned long long nListed = 0, nTests = 0;
        /* instrumented build: how evenly the hits of a drain call are spread over the wave's lanes, and what merging drain
         * calls would make of it (the iterations of a drain = the LONGEST of 64 lists) */
        uint32_t drainCalls = 0, sumMost = 0, sumMostPairs = 0, sumMostRound = 0, prevCnt = 0, roundCnt = 0;
        bool havePrev = false;
        typedef __attribute__((address_space(3))) uint16_t LdsSlot;
        LdsSlot *const mySlots = (LdsSlot *) sSlot[wave];
        const uint32_t myCube = ((lane >> 1) & 1u) | (((lane >> 3) & 1u) << 1) | (((lane >> 5) & 1u) << 2);
        LdsSlot *const myList = (LdsSlot *) sList[wave][myCube];
        f32x2 sWpxy = {0.0f, 0.0f}, sWnxy = {0.0f, 0.0f};
        const f32x2 cxy = {cx, cy};
        uint32_t nt = 0;            /* relevant splats in the window (wave-uniform) */

        /* every list entry that is ever read is a valid offset: zero until written */
#pragma unroll
        for (int c = 0; c < 8; c++)
            sList[wave][c][lane] = 0;

        /* accumulate this lane's hits of one chunk of iterations, in list order */
        auto drain = [&](uint32_t cur, LdsSlot *chunk)
        {
            const uint32_t cnt = (uint32_t) __popc(cur);
            fit.hits += cnt;
            if (STATS)
            {
                drainCalls++;
                sumMost += waveMax(cnt);
                atomicAdd(&sHist[cnt], 1u);
                roundCnt += cnt;
                if (havePrev)
                    sumMostPairs += waveMax(prevCnt + cnt);
                else
                    prevCnt = cnt;
                havePrev = !havePrev;
            }
            /* a lane leaves the loop with its last hit; the wave runs as many iterations as its longest list (no wave-wide
             * maximum is computed for a counted loop: 577 -> 572.5 us per launch) */
            while (cur != 0)
            {
                {
                    const uint32_t t = (uint32_t) __builtin_clz(cur);
                    cur ^= 0x80000000u >> t;
                    const uint32_t off = chunk[t];
                    const float4 pr = *(const float4 *) ((const char *) sPosRad + off);
                    const float4 nq = *(const float4 *) ((const char *) sNormQ + off);
                    const f32x2 pxy = f32x2{pr.x, pr.y} - cxy;
                    const float pz = pr.z - cz;
                    const float pp = fmaf(pxy.x, pxy.x, fmaf(pxy.y, pxy.y, pz * pz));
                    const float d = pp * pr.w;
                    float w = 1.0f - d;
                    w *= w;
                    w *= w;
                    w *= nq.w;
                    const f32x2 ww = {w, w};
                    const f32x2 nxy = {nq.x, nq.y};
                    const f32x2 wnxy = ww * nxy;
                    const float wnz = w * nq.z;
                    fit.sumW = fit.sumW + w;
                    sWpxy = __builtin_elementwise_fma(ww, pxy, sWpxy);
                    fit.sumWpz = fmaf(w, pz, fit.sumWpz);
                    sWnxy = __builtin_elementwise_fma(ww, nxy, sWnxy);
                    fit.sumWnz = fmaf(w, nq.z, fit.sumWnz);
                    fit.sumWpp = fmaf(w, pp, fit.sumWpp);
                    fit.sumWpn = fit.sumWpn + fmaf(wnxy.x, pxy.x, fmaf(wnxy.y, pxy.y, wnz * pz));
                }
            }
        };

        /* the window's relevant splats: cube lists, tests, accumulation */
        auto processWindow = [&](const uint32_t nt)
        {
            /* (the table was written by other lanes of this wave: LDS operations of a wave execute in order) */
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            /* 1. one lane per relevant splat: which of the eight cubes can its support reach? */
            uint32_t fine = 0, off = 0;
            if (lane < nt)
            {
                off = mySlots[lane];
                const float4 pr = *(const float4 *) ((const char *) sPosRad + off);
                float d[3][2];
                const float p[3] = {pr.x, pr.y, pr.z};
                const float b[3] = {sbx, sby, sbz};
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int h = 0; h < 2; h++)
                    {
                        const float lo = b[a] + (float) (2 * h), hi = b[a] + (float) (2 * h + 1);
                        d[a][h] = fmaxf(fmaxf(lo - p[a], p[a] - hi), 0.0f);
                    }
#pragma unroll
                for (int c = 7; c >= 0; c--)
                {
                    const float dx = d[0][c & 1], dy = d[1][(c >> 1) & 1], dz = d[2][c >> 2];
                    const float dd = dot3(dx, dy, dz, dx, dy, dz) * pr.w;
                    fine = __builtin_amdgcn_alignbit(fine, __float_as_uint(dd - RADIUS_CUTOFF), 31);   /* hitBit(dd) */
                }
                if (splatNeverCulled(pr))
                    fine = 0xFFu;
            }
            /* 2. the cubes' lists, in window order */
            uint32_t myCnt = 0, maxCnt = 0;
#pragma unroll
            for (uint32_t c = 0; c < 8; c++)
            {
                const uint64_t bal = __ballot((fine >> c) & 1u);
                const uint32_t n = (uint32_t) __popcll(bal);
                if ((fine >> c) & 1u)
                    sList[wave][c][popcBelow(bal)] = (uint16_t) off;
                myCnt = myCube == c ? n : myCnt;
                maxCnt = n > maxCnt ? n : maxCnt;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            /* 3. chunks of up to 32 iterations: lane l tests the k-th splat of its own cube's list */
            for (uint32_t k0 = 0; k0 < maxCnt; k0 += 32)
            {
                const uint32_t K = maxCnt - k0 < 32u ? maxCnt - k0 : 32u;
                LdsSlot *const chunk = myList + k0;
                uint32_t acc = 0;
#pragma unroll 4
                for (uint32_t k = 0; k < K; k++)
                {
                    const uint32_t o = chunk[k];
                    const float4 a = *(const float4 *) ((const char *) sPosRad + o);
                    const f32x2 pxy = f32x2{a.x, a.y} - cxy;
                    const float pz = a.z - cz;
                    const float pp = fmaf(pxy.x, pxy.x, fmaf(pxy.y, pxy.y, pz * pz));
                    const float d = pp * a.w;
                    acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(d - RADIUS_CUTOFF), 31);     /* acc = 2 acc + hitBit(d) */
                }
                if (STATS)
                    nTests += 64ull * K;
                /* iteration k sits in bit K - 1 - k; the lane's list ends after v of the K iterations (what it read beyond
                 * is somebody's older entry): drop those bits and left-align, so that the leading-zero count is k */
                const uint32_t v = myCnt > k0 ? (myCnt - k0 < K ? myCnt - k0 : K) : 0u;
                const uint32_t cur = v == 0 ? 0u : (acc >> (K - v)) << (32u - v);
                drain(cur, chunk);
            }
        };

        int32_t end = A.commands[pos++];
        /* a round's splat ids are requested while the round before it is processed */
        int32_t idAhead = pos + (int32_t) tid < end ? A.commands[pos + (int32_t) tid] : -1;
        while (pos < end)
        {
#pragma unroll
            for (int part = 0; part < 2; part++)
            {
                const uint32_t slot = tid + 512u * (uint32_t) part;
                if (slot >= CUBE_STAGE)
                    continue;
                uint32_t mask = 0;
                const int32_t lpos = pos + (int32_t) slot;
                const int32_t mine = part == 0 ? idAhead : (lpos < end ? A.commands[lpos] : -1);
                if (mine >= 0)
                {
                    const float4 pr = stagedPosRad(A, mine);
                    const float4 nq = A.splats[2 * (int64_t) mine + 1];
                    sPosRad[slot] = pr;
                    sNormQ[slot] = nq;
                    float d[3][2];
                    const float p[3] = {pr.x, pr.y, pr.z};
                    const float b[3] = {bx0, by0, bz0};
#pragma unroll
                    for (int a = 0; a < 3; a++)
#pragma unroll
                        for (int h = 0; h < 2; h++)
                        {
                            const float lo = b[a] + (float) (4 * h), hi = b[a] + (float) (4 * h + 3);
                            d[a][h] = fmaxf(fmaxf(lo - p[a], p[a] - hi), 0.0f);
                        }
#pragma unroll
                    for (int s_ = 7; s_ >= 0; s_--)
                    {
                        const float dx = d[0][s_ & 1], dy = d[1][(s_ >> 1) & 1], dz = d[2][s_ >> 2];
                        const float dd = dot3(dx, dy, dz, dx, dy, dz) * pr.w;
                        mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(dd - RADIUS_CUTOFF), 31);   /* hitBit(dd) */
                    }
                    if (splatNeverCulled(pr))
                        mask = 0xFFu;
                    if (splatIsNan(pr))
                        mask = 0;       /* no corner accepts it (see hitBit): it never enters a window */
                }
                sMask[slot] = (uint8_t) mask;
                if (STATS)
                    nListed += __popcll(__ballot(mine >= 0));
            }
            const int32_t staged = min(end - pos, (int32_t) CUBE_STAGE);
            pos += CUBE_STAGE;
            if (pos >= end)
            {
                pos = A.commands[end];
                end = (pos >= 0) ? A.commands[pos++] : INT32_MIN;
            }
            idAhead = pos + (int32_t) tid < end ? A.commands[pos + (int32_t) tid] : -1;
            __syncthreads();

            for (int32_t g = 0; g < staged; g += 64)
            {
                const uint32_t m = sMask[g + lane];
                const uint64_t todo = __ballot((m >> wave) & 1u);
                const uint32_t nGroup = (uint32_t) __popcll(todo);
                if (nGroup == 0)
                    continue;
                if ((todo >> lane) & 1ull)
                    mySlots[nt + popcBelow(todo)] = (uint16_t) ((g + (int32_t) lane) * (int32_t) sizeof(float4));
                nt += nGroup;
                if (nt >= 64u)
                {
                    /* always a FULL window: the first 64 entries now, the rest of this group moves to the front */
                    processWindow(64u);
                    nt -= 64u;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const uint16_t carried = lane < nt ? mySlots[64u + lane] : (uint16_t) 0;
                    __builtin_amdgcn_wave_barrier();
                    if (lane < nt)
                        mySlots[lane] = carried;
                }
            }
            if (nt != 0)
                processWindow(nt);          /* the offsets point into this round's staging buffers */
            nt = 0;
            if (STATS)
            {
                if (havePrev)
                    sumMostPairs += waveMax(prevCnt);       /* an odd call out: it stays alone */
                havePrev = false;
                sumMostRound += waveMax(roundCnt);
                roundCnt = 0;
            }
            __syncthreads();
        }
        fit.sumWpx = sWpxy.x;
        fit.sumWpy = sWpxy.y;
        fit.sumWnx = sWnxy.x;
        fit.sumWny = sWnxy.y;
        f = finishCorner<SHAPE>(fit, A.boundaryFactor);
        if (STATS)
        {
            const unsigned long long hits = waveSum(fit.hits);
            const uint32_t mostBlock = waveMax(fit.hits);
            if (lane == 0)
            {
                atomicAdd(&A.stats[0], nListed);
                atomicAdd(&A.stats[1], nTests);
                atomicAdd(&A.stats[2], hits);
                atomicAdd(&A.stats[3], (unsigned long long) drainCalls);
                atomicAdd(&A.stats[4], (unsigned long long) sumMost);
                atomicAdd(&A.stats[5], (unsigned long long) sumMostPairs);
                atomicAdd(&A.stats[6], (unsigned long long) sumMostRound);
                atomicAdd(&A.stats[7], (unsigned long long) mostBlock);
            }
            __syncthreads();
            if (tid < 33 && sHist[tid] != 0)
                atomicAdd(&A.stats[8 + tid], (unsigned long long) sHist[tid]);
        }
    }

    const int64_t row = (int64_t) (wy + ly) + (int64_t) (wz + lz) * A.zStride + A.zBias;
    A.field[row * (int64_t) A.pitch + (wx + lx)] = f;
}

/*
 * Variant 5 ("matrix prefilter").  Variant 4 is bound by issuing vector instructions, and close to half of them only FIND
 * hits (the per-cube cull and the distance tests); the matrix pipe is idle.  The distance test is a contraction in
 * disguise: with a = splat position - block origin and c' in {0..7}^3 the corner's coordinates inside the block,
 *     d - cutoff = invr2 * (|c'|^2 - 2 a.c' + |a|^2) - 0.99
 * is the dot product of a per-corner row [c'x, c'y, c'z, |c'|^2, 1] (small integers: exact in bf16) with a per-splat
 * column [-2 invr2 ax, -2 invr2 ay, -2 invr2 az, invr2, invr2 |a|^2 - 0.99 - margin], each f32 of which is the exact sum
 * of three bf16 pieces (8 + 8 + 8 significand bits): 15 of the 16 k-slots of ONE v_mfma_f32_32x32x16_bf16, every product
 * exact, f32 accumulation.  One lane per staged splat prepares the column at staging (beside the sub-block masks, as
 * before); a wave compacts a ROUND's relevant splats into its slot table and walks them in tiles of 32: the splats are the
 * rows (A), the sub-block's 64 corners the columns of two MFMAs (B: lane constants), a lane receives 16 + 16 results
 * for its column and packs their sign bits with 32 v_alignbit; one v_permlane32_swap brings the two halves of a corner's 32
 * bits to the lane that owns the corner.  The rows of a tile are dealt so that splat s of the tile ends up in bit 31 - s:
 * the drain walks the mask from the top, in list order, as in variant 4.
 *
 * The prefilter only has to be a SUPERSET of the reference's hits: the drain recomputes d exactly as the reference does
 * and accumulates under `d < cutoff` (kernels/mls.cl:371-373), so results are bit-identical by construction.  Superset:
 * every term of the contraction is bounded by T = |invr2| (|a|_1 + 21)^2 (|c'|_1 <= 21); rounding a, the five column
 * values, the reference's own d and the matrix unit's f32 accumulation (exact products, <= 16 roundings, each <= one ulp of a
 * partial sum <= T whether it rounds or truncates) move the difference by less than 40 x 2^-23 T; margin = 2^-16 T =
 * 128 x 2^-23 T.  (d near the cutoff needs T >= 0.99, so pieces lost to bf16 underflow, < 2^-126, do not count.)  A column
 * with T >= 1e30 or not finite (which includes 1/r^2 = -inf, never culled) is [0, 0, 0, 0, -1]: a candidate for every
 * corner.  A record with a NaN keeps its empty sub-block mask.  The instrumented build checks the superset on every tile
 * (stats word 42: exact hits the mask missed, must be 0; bench.py refuses to report otherwise).  False positives: under one
 * percent of the candidates.
 *
 * The kernel lives in 64 VGPRs and 40 KB of LDS (four workgroups per CU, as variant 4): the two MFMAs of a tile run one after
 * the other (16 accumulators at a time), wave-uniform floats are kept in scalar registers, the column is built in stages.
 * At 76 VGPRs / 41 KB (three workgroups per CU) the same code was 4 % SLOWER than variant 4, at 64 / 40 it is 9-10 % faster
 * (cfg3: 563-576 -> 510-516 us per launch of two buckets; shells cloud 497 -> 411): it executes 19 % fewer vector
 * instructions and 25 % fewer LDS instructions (profiles/r06_cfg3_processCorners_sq_counters.csv), the matrix pipe is 4 %
 * busy.  The drain requests the next candidate's records before it works on this one's (-1.5 %).
 *
 * Where a wave's time goes (s_memtime at the phase boundaries, -DMLSGPU_MLS5_CLOCK, tools/mls_clock.sh; cfg3 uniform): 15 %
 * before its first round (kernel arguments, start[], the list head, the ids: four dependent loads), 12 % staging, 6 % at the
 * barrier behind it, 9 % compaction, 43 % tiles and drains, 10 % at the round's last barrier.  Two things follow from it:
 * a round's records are requested BEFORE the wave waits for the others to leave the round before (the barrier that guards
 * the staged arrays sits between the loads and the LDS writes, and waits for LDS operations only), and a lane's eight
 * sub-block masks lie side by side so that the compaction reads them once (together -1.8 %).  -DMLSGPU_MLS5_DUMP leaves
 * the candidate masks of a sample of tiles behind the work counters: tools/drain_sim.py replays them under other ways of
 * walking the masks (profiles/NOTES_r06.md, section 10).
 */
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

/* the three bf16 pieces of an f32, most significant first, as the HIGH halves of three words (exact: see above) */
__device__ __forceinline__ void splitBf16(float x, uint32_t &hi, uint32_t &mid, uint32_t &lo)
{
    hi = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(hi);
    mid = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(mid);
    lo = __float_as_uint(r2) & 0xFFFF0000u;
}

/* words of two bf16: element 0 in the low half */
__device__ __forceinline__ uint32_t packHi(uint32_t e0, uint32_t e1)
{
    return __builtin_amdgcn_perm(e1, e0, 0x07060302u);     /* {e1[31:16], e0[31:16]} */
}

#define MATRIX_STAGE 512
#define MATRIX_SLOTS 440          /* + 4 entries of leading pad + 36 of slack = 480 per wave */
#ifndef MLSGPU_MLS5_WAVES
#define MLSGPU_MLS5_WAVES 8
#endif


template<int SHAPE, bool STATS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(MLSGPU_MLS5_WAVES, MLSGPU_MLS5_WAVES)))
void processCornersMatrixKernel(Lanes<MlsArgs> lanes)
{
    __shared__ float4 sPosRad[MATRIX_STAGE];
    __shared__ float4 sNormQ[MATRIX_STAGE];
    __shared__ uint4 sColLo[MATRIX_STAGE];      /* k = 0..7 of a splat's column: x pieces, y pieces, constant hi, mid */
    __shared__ uint4 sColHi[MATRIX_STAGE];      /* k = 8..15: z pieces, 1/r^2 pieces, constant lo, 0 */
    __shared__ uint8_t sMask[MATRIX_STAGE];
    __shared__ uint16_t sSlot[8][MATRIX_SLOTS + 40];  /* per wave: byte offsets of the round's relevant splats (flushed when
                                                       * full), a tile of slack; 40 KB in all: four workgroups per CU */
    __shared__ uint32_t sHist[STATS ? 33 : 1];

    const MlsArgs A = lanes.a[blockIdx.y];
    if (blockIdx.x >= A.numBlocks)
        return;
#ifdef MLSGPU_MLS5_CLOCK
    /* where a wave's time goes (an instrumented build for profiles/NOTES only: tools/mls_clock.sh); s_memtime at the points
     * where the wave waits for its LDS operations anyway */
    uint64_t clkMark = __builtin_amdgcn_s_memtime();
    const uint64_t clkStart = clkMark;
    uint64_t clkHead = 0, clkStage = 0, clkBar1 = 0, clkCompact = 0, clkTiles = 0, clkBar2 = 0;
#define MLS5_CLOCK(sum) do { uint64_t now_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) : : "memory"); sum += now_ - clkMark; clkMark = now_; } while (0)
#else
#define MLS5_CLOCK(sum) do { } while (0)
#endif
    if (STATS)
    {
        if (threadIdx.x < 33)
            sHist[threadIdx.x] = 0;
        __syncthreads();
    }
    uint32_t gx, gy, gz;
    if (A.superShift != 0)
    {
        /* xcdRemap and the three coordinates by shifts and two multiplications (see MlsArgs) */
        const uint32_t id = blockIdx.x, sh = A.superShift;
        const uint32_t j = id & ((1u << sh) - 1u);
        const uint32_t bid = id >= A.full ? id : (id >> sh << sh) + ((j & 7u) << (sh - 3u)) + (j >> 3);
        const uint32_t t = __umulhi(bid, A.magicX);
        gx = bid - t * A.blocksX;
        gz = __umulhi(t, A.magicY);
        gy = t - gz * A.blocksY;
    }
    else
    {
        const uint32_t bid = xcdRemap(blockIdx.x, A.numBlocks, A.xcdChunk);
        gx = bid % A.blocksX;
        gy = (bid / A.blocksX) % A.blocksY;
        gz = bid / (A.blocksX * A.blocksY);
    }
    const int wx = (int) (gx * 8), wy = (int) (gy * 8), wz = (int) (gz * 8 + A.zFirst);
    const uint32_t sub = A.startShift / 3;
    const uint32_t code = spread3((uint32_t) wx >> sub) | (spread3((uint32_t) wy >> sub) << 1) | (spread3((uint32_t) wz >> sub) << 2);
    int32_t pos = A.start[code];

    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6, lane = tid & 63;
    const int lx = (int) ((wave & 1) * 4 + (lane & 3));
    const int ly = (int) (((wave >> 1) & 1) * 4 + ((lane >> 2) & 3));
    const int lz = (int) ((wave >> 2) * 4 + (lane >> 4));

    float f = __int_as_float(0x7FC00000);
    if (pos >= 0)       /* uniform over the workgroup */
    {
        const float cx = (float) (wx + lx + A.ox), cy = (float) (wy + ly + A.oy), cz = (float) (wz + lz + A.oz);
        /* wave-uniform floats live in scalar registers (a float made by a vector instruction would be hoisted out of the
         * loops into a vector register each: twelve of the kernel's 64) */
        /* (the builtin is folded away for a value the compiler knows to be uniform.)  The wait states are part of the
         * statement: the compiler's hazard recogniser does not look into inline assembly, and a v_readfirstlane issued
         * right behind the conversion that writes its source READ THE OLD REGISTER on gfx950 -- one block of one test case,
         * found when a rearrangement of this kernel put the two back to back (profiles/NOTES_r06.md, section 1).  Four bounds
         * of an axis per statement: one pair of wait states for the four. */
        float bx0, by0, bz0;
        float boxLo[3][2], boxHi[3][2];
        auto axisBounds = [](int o, float &lo0, float &hi0, float &lo1, float &hi1)
        {
            asm("s_nop 3\n\tv_readfirstlane_b32 %0, %4\n\tv_readfirstlane_b32 %1, %5\n\tv_readfirstlane_b32 %2, %6\n\t"
                "v_readfirstlane_b32 %3, %7\n\ts_nop 1"
                : "=s"(lo0), "=s"(hi0), "=s"(lo1), "=s"(hi1)
                : "v"((float) o), "v"((float) (o + 3)), "v"((float) (o + 4)), "v"((float) (o + 7)));
        };
        axisBounds(wx + A.ox, boxLo[0][0], boxHi[0][0], boxLo[0][1], boxHi[0][1]);
        axisBounds(wy + A.oy, boxLo[1][0], boxHi[1][0], boxLo[1][1], boxHi[1][1]);
        axisBounds(wz + A.oz, boxLo[2][0], boxHi[2][0], boxLo[2][1], boxHi[2][1]);
        bx0 = boxLo[0][0];
        by0 = boxLo[1][0];
        bz0 = boxLo[2][0];
        Fit fit;
        fitInit(fit);
        unsigned long long nListed = 0, nTests = 0, nCand = 0, nMissed = 0;
        uint32_t drainCalls = 0, sumMost = 0, sumMostRound = 0, roundCnt = 0;
        typedef __attribute__((address_space(3))) uint16_t LdsSlot;
        LdsSlot *const mySlots = (LdsSlot *) sSlot[wave] + 4;
        f32x2 sWpxy = {0.0f, 0.0f}, sWnxy = {0.0f, 0.0f};
        const f32x2 cxy = {cx, cy};

        /* B operands: lane l = column n = l & 31 of both MFMAs, k = 8 (l >> 5) + j.  MFMA j covers the corners
         * n + 32 j of the sub-block (z layers 2 j, 2 j + 1); coordinates relative to the block origin */
        uint4 bFrag[2];
        {
            const uint32_t n = lane & 31u, h = lane >> 5;
#pragma unroll
            for (uint32_t j = 0; j < 2; j++)
            {
                const uint32_t q = n + 32u * j;
                const uint32_t qx = (wave & 1u) * 4u + (q & 3u), qy = ((wave >> 1) & 1u) * 4u + ((q >> 2) & 3u), qz = (wave >> 2) * 4u + (q >> 4);
                const uint32_t u = h ? qz : qx, v = h ? qx * qx + qy * qy + qz * qz : qy;
                const uint32_t ub = __float_as_uint((float) u), vb = __float_as_uint((float) v);     /* exact in bf16 */
                bFrag[j] = make_uint4(packHi(ub, ub), packHi(ub, vb), packHi(vb, vb), 0x3F803F80u);
            }
        }
        /* A operand: lane l holds row m = l & 31, which carries splat s(m) of the tile so that the sign bits line up in
         * list order: row 8 (r >> 2) + 4 half + (r & 3) <-> splat 16 half + r */
        const uint32_t rowM = lane & 31u;
        const uint32_t rowSplat = 16u * ((rowM >> 2) & 1u) + 4u * (rowM >> 3) + (rowM & 3u);
        const char *const colBase = (lane >> 5) ? (const char *) sColHi : (const char *) sColLo;

        /* every slot that is ever read is a valid offset: zero until written (960 bytes per wave: one 16-byte store of 60 lanes) */
        static_assert((MATRIX_SLOTS + 40) * sizeof(uint16_t) % 16 == 0 && (MATRIX_SLOTS + 40) * sizeof(uint16_t) <= 64 * 16, "one store per lane");
        if (lane < (MATRIX_SLOTS + 40) * sizeof(uint16_t) / 16)
            reinterpret_cast<uint4 *>(sSlot[wave])[lane] = make_uint4(0u, 0u, 0u, 0u);

        /* one candidate: the reference's test and sums (kernels/mls.cl:362-390) */
        auto accumulate = [&](const float4 pr, const float4 nq)
        {
            const f32x2 pxy = f32x2{pr.x, pr.y} - cxy;
            const float pz = pr.z - cz;
            const float pp = fmaf(pxy.x, pxy.x, fmaf(pxy.y, pxy.y, pz * pz));
            const float d = pp * pr.w;
            if (d < RADIUS_CUTOFF)
            {
                float w = 1.0f - d;
                w *= w;
                w *= w;
                w *= nq.w;
                const f32x2 ww = {w, w};
                const f32x2 nxy = {nq.x, nq.y};
                const f32x2 wnxy = ww * nxy;
                const float wnz = w * nq.z;
                fit.sumW = fit.sumW + w;
                sWpxy = __builtin_elementwise_fma(ww, pxy, sWpxy);
                fit.sumWpz = fmaf(w, pz, fit.sumWpz);
                sWnxy = __builtin_elementwise_fma(ww, nxy, sWnxy);
                fit.sumWnz = fmaf(w, nq.z, fit.sumWnz);
                fit.sumWpp = fmaf(w, pp, fit.sumWpp);
                fit.sumWpn = fit.sumWpn + fmaf(wnxy.x, pxy.x, fmaf(wnxy.y, pxy.y, wnz * pz));
                fit.hits++;
            }
        };
        /* position of the highest set bit from the top; -1 for 0 (the hardware's answer, which __builtin_clz does not promise) */
        auto ffbh = [](uint32_t v)
        {
            uint32_t t;
            asm("v_ffbh_u32 %0, %1" : "=v"(t) : "v"(v));
            return t;
        };
        auto posRadAt = [&](uint32_t off) { return *(const float4 *) ((const char *) sPosRad + off); };
        auto normQAt = [&](uint32_t off) { return *(const float4 *) ((const char *) sNormQ + off); };

        /* accumulate this lane's candidates of one tile, in list order, under the reference's own test */
        auto drain = [&](uint32_t cur, LdsSlot *chunk)
        {
            if (STATS)
            {
                const uint32_t cnt = (uint32_t) __popc(cur);
                drainCalls++;
                sumMost += waveMax(cnt);
                atomicAdd(&sHist[cnt], 1u);
                roundCnt += cnt;
                nCand += waveSum(cnt);
            }
            /* the next candidate's record is requested before this one is worked on (two copies of the body: the records
             * change places without register moves).  A mask that has run out reads the entry before the tile's first: a
             * valid offset (the table's leading pad, or the tile before) */
            if (cur != 0)
            {
                uint32_t t = ffbh(cur);
                uint32_t off = chunk[t];
                float4 prA = posRadAt(off), nqA = normQAt(off), prB, nqB;
                for (;;)
                {
                    cur ^= 0x80000000u >> t;
                    t = ffbh(cur);
                    off = chunk[(int32_t) t];
                    prB = posRadAt(off);
                    nqB = normQAt(off);
                    accumulate(prA, nqA);
                    if (cur == 0)
                        break;
                    cur ^= 0x80000000u >> t;
                    t = ffbh(cur);
                    off = chunk[(int32_t) t];
                    prA = posRadAt(off);
                    nqA = normQAt(off);
                    accumulate(prB, nqB);
                    if (cur == 0)
                        break;
                }
            }
        };

        int32_t end = A.commands[pos++];
        /* a round's splat ids are requested while the round before it is processed */
        int32_t idAhead = pos + (int32_t) tid < end ? A.commands[pos + (int32_t) tid] : -1;
        MLS5_CLOCK(clkHead);
        bool laterRound = false;
        while (pos < end)
        {
            {
                uint32_t mask = 0;
                const int32_t mine = idAhead;
                /* the round's records are requested BEFORE the wave waits for the others to be done with the round before:
                 * the barrier only guards the staged arrays */
                float4 pr = {0.0f, 0.0f, 0.0f, 0.0f}, nq = {0.0f, 0.0f, 0.0f, 0.0f};
                if (mine >= 0)
                {
                    pr = A.splats[2 * (int64_t) mine];
                    nq = A.splats[2 * (int64_t) mine + 1];
                }
                if (laterRound)
                {
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");
                    MLS5_CLOCK(clkBar2);
                }
                laterRound = true;
                if (mine >= 0)
                {
                    if (A.rawRadius)
                        pr.w = 1.0f / (pr.w * pr.w);
                    sPosRad[tid] = pr;
                    sNormQ[tid] = nq;
                    float d[3][2];
                    const float p[3] = {pr.x, pr.y, pr.z};
#pragma unroll
                    for (int a = 0; a < 3; a++)
#pragma unroll
                        for (int h = 0; h < 2; h++)
                            d[a][h] = fmaxf(fmaxf(boxLo[a][h] - p[a], p[a] - boxHi[a][h]), 0.0f);
#pragma unroll
                    for (int s_ = 7; s_ >= 0; s_--)
                    {
                        const float dx = d[0][s_ & 1], dy = d[1][(s_ >> 1) & 1], dz = d[2][s_ >> 2];
                        const float dd = dot3(dx, dy, dz, dx, dy, dz) * pr.w;
                        mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(dd - RADIUS_CUTOFF), 31);   /* hitBit(dd) */
                    }
                    if (splatNeverCulled(pr))
                        mask = 0xFFu;
                    if (splatIsNan(pr))
                        mask = 0;       /* no corner accepts it: it never enters a tile */

                    /* the splat's column of the contraction (in stages: the kernel lives in 64 registers) */
                    __builtin_amdgcn_sched_barrier(0);
                    const float ax = pr.x - bx0, ay = pr.y - by0, az = pr.z - bz0;
                    const float s1 = fabsf(ax) + fabsf(ay) + fabsf(az) + 21.0f;
                    const float T = fabsf(pr.w) * (s1 * s1);
                    const bool plain = T < 1e30f;                   /* false for NaN and infinities too */
                    const float m2 = plain ? -2.0f * pr.w : 0.0f;
                    const float gc = plain ? fmaf(pr.w, dot3(ax, ay, az, ax, ay, az), -RADIUS_CUTOFF) - T * 0x1p-16f : -1.0f;
                    uint32_t c0, c1, c2;
                    splitBf16(gc, c0, c1, c2);
                    {
                        uint32_t x0, x1, x2, y0, y1, y2;
                        splitBf16(m2 * ax, x0, x1, x2);
                        splitBf16(m2 * ay, y0, y1, y2);
                        sColLo[tid] = make_uint4(packHi(x0, x1), packHi(x2, y0), packHi(y1, y2), packHi(c0, c1));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        uint32_t z0, z1, z2, q0, q1, q2;
                        splitBf16(m2 * az, z0, z1, z2);
                        splitBf16(plain ? pr.w : 0.0f, q0, q1, q2);
                        sColHi[tid] = make_uint4(packHi(z0, z1), packHi(z2, q0), packHi(q1, q2), c2 >> 16);
                    }
                }
                sMask[lane * 8 + wave] = (uint8_t) mask;        /* a lane's eight groups side by side: one read in the compaction */
                if (STATS)
                    nListed += __popcll(__ballot(mine >= 0));
            }
            const int32_t staged = min(end - pos, (int32_t) MATRIX_STAGE);
            pos += MATRIX_STAGE;
            if (pos >= end)
            {
                pos = A.commands[end];
                end = (pos >= 0) ? A.commands[pos++] : INT32_MIN;
            }
            idAhead = pos + (int32_t) tid < end ? A.commands[pos + (int32_t) tid] : -1;
            MLS5_CLOCK(clkStage);
            __syncthreads();
            MLS5_CLOCK(clkBar1);

            /* the round's relevant splats of this wave's sub-block, in list order; a table that cannot take the next group of
             * 64 is worked off first (no cloud of the BASELINE configs gets there) */
            int32_t g = 0;
            do
            {
            uint32_t nt = 0;
            const uint64_t mm = *(const uint64_t *) (sMask + lane * 8);
            for (; g < staged; g += 64)
            {
                const uint32_t m = (uint32_t) (mm >> ((uint32_t) g >> 3));       /* byte g / 64 */
                const uint64_t todo = __ballot((m >> wave) & 1u);
                const uint32_t n = (uint32_t) __popcll(todo);
                if (nt + n > MATRIX_SLOTS)
                    break;
                if ((m >> wave) & 1u)
                    mySlots[nt + popcBelow(todo)] = (uint16_t) ((g + (int32_t) lane) * (int32_t) sizeof(float4));
                nt += n;
            }
            /* (the table was written by other lanes of this wave: LDS operations of a wave execute in order) */
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            MLS5_CLOCK(clkCompact);
            /* the tiles and drains issue ahead of the waves that stage, compact or wait: the vector work is the long pole, and
             * a wave kept waiting in it holds its workgroup's next barrier back (-1 % on the uniform cloud, -1.5 % on the shells
             * cloud; the other way round -- the waiting phases first -- costs 4 %) */
            __builtin_amdgcn_s_setprio(1);

            LdsSlot *tile = mySlots;        /* the tile's first slot; this lane's row reads tile[rowSplat] */
            for (uint32_t t0 = 0; t0 < nt; t0 += 32, tile += 32)
            {
                const uint32_t v = nt - t0 < 32u ? nt - t0 : 32u;
                const uint32_t off = tile[rowSplat];
                const uint4 aw = *(const uint4 *) (colBase + off);
                const bf16x8 aFrag = __builtin_bit_cast(bf16x8, aw);
                const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                /* one MFMA's 16 results at a time: 64 VGPRs, 8 waves per SIMD */
                uint32_t x0 = 0, x1 = 0;
                {
                    const f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aFrag, __builtin_bit_cast(bf16x8, bFrag[0]), zero, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        x0 = __builtin_amdgcn_alignbit(x0, __float_as_uint(acc[r]), 31);
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    const f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aFrag, __builtin_bit_cast(bf16x8, bFrag[1]), zero, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        x1 = __builtin_amdgcn_alignbit(x1, __float_as_uint(acc[r]), 31);
                }
                /* lanes 32-63 of x0 (corner l - 32, splats 16-31) <-> lanes 0-31 of x1 (corner l + 32, splats 0-15):
                 * afterwards x0 = this lane's corner against splats 0-15, x1 = against splats 16-31 */
                const auto sw = __builtin_amdgcn_permlane32_swap(x0, x1, false, false);
                uint32_t cur = (sw[0] << 16) | sw[1];
                cur &= 0xFFFFFFFFu << (32u - v);          /* rows beyond the tile's end hold somebody's older splat */
                if (STATS)
                {
                    nTests += 64ull * 32ull;
                    /* the superset check: the reference's test on every (corner, splat) of the tile */
                    uint32_t exact = 0;
                    for (uint32_t s = 0; s < v; s++)
                    {
                        const float4 pr = *(const float4 *) ((const char *) sPosRad + tile[s]);
                        const float px = pr.x - cx, py = pr.y - cy, pz = pr.z - cz;
                        const float d = dot3(px, py, pz, px, py, pz) * pr.w;
                        exact |= d < RADIUS_CUTOFF ? 0x80000000u >> s : 0u;
                    }
                    nMissed += waveSum((uint32_t) __popc(exact & ~cur));
                }
#ifdef MLSGPU_MLS5_DUMP
                /* a sample of the drain's input for offline what-ifs (tools/drain_sim.py): per tile a header word and the 64
                 * lanes' candidate masks, behind a word count in word 0 */
                if (A.stats != nullptr && blockIdx.x % 499 == 0)
                {
                    unsigned long long at = 0;
                    if (lane == 0)
                        at = atomicAdd(&A.stats[0], 65ull);
                    at = __shfl(at, 0);
                    if (at + 65 < MLSGPU_MLS5_DUMP)
                    {
                        if (lane == 0)
                            A.stats[1 + at] = ((unsigned long long) blockIdx.y << 56) | ((unsigned long long) blockIdx.x << 24) | (wave << 16) | t0;
                        A.stats[2 + at + lane] = cur;
                    }
                }
#endif
                drain(cur, tile);
            }
            MLS5_CLOCK(clkTiles);
            } while (g < staged);
            __builtin_amdgcn_s_setprio(0);
            if (STATS)
            {
                sumMostRound += waveMax(roundCnt);
                roundCnt = 0;
            }
        }
        fit.sumWpx = sWpxy.x;
        fit.sumWpy = sWpxy.y;
        fit.sumWnx = sWnxy.x;
        fit.sumWny = sWnxy.y;
        f = finishCorner<SHAPE>(fit, A.boundaryFactor);
#ifdef MLSGPU_MLS5_CLOCK
        if (A.stats != nullptr && lane == 0 && blockIdx.x % 61 == 0)     /* a sample: 18 M atomics on eight words would be the measurement */
        {
            /* words 0-5: cycles of a wave before its first round / staging (its loads included) / at the barrier behind it /
             * compaction / tiles and drains / at the round's last barrier; 6: from the first instruction to here; 7: waves */
            atomicAdd(&A.stats[0], (unsigned long long) clkHead);
            atomicAdd(&A.stats[1], (unsigned long long) clkStage);
            atomicAdd(&A.stats[2], (unsigned long long) clkBar1);
            atomicAdd(&A.stats[3], (unsigned long long) clkCompact);
            atomicAdd(&A.stats[4], (unsigned long long) clkTiles);
            atomicAdd(&A.stats[5], (unsigned long long) clkBar2);
            atomicAdd(&A.stats[6], (unsigned long long) (__builtin_amdgcn_s_memtime() - clkStart));
            atomicAdd(&A.stats[7], 1ull);
        }
#endif
        if (STATS)
        {
            const unsigned long long hits = waveSum(fit.hits);
            const uint32_t mostBlock = waveMax(fit.hits);
            if (lane == 0)
            {
                atomicAdd(&A.stats[0], nListed);
                atomicAdd(&A.stats[1], nTests);
                atomicAdd(&A.stats[2], hits);
                atomicAdd(&A.stats[3], (unsigned long long) drainCalls);
                atomicAdd(&A.stats[4], (unsigned long long) sumMost);
                atomicAdd(&A.stats[6], (unsigned long long) sumMostRound);
                atomicAdd(&A.stats[7], (unsigned long long) mostBlock);
                atomicAdd(&A.stats[41], nCand);                         /* candidates the prefilter passed to the drain */
                atomicAdd(&A.stats[42], nMissed);                       /* hits it missed: must stay 0 */
            }
            __syncthreads();
            if (tid < 33 && sHist[tid] != 0)
                atomicAdd(&A.stats[8 + tid], (unsigned long long) sHist[tid]);
        }
    }

    const int64_t row = (int64_t) (wy + ly) + (int64_t) (wz + lz) * A.zStride + A.zBias;
    A.field[row * (int64_t) A.pitch + (wx + lx)] = f;
}

} // namespace

/* ------------------------------------------------------------------ C-ABI */

MLSGPU_API int mlsgpu_hip_mls_create(mlsgpu_ctx *ctx, int shape, mlsgpu_mls **out)
{
    REQUIRE(ctx != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(shape == MLSGPU_SHAPE_SPHERE || shape == MLSGPU_SHAPE_PLANE, MLSGPU_ERR_INVALID);
    mlsgpu_mls *m = new mlsgpu_mls;
    m->ctx = ctx;
    m->shape = shape;
    *out = m;
    return mlsgpu_hip_mls_set_boundary_limit(m, 1.0f);     /* src/mls.cpp:72 */
}

MLSGPU_API void mlsgpu_hip_mls_destroy(mlsgpu_mls *m) { delete m; }

MLSGPU_API int mlsgpu_hip_mls_set_buffers(mlsgpu_mls *m, const int32_t offset[3], const mlsgpu_splat *dSplats,
                                          const int32_t *dCommands, const int32_t *dStart, uint32_t subsamplingShift)
{
    REQUIRE(m != nullptr && offset != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(dSplats != nullptr && dCommands != nullptr && dStart != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(subsamplingShift >= 3 && subsamplingShift <= 10, MLSGPU_ERR_LENGTH);   /* subsamplingMin, src/mls.cpp:54 */
    m->dSplats = dSplats;
    m->dCommands = dCommands;
    m->dStart = dStart;
    m->startShift = 3 * subsamplingShift;                   /* src/mls.cpp:86 */
    for (int i = 0; i < 3; i++)
        m->offset[i] = offset[i];
    m->isSet = true;
    m->rawRadius = false;
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_mls_set_raw_radius(mlsgpu_mls *m, int rawRadius)
{
    REQUIRE(m != nullptr, MLSGPU_ERR_INVALID);
    m->rawRadius = rawRadius != 0;
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_mls_set(mlsgpu_mls *m, const int32_t offset[3], const mlsgpu_tree *tree, uint32_t subsamplingShift)
{
    REQUIRE(m != nullptr && tree != nullptr, MLSGPU_ERR_INVALID);
    PROPAGATE(mlsgpu_hip_mls_set_buffers(m, offset, mlsgpu_hip_tree_splats(tree), mlsgpu_hip_tree_commands(tree),
                                         mlsgpu_hip_tree_start(tree), subsamplingShift));
    m->rawRadius = mlsgpu_hip_tree_mutates(tree) == 0;
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_mls_set_boundary_limit(mlsgpu_mls *m, float limit)
{
    REQUIRE(m != nullptr, MLSGPU_ERR_INVALID);
    /* src/mls.cpp:137-144 */
    const float pi = 3.14159265358979323846f;
    const float boundaryScale = (sqrtf(6.0f) * 512) / (693 * pi);
    const float gamma = boundaryScale * limit;
    m->boundaryFactor = 1.0f - gamma * gamma;
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_mls_set_variant(mlsgpu_mls *m, int variant)
{
    /* 5 = matrix prefilter (default), 4 = cube streams, 1 = the reference's structure; 0, 2 and 3 were intermediate designs of rounds 1-3 */
    REQUIRE(m != nullptr && (variant == 1 || variant == 4 || variant == 5), MLSGPU_ERR_INVALID);
    m->variant = variant;
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_mls_set_stats(mlsgpu_mls *m, uint64_t *dCounters)
{
    REQUIRE(m != nullptr, MLSGPU_ERR_INVALID);
    m->dStats = reinterpret_cast<unsigned long long *>(dCounters);
    return MLSGPU_OK;
}

/* MlsFunctor::enqueue's checks (src/mls.cpp:108-116) and one lane's kernel arguments */
static int mlsLaneArgs(mlsgpu_mls *m, float *dField, uint64_t pitch, uint64_t fieldRows, const mlsgpu_swathe *sw, MlsArgs *out)
{
    REQUIRE(m != nullptr && dField != nullptr && sw != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(m->isSet, MLSGPU_ERR_INVALID);
    /* src/mls.cpp:108-116 */
    const uint32_t width = roundUp(sw->width, 8), height = roundUp(sw->height, 8);
    REQUIRE(sw->width > 0 && sw->height > 0, MLSGPU_ERR_INVALID);
    REQUIRE(sw->zStride >= height, MLSGPU_ERR_INVALID);
    REQUIRE(sw->zFirst <= sw->zLast, MLSGPU_ERR_INVALID);
    REQUIRE(sw->zFirst % 8 == 0, MLSGPU_ERR_INVALID);
    REQUIRE(pitch >= width, MLSGPU_ERR_LENGTH);
    const uint32_t blocksZ = divUp((uint64_t) sw->zLast - sw->zFirst + 1, 8);
    /* the kernel writes whole 8x8x8 blocks (kernels/mls.cl:429-432): every row it touches must exist */
    const int64_t firstRow = (int64_t) sw->zFirst * sw->zStride + sw->zBias;
    const int64_t lastRow = (int64_t) (sw->zFirst + 8 * blocksZ - 1) * sw->zStride + sw->zBias + height - 1;
    REQUIRE(firstRow >= 0 && (uint64_t) lastRow < fieldRows, MLSGPU_ERR_LENGTH);
    const uint32_t sub = m->startShift / 3;
    REQUIRE(((sw->zFirst + 8 * blocksZ - 1) >> sub) < 1024 && ((width - 1) >> sub) < 1024 && ((height - 1) >> sub) < 1024,
            MLSGPU_ERR_LENGTH);
    MlsArgs A;
    A.field = dField;
    A.pitch = pitch;
    A.splats = reinterpret_cast<const float4 *>(m->dSplats);
    A.commands = m->dCommands;
    A.start = m->dStart;
    A.startShift = m->startShift;
    A.ox = m->offset[0]; A.oy = m->offset[1]; A.oz = m->offset[2];
    A.zStride = sw->zStride;
    A.zBias = sw->zBias;
    A.zFirst = sw->zFirst;
    A.blocksX = width / 8;
    A.blocksY = height / 8;
    A.blocksZ = blocksZ;
    A.numBlocks = A.blocksX * A.blocksY * A.blocksZ;
    A.boundaryFactor = m->boundaryFactor;
    /* measured on cfg3 (ms per step shells / noise cloud, HBM fetch per launch on the noise cloud):
     *   one contiguous eighth per XCD 12.5 / 10.6, 424 MB;  runs of 4: 9.5 / 10.5, 700 MB;  16: 9.5 / 10.5, 624 MB;
     *   64: 10.2 / 10.5, 402 MB;  128: 10.1 / 10.5.  Long runs keep a hot list in ONE L2, which then limits the heavy
     *   blocks of surface-like data; short runs re-fetch it into several L2s (more HBM traffic, more L2 bandwidth).
     *   The kernel is not HBM-bound, so the faster setting wins. */
    static const uint32_t xcdChunk = getenv("MLSGPU_HIP_MLS_XCD_CHUNK") ? (uint32_t) atoi(getenv("MLSGPU_HIP_MLS_XCD_CHUNK")) : 16u;
    A.xcdChunk = xcdChunk;
    A.superShift = A.full = A.magicX = A.magicY = 0;
    if (xcdChunk != 0 && (xcdChunk & (xcdChunk - 1)) == 0 && xcdChunk <= (1u << 20) && A.blocksX >= 2 && A.blocksY >= 2)
    {
        /* q = umulhi(n, ceil(2^32 / d)) is floor(n / d) when n * (ceil(2^32 / d) * d - 2^32) < 2^32: with n = q d + r the
         * product is q + r / d + n e / (d 2^32), and r <= d - 1 */
        const uint64_t two32 = uint64_t(1) << 32;
        const uint64_t mx = (two32 + A.blocksX - 1) / A.blocksX, my = (two32 + A.blocksY - 1) / A.blocksY;
        const uint64_t nx = A.numBlocks - 1, ny = nx / A.blocksX;
        if (nx * (mx * A.blocksX - two32) < two32 && ny * (my * A.blocksY - two32) < two32)
        {
            uint32_t shift = 3;
            while ((1u << (shift - 3)) < xcdChunk)
                shift++;
            A.superShift = shift;
            A.full = A.numBlocks >> shift << shift;
            A.magicX = (uint32_t) mx;
            A.magicY = (uint32_t) my;
        }
    }
    A.rawRadius = m->rawRadius ? 1u : 0u;
    A.stats = m->dStats;
    *out = A;
    return MLSGPU_OK;
}

/* MlsFunctor::enqueue for the buckets of a batch: one launch, blockIdx.y = bucket.  The functors share a context, the
 * shape, the kernel variant and whether work counters are collected (the first functor's settings select the kernel). */
static int mlsEnqueueLanes(mlsgpu_mls *const *ms, const MlsArgs *args, uint32_t count)
{
    mlsgpu_mls *const m = ms[0];
    mlsgpu_ctx *ctx = m->ctx;
    for (uint32_t k = 1; k < count; k++)
        REQUIRE(ms[k]->ctx == ctx && ms[k]->shape == m->shape && ms[k]->variant == m->variant
                && (ms[k]->dStats != nullptr) == (m->dStats != nullptr), MLSGPU_ERR_INVALID);
    HIP_CHECK(hipSetDevice(ctx->device));
    Lanes<MlsArgs> L;
    uint32_t maxBlocks = 0;
    for (uint32_t k = 0; k < MAX_LANES; k++)
    {
        L.a[k] = args[k < count ? k : 0];
        if (k < count)
            maxBlocks = std::max(maxBlocks, L.a[k].numBlocks);
    }
    const dim3 grid(maxBlocks, count), block(512);
    const char *stat = "kernel.mls.processCorners.time";      /* src/mls.cpp:57 */
    /* tuning aid: dynamic LDS that the kernel never touches lowers its occupancy (3 workgroups per CU from 8 KB, 2 from
     * 20 KB), leaving wave slots to the memory-bound kernels of the other device workers */
    static const uint32_t ldsPad = [] {
        /* clamped to what still launches beside the kernels' static LDS (36 KB at most) in the CU's 160 KB */
        const char *e = getenv("MLSGPU_HIP_MLS_LDS_PAD");
        const long v = e ? atol(e) : 0;
        return (uint32_t) (v < 0 ? 0 : (v > 120 * 1024 ? 120 * 1024 : v));
    }();
#if defined(MLSGPU_MLS5_CLOCK) || defined(MLSGPU_MLS5_DUMP)
    /* the clock build times the PLAIN kernel: it writes its cycle sums where the work counters would go */
    const bool sphere = m->shape == MLSGPU_SHAPE_SPHERE, stats = m->dStats != nullptr && m->variant != 5;
#else
    const bool sphere = m->shape == MLSGPU_SHAPE_SPHERE, stats = m->dStats != nullptr;    /* stats: an instrumented build,
                                                                                            * never in a timed run */
#endif
#define MLS_LAUNCH(KERNEL, SHAPE, STATS) LAUNCH_LDS(ctx, stat, (KERNEL<SHAPE, STATS>), grid, block, ldsPad, L)
#define MLS_LAUNCH_ANY(KERNEL)                                                                                   \
    do {                                                                                                         \
        if (stats) { if (sphere) MLS_LAUNCH(KERNEL, MLSGPU_SHAPE_SPHERE, true); else MLS_LAUNCH(KERNEL, MLSGPU_SHAPE_PLANE, true); } \
        else { if (sphere) MLS_LAUNCH(KERNEL, MLSGPU_SHAPE_SPHERE, false); else MLS_LAUNCH(KERNEL, MLSGPU_SHAPE_PLANE, false); }     \
    } while (0)
    if (m->variant == 5)
        MLS_LAUNCH_ANY(processCornersMatrixKernel);
    else if (m->variant == 4)
        MLS_LAUNCH_ANY(processCornersCubeKernel);
    else
        MLS_LAUNCH_ANY(processCornersKernel);
#undef MLS_LAUNCH_ANY
#undef MLS_LAUNCH
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_mls_enqueue(mlsgpu_mls *m, float *dField, uint64_t pitch, uint64_t fieldRows,
                                      const mlsgpu_swathe *sw)
{
    MlsArgs A;
    PROPAGATE(mlsLaneArgs(m, dField, pitch, fieldRows, sw, &A));
    return mlsEnqueueLanes(&m, &A, 1);
}

MLSGPU_API int mlsgpu_hip_mls_enqueue_batch(mlsgpu_mls *const *ms, float *const *dFields, const uint64_t *pitches,
                                            const uint64_t *fieldRows, const mlsgpu_swathe *swathes, uint32_t count)
{
    REQUIRE(ms != nullptr && dFields != nullptr && pitches != nullptr && swathes != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(count >= 1 && count <= MLSGPU_MAX_BATCH, MLSGPU_ERR_LENGTH);
    MlsArgs args[MAX_LANES];
    for (uint32_t k = 0; k < count; k++)
        PROPAGATE(mlsLaneArgs(ms[k], dFields[k], pitches[k], fieldRows != nullptr ? fieldRows[k] : UINT64_MAX, &swathes[k], &args[k]));
    return mlsEnqueueLanes(ms, args, count);
}

static int mlsGeneratorEnqueue(void *user, void *stream, float *dField, uint64_t pitch, const mlsgpu_swathe *swathe)
{
    mlsgpu_mls *m = static_cast<mlsgpu_mls *>(user);
    (void) stream;   /* the functor and Marching share the worker's context, hence its stream */
    /* Marching guarantees the rows (src/marching.h:232-241); pass "unbounded" and let Marching's own check stand */
    return mlsgpu_hip_mls_enqueue(m, dField, pitch, UINT64_MAX, swathe);
}

MLSGPU_API int mlsgpu_hip_mls_generator(mlsgpu_mls *m, mlsgpu_generator *gen)
{
    REQUIRE(m != nullptr && gen != nullptr, MLSGPU_ERR_INVALID);
    gen->alignment[0] = gen->alignment[1] = gen->alignment[2] = 8;   /* MlsFunctor::wgs, src/mls.cpp:53 */
    gen->enqueue = mlsGeneratorEnqueue;
    gen->user = m;
    return MLSGPU_OK;
}

/* the MlsFunctor behind a generator made by mlsgpu_hip_mls_generator, or NULL: Marching's batched generate launches
 * processCorners for all its buckets at once when every generator is one */
MLSGPU_API mlsgpu_mls *mlsgpu_hip_mls_of_generator(const mlsgpu_generator *gen)
{
    return gen != nullptr && gen->enqueue == mlsGeneratorEnqueue ? static_cast<mlsgpu_mls *>(gen->user) : nullptr;
}

/* kernel variant, work counters and boundary limit of `src` for `dst` (the lanes of a batched worker follow lane 0) */
MLSGPU_API int mlsgpu_hip_mls_copy_settings(mlsgpu_mls *dst, const mlsgpu_mls *src)
{
    REQUIRE(dst != nullptr && src != nullptr && dst->shape == src->shape, MLSGPU_ERR_INVALID);
    dst->variant = src->variant;
    dst->dStats = src->dStats;
    dst->boundaryFactor = src->boundaryFactor;
    return MLSGPU_OK;
}

/* ---- one-work-item test kernels of kernels/mls.cl:439-469 ---- */
namespace
{
__global__ void testMlsKernel(int op, const float *in, uint32_t n, float *out)
{
    if (op == 0)
        out[0] = solveQuadratic(in[0], in[1], in[2]);
    else
    {
        /* testFitSphere: weights in the quality slot, positions in the local frame */
        Fit fit;
        fitInit(fit);
        for (uint32_t i = 0; i < n; i++)
        {
            const float *s = in + 8 * i;
            fitAdd(fit, s[7], s[0], s[1], s[2], dot3(s[0], s[1], s[2], s[0], s[1], s[2]), s[4], s[5], s[6]);
        }
        const float invSumW = 1.0f / fit.sumW;
        const float mx = fit.sumWpx * invSumW, my = fit.sumWpy * invSumW, mz = fit.sumWpz * invSumW;
        const float qNum = fit.sumWpn - dot3(mx, my, mz, fit.sumWnx, fit.sumWny, fit.sumWnz);
        const float qDen = fit.sumWpp - dot3(mx, my, mz, fit.sumWpx, fit.sumWpy, fit.sumWpz);
        float q = qNum / qDen;
        if (fabsf(qDen) < (4 * 1.1920928955078125e-07f) * (float) fit.hits * fabsf(fit.sumWpp) || !isfinite(q))
            q = 0.0f;
        const float a = 0.5f * q;
        const float bx = (fit.sumWnx - q * fit.sumWpx) * invSumW;
        const float by = (fit.sumWny - q * fit.sumWpy) * invSumW;
        const float bz = (fit.sumWnz - q * fit.sumWpz) * invSumW;
        out[0] = bx; out[1] = by; out[2] = bz; out[3] = a;
        out[4] = (-a * fit.sumWpp - dot3(bx, by, bz, fit.sumWpx, fit.sumWpy, fit.sumWpz)) * invSumW;
    }
}

int runMlsTest(mlsgpu_ctx *ctx, int op, const float *in, size_t inFloats, uint32_t n, float *out, int outFloats)
{
    float *dIn = nullptr, *dOut = nullptr;
    HIP_CHECK(hipSetDevice(ctx->device));
    HIP_CHECK(hipMalloc(&dIn, inFloats * 4 + 4));
    HIP_CHECK(hipMalloc(&dOut, 32));
    HIP_CHECK(hipMemcpyAsync(dIn, in, inFloats * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(testMlsKernel, dim3(1), dim3(1), 0, ctx->stream, op, (const float *) dIn, n, dOut);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(out, dOut, outFloats * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    hipFree(dIn); hipFree(dOut);
    return MLSGPU_OK;
}
} // namespace

MLSGPU_API int mlsgpu_hip_test_solve_quadratic(mlsgpu_ctx *ctx, float a, float b, float c, float *out)
{
    REQUIRE(ctx != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    const float in[3] = {a, b, c};
    return runMlsTest(ctx, 0, in, 3, 0, out, 1);
}

MLSGPU_API int mlsgpu_hip_test_fit_sphere(mlsgpu_ctx *ctx, const mlsgpu_splat *hSplats, uint32_t n, float out[5])
{
    REQUIRE(ctx != nullptr && hSplats != nullptr && out != nullptr && n > 0, MLSGPU_ERR_INVALID);
    return runMlsTest(ctx, 1, reinterpret_cast<const float *>(hSplats), (size_t) n * 8, n, out, 5);
}
