/*
 * Splat octree build on gfx950 -- the device half of SplatTreeCL
 * (reference: src/splat_tree_cl.{h,cpp}, kernels/octree.cl).
 *
 * Output contract (identical to the reference, checked bit-for-bit against the oracle):
 * `start` (one int per node, levels stored finest first) and `commands`
 * ([end][ids...][jump] per non-empty node), see src/splat_tree.h:40-74.
 *
 * Differences in mechanism (not in results):
 *  - the indicator (countCommands) and commandMap arrays are never materialised: the indicator
 *    is recomputed from the sorted keys inside the scan, and writeSplatIds is the scan's consumer;
 *  - the per-level writeStart launches (levels dependent launches in the reference) are one
 *    launch: every node finds its nearest non-empty strict ancestor by itself;
 *  - the sort handles 3*(maxShift-minShift)+1 key bits in ceil(bits/10) stable passes.
 */
#include "common.hpp"
#include "primitives.hpp"

#include <cstdlib>

using namespace mlsgpu;

struct LevelOffsets
{
    uint32_t v[32];
};

struct mlsgpu_tree
{
    mlsgpu_ctx *ctx = nullptr;
    uint64_t maxLevels = 0, maxSplats = 0;
    uint64_t maxStart = 0, commandsSize = 0;
    uint32_t numLevels = 0;
    int32_t *dStart = nullptr, *dJumpPos = nullptr, *dCommands = nullptr;
    uint32_t *dKeysA = nullptr, *dKeysB = nullptr, *dValsA = nullptr, *dValsB = nullptr;
    uint32_t *dHist = nullptr, *dTileSums = nullptr, *dNumEntries = nullptr;
    uint32_t *dNodeCounts = nullptr, *dNodeBase = nullptr;     /* per node: entries, and twice the non-empty nodes before it */
    uint32_t *dDigitBase = nullptr;     /* 257 words: where the groups of the fused first pass begin (packed entries) */
    U3 *dNodeTiles = nullptr;           /* tile sums of the scan over the nodes */
    HostMailbox entryBox;               /* the entry count comes back to the host once per build */
    uint8_t *dSlotMasks = nullptr;      /* per splat: which of its 8 candidate slots are real entries */
    uint64_t *dEntryNotes = nullptr;    /* per splat, the fused front end: slot mask, level, node coordinates (packNote) */
    mlsgpu_splat *dSplats = nullptr;   /* borrowed between build and clear_splats */
    bool mutate = true;                 /* radius -> 1/radius^2 in place (the reference); false: the splats stay as they came */
};

namespace
{

/* kernels/octree.cl:121-136 (z-major Morton); coordinates here are < 2^10 */
__device__ __forceinline__ uint32_t spread3(uint32_t v)
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__device__ __forceinline__ uint32_t makeCode(int x, int y, int z)
{
    return spread3((uint32_t) x) | (spread3((uint32_t) y) << 1) | (spread3((uint32_t) z) << 2);
}

/* general form for the test kernel (any non-negative ints), octree.cl:121-136 verbatim semantics */
__device__ uint32_t makeCodeLoop(int x, int y, int z)
{
    uint32_t ans = 0, scale = 1;
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

/* kernels/octree.cl:50-55 */
__device__ __forceinline__ int levelShift(int lox, int loy, int loz, int hix, int hiy, int hiz)
{
    int big = max(max(hix - lox, hiy - loy), hiz - loz);
    return big > 1 ? 32 - __clz(big - 1) : 0;
}

/* kernels/octree.cl:60-66: OpenCL dot() restated as a plain sum of products (no fma) */
__device__ __forceinline__ float pointBoxDist2(float px, float py, float pz, float lx, float ly, float lz,
                                                float hx, float hy, float hz)
{
    float dx = fmaxf(lx, fminf(hx, px)) - px;
    float dy = fmaxf(ly, fminf(hy, py)) - py;
    float dz = fmaxf(lz, fminf(hz, pz)) - pz;
    return dx * dx + dy * dy + dz * dz;
}

/* convert_int_rtn: floor then saturating convert (v_cvt_i32_f32 saturates) */
__device__ __forceinline__ int floorToInt(float v)
{
    float f = floorf(v);
    if (!(f > -2147483648.0f)) return INT32_MIN;
    if (!(f < 2147483648.0f)) return INT32_MAX;
    return (int) f;
}

/* The per-splat part of writeEntries, kernels/octree.cl:159-214 (+prepare :79-90, goodEntry :100-110):
 * the up-to-8 sort keys of one splat; returns the bit mask of the slots that hold a real entry
 * (the reference writes UINT_MAX into the others). */
struct EntryParams
{
    mlsgpu_splat *splats;
    int bx, by, bz;
    LevelOffsets levelOffsets;
    int minShift, maxShift;
    uint32_t firstSplat;
    uint32_t mutate;            /* write 1/r^2 into the radius slot (kernels/octree.cl:193), or leave the splats untouched */
};

/* ... and where its 2 x 2 x 2 candidate nodes are: level and lowest node coordinates (what a key is made of) */
struct EntryPlace
{
    int shift, ilx, ily, ilz;
};

__device__ __forceinline__ uint32_t splatEntries(const EntryParams &P, const float4 pr, uint32_t k[8], EntryPlace *place = nullptr)
{
    /* prepare, octree.cl:79-90 */
    const int lox = floorToInt(pr.x - pr.w), loy = floorToInt(pr.y - pr.w), loz = floorToInt(pr.z - pr.w);
    const int hix = floorToInt(pr.x + pr.w), hiy = floorToInt(pr.y + pr.w), hiz = floorToInt(pr.z + pr.w);
    int shift = levelShift(lox, loy, loz, hix, hiy, hiz);
    shift = min(max(shift, P.minShift), P.maxShift);
    const int ilx = max(lox - P.bx, 0) >> shift;
    const int ily = max(loy - P.by, 0) >> shift;
    const int ilz = max(loz - P.bz, 0) >> shift;
    float radius2 = pr.w * pr.w;
    radius2 *= 1.00001f;
    const uint32_t levelOffset = P.levelOffsets.v[shift];
    const int bound = 1 << (P.maxShift - shift);
    if (place != nullptr)
        *place = EntryPlace{shift, ilx, ily, ilz};
    uint32_t mask = 0;
#pragma unroll
    for (int o = 0; o < 8; o++)
    {
        const int ax = ilx + (o & 1), ay = ily + ((o >> 1) & 1), az = ilz + (o >> 2);
        /* goodEntry, octree.cl:100-110; int arithmetic wraps like OpenCL's */
        const int blx = (int) ((uint32_t) ax << shift) + P.bx, bhx = (int) ((uint32_t) (ax + 1) << shift) + P.bx;
        const int bly = (int) ((uint32_t) ay << shift) + P.by, bhy = (int) ((uint32_t) (ay + 1) << shift) + P.by;
        const int blz = (int) ((uint32_t) az << shift) + P.bz, bhz = (int) ((uint32_t) (az + 1) << shift) + P.bz;
        bool isect = pointBoxDist2(pr.x, pr.y, pr.z, (float) blx, (float) bly, (float) blz,
                                   (float) bhx, (float) bhy, (float) bhz) < radius2;
        isect = isect && ax < bound && ay < bound && az < bound;
        /* inside the bounds the coordinates fit 10 bits (maxShift - shift <= 9 levels) */
        k[o] = makeCode(ax, ay, az) + levelOffset;
        mask |= (isect ? 1u : 0u) << o;
    }
    return mask;
}

/*
 * writeEntries as a compaction: the reference writes 8 (key, id) slots per splat and lets the UINT_MAX
 * ones sort to the end, where writeSplatIds ignores them (kernels/octree.cl:263).  Here only the real
 * entries are written, densely, in the same (splat, slot) order -- a stable sort of the survivors gives
 * exactly the order the padded array would have had before its UINT_MAX tail, so `commands` and `start`
 * are unchanged while sort, scan and writeSplatIds touch about half the data.
 * Producer: number of real entries of splat i.  Consumer: writes them at the scanned position and
 * replaces splat.w by 1/r^2 (:193).
 */
struct EntryCountIn            /* phase 1: evaluates the splat, remembers which slots are real entries */
{
    EntryParams P;
    uint8_t *slotMasks;
    __device__ __forceinline__ uint32_t operator()(uint64_t i) const
    {
        const float4 pr = reinterpret_cast<const float4 *>(P.splats + (i + P.firstSplat))[0];
        uint32_t k[8];
        const uint32_t mask = splatEntries(P, pr, k);
        slotMasks[i] = (uint8_t) mask;
        return (uint32_t) __popc(mask);
    }
};

struct EntryMaskIn             /* phase 2: the count again, from the remembered mask */
{
    const uint8_t *slotMasks;
    __device__ __forceinline__ uint32_t operator()(uint64_t i) const { return (uint32_t) __popc((uint32_t) slotMasks[i]); }
};

struct EntryWriteOut
{
    EntryParams P;
    const uint8_t *slotMasks;
    uint32_t *keys, *values;
    __device__ __forceinline__ void operator()(uint64_t i, uint32_t pos, uint32_t) const
    {
        const uint32_t gid = (uint32_t) i + P.firstSplat;
        float4 *sp = reinterpret_cast<float4 *>(P.splats + gid);
        const float4 pr = sp[0];
        if (P.mutate)
            reinterpret_cast<float *>(sp)[3] = 1.0f / (pr.w * pr.w);
        const uint32_t mask = slotMasks[i];
        if (mask == 0)
            return;
        /* prepare (octree.cl:79-90) again for the node coordinates; the box tests are not repeated */
        const int lox = floorToInt(pr.x - pr.w), loy = floorToInt(pr.y - pr.w), loz = floorToInt(pr.z - pr.w);
        const int hix = floorToInt(pr.x + pr.w), hiy = floorToInt(pr.y + pr.w), hiz = floorToInt(pr.z + pr.w);
        int shift = levelShift(lox, loy, loz, hix, hiy, hiz);
        shift = min(max(shift, P.minShift), P.maxShift);
        const int ilx = max(lox - P.bx, 0) >> shift;
        const int ily = max(loy - P.by, 0) >> shift;
        const int ilz = max(loz - P.bz, 0) >> shift;
        const uint32_t levelOffset = P.levelOffsets.v[shift];
#pragma unroll
        for (int o = 0; o < 8; o++)
            if (mask & (1u << o))
            {
                keys[pos] = makeCode(ilx + (o & 1), ily + ((o >> 1) & 1), ilz + (o >> 2)) + levelOffset;
                values[pos] = gid;
                pos++;
            }
    }
};

/*
 * writeEntries fused with the FIRST pass of the entry sort (round 3).  The entries of a tile of ENT_TILE splats are never
 * written in (splat, slot) order: the counting kernel histograms the low digit of their keys per tile while it evaluates the
 * box tests, one digit scan turns the histograms into positions (the same kernel the sort uses), and the scattering kernel
 * regenerates the tile's entries from the remembered slot masks, lines them up in LDS in (splat, slot) order -- the order a
 * stable sort has to start from -- and sends them straight to where the first LSD pass would have put them.  That is one
 * write and one read of all entries (16 bytes each) and two launches less than writeEntries + histogram + scatter; the
 * remaining passes of the sort are unchanged.  Results are identical: same keys, same stable order.
 */
enum
{
    ENT_THREADS = 512,              /* threads per workgroup */
    ENT_PER = 2,                    /* consecutive splats per thread.  entryScatter per launch of four buckets (cfg3): 1 splat
                                     * per thread 193.5 us, 2: 170.0, 3: 208.6, 4: 211.9 -- twice the entries between two
                                     * barriers and twice as long a run of every digit in memory, while three workgroups still
                                     * fit a CU's LDS (two at 3 and 4) */
    ENT_TILE = ENT_THREADS * ENT_PER,   /* splats per workgroup */
    ENT_CAP = 8 * ENT_TILE,         /* at most eight entries per splat */
    ENT_BIN_BITS = 8,               /* the fused pass handles digits of up to 8 bits */
    ENT_KEY_BITS = 22               /* ... of keys that leave ten bits of a word for the splat's place in the tile */
};

/* What entryHist leaves entryScatter per splat, 8 bytes: low word = slot mask (bits 0-7), level (8-12), lowest node x
 * (13-22); high word = lowest node y (0-9) and z (10-19); ten bits per coordinate inside the bounds -- so the scatter reads
 * 8 bytes per splat instead of the 32-byte record (of which it needed 16) and repeats none of the arithmetic.  0 = no entry. */
__device__ __forceinline__ uint64_t packNote(uint32_t mask, const EntryPlace &pl)
{
    if (mask == 0)
        return 0;
    const uint32_t lo = mask | (uint32_t) pl.shift << 8 | ((uint32_t) pl.ilx & 0x3FFu) << 13;
    const uint32_t hi = ((uint32_t) pl.ily & 0x3FFu) | ((uint32_t) pl.ilz & 0x3FFu) << 10;
    return (uint64_t) lo | (uint64_t) hi << 32;
}

struct EntryHistArgs
{
    EntryParams P;
    uint64_t *notes;
    uint32_t *hist;
    uint32_t numTiles;
    uint64_t n;
};

__global__ __launch_bounds__(ENT_THREADS) void entryHistKernel(Lanes<EntryHistArgs> lanes, uint32_t digitBits)
{
    const EntryHistArgs &A = lanes.a[blockIdx.y];     /* by reference: the level offsets are indexed dynamically and stay
                                                         * in the kernel-argument segment (a copy would live in scratch) */
    if (blockIdx.x >= A.numTiles)
        return;
    __shared__ uint32_t bins[1 << ENT_BIN_BITS];
    const uint32_t numBins = 1u << digitBits, dmask = numBins - 1;
    for (uint32_t d = threadIdx.x; d < numBins; d += ENT_THREADS)
        bins[d] = 0;
    __syncthreads();
    /* a thread's splats are neighbours in the cloud (one 32 x ENT_PER-byte stretch): both records are requested before the
     * first is looked at */
    const uint32_t tile = tileOfWorkgroup(blockIdx.x, A.numTiles);      /* its counters' lines are completed in one XCD's L2 */
    const uint64_t i0 = (uint64_t) tile * ENT_TILE + (uint64_t) threadIdx.x * ENT_PER;
    float4 pr[ENT_PER];
#pragma unroll
    for (int s_ = 0; s_ < ENT_PER; s_++)
        pr[s_] = reinterpret_cast<const float4 *>(A.P.splats + ((i0 + s_ < A.n ? i0 + s_ : 0) + A.P.firstSplat))[0];
#pragma unroll
    for (int s_ = 0; s_ < ENT_PER; s_++)
        if (i0 + s_ < A.n)
        {
            uint32_t k[8];
            EntryPlace pl;
            const uint32_t mask = splatEntries(A.P, pr[s_], k, &pl);
            A.notes[i0 + s_] = packNote(mask, pl);
            if (A.P.mutate)     /* kernels/octree.cl:193; nothing behind this kernel reads the radius */
                reinterpret_cast<float *>(A.P.splats + ((i0 + s_) + A.P.firstSplat))[3] = 1.0f / (pr[s_].w * pr[s_].w);
#pragma unroll
            for (int o = 0; o < 8; o++)
                if (mask & (1u << o))
                    atomicAdd(&bins[k[o] & dmask], 1u);
        }
    __syncthreads();
    uint32_t *const hist = A.hist;
    const uint32_t numTiles = A.numTiles;
    for (uint32_t d = threadIdx.x; d < numBins; d += ENT_THREADS)
        hist[(uint64_t) d * numTiles + tile] = bins[d];
}

struct EntryTotalArgs
{
    const uint32_t *digitTotals;
    uint32_t *total;
    uint32_t *digitBase;        /* [numBins + 1]: exclusive prefix of the totals -- where every digit's group begins */
};

/* the number of entries = the sum of the digit totals; one workgroup adds up every lane's in turn */
__global__ __launch_bounds__(256) void entryTotalKernel(Lanes<EntryTotalArgs> lanes, uint32_t count, uint32_t numBins,
                                                        uint32_t *box, uint32_t seq)
{
    __shared__ uint32_t waveTotals[4];
    for (uint32_t k = 0; k < count; k++)
    {
        const uint32_t *const digitTotals = lanes.a[k].digitTotals;
        /* numBins <= 256 (ENT_BIN_BITS): one digit per thread */
        const uint32_t mine = threadIdx.x < numBins ? digitTotals[threadIdx.x] : 0u;
        const uint32_t incl = waveInclusiveScan(mine);
        if ((threadIdx.x & 63) == 63)
            waveTotals[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++)
            before += waveTotals[w];
        if (threadIdx.x < numBins)
            lanes.a[k].digitBase[threadIdx.x] = before + incl - mine;
        if (threadIdx.x == 0)
        {
            const uint32_t sum = waveTotals[0] + waveTotals[1] + waveTotals[2] + waveTotals[3];
            lanes.a[k].digitBase[numBins] = sum;
            *lanes.a[k].total = sum;
            /* ... and straight to the host (HostMailbox): the count sizes the launches that follow */
            __hip_atomic_store(box + 1 + k, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        __threadfence_system();
        __hip_atomic_store(box, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

struct EntryScatterArgs
{
    EntryParams P;
    const uint64_t *notes;
    const uint32_t *hist;
    const uint32_t *digitTotals;
    uint32_t numTiles;
    uint64_t n;
    uint32_t *keysOut, *valsOut;
    uint32_t idBits;            /* != 0: an entry leaves as ONE word, (key >> digitBits) << idBits | (id - firstSplat), in keysOut */
};

__global__ __launch_bounds__(ENT_THREADS) __attribute__((amdgpu_waves_per_eu(6, 8))) void entryScatterKernel(Lanes<EntryScatterArgs> lanes, uint32_t digitBits)
{
    enum { BINS = 1 << ENT_BIN_BITS, WAVES = ENT_THREADS / 64, MAX_ROUNDS = ENT_CAP / ENT_THREADS };
    const EntryScatterArgs &A = lanes.a[blockIdx.y];  /* by reference, as in entryHistKernel */
    if (blockIdx.x >= A.numTiles)
        return;
    const EntryParams &P = A.P;
    const uint64_t *const notes = A.notes;
    const uint32_t *const hist = A.hist;
    const uint32_t *const digitTotals = A.digitTotals;
    const uint32_t numTiles = A.numTiles;
    const uint64_t n = A.n;
    uint32_t *const keysOut = A.keysOut, *const valsOut = A.valsOut;
    __shared__ uint32_t waveBins[WAVES][BINS];
    __shared__ uint32_t tileBase[BINS];
    __shared__ uint32_t waveTotals[WAVES], waveTotalsAll[WAVES], waveCnt[WAVES];
    /* An entry in LDS is ONE word: its key (ENT_KEY_BITS at most on this route) below the thread that holds its splat (the
     * id is the tile's first id + that).  Key and id travel together through the reorder, so the ids need no pass of their
     * own: six workgroup barriers instead of eight, and 41 KB of LDS for a tile of 1024 splats (three workgroups per CU). */
    __shared__ uint32_t sEnt[ENT_CAP];          /* the tile's entries in (splat, slot) order; afterwards in the pass's order */
    __shared__ uint32_t sMatch[WAVES][BINS];    /* per wave and digit: the lanes of HALF a wave that hold it (sortScatterKernel's
                                                 * way of ranking, 32 lanes at a time: 64-bit words would cost the third workgroup
                                                 * of a CU its LDS) */
    const uint32_t numBins = 1u << digitBits, dmask = numBins - 1;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t d = threadIdx.x; d < numBins; d += ENT_THREADS)
    {
#pragma unroll
        for (int w = 0; w < WAVES; w++)
        {
            waveBins[w][d] = 0;
            sMatch[w][d] = 0u;
        }
    }
    /* 1. the tile's entries, lined up in (splat, slot) order.  What the thread reads that does not depend on another read is
     * requested here, together: its slot mask, its splat, and the digit totals and tile offsets of the bins it owns in the
     * scan further down (one round of memory latency instead of three). */
    const uint32_t tile = tileOfWorkgroup(blockIdx.x, numTiles);
    const uint64_t i0 = (uint64_t) tile * ENT_TILE + (uint64_t) threadIdx.x * ENT_PER;
    uint64_t note[ENT_PER];
    uint32_t cnt = 0;
#pragma unroll
    for (int s_ = 0; s_ < ENT_PER; s_++)
        note[s_] = i0 + s_ < n ? notes[i0 + s_] : 0ull;
    const uint32_t per = numBins > ENT_THREADS ? numBins / ENT_THREADS : 1;
    const uint32_t d0 = threadIdx.x * per;
    const uint32_t totalOfBin = d0 < numBins ? digitTotals[d0] : 0u;
    const uint32_t histOfBin = d0 < numBins ? hist[(uint64_t) d0 * numTiles + tile] : 0u;
#pragma unroll
    for (int s_ = 0; s_ < ENT_PER; s_++)
        cnt += (uint32_t) __popc((uint32_t) note[s_] & 0xFFu);
    const uint32_t incl = waveInclusiveScan(cnt);
    if (lane == 63)
        waveCnt[wave] = incl;
    __syncthreads();
    uint32_t pos = incl - cnt, tileCount = 0;
#pragma unroll
    for (uint32_t w = 0; w < WAVES; w++)
    {
        if (w < wave)
            pos += waveCnt[w];
        tileCount += waveCnt[w];
    }
#pragma unroll
    for (int s_ = 0; s_ < ENT_PER; s_++)
    {
        const uint32_t mask = (uint32_t) note[s_] & 0xFFu;
        if (mask != 0)
        {
            const uint32_t lo32 = (uint32_t) note[s_], hi32 = (uint32_t) (note[s_] >> 32);
            const int shift = (int) ((lo32 >> 8) & 0x1Fu);
            const int ilx = (int) ((lo32 >> 13) & 0x3FFu), ily = (int) (hi32 & 0x3FFu), ilz = (int) ((hi32 >> 10) & 0x3FFu);
            const uint32_t levelOffset = P.levelOffsets.v[shift];
            const uint32_t holder = (threadIdx.x * ENT_PER + (uint32_t) s_) << ENT_KEY_BITS;
#pragma unroll
            for (int o = 0; o < 8; o++)
                if (mask & (1u << o))
                    sEnt[pos++] = (makeCode(ilx + (o & 1), ily + ((o >> 1) & 1), ilz + (o >> 2)) + levelOffset) | holder;
        }
    }
    __syncthreads();
    if (tileCount == 0)
        return;
    /* 2. one stable LSD pass over the tile, as sortScatterKernel: wave w owns `rounds` x 64 consecutive elements */
    const uint32_t rounds = (tileCount + ENT_THREADS - 1) / ENT_THREADS;
    const uint32_t first = wave * rounds * 64 + lane;
    uint32_t ent[MAX_ROUNDS];
#pragma unroll
    for (int j = 0; j < MAX_ROUNDS; j++)
    {
        const uint32_t e = first + j * 64;
        const bool valid = (uint32_t) j < rounds && e < tileCount;
        ent[j] = valid ? sEnt[e] : 0u;
        if (valid)
            atomicAdd(&waveBins[wave][ent[j] & dmask], 1u);
    }
    __syncthreads();
    {
        uint32_t mine = 0, mineAll = 0;
        if (d0 < numBins)
            for (uint32_t k = 0; k < per; k++)
            {
                mineAll += k == 0 ? totalOfBin : digitTotals[d0 + k];
#pragma unroll
                for (int w = 0; w < WAVES; w++)
                    mine += waveBins[w][d0 + k];
            }
        const uint32_t inclMine = waveInclusiveScan(mine), inclAll = waveInclusiveScan(mineAll);
        if (lane == 63)
        {
            waveTotals[wave] = inclMine;
            waveTotalsAll[wave] = inclAll;
        }
        __syncthreads();
        uint32_t run = inclMine - mine, base = inclAll - mineAll;
        for (uint32_t w = 0; w < wave; w++)
        {
            run += waveTotals[w];
            base += waveTotalsAll[w];
        }
        if (d0 < numBins)
            for (uint32_t k = 0; k < per; k++)
            {
                const uint32_t d = d0 + k;
                tileBase[d] = base + (k == 0 ? histOfBin : hist[(uint64_t) d * numTiles + tile]) - run;
                base += k == 0 ? totalOfBin : digitTotals[d];
#pragma unroll
                for (int w = 0; w < WAVES; w++)
                {
                    const uint32_t c = waveBins[w][d];
                    waveBins[w][d] = run;
                    run += c;
                }
            }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MAX_ROUNDS; j++)
    {
        if ((uint32_t) j >= rounds)     /* uniform: a tile holds 3.8 entries per splat on average, 8 at most */
            continue;
        const uint32_t e = first + j * 64;
        const bool valid = e < tileCount;
        const uint32_t digit = ent[j] & dmask;
        /* the lanes with the same digit, through LDS: see sortScatterKernel */
#pragma unroll
        for (uint32_t half = 0; half < 2; half++)       /* the lower lanes first: the order of the elements */
        {
            const bool mine = valid && (lane >> 5) == half;
            const uint32_t bit = 1u << (lane & 31u);
            if (mine)
                atomicOr(&sMatch[wave][digit], bit);
            __builtin_amdgcn_wave_barrier();
            const uint32_t peers = mine ? sMatch[wave][digit] : 0u;
            __builtin_amdgcn_wave_barrier();
            if (mine)
            {
                const uint32_t rank = (uint32_t) __popc(peers & (bit - 1u));
                const uint32_t dst = waveBins[wave][digit] + rank;
                sEnt[dst] = ent[j];      /* (every wave took its elements into registers before the last two barriers) */
                if (rank == 0)
                {
                    waveBins[wave][digit] = dst + (uint32_t) __popc(peers);
                    sMatch[wave][digit] = 0u;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    /* 3. out, in the order of the pass: the run of a digit is contiguous in the tile and in memory */
    const uint32_t idBits = A.idBits;
    const uint32_t tileFirstId = tile * ENT_TILE + (idBits != 0 ? 0u : P.firstSplat);
#pragma unroll
    for (int k = 0; k < MAX_ROUNDS; k++)
    {
        const uint32_t p = threadIdx.x + k * ENT_THREADS;
        if (p < tileCount)
        {
            const uint32_t e = sEnt[p];
            const uint32_t key = e & ((1u << ENT_KEY_BITS) - 1u);
            const uint32_t out = tileBase[key & dmask] + p;
            const uint32_t id = tileFirstId + (e >> ENT_KEY_BITS);
            if (idBits != 0)
                keysOut[out] = (key >> digitBits) << idBits | id;     /* the digit is where the entry lies from now on */
            else
            {
                keysOut[out] = key;
                valsOut[out] = id;
            }
        }
    }
}

/* countCommands (kernels/octree.cl:230-239) as the scan's producer.  The reference leaves the
 * last indicator unwritten; an exclusive scan never reads it, so any value serves. */
struct IndicatorIn
{
    const uint32_t *keys;
    const uint32_t *n;          /* number of entries, on the device */
    __device__ __forceinline__ uint32_t operator()(uint64_t i) const
    {
        if (i + 1 >= *n)
            return 1u;
        return keys[i] != keys[i + 1] ? 3u : 1u;
    }
};

/* writeSplatIds (kernels/octree.cl:256-279) as the scan's consumer: cpos is the exclusive prefix. */
struct SplatIdsOut
{
    int32_t *commands, *start, *jumpPos;
    const uint32_t *keys, *ids;
    const uint32_t *n;
    __device__ __forceinline__ void operator()(uint64_t pos, uint32_t cpos, uint32_t) const
    {
        const uint32_t curKey = keys[pos];
        if (curKey != 0xFFFFFFFFu)
        {
            commands[cpos] = (int32_t) ids[pos];
            const uint32_t prevKey = pos > 0 ? keys[pos - 1] : 0xFFFFFFFFu;
            const uint32_t nextKey = pos + 1 < *n ? keys[pos + 1] : 0xFFFFFFFFu;
            if (prevKey != curKey)
                start[curKey] = (int32_t) (cpos - 1);
            if (curKey != nextKey)
                jumpPos[curKey] = (int32_t) (cpos + 1);
        }
    }
};

/*
 * writeStartTop + writeStart for all levels in one launch (kernels/octree.cl:292-341, loop
 * src/splat_tree_cl.cpp:320-331).  The reference walks levels coarse to fine and hands each node
 * its parent's start.  Unrolled, that value is: the start of the nearest NON-EMPTY strict ancestor,
 * or -1.  start[] of a non-empty node is written by writeSplatIds and never changes, so every node
 * can fetch it independently.
 */
/*
 * The command list without a pass over the entries (round 4).  In the sorted entry list a node's ids are one run, and
 * writeSplatIds (kernels/octree.cl:256-279) puts the id of sorted rank r at command position 1 + r + 2 * (non-empty nodes
 * before the id's node): the scan of countCommands' 1 / 3 indicators (src/splat_tree_cl.cpp:310-317) is that sum, entry by
 * entry.  With the number of entries of every node at hand -- the last pass's histogram kernel counts whole keys on the way
 * (sortHistKernel<KEY_COUNTS>) -- a scan over the NODES (37 449 of them for the default tree, not 7 million entries) gives
 * each node its first rank and the number of non-empty nodes before it, hence start[] and the jump slots of the non-empty
 * nodes (NodeOut below), and the last scatter pass of the sort writes every id straight to its command position
 * (sortScatterKernel<SPREAD>): the sorted (key, id) arrays are never written and countCommands / scan / writeSplatIds do
 * not run.  `commands` and `start` are the reference's, word for word (tests/test_gpu_tree.py).
 */
struct NodeIn
{
    const uint32_t *counts;
    __device__ __forceinline__ U3 operator()(uint64_t i) const
    {
        const uint32_t c = counts[i];
        return U3{c, c != 0 ? 1u : 0u, 0u};
    }
};

struct NodeOut
{
    int32_t *start, *jumpPos;
    uint32_t *nodeBase;
    __device__ __forceinline__ void operator()(uint64_t i, U3 excl, U3 val) const
    {
        if (val.a != 0)
        {
            /* first id at 1 + excl.a + 2 excl.b: start = that - 1 (octree.cl:272-274), the jump slot behind the last id (:275-277) */
            const uint32_t first = excl.a + 2 * excl.b;
            start[i] = (int32_t) first;
            jumpPos[i] = (int32_t) (first + val.a + 1);
            nodeBase[i] = 2 * excl.b;
        }
        else
            jumpPos[i] = -1;            /* fill(jumpPos, -1), kernels/octree.cl:346 */
    }
};

struct WriteStartArgs
{
    int32_t *start, *commands;
    const int32_t *jumpPos;
};

__global__ __launch_bounds__(256) void writeStartKernel(Lanes<WriteStartArgs> lanes, LevelOffsets levelOffsets, int minShift,
                                                        int maxShift, uint32_t numStart)
{
    const WriteStartArgs A = lanes.a[blockIdx.y];
    int32_t *const start = A.start, *const commands = A.commands;
    const int32_t *const jumpPos = A.jumpPos;
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= numStart)
        return;
    int level = minShift;
    while (level < maxShift && p >= levelOffsets.v[level + 1])
        level++;
    uint32_t code = p - levelOffsets.v[level];
    int32_t prev = -1;
    for (int l = level + 1; l <= maxShift; l++)
    {
        code >>= 3;
        const uint32_t a = levelOffsets.v[l] + code;
        if (jumpPos[a] >= 0)
        {
            prev = start[a];
            break;
        }
    }
    const int32_t jp = jumpPos[p];
    if (jp >= 0)
    {
        commands[jp] = prev;
        commands[start[p]] = jp;
    }
    else
        start[p] = prev;
}

/* fill(jumpPos, -1), kernels/octree.cl:346, for every lane */
__global__ __launch_bounds__(256) void fillKernel(Lanes<int32_t *> lanes, uint32_t n, int32_t value)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        lanes.a[blockIdx.y][i] = value;
}

__global__ void testHelpersKernel(int op, const int32_t *iargs, const float *fargs, uint32_t *out)
{
    if (op == 0)
        out[0] = makeCodeLoop(iargs[0], iargs[1], iargs[2]);
    else if (op == 1)
        out[0] = (uint32_t) levelShift(iargs[0], iargs[1], iargs[2], iargs[3], iargs[4], iargs[5]);
    else if (op == 2)
    {
        float r = pointBoxDist2(fargs[0], fargs[1], fargs[2], fargs[3], fargs[4], fargs[5], fargs[6], fargs[7], fargs[8]);
        out[0] = __float_as_uint(r);
    }
    else if (op == 3)
        out[0] = makeCode(iargs[0], iargs[1], iargs[2]);   /* fast path must agree with the loop form */
}

int runTestHelper(mlsgpu_ctx *ctx, int op, const int32_t *iargs, int ni, const float *fargs, int nf, uint32_t *out)
{
    int32_t *dI = nullptr;
    float *dF = nullptr;
    uint32_t *dO = nullptr;
    HIP_CHECK(hipSetDevice(ctx->device));
    HIP_CHECK(hipMalloc(&dI, 64));
    HIP_CHECK(hipMalloc(&dF, 64));
    HIP_CHECK(hipMalloc(&dO, 16));
    if (ni) HIP_CHECK(hipMemcpyAsync(dI, iargs, ni * 4, hipMemcpyHostToDevice, ctx->stream));
    if (nf) HIP_CHECK(hipMemcpyAsync(dF, fargs, nf * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(testHelpersKernel, dim3(1), dim3(1), 0, ctx->stream, op, (const int32_t *) dI, (const float *) dF, dO);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(out, dO, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    hipFree(dI); hipFree(dF); hipFree(dO);
    return MLSGPU_OK;
}

} // namespace

/* ------------------------------------------------------------------ C-ABI */

static void treeSizes(uint64_t maxLevels, uint64_t maxSplats, uint64_t *maxStart, uint64_t *commandsSize)
{
    /* src/splat_tree_cl.cpp:112-123 */
    *maxStart = (uint64_t(1) << (3 * maxLevels)) / 7;
    const uint64_t maxRanges = *maxStart < 8 * maxSplats ? *maxStart : 8 * maxSplats;
    *commandsSize = maxSplats * 8 + maxRanges * 2;
}

MLSGPU_API uint64_t mlsgpu_hip_tree_resource_usage(uint64_t maxLevels, uint64_t maxSplats)
{
    uint64_t maxStart, commandsSize;
    treeSizes(maxLevels, maxSplats, &maxStart, &commandsSize);
    const uint64_t entries = maxSplats * 8;
    /* start, jumpPos, node counts and bases; commands; keys and values x2; histogram, tile sums; slot masks; node tiles */
    return maxStart * 4 * 4 + commandsSize * 4 + entries * 4 * 4
        + sortHistElems(entries) * 4 + ((uint64_t) scanTiles(sortHistElems(entries) > entries ? sortHistElems(entries) : entries) + 1) * 4
        + 4 + maxSplats * 9 + ((uint64_t) scanTiles(maxStart) + 1) * sizeof(U3);
}

MLSGPU_API int mlsgpu_hip_tree_create(mlsgpu_ctx *ctx, uint64_t maxLevels, uint64_t maxSplats, mlsgpu_tree **out)
{
    REQUIRE(ctx != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(1 <= maxSplats && maxSplats <= MLSGPU_TREE_MAX_SPLATS, MLSGPU_ERR_LENGTH);   /* src/splat_tree_cl.cpp:106 */
    REQUIRE(1 <= maxLevels && maxLevels <= MLSGPU_TREE_MAX_LEVELS, MLSGPU_ERR_LENGTH);   /* :107 */
    HIP_CHECK(hipSetDevice(ctx->device));
    mlsgpu_tree *t = new mlsgpu_tree;
    t->ctx = ctx;
    t->maxLevels = maxLevels;
    t->maxSplats = maxSplats;
    treeSizes(maxLevels, maxSplats, &t->maxStart, &t->commandsSize);
    const uint64_t entries = maxSplats * 8;
    const uint64_t histElems = sortHistElems(entries);
    const uint64_t tileSums = scanTiles(histElems > entries ? histElems : entries) + 1;
    int rc = MLSGPU_OK;
    auto alloc = [&](void **p, uint64_t bytes) {
        if (rc == MLSGPU_OK && hipMalloc(p, bytes ? bytes : 4) != hipSuccess)
            rc = setError(MLSGPU_ERR_NOMEM, "SplatTreeCL: cannot allocate %llu bytes", (unsigned long long) bytes);
    };
    alloc((void **) &t->dStart, t->maxStart * 4);
    alloc((void **) &t->dJumpPos, t->maxStart * 4);
    alloc((void **) &t->dCommands, t->commandsSize * 4);
    alloc((void **) &t->dKeysA, entries * 4);
    alloc((void **) &t->dKeysB, entries * 4);
    alloc((void **) &t->dValsA, entries * 4);
    alloc((void **) &t->dValsB, entries * 4);
    alloc((void **) &t->dHist, histElems * 4);
    alloc((void **) &t->dTileSums, tileSums * 4);
    alloc((void **) &t->dNumEntries, 4);
    alloc((void **) &t->dSlotMasks, maxSplats);
    alloc((void **) &t->dEntryNotes, maxSplats * 8);
    alloc((void **) &t->dNodeCounts, t->maxStart * 4);
    alloc((void **) &t->dNodeBase, t->maxStart * 4);
    alloc((void **) &t->dDigitBase, 257 * 4);
    alloc((void **) &t->dNodeTiles, ((uint64_t) scanTiles(t->maxStart) + 1) * sizeof(U3));
    if (rc == MLSGPU_OK)
        rc = t->entryBox.create();
    if (rc != MLSGPU_OK)
    {
        mlsgpu_hip_tree_destroy(t);
        return rc;
    }
    *out = t;
    return MLSGPU_OK;
}

MLSGPU_API void mlsgpu_hip_tree_destroy(mlsgpu_tree *t)
{
    if (!t)
        return;
    hipSetDevice(t->ctx->device);
    hipFree(t->dStart); hipFree(t->dJumpPos); hipFree(t->dCommands);
    hipFree(t->dKeysA); hipFree(t->dKeysB); hipFree(t->dValsA); hipFree(t->dValsB);
    hipFree(t->dHist); hipFree(t->dTileSums); hipFree(t->dNumEntries); hipFree(t->dSlotMasks); hipFree(t->dEntryNotes);
    hipFree(t->dNodeCounts); hipFree(t->dNodeBase); hipFree(t->dNodeTiles); hipFree(t->dDigitBase);
    t->entryBox.destroy();
    delete t;
}

/* SplatTreeCL::enqueueBuild (src/splat_tree_cl.cpp:269-335) for the buckets of a batch in lock-step: the trees (one per
 * lane, same levels, one context) are built by ONE set of launches whose workgroups pick their lane by blockIdx.y; the
 * entry counts of all lanes come back through one mailbox publication.  A single build is a batch of one. */
static int treeBuildBatch(mlsgpu_tree *const *trees, const mlsgpu_tree_build *reqs, uint32_t count, uint32_t subsamplingShift)
{
    REQUIRE(trees != nullptr && reqs != nullptr && count >= 1 && count <= MAX_LANES, MLSGPU_ERR_INVALID);
    mlsgpu_tree *const t0 = trees[0];
    REQUIRE(t0 != nullptr, MLSGPU_ERR_INVALID);
    mlsgpu_ctx *ctx = t0->ctx;
    /* src/splat_tree_cl.cpp:277-281 */
    REQUIRE(t0->maxLevels + subsamplingShift - 1 < 31, MLSGPU_ERR_LENGTH);
    const uint32_t maxSize = 1u << (t0->maxLevels + subsamplingShift - 1);
    for (uint32_t k = 0; k < count; k++)
    {
        const mlsgpu_tree *t = trees[k];
        const mlsgpu_tree_build &r = reqs[k];
        REQUIRE(t != nullptr && r.dSplats != nullptr, MLSGPU_ERR_INVALID);
        REQUIRE(t->ctx == ctx && t->maxLevels == t0->maxLevels, MLSGPU_ERR_INVALID);
        for (uint32_t j = 0; j < k; j++)
            REQUIRE(trees[j] != t, MLSGPU_ERR_INVALID);
        REQUIRE(r.numSplats <= t->maxSplats, MLSGPU_ERR_LENGTH);
        REQUIRE(r.firstSplat < 0xFFFFFFFFull - r.numSplats, MLSGPU_ERR_LENGTH);
        REQUIRE(r.size[0] <= maxSize && r.size[1] <= maxSize && r.size[2] <= maxSize, MLSGPU_ERR_LENGTH);
    }
    const int maxShift = (int) (t0->maxLevels + subsamplingShift - 1);
    const int minShift = (int) subsamplingShift < maxShift ? (int) subsamplingShift : maxShift;
    HIP_CHECK(hipSetDevice(ctx->device));

    LevelOffsets lo;
    std::memset(&lo, 0, sizeof(lo));
    uint64_t pos = 0;
    for (int i = minShift; i <= maxShift; i++)
    {
        lo.v[i] = (uint32_t) pos;
        pos += uint64_t(1) << (3 * (maxShift - i));
    }
    const uint32_t numStart = (uint32_t) pos;
    const uint32_t keyBits = (uint32_t) (3 * (maxShift - minShift) + 1);
    const uint32_t passes = sortPasses(keyBits, SortCaps<uint32_t>::MAX_DIGIT_BITS);
    const uint32_t perPass = (keyBits + passes - 1) / passes;
    static const bool fusedOff = getenv("MLSGPU_HIP_OCTREE_FUSED") != nullptr && atoi(getenv("MLSGPU_HIP_OCTREE_FUSED")) == 0;
    /* wider digits (deep trees) take the separate passes */
    const bool fused = perPass <= ENT_BIN_BITS && keyBits <= ENT_KEY_BITS && !fusedOff;

    /* the command list straight from the sort's last pass (NodeIn / NodeOut above): for the fused front end with ONE pass left */
    const bool direct = fused && keyBits <= 2 * perPass;
    /* One word per entry between the two passes (entryScatterKernel / SortHistArgs): after the fused pass an entry's low
     * digit is its position, so the word holds the rest of the key above the splat's number inside the bucket -- when both
     * fit.  4 bytes less written and read again per entry (8 E of the 24 E the two passes move). */
    uint32_t idBits = 0;
    {
        const char *const env = getenv("MLSGPU_HIP_OCTREE_PACKED");      /* "0": two words per entry, as before (tests, A/B) */
        const bool packedOff = env != nullptr && atoi(env) == 0;
        uint64_t most = 1;
        for (uint32_t k = 0; k < count; k++)
            most = std::max<uint64_t>(most, reqs[k].numSplats);
        uint32_t bits = 1;
        while (bits < 32 && ((most - 1) >> bits) != 0)
            bits++;
        if (direct && !packedOff && (keyBits - perPass) + bits <= 32)
            idBits = bits;
    }
    Lanes<int32_t *> jump;
    Lanes<WriteStartArgs> ws;
    for (uint32_t k = 0; k < MAX_LANES; k++)
    {
        mlsgpu_tree *t = trees[k < count ? k : 0];
        if (k < count)
        {
            REQUIRE(numStart <= t->maxStart, MLSGPU_ERR_LENGTH);
            t->numLevels = (uint32_t) (maxShift - minShift + 1);
            t->dSplats = reqs[k].dSplats;
        }
        jump.a[k] = direct ? reinterpret_cast<int32_t *>(t->dNodeCounts) : t->dJumpPos;
        ws.a[k] = WriteStartArgs{t->dStart, t->dCommands, t->dJumpPos};
    }
    /* fill(jumpPos, -1), kernels/octree.cl:346 -- or, on the direct route, the node counters (NodeOut writes jumpPos) */
    LAUNCH(ctx, "kernel.octree.fill.time", fillKernel, dim3(divUp(numStart, 256), count), dim3(256), jump, numStart,
           (int32_t) (direct ? 0 : -1));

    /* the lanes that hold splats */
    uint32_t act[MAX_LANES], na = 0;
    for (uint32_t k = 0; k < count; k++)
        if (reqs[k].numSplats > 0)
            act[na++] = k;
    SortJob<uint32_t> sortJobs[MAX_LANES];
    uint64_t sortN[MAX_LANES];
    if (na > 0)
    {
        auto params = [&](uint32_t k) {
            const mlsgpu_tree_build &r = reqs[k];
            return EntryParams{r.dSplats, r.offset[0], r.offset[1], r.offset[2], lo, minShift, maxShift, (uint32_t) r.firstSplat,
                               trees[k]->mutate ? 1u : 0u};
        };
        if (fused)
        {
            /* writeEntries + the sort's first pass as one count / digit scan / scatter, see entryScatterKernel */
            const char *stat = "kernel.octree.writeEntries.time";
            Lanes<EntryHistArgs> eh;
            Lanes<SortDigitScanArgs> ds;
            Lanes<EntryTotalArgs> et;
            Lanes<EntryScatterArgs> es;
            uint32_t maxTiles = 0;
            for (uint32_t a = 0; a < MAX_LANES; a++)
            {
                const uint32_t k = act[a < na ? a : 0];
                mlsgpu_tree *t = trees[k];
                const uint32_t tilesE = a < na ? divUp(reqs[k].numSplats, ENT_TILE) : 0u;
                uint32_t *const dDigitTotals = t->dHist + (uint64_t) (1u << perPass) * divUp(reqs[k].numSplats, ENT_TILE);
                const EntryParams P = params(k);
                eh.a[a] = EntryHistArgs{P, t->dEntryNotes, t->dHist, tilesE, reqs[k].numSplats};
                ds.a[a] = SortDigitScanArgs{t->dHist, dDigitTotals, tilesE};
                et.a[a] = EntryTotalArgs{dDigitTotals, t->dNumEntries, t->dDigitBase};
                es.a[a] = EntryScatterArgs{P, t->dEntryNotes, t->dHist, dDigitTotals, tilesE, reqs[k].numSplats, t->dKeysB, t->dValsB, idBits};
                maxTiles = std::max(maxTiles, tilesE);
            }
            LAUNCH(ctx, stat, entryHistKernel, dim3(maxTiles, na), dim3(ENT_THREADS), eh, perPass);
            LAUNCH(ctx, stat, (sortDigitScanKernel<uint32_t>), dim3(1u << perPass, na), dim3(PRIM_BLOCK), ds);
            /* The entry counts (2.4 .. 3.8 per splat on the BASELINE clouds, 8 at most) come back to the host: the remaining
             * sort pass and the command scan launch on n instead of 8N elements. */
            const uint32_t seq = t0->entryBox.reserve();
            LAUNCH(ctx, stat, entryTotalKernel, dim3(1), dim3(256), et, na, 1u << perPass, t0->entryBox.dev, seq);
            LAUNCH(ctx, stat, entryScatterKernel, dim3(maxTiles, na), dim3(ENT_THREADS), es, perPass);
            PROPAGATE(t0->entryBox.wait(ctx->stream));
            for (uint32_t a = 0; a < na; a++)
            {
                mlsgpu_tree *t = trees[act[a]];
                sortN[a] = t0->entryBox.payload()[a];
                sortJobs[a] = SortJob<uint32_t>{t->dKeysB, t->dValsB, t->dKeysA, t->dValsA, sortN[a], t->dHist, t->dNumEntries,
                                                SortResult<uint32_t>{nullptr, nullptr}};
            }
            if (!direct)
                PROPAGATE(radixSortBatch<uint32_t>(ctx, "kernel.octree.sort.time", sortJobs, na, keyBits, false, perPass));
        }
        else
        {
            /* writeEntries: count, scan, write compacted; the entry count stays on the device (t->dNumEntries) */
            typedef ScanJob<uint32_t, EntryCountIn, EntryMaskIn, EntryWriteOut> EntryJob;
            EntryJob jobs[MAX_LANES];
            const void *counts[MAX_LANES];
            for (uint32_t a = 0; a < na; a++)
            {
                const uint32_t k = act[a];
                mlsgpu_tree *t = trees[k];
                const EntryParams P = params(k);
                jobs[a] = EntryJob{EntryCountIn{P, t->dSlotMasks}, EntryMaskIn{t->dSlotMasks},
                                   EntryWriteOut{P, t->dSlotMasks, t->dKeysA, t->dValsA}, reqs[k].numSplats, 0u, t->dTileSums,
                                   t->dNumEntries, nullptr};
                counts[a] = t->dNumEntries;
            }
            PROPAGATE((exclusiveScanBatch<uint32_t, EntryCountIn, EntryMaskIn, EntryWriteOut>(
                ctx, "kernel.octree.writeEntries.time", jobs, na)));
            PROPAGATE(t0->entryBox.publishGather(ctx->stream, counts, na, 1));
            PROPAGATE(t0->entryBox.wait(ctx->stream));
            for (uint32_t a = 0; a < na; a++)
            {
                mlsgpu_tree *t = trees[act[a]];
                sortN[a] = t0->entryBox.payload()[a];
                sortJobs[a] = SortJob<uint32_t>{t->dKeysA, t->dValsA, t->dKeysB, t->dValsB, sortN[a], t->dHist, t->dNumEntries,
                                                SortResult<uint32_t>{nullptr, nullptr}};
            }
            PROPAGATE(radixSortBatch<uint32_t>(ctx, "kernel.octree.sort.time", sortJobs, na, keyBits, false));
        }
        if (direct)
        {
            /* the sort's last pass: histogram + whole-key counts */
            const uint32_t shift = perPass, digitBits = keyBits - perPass;
            Lanes<SortHistArgs<uint32_t> > h;
            Lanes<SortDigitScanArgs> d;
            uint32_t maxTiles = 0;
            for (uint32_t a = 0; a < MAX_LANES; a++)
            {
                const SortJob<uint32_t> &j = sortJobs[a < na ? a : 0];
                mlsgpu_tree *t = trees[act[a < na ? a : 0]];
                const uint32_t tiles = a < na ? sortTiles(j.n) : 0u;
                uint32_t *const dDigitTotals = j.dHist + (uint64_t) SORT_MAX_BINS * sortTiles(j.n);
                h.a[a] = SortHistArgs<uint32_t>{j.keysA, j.dHist, j.n, j.nDev, tiles, t->dNodeCounts, t->dDigitBase, idBits};
                d.a[a] = SortDigitScanArgs{j.dHist, dDigitTotals, tiles};
                maxTiles = std::max(maxTiles, tiles);
            }
            if (maxTiles > 0)
            {
                LAUNCH(ctx, "kernel.octree.sort.time", (sortHistKernel<uint32_t, true>), dim3(maxTiles, na), dim3(PRIM_BLOCK), h, shift,
                       digitBits);
                LAUNCH(ctx, "kernel.octree.sort.time", (sortDigitScanKernel<uint32_t>), dim3(1u << digitBits, na), dim3(PRIM_BLOCK), d);
            }
        }
        else
        {
            /* countCommands + scan(seed 1) + writeSplatIds, src/splat_tree_cl.cpp:310-317 */
            typedef ScanJob<uint32_t, IndicatorIn, IndicatorIn, SplatIdsOut> CommandJob;
            CommandJob cj[MAX_LANES];
            for (uint32_t a = 0; a < na; a++)
            {
                mlsgpu_tree *t = trees[act[a]];
                const SortResult<uint32_t> &sorted = sortJobs[a].result;
                const IndicatorIn in{sorted.keys, t->dNumEntries};
                cj[a] = CommandJob{in, in, SplatIdsOut{t->dCommands, t->dStart, t->dJumpPos, sorted.keys, sorted.vals, t->dNumEntries},
                                   sortN[a], 1u, t->dTileSums, (uint32_t *) nullptr, t->dNumEntries};
            }
            PROPAGATE((exclusiveScanBatch<uint32_t, IndicatorIn, IndicatorIn, SplatIdsOut>(ctx, "kernel.octree.scan.time", cj, na)));
        }
    }
    if (direct)
    {
        /* the scan over the NODES of every lane (a lane without splats has no entries anywhere: jumpPos = -1 throughout) */
        typedef ScanJob<U3, NodeIn, NodeIn, NodeOut> NodeJob;
        NodeJob nj[MAX_LANES];
        for (uint32_t k = 0; k < count; k++)
        {
            mlsgpu_tree *t = trees[k];
            const NodeIn in{t->dNodeCounts};
            nj[k] = NodeJob{in, in, NodeOut{t->dStart, t->dJumpPos, t->dNodeBase}, numStart, U3{0, 0, 0}, t->dNodeTiles, (U3 *) nullptr,
                            nullptr};
        }
        PROPAGATE((exclusiveScanBatch<U3, NodeIn, NodeIn, NodeOut>(ctx, "kernel.octree.scan.time", nj, count)));
        if (na > 0)
        {
            /* ... and the last scatter: every id to its command position */
            const uint32_t shift = perPass, digitBits = keyBits - perPass;
            Lanes<SortScatterArgs<uint32_t> > sc;
            uint32_t maxTiles = 0;
            for (uint32_t a = 0; a < MAX_LANES; a++)
            {
                const SortJob<uint32_t> &j = sortJobs[a < na ? a : 0];
                mlsgpu_tree *t = trees[act[a < na ? a : 0]];
                const uint32_t tiles = a < na ? sortTiles(j.n) : 0u;
                uint32_t *const dDigitTotals = j.dHist + (uint64_t) SORT_MAX_BINS * sortTiles(j.n);
                sc.a[a] = SortScatterArgs<uint32_t>{j.keysA, j.valsA, (uint32_t *) nullptr, reinterpret_cast<uint32_t *>(t->dCommands), j.dHist,
                                                   dDigitTotals, j.n, j.nDev, tiles, t->dNodeBase, t->dDigitBase, idBits,
                                                   (uint32_t) reqs[act[a < na ? a : 0]].firstSplat};
                maxTiles = std::max(maxTiles, tiles);
            }
            if (maxTiles > 0)
                LAUNCH(ctx, "kernel.octree.sort.time", (sortScatterKernel<uint32_t, false, 8, true>), dim3(maxTiles, na), dim3(PRIM_BLOCK), sc,
                       shift, digitBits);
        }
    }
    LAUNCH(ctx, "kernel.octree.writeStart.time", writeStartKernel, dim3(divUp(numStart, 256), count), dim3(256),
           ws, lo, minShift, maxShift, numStart);
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_tree_build(mlsgpu_tree *t, mlsgpu_splat *dSplats, uint64_t firstSplat, uint64_t numSplats,
                                     const uint32_t size[3], const int32_t offset[3], uint32_t subsamplingShift)
{
    REQUIRE(t != nullptr && dSplats != nullptr && size != nullptr && offset != nullptr, MLSGPU_ERR_INVALID);
    mlsgpu_tree_build r;
    r.dSplats = dSplats;
    r.firstSplat = firstSplat;
    r.numSplats = numSplats;
    for (int i = 0; i < 3; i++)
    {
        r.size[i] = size[i];
        r.offset[i] = offset[i];
    }
    return treeBuildBatch(&t, &r, 1, subsamplingShift);
}

MLSGPU_API int mlsgpu_hip_tree_build_batch(mlsgpu_tree *const *trees, const mlsgpu_tree_build *builds, uint32_t count,
                                           uint32_t subsamplingShift)
{
    REQUIRE(trees != nullptr && builds != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(count >= 1 && count <= MLSGPU_MAX_BATCH, MLSGPU_ERR_LENGTH);
    return treeBuildBatch(trees, builds, count, subsamplingShift);
}

MLSGPU_API int mlsgpu_hip_tree_num_entries(mlsgpu_tree *t, uint64_t *out)
{
    REQUIRE(t != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    HIP_CHECK(hipSetDevice(t->ctx->device));
    uint32_t n = 0;
    HIP_CHECK(hipMemcpyAsync(&n, t->dNumEntries, 4, hipMemcpyDeviceToHost, t->ctx->stream));
    HIP_CHECK(hipStreamSynchronize(t->ctx->stream));
    *out = n;
    return MLSGPU_OK;
}

MLSGPU_API void mlsgpu_hip_tree_clear_splats(mlsgpu_tree *t) { if (t) t->dSplats = nullptr; }
MLSGPU_API const mlsgpu_splat *mlsgpu_hip_tree_splats(const mlsgpu_tree *t) { return t->dSplats; }
MLSGPU_API int mlsgpu_hip_tree_set_mutate(mlsgpu_tree *t, int mutate)
{
    REQUIRE(t != nullptr, MLSGPU_ERR_INVALID);
    t->mutate = mutate != 0;
    return MLSGPU_OK;
}
MLSGPU_API int mlsgpu_hip_tree_mutates(const mlsgpu_tree *t) { return t != nullptr && t->mutate ? 1 : 0; }
MLSGPU_API const int32_t *mlsgpu_hip_tree_commands(const mlsgpu_tree *t) { return t->dCommands; }
MLSGPU_API const int32_t *mlsgpu_hip_tree_start(const mlsgpu_tree *t) { return t->dStart; }
MLSGPU_API uint64_t mlsgpu_hip_tree_commands_size(const mlsgpu_tree *t) { return t->commandsSize; }
MLSGPU_API uint64_t mlsgpu_hip_tree_start_size(const mlsgpu_tree *t) { return t->maxStart; }
MLSGPU_API uint32_t mlsgpu_hip_tree_num_levels(const mlsgpu_tree *t) { return t->numLevels; }

MLSGPU_API int mlsgpu_hip_test_make_code(mlsgpu_ctx *ctx, int x, int y, int z, uint32_t *out)
{
    REQUIRE(ctx != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    int32_t a[3] = {x, y, z};
    uint32_t loopForm = 0, fast = 0;
    PROPAGATE(runTestHelper(ctx, 0, a, 3, nullptr, 0, &loopForm));
    if (x >= 0 && y >= 0 && z >= 0 && x < 1024 && y < 1024 && z < 1024)
    {
        PROPAGATE(runTestHelper(ctx, 3, a, 3, nullptr, 0, &fast));
        if (fast != loopForm)
            return setError(MLSGPU_ERR_INVALID, "makeCode fast path disagrees with loop form");
    }
    *out = loopForm;
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_test_level_shift(mlsgpu_ctx *ctx, const int32_t lo[3], const int32_t hi[3], int32_t *out)
{
    REQUIRE(ctx != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    int32_t a[6] = {lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]};
    uint32_t r = 0;
    PROPAGATE(runTestHelper(ctx, 1, a, 6, nullptr, 0, &r));
    *out = (int32_t) r;
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_test_point_box_dist2(mlsgpu_ctx *ctx, const float p[3], const float lo[3], const float hi[3],
                                               float *out)
{
    REQUIRE(ctx != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    float a[9] = {p[0], p[1], p[2], lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]};
    uint32_t r = 0;
    PROPAGATE(runTestHelper(ctx, 2, nullptr, 0, a, 9, &r));
    std::memcpy(out, &r, 4);
    return MLSGPU_OK;
}

/* ---- primitive tests ---- */
MLSGPU_API int mlsgpu_hip_test_scan_u32(mlsgpu_ctx *ctx, uint32_t *dData, uint64_t n, uint32_t seed)
{
    REQUIRE(ctx != nullptr, MLSGPU_ERR_INVALID);
    HIP_CHECK(hipSetDevice(ctx->device));
    uint32_t *dTiles = nullptr;
    HIP_CHECK(hipMalloc(&dTiles, ((uint64_t) scanTiles(n) + 1) * 4));
    int rc = exclusiveScan<uint32_t>(ctx, "test.scan", ArrayIn<uint32_t>{dData}, ArrayOut<uint32_t>{dData}, n, seed,
                                     dTiles, (uint32_t *) nullptr);
    hipStreamSynchronize(ctx->stream);
    hipFree(dTiles);
    return rc;
}

/* `repeats` scans of `count` lanes each (one set of launches per repeat, as the buckets of a batch), dIn[k] -> dOut[k],
 * enqueued back to back without a host synchronisation in between; returns when the last one has finished */
MLSGPU_API int mlsgpu_hip_test_scan_u32_batch(mlsgpu_ctx *ctx, const uint32_t *const *dIn, uint32_t *const *dOut, const uint64_t *n,
                                              const uint32_t *seeds, uint32_t count, uint32_t repeats)
{
    REQUIRE(ctx != nullptr && dIn != nullptr && dOut != nullptr && n != nullptr && seeds != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(count >= 1 && count <= MAX_LANES, MLSGPU_ERR_LENGTH);
    HIP_CHECK(hipSetDevice(ctx->device));
    uint32_t *dTiles[MAX_LANES] = {};
    typedef ScanJob<uint32_t, ArrayIn<uint32_t>, ArrayIn<uint32_t>, ArrayOut<uint32_t> > Job;
    Job jobs[MAX_LANES];
    int rc = MLSGPU_OK;
    for (uint32_t k = 0; k < count && rc == MLSGPU_OK; k++)
    {
        if (hipMalloc(&dTiles[k], ((uint64_t) scanTiles(n[k]) + 1) * 4) != hipSuccess)
            rc = setError(MLSGPU_ERR_NOMEM, "hipMalloc failed");
        jobs[k] = Job{ArrayIn<uint32_t>{dIn[k]}, ArrayIn<uint32_t>{dIn[k]}, ArrayOut<uint32_t>{dOut[k]}, n[k], seeds[k], dTiles[k],
                      (uint32_t *) nullptr, (const uint32_t *) nullptr};
    }
    for (uint32_t r = 0; r < repeats && rc == MLSGPU_OK; r++)
        rc = exclusiveScanBatch<uint32_t>(ctx, "test.scan", jobs, count);
    hipStreamSynchronize(ctx->stream);
    for (uint32_t k = 0; k < count; k++)
        hipFree(dTiles[k]);
    return rc;
}

template<typename K>
static int testSort(mlsgpu_ctx *ctx, K *dKeys, uint32_t *dValues, uint64_t n, uint32_t bits)
{
    HIP_CHECK(hipSetDevice(ctx->device));
    K *kb = nullptr;
    uint32_t *vb = nullptr, *hist = nullptr, *tiles = nullptr;
    HIP_CHECK(hipMalloc(&kb, (n + 1) * sizeof(K)));
    HIP_CHECK(hipMalloc(&vb, (n + 1) * 4));
    HIP_CHECK(hipMalloc(&hist, (sortHistElems(n) + 1) * 4));
    HIP_CHECK(hipMalloc(&tiles, ((uint64_t) scanTiles(sortHistElems(n)) + 1) * 4));
    SortResult<K> res;
    int rc = radixSort<K>(ctx, "test.sort", dKeys, dValues, kb, vb, n, bits, false, hist, tiles, &res);
    if (rc == MLSGPU_OK && res.keys != dKeys && n > 0)
    {
        hipMemcpyAsync(dKeys, res.keys, n * sizeof(K), hipMemcpyDeviceToDevice, ctx->stream);
        hipMemcpyAsync(dValues, res.vals, n * 4, hipMemcpyDeviceToDevice, ctx->stream);
    }
    hipStreamSynchronize(ctx->stream);
    hipFree(kb); hipFree(vb); hipFree(hist); hipFree(tiles);
    return rc;
}

MLSGPU_API int mlsgpu_hip_test_sort_u32(mlsgpu_ctx *ctx, uint32_t *dKeys, uint32_t *dValues, uint64_t n, uint32_t bits)
{
    REQUIRE(ctx != nullptr, MLSGPU_ERR_INVALID);
    return testSort<uint32_t>(ctx, dKeys, dValues, n, bits);
}

MLSGPU_API int mlsgpu_hip_test_sort_u64(mlsgpu_ctx *ctx, uint64_t *dKeys, uint32_t *dValues, uint64_t n, uint32_t bits)
{
    REQUIRE(ctx != nullptr, MLSGPU_ERR_INVALID);
    return testSort<uint64_t>(ctx, dKeys, dValues, n, bits);
}
