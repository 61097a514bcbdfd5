/*
 * Device-wide primitives of the HIP path, replacing third-party clogs 1.1 in the reference:
 *  - exclusive prefix sum with a seed (clogs::Scan; call sites src/splat_tree_cl.cpp:313,
 *    src/marching.cpp:583,721), as reduce -> scan of tile sums -> apply, with the element
 *    producer and consumer fused in as functors so indicator / map arrays never hit HBM;
 *  - stable LSD radix sort of (key, uint32 value) pairs on the low `bits` of the key
 *    (clogs::Radixsort; src/splat_tree_cl.cpp:308, src/marching.cpp:572), ranking with
 *    wave64 ballots so each wave's 64 keys are split stably without an LDS sort.
 * Everything is integer and bit-exact by construction.
 *
 * Element order inside a tile: wave w owns PRIM_WAVE_SPAN contiguous elements and walks them
 * in PRIM_ITEMS rounds of 64 contiguous elements (lane = element), so every global access is a
 * fully coalesced 256-byte (u32) or 512-byte (u64) wave access and memory order == scan order.
 */
#ifndef MLSGPU_AMD_PRIMITIVES_HPP
#define MLSGPU_AMD_PRIMITIVES_HPP

#include "common.hpp"

#include <cstdlib>

namespace mlsgpu
{

enum
{
    PRIM_BLOCK = 256,                       /* threads per block (4 waves) */
    PRIM_WAVES = PRIM_BLOCK / 64,
    PRIM_ITEMS = 8,                         /* rounds of 64 contiguous elements per wave */
    PRIM_TILE = PRIM_BLOCK * PRIM_ITEMS,    /* elements per block */
    PRIM_WAVE_SPAN = 64 * PRIM_ITEMS,       /* contiguous elements per wave */
    /* the sort's tiles are twice the scans': measured on cfg3 (ms per step) scans 2048 / sort 4096 elements per workgroup:
     * writeEntries 2.08, octree.scan 1.41, marching scans 1.12, octree.sort 4.2; all at 4096: 2.41, 1.76, 1.45, 4.2; all at
     * 2048: sort 4.35 */
    SORT_ITEMS = 16,
    SORT_TILE = PRIM_BLOCK * SORT_ITEMS,
    SORT_WAVE_SPAN = 64 * SORT_ITEMS,
    SORT_MAX_DIGIT_BITS = 10,
    SORT_MAX_BINS = 1 << SORT_MAX_DIGIT_BITS
};

/* (occupied cells, vertices, indices) triple scanned by Marching */
struct U3
{
    uint32_t a, b, c;
};

#ifdef __HIPCC__

__host__ __device__ __forceinline__ U3 operator+(const U3 &x, const U3 &y) { return U3{x.a + y.a, x.b + y.b, x.c + y.c}; }
__host__ __device__ __forceinline__ uint32_t zeroOf(uint32_t) { return 0u; }
__host__ __device__ __forceinline__ U3 zeroOf(U3) { return U3{0u, 0u, 0u}; }

__device__ __forceinline__ uint32_t waveInclusiveScanT(uint32_t v) { return waveInclusiveScan(v); }
__device__ __forceinline__ U3 waveInclusiveScanT(U3 v)
{
    return U3{waveInclusiveScan(v.a), waveInclusiveScan(v.b), waveInclusiveScan(v.c)};
}
/* value of lane-1; lane 0 gets zero */
__device__ __forceinline__ uint32_t waveShiftUpT(uint32_t v) { return waveShiftUp1(v); }
__device__ __forceinline__ U3 waveShiftUpT(U3 v) { return U3{waveShiftUpT(v.a), waveShiftUpT(v.b), waveShiftUpT(v.c)}; }
__device__ __forceinline__ uint32_t readLaneT(uint32_t v, int lane) { return readLane(v, lane); }
__device__ __forceinline__ U3 readLaneT(U3 v, int lane)
{
    return U3{readLane(v.a, lane), readLane(v.b, lane), readLane(v.c, lane)};
}

/* ------------------------------------------------------------------ scan */

/* tileSums[b] = sum of in(i) over tile b */
template<typename T, typename In>
__global__ __launch_bounds__(PRIM_BLOCK) void scanReduceKernel(In in, T *tileSums, uint64_t n, const uint32_t *nDev)
{
    if (nDev != nullptr && *nDev < n)
        n = *nDev;      /* element count produced on the device; the grid covers the upper bound */
    __shared__ T waveTotals[PRIM_WAVES];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t) blockIdx.x * PRIM_TILE + (uint64_t) wave * PRIM_WAVE_SPAN + lane;
    T sum = zeroOf(T());
#pragma unroll 4
    for (int j = 0; j < PRIM_ITEMS; j++)
    {
        uint64_t i = base + (uint64_t) j * 64;
        if (i < n)
            sum = sum + in(i);
    }
    T incl = waveInclusiveScanT(sum);
    if (lane == 63)
        waveTotals[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        T t = waveTotals[0];
        for (int w = 1; w < PRIM_WAVES; w++)
            t = t + waveTotals[w];
        tileSums[blockIdx.x] = t;
    }
}

/* Single block: exclusive scan of the tile sums in place, starting at `seed`; grand total (incl. seed) to *total. */
template<typename T>
__global__ __launch_bounds__(PRIM_BLOCK) void scanTileSumsKernel(T *tileSums, uint32_t numTiles, T seed, T *total)
{
    __shared__ T waveTotals[PRIM_WAVES];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T carry = seed;
    for (uint32_t base = 0; base < numTiles; base += PRIM_BLOCK)
    {
        const uint32_t i = base + threadIdx.x;
        T v = i < numTiles ? tileSums[i] : zeroOf(T());
        T incl = waveInclusiveScanT(v);
        if (lane == 63)
            waveTotals[wave] = incl;
        __syncthreads();
        T before = carry;
        T all = carry;
        for (uint32_t w = 0; w < PRIM_WAVES; w++)
        {
            if (w < wave)
                before = before + waveTotals[w];
            all = all + waveTotals[w];
        }
        T excl = before + waveShiftUpT(incl);
        if (i < numTiles)
            tileSums[i] = excl;
        carry = all;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total != nullptr)
        *total = carry;
}

/* out(i, exclusivePrefix(i), in(i)) for every i < n.
 * FUSED = false: tileSums already hold each tile's exclusive prefix (scanTileSumsKernel ran).
 * FUSED = true:  tileSums are the raw tile sums; every workgroup adds up its predecessors' itself (at most
 *                SCAN_FUSED_MAX_TILES L2-resident values), which saves the single-workgroup launch in between;
 *                the last workgroup writes the grand total. */
template<typename T, typename In, typename Out, bool FUSED>
__global__ __launch_bounds__(PRIM_BLOCK) void scanApplyKernel(In in, Out out, const T *tileSums, uint64_t n, const uint32_t *nDev,
                                                              T seed, T *total)
{
    if (nDev != nullptr && *nDev < n)
        n = *nDev;
    __shared__ T waveTotals[PRIM_WAVES];
    __shared__ T wavePrefix[PRIM_WAVES];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t) blockIdx.x * PRIM_TILE + (uint64_t) wave * PRIM_WAVE_SPAN + lane;
    /* the tile holding the last element (tile 0 of an empty scan) reports the total; later tiles are empty */
    const uint32_t lastTile = n > 0 ? (uint32_t) ((n - 1) / PRIM_TILE) : 0u;
    if (FUSED && blockIdx.x > lastTile)
        return;
    T before = zeroOf(T());
    if (FUSED)
    {
        /* the predecessors' sums are requested first: their latency hides behind the tile's own loads */
        for (uint32_t t = threadIdx.x; t < blockIdx.x; t += PRIM_BLOCK)
            before = before + tileSums[t];
    }
    T vals[PRIM_ITEMS];
    T excl[PRIM_ITEMS];
    T running = zeroOf(T());        /* wave-uniform: sum of the previous rounds of this wave */
#pragma unroll
    for (int j = 0; j < PRIM_ITEMS; j++)
    {
        uint64_t i = base + (uint64_t) j * 64;
        vals[j] = i < n ? in(i) : zeroOf(T());
        T incl = waveInclusiveScanT(vals[j]);
        excl[j] = running + waveShiftUpT(incl);
        running = running + readLaneT(incl, 63);
    }
    if (FUSED)
    {
        const T inclBefore = waveInclusiveScanT(before);
        if (lane == 63)
            wavePrefix[wave] = inclBefore;
    }
    if (lane == 0)
        waveTotals[wave] = running;
    __syncthreads();
    if (FUSED)
    {
        before = seed;
#pragma unroll
        for (int w = 0; w < PRIM_WAVES; w++)
            before = before + wavePrefix[w];
        if (total != nullptr && blockIdx.x == lastTile && threadIdx.x == 0)
        {
            T all = before;
#pragma unroll
            for (int w = 0; w < PRIM_WAVES; w++)
                all = all + waveTotals[w];
            *total = all;
        }
    }
    else
        before = tileSums[blockIdx.x];
    for (uint32_t w = 0; w < wave; w++)
        before = before + waveTotals[w];
#pragma unroll
    for (int j = 0; j < PRIM_ITEMS; j++)
    {
        uint64_t i = base + (uint64_t) j * 64;
        if (i < n)
            out(i, before + excl[j], vals[j]);
    }
}

#endif /* __HIPCC__ */

/* Host driver.  Workspace: tileSums must hold scanTiles(n) elements of T. */
static inline uint32_t scanTiles(uint64_t n) { return divUp(n, PRIM_TILE); }

#ifdef __HIPCC__
/* Phase 1: tile sums of in() scanned from `seed`; grand total (incl. seed) to *dTotal (may be null). */
template<typename T, typename In>
static int scanPhase1(mlsgpu_ctx *ctx, const char *statName, In in, uint64_t n, T seed, T *dTileSums, T *dTotal,
                      const uint32_t *nDev = nullptr)
{
    const uint32_t tiles = scanTiles(n);
    if (tiles > 0)
        LAUNCH(ctx, statName, (scanReduceKernel<T, In>), dim3(tiles), dim3(PRIM_BLOCK), in, dTileSums, n, nDev);
    LAUNCH(ctx, statName, (scanTileSumsKernel<T>), dim3(1), dim3(PRIM_BLOCK), dTileSums, tiles, seed, dTotal);
    return MLSGPU_OK;
}

/* Phase 2: out(i, exclusivePrefix(i), in(i)) for all i, using the tile sums of phase 1. */
template<typename T, typename In, typename Out>
static int scanPhase2(mlsgpu_ctx *ctx, const char *statName, In in, Out out, uint64_t n, const T *dTileSums,
                      const uint32_t *nDev = nullptr)
{
    const uint32_t tiles = scanTiles(n);
    if (tiles > 0)
        LAUNCH(ctx, statName, (scanApplyKernel<T, In, Out, false>), dim3(tiles), dim3(PRIM_BLOCK), in, out, dTileSums, n, nDev,
               zeroOf(T()), (T *) nullptr);
    return MLSGPU_OK;
}

/* largest scan (in tiles) whose workgroups add up their predecessors' tile sums themselves */
#define SCAN_FUSED_MAX_TILES 4096u

/* in1 feeds the tile sums, in2 the scan proper (they must agree; a first pass may cache what the second reads) */
template<typename T, typename In1, typename In2, typename Out>
static int exclusiveScan2(mlsgpu_ctx *ctx, const char *statName, In1 in1, In2 in2, Out out, uint64_t n, T seed,
                          T *dTileSums, T *dTotal, const uint32_t *nDev = nullptr)
{
    const uint32_t tiles = scanTiles(n);
    if (tiles > 0 && tiles <= SCAN_FUSED_MAX_TILES)
    {
        /* two launches: raw tile sums, then the scan proper */
        LAUNCH(ctx, statName, (scanReduceKernel<T, In1>), dim3(tiles), dim3(PRIM_BLOCK), in1, dTileSums, n, nDev);
        LAUNCH(ctx, statName, (scanApplyKernel<T, In2, Out, true>), dim3(tiles), dim3(PRIM_BLOCK), in2, out, (const T *) dTileSums,
               n, nDev, seed, dTotal);
        return MLSGPU_OK;
    }
    PROPAGATE((scanPhase1<T, In1>(ctx, statName, in1, n, seed, dTileSums, dTotal, nDev)));
    return scanPhase2<T, In2, Out>(ctx, statName, in2, out, n, (const T *) dTileSums, nDev);
}

template<typename T, typename In, typename Out>
static int exclusiveScan(mlsgpu_ctx *ctx, const char *statName, In in, Out out, uint64_t n, T seed,
                         T *dTileSums, T *dTotal, const uint32_t *nDev = nullptr)
{
    return exclusiveScan2<T, In, In, Out>(ctx, statName, in, in, out, n, seed, dTileSums, dTotal, nDev);
}

/* plain array in / array out functors */
template<typename T>
struct ArrayIn
{
    const T *p;
    __device__ __forceinline__ T operator()(uint64_t i) const { return p[i]; }
};
template<typename T>
struct ArrayOut
{
    T *p;
    __device__ __forceinline__ void operator()(uint64_t i, T excl, T) const { p[i] = excl; }
};

/* ------------------------------------------------------------------ radix sort */

template<typename K>
__global__ __launch_bounds__(PRIM_BLOCK) void sortHistKernel(const K *keys, uint32_t *hist, uint64_t n,
                                                             uint32_t shift, uint32_t digitBits, uint32_t numTiles,
                                                             const uint32_t *nDev)
{
    if (nDev != nullptr && *nDev < n)
        n = *nDev;
    __shared__ uint32_t bins[SORT_MAX_BINS];
    const uint32_t numBins = 1u << digitBits;
    const K mask = (K) (numBins - 1);
    for (uint32_t d = threadIdx.x; d < numBins; d += PRIM_BLOCK)
        bins[d] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t) blockIdx.x * SORT_TILE + (uint64_t) wave * SORT_WAVE_SPAN + lane;
#pragma unroll 4
    for (int j = 0; j < SORT_ITEMS; j++)
    {
        uint64_t i = base + (uint64_t) j * 64;
        if (i < n)
            atomicAdd(&bins[(uint32_t) ((keys[i] >> shift) & mask)], 1u);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < numBins; d += PRIM_BLOCK)
        hist[(uint64_t) d * numTiles + blockIdx.x] = bins[d];
}

/* hist[d][0 .. numTiles) -> its exclusive prefix along the tiles, in place; digitTotals[d] = the digit's count.  One
 * workgroup per digit: the whole scan of a pass's histogram is this ONE small launch (the scatter kernel turns the 2^bits
 * digit totals into digit bases itself), where a generic scan of the 2^bits x numTiles array took two. */
template<typename T>    /* T = uint32_t; a template only so that the header can be included by several translation units */
__global__ __launch_bounds__(PRIM_BLOCK) void sortDigitScanKernel(T *hist, T *digitTotals, uint32_t numTiles)
{
    __shared__ T waveTotals[PRIM_WAVES];
    T *row = hist + (uint64_t) blockIdx.x * numTiles;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < numTiles; base += PRIM_BLOCK)
    {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < numTiles ? row[i] : 0u;
        const uint32_t incl = waveInclusiveScan(v);
        if (lane == 63)
            waveTotals[wave] = incl;
        __syncthreads();
        uint32_t before = carry, all = carry;
#pragma unroll
        for (uint32_t w = 0; w < PRIM_WAVES; w++)
        {
            if (w < wave)
                before += waveTotals[w];
            all += waveTotals[w];
        }
        if (i < numTiles)
            row[i] = before + incl - v;
        carry = all;
        __syncthreads();
    }
    if (threadIdx.x == 0)
        digitTotals[blockIdx.x] = carry;
}

/* Largest digit a key type can be sorted by per pass: bounded by the LDS the scatter kernel needs
 * (tile of keys + values, per-wave bins).  64 KB of static LDS per workgroup. */
template<typename K> struct SortCaps { enum { MAX_DIGIT_BITS = sizeof(K) == 8 ? 9 : SORT_MAX_DIGIT_BITS }; };

/*
 * Scatter pass.  hist[d][tile] = keys with digit d in the tiles before this one (sortDigitScanKernel), digitTotals[d] =
 * keys with digit d: the digit's base is the exclusive prefix of the totals, computed here by every workgroup (2^bits
 * values from L2), so hist[d][tile] + base[d] is where this tile's keys with digit d start in the output.
 *   1. per-wave digit counts (LDS atomics);
 *   2. tile-local start of every digit (block scan over the bins) and of every (wave, digit);
 *   3. stable rank of each key among the keys of its wave with the same digit, by wave64 ballots
 *      (one __ballot per digit bit gives the peer mask, mbcnt the rank), rounds in order: the key's
 *      position inside the TILE-LOCALLY SORTED sequence;
 *   4. the tile is assembled in that order in LDS and copied out by consecutive threads, so each
 *      digit's run leaves as one contiguous, coalesced burst instead of 64 scattered dwords per store.
 */
template<typename K, bool IOTA, int BIN_BITS>
__global__ __launch_bounds__(PRIM_BLOCK) void sortScatterKernel(const K *keysIn, const uint32_t *valsIn,
                                                                K *keysOut, uint32_t *valsOut,
                                                                const uint32_t *hist, const uint32_t *digitTotals,
                                                                uint64_t n, uint32_t shift, uint32_t digitBits,
                                                                uint32_t numTiles, const uint32_t *nDev)
{
    enum { BINS = 1 << BIN_BITS };
    __shared__ uint32_t waveBins[PRIM_WAVES][BINS];
    __shared__ uint32_t tileBase[BINS];        /* global start of the digit minus its tile-local start */
    __shared__ uint32_t waveTotals[PRIM_WAVES], waveTotalsAll[PRIM_WAVES];
    /* the tile is reordered in two phases through ONE buffer (keys, then values): half the LDS, twice the
     * resident workgroups, which is what this latency-bound kernel needs */
    __shared__ K sTile[SORT_TILE];
    if (nDev != nullptr && *nDev < n)
        n = *nDev;
    const uint64_t tileFirst = (uint64_t) blockIdx.x * SORT_TILE;
    if (tileFirst >= n)
        return;
    const uint32_t tileCount = (uint32_t) (n - tileFirst < SORT_TILE ? n - tileFirst : SORT_TILE);
    const uint32_t numBins = 1u << digitBits;
    const K mask = (K) (numBins - 1);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t d = threadIdx.x; d < numBins; d += PRIM_BLOCK)
    {
#pragma unroll
        for (int w = 0; w < PRIM_WAVES; w++)
            waveBins[w][d] = 0;
    }
    __syncthreads();
    const uint64_t base = tileFirst + (uint64_t) wave * SORT_WAVE_SPAN + lane;
    K keys[SORT_ITEMS];
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++)
    {
        uint64_t i = base + (uint64_t) j * 64;
        keys[j] = i < n ? keysIn[i] : (K) 0;
        if (i < n)
            atomicAdd(&waveBins[wave][(uint32_t) ((keys[j] >> shift) & mask)], 1u);
    }
    __syncthreads();
    /* tile-local exclusive prefix over the digits: thread t owns the `per` consecutive bins from t * per */
    {
        const uint32_t per = numBins > PRIM_BLOCK ? numBins / PRIM_BLOCK : 1;
        const uint32_t d0 = threadIdx.x * per;
        uint32_t mine = 0, mineAll = 0;         /* this tile's / the whole input's keys in my bins */
        if (d0 < numBins)
            for (uint32_t k = 0; k < per; k++)
            {
                mineAll += digitTotals[d0 + k];
#pragma unroll
                for (int w = 0; w < PRIM_WAVES; w++)
                    mine += waveBins[w][d0 + k];
            }
        const uint32_t incl = waveInclusiveScan(mine), inclAll = waveInclusiveScan(mineAll);
        if (lane == 63)
        {
            waveTotals[wave] = incl;
            waveTotalsAll[wave] = inclAll;
        }
        __syncthreads();
        uint32_t run = incl - mine, base = inclAll - mineAll;
        for (uint32_t w = 0; w < wave; w++)
        {
            run += waveTotals[w];
            base += waveTotalsAll[w];
        }
        if (d0 < numBins)
            for (uint32_t k = 0; k < per; k++)
            {
                const uint32_t d = d0 + k;
                tileBase[d] = base + hist[(uint64_t) d * numTiles + blockIdx.x] - run;
                base += digitTotals[d];
#pragma unroll
                for (int w = 0; w < PRIM_WAVES; w++)
                {
                    const uint32_t c = waveBins[w][d];
                    waveBins[w][d] = run;
                    run += c;
                }
            }
    }
    __syncthreads();
    /* stable split of each round: rank among the lanes of the wave holding the same digit */
    uint32_t dst[SORT_ITEMS];
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++)
    {
        const uint64_t i = base + (uint64_t) j * 64;
        const bool valid = i < n;
        const uint32_t digit = (uint32_t) ((keys[j] >> shift) & mask);
        uint64_t peers = __ballot(valid);
        for (uint32_t b = 0; b < digitBits; b++)
        {
            const bool bit = (digit >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        dst[j] = 0;
        if (valid)
        {
            const uint32_t rank = popcBelow(peers);
            dst[j] = waveBins[wave][digit] + rank;
            sTile[dst[j]] = keys[j];
            if (rank == 0)
                waveBins[wave][digit] = dst[j] + (uint32_t) __popcll(peers);
        }
        /* LDS operations of one wave complete in program order, so the next round sees the update */
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    /* keys leave in tile-sorted order: each digit's run is one contiguous, coalesced burst */
    uint32_t out[SORT_ITEMS];
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++)
    {
        const uint32_t p = threadIdx.x + k * PRIM_BLOCK;
        out[k] = 0;
        if (p < tileCount)
        {
            const K key = sTile[p];
            out[k] = tileBase[(uint32_t) ((key >> shift) & mask)] + p;
            keysOut[out[k]] = key;
        }
    }
    __syncthreads();
    /* the values take the same route through the same buffer */
    uint32_t *sVals = reinterpret_cast<uint32_t *>(sTile);
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++)
    {
        const uint64_t i = base + (uint64_t) j * 64;
        if (i < n)
            sVals[dst[j]] = IOTA ? (uint32_t) i : valsIn[i];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++)
    {
        const uint32_t p = threadIdx.x + k * PRIM_BLOCK;
        if (p < tileCount)
            valsOut[out[k]] = sVals[p];
    }
}

/* Where the sorted data ends up. */
template<typename K>
struct SortResult
{
    K *keys;
    uint32_t *vals;
};

static inline uint32_t sortPasses(uint32_t bits, uint32_t maxDigitBits) { return (bits + maxDigitBits - 1) / maxDigitBits; }
/* elements of uint32 needed for the histogram of a sort of n keys: 2^bits x tiles counters and, behind them, the 2^bits
 * digit totals of the pass */
static inline uint32_t sortTiles(uint64_t n) { return divUp(n, SORT_TILE); }
static inline uint64_t sortHistElems(uint64_t n) { return (uint64_t) SORT_MAX_BINS * ((uint64_t) sortTiles(n) + 1); }

/*
 * Sorts (keysA, valsA)[0..n) stably by key bits [0, bits).  keysB/valsB are same-sized temporaries
 * (the reference aliases its sort temporaries onto other buffers the same way,
 * src/splat_tree_cl.cpp:129, src/marching.cpp:405).  iota: values are 0..n-1 and valsA is not read.
 * dHist: sortHistElems(n) uint32; dTileSums: not used any more (kept for the call sites' sake).
 */
template<typename K>
static int radixSort(mlsgpu_ctx *ctx, const char *statName, K *keysA, uint32_t *valsA, K *keysB, uint32_t *valsB,
                     uint64_t n, uint32_t bits, bool iota, uint32_t *dHist, uint32_t *dTileSums,
                     SortResult<K> *result, const uint32_t *nDev = nullptr, uint32_t doneBits = 0)
{
    /* doneBits: the lowest doneBits key bits have been sorted already (by a pass of that width fused into the producer of
     * the keys); the remaining passes keep the split of the whole sort */
    result->keys = keysA;
    result->vals = valsA;
    if (n == 0)
        return MLSGPU_OK;
    const uint32_t tiles = sortTiles(n);
    (void) dTileSums;
    uint32_t *const dDigitTotals = dHist + (uint64_t) SORT_MAX_BINS * tiles;
    if (bits == 0)
        bits = 1;    /* still run one pass so that iota values are materialised */
    uint32_t maxDigit = SortCaps<K>::MAX_DIGIT_BITS;
    if (const char *e = getenv("MLSGPU_HIP_SORT_DIGIT_BITS"))     /* tuning aid: narrower digits, more passes */
    {
        const uint32_t v = (uint32_t) atoi(e);
        if (v >= 1 && v < maxDigit)
            maxDigit = v;
    }
    const uint32_t passes = sortPasses(bits, maxDigit);
    const uint32_t perPass = (bits + passes - 1) / passes;
    uint32_t shift = doneBits;
    K *kin = keysA, *kout = keysB;
    uint32_t *vin = valsA, *vout = valsB;
    for (uint32_t p = 0; shift < bits; p++)
    {
        const uint32_t digitBits = (bits - shift) < perPass ? (bits - shift) : perPass;
        LAUNCH(ctx, statName, (sortHistKernel<K>), dim3(tiles), dim3(PRIM_BLOCK),
               (const K *) kin, dHist, n, shift, digitBits, tiles, nDev);
        LAUNCH(ctx, statName, (sortDigitScanKernel<uint32_t>), dim3(1u << digitBits), dim3(PRIM_BLOCK), dHist, dDigitTotals, tiles);
#define SORT_SCATTER(IOTA, BITS)                                                                                       \
        LAUNCH(ctx, statName, (sortScatterKernel<K, IOTA, BITS>), dim3(tiles), dim3(PRIM_BLOCK), (const K *) kin,      \
               (const uint32_t *) vin, kout, vout, (const uint32_t *) dHist, (const uint32_t *) dDigitTotals, n, shift,\
               digitBits, tiles, nDev)
        /* the kernel's bin tables are sized for the digit in use: fewer bins, more resident workgroups */
        const bool first = iota && p == 0;
        if (digitBits <= 8) { if (first) SORT_SCATTER(true, 8); else SORT_SCATTER(false, 8); }
        else if (digitBits <= 9) { if (first) SORT_SCATTER(true, 9); else SORT_SCATTER(false, 9); }
        else { if (first) SORT_SCATTER(true, SortCaps<K>::MAX_DIGIT_BITS); else SORT_SCATTER(false, SortCaps<K>::MAX_DIGIT_BITS); }
#undef SORT_SCATTER
        shift += digitBits;
        K *tk = kin; kin = kout; kout = tk;
        uint32_t *tv = vin; vin = vout; vout = tv;
    }
    result->keys = kin;
    result->vals = vin;
    return MLSGPU_OK;
}
#endif /* __HIPCC__ */

} // namespace mlsgpu
#endif
