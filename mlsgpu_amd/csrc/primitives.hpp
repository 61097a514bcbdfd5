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
    PRIM_ITEMS = 16,                        /* rounds of 64 contiguous elements per wave */
    PRIM_TILE = PRIM_BLOCK * PRIM_ITEMS,    /* elements per block */
    PRIM_WAVE_SPAN = 64 * PRIM_ITEMS,       /* contiguous elements per wave */
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
__device__ __forceinline__ uint32_t waveShiftUpT(uint32_t v)
{
    uint32_t t = __shfl_up(v, 1, 64);
    return laneId() == 0 ? 0u : t;
}
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

/* out(i, exclusivePrefix(i), in(i)) for every i < n; tileSums already hold each tile's exclusive prefix */
template<typename T, typename In, typename Out>
__global__ __launch_bounds__(PRIM_BLOCK) void scanApplyKernel(In in, Out out, const T *tileSums, uint64_t n, const uint32_t *nDev)
{
    if (nDev != nullptr && *nDev < n)
        n = *nDev;
    __shared__ T waveTotals[PRIM_WAVES];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t) blockIdx.x * PRIM_TILE + (uint64_t) wave * PRIM_WAVE_SPAN + lane;
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
    if (lane == 0)
        waveTotals[wave] = running;
    __syncthreads();
    T before = tileSums[blockIdx.x];
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

/* ------------------------------------------------------------------ single-pass scan (decoupled look-back)
 *
 * One launch, the input read once.  A workgroup takes the next tile by ticket (so every predecessor is already
 * running), scans it locally, PUBLISHES the tile's aggregate, adds up its predecessors' aggregates until it meets one
 * that has published an inclusive prefix, then publishes its own prefix (Merrill & Garland, "Single-pass parallel
 * prefix scan with decoupled look-back").  A descriptor is ONE 64-bit word -- value in the low half, (epoch << 2 |
 * state) in the high half -- written and read with relaxed agent-scope atomics, so no fences are needed and words of
 * an earlier scan (other epoch) read as "not yet published" without clearing the buffer.  A U3 scan is three
 * independent chains over the same tiles.  Waiting is bounded: a workgroup that gives up sets *failed (checked at the
 * context's next synchronisation) instead of hanging the GPU.
 */
enum { SCAN_INVALID = 0, SCAN_AGGREGATE = 1, SCAN_PREFIX = 2 };

__device__ __forceinline__ int scanComponents(uint32_t) { return 1; }
__device__ __forceinline__ int scanComponents(U3) { return 3; }
__device__ __forceinline__ uint32_t scanGet(uint32_t v, int) { return v; }
__device__ __forceinline__ uint32_t scanGet(const U3 &v, int c) { return c == 0 ? v.a : (c == 1 ? v.b : v.c); }
__device__ __forceinline__ void scanSet(uint32_t &v, int, uint32_t x) { v = x; }
__device__ __forceinline__ void scanSet(U3 &v, int c, uint32_t x) { if (c == 0) v.a = x; else if (c == 1) v.b = x; else v.c = x; }

__device__ __forceinline__ void scanPublish(uint64_t *desc, uint32_t epoch, uint32_t state, uint32_t value)
{
    __hip_atomic_store(desc, ((uint64_t) ((epoch << 2) | state) << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* exclusive prefix of `tile` for one component chain; called by the whole of wave 0 */
__device__ __forceinline__ uint32_t scanLookBack(const uint64_t *desc, uint32_t stride, uint32_t tile, uint32_t epoch, uint32_t *failed)
{
    const uint32_t lane = laneId();
    uint32_t acc = 0;
    int64_t first = (int64_t) tile - 1;         /* nearest predecessor looked at by lane 0 */
    uint32_t spins = 0;
    while (true)
    {
        const int64_t pred = first - lane;
        uint32_t state = SCAN_PREFIX, value = 0;    /* lanes before tile 0 read as an empty prefix */
        if (pred >= 0)
        {
            const uint64_t d = __hip_atomic_load(&desc[(uint64_t) pred * stride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t tag = (uint32_t) (d >> 32);
            state = (tag >> 2) == epoch ? (tag & 3u) : (uint32_t) SCAN_INVALID;
            value = (uint32_t) d;
        }
        const uint64_t prefixes = __ballot(state == SCAN_PREFIX);
        const uint64_t invalid = __ballot(state == SCAN_INVALID);
        const uint32_t stop = prefixes != 0 ? (uint32_t) __builtin_ctzll(prefixes) : 64u;   /* nearest published prefix */
        const uint64_t needed = stop >= 63 ? ~0ull : ((2ull << stop) - 1);                   /* lanes 0 .. stop */
        if (invalid & needed)
        {
            if (++spins > (1u << 22))
            {
                if (lane == 0)
                    *failed = 1;
                return acc;
            }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        acc += waveSum(lane <= stop ? value : 0u);
        if (stop < 64)
            return acc;
        first -= 64;
    }
}

template<typename T, typename In, typename Out>
__global__ __launch_bounds__(PRIM_BLOCK) void scanOnePassKernel(In in, Out out, uint64_t n, const uint32_t *nDev, T seed, T *total,
                                                                uint64_t *desc, uint32_t *ticket, uint32_t epoch, uint32_t *failed)
{
    if (nDev != nullptr && *nDev < n)
        n = *nDev;
    __shared__ uint32_t sTile;
    __shared__ T sPrefix;
    __shared__ T waveTotals[PRIM_WAVES];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0)
    {
        sTile = atomicAdd(ticket, 1u);
        if (sTile == gridDim.x - 1)
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    /* every ticket is taken */
    }
    __syncthreads();
    const uint32_t tile = sTile;
    const int C = scanComponents(T());
    const uint64_t base = (uint64_t) tile * PRIM_TILE + (uint64_t) wave * PRIM_WAVE_SPAN + lane;
    T vals[PRIM_ITEMS];
    T excl[PRIM_ITEMS];
    T running = zeroOf(T());
#pragma unroll
    for (int j = 0; j < PRIM_ITEMS; j++)
    {
        const uint64_t i = base + (uint64_t) j * 64;
        vals[j] = i < n ? in(i) : zeroOf(T());
        const T incl = waveInclusiveScanT(vals[j]);
        excl[j] = running + waveShiftUpT(incl);
        running = running + readLaneT(incl, 63);
    }
    if (lane == 0)
        waveTotals[wave] = running;
    __syncthreads();
    if (wave == 0)
    {
        T sum = waveTotals[0];
#pragma unroll
        for (int w = 1; w < PRIM_WAVES; w++)
            sum = sum + waveTotals[w];
        T prefix = seed;
        if (tile == 0)
        {
            if (lane == 0)
                for (int c = 0; c < C; c++)
                    scanPublish(&desc[c], epoch, SCAN_PREFIX, scanGet(seed, c) + scanGet(sum, c));
        }
        else
        {
            if (lane == 0)
                for (int c = 0; c < C; c++)
                    scanPublish(&desc[(uint64_t) tile * C + c], epoch, SCAN_AGGREGATE, scanGet(sum, c));
            for (int c = 0; c < C; c++)
            {
                const uint32_t before = scanLookBack(desc + c, (uint32_t) C, tile, epoch, failed);
                scanSet(prefix, c, before);
                if (lane == 0)
                    scanPublish(&desc[(uint64_t) tile * C + c], epoch, SCAN_PREFIX, before + scanGet(sum, c));
            }
        }
        if (lane == 0)
        {
            sPrefix = prefix;
            /* the tile holding the last element (tile 0 of an empty scan) reports the total; later tiles are empty */
            const uint32_t lastTile = n > 0 ? (uint32_t) ((n - 1) / PRIM_TILE) : 0u;
            if (total != nullptr && tile == lastTile)
                *total = prefix + sum;
        }
    }
    __syncthreads();
    T before = sPrefix;
    for (uint32_t w = 0; w < wave; w++)
        before = before + waveTotals[w];
#pragma unroll
    for (int j = 0; j < PRIM_ITEMS; j++)
    {
        const uint64_t i = base + (uint64_t) j * 64;
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
        LAUNCH(ctx, statName, (scanApplyKernel<T, In, Out>), dim3(tiles), dim3(PRIM_BLOCK), in, out, dTileSums, n, nDev);
    return MLSGPU_OK;
}

/* largest scan (in tiles) whose workgroups add up their predecessors' tile sums themselves */
/* descriptors, ticket and failure flag of the single-pass scans of one context (its scans are stream-ordered) */
struct ScanState
{
    uint64_t *desc = nullptr;
    uint64_t capWords = 0;
    uint32_t *ticket = nullptr;     /* [0] ticket, [1] failed */
    uint32_t epoch = 0;
    ~ScanState()
    {
        hipFree(desc);
        hipFree(ticket);
    }
};

static int scanState(mlsgpu_ctx *ctx, uint64_t words, ScanState **out)
{
    std::shared_ptr<void> &cached = ctx->scratchCache["scan"];
    if (!cached)
        cached = std::shared_ptr<void>(new ScanState, [](void *p) { delete static_cast<ScanState *>(p); });
    ScanState *st = static_cast<ScanState *>(cached.get());
    if (st->ticket == nullptr)
    {
        HIP_CHECK(hipMalloc((void **) &st->ticket, 2 * sizeof(uint32_t)));
        HIP_CHECK(hipMemsetAsync(st->ticket, 0, 2 * sizeof(uint32_t), ctx->stream));
    }
    if (st->capWords < words)
    {
        /* earlier scans on this stream may still be running: the old buffer is released by hipFree after they finish */
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        hipFree(st->desc);
        st->desc = nullptr;
        st->capWords = 0;
        const uint64_t cap = std::max<uint64_t>(words + words / 2, 4096);
        HIP_CHECK(hipMalloc((void **) &st->desc, cap * sizeof(uint64_t)));
        HIP_CHECK(hipMemsetAsync(st->desc, 0, cap * sizeof(uint64_t), ctx->stream));
        st->capWords = cap;
    }
    *out = st;
    return MLSGPU_OK;
}

/* MLSGPU_ERR_HIP if a look-back of an earlier scan on this context gave up (never observed; see scanLookBack) */
static inline int scanCheck(mlsgpu_ctx *ctx)
{
    auto it = ctx->scratchCache.find("scan");
    if (it == ctx->scratchCache.end())
        return MLSGPU_OK;
    ScanState *st = static_cast<ScanState *>(it->second.get());
    if (st->ticket == nullptr)
        return MLSGPU_OK;
    uint32_t failed = 0;
    /* on the context's own stream: a null-stream copy would wait for every other worker's stream too */
    HIP_CHECK(hipMemcpyAsync(&failed, st->ticket + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (failed != 0)
        return setError(MLSGPU_ERR_HIP, "single-pass scan: look-back gave up");
    return MLSGPU_OK;
}

/* in1 is what the scan reads (once); in2 is the same values for a second reader and is only used by the three-launch
 * form that very large scans fall back to */
template<typename T, typename In1, typename In2, typename Out>
static int exclusiveScan2(mlsgpu_ctx *ctx, const char *statName, In1 in1, In2 in2, Out out, uint64_t n, T seed,
                          T *dTileSums, T *dTotal, const uint32_t *nDev = nullptr)
{
    const uint32_t tiles = scanTiles(n);
    if (tiles == 0)
    {
        if (dTotal != nullptr)
            HIP_CHECK(hipMemcpyAsync(dTotal, &seed, sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        return MLSGPU_OK;
    }
    if (tiles <= (1u << 24))
    {
        ScanState *st = nullptr;
        PROPAGATE(scanState(ctx, (uint64_t) tiles * (sizeof(T) / 4), &st));
        st->epoch = (st->epoch + 1) & 0x3FFFFFFFu;
        if (st->epoch == 0)
        {
            /* the epoch wrapped: forget every descriptor once */
            HIP_CHECK(hipMemsetAsync(st->desc, 0, st->capWords * sizeof(uint64_t), ctx->stream));
            st->epoch = 1;
        }
        LAUNCH(ctx, statName, (scanOnePassKernel<T, In1, Out>), dim3(tiles), dim3(PRIM_BLOCK), in1, out, n, nDev, seed, dTotal,
               st->desc, st->ticket, st->epoch, st->ticket + 1);
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
    const uint64_t base = (uint64_t) blockIdx.x * PRIM_TILE + (uint64_t) wave * PRIM_WAVE_SPAN + lane;
#pragma unroll 4
    for (int j = 0; j < PRIM_ITEMS; j++)
    {
        uint64_t i = base + (uint64_t) j * 64;
        if (i < n)
            atomicAdd(&bins[(uint32_t) ((keys[i] >> shift) & mask)], 1u);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < numBins; d += PRIM_BLOCK)
        hist[(uint64_t) d * numTiles + blockIdx.x] = bins[d];
}

/* Largest digit a key type can be sorted by per pass: bounded by the LDS the scatter kernel needs
 * (tile of keys + values, per-wave bins).  64 KB of static LDS per workgroup. */
template<typename K> struct SortCaps { enum { MAX_DIGIT_BITS = sizeof(K) == 8 ? 9 : SORT_MAX_DIGIT_BITS }; };

/*
 * Scatter pass.  hist has been exclusively scanned over the (digit-major, tile-minor) sequence, so
 * hist[d][tile] is where this tile's keys with digit d start in the output.
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
                                                                const uint32_t *hist, uint64_t n,
                                                                uint32_t shift, uint32_t digitBits, uint32_t numTiles,
                                                                const uint32_t *nDev)
{
    enum { BINS = 1 << BIN_BITS };
    __shared__ uint32_t waveBins[PRIM_WAVES][BINS];
    __shared__ uint32_t tileBase[BINS];        /* global start of the digit minus its tile-local start */
    __shared__ uint32_t waveTotals[PRIM_WAVES];
    /* the tile is reordered in two phases through ONE buffer (keys, then values): half the LDS, twice the
     * resident workgroups, which is what this latency-bound kernel needs */
    __shared__ K sTile[PRIM_TILE];
    if (nDev != nullptr && *nDev < n)
        n = *nDev;
    const uint64_t tileFirst = (uint64_t) blockIdx.x * PRIM_TILE;
    if (tileFirst >= n)
        return;
    const uint32_t tileCount = (uint32_t) (n - tileFirst < PRIM_TILE ? n - tileFirst : PRIM_TILE);
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
    const uint64_t base = tileFirst + (uint64_t) wave * PRIM_WAVE_SPAN + lane;
    K keys[PRIM_ITEMS];
#pragma unroll
    for (int j = 0; j < PRIM_ITEMS; j++)
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
        uint32_t mine = 0;
        if (d0 < numBins)
            for (uint32_t k = 0; k < per; k++)
#pragma unroll
                for (int w = 0; w < PRIM_WAVES; w++)
                    mine += waveBins[w][d0 + k];
        const uint32_t incl = waveInclusiveScan(mine);
        if (lane == 63)
            waveTotals[wave] = incl;
        __syncthreads();
        uint32_t run = incl - mine;
        for (uint32_t w = 0; w < wave; w++)
            run += waveTotals[w];
        if (d0 < numBins)
            for (uint32_t k = 0; k < per; k++)
            {
                const uint32_t d = d0 + k;
                tileBase[d] = hist[(uint64_t) d * numTiles + blockIdx.x] - run;
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
    uint32_t dst[PRIM_ITEMS];
#pragma unroll
    for (int j = 0; j < PRIM_ITEMS; j++)
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
    uint32_t out[PRIM_ITEMS];
#pragma unroll
    for (int k = 0; k < PRIM_ITEMS; k++)
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
    for (int j = 0; j < PRIM_ITEMS; j++)
    {
        const uint64_t i = base + (uint64_t) j * 64;
        if (i < n)
            sVals[dst[j]] = IOTA ? (uint32_t) i : valsIn[i];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PRIM_ITEMS; k++)
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
/* elements of uint32 needed for the histogram of a sort of n keys */
static inline uint64_t sortHistElems(uint64_t n) { return (uint64_t) SORT_MAX_BINS * scanTiles(n); }

/*
 * Sorts (keysA, valsA)[0..n) stably by key bits [0, bits).  keysB/valsB are same-sized temporaries
 * (the reference aliases its sort temporaries onto other buffers the same way,
 * src/splat_tree_cl.cpp:129, src/marching.cpp:405).  iota: values are 0..n-1 and valsA is not read.
 * dHist: sortHistElems(n) uint32; dTileSums: scanTiles(sortHistElems(n)) uint32.
 */
template<typename K>
static int radixSort(mlsgpu_ctx *ctx, const char *statName, K *keysA, uint32_t *valsA, K *keysB, uint32_t *valsB,
                     uint64_t n, uint32_t bits, bool iota, uint32_t *dHist, uint32_t *dTileSums,
                     SortResult<K> *result, const uint32_t *nDev = nullptr)
{
    result->keys = keysA;
    result->vals = valsA;
    if (n == 0)
        return MLSGPU_OK;
    const uint32_t tiles = scanTiles(n);
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
    uint32_t shift = 0;
    K *kin = keysA, *kout = keysB;
    uint32_t *vin = valsA, *vout = valsB;
    for (uint32_t p = 0; p < passes; p++)
    {
        const uint32_t digitBits = (bits - shift) < perPass ? (bits - shift) : perPass;
        const uint64_t histN = ((uint64_t) 1 << digitBits) * tiles;
        LAUNCH(ctx, statName, (sortHistKernel<K>), dim3(tiles), dim3(PRIM_BLOCK),
               (const K *) kin, dHist, n, shift, digitBits, tiles, nDev);
        PROPAGATE((exclusiveScan<uint32_t>(ctx, statName, ArrayIn<uint32_t>{dHist}, ArrayOut<uint32_t>{dHist},
                                           histN, 0u, dTileSums, (uint32_t *) nullptr)));
#define SORT_SCATTER(IOTA, BITS)                                                                                       \
        LAUNCH(ctx, statName, (sortScatterKernel<K, IOTA, BITS>), dim3(tiles), dim3(PRIM_BLOCK), (const K *) kin,      \
               (const uint32_t *) vin, kout, vout, (const uint32_t *) dHist, n, shift, digitBits, tiles, nDev)
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
