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

#include <algorithm>
#include <cstdlib>
#include <type_traits>

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

/* Workgroup -> tile for the kernels that write or read ONE word per (digit, tile) of a [digit][tile] array.  Workgroups are
 * dealt round-robin to the 8 XCDs, each with its own write-back L2: with tile = workgroup id, the 32 words of a 128-byte line
 * of a digit's row come from 32 workgroups on 8 different XCDs, and every L2 writes its four words back as a masked partial
 * line (10^9 pairs in 1 024 digits: 11.75 GB of write requests for 2.1 GB of counters).  Here runs of 32 consecutive tiles stay
 * on one XCD (runs dealt round-robin), so a line is completed in ONE L2.  Any bijection gives the same results. */
template<uint32_t RUN>
__device__ __forceinline__ uint32_t runOfWorkgroup(uint32_t id, uint32_t count)
{
    /* RUN consecutive work items per XCD, the runs dealt round-robin; the items past the last whole group of 8 runs as they come */
    const uint32_t full = count / (8u * RUN) * (8u * RUN);
    if (id >= full)
        return id;
    const uint32_t j = id % (8u * RUN);
    return (id - j) + (j & 7u) * RUN + (j >> 3);
}
__device__ __forceinline__ uint32_t tileOfWorkgroup(uint32_t id, uint32_t numTiles) { return runOfWorkgroup<32>(id, numTiles); }

/* ------------------------------------------------------------------ scan */

/* Every kernel here has a bucket dimension (common.hpp, Lanes): blockIdx.y is the lane, the lane's arguments are one
 * element of the by-value array, and workgroups beyond the lane's own number of tiles leave at once. */

template<typename T, typename In>
struct ScanReduceArgs
{
    In in;
    T *tileSums;
    uint64_t n;
    const uint32_t *nDev;       /* element count produced on the device; the grid covers the upper bound n */
    uint32_t numTiles;
};

/* tileSums[b] = sum of in(i) over tile b */
template<typename T, typename In>
__global__ __launch_bounds__(PRIM_BLOCK) void scanReduceKernel(Lanes<ScanReduceArgs<T, In> > lanes)
{
    const ScanReduceArgs<T, In> A = lanes.a[blockIdx.y];
    if (blockIdx.x >= A.numTiles)
        return;
    uint64_t n = A.n;
    if (A.nDev != nullptr && *A.nDev < n)
        n = *A.nDev;
    __shared__ T waveTotals[PRIM_WAVES];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t) blockIdx.x * PRIM_TILE + (uint64_t) wave * PRIM_WAVE_SPAN + lane;
    T sum = zeroOf(T());
#pragma unroll 4
    for (int j = 0; j < PRIM_ITEMS; j++)
    {
        uint64_t i = base + (uint64_t) j * 64;
        if (i < n)
            sum = sum + A.in(i);
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
        A.tileSums[blockIdx.x] = t;
    }
}

template<typename T>
struct ScanTileSumsArgs
{
    T *tileSums;
    uint32_t numTiles;
    T seed;
    T *total;
};

/* One workgroup per lane: exclusive scan of the tile sums in place, starting at `seed`; grand total (incl. seed) to *total. */
template<typename T>
__global__ __launch_bounds__(PRIM_BLOCK) void scanTileSumsKernel(Lanes<ScanTileSumsArgs<T> > lanes)
{
    const ScanTileSumsArgs<T> A = lanes.a[blockIdx.y];
    __shared__ T waveTotals[PRIM_WAVES];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t numTiles = A.numTiles;
    T *const tileSums = A.tileSums;
    T carry = A.seed;
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
    if (threadIdx.x == 0 && A.total != nullptr)
        *A.total = carry;
}

template<typename T, typename In, typename Out>
struct ScanApplyArgs
{
    In in;
    Out out;
    const T *tileSums;
    uint64_t n;
    const uint32_t *nDev;
    T seed;
    T *total;
    uint32_t numTiles;
};

/* out(i, exclusivePrefix(i), in(i)) for every i < n.
 * FUSED = false: tileSums already hold each tile's exclusive prefix (scanTileSumsKernel ran).
 * FUSED = true:  tileSums are the raw tile sums; every workgroup adds up its predecessors' itself (at most
 *                SCAN_FUSED_MAX_TILES L2-resident values), which saves the single-workgroup launch in between;
 *                the last workgroup writes the grand total. */
template<typename T, typename In, typename Out, bool FUSED>
__global__ __launch_bounds__(PRIM_BLOCK) void scanApplyKernel(Lanes<ScanApplyArgs<T, In, Out> > lanes)
{
    const ScanApplyArgs<T, In, Out> A = lanes.a[blockIdx.y];
    if (blockIdx.x >= A.numTiles)
        return;
    uint64_t n = A.n;
    if (A.nDev != nullptr && *A.nDev < n)
        n = *A.nDev;
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
        const T *const tileSums = A.tileSums;
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
        vals[j] = i < n ? A.in(i) : zeroOf(T());
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
        before = A.seed;
#pragma unroll
        for (int w = 0; w < PRIM_WAVES; w++)
            before = before + wavePrefix[w];
        if (A.total != nullptr && blockIdx.x == lastTile && threadIdx.x == 0)
        {
            T all = before;
#pragma unroll
            for (int w = 0; w < PRIM_WAVES; w++)
                all = all + waveTotals[w];
            *A.total = all;
        }
    }
    else
        before = A.tileSums[blockIdx.x];
    for (uint32_t w = 0; w < wave; w++)
        before = before + waveTotals[w];
#pragma unroll
    for (int j = 0; j < PRIM_ITEMS; j++)
    {
        uint64_t i = base + (uint64_t) j * 64;
        if (i < n)
            A.out(i, before + excl[j], vals[j]);
    }
}

/* ---- the scan in ONE launch.  A workgroup publishes its tile's sum (value, then a flag behind a release), and adds up
 * the sums of ALL its predecessors as they appear -- no chain from tile to tile: a workgroup waits for the slowest of the
 * earlier tiles' loads, not for a sequence of look-backs.  The flag is the launch's EPOCH (the context counts its one-pass
 * launches), so flags are never cleared: what an earlier launch left behind does not match.  A workgroup's TILE is a ticket
 * it draws when it starts (one returning agent-scope atomic on a counter per lane that is never reset: the host knows what it
 * read before the launch), not blockIdx.x: HIP promises nothing about the order workgroups are dispatched in, and a
 * workgroup spinning on a tile whose workgroup is not resident would hang the stream.  With tickets the tiles a workgroup
 * waits for belong to workgroups that have started, and those publish before they wait (up to SCAN_ONEPASS_MAX_TILES
 * tiles; longer scans take the two-launch form above).  The waits and the sums' loads are agent-scope atomics: they see
 * what another XCD's workgroup published. ---- */
__device__ __forceinline__ uint32_t loadAgent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ U3 loadAgent(const U3 *p) { return U3{loadAgent(&p->a), loadAgent(&p->b), loadAgent(&p->c)}; }
__device__ __forceinline__ void storeAgent(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void storeAgent(U3 *p, U3 v)
{
    storeAgent(&p->a, v.a);
    storeAgent(&p->b, v.b);
    storeAgent(&p->c, v.c);
}

template<typename T, typename In, typename Out>
struct ScanOnePassArgs
{
    In in;
    Out out;
    T *tileSums;
    uint32_t *flags;            /* one word per tile of this lane */
    uint32_t *ticket;           /* this lane's ticket counter ... */
    uint32_t ticketBase;        /* ... and what it reads when this launch begins */
    uint64_t n;
    const uint32_t *nDev;
    T seed;
    T *total;
    uint32_t numTiles;
};

template<typename T, typename In, typename Out>
__global__ __launch_bounds__(PRIM_BLOCK) void scanOnePassKernel(Lanes<ScanOnePassArgs<T, In, Out> > lanes, uint32_t epoch)
{
    const ScanOnePassArgs<T, In, Out> A = lanes.a[blockIdx.y];
    /* every workgroup of the lane draws a ticket (the lane's counter advances by gridDim.x per launch, the host counts on it) */
    __shared__ uint32_t sTile;
    if (threadIdx.x == 0)
        sTile = __hip_atomic_fetch_add(A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - A.ticketBase;
    __syncthreads();
    const uint32_t tile = sTile;
    if (tile >= A.numTiles)
        return;
    uint64_t n = A.n;
    if (A.nDev != nullptr && *A.nDev < n)
        n = *A.nDev;
    /* the tile holding the last element (tile 0 of an empty scan) reports the total; later tiles are empty, publish nothing
     * and nobody waits for them */
    const uint32_t lastTile = n > 0 ? (uint32_t) ((n - 1) / PRIM_TILE) : 0u;
    if (tile > lastTile)
        return;
    __shared__ T waveTotals[PRIM_WAVES];
    __shared__ T wavePrefix[PRIM_WAVES];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t) tile * PRIM_TILE + (uint64_t) wave * PRIM_WAVE_SPAN + lane;
    T vals[PRIM_ITEMS];
    T excl[PRIM_ITEMS];
    T running = zeroOf(T());        /* wave-uniform: sum of the previous rounds of this wave */
#pragma unroll
    for (int j = 0; j < PRIM_ITEMS; j++)
    {
        uint64_t i = base + (uint64_t) j * 64;
        vals[j] = i < n ? A.in(i) : zeroOf(T());
        T incl = waveInclusiveScanT(vals[j]);
        excl[j] = running + waveShiftUpT(incl);
        running = running + readLaneT(incl, 63);
    }
    if (lane == 0)
        waveTotals[wave] = running;
    __syncthreads();
    T mine = waveTotals[0];
#pragma unroll
    for (int w = 1; w < PRIM_WAVES; w++)
        mine = mine + waveTotals[w];
    if (threadIdx.x == 0 && tile < lastTile)
    {
        storeAgent(&A.tileSums[tile], mine);
        __hip_atomic_store(&A.flags[tile], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    T before = zeroOf(T());
    for (uint32_t t = threadIdx.x; t < tile; t += PRIM_BLOCK)
    {
        while (__hip_atomic_load(&A.flags[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch)
            __builtin_amdgcn_s_sleep(2);
        before = before + loadAgent(&A.tileSums[t]);
    }
    const T inclBefore = waveInclusiveScanT(before);
    if (lane == 63)
        wavePrefix[wave] = inclBefore;
    __syncthreads();
    before = A.seed;
#pragma unroll
    for (int w = 0; w < PRIM_WAVES; w++)
        before = before + wavePrefix[w];
    if (A.total != nullptr && tile == lastTile && threadIdx.x == 0)
        *A.total = before + mine;
    for (uint32_t w = 0; w < wave; w++)
        before = before + waveTotals[w];
#pragma unroll
    for (int j = 0; j < PRIM_ITEMS; j++)
    {
        uint64_t i = base + (uint64_t) j * 64;
        if (i < n)
            A.out(i, before + excl[j], vals[j]);
    }
}

#endif /* __HIPCC__ */

/* largest scan (in tiles) that runs as one launch */
#define SCAN_ONEPASS_MAX_TILES 1024u
/* words between the lanes' ticket counters (a 128-byte line each) */
#define SCAN_TICKET_STRIDE 32u

/* Host drivers.  Workspace: dTileSums must hold scanTiles(n) + 1 elements of T. */
static inline uint32_t scanTiles(uint64_t n) { return divUp(n, PRIM_TILE); }

/* largest scan (in tiles) whose workgroups add up their predecessors' tile sums themselves */
#define SCAN_FUSED_MAX_TILES 4096u

/* MLSGPU_HIP_SCAN_ONEPASS=0: every scan in the two- or three-launch form (A/B runs) */
static inline bool scanOnePassOff()
{
    static const bool off = getenv("MLSGPU_HIP_SCAN_ONEPASS") != nullptr && atoi(getenv("MLSGPU_HIP_SCAN_ONEPASS")) == 0;
    return off;
}

#ifdef __HIPCC__
/* One scan of a batch: in1 feeds the tile sums, in2 the scan proper (they must agree; a first pass may cache what the
 * second reads); out(i, prefix, value) consumes; grand total (incl. seed) to *dTotal (may be null). */
template<typename T, typename In1, typename In2, typename Out>
struct ScanJob
{
    In1 in1;
    In2 in2;
    Out out;
    uint64_t n;
    T seed;
    T *dTileSums;
    T *dTotal;
    const uint32_t *nDev;
};

/* Phase 1 for every lane: tile sums of in1() scanned from the seed; grand totals written. */
template<typename T, typename In1, typename In2, typename Out>
static int scanPhase1Batch(mlsgpu_ctx *ctx, const char *statName, const ScanJob<T, In1, In2, Out> *jobs, uint32_t count)
{
    REQUIRE(count >= 1 && count <= MAX_LANES, MLSGPU_ERR_INVALID);
    Lanes<ScanReduceArgs<T, In1> > r;
    Lanes<ScanTileSumsArgs<T> > t;
    uint32_t maxTiles = 0;
    for (uint32_t k = 0; k < MAX_LANES; k++)
    {
        const ScanJob<T, In1, In2, Out> &j = jobs[k < count ? k : 0];
        const uint32_t tiles = k < count ? scanTiles(j.n) : 0u;
        r.a[k] = ScanReduceArgs<T, In1>{j.in1, j.dTileSums, j.n, j.nDev, tiles};
        t.a[k] = ScanTileSumsArgs<T>{j.dTileSums, tiles, j.seed, j.dTotal};
        maxTiles = tiles > maxTiles ? tiles : maxTiles;
    }
    if (maxTiles > 0)
        LAUNCH(ctx, statName, (scanReduceKernel<T, In1>), dim3(maxTiles, count), dim3(PRIM_BLOCK), r);
    LAUNCH(ctx, statName, (scanTileSumsKernel<T>), dim3(1, count), dim3(PRIM_BLOCK), t);
    return MLSGPU_OK;
}

/* Phase 2 for every lane: out(i, exclusivePrefix(i), in2(i)) for all i, using the tile sums of phase 1. */
template<typename T, typename In1, typename In2, typename Out>
static int scanPhase2Batch(mlsgpu_ctx *ctx, const char *statName, const ScanJob<T, In1, In2, Out> *jobs, uint32_t count)
{
    REQUIRE(count >= 1 && count <= MAX_LANES, MLSGPU_ERR_INVALID);
    Lanes<ScanApplyArgs<T, In2, Out> > a;
    uint32_t maxTiles = 0;
    for (uint32_t k = 0; k < MAX_LANES; k++)
    {
        const ScanJob<T, In1, In2, Out> &j = jobs[k < count ? k : 0];
        const uint32_t tiles = k < count ? scanTiles(j.n) : 0u;
        a.a[k] = ScanApplyArgs<T, In2, Out>{j.in2, j.out, (const T *) j.dTileSums, j.n, j.nDev, zeroOf(T()), (T *) nullptr, tiles};
        maxTiles = tiles > maxTiles ? tiles : maxTiles;
    }
    if (maxTiles > 0)
        LAUNCH(ctx, statName, (scanApplyKernel<T, In2, Out, false>), dim3(maxTiles, count), dim3(PRIM_BLOCK), a);
    return MLSGPU_OK;
}

/* The whole scan for every lane of a batch. */
template<typename T, typename In1, typename In2, typename Out>
static int exclusiveScanBatch(mlsgpu_ctx *ctx, const char *statName, const ScanJob<T, In1, In2, Out> *jobs, uint32_t count)
{
    REQUIRE(count >= 1 && count <= MAX_LANES, MLSGPU_ERR_INVALID);
    uint32_t maxTiles = 0;
    for (uint32_t k = 0; k < count; k++)
        maxTiles = std::max(maxTiles, scanTiles(jobs[k].n));
    if (std::is_same<In1, In2>::value && maxTiles <= SCAN_ONEPASS_MAX_TILES && !scanOnePassOff())
    {
        /* one launch (scanOnePassKernel).  An empty lane still runs its tile 0, which reports the total. */
        uint32_t *flags = nullptr, *tickets = nullptr, epoch = 0, bases[MAX_LANES];
        maxTiles = std::max(maxTiles, 1u);
        PROPAGATE(ctx->scanFlags(&flags, &epoch, &tickets, bases, maxTiles, count));
        Lanes<ScanOnePassArgs<T, In2, Out> > a;
        for (uint32_t k = 0; k < MAX_LANES; k++)
        {
            const ScanJob<T, In1, In2, Out> &j = jobs[k < count ? k : 0];
            const uint32_t tiles = k < count ? std::max(scanTiles(j.n), 1u) : 0u;
            a.a[k] = ScanOnePassArgs<T, In2, Out>{j.in2, j.out, j.dTileSums, flags + (uint64_t) k * SCAN_ONEPASS_MAX_TILES,
                                                  tickets + (uint64_t) k * SCAN_TICKET_STRIDE, bases[k], j.n, j.nDev,
                                                  j.seed, j.dTotal, tiles};
        }
        LAUNCH(ctx, statName, (scanOnePassKernel<T, In2, Out>), dim3(maxTiles, count), dim3(PRIM_BLOCK), a, epoch);
        return MLSGPU_OK;
    }
    if (maxTiles <= SCAN_FUSED_MAX_TILES)
    {
        /* two launches: raw tile sums, then the scan proper.  An empty lane still runs its tile 0, which reports the total. */
        Lanes<ScanReduceArgs<T, In1> > r;
        Lanes<ScanApplyArgs<T, In2, Out> > a;
        for (uint32_t k = 0; k < MAX_LANES; k++)
        {
            const ScanJob<T, In1, In2, Out> &j = jobs[k < count ? k : 0];
            const uint32_t tiles = k < count ? std::max(scanTiles(j.n), 1u) : 0u;
            r.a[k] = ScanReduceArgs<T, In1>{j.in1, j.dTileSums, j.n, j.nDev, tiles};
            a.a[k] = ScanApplyArgs<T, In2, Out>{j.in2, j.out, (const T *) j.dTileSums, j.n, j.nDev, j.seed, j.dTotal, tiles};
        }
        maxTiles = std::max(maxTiles, 1u);
        LAUNCH(ctx, statName, (scanReduceKernel<T, In1>), dim3(maxTiles, count), dim3(PRIM_BLOCK), r);
        LAUNCH(ctx, statName, (scanApplyKernel<T, In2, Out, true>), dim3(maxTiles, count), dim3(PRIM_BLOCK), a);
        return MLSGPU_OK;
    }
    PROPAGATE((scanPhase1Batch<T, In1, In2, Out>(ctx, statName, jobs, count)));
    return scanPhase2Batch<T, In1, In2, Out>(ctx, statName, jobs, count);
}

/* single-bucket forms */
struct NoIn { };
struct NoOut { };

template<typename T, typename In>
static int scanPhase1(mlsgpu_ctx *ctx, const char *statName, In in, uint64_t n, T seed, T *dTileSums, T *dTotal,
                      const uint32_t *nDev = nullptr)
{
    const ScanJob<T, In, NoIn, NoOut> job{in, NoIn(), NoOut(), n, seed, dTileSums, dTotal, nDev};
    return scanPhase1Batch<T, In, NoIn, NoOut>(ctx, statName, &job, 1);
}

template<typename T, typename In, typename Out>
static int scanPhase2(mlsgpu_ctx *ctx, const char *statName, In in, Out out, uint64_t n, const T *dTileSums,
                      const uint32_t *nDev = nullptr)
{
    const ScanJob<T, NoIn, In, Out> job{NoIn(), in, out, n, zeroOf(T()), const_cast<T *>(dTileSums), (T *) nullptr, nDev};
    return scanPhase2Batch<T, NoIn, In, Out>(ctx, statName, &job, 1);
}

template<typename T, typename In1, typename In2, typename Out>
static int exclusiveScan2(mlsgpu_ctx *ctx, const char *statName, In1 in1, In2 in2, Out out, uint64_t n, T seed,
                          T *dTileSums, T *dTotal, const uint32_t *nDev = nullptr)
{
    const ScanJob<T, In1, In2, Out> job{in1, in2, out, n, seed, dTileSums, dTotal, nDev};
    return exclusiveScanBatch<T, In1, In2, Out>(ctx, statName, &job, 1);
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
struct SortHistArgs
{
    const K *keys;
    uint32_t *hist;
    uint64_t n;
    const uint32_t *nDev;
    uint32_t numTiles;
    uint32_t *keyCounts;        /* KEY_COUNTS: occurrences of every whole key (the pass's digit is the key's top digit) */
    /* KEY_COUNTS, packed words (idBits != 0): an element is (top digit << idBits | value) -- the low part of its key is not
     * stored: the input is grouped by it, group g at positions [lowBase[g], lowBase[g + 1]) */
    const uint32_t *lowBase = nullptr;
    uint32_t idBits = 0;
};

/* the group (low part) position i lies in, for groups that begin at lowBase[0 .. numLow]: at or behind `from` */
__device__ __forceinline__ uint32_t lowPartAt(const uint32_t *lowBase, uint32_t from, uint32_t numLow, uint64_t i)
{
    while (from + 1 < numLow && i >= lowBase[from + 1])
        from++;
    return from;
}

enum { KEY_COUNT_SLOTS = 4 };   /* low parts of the key a tile may span and still count in LDS */

/*
 * Histogram of one digit per tile.  KEY_COUNTS (the LAST pass of a sort whose keys have `shift + digitBits` bits, 256 bins at
 * most): the kernel also counts the occurrences of every whole key, keyCounts[key].  The input is sorted by the low `shift`
 * bits already, so a tile of 4096 keys holds one low part, sometimes two: the counts are gathered in LDS per (low part, top
 * digit) and leave as one global atomic per non-empty bin -- a sixteenth of the atomics a key-by-key count would take
 * (~16 keys per bin on the octree's entries).  Keys of a tile beyond KEY_COUNT_SLOTS low parts (tiny groups) go straight
 * to memory.
 */
template<typename K, bool KEY_COUNTS = false>
__global__ __launch_bounds__(PRIM_BLOCK) void sortHistKernel(Lanes<SortHistArgs<K> > lanes, uint32_t shift, uint32_t digitBits)
{
    const SortHistArgs<K> A = lanes.a[blockIdx.y];
    if (blockIdx.x >= A.numTiles)
        return;
    uint64_t n = A.n;
    if (A.nDev != nullptr && *A.nDev < n)
        n = *A.nDev;
    __shared__ uint32_t bins[SORT_MAX_BINS];
    __shared__ uint32_t keyBins[KEY_COUNTS ? KEY_COUNT_SLOTS * 256 : 1];
    __shared__ uint32_t sLow[KEY_COUNTS ? 256 : 1];
    __shared__ uint32_t sGroup[PRIM_WAVES][2];
    const uint32_t numBins = 1u << digitBits;
    const K mask = (K) (numBins - 1);
    const bool packed = KEY_COUNTS && A.idBits != 0;
    const uint32_t numLow = 1u << shift;
    const uint32_t digitShift = packed ? A.idBits : shift;
    const K *const keys = A.keys;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = tileOfWorkgroup(blockIdx.x, A.numTiles);
    const uint64_t tileFirst = (uint64_t) tile * SORT_TILE;
    const uint64_t base = tileFirst + (uint64_t) wave * SORT_WAVE_SPAN + lane;
    const K lowMask = (K) (((K) 1 << shift) - 1);
    /* all of the thread's keys are requested before the first is counted (one round of memory latency, not four) -- and,
     * with them, the group bounds of packed words */
    K mine[SORT_ITEMS];
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++)
    {
        const uint64_t i = base + (uint64_t) j * 64;
        mine[j] = i < n ? keys[i] : (K) 0;
    }
    for (uint32_t d = threadIdx.x; d < numBins; d += PRIM_BLOCK)
        bins[d] = 0;
    if (KEY_COUNTS)
    {
        for (uint32_t d = threadIdx.x; d < KEY_COUNT_SLOTS * 256; d += PRIM_BLOCK)
            keyBins[d] = 0;
        if (packed)
        {
            /* thread g holds where group g begins (numLow <= 256 = PRIM_BLOCK): the groups the tile's first and last element
             * lie in are counts of begins at or before them -- two ballots, no search */
            const uint64_t tileLast = tileFirst + SORT_TILE - 1 < n ? tileFirst + SORT_TILE - 1 : n - 1;
            const uint32_t myBase = threadIdx.x < numLow ? A.lowBase[threadIdx.x] : 0xFFFFFFFFu;
            sLow[threadIdx.x] = myBase;
            const uint32_t c0 = (uint32_t) __popcll(__ballot(threadIdx.x < numLow && myBase <= tileFirst));
            const uint32_t c1 = (uint32_t) __popcll(__ballot(threadIdx.x < numLow && myBase <= tileLast));
            if (lane == 0)
            {
                sGroup[wave][0] = c0;
                sGroup[wave][1] = c1;
            }
        }
    }
    __syncthreads();
    uint32_t lowFirst = 0;
    bool oneLow = false;        /* the whole tile has one low part: the digit bins ARE the key counts (six tiles of seven) */
    if (KEY_COUNTS && tileFirst < n)
    {
        const uint64_t tileLast = tileFirst + SORT_TILE - 1 < n ? tileFirst + SORT_TILE - 1 : n - 1;
        if (packed)
        {
            uint32_t c0 = 0, c1 = 0;
#pragma unroll
            for (int w = 0; w < PRIM_WAVES; w++)
            {
                c0 += sGroup[w][0];
                c1 += sGroup[w][1];
            }
            lowFirst = c0 - 1;          /* group 0 begins at 0 */
            oneLow = c1 == c0;
        }
        else
        {
            lowFirst = (uint32_t) (keys[tileFirst] & lowMask);
            oneLow = (uint32_t) (keys[tileLast] & lowMask) == lowFirst;      /* sorted by the low part */
        }
    }
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++)
    {
        uint64_t i = base + (uint64_t) j * 64;
        if (i < n)
        {
            const K key = mine[j];
            const uint32_t digit = (uint32_t) ((key >> digitShift) & mask);
            atomicAdd(&bins[digit], 1u);
            if (KEY_COUNTS && !oneLow)
            {
                const uint32_t low = packed ? lowPartAt(sLow, lowFirst, numLow, i) : (uint32_t) (key & lowMask);
                const uint32_t slot = low - lowFirst;
                if (slot < KEY_COUNT_SLOTS)
                    atomicAdd(&keyBins[slot * 256 + digit], 1u);
                else
                    atomicAdd(&A.keyCounts[packed ? (digit << shift) | low : (uint32_t) key], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t *const hist = A.hist;
    const uint32_t numTiles = A.numTiles;
    for (uint32_t d = threadIdx.x; d < numBins; d += PRIM_BLOCK)
        hist[(uint64_t) d * numTiles + tile] = bins[d];
    if (KEY_COUNTS && oneLow)
    {
        for (uint32_t d = threadIdx.x; d < numBins; d += PRIM_BLOCK)
            if (bins[d] != 0)
                atomicAdd(&A.keyCounts[(d << shift) | lowFirst], bins[d]);
    }
    else if (KEY_COUNTS)
        for (uint32_t d = threadIdx.x; d < KEY_COUNT_SLOTS * 256; d += PRIM_BLOCK)
        {
            const uint32_t c = keyBins[d];
            if (c != 0)
                atomicAdd(&A.keyCounts[((d & 255u) << shift) | (lowFirst + (d >> 8))], c);
        }
}

struct SortDigitScanArgs
{
    uint32_t *hist;
    uint32_t *digitTotals;
    uint32_t numTiles;
};

/* hist[d][0 .. numTiles) -> its exclusive prefix along the tiles, in place; digitTotals[d] = the digit's count.  One
 * workgroup per digit (and lane): the whole scan of a pass's histogram is this ONE small launch (the scatter kernel turns
 * the 2^bits digit totals into digit bases itself), where a generic scan of the 2^bits x numTiles array took two. */
template<typename T>    /* T = uint32_t; a template only so that the header can be included by several translation units */
__global__ __launch_bounds__(PRIM_BLOCK) void sortDigitScanKernel(Lanes<SortDigitScanArgs> lanes)
{
    const SortDigitScanArgs A = lanes.a[blockIdx.y];
    __shared__ T waveTotals[PRIM_WAVES];
    const uint32_t numTiles = A.numTiles;
    T *row = A.hist + (uint64_t) blockIdx.x * numTiles;
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
        A.digitTotals[blockIdx.x] = carry;
}

/* Largest digit a key type can be sorted by per pass: bounded by the LDS the scatter kernel needs
 * (tile of keys + values, per-wave bins).  64 KB of static LDS per workgroup. */
template<typename K> struct SortCaps { enum { MAX_DIGIT_BITS = sizeof(K) == 8 ? 9 : SORT_MAX_DIGIT_BITS }; };

template<typename K>
struct SortScatterArgs
{
    const K *keysIn;
    const uint32_t *valsIn;
    K *keysOut;
    uint32_t *valsOut;
    const uint32_t *hist;
    const uint32_t *digitTotals;
    uint64_t n;
    const uint32_t *nDev;
    uint32_t numTiles;
    const uint32_t *keyBase;    /* SPREAD: the value of sorted rank r with key k goes to valsOut[1 + r + keyBase[k]] */
    /* SPREAD over packed words (idBits != 0, see SortHistArgs): keysIn[i] = top digit << idBits | value, the key's low part is
     * the group position i lies in; what goes out is value + valueBias */
    const uint32_t *lowBase = nullptr;
    uint32_t idBits = 0, valueBias = 0;
};

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
 *
 * SPREAD (the last pass of the octree's entry sort): the sorted keys are not written at all, and a value does not go to its
 * sorted rank r but to 1 + r + keyBase[key] -- the runs of equal keys keep their order and open up by keyBase, which is how
 * the octree's command list interleaves its per-node markers with the splat ids (octree.hip).
 */
template<typename K, bool IOTA, int BIN_BITS, bool SPREAD = false>
__global__ __launch_bounds__(PRIM_BLOCK) void sortScatterKernel(Lanes<SortScatterArgs<K> > lanes, uint32_t shift, uint32_t digitBits)
{
    enum { BINS = 1 << BIN_BITS };
    __shared__ uint32_t waveBins[PRIM_WAVES][BINS];
    __shared__ uint32_t tileBase[BINS];        /* global start of the digit minus its tile-local start */
    __shared__ uint32_t waveTotals[PRIM_WAVES], waveTotalsAll[PRIM_WAVES];
    /* the tile is reordered in two phases through ONE buffer (keys, then values): half the LDS, twice the
     * resident workgroups, which is what this latency-bound kernel needs */
    __shared__ K sTile[SORT_TILE];
    /* digits of up to 8 bits: the lanes of a wave that hold the same digit find each other through LDS -- every lane ORs its
     * bit into the digit's 64-bit word, reads the word back (LDS operations of a wave execute in order) and the first of
     * them clears it -- instead of one ballot and four vector instructions per digit BIT (the scatter was bound by issuing
     * vector instructions, half of them these) */
    enum { LDS_MATCH = BIN_BITS <= 8, HALF_MATCH = !LDS_MATCH };     /* wider digits: 32 lanes at a time, half the LDS */
    __shared__ unsigned long long sMatch[LDS_MATCH ? PRIM_WAVES : 1][LDS_MATCH ? BINS : 1];
    __shared__ uint32_t sMatch32[HALF_MATCH ? PRIM_WAVES : 1][HALF_MATCH ? BINS : 1];
    __shared__ uint32_t sLow[SPREAD ? 256 : 1];
    __shared__ uint32_t sGroup[PRIM_WAVES][2];
    const SortScatterArgs<K> A = lanes.a[blockIdx.y];
    if (blockIdx.x >= A.numTiles)
        return;
    const bool packed = SPREAD && A.idBits != 0;
    const uint32_t digitShift = packed ? A.idBits : shift;
    const uint32_t numLow = 1u << shift;
    uint64_t n = A.n;
    if (A.nDev != nullptr && *A.nDev < n)
        n = *A.nDev;
    const uint32_t tile = tileOfWorkgroup(blockIdx.x, A.numTiles);     /* its words of the [digit][tile] array: one L2's */
    const uint64_t tileFirst = (uint64_t) tile * SORT_TILE;
    if (tileFirst >= n)
        return;
    const K *const keysIn = A.keysIn;
    const uint32_t *const valsIn = A.valsIn;
    K *const keysOut = A.keysOut;
    uint32_t *const valsOut = A.valsOut;
    const uint32_t *const hist = A.hist;
    const uint32_t *const digitTotals = A.digitTotals;
    const uint32_t numTiles = A.numTiles;
    const uint32_t tileCount = (uint32_t) (n - tileFirst < SORT_TILE ? n - tileFirst : SORT_TILE);
    const uint32_t numBins = 1u << digitBits;
    const K mask = (K) (numBins - 1);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t d = threadIdx.x; d < numBins; d += PRIM_BLOCK)
    {
#pragma unroll
        for (int w = 0; w < PRIM_WAVES; w++)
        {
            waveBins[w][d] = 0;
            if (LDS_MATCH)
                sMatch[w][d] = 0ull;
            else
                sMatch32[w][d] = 0u;
        }
    }
    __syncthreads();
    const uint64_t base = tileFirst + (uint64_t) wave * SORT_WAVE_SPAN + lane;
    /* Everything the workgroup reads that does not depend on another read is requested here, together: its keys, their
     * values, and the digit totals and tile offsets of the bins this thread will own in the scan below.  (The kernel waits
     * four fifths of its time: one round of memory latency instead of three.) */
    enum { MAX_PER = BINS > PRIM_BLOCK ? BINS / PRIM_BLOCK : 1 };
    const uint32_t per = numBins > PRIM_BLOCK ? numBins / PRIM_BLOCK : 1;
    const uint32_t d0 = threadIdx.x * per;
    K keys[SORT_ITEMS];
    uint32_t vals[SORT_ITEMS];
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++)
    {
        const uint64_t i = base + (uint64_t) j * 64;
        keys[j] = i < n ? keysIn[i] : (K) 0;
        vals[j] = IOTA ? (uint32_t) i : (i < n && !packed ? valsIn[i] : 0u);
    }
    uint32_t totalOf[MAX_PER], histOf[MAX_PER];
#pragma unroll
    for (int k = 0; k < MAX_PER; k++)
    {
        const bool mineToo = (uint32_t) k < per && d0 + k < numBins;
        totalOf[k] = mineToo ? digitTotals[d0 + k] : 0u;
        histOf[k] = mineToo ? hist[(uint64_t) (d0 + k) * numTiles + tile] : 0u;
    }
    if (packed)
    {
        /* thread g holds where group g begins: see sortHistKernel */
        const uint32_t myBase = threadIdx.x < numLow ? A.lowBase[threadIdx.x] : 0xFFFFFFFFu;
        sLow[threadIdx.x] = myBase;
        const uint32_t c0 = (uint32_t) __popcll(__ballot(threadIdx.x < numLow && myBase <= tileFirst));
        const uint32_t c1 = (uint32_t) __popcll(__ballot(threadIdx.x < numLow && myBase <= tileFirst + tileCount - 1));
        if (lane == 0)
        {
            sGroup[wave][0] = c0;
            sGroup[wave][1] = c1;
        }
    }
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++)
    {
        const uint64_t i = base + (uint64_t) j * 64;
        if (i < n)
            atomicAdd(&waveBins[wave][(uint32_t) ((keys[j] >> digitShift) & mask)], 1u);
    }
    __syncthreads();
    uint32_t lowFirst = 0;
    bool oneLow = false;
    if (packed)
    {
        /* What travels behind the words through the tile's reorder is the low part of every element's key -- nothing at all
         * where the tile lies inside one group (most do: a group is thousands of entries) */
        uint32_t c0 = 0, c1 = 0;
#pragma unroll
        for (int w = 0; w < PRIM_WAVES; w++)
        {
            c0 += sGroup[w][0];
            c1 += sGroup[w][1];
        }
        lowFirst = c0 - 1;              /* group 0 begins at 0 */
        oneLow = c1 == c0;
        if (!oneLow)
        {
            uint32_t low = lowFirst;
#pragma unroll
            for (int j = 0; j < SORT_ITEMS; j++)
            {
                const uint64_t i = base + (uint64_t) j * 64;
                if (i < n)
                    low = lowPartAt(sLow, low, numLow, i);
                vals[j] = low;
            }
        }
    }
    /* tile-local exclusive prefix over the digits: thread t owns the `per` consecutive bins from t * per */
    {
        uint32_t mine = 0, mineAll = 0;         /* this tile's / the whole input's keys in my bins */
#pragma unroll
        for (int k = 0; k < MAX_PER; k++)
            if ((uint32_t) k < per && d0 + k < numBins)
            {
                mineAll += totalOf[k];
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
#pragma unroll
        for (int k = 0; k < MAX_PER; k++)
            if ((uint32_t) k < per && d0 + k < numBins)
            {
                const uint32_t d = d0 + k;
                tileBase[d] = base + histOf[k] - run;
                base += totalOf[k];
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
        const uint32_t digit = (uint32_t) ((keys[j] >> digitShift) & mask);
        uint64_t peers;
        if (LDS_MATCH)
        {
            if (valid)
                atomicOr(&sMatch[wave][digit], 1ull << lane);
            __builtin_amdgcn_wave_barrier();
            peers = valid ? sMatch[wave][digit] : 0ull;
            __builtin_amdgcn_wave_barrier();
        }
        else
        {
            /* two halves of 32 lanes, one after the other, through ONE 32-bit word per digit: a half's lanes OR their bits in,
             * EVERY lane with the digit reads the word (the other half needs it for its rank and for the run's new end), the
             * half's first lane with the digit clears it */
            const uint32_t bit = 1u << (lane & 31u);
            uint32_t lowerPeers = 0, upperPeers = 0;
            if (valid && lane < 32)
                atomicOr(&sMatch32[wave][digit], bit);
            __builtin_amdgcn_wave_barrier();
            if (valid)
                lowerPeers = sMatch32[wave][digit];
            __builtin_amdgcn_wave_barrier();
            if (valid && lane < 32 && (lowerPeers & (bit - 1u)) == 0)
                sMatch32[wave][digit] = 0u;
            __builtin_amdgcn_wave_barrier();
            if (valid && lane >= 32)
                atomicOr(&sMatch32[wave][digit], bit);
            __builtin_amdgcn_wave_barrier();
            if (valid)
                upperPeers = sMatch32[wave][digit];
            __builtin_amdgcn_wave_barrier();
            if (valid && lane >= 32 && (upperPeers & (bit - 1u)) == 0)
                sMatch32[wave][digit] = 0u;
            peers = (uint64_t) lowerPeers | (uint64_t) upperPeers << 32;
        }
        dst[j] = 0;
        if (valid)
        {
            const uint32_t rank = popcBelow(peers);
            dst[j] = waveBins[wave][digit] + rank;
            sTile[dst[j]] = keys[j];
            if (rank == 0)
            {
                waveBins[wave][digit] = dst[j] + (uint32_t) __popcll(peers);
                if (LDS_MATCH)
                    sMatch[wave][digit] = 0ull;
            }
        }
        /* LDS operations of one wave complete in program order, so the next round sees the update */
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    /* keys leave in tile-sorted order: each digit's run is one contiguous, coalesced burst */
    uint32_t out[SORT_ITEMS];
    K word[SPREAD ? SORT_ITEMS : 1];           /* packed: the element of tile-sorted rank p, kept for its value */
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++)
    {
        const uint32_t p = threadIdx.x + k * PRIM_BLOCK;
        out[k] = 0;
        if (SPREAD)
            word[k] = (K) 0;
        if (p < tileCount)
        {
            const K key = sTile[p];
            out[k] = tileBase[(uint32_t) ((key >> digitShift) & mask)] + p;
            if (SPREAD && packed)
                word[k] = key;
            else if (SPREAD)
                out[k] += 1u + A.keyBase[(uint32_t) key];
            else if (keysOut != nullptr)
                keysOut[out[k]] = key;
        }
    }
    uint32_t *sVals = reinterpret_cast<uint32_t *>(sTile);
    if (!(SPREAD && packed && oneLow))          /* (uniform over the workgroup) */
    {
        __syncthreads();
        /* the values take the same route through the same buffer */
#pragma unroll
        for (int j = 0; j < SORT_ITEMS; j++)
        {
            const uint64_t i = base + (uint64_t) j * 64;
            if (i < n)
                sVals[dst[j]] = vals[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++)
    {
        const uint32_t p = threadIdx.x + k * PRIM_BLOCK;
        if (p < tileCount)
        {
            if (SPREAD && packed)
            {
                const uint32_t w = (uint32_t) word[SPREAD ? k : 0];
                const uint32_t key = (((w >> digitShift) & (uint32_t) mask) << shift) | (oneLow ? lowFirst : sVals[p]);
                valsOut[out[k] + 1u + A.keyBase[key]] = (w & ((1u << digitShift) - 1u)) + A.valueBias;
            }
            else
                valsOut[out[k]] = sVals[p];
        }
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

/* one lane of a batched sort: (keysA, valsA)[0..n) sorted stably by key bits [0, bits); keysB/valsB are same-sized
 * temporaries (the reference aliases its sort temporaries onto other buffers the same way, src/splat_tree_cl.cpp:129,
 * src/marching.cpp:405); dHist: sortHistElems(n) uint32 */
template<typename K>
struct SortJob
{
    K *keysA;
    uint32_t *valsA;
    K *keysB;
    uint32_t *valsB;
    uint64_t n;
    uint32_t *dHist;
    const uint32_t *nDev;
    SortResult<K> result;       /* out */
};

/*
 * Sorts every lane's pairs.  iota: values are 0..n-1 and valsA is not read.  doneBits: the lowest doneBits key bits have
 * been sorted already (by a pass of that width fused into the producer of the keys); the remaining passes keep the split of
 * the whole sort.  The passes -- digit widths, hence the kernels -- are the same for every lane.
 */
template<typename K>
static int radixSortBatch(mlsgpu_ctx *ctx, const char *statName, SortJob<K> *jobs, uint32_t count, uint32_t bits, bool iota,
                          uint32_t doneBits = 0, bool keysWanted = true)
{
    REQUIRE(count >= 1 && count <= MAX_LANES, MLSGPU_ERR_INVALID);
    uint32_t maxTiles = 0;
    uint32_t tiles[MAX_LANES];
    for (uint32_t k = 0; k < count; k++)
    {
        jobs[k].result.keys = jobs[k].keysA;
        jobs[k].result.vals = jobs[k].valsA;
        tiles[k] = sortTiles(jobs[k].n);
        maxTiles = std::max(maxTiles, tiles[k]);
    }
    if (maxTiles == 0)
        return MLSGPU_OK;
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
    bool flipped = false;
    for (uint32_t p = 0; shift < bits; p++)
    {
        const uint32_t digitBits = (bits - shift) < perPass ? (bits - shift) : perPass;
        Lanes<SortHistArgs<K> > h;
        Lanes<SortDigitScanArgs> d;
        Lanes<SortScatterArgs<K> > s;
        for (uint32_t k = 0; k < MAX_LANES; k++)
        {
            const SortJob<K> &j = jobs[k < count ? k : 0];
            const uint32_t t = k < count ? tiles[k] : 0u;
            /* keysWanted = false: the caller reads the sorted VALUES only, so the last pass leaves the keys unwritten */
            const bool lastPass = shift + digitBits >= bits;
            K *const kin = flipped ? j.keysB : j.keysA, *const kout = !keysWanted && lastPass ? (K *) nullptr : flipped ? j.keysA : j.keysB;
            uint32_t *const vin = flipped ? j.valsB : j.valsA, *const vout = flipped ? j.valsA : j.valsB;
            uint32_t *const dDigitTotals = j.dHist + (uint64_t) SORT_MAX_BINS * sortTiles(j.n);
            h.a[k] = SortHistArgs<K>{kin, j.dHist, j.n, j.nDev, t};
            d.a[k] = SortDigitScanArgs{j.dHist, dDigitTotals, t};
            s.a[k] = SortScatterArgs<K>{kin, vin, kout, vout, j.dHist, dDigitTotals, j.n, j.nDev, t};
        }
        LAUNCH(ctx, statName, (sortHistKernel<K>), dim3(maxTiles, count), dim3(PRIM_BLOCK), h, shift, digitBits);
        LAUNCH(ctx, statName, (sortDigitScanKernel<uint32_t>), dim3(1u << digitBits, count), dim3(PRIM_BLOCK), d);
#define SORT_SCATTER(IOTA, BITS)                                                                                       \
        LAUNCH(ctx, statName, (sortScatterKernel<K, IOTA, BITS>), dim3(maxTiles, count), dim3(PRIM_BLOCK), s, shift, digitBits)
        /* the kernel's bin tables are sized for the digit in use: fewer bins, more resident workgroups */
        const bool first = iota && p == 0;
        if (digitBits <= 8) { if (first) SORT_SCATTER(true, 8); else SORT_SCATTER(false, 8); }
        else if (digitBits <= 9) { if (first) SORT_SCATTER(true, 9); else SORT_SCATTER(false, 9); }
        else { if (first) SORT_SCATTER(true, SortCaps<K>::MAX_DIGIT_BITS); else SORT_SCATTER(false, SortCaps<K>::MAX_DIGIT_BITS); }
#undef SORT_SCATTER
        shift += digitBits;
        flipped = !flipped;
    }
    for (uint32_t k = 0; k < count; k++)
        if (flipped && tiles[k] > 0)
        {
            jobs[k].result.keys = jobs[k].keysB;
            jobs[k].result.vals = jobs[k].valsB;
        }
    return MLSGPU_OK;
}

/* single-bucket form; dTileSums: not used any more (kept for the call sites' sake) */
template<typename K>
static int radixSort(mlsgpu_ctx *ctx, const char *statName, K *keysA, uint32_t *valsA, K *keysB, uint32_t *valsB,
                     uint64_t n, uint32_t bits, bool iota, uint32_t *dHist, uint32_t *dTileSums,
                     SortResult<K> *result, const uint32_t *nDev = nullptr, uint32_t doneBits = 0, bool keysWanted = true)
{
    (void) dTileSums;
    SortJob<K> job{keysA, valsA, keysB, valsB, n, dHist, nDev, SortResult<K>{keysA, valsA}};
    PROPAGATE(radixSortBatch<K>(ctx, statName, &job, 1, bits, iota, doneBits, keysWanted));
    *result = job.result;
    return MLSGPU_OK;
}
#endif /* __HIPCC__ */

} // namespace mlsgpu
#endif
