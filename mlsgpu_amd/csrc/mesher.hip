/*
 * Device-side mesh sink: what OOCMesher (src/mesher.cpp) does with the meshes Marching ships out -- weld of
 * external vertices by 64-bit key across buckets, connected components, pruning of components below a fraction
 * of the total vertex count, one output mesh per chunk -- for meshes that stay RESIDENT IN HBM.  Row f3 of
 * SURVEY.md section 8.
 *
 * The reference copies every ship-out to the host, runs a union-find per block plus a hash map of keys on ONE
 * mesher thread, spills to temporary files and re-reads them to write the PLY (src/mesher.cpp:220-469,763-852);
 * its manual names that thread as the scaling limit (doc/mlsgpu-user-manual.xml:508-511).  Here add() is a
 * device-to-device append of the DeviceKeyMesh into arenas (288 GB: the 13.5 GB mesh of the cfg3 noise cloud fits
 * many times) and finalize() is a handful of passes:
 *   1. externals: stable radix sort of (key, slot) -- blocks arrive grouped by chunk, so equal keys end up ordered
 *      by chunk; the first vertex of a key run is the vertex's identity for components (updateClumpKeyMap,
 *      :286-311), the first of a (key, chunk) run its identity in that chunk's file (externalRemap, :538-567);
 *   2. components: lock-free union-find over the triangles' two edges (computeLocalComponents uses the same two,
 *      :228-235) with CAS hooking, then full path compression;
 *   3. component sizes (each welded vertex once), the prune threshold uint64(total * threshold) and the
 *      keep test `>=` of getStatistics (:491-536);
 *   4. two scans compact the kept vertices and triangles; a chunk's indices are relative to its first vertex.
 * Output order is (chunk, block arrival, vertex / triangle order inside the block); the reference's differs
 * (clump order, reorder buffer) and its own tests compare up to isomorphism (test/test_mesher.cpp:401-460).
 */
#include "common.hpp"
#include "primitives.hpp"

#include <algorithm>
#include <cstdlib>
#include <mutex>

using namespace mlsgpu;

namespace
{

struct MeshRecord
{
    uint64_t chunk;
    uint32_t vBase, nv, nInternal;
    uint64_t tBase, nt;
    uint32_t eBase;
};

/* per block, on the device: what the finalize kernels look up by binary search over tBase / vBase */
struct BlockTable
{
    const uint64_t *tBase;      /* [blocks + 1] */
    const uint32_t *vBase;      /* [blocks + 1] */
    const uint32_t *chunkOf;    /* [blocks] dense chunk index */
    const uint32_t *chunkVStart;/* [chunks + 1] output vertex index where the chunk starts (after the vertex scan) */
    uint32_t blocks;

    __device__ __forceinline__ uint32_t blockOfTriangle(uint64_t t) const
    {
        uint32_t lo = 0, hi = blocks;           /* last block with tBase <= t */
        while (hi - lo > 1)
        {
            const uint32_t mid = (lo + hi) >> 1;
            if (tBase[mid] <= t) lo = mid; else hi = mid;
        }
        return lo;
    }
};

__global__ void rebaseTrianglesKernel(uint32_t *tri, uint64_t n3, uint32_t base)
{
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3)
        tri[i] += base;
}

/* copy + rebase in one pass (same-device appends): four indices per thread */
__global__ __launch_bounds__(256) void copyRebaseTrianglesKernel(uint32_t *dst, const uint32_t *src, uint64_t n3, uint32_t base)
{
    const uint64_t i = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 4 <= n3 && ((((uintptr_t) dst) | ((uintptr_t) src)) & 15) == 0)
    {
        uint4 v = *reinterpret_cast<const uint4 *>(src + i);
        v.x += base; v.y += base; v.z += base; v.w += base;
        *reinterpret_cast<uint4 *>(dst + i) = v;
    }
    else
        for (uint64_t k = i; k < n3 && k < i + 4; k++)
            dst[k] = src[k] + base;
}

__global__ void fillExternalsKernel(uint32_t *gid, uint32_t *chunk, uint64_t n, uint32_t firstGid, uint32_t chunkIndex)
{
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
    {
        gid[i] = firstGid + (uint32_t) i;
        chunk[i] = chunkIndex;
    }
}

__global__ void iotaKernel(uint32_t *a, uint64_t n)
{
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        a[i] = (uint32_t) i;
}

/* externals sorted by key (ties in arrival = chunk order): identities of every external vertex */
__global__ void externalRepsKernel(const uint64_t *keys, const uint32_t *slots, const uint32_t *extGid, const uint32_t *extChunk,
                                   uint64_t n, uint32_t *compRep, uint32_t *outRep)
{
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint64_t key = keys[i];
    const uint32_t chunk = extChunk[slots[i]];
    uint64_t c = i, o = i;
    /* runs are as long as the number of blocks that share the vertex (at most 8 for bucket corners) */
    while (c > 0 && keys[c - 1] == key)
        c--;
    while (o > 0 && keys[o - 1] == key && extChunk[slots[o - 1]] == chunk)
        o--;
    const uint32_t g = extGid[slots[i]];
    compRep[g] = extGid[slots[c]];
    outRep[g] = extGid[slots[o]];
}

/* parent[] is read while other workgroups hook roots: the loads must come from the coherence point (a line cached
 * in this CU's vector L1 would never show the new parent and the retry loop below would not end) */
__device__ __forceinline__ uint32_t loadParent(const uint32_t *parent, uint32_t v)
{
    return __hip_atomic_load(&parent[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t findRoot(const uint32_t *parent, uint32_t v)
{
    uint32_t p = loadParent(parent, v);
    while (p != v)
    {
        v = p;
        p = loadParent(parent, v);
    }
    return v;
}

/* findRoot that shortens LONG walks: a start vertex more than `shortcut` steps from its root is re-parented to the root.
 * Parents only ever point to SMALLER ids and hooks re-parent roots only, so an ancestor stays an ancestor whatever the other
 * workgroups do meanwhile: a stale or lost store costs time, never correctness, and the root of a finished component is its
 * smallest id either way (the result does not depend on the schedule).  Measured: unconditional pointer jumping (a store
 * per step) takes the shells cloud's finalize from 17.4 to 14.3 ms but the noise cloud's from 96.5 to 115.9 (its chains are
 * short already: the stores are pure cost there); the shortcut beyond 1 / 3 / 8 steps: shells 13.3 / 12.8 / 14.1 ms, noise
 * 113.9 / 103.9 / 96.3 against 17.9 and 97.4 without.
 * Round 4: the threshold is a launch parameter chosen by the size of the mesh (unionShortcut below): surface-like jobs (tens of
 * millions of vertices in a handful of sheets: long chains) take 3, the hundreds of millions of vertices of a noise cloud 8. */
__device__ __forceinline__ uint32_t findRootHalving(uint32_t *parent, uint32_t v, uint32_t shortcut)
{
    const uint32_t start = v;
    uint32_t steps = 0;
    uint32_t p = loadParent(parent, v);
    while (p != v)
    {
        v = p;
        p = loadParent(parent, v);
        steps++;
    }
    if (steps > shortcut)
        __hip_atomic_store(&parent[start], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}

/* union of the endpoints of two edges per triangle (the third is redundant, src/mesher.cpp:231-234) */
__global__ void unionKernel(const uint32_t *tri, uint64_t nt, const uint32_t *compRep, uint32_t *parent, uint32_t *failed,
                            uint32_t shortcut)
{
    const uint64_t t = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt)
        return;
    const uint32_t v[3] = {compRep[tri[3 * t]], compRep[tri[3 * t + 1]], compRep[tri[3 * t + 2]]};
#pragma unroll
    for (int e = 0; e < 2; e++)
    {
        uint32_t a = v[e], b = v[e + 1];
        for (uint32_t attempt = 0;; attempt++)
        {
            if (attempt == (1u << 20))      /* cannot happen; a bound instead of a hung GPU if it ever does */
            {
                *failed = 1;
                break;
            }
            a = findRootHalving(parent, a, shortcut);
            b = findRootHalving(parent, b, shortcut);
            if (a == b)
                break;
            if (a < b)
            {
                const uint32_t s = a; a = b; b = s;
            }
            /* hook the larger root under the smaller; retry if someone re-parented it first */
            if (atomicCAS(&parent[a], a, b) == a)
                break;
        }
    }
}

__global__ void compressKernel(uint32_t *parent, const uint32_t *compRep, uint64_t n, uint32_t *root)
{
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        root[i] = findRoot(parent, compRep[i]);
}

/* vertices of a component: every welded vertex once.  Neighbouring vertices mostly share their component (a noise
 * cloud is ONE component of hundreds of millions of vertices, and 6 million same-address atomics take 67 ms), so each
 * wave walks a contiguous span and keeps one pending (root, count) in registers; it only touches memory when the root
 * changes. */
#define SIZE_SPAN 64        /* steps of 64 vertices per wave */
__global__ __launch_bounds__(256) void componentSizeKernel(const uint32_t *compRep, const uint32_t *root, uint64_t n, uint32_t *size)
{
    const uint64_t wave = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t first = wave * (64 * SIZE_SPAN);
    uint32_t pendingRoot = 0, pendingCount = 0;         /* wave-uniform */
    for (uint32_t s = 0; s < SIZE_SPAN; s++)
    {
        const uint64_t i = first + (uint64_t) s * 64 + laneId();
        if (first + (uint64_t) s * 64 >= n)
            break;
        const bool counts = i < n && compRep[i] == (uint32_t) i;
        const uint32_t mine = counts ? root[i] : 0u;
        uint64_t todo = __ballot(counts);
        while (todo != 0)
        {
            const uint32_t r = readLane(mine, (int) __builtin_ctzll(todo));
            const uint64_t same = __ballot(counts && mine == r) & todo;
            const uint32_t c = (uint32_t) __popcll(same);
            if (pendingCount != 0 && r != pendingRoot)
            {
                if (laneId() == 0)
                    atomicAdd(&size[pendingRoot], pendingCount);
                pendingCount = 0;
            }
            pendingRoot = r;
            pendingCount += c;
            todo &= ~same;
        }
    }
    if (pendingCount != 0 && laneId() == 0)
        atomicAdd(&size[pendingRoot], pendingCount);
}

/* [0] welded vertices, [1] components, [2] kept components, [3] kept vertices */
__global__ __launch_bounds__(256) void componentStatsKernel(const uint32_t *compRep, const uint32_t *root, const uint32_t *size,
                                                           uint64_t n, uint64_t threshold, unsigned long long *stats, int pass)
{
    /* grid-stride with per-thread sums, one atomic per wave at the end */
    unsigned long long reps = 0, comps = 0, kept = 0, keptV = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
    {
        const bool rep = compRep[i] == (uint32_t) i;
        if (pass == 0)
            reps += rep ? 1 : 0;
        else if (rep && root[i] == (uint32_t) i)
        {
            comps++;
            if (size[i] >= threshold)
            {
                kept++;
                keptV += size[i];
            }
        }
    }
    const unsigned long long vals[4] = {reps, comps, kept, keptV};
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        unsigned long long v = vals[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
            v += __shfl_xor(v, d, 64);
        if (laneId() == 0 && v != 0)
            atomicAdd(&stats[k], v);
    }
}

struct KeepVertexIn
{
    const uint32_t *outRep, *root, *size;
    uint64_t threshold;
    __device__ __forceinline__ uint32_t operator()(uint64_t i) const
    {
        return (outRep[i] == (uint32_t) i && size[root[i]] >= threshold) ? 1u : 0u;
    }
};

struct VertexOut
{
    const float *vertices;
    float *out;
    uint32_t *index;
    __device__ __forceinline__ void operator()(uint64_t i, uint32_t excl, uint32_t keep) const
    {
        index[i] = excl;
        if (keep)
        {
            out[3 * (uint64_t) excl + 0] = vertices[3 * i + 0];
            out[3 * (uint64_t) excl + 1] = vertices[3 * i + 1];
            out[3 * (uint64_t) excl + 2] = vertices[3 * i + 2];
        }
    }
};

struct KeepTriangleIn
{
    const uint32_t *tri, *root, *size;
    uint64_t threshold;
    __device__ __forceinline__ uint32_t operator()(uint64_t t) const
    {
        return size[root[tri[3 * t]]] >= threshold ? 1u : 0u;
    }
};

struct TriangleOut
{
    const uint32_t *tri, *outRep, *vIndex;
    BlockTable B;
    uint32_t *out;
    __device__ __forceinline__ void operator()(uint64_t t, uint32_t excl, uint32_t keep) const
    {
        if (!keep)
            return;
        const uint32_t first = B.chunkVStart[B.chunkOf[B.blockOfTriangle(t)]];
#pragma unroll
        for (int j = 0; j < 3; j++)
            out[3 * (uint64_t) excl + j] = vIndex[outRep[tri[3 * t + j]]] - first;
    }
};

struct TriangleOutIdx
{
    TriangleOut inner;
    uint32_t *tIndex;
    __device__ __forceinline__ void operator()(uint64_t t, uint32_t excl, uint32_t keep) const
    {
        tIndex[t] = excl;
        inner(t, excl, keep);
    }
};

__global__ void gatherU32Kernel(const uint32_t *src, const uint64_t *at, uint32_t n, uint64_t limit, uint32_t last, uint32_t *dst)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        dst[i] = at[i] < limit ? src[at[i]] : last;
}

/* ---- boundary export for several meshers, one job (see mlsgpu_hip_mesher_boundary) ---- */

/* the roots of the components, densely numbered in vertex order */
struct IsRootIn
{
    const uint32_t *compRep, *root;
    __device__ __forceinline__ uint32_t operator()(uint64_t i) const
    {
        return (compRep[i] == (uint32_t) i && root[i] == (uint32_t) i) ? 1u : 0u;
    }
};
struct RootOut
{
    uint32_t *rootId, *denseOf;
    __device__ __forceinline__ void operator()(uint64_t i, uint32_t excl, uint32_t isRoot) const
    {
        denseOf[i] = excl;
        if (isRoot)
            rootId[excl] = (uint32_t) i;
    }
};

/* triangles per component, by the span trick of componentSizeKernel (a noise cloud is one component) */
__global__ __launch_bounds__(256) void componentTrianglesKernel(const uint32_t *tri, const uint32_t *root, const uint32_t *denseOf,
                                                                uint64_t nt, uint32_t *count)
{
    const uint64_t wave = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t first = wave * (64 * SIZE_SPAN);
    uint32_t pendingRoot = 0, pendingCount = 0;
    for (uint32_t s = 0; s < SIZE_SPAN; s++)
    {
        const uint64_t t = first + (uint64_t) s * 64 + laneId();
        if (first + (uint64_t) s * 64 >= nt)
            break;
        const bool counts = t < nt;
        const uint32_t mine = counts ? denseOf[root[tri[3 * t]]] : 0u;
        uint64_t todo = __ballot(counts);
        while (todo != 0)
        {
            const uint32_t r = readLane(mine, (int) __builtin_ctzll(todo));
            const uint64_t same = __ballot(counts && mine == r) & todo;
            if (pendingCount != 0 && r != pendingRoot)
            {
                if (laneId() == 0)
                    atomicAdd(&count[pendingRoot], pendingCount);
                pendingCount = 0;
            }
            pendingRoot = r;
            pendingCount += (uint32_t) __popcll(same);
            todo &= ~same;
        }
    }
    if (pendingCount != 0 && laneId() == 0)
        atomicAdd(&count[pendingRoot], pendingCount);
}

/* The same count for FEW components (surface-like data: a handful of sheets whose triangles alternate along every row of
 * cells, so the run-length trick above flushes at almost every step and a million global atomics land on a dozen
 * addresses): per-workgroup bins in LDS, one global atomic per bin and workgroup. */
#define FEW_COMPONENTS 2048
__global__ __launch_bounds__(256) void componentTrianglesFewKernel(const uint32_t *tri, const uint32_t *root, const uint32_t *denseOf,
                                                                   uint64_t nt, uint32_t numRoots, uint32_t *count)
{
    __shared__ uint32_t bins[FEW_COMPONENTS];
    for (uint32_t d = threadIdx.x; d < numRoots; d += 256)
        bins[d] = 0;
    __syncthreads();
    const uint64_t wave = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t first = wave * (64 * SIZE_SPAN);
    for (uint32_t s = 0; s < SIZE_SPAN; s++)
    {
        const uint64_t t = first + (uint64_t) s * 64 + laneId();
        if (first + (uint64_t) s * 64 >= nt)
            break;
        const bool counts = t < nt;
        const uint32_t mine = counts ? denseOf[root[tri[3 * t]]] : 0u;
        uint64_t todo = __ballot(counts);
        while (todo != 0)
        {
            const uint32_t r = readLane(mine, (int) __builtin_ctzll(todo));
            const uint64_t same = __ballot(counts && mine == r) & todo;
            if (laneId() == 0)
                atomicAdd(&bins[r], (uint32_t) __popcll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < numRoots; d += 256)
        if (bins[d] != 0)
            atomicAdd(&count[d], bins[d]);
}

__global__ void gatherSizesKernel(const uint32_t *rootId, const uint32_t *size, uint32_t n, uint32_t *out)
{
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < n)
        out[d] = size[rootId[d]];
}

/* every distinct external key (the keys are sorted) with the dense root of its vertex */
struct FirstOfRunIn
{
    const uint64_t *keys;
    __device__ __forceinline__ uint32_t operator()(uint64_t i) const { return (i == 0 || keys[i - 1] != keys[i]) ? 1u : 0u; }
};
struct KeyRootOut
{
    const uint64_t *keys;
    const uint32_t *slots, *extGid, *root, *denseOf;
    uint64_t *keyOut;
    uint32_t *rootOut;
    __device__ __forceinline__ void operator()(uint64_t i, uint32_t excl, uint32_t first) const
    {
        if (first)
        {
            keyOut[excl] = keys[i];
            rootOut[excl] = denseOf[root[extGid[slots[i]]]];
        }
    }
};

/* the caller's verdict per dense root becomes the "size" the output pass tests against a threshold of 1 */
__global__ void applyVerdictKernel(const uint32_t *rootId, const uint8_t *keep, uint32_t n, uint32_t *size)
{
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < n)
        size[rootId[d]] = keep[d] ? 0xFFFFFFFFu : 0u;
}

template<typename T>
struct Arena
{
    T *ptr = nullptr;
    uint64_t used = 0, cap = 0;

    /* the mesher's device must be current; `stream` is one of its streams */
    int reserve(hipStream_t stream, uint64_t need)
    {
        if (need <= cap)
            return MLSGPU_OK;
        const uint64_t newCap = std::max<uint64_t>(need, cap + cap / 2 + 1024);
        T *np = nullptr;
        HIP_CHECK(hipMalloc((void **) &np, newCap * sizeof(T)));
        if (used > 0)
        {
            hipError_t e = hipMemcpyAsync(np, ptr, used * sizeof(T), hipMemcpyDeviceToDevice, stream);
            if (e == hipSuccess)
                e = hipStreamSynchronize(stream);
            if (e != hipSuccess)
            {
                hipFree(np);
                return setError(MLSGPU_ERR_HIP, "mesher: arena move failed: %s", hipGetErrorString(e));
            }
        }
        hipFree(ptr);
        ptr = np;
        cap = newCap;
        return MLSGPU_OK;
    }
    ~Arena() { hipFree(ptr); }
};

} // namespace

namespace
{
/* bump allocator over a slab the mesher keeps between calls: allocating and freeing tens of GB of scratch costs
 * several times the kernels that use it */
struct Scratch
{
    char *base;
    uint64_t cap, used = 0;
    Scratch(void *base, uint64_t cap) : base(static_cast<char *>(base)), cap(cap) {}
    template<typename T> int get(T **p, uint64_t n)
    {
        const uint64_t bytes = (std::max<uint64_t>(n, 1) * sizeof(T) + 255) & ~uint64_t(255);
        if (used + bytes > cap)
            return setError(MLSGPU_ERR_NOMEM, "mesher: scratch slab too small (%llu of %llu bytes used, %llu more wanted)",
                            (unsigned long long) used, (unsigned long long) cap, (unsigned long long) bytes);
        *p = reinterpret_cast<T *>(base + used);
        used += bytes;
        return MLSGPU_OK;
    }
};

uint64_t scratchBytes(uint64_t nv, uint64_t nt, uint64_t ne, uint64_t nb, uint64_t nc)
{
    auto al = [](uint64_t b) { return (std::max<uint64_t>(b, 1) + 255) & ~uint64_t(255); };
    uint64_t total = 6 * al(nv * 4) + al(nt * 4);                                   /* per-vertex arrays, tIndex */
    total += 2 * al(ne * 8) + 2 * al(ne * 4) + al(sortHistElems(ne) * 4)            /* key sort */
        + al(scanTiles(std::max<uint64_t>(sortHistElems(ne), ne)) * 4);
    total += al(scanTiles(std::max(nv, nt)) * 4);                                   /* scans */
    total += 2 * al((nb + 1) * 8) + 2 * al((nb + 1) * 4) + 3 * al((nc + 1) * 8) + 8 * 256;
    total += 2 * al(scanTiles(nv) * 4) + al(scanTiles(ne) * 4) + al(nv);            /* boundary export / verdict */
    total += 2 * al(nv * 4);                                                        /* ... vertices and triangles per component */
    return total + (1 << 20);
}
} // namespace

struct mlsgpu_mesher
{
    mlsgpu_ctx *ctx = nullptr;
    std::mutex mutex;
    double pruneThreshold = 0.0;
    bool background = false;        /* finalize shares the GPU with other work: see the union-find launch */
    Arena<float> vertices;          /* 3 per vertex */
    Arena<uint32_t> triangles;      /* 3 per triangle, global vertex ids */
    Arena<uint64_t> extKeys;
    Arena<uint32_t> extGid, extChunk;
    std::vector<MeshRecord> blocks;
    std::vector<uint64_t> chunkIds;             /* dense index -> caller's id, arrival order */
    bool finalized = false;
    bool peerEnabled[16] = {};                  /* peer access towards the GPUs whose workers have appended */
    /* What the last analysis left in the slab, valid while nothing is added (same arenas):
     *   1  after boundary(): the weld and the components (representatives, roots, sizes) AND the dense root numbering -- a
     *      finalize_with() right behind it starts from them instead of sorting and uniting again (one use: the verdict
     *      overwrites the sizes);
     *   2  after a plain finalize(): the weld and the components (the output pass only rewrote the vertex index array) -- a
     *      boundary() behind it numbers the roots and exports, without the key sort and the union-find;
     *   0  nothing. */
    int cacheKind = 0;
    bool cacheSortedInA = false;                /* which of the sort's two buffers holds the sorted external keys */
    uint64_t cacheDims[3] = {0, 0, 0};
    uint32_t cacheRootCount = 0;
    /* results */
    float *outVertices = nullptr;
    uint32_t *outTriangles = nullptr;
    std::vector<uint64_t> chunkVStart, chunkTStart;     /* [chunks + 1] */
    std::vector<uint32_t> outChunks;                    /* dense chunk indices that have triangles */
    uint64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    /* boundary export (mlsgpu_hip_mesher_boundary): valid until the next add */
    bool analyzed = false;
    /* the export's keys and their roots land in PINNED memory (tens of MB: a copy into pageable memory goes through the
     * runtime's staging buffers and took anything from 5 to 80 ms) */
    template<typename T>
    struct PinnedArray
    {
        T *p = nullptr;
        size_t n = 0, cap = 0;
        ~PinnedArray() { if (p) hipHostFree(p); }
        int resize(size_t count)
        {
            if (count > cap)
            {
                if (p) hipHostFree(p);
                p = nullptr;
                cap = 0;
                const size_t want = count + count / 4 + 1024;
                if (hipHostMalloc((void **) &p, want * sizeof(T)) != hipSuccess)
                    return setError(MLSGPU_ERR_NOMEM, "mesher: cannot allocate %zu bytes of pinned export buffer", want * sizeof(T));
                cap = want;
            }
            n = count;
            return MLSGPU_OK;
        }
        void clear() { n = 0; }
        size_t size() const { return n; }
        bool empty() const { return n == 0; }
        T *data() { return p; }
        const T *begin() const { return p; }
        const T *end() const { return p + n; }
    };
    PinnedArray<uint64_t> bKeys;
    PinnedArray<uint32_t> bKeyRoot;
    std::vector<uint64_t> bRootVertices, bRootTriangles;

    /* kept between finalize calls; grown on demand (or up front by reserve) */
    void *slab = nullptr;
    uint64_t slabCap = 0;
    uint64_t outVCap = 0, outTCap = 0;

    int ensureSlab(uint64_t bytes)
    {
        if (bytes <= slabCap)
            return MLSGPU_OK;
        cacheKind = 0;          /* whatever an analysis left in the old slab is gone */
        hipFree(slab);
        slab = nullptr;
        slabCap = 0;
        HIP_CHECK(hipMalloc(&slab, bytes));
        slabCap = bytes;
        return MLSGPU_OK;
    }
    int ensureOutputs(uint64_t nv, uint64_t nt)
    {
        if (outVCap < nv)
        {
            hipFree(outVertices);
            outVertices = nullptr;
            outVCap = 0;
            HIP_CHECK(hipMalloc((void **) &outVertices, std::max<uint64_t>(3 * nv, 1) * sizeof(float)));
            outVCap = nv;
        }
        if (outTCap < nt)
        {
            hipFree(outTriangles);
            outTriangles = nullptr;
            outTCap = 0;
            HIP_CHECK(hipMalloc((void **) &outTriangles, std::max<uint64_t>(3 * nt, 1) * sizeof(uint32_t)));
            outTCap = nt;
        }
        return MLSGPU_OK;
    }
    /* a stream of the mesher's own: add() is called from worker threads whose contexts (and devices) differ, and the
     * mesher's context belongs to the thread that finalizes; everything on it runs under the mutex */
    hipStream_t addStream = nullptr;
    int ensureAddStream()
    {
        if (addStream == nullptr)
            HIP_CHECK(hipStreamCreateWithFlags(&addStream, hipStreamNonBlocking));
        return MLSGPU_OK;
    }
    /* appends that are still running on their producers' streams (same-device appends do not wait on the host: the
     * producer's next kernels are behind the copies in stream order).  Waited for before an arena moves, before any
     * analysis and before a reset. */
    struct PendingAppend
    {
        int device;
        hipEvent_t done;
    };
    std::vector<PendingAppend> pending, eventPool;
    int takeEvent(int device, hipEvent_t *ev)      /* `device` is current */
    {
        for (size_t i = 0; i < eventPool.size(); i++)
            if (eventPool[i].device == device)
            {
                *ev = eventPool[i].done;
                eventPool.erase(eventPool.begin() + (long) i);
                return MLSGPU_OK;
            }
        HIP_CHECK(hipEventCreateWithFlags(ev, hipEventDisableTiming));
        return MLSGPU_OK;
    }
    int drainPending(size_t keep = 0)
    {
        while (pending.size() > keep)
        {
            const hipError_t e = hipEventSynchronize(pending.front().done);
            eventPool.push_back(pending.front());
            pending.erase(pending.begin());
            if (e != hipSuccess)
                return setError(MLSGPU_ERR_HIP, "mesher: an append failed: %s", hipGetErrorString(e));
        }
        return MLSGPU_OK;
    }
    int regroupByChunk();
    void dropResults() { finalized = false; }
    int finalizeImpl(uint32_t *numChunks, bool analyzeOnly, const uint8_t *keepRoots, uint64_t numRoots);
    ~mlsgpu_mesher()
    {
        hipFree(outVertices);
        hipFree(outTriangles);
        hipFree(slab);
        if (addStream) hipStreamDestroy(addStream);
        drainPending();
        for (auto &e : eventPool)
            hipEventDestroy(e.done);
    }
};

MLSGPU_API int mlsgpu_hip_mesher_create(mlsgpu_ctx *ctx, mlsgpu_mesher **out)
{
    REQUIRE(ctx != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    mlsgpu_mesher *m = new mlsgpu_mesher;
    m->ctx = ctx;
    *out = m;
    return MLSGPU_OK;
}

MLSGPU_API void mlsgpu_hip_mesher_destroy(mlsgpu_mesher *m) { delete m; }

/* on = 1: this sink's finalize usually runs while other work (the next job's buckets) is on the GPU: its union-find pass
 * holds back to a quarter of the wave slots.  Shells cloud, jobs in rotation: 40.6 -> 37.1 ms per job (the ship-out route:
 * 36.3); a finalize that has the GPU to itself takes 11-14 ms either way, the noise cloud's (378 M vertices) 97 -> 108 ms. */
MLSGPU_API int mlsgpu_hip_mesher_set_background(mlsgpu_mesher *m, int on)
{
    REQUIRE(m != nullptr, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    m->background = on != 0;
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_mesher_set_prune_threshold(mlsgpu_mesher *m, double threshold)
{
    REQUIRE(m != nullptr && threshold >= 0.0 && threshold <= 1.0, MLSGPU_ERR_INVALID);
    m->pruneThreshold = threshold;
    return MLSGPU_OK;
}

/* room for the whole job up front: an arena that has to grow is reallocated and copied */
MLSGPU_API int mlsgpu_hip_mesher_reserve(mlsgpu_mesher *m, uint64_t numVertices, uint64_t numTriangles, uint64_t numExternal)
{
    REQUIRE(m != nullptr, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    HIP_CHECK(hipSetDevice(m->ctx->device));
    PROPAGATE(m->ensureAddStream());
    /* an arena that grows moves (copy on addStream, old buffer freed): same-device appends may still be running on their
     * producers' streams, so they must have landed first -- the rule mlsgpu_hip_mesher_add follows before it grows one */
    if (3 * numVertices > m->vertices.cap || 3 * numTriangles > m->triangles.cap || numExternal > m->extKeys.cap
        || numExternal > m->extGid.cap || numExternal > m->extChunk.cap)
        PROPAGATE(m->drainPending());
    PROPAGATE(m->vertices.reserve(m->addStream, 3 * numVertices));
    PROPAGATE(m->triangles.reserve(m->addStream, 3 * numTriangles));
    PROPAGATE(m->extKeys.reserve(m->addStream, numExternal));
    PROPAGATE(m->extGid.reserve(m->addStream, numExternal));
    PROPAGATE(m->extChunk.reserve(m->addStream, numExternal));
    /* finalize's scratch and outputs too (a few hundred blocks and chunks are assumed; finalize grows it otherwise) */
    PROPAGATE(m->ensureSlab(scratchBytes(numVertices, numTriangles, numExternal, 4096, 4096)));
    PROPAGATE(m->ensureOutputs(numVertices, numTriangles));
    return MLSGPU_OK;
}

/* MesherBase::InputFunctor (src/mesher.h:204-210) for a mesh that is still on a device */
MLSGPU_API int mlsgpu_hip_mesher_add(mlsgpu_mesher *m, mlsgpu_ctx *from, uint64_t chunkId, const mlsgpu_mesh *mesh)
{
    REQUIRE(m != nullptr && from != nullptr && mesh != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(mesh->numInternalVertices <= mesh->numVertices, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    REQUIRE(!m->finalized, MLSGPU_ERR_INVALID);
    const int home = m->ctx->device;
    static const bool forcePeer = [] {          /* tests: the peer route on one GPU; said once, it synchronises every append */
        const bool on = getenv("MLSGPU_HIP_MESHER_FORCE_PEER") != nullptr;
        if (on)
            fprintf(stderr, "mlsgpu_hip: MLSGPU_HIP_MESHER_FORCE_PEER is set: ship-outs take the peer route (test hook)\n");
        return on;
    }();
    const bool peer = from->device != home || forcePeer;
    DeviceGuard restore;        /* the caller is a worker in the middle of ITS device's work (Marching's output functor) */
    if (from->device != home && !m->peerEnabled[from->device & 15])
    {
        enablePeerAccess(home, from->device);
        m->peerEnabled[from->device & 15] = true;
    }
    HIP_CHECK(hipSetDevice(home));
    PROPAGATE(m->ensureAddStream());
    /* OOCMesher::add indexes chunks[chunkId.gen] (src/mesher.cpp:380-384): any arrival order; dense index = first arrival */
    uint32_t chunk = 0;
    while (chunk < m->chunkIds.size() && m->chunkIds[chunk] != chunkId)
        chunk++;
    const bool newChunk = chunk == m->chunkIds.size();     /* recorded with the block, once everything has succeeded */
    const uint64_t nv = mesh->numVertices, nt = mesh->numTriangles, ne = nv - mesh->numInternalVertices;
    REQUIRE(m->vertices.used / 3 + nv < (uint64_t(1) << 32), MLSGPU_ERR_LENGTH);
    /* an arena that has to grow moves: every earlier append must have landed first (reserve() up front avoids both) */
    if (m->vertices.used + 3 * nv > m->vertices.cap || m->triangles.used + 3 * nt > m->triangles.cap
        || m->extKeys.used + ne > m->extKeys.cap || m->extGid.used + ne > m->extGid.cap || m->extChunk.used + ne > m->extChunk.cap)
        PROPAGATE(m->drainPending());
    PROPAGATE(m->vertices.reserve(m->addStream, m->vertices.used + 3 * nv));
    PROPAGATE(m->triangles.reserve(m->addStream, m->triangles.used + 3 * nt));
    PROPAGATE(m->extKeys.reserve(m->addStream, m->extKeys.used + ne));
    PROPAGATE(m->extGid.reserve(m->addStream, m->extGid.used + ne));
    PROPAGATE(m->extChunk.reserve(m->addStream, m->extChunk.used + ne));
    MeshRecord r;
    r.chunk = chunk;
    r.vBase = (uint32_t) (m->vertices.used / 3);
    r.nv = (uint32_t) nv;
    r.nInternal = (uint32_t) mesh->numInternalVertices;
    r.tBase = m->triangles.used / 3;
    r.nt = nt;
    r.eBase = (uint32_t) m->extKeys.used;
    /* the copies run on the producing worker's stream, behind the kernels that made the mesh; from another GPU they are
     * peer copies over the fabric.  The index fix-ups run on the mesher's device, after the copies. */
    HIP_CHECK(hipSetDevice(from->device));
    auto append = [&](void *dst, const void *src, size_t bytes) -> hipError_t
    {
        if (bytes == 0)
            return hipSuccess;
        return peer ? hipMemcpyPeerAsync(dst, home, src, from->device, bytes, from->stream)
                    : hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, from->stream);
    };
    /* A failure after the first enqueue leaves work on the producer's stream that writes [used, used + n) of the arenas
     * while `used` does not advance: the next append (possibly from another worker's stream) would write the same range
     * under it, and finalize / reset would not wait for it.  Every failing path therefore drains the producer's stream
     * (and the mesher's, for a peer append) before it reports. */
    hipStream_t fix = from->stream;
    auto enqueue = [&]() -> int
    {
        HIP_CHECK(append(m->vertices.ptr + m->vertices.used, mesh->dVertices, 3 * nv * sizeof(float)));
        if (peer)
            HIP_CHECK(append(m->triangles.ptr + m->triangles.used, mesh->dTriangles, 3 * nt * sizeof(uint32_t)));
        else if (nt > 0)
            hipLaunchKernelGGL(copyRebaseTrianglesKernel, dim3(divUp(divUp(3 * nt, 4), 256)), dim3(256), 0, from->stream,
                               m->triangles.ptr + m->triangles.used, mesh->dTriangles, 3 * nt, (uint32_t) (m->vertices.used / 3));
        HIP_CHECK(append(m->extKeys.ptr + m->extKeys.used, mesh->dVertexKeys + mesh->numInternalVertices, ne * sizeof(uint64_t)));
        if (peer)
        {
            HIP_CHECK(hipStreamSynchronize(from->stream));
            HIP_CHECK(hipSetDevice(home));
            fix = m->addStream;
        }
        if (nt > 0 && peer)
            hipLaunchKernelGGL(rebaseTrianglesKernel, dim3(divUp(3 * nt, 256)), dim3(256), 0, fix,
                               m->triangles.ptr + m->triangles.used, 3 * nt, r.vBase);
        if (ne > 0)
            hipLaunchKernelGGL(fillExternalsKernel, dim3(divUp(ne, 256)), dim3(256), 0, fix,
                               m->extGid.ptr + m->extGid.used, m->extChunk.ptr + m->extChunk.used, ne, r.vBase + r.nInternal, chunk);
        HIP_CHECK(hipGetLastError());
        return MLSGPU_OK;
    };
    {
        const int rcEnqueue = enqueue();
        if (rcEnqueue != MLSGPU_OK)
        {
            (void) hipStreamSynchronize(from->stream);
            if (fix != from->stream)
                (void) hipStreamSynchronize(fix);
            return rcEnqueue;
        }
    }
    if (peer)
    {
        /* the fix-ups ran on the mesher's stream, not the producer's: the mesh is Marching's and is reused for the next
         * ship-out, so they must have finished */
        HIP_CHECK(hipStreamSynchronize(fix));
    }
    else
    {
        /* same device: copies and fix-ups are on the PRODUCER's stream, ahead of whatever it does to the mesh next -- the
         * worker goes on without waiting; the mesher remembers that this append is still in flight */
        mlsgpu_mesher::PendingAppend p{from->device, nullptr};
        PROPAGATE(m->takeEvent(from->device, &p.done));
        if (hipEventRecord(p.done, fix) != hipSuccess)
        {
            m->eventPool.push_back(p);
            HIP_CHECK(hipStreamSynchronize(fix));
        }
        else
        {
            m->pending.push_back(p);
            PROPAGATE(m->drainPending(512));       /* bounds the list on jobs of thousands of ship-outs */
        }
    }
    m->vertices.used += 3 * nv;
    m->triangles.used += 3 * nt;
    m->extKeys.used += ne;
    m->extGid.used += ne;
    m->extChunk.used += ne;
    if (newChunk)
        m->chunkIds.push_back(chunkId);
    m->blocks.push_back(r);
    m->analyzed = false;
    m->cacheKind = 0;
    return MLSGPU_OK;
}

/* a mlsgpu_farm_output_fn whose `user` is the mesher: the farm's device workers append their ship-outs */
MLSGPU_API int mlsgpu_hip_mesher_farm_output(void *mesher, int device, uint64_t chunkId, mlsgpu_ctx *ctx, const mlsgpu_mesh *mesh)
{
    (void) device;
    return mlsgpu_hip_mesher_add(static_cast<mlsgpu_mesher *>(mesher), ctx, chunkId, mesh);
}

/* empties the mesher (arenas, scratch and outputs keep their capacity): the next job starts from nothing */
MLSGPU_API int mlsgpu_hip_mesher_reset(mlsgpu_mesher *m)
{
    REQUIRE(m != nullptr, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    PROPAGATE(m->drainPending());
    m->vertices.used = m->triangles.used = 0;
    m->extKeys.used = m->extGid.used = m->extChunk.used = 0;
    m->blocks.clear();
    m->chunkIds.clear();
    m->outChunks.clear();
    m->finalized = false;
    m->analyzed = false;
    m->cacheKind = 0;
    return MLSGPU_OK;
}

/* finalize wants the blocks of a chunk adjacent in the arenas (output order is arena order, and equal keys must sort by
 * chunk).  With several workers and several chunks they arrive interleaved: move the blocks into (chunk by first
 * arrival, arrival) order -- device-to-device copies into fresh arenas, vertex ids shifted by the block's move. */
int mlsgpu_mesher::regroupByChunk()
{
    bool grouped = true;
    for (size_t b = 1; b < blocks.size(); b++)
        grouped = grouped && blocks[b - 1].chunk <= blocks[b].chunk;
    if (grouped)
        return MLSGPU_OK;
    std::vector<uint32_t> order(blocks.size());
    for (uint32_t i = 0; i < order.size(); i++)
        order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return blocks[a].chunk < blocks[b].chunk; });
    PROPAGATE(ensureAddStream());
    Arena<float> nVertices;
    Arena<uint32_t> nTriangles, nGid, nChunk;
    Arena<uint64_t> nKeys;
    PROPAGATE(nVertices.reserve(addStream, std::max<uint64_t>(vertices.cap, 1)));
    PROPAGATE(nTriangles.reserve(addStream, std::max<uint64_t>(triangles.cap, 1)));
    PROPAGATE(nKeys.reserve(addStream, std::max<uint64_t>(extKeys.cap, 1)));
    PROPAGATE(nGid.reserve(addStream, std::max<uint64_t>(extGid.cap, 1)));
    PROPAGATE(nChunk.reserve(addStream, std::max<uint64_t>(extChunk.cap, 1)));
    std::vector<MeshRecord> moved;
    for (uint32_t idx : order)
    {
        const MeshRecord &o = blocks[idx];
        MeshRecord r = o;
        r.vBase = (uint32_t) (nVertices.used / 3);
        r.tBase = nTriangles.used / 3;
        r.eBase = (uint32_t) nKeys.used;
        const uint64_t ne = o.nv - o.nInternal;
        if (o.nv > 0)
            HIP_CHECK(hipMemcpyAsync(nVertices.ptr + nVertices.used, vertices.ptr + 3 * (uint64_t) o.vBase, 3 * (uint64_t) o.nv * sizeof(float),
                                     hipMemcpyDeviceToDevice, addStream));
        if (o.nt > 0)
        {
            HIP_CHECK(hipMemcpyAsync(nTriangles.ptr + nTriangles.used, triangles.ptr + 3 * o.tBase, 3 * o.nt * sizeof(uint32_t),
                                     hipMemcpyDeviceToDevice, addStream));
            hipLaunchKernelGGL(rebaseTrianglesKernel, dim3(divUp(3 * o.nt, 256)), dim3(256), 0, addStream,
                               nTriangles.ptr + nTriangles.used, 3 * o.nt, r.vBase - o.vBase);      /* modulo 2^32 */
        }
        if (ne > 0)
        {
            HIP_CHECK(hipMemcpyAsync(nKeys.ptr + nKeys.used, extKeys.ptr + o.eBase, ne * sizeof(uint64_t), hipMemcpyDeviceToDevice,
                                     addStream));
            hipLaunchKernelGGL(fillExternalsKernel, dim3(divUp(ne, 256)), dim3(256), 0, addStream, nGid.ptr + nGid.used,
                               nChunk.ptr + nChunk.used, ne, r.vBase + r.nInternal, (uint32_t) r.chunk);
        }
        nVertices.used += 3 * (uint64_t) o.nv;
        nTriangles.used += 3 * o.nt;
        nKeys.used += ne;
        nGid.used += ne;
        nChunk.used += ne;
        moved.push_back(r);
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(addStream));
    std::swap(vertices.ptr, nVertices.ptr);   std::swap(vertices.cap, nVertices.cap);
    std::swap(triangles.ptr, nTriangles.ptr); std::swap(triangles.cap, nTriangles.cap);
    std::swap(extKeys.ptr, nKeys.ptr);        std::swap(extKeys.cap, nKeys.cap);
    std::swap(extGid.ptr, nGid.ptr);          std::swap(extGid.cap, nGid.cap);
    std::swap(extChunk.ptr, nChunk.ptr);      std::swap(extChunk.cap, nChunk.cap);
    blocks.swap(moved);
    return MLSGPU_OK;
}


/* MesherBase::write's finalisation (src/mesher.cpp:763-852) up to the point where files are written */
MLSGPU_API int mlsgpu_hip_mesher_finalize(mlsgpu_mesher *m, uint32_t *numChunks)
{
    REQUIRE(m != nullptr, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    return m->finalizeImpl(numChunks, false, nullptr, 0);
}

/* analyzeOnly: weld + components, then the boundary export instead of any output.  keepRoots: the caller's verdict per
 * dense root (the numbering of the last export; the union-find hooks larger roots under smaller ones, so a component's
 * root is its smallest welded vertex whatever the schedule and the numbering is reproducible) replaces the prune rule. */
int mlsgpu_mesher::finalizeImpl(uint32_t *numChunks, bool analyzeOnly, const uint8_t *keepRoots, uint64_t numRoots)
{
    mlsgpu_mesher *const m = this;
    mlsgpu_ctx *ctx = m->ctx;
    PROPAGATE(m->drainPending());
    HIP_CHECK(hipSetDevice(ctx->device));
    m->dropResults();
    PROPAGATE(m->regroupByChunk());
    const uint64_t nv = m->vertices.used / 3, nt = m->triangles.used / 3, ne = m->extKeys.used;
    const uint32_t nb = (uint32_t) m->blocks.size(), nc = (uint32_t) m->chunkIds.size();
    /* the verdict pass right behind an export: weld, components and root numbering are still in the slab */
    const bool sameArenas = m->cacheDims[0] == nv && m->cacheDims[1] == nt && m->cacheDims[2] == ne;
    bool reuseDense = keepRoots != nullptr && !analyzeOnly && m->cacheKind == 1 && sameArenas;
    /* the weld and the components are in the slab: after an export (for the verdict pass, or for another export) or after a
     * plain finalize (for an export) */
    bool reuse = reuseDense || (analyzeOnly && m->cacheKind != 0 && sameArenas);
    m->cacheKind = 0;
    m->chunkVStart.assign(nc + 1, 0);
    m->chunkTStart.assign(nc + 1, 0);
    m->outChunks.clear();
    std::fill(m->stats, m->stats + 8, 0);
    if (analyzeOnly)
    {
        m->bKeys.clear();
        m->bKeyRoot.clear();
        m->bRootVertices.clear();
        m->bRootTriangles.clear();
        m->analyzed = true;
    }
    if (nv == 0 || nt == 0)
    {
        m->finalized = !analyzeOnly;
        if (numChunks)
            *numChunks = 0;
        return MLSGPU_OK;
    }
    {
        const void *const slabBefore = m->slab;
        PROPAGATE(m->ensureSlab(scratchBytes(nv, nt, ne, nb, nc)));
        if (m->slab != slabBefore)
            reuse = reuseDense = false;         /* (cannot happen for unchanged arenas; the cached analysis lived there) */
    }
    PROPAGATE(m->ensureOutputs(nv, nt));
    Scratch S(m->slab, m->slabCap);
    uint32_t *compRep, *outRep, *parent, *root, *size, *vIndex;
    PROPAGATE(S.get(&compRep, nv));
    PROPAGATE(S.get(&outRep, nv));
    PROPAGATE(S.get(&parent, nv));
    PROPAGATE(S.get(&root, nv));
    PROPAGATE(S.get(&size, nv));
    PROPAGATE(S.get(&vIndex, nv));
    const dim3 B(256);
    if (!reuse)
    {
        LAUNCH(ctx, "mesher.weld.time", iotaKernel, dim3(divUp(nv, 256)), B, compRep, nv);
        LAUNCH(ctx, "mesher.weld.time", iotaKernel, dim3(divUp(nv, 256)), B, outRep, nv);
        LAUNCH(ctx, "mesher.weld.time", iotaKernel, dim3(divUp(nv, 256)), B, parent, nv);
        HIP_CHECK(hipMemsetAsync(size, 0, nv * 4, ctx->stream));
    }

    /* 1. weld */
    SortResult<uint64_t> sorted = {nullptr, nullptr};
    uint64_t *keysA = nullptr, *keysB = nullptr;
    uint32_t *slotsA = nullptr, *slotsB = nullptr;
    if (ne > 0)
    {
        uint32_t *hist, *tileSums;
        PROPAGATE(S.get(&keysA, ne));
        PROPAGATE(S.get(&keysB, ne));
        PROPAGATE(S.get(&slotsA, ne));
        PROPAGATE(S.get(&slotsB, ne));
        PROPAGATE(S.get(&hist, sortHistElems(ne)));
        PROPAGATE(S.get(&tileSums, scanTiles(std::max<uint64_t>(sortHistElems(ne), ne))));
        if (!reuse)
        {
            HIP_CHECK(hipMemcpyAsync(keysA, m->extKeys.ptr, ne * 8, hipMemcpyDeviceToDevice, ctx->stream));
            PROPAGATE(radixSort<uint64_t>(ctx, "mesher.weld.time", keysA, slotsA, keysB, slotsB, ne, 64, true, hist, tileSums, &sorted));
            m->cacheSortedInA = sorted.keys == keysA;
            LAUNCH(ctx, "mesher.weld.time", externalRepsKernel, dim3(divUp(ne, 256)), B, (const uint64_t *) sorted.keys,
                   (const uint32_t *) sorted.vals, (const uint32_t *) m->extGid.ptr, (const uint32_t *) m->extChunk.ptr, ne, compRep, outRep);
        }
        else
        {
            sorted.keys = m->cacheSortedInA ? keysA : keysB;
            sorted.vals = m->cacheSortedInA ? slotsA : slotsB;
        }
    }

    /* 2. components */
    uint32_t *dFailed;
    PROPAGATE(S.get(&dFailed, 1));
    HIP_CHECK(hipMemsetAsync(dFailed, 0, 4, ctx->stream));
    if (!reuse)
    {
        /* measured in round 3 (ms per finalize, shortcut beyond 1 / 3 / 8 steps): shells cloud (17.8 M vertices) 13.3 / 12.8 /
         * 14.1, noise cloud (378 M) 113.9 / 103.9 / 96.3 */
        static const int forced = getenv("MLSGPU_HIP_UF_SHORTCUT") ? atoi(getenv("MLSGPU_HIP_UF_SHORTCUT")) : -1;
        const uint32_t unionShortcut = forced >= 0 ? (uint32_t) forced : (nv < (uint64_t(100) << 20) ? 3u : 8u);
        /* background (mlsgpu_hip_mesher_set_background): dynamic LDS the kernel never touches leaves it four workgroups per
         * CU.  The pass waits on L2 atomics, not on wave slots -- a surface-like mesh takes as long either way -- but at full
         * occupancy it slows the kernels of the job that is streaming in behind it. */
        const uint32_t ufPad = m->background && nv < (uint64_t(100) << 20) ? 40000u : 0u;     /* (a mesh of hundreds of millions
                                                                                                * of vertices does need the slots) */
        LAUNCH_LDS(ctx, "mesher.components.time", unionKernel, dim3(divUp(nt, 256)), B, ufPad, (const uint32_t *) m->triangles.ptr, nt,
                   (const uint32_t *) compRep, parent, dFailed, unionShortcut);
        LAUNCH(ctx, "mesher.components.time", compressKernel, dim3(divUp(nv, 256)), B, parent, (const uint32_t *) compRep, nv, root);
        LAUNCH(ctx, "mesher.components.time", componentSizeKernel, dim3(divUp(divUp(nv, 64 * SIZE_SPAN), 4)), B,
               (const uint32_t *) compRep, (const uint32_t *) root, nv, size);
    }

    /* 2b. several meshers, one job: the dense roots (needed by both the export and the verdict) */
    uint32_t rootCount = 0;
    uint32_t *const rootId = parent;            /* the union-find forest is not needed any more */
    uint32_t *const denseOf = vIndex;           /* the output pass rewrites it later */
    if (analyzeOnly || keepRoots != nullptr)
    {
        uint32_t *dRootTotal, *rootTiles;
        PROPAGATE(S.get(&dRootTotal, 1));
        PROPAGATE(S.get(&rootTiles, scanTiles(nv)));
        if (reuseDense)
            rootCount = m->cacheRootCount;
        else
        {
            PROPAGATE((exclusiveScan<uint32_t>(ctx, "mesher.components.time", IsRootIn{compRep, root}, RootOut{rootId, denseOf}, nv, 0u,
                                               rootTiles, dRootTotal)));
            HIP_CHECK(hipMemcpyAsync(&rootCount, dRootTotal, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
        }
    }
    if (analyzeOnly)
    {
        uint32_t hFailed = 0;
        HIP_CHECK(hipMemcpyAsync(&hFailed, dFailed, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (hFailed != 0)
            return setError(MLSGPU_ERR_HIP, "mesher: component union did not converge");
        /* vertices and triangles per root, in buffers of their own: representatives, roots and sizes stay intact for the
         * verdict pass (finalize_with) that follows an export */
        uint32_t *tcount, *vcount;
        PROPAGATE(S.get(&tcount, std::max<uint64_t>(rootCount, 1)));
        PROPAGATE(S.get(&vcount, std::max<uint64_t>(rootCount, 1)));
        HIP_CHECK(hipMemsetAsync(tcount, 0, (size_t) rootCount * 4, ctx->stream));
        if (rootCount <= FEW_COMPONENTS)
            LAUNCH(ctx, "mesher.components.time", componentTrianglesFewKernel, dim3(divUp(divUp(nt, 64 * SIZE_SPAN), 4)), B,
                   (const uint32_t *) m->triangles.ptr, (const uint32_t *) root, (const uint32_t *) denseOf, nt, rootCount, tcount);
        else
            LAUNCH(ctx, "mesher.components.time", componentTrianglesKernel, dim3(divUp(divUp(nt, 64 * SIZE_SPAN), 4)), B,
                   (const uint32_t *) m->triangles.ptr, (const uint32_t *) root, (const uint32_t *) denseOf, nt, tcount);
        LAUNCH(ctx, "mesher.components.time", gatherSizesKernel, dim3(divUp(rootCount, 256)), B, (const uint32_t *) rootId,
               (const uint32_t *) size, rootCount, vcount);
        std::vector<uint32_t> hv(rootCount), ht(rootCount);
        HIP_CHECK(hipMemcpyAsync(hv.data(), vcount, (size_t) rootCount * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(ht.data(), tcount, (size_t) rootCount * 4, hipMemcpyDeviceToHost, ctx->stream));
        uint32_t keyCount = 0;
        if (ne > 0)
        {
            /* the distinct keys with their roots, into the sort's spare buffers */
            uint64_t *keyOut = sorted.keys == keysA ? keysB : keysA;
            uint32_t *rootOut = sorted.vals == slotsA ? slotsB : slotsA;
            uint32_t *dKeyTotal, *keyTiles;
            PROPAGATE(S.get(&dKeyTotal, 1));
            PROPAGATE(S.get(&keyTiles, scanTiles(ne)));
            PROPAGATE((exclusiveScan<uint32_t>(ctx, "mesher.weld.time", FirstOfRunIn{sorted.keys},
                                               KeyRootOut{sorted.keys, sorted.vals, m->extGid.ptr, root, denseOf, keyOut, rootOut},
                                               ne, 0u, keyTiles, dKeyTotal)));
            HIP_CHECK(hipMemcpyAsync(&keyCount, dKeyTotal, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            PROPAGATE(m->bKeys.resize(keyCount));
            PROPAGATE(m->bKeyRoot.resize(keyCount));
            HIP_CHECK(hipMemcpyAsync(m->bKeys.data(), keyOut, (size_t) keyCount * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipMemcpyAsync(m->bKeyRoot.data(), rootOut, (size_t) keyCount * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        m->bRootVertices.assign(hv.begin(), hv.end());
        m->bRootTriangles.assign(ht.begin(), ht.end());
        m->cacheKind = 1;
        m->cacheDims[0] = nv; m->cacheDims[1] = nt; m->cacheDims[2] = ne;
        m->cacheRootCount = rootCount;
        if (numChunks)
            *numChunks = 0;
        return MLSGPU_OK;
    }
    if (keepRoots != nullptr)
    {
        if (numRoots != rootCount || !m->analyzed || m->bRootVertices.size() != rootCount)
            return setError(MLSGPU_ERR_LENGTH, "mesher: %llu verdicts for %u components (call boundary first, no add in between)",
                            (unsigned long long) numRoots, rootCount);
        uint8_t *dKeep;
        PROPAGATE(S.get(&dKeep, rootCount));
        HIP_CHECK(hipMemcpyAsync(dKeep, keepRoots, rootCount, hipMemcpyHostToDevice, ctx->stream));
        LAUNCH(ctx, "mesher.components.time", applyVerdictKernel, dim3(divUp(rootCount, 256)), B, (const uint32_t *) rootId,
               (const uint8_t *) dKeep, rootCount, size);
    }

    /* 3. prune threshold, src/mesher.cpp:498-527 */
    unsigned long long *dStats;
    PROPAGATE(S.get(&dStats, 4));
    HIP_CHECK(hipMemsetAsync(dStats, 0, 32, ctx->stream));
    const dim3 statsGrid((uint32_t) std::min<uint64_t>(divUp(nv, 256), 8192));
    LAUNCH(ctx, "mesher.components.time", componentStatsKernel, statsGrid, B, (const uint32_t *) compRep,
           (const uint32_t *) root, (const uint32_t *) size, nv, (uint64_t) 0, dStats, 0);
    unsigned long long hStats[4];
    uint32_t hFailed = 0;
    HIP_CHECK(hipMemcpyAsync(hStats, dStats, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(&hFailed, dFailed, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (hFailed != 0)
        return setError(MLSGPU_ERR_HIP, "mesher: component union did not converge");
    const uint64_t totalVertices = hStats[0];
    /* with a verdict the "sizes" are all-or-nothing and the threshold is 1 */
    const uint64_t threshold = keepRoots != nullptr ? 1 : (uint64_t) ((double) totalVertices * m->pruneThreshold);
    LAUNCH(ctx, "mesher.components.time", componentStatsKernel, statsGrid, B, (const uint32_t *) compRep,
           (const uint32_t *) root, (const uint32_t *) size, nv, threshold, dStats, 1);

    /* 4. output */
    std::vector<uint64_t> tBase(nb + 1), chunkFirstV(nc + 1), chunkFirstT(nc + 1);
    std::vector<uint32_t> vBase(nb + 1), chunkOf(nb);
    for (uint32_t b = 0; b < nb; b++)
    {
        tBase[b] = m->blocks[b].tBase;
        vBase[b] = m->blocks[b].vBase;
        chunkOf[b] = (uint32_t) m->blocks[b].chunk;
    }
    tBase[nb] = nt;
    vBase[nb] = (uint32_t) nv;
    for (uint32_t c = 0, b = 0; c <= nc; c++)
    {
        while (b < nb && chunkOf[b] < c)
            b++;
        chunkFirstV[c] = b < nb ? vBase[b] : nv;
        chunkFirstT[c] = b < nb ? tBase[b] : nt;
    }
    uint64_t *dTBase, *dAt;
    uint32_t *dVBase, *dChunkOf, *dChunkVStart, *dChunkTStart, *dTotals, *tileSums;
    PROPAGATE(S.get(&dTBase, nb + 1));
    PROPAGATE(S.get(&dVBase, nb + 1));
    PROPAGATE(S.get(&dChunkOf, nb));
    PROPAGATE(S.get(&dChunkVStart, nc + 1));
    PROPAGATE(S.get(&dChunkTStart, nc + 1));
    PROPAGATE(S.get(&dAt, nc + 1));
    PROPAGATE(S.get(&dTotals, 2));
    PROPAGATE(S.get(&tileSums, scanTiles(std::max(nv, nt))));
    HIP_CHECK(hipMemcpyAsync(dTBase, tBase.data(), (nb + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(dVBase, vBase.data(), (nb + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(dChunkOf, chunkOf.data(), nb * 4, hipMemcpyHostToDevice, ctx->stream));

    /* vertices: the scan's size is not known before it ran; the output is sized for every vertex and trimmed by count */
    const KeepVertexIn keepV{outRep, root, size, threshold};
    PROPAGATE((exclusiveScan<uint32_t>(ctx, "mesher.output.time", keepV, VertexOut{m->vertices.ptr, m->outVertices, vIndex},
                                       nv, 0u, tileSums, dTotals)));
    HIP_CHECK(hipMemcpyAsync(dAt, chunkFirstV.data(), (nc + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    LAUNCH(ctx, "mesher.output.time", gatherU32Kernel, dim3(divUp(nc + 1, 256)), B, (const uint32_t *) vIndex,
           (const uint64_t *) dAt, nc + 1, nv, 0u, dChunkVStart);
    /* the entry for "one past the last vertex" is the scan total */
    std::vector<uint32_t> hV(nc + 1), hT(nc + 1);
    uint32_t totals[2] = {0, 0};
    HIP_CHECK(hipMemcpyAsync(hV.data(), dChunkVStart, (nc + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(&totals[0], dTotals, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (uint32_t c = 0; c <= nc; c++)
        if (chunkFirstV[c] >= nv)
            hV[c] = totals[0];
    HIP_CHECK(hipMemcpyAsync(dChunkVStart, hV.data(), (nc + 1) * 4, hipMemcpyHostToDevice, ctx->stream));

    /* triangles */
    uint32_t *tIndex;
    PROPAGATE(S.get(&tIndex, nt));
    const BlockTable T{dTBase, dVBase, dChunkOf, dChunkVStart, nb};
    const KeepTriangleIn keepT{m->triangles.ptr, root, size, threshold};
    PROPAGATE((exclusiveScan<uint32_t>(ctx, "mesher.output.time", keepT,
                                       TriangleOutIdx{TriangleOut{m->triangles.ptr, outRep, vIndex, T, m->outTriangles}, tIndex},
                                       nt, 0u, tileSums, dTotals + 1)));
    HIP_CHECK(hipMemcpyAsync(dAt, chunkFirstT.data(), (nc + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    LAUNCH(ctx, "mesher.output.time", gatherU32Kernel, dim3(divUp(nc + 1, 256)), B, (const uint32_t *) tIndex,
           (const uint64_t *) dAt, nc + 1, nt, 0u, dChunkTStart);
    HIP_CHECK(hipMemcpyAsync(hT.data(), dChunkTStart, (nc + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(&totals[1], dTotals + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(hStats, dStats, 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (uint32_t c = 0; c <= nc; c++)
    {
        if (chunkFirstT[c] >= nt)
            hT[c] = totals[1];
        m->chunkVStart[c] = hV[c];
        m->chunkTStart[c] = hT[c];
    }
    for (uint32_t c = 0; c < nc; c++)
        if (m->chunkTStart[c + 1] > m->chunkTStart[c])      /* no output for a chunk without triangles, :820 */
            m->outChunks.push_back(c);
    m->stats[0] = totalVertices;
    m->stats[1] = threshold;
    m->stats[2] = hStats[1];
    m->stats[3] = hStats[2];
    m->stats[4] = hStats[3];
    if (keepRoots != nullptr)
    {
        /* this mesher's own part of the job: kept components and their welded vertices from the export */
        m->stats[4] = 0;
        for (uint64_t r = 0; r < numRoots; r++)
            if (keepRoots[r])
                m->stats[4] += m->bRootVertices[r];
    }
    m->stats[5] = totals[1];
    m->stats[6] = nv;
    m->stats[7] = nt;
    if (keepRoots == nullptr)
    {
        /* the weld and the components stay in the slab for an export that may follow (boundary) */
        m->cacheKind = 2;
        m->cacheDims[0] = nv; m->cacheDims[1] = nt; m->cacheDims[2] = ne;
    }
    m->finalized = true;
    if (numChunks)
        *numChunks = (uint32_t) m->outChunks.size();
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_mesher_chunk(mlsgpu_mesher *m, uint32_t i, uint64_t *chunkId, uint64_t *numVertices,
                                       uint64_t *numTriangles, const float **dVertices, const uint32_t **dTriangles)
{
    REQUIRE(m != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(m->finalized && i < m->outChunks.size(), MLSGPU_ERR_INVALID);
    const uint32_t c = m->outChunks[i];
    if (chunkId) *chunkId = m->chunkIds[c];
    if (numVertices) *numVertices = m->chunkVStart[c + 1] - m->chunkVStart[c];
    if (numTriangles) *numTriangles = m->chunkTStart[c + 1] - m->chunkTStart[c];
    if (dVertices) *dVertices = m->outVertices + 3 * m->chunkVStart[c];
    if (dTriangles) *dTriangles = m->outTriangles + 3 * m->chunkTStart[c];
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_mesher_stats(mlsgpu_mesher *m, uint64_t out[8])
{
    REQUIRE(m != nullptr && out != nullptr && m->finalized, MLSGPU_ERR_INVALID);
    std::copy(m->stats, m->stats + 8, out);
    return MLSGPU_OK;
}

/* ---- several meshers, one job: the device sink's side of mlsgpu_amd/dist_sink.py (see include/mlsgpu_hip.h) ---- */
MLSGPU_API int mlsgpu_hip_mesher_boundary(mlsgpu_mesher *m, uint64_t *numKeys, uint64_t *numRoots)
{
    REQUIRE(m != nullptr && numKeys != nullptr && numRoots != nullptr, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    PROPAGATE(m->finalizeImpl(nullptr, true, nullptr, 0));
    *numKeys = m->bKeys.size();
    *numRoots = m->bRootVertices.size();
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_mesher_boundary_read(mlsgpu_mesher *m, uint64_t *keys, uint32_t *keyRoot, uint64_t *rootVertices,
                                               uint64_t *rootTriangles)
{
    REQUIRE(m != nullptr, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    REQUIRE(m->analyzed, MLSGPU_ERR_INVALID);
    REQUIRE((m->bKeys.empty() || (keys != nullptr && keyRoot != nullptr))
            && (m->bRootVertices.empty() || (rootVertices != nullptr && rootTriangles != nullptr)), MLSGPU_ERR_INVALID);
    std::copy(m->bKeys.begin(), m->bKeys.end(), keys);
    std::copy(m->bKeyRoot.begin(), m->bKeyRoot.end(), keyRoot);
    std::copy(m->bRootVertices.begin(), m->bRootVertices.end(), rootVertices);
    std::copy(m->bRootTriangles.begin(), m->bRootTriangles.end(), rootTriangles);
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_mesher_finalize_with(mlsgpu_mesher *m, const uint8_t *keepRoot, uint64_t numRoots, uint32_t *numChunks)
{
    REQUIRE(m != nullptr && (numRoots == 0 || keepRoot != nullptr), MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    REQUIRE(m->analyzed, MLSGPU_ERR_INVALID);
    static const uint8_t none = 0;
    return m->finalizeImpl(numChunks, false, numRoots ? keepRoot : &none, numRoots);
}

namespace
{
std::string plyHeader(uint64_t numVertices, uint64_t numTriangles, const char *const *comments, uint32_t numComments)
{
    std::string head = "ply\nformat binary_little_endian 1.0\n";
    for (uint32_t i = 0; i < numComments; i++)
        head += std::string("comment ") + comments[i] + "\n";
    head += "element vertex " + std::to_string(numVertices) + "\nproperty float32 x\nproperty float32 y\nproperty float32 z\n";
    head += "element face " + std::to_string(numTriangles) + "\nproperty list uint8 uint32 vertex_indices\ncomment padding:";
    size_t size = head.size() + 12;
    while (size % 4 != 0)
    {
        head += 'X';
        size++;
    }
    head += "\nend_header\n";
    return head;
}

/* a face of the file: the count byte and three indices, 13 bytes, packed on the device */
__global__ __launch_bounds__(256) void packFacesKernel(const uint32_t *triangles, uint64_t n, uint8_t *out)
{
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    uint8_t *o = out + 13 * i;
    o[0] = 3;
#pragma unroll
    for (int k = 0; k < 3; k++)
    {
        const uint32_t v = triangles[3 * i + k];
        o[1 + 4 * k] = (uint8_t) v;
        o[2 + 4 * k] = (uint8_t) (v >> 8);
        o[3 + 4 * k] = (uint8_t) (v >> 16);
        o[4 + 4 * k] = (uint8_t) (v >> 24);
    }
}
} // namespace

/*
 * One output chunk of a finalized mesher straight from HBM into FastPly::Writer's file, through two pinned buffers of
 * bufferBytes / 2: while one piece is written, the next travels (the role of src/async_io.h:95-140 behind the reference's
 * writer: the host never holds more of the mesh than the buffer).  Faces are packed into the file's 13-byte records on the
 * device.  The file is byte for byte what mlsgpu_hip_write_ply makes of the downloaded arrays.
 */
MLSGPU_API int mlsgpu_hip_mesher_write_ply(mlsgpu_mesher *m, uint32_t i, const char *path, const char *const *comments,
                                           uint32_t numComments, uint64_t bufferBytes)
{
    REQUIRE(m != nullptr && path != nullptr, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    REQUIRE(m->finalized && i < m->outChunks.size(), MLSGPU_ERR_INVALID);
    mlsgpu_ctx *ctx = m->ctx;
    HIP_CHECK(hipSetDevice(ctx->device));
    const uint32_t c = m->outChunks[i];
    const uint64_t nv = m->chunkVStart[c + 1] - m->chunkVStart[c], nt = m->chunkTStart[c + 1] - m->chunkTStart[c];
    const uint8_t *dV = reinterpret_cast<const uint8_t *>(m->outVertices + 3 * (uint64_t) m->chunkVStart[c]);
    const uint32_t *dT = m->outTriangles + 3 * (uint64_t) m->chunkTStart[c];
    if (bufferBytes == 0)
        bufferBytes = uint64_t(64) << 20;
    /* a piece holds whole 12-byte vertices and whole 13-byte faces */
    const uint64_t piece = std::max<uint64_t>(bufferBytes / 2 / 156, 1) * 156;
    uint8_t *pinned[2] = {nullptr, nullptr}, *dPack = nullptr;
    hipEvent_t done[2] = {nullptr, nullptr};
    FILE *f = nullptr;
    int rc = MLSGPU_OK;
    auto cleanup = [&]()
    {
        for (int k = 0; k < 2; k++)
        {
            if (pinned[k]) hipHostFree(pinned[k]);
            if (done[k]) hipEventDestroy(done[k]);
        }
        hipFree(dPack);
        if (f != nullptr)
            std::fclose(f);
    };
    for (int k = 0; k < 2 && rc == MLSGPU_OK; k++)
        if (hipHostMalloc((void **) &pinned[k], piece) != hipSuccess || hipEventCreateWithFlags(&done[k], hipEventDisableTiming) != hipSuccess)
            rc = setError(MLSGPU_ERR_NOMEM, "mesher: cannot allocate %llu bytes of pinned write buffer", (unsigned long long) piece);
    if (rc == MLSGPU_OK && hipMalloc((void **) &dPack, 2 * piece) != hipSuccess)
        rc = setError(MLSGPU_ERR_NOMEM, "mesher: cannot allocate the face packing buffer");
    if (rc == MLSGPU_OK && (f = std::fopen(path, "wb")) == nullptr)
        rc = setError(MLSGPU_ERR_INVALID, "cannot open %s for writing", path);
    if (rc != MLSGPU_OK)
    {
        cleanup();
        return rc;
    }
    const std::string head = plyHeader(nv, nt, comments, numComments);
    bool ok = std::fwrite(head.data(), 1, head.size(), f) == head.size();
    /* the pieces of the file body in order: vertex bytes, then face records; piece k + 1 is on its way while k is written */
    const uint64_t vBytes = 12 * nv, fBytes = 13 * nt, total = vBytes + fBytes;
    const uint64_t vPieces = (vBytes + piece - 1) / piece, fPieces = (fBytes + piece - 1) / piece, pieces = vPieces + fPieces;
    auto issue = [&](uint64_t k) -> hipError_t
    {
        const int slot = (int) (k & 1);
        if (k < vPieces)
        {
            const uint64_t off = k * piece, n = std::min(piece, vBytes - off);
            hipError_t e = hipMemcpyAsync(pinned[slot], dV + off, n, hipMemcpyDeviceToHost, ctx->stream);
            return e != hipSuccess ? e : hipEventRecord(done[slot], ctx->stream);
        }
        const uint64_t off = (k - vPieces) * piece, n = std::min(piece, fBytes - off);
        const uint64_t firstFace = off / 13, faces = n / 13;
        uint8_t *pack = dPack + (uint64_t) slot * piece;
        hipLaunchKernelGGL(packFacesKernel, dim3(divUp(faces, 256)), dim3(256), 0, ctx->stream, dT + 3 * firstFace, faces, pack);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess)
            e = hipMemcpyAsync(pinned[slot], pack, n, hipMemcpyDeviceToHost, ctx->stream);
        return e != hipSuccess ? e : hipEventRecord(done[slot], ctx->stream);
    };
    hipError_t e = pieces > 0 ? issue(0) : hipSuccess;
    uint64_t written = 0;
    for (uint64_t k = 0; k < pieces && ok && e == hipSuccess; k++)
    {
        if (k + 1 < pieces)
            e = issue(k + 1);
        if (e == hipSuccess)
            e = hipEventSynchronize(done[k & 1]);
        if (e != hipSuccess)
            break;
        const uint64_t n = k < vPieces ? std::min(piece, vBytes - k * piece) : std::min(piece, fBytes - (k - vPieces) * piece);
        ok = std::fwrite(pinned[k & 1], 1, n, f) == n;
        written += n;
    }
    hipStreamSynchronize(ctx->stream);
    ok = (std::fclose(f) == 0) && ok && written == total;
    f = nullptr;
    cleanup();
    if (e != hipSuccess)
        return setError(MLSGPU_ERR_HIP, "mesher: reading the mesh back failed: %s", hipGetErrorString(e));
    if (!ok)
        return setError(MLSGPU_ERR_INVALID, "writing %s failed", path);
    return MLSGPU_OK;
}

/* FastPly::Writer's file (src/fast_ply.cpp:443-521): header padded to a multiple of 4, float32 x y z per vertex,
 * uint8 3 + 3 x uint32 per face; host memory in, one file out */
MLSGPU_API int mlsgpu_hip_write_ply(const char *path, const float *vertices, uint64_t numVertices, const uint32_t *triangles,
                                    uint64_t numTriangles, const char *const *comments, uint32_t numComments)
{
    REQUIRE(path != nullptr && (numVertices == 0 || vertices != nullptr) && (numTriangles == 0 || triangles != nullptr),
            MLSGPU_ERR_INVALID);
    const std::string head = plyHeader(numVertices, numTriangles, comments, numComments);
    FILE *f = std::fopen(path, "wb");
    if (f == nullptr)
        return setError(MLSGPU_ERR_INVALID, "cannot open %s for writing", path);
    bool ok = std::fwrite(head.data(), 1, head.size(), f) == head.size();
    ok = ok && std::fwrite(vertices, 12, numVertices, f) == numVertices;
    std::vector<unsigned char> faces;
    const uint64_t batch = 1 << 20;
    for (uint64_t first = 0; ok && first < numTriangles; first += batch)
    {
        const uint64_t n = std::min(batch, numTriangles - first);
        faces.resize(13 * n);
        for (uint64_t i = 0; i < n; i++)
        {
            faces[13 * i] = 3;
            std::memcpy(&faces[13 * i + 1], triangles + 3 * (first + i), 12);
        }
        ok = std::fwrite(faces.data(), 13, n, f) == n;
    }
    ok = (std::fclose(f) == 0) && ok;
    if (!ok)
        return setError(MLSGPU_ERR_INVALID, "writing %s failed", path);
    return MLSGPU_OK;
}
