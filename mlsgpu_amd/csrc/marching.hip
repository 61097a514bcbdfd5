/*
 * Marching tetrahedra with on-device welding on gfx950 -- the device half of Marching
 * (reference: src/marching.{h,cpp}, kernels/marching.cl) plus ScaleBiasFilter
 * (src/mesh_filter.cpp:69-113, kernels/scale_bias.cl) and enqueueReadMesh (src/mesh.cpp:62-102).
 *
 * Data contracts kept from the reference: lookup tables (makeTables), vertex-key bit layout,
 * external-vertex rule, internal-first / external-last output order, DeviceKeyMesh layout, the
 * swathe loop and the mesh-memory overflow protocol of addSlices / shipOut.
 *
 * Mechanism, MI355X-first:
 *  - the distance field is a plain HBM buffer addressed like the reference's packed 2-D image
 *    (CDNA has no image path); with no 8192-row image limit a swathe may span the whole bucket;
 *  - occupied cells are compacted in cell-linear (z, y, x) order by a three-component scan
 *    (cells, vertices, indices) whose producer classifies the cell and whose consumer writes the
 *    cell and its output positions: deterministic, and no global atomics (the reference appends
 *    with atomic_inc, kernels/marching.cl:114);
 *  - welding sorts (compact key, vertex index) pairs on only the significant key bits
 *    (1 + bits(2W) + bits(2H) + bits(2D) instead of 64) and moves the 16-byte positions once, in
 *    the compaction pass, instead of through every sort pass;
 *  - count-unique, its scan, compactVertices and the index remap table are one scan launch set.
 *  - host synchronisations per bucket: one per swathe (totals) + one per ship-out (welded sizes).
 */
#include "common.hpp"
#include <type_traits>
#include "primitives.hpp"

#include <algorithm>
#include <cassert>
#include <cstdlib>

using namespace mlsgpu;

namespace
{

enum
{
    NUM_EDGES = 19,
    NUM_TETRAHEDRA = 6,
    NUM_CUBES = 256,
    MAX_CELL_VERTICES = 13,
    MAX_CELL_INDICES = 36,
    KEY_AXIS_BITS = 21
};
const uint64_t KEY_EXTERNAL_FLAG = uint64_t(1) << 63;

/* src/marching.cpp:50-81 */
const unsigned char edgeIndices[NUM_EDGES][2] =
{
    {0, 1}, {0, 2}, {0, 3}, {1, 3}, {2, 3}, {0, 4}, {0, 5}, {1, 5}, {4, 5}, {0, 6},
    {2, 6}, {4, 6}, {0, 7}, {1, 7}, {2, 7}, {3, 7}, {4, 7}, {5, 7}, {6, 7}
};
const unsigned char tetrahedronIndices[NUM_TETRAHEDRA][4] =
{
    {0, 7, 1, 3}, {0, 7, 3, 2}, {0, 7, 2, 6}, {0, 7, 6, 4}, {0, 7, 4, 5}, {0, 7, 5, 1}
};

struct HostTables
{
    uint8_t count[NUM_CUBES][2];
    uint16_t start[NUM_CUBES + 1][2];
    std::vector<uint8_t> data;      /* vertex edge ids, then index lists */
    std::vector<uint32_t> key;      /* 3 per vertex entry */
};

unsigned int findEdge(unsigned int v0, unsigned int v1)
{
    if (v0 > v1) std::swap(v0, v1);
    for (unsigned int i = 0; i < NUM_EDGES; i++)
        if (edgeIndices[i][0] == v0 && edgeIndices[i][1] == v1)
            return i;
    return ~0u;
}

struct TetVertex
{
    unsigned char id;
    bool outside;
    bool operator<(const TetVertex &o) const { return id != o.id ? id < o.id : outside < o.outside; }
    bool operator>(const TetVertex &o) const { return o < *this; }
};

unsigned int parity4(const TetVertex *v)
{
    unsigned int p = 0;
    for (int i = 0; i < 4; i++)
        for (int j = i + 1; j < 4; j++)
            if (v[i] > v[j])
                p ^= 1;
    return p;
}

/* Marching::makeTables, src/marching.cpp:109-252: per cube code, walk the six tetrahedra around
 * diagonal 0-7; canonicalise each to "outside vertices first" by the first even permutation (w.r.t.
 * the tetrahedron's own orientation) that std::next_permutation yields, and emit 0, 1 or 2 triangles. */
void makeTables(HostTables &t)
{
    std::vector<uint8_t> vertexTable, indexTable;
    for (unsigned int cube = 0; cube < NUM_CUBES; cube++)
    {
        t.start[cube][0] = (uint16_t) vertexTable.size();
        t.start[cube][1] = (uint16_t) indexTable.size();
        std::vector<uint8_t> tri;
        for (unsigned int j = 0; j < NUM_TETRAHEDRA; j++)
        {
            TetVertex tv[4];
            unsigned int outside = 0;
            for (int k = 0; k < 4; k++)
            {
                tv[k].id = tetrahedronIndices[j][k];
                tv[k].outside = (cube >> tv[k].id) & 1;
                outside += tv[k].outside;
            }
            unsigned int baseParity = parity4(tv);
            if (outside > 2)
            {
                baseParity ^= 1;
                for (int k = 0; k < 4; k++)
                    tv[k].outside = !tv[k].outside;
            }
            std::sort(tv, tv + 4);
            do
            {
                if (parity4(tv) != baseParity)
                    continue;
                unsigned int mask = 0;
                for (int k = 0; k < 4; k++)
                    mask |= (unsigned int) tv[k].outside << k;
                const unsigned int t0 = tv[0].id, t1 = tv[1].id, t2 = tv[2].id, t3 = tv[3].id;
                if (mask == 0)
                    break;
                if (mask == 1)
                {
                    tri.push_back(findEdge(t0, t1)); tri.push_back(findEdge(t0, t3)); tri.push_back(findEdge(t0, t2));
                    break;
                }
                if (mask == 3)
                {
                    tri.push_back(findEdge(t0, t2)); tri.push_back(findEdge(t1, t2)); tri.push_back(findEdge(t1, t3));
                    tri.push_back(findEdge(t1, t3)); tri.push_back(findEdge(t0, t3)); tri.push_back(findEdge(t0, t2));
                    break;
                }
            } while (std::next_permutation(tv, tv + 4));
        }
        int compact[NUM_EDGES];
        int pool = 0;
        for (unsigned int e = 0; e < NUM_EDGES; e++)
            if (std::count(tri.begin(), tri.end(), (uint8_t) e))
            {
                compact[e] = pool++;
                vertexTable.push_back((uint8_t) e);
                for (int axis = 0; axis < 3; axis++)
                    t.key.push_back(((edgeIndices[e][0] >> axis) & 1) + ((edgeIndices[e][1] >> axis) & 1));
            }
        for (size_t k = 0; k < tri.size(); k++)
            indexTable.push_back((uint8_t) compact[tri[k]]);
        t.count[cube][0] = (uint8_t) (vertexTable.size() - t.start[cube][0]);
        t.count[cube][1] = (uint8_t) (indexTable.size() - t.start[cube][1]);
    }
    t.start[NUM_CUBES][0] = (uint16_t) vertexTable.size();
    t.start[NUM_CUBES][1] = (uint16_t) indexTable.size();
    for (unsigned int i = 0; i <= NUM_CUBES; i++)
        t.start[i][1] = (uint16_t) (t.start[i][1] + vertexTable.size());
    t.data = vertexTable;
    t.data.insert(t.data.end(), indexTable.begin(), indexTable.end());
}

/* ------------------------------------------------------------------ device side */

struct DevTables
{
    const uchar2 *count;     /* [256]  (vertices, indices) */
    const ushort2 *start;    /* [257] */
    const uint8_t *data;     /* [8192] */
    const uint32_t *key;     /* [2432] kx | ky<<8 | kz<<16 */
    const uint32_t *rec;     /* [256][16] per-code record for the lattice weld: words 0-1 = 6-bit keys (kx | ky<<2 | kz<<4) of
                              * vertices 0..9, word 2 = vertices 10..12, word 3 = nv | ni << 8, words 4..12 = index bytes,
                              * word 13 = the index word that is only partly used (0 if ni % 4 == 0), word 14 = the edges
                              * that carry a vertex (bit e), word 15 = word 3 again (so that words 4..15 are all the
                              * triangle emission reads) */
};

struct FieldView
{
    const float *field;
    uint64_t pitch;
    uint32_t zStride;
    int32_t zBias;
    __device__ __forceinline__ float at(uint32_t x, uint32_t row) const { return field[(uint64_t) row * pitch + x]; }
};

/* cell range of one (sub-)swathe: cells (x, y, z) with z in [zFirst, zLast) in (z, y, x) linear order */
struct CellRange
{
    uint32_t cw, ch;         /* cells per row / rows per slice (width-1, height-1) */
    uint32_t zFirst;
    __device__ __forceinline__ void decode(uint64_t i64, uint32_t &x, uint32_t &y, uint32_t &z) const
    {
        const uint32_t i = (uint32_t) i64;      /* swathe cells < 2^32, checked at creation */
        x = i % cw;
        const uint32_t r = i / cw;
        y = r % ch;
        z = r / ch + zFirst;
    }
};

__device__ __forceinline__ void loadIso(const FieldView &F, uint32_t x, uint32_t y, uint32_t z, float iso[8])
{
    /* kernels/marching.cl:95-107 */
    const uint32_t y0 = y + F.zStride * z + (uint32_t) F.zBias;
    const uint32_t y1 = y0 + F.zStride;
    iso[0] = F.at(x, y0);     iso[1] = F.at(x + 1, y0);
    iso[2] = F.at(x, y0 + 1); iso[3] = F.at(x + 1, y0 + 1);
    iso[4] = F.at(x, y1);     iso[5] = F.at(x + 1, y1);
    iso[6] = F.at(x, y1 + 1); iso[7] = F.at(x + 1, y1 + 1);
}

/* makeCode / isValid, kernels/marching.cl:41-63 */
__device__ __forceinline__ uint32_t cellCode(const float iso[8], bool &valid)
{
    uint32_t code = 0;
    valid = true;
#pragma unroll
    for (int i = 0; i < 8; i++)
    {
        code |= (iso[i] >= 0.0f ? 1u : 0u) << i;
        valid = valid && isfinite(iso[i]);
    }
    return code;
}

/* genOccupied (kernels/marching.cl:84-120), step 1: one byte per cell of the swathe = the cell's cube code if
 * the cell is occupied (all 8 corners finite, code not 0 / 255), else 0.  Every later pass (counting,
 * compaction, per-slice histogram, lattice weld) reads this byte instead of 8 floats. */
struct CodeView
{
    const uint8_t *codes;
    uint32_t cw, ch;
    uint32_t z0;             /* cell slice stored first */
    __device__ __forceinline__ uint32_t at(uint32_t x, uint32_t y, uint32_t z) const
    {
        return codes[((uint64_t) (z - z0) * ch + y) * cw + x];
    }
};

/* Code bytes plus each row's (occupied, vertices, indices) totals.  One wave per 2 x 2 group of rows of cells -- rows (y, z),
 * (y + 1, z), (y, z + 1), (y + 1, z + 1) -- lane = cell x: the four rows need nine rows of corners between them, not
 * sixteen, and (the kernel is bound by latency times resident waves, as latticeVerticesKernel) a wave has four rows' worth
 * of independent work behind one round of loads.  The row totals feed the swathe totals, the per-slice histogram and, in the
 * lattice weld, each row's first cell / index slot -- a 256x smaller scan than one over cells. */
#ifndef CELLCODE_RUN
#define CELLCODE_RUN 64
#endif
/* (the lattice kernels -- mask, vertices, triangle rows -- were tried the same way: no change, they do not wait for those rows) */
struct CellCodeArgs
{
    uint8_t *codes;
    U3 *rowCounts;
    FieldView F;
    uint32_t cw, ch, zFirst, numRows;
    const uchar2 *countTable;
};

__global__ __launch_bounds__(256) void cellCodeKernel(Lanes<CellCodeArgs> lanes)
{
    const CellCodeArgs A = lanes.a[blockIdx.y];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t cw = A.cw, ch = A.ch, zFirst = A.zFirst;
    const uint32_t slices = ch > 0 ? A.numRows / ch : 0u;
    const uint32_t groupsY = (ch + 1) / 2, groupsZ = (slices + 1) / 2;
    /* a group shares a third of its corner rows with the group one slice pair up, groupsY groups on: runs of CELLCODE_RUN
     * workgroups (a few slice pairs) stay on one XCD, whose L2 then serves that third */
    const uint32_t wg = runOfWorkgroup<CELLCODE_RUN>(blockIdx.x, (groupsY * groupsZ + 3) / 4);
    const uint32_t group = __builtin_amdgcn_readfirstlane(wg * 4 + (threadIdx.x >> 6));
    if (group >= groupsY * groupsZ)
        return;
    uint8_t *const codes = A.codes;
    U3 *const rowCounts = A.rowCounts;
    const FieldView F = A.F;
    const uchar2 *const countTable = A.countTable;
    const uint32_t y0 = 2 * (group % groupsY), zs0 = 2 * (group / groupsY);     /* zs: slice within the swathe */
    const bool hasY = y0 + 1 < ch, hasZ = zs0 + 1 < slices;
    /* the corner rows (y0 + a, z0 + b), a, b = 0..2, clamped into the rows the group uses */
    uint32_t corner[3][3];
#pragma unroll
    for (int b = 0; b < 3; b++)
#pragma unroll
        for (int a = 0; a < 3; a++)
            corner[b][a] = (y0 + min((uint32_t) a, hasY ? 2u : 1u)) + F.zStride * (zs0 + zFirst + min((uint32_t) b, hasZ ? 2u : 1u))
                           + (uint32_t) F.zBias;
    U3 sum[2][2] = {{U3{0u, 0u, 0u}, U3{0u, 0u, 0u}}, {U3{0u, 0u, 0u}, U3{0u, 0u, 0u}}};
    /* 192 cells at a time: the corner values of the three chunks are requested before the first code byte is stored (a load
     * issued behind a store waits for that store: loads and stores retire in order) */
    constexpr int C = 3;
    for (uint32_t x0 = 0; x0 < cw; x0 += 64 * C)
    {
        /* corner x of the nine corner rows; corner x + 1 is the next lane's, and lane 63's is lane 0's of the next chunk (the
         * corner after the last chunk is one more load) */
        float v[C + 1][3][3];
#pragma unroll
        for (int c = 0; c <= C; c++)
        {
            const uint32_t xc = min(x0 + 64 * c + (c < C ? lane : 0u), cw);
#pragma unroll
            for (int b = 0; b < 3; b++)
#pragma unroll
                for (int a = 0; a < 3; a++)
                    v[c][b][a] = F.at(xc, corner[b][a]);
        }
#pragma unroll
        for (int c = 0; c < C; c++)
        {
            if (x0 + 64 * c >= cw)
                break;
            const uint32_t x = x0 + 64 * c + lane;
            float n[3][3];      /* the same corners at x + 1 */
#pragma unroll
            for (int b = 0; b < 3; b++)
#pragma unroll
                for (int a = 0; a < 3; a++)
                {
                    const uint32_t next = readLane(__float_as_uint(v[c + 1][b][a]), 0);
                    const uint32_t down = waveShiftDown1(__float_as_uint(v[c][b][a]));
                    n[b][a] = __uint_as_float(lane == 63 ? next : down);
                }
#pragma unroll
            for (int dz = 0; dz < 2; dz++)
            {
                if (dz == 1 && !hasZ)
                    break;
#pragma unroll
                for (int dy = 0; dy < 2; dy++)
                {
                    if (dy == 1 && !hasY)
                        break;
                    if (x < cw)
                    {
                        /* loadIso's order, kernels/marching.cl:95-107 */
                        const float iso[8] = {v[c][dz][dy], n[dz][dy], v[c][dz][dy + 1], n[dz][dy + 1],
                                              v[c][dz + 1][dy], n[dz + 1][dy], v[c][dz + 1][dy + 1], n[dz + 1][dy + 1]};
                        bool valid;
                        uint32_t code = cellCode(iso, valid);
                        if (!valid || code == 255)
                            code = 0;
                        codes[((uint64_t) (zs0 + dz) * ch + (y0 + dy)) * cw + x] = (uint8_t) code;
                        if (code != 0)
                        {
                            const uchar2 cnt = countTable[code];
                            sum[dz][dy].a += 1;
                            sum[dz][dy].b += cnt.x;
                            sum[dz][dy].c += cnt.y;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int dz = 0; dz < 2; dz++)
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
        {
            if ((dz == 1 && !hasZ) || (dy == 1 && !hasY))
                continue;
            U3 t = sum[dz][dy];
            t.a = waveSum(t.a);
            t.b = waveSum(t.b);
            t.c = waveSum(t.c);
            if (lane == 0)
                rowCounts[(uint64_t) (zs0 + dz) * ch + (y0 + dy)] = t;
        }
}

/* ... step 2: the producer of the compaction scan */
struct ClassifyIn
{
    CodeView C;
    CellRange R;
    const uchar2 *countTable;
    __device__ __forceinline__ U3 operator()(uint64_t i) const
    {
        uint32_t x, y, z;
        R.decode(i, x, y, z);
        const uint32_t code = C.at(x, y, z);
        if (code != 0)
        {
            const uchar2 c = countTable[code];
            return U3{1u, c.x, c.y};
        }
        return U3{0u, 0u, 0u};
    }
};

/* ... and its consumer: the compacted cell list and each cell's first vertex / index slot
 * (occupied / viCount after scanElements in the reference, src/marching.cpp:721) */
struct CompactCellsOut
{
    CellRange R;
    uint2 *cells;        /* x | y<<16, z */
    uint2 *viStart;
    uint32_t vBase, iBase;
    __device__ __forceinline__ void operator()(uint64_t i, U3 excl, U3 val) const
    {
        if (val.a)
        {
            uint32_t x, y, z;
            R.decode(i, x, y, z);
            cells[excl.a] = make_uint2(x | (y << 16), z);
            viStart[excl.a] = make_uint2(excl.b + vBase, excl.c + iBase);
        }
    }
};

/* per-slice (vertices, indices) histogram, only needed by the overflow path (src/marching.cpp:652-701):
 * one wave per slice sums that slice's row totals */
/* The swathe totals of every bucket of a batch -- the sum of its row totals -- and their way to the host in ONE launch: a
 * workgroup per bucket adds up the bucket's rows, the last one to finish (a counter that wraps by itself) writes all the
 * totals and the sequence number into the mailbox.  (Before: a reduction, a scan of its tile sums and the mailbox kernel,
 * three launches of a few microseconds each behind one another.) */
struct RowTotalsArgs
{
    const U3 *rowCounts;
    uint64_t rows;
    U3 *total;
};

__global__ __launch_bounds__(1024) void rowTotalsKernel(Lanes<RowTotalsArgs> lanes, uint32_t count, uint32_t *gate,
                                                        uint32_t *box, uint32_t seq)
{
    const RowTotalsArgs A = lanes.a[blockIdx.x];
    __shared__ U3 waveTotals[16];
    U3 sum{0u, 0u, 0u};
    for (uint64_t i = threadIdx.x; i < A.rows; i += 1024)
        sum = sum + A.rowCounts[i];
    const U3 incl = waveInclusiveScanT(sum);
    if ((threadIdx.x & 63) == 63)
        waveTotals[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (threadIdx.x != 0)
        return;
    U3 total = waveTotals[0];
    for (int w = 1; w < 16; w++)
        total = total + waveTotals[w];
    storeAgent(A.total, total);
    __threadfence();
    if (atomicInc(gate, count - 1) != count - 1)
        return;
    __threadfence();
    for (uint32_t k = 0; k < count; k++)
    {
        const U3 t = loadAgent(lanes.a[k].total);
        __hip_atomic_store(box + 1 + 3 * k, t.a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(box + 2 + 3 * k, t.b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(box + 3 + 3 * k, t.c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __hip_atomic_store(box, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(64) void sliceHistogramKernel(const U3 *rowCounts, uint32_t ch, uint32_t zFirst, uint32_t z0,
                                                           uint2 *histogram)
{
    const uint32_t z = zFirst + blockIdx.x;
    uint32_t v = 0, idx = 0;
    for (uint32_t y = threadIdx.x; y < ch; y += 64)
    {
        const U3 c = rowCounts[(uint64_t) (z - z0) * ch + y];
        v += c.b;
        idx += c.c;
    }
    v = waveSum(v);
    idx = waveSum(idx);
    if (threadIdx.x == 0)
        histogram[z] = make_uint2(v, idx);
}

/* Lattice weld: the occupied cells of the batch in cell-linear order with each cell's first index slot.
 * One wave per row of cells; a row's first cell / index slot come from the exclusive scan of the row totals,
 * positions inside the row from a ballot rank and a wave scan. */
struct CompactRowCellsArgs
{
    const uint8_t *codes;
    uint32_t cw, ch, z0, zFirst;
    const U3 *rowStarts;
    const uchar2 *countTable;
    uint2 *cells, *viStart;
    uint32_t numRows;
};

__global__ __launch_bounds__(256) void compactRowCellsKernel(Lanes<CompactRowCellsArgs> lanes)
{
    const CompactRowCellsArgs A = lanes.a[blockIdx.y];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= A.numRows)
        return;
    const uint8_t *const codes = A.codes;
    const uint32_t cw = A.cw, ch = A.ch, z0 = A.z0, zFirst = A.zFirst;
    const U3 *const rowStarts = A.rowStarts;
    const uchar2 *const countTable = A.countTable;
    uint2 *const cells = A.cells, *const viStart = A.viStart;
    const uint32_t y = r % ch, z = r / ch + zFirst;
    const uint8_t *row = codes + ((uint64_t) (z - z0) * ch + y) * cw;
    const U3 start = rowStarts[r];
    uint32_t cellBase = start.a, indexBase = start.c;
    for (uint32_t x0 = 0; x0 < cw; x0 += 64)
    {
        const uint32_t x = x0 + lane;
        const uint32_t code = x < cw ? row[x] : 0u;
        const uint32_t ni = code != 0 ? countTable[code].y : 0u;
        const uint64_t occ = __ballot(code != 0);
        const uint32_t incl = waveInclusiveScan(ni);
        if (code != 0)
        {
            const uint32_t pos = cellBase + popcBelow(occ);
            cells[pos] = make_uint2(x | (y << 16), z | (code << 16));     /* the code rides along: one dependent load less */
            viStart[pos] = make_uint2(0u, indexBase + incl - ni);
        }
        cellBase += (uint32_t) __popcll(occ);
        indexBase += readLane(incl, 63);
    }
}

/* ScaleBiasFilter (kernels/scale_bias.cl:33-41) folded into vertex emission: v = fma(v, scale, bias).
 * Same operation on the same rounded value as the separate in-place pass, one HBM round trip less. */
struct VertexTransform
{
    float scale, bx, by, bz;
    int enabled;
};

/* Layout of the compact sort key: [ext][z2 : bz][y2 : by][x2 : bx], order-isomorphic to the
 * reference's 64-bit key (kernels/marching.cl:148-154) for coordinates that fit the field widths. */
struct KeyLayout
{
    uint32_t bx, by, bz;
    __host__ __device__ __forceinline__ uint32_t bits() const { return bx + by + bz + 1; }
};

/* interp, kernels/marching.cl:130-138 (explicit fma, no other contraction) */
__device__ __forceinline__ void interp(float iso0, float iso1, uint32_t gx, uint32_t gy, uint32_t gz,
                                       uint32_t c0, uint32_t c1, float &ox, float &oy, float &oz)
{
    const float inv = 1.0f / (iso0 - iso1);
    const float t = iso0 * inv;
    const uint32_t o0x = c0 & 1, o0y = (c0 >> 1) & 1, o0z = (c0 >> 2) & 1;
    const uint32_t o1x = c1 & 1, o1y = (c1 >> 1) & 1, o1z = (c1 >> 2) & 1;
    ox = fmaf(t, (float) (o1x - o0x), (float) (gx + o0x));
    oy = fmaf(t, (float) (o1y - o0y), (float) (gy + o0y));
    oz = fmaf(t, (float) (o1z - o0z), (float) (gz + o0z));
}

__constant__ unsigned char dEdgeIndices[NUM_EDGES][2] =
{
    {0, 1}, {0, 2}, {0, 3}, {1, 3}, {2, 3}, {0, 4}, {0, 5}, {1, 5}, {4, 5}, {0, 6},
    {2, 6}, {4, 6}, {0, 7}, {1, 7}, {2, 7}, {3, 7}, {4, 7}, {5, 7}, {6, 7}
};

/* generateElements, kernels/marching.cl:184-258.  One thread per occupied cell.  Only the vertices
 * the cell's code uses are interpolated (the reference computes all 19 into local memory first);
 * a vertex's value does not depend on which others are computed. */
template<typename K>
__global__ __launch_bounds__(256) void generateElementsKernel(float4 *vertices, K *sortKeys, uint32_t *indices,
                                                              const uint2 *viStart, const uint2 *cells,
                                                              FieldView F, DevTables T,
                                                              uint32_t gox, uint32_t goy, uint32_t goz,
                                                              uint32_t topx, uint32_t topy, uint32_t topz,
                                                              KeyLayout L, uint32_t numCells)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= numCells)
        return;
    const uint2 cell = cells[gid];
    const uint32_t x = cell.x & 0xFFFFu, y = cell.x >> 16, z = cell.y;
    float iso[8];
    loadIso(F, x, y, z, iso);
    bool valid;
    const uint32_t code = cellCode(iso, valid);
    const uint2 vi = viStart[gid];
    const ushort2 st = T.start[code], en = T.start[code + 1];
    const uint32_t nv = en.x - st.x, ni = en.y - st.y;
    const uint32_t gx = x + gox, gy = y + goy, gz = z + goz;
    for (uint32_t i = 0; i < nv; i++)
    {
        const uint32_t e = T.data[st.x + i];
        const uint32_t c0 = dEdgeIndices[e][0], c1 = dEdgeIndices[e][1];
        float vx, vy, vz;
        interp(iso[c0], iso[c1], gx, gy, gz, c0, c1, vx, vy, vz);
        vertices[vi.x + i] = make_float4(vx, vy, vz, __uint_as_float(vi.x + i));
        /* computeKey(2 * cell + keyTable[..], top), kernels/marching.cl:148-154,252 */
        const uint32_t k = T.key[st.x + i];
        const uint32_t kx = 2 * x + (k & 0xFF), ky = 2 * y + ((k >> 8) & 0xFF), kz = 2 * z + (k >> 16);
        const bool ext = kx == 0 || ky == 0 || kx == topx || ky == topy || kz == topz;
        sortKeys[vi.x + i] = ((K) (ext ? 1u : 0u) << (L.bx + L.by + L.bz)) | ((K) kz << (L.bx + L.by)) | ((K) ky << L.bx) | (K) kx;
    }
    for (uint32_t i = 0; i < ni; i++)
        indices[vi.y + i] = vi.x + T.data[st.y + i];
}

/* countUniqueVertices (kernels/marching.cl:271-279): 1 for the LAST of each run of equal keys.
 * The reference appends a ULONG_MAX sentinel; here the last element is simply always a run end. */
template<typename K>
struct UniqueIn
{
    const K *keys;
    uint64_t n;
    __device__ __forceinline__ uint32_t operator()(uint64_t i) const
    {
        return (i + 1 >= n || keys[i] != keys[i + 1]) ? 1u : 0u;
    }
};

/* compactVertices (kernels/marching.cl:295-326) as the consumer of that scan.  Positions are
 * gathered here through the sorted vertex index instead of having travelled through the sort. */
template<typename K>
struct CompactVerticesOut
{
    const K *keys;
    const uint32_t *order;       /* sorted original vertex indices */
    const float4 *inVertices;
    float *outVertices;
    uint64_t *outKeys;
    uint32_t *indexRemap;
    uint32_t *firstExternal;
    KeyLayout L;
    uint32_t zMax2;              /* 2 * zMax: keys with z >= this are external (src/marching.cpp:593) */
    uint64_t keyOffset;
    uint64_t n;
    VertexTransform X;

    __device__ __forceinline__ bool isExternal(K key) const
    {
        const uint32_t z2 = (uint32_t) ((key >> (L.bx + L.by)) & (((K) 1 << L.bz) - 1));
        return ((key >> (L.bx + L.by + L.bz)) & 1) != 0 || z2 >= zMax2;
    }

    __device__ __forceinline__ void operator()(uint64_t i, uint32_t u, uint32_t isLast) const
    {
        const K key = keys[i];
        const uint32_t orig = order[i];
        if (isLast)
        {
            float4 v = inVertices[orig];
            if (X.enabled)
            {
                v.x = fmaf(v.x, X.scale, X.bx);
                v.y = fmaf(v.y, X.scale, X.by);
                v.z = fmaf(v.z, X.scale, X.bz);
            }
            outVertices[3 * (uint64_t) u + 0] = v.x;
            outVertices[3 * (uint64_t) u + 1] = v.y;
            outVertices[3 * (uint64_t) u + 2] = v.z;
            const bool ext = isExternal(key);
            const bool nextExt = (i + 1 >= n) || isExternal(keys[i + 1]);   /* sentinel counts as external */
            if (ext)
            {
                const uint64_t x2 = (uint64_t) (key & (((K) 1 << L.bx) - 1));
                const uint64_t y2 = (uint64_t) ((key >> L.bx) & (((K) 1 << L.by) - 1));
                const uint64_t z2 = (uint64_t) ((key >> (L.bx + L.by)) & (((K) 1 << L.bz) - 1));
                outKeys[u] = ((z2 << (2 * KEY_AXIS_BITS)) | (y2 << KEY_AXIS_BITS) | x2) + keyOffset;
                if (u == 0)
                    *firstExternal = 0;
            }
            else if (nextExt)
                *firstExternal = u + 1;
        }
        indexRemap[orig] = u;
    }
};

/* reindex, kernels/marching.cl:334-340 */
__global__ __launch_bounds__(256) void reindexKernel(uint32_t *indices, const uint32_t *indexRemap, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        indices[i] = indexRemap[indices[i]];
}

/* copySlice, kernels/marching.cl:349-358 / clEnqueueCopyImage in src/marching.cpp:447-498 */
__global__ __launch_bounds__(256) void copySliceKernel(float *field, uint64_t pitch, uint32_t srcRow, uint32_t trgRow,
                                                       uint32_t width, uint32_t height)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < width * height)
    {
        const uint32_t x = i % width, y = i / width;
        field[(uint64_t) (trgRow + y) * pitch + x] = field[(uint64_t) (srcRow + y) * pitch + x];
    }
}

/* scaleBiasVertices, kernels/scale_bias.cl:33-41 */
__global__ __launch_bounds__(256) void scaleBiasKernel(float *vertices, uint64_t numFloats, float scale,
                                                       float bx, float by, float bz)
{
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < numFloats)
    {
        const uint32_t axis = (uint32_t) (i % 3);
        const float b = axis == 0 ? bx : (axis == 1 ? by : bz);
        vertices[i] = fmaf(vertices[i], scale, b);
    }
}

/* the reference's compactVertices kernel verbatim in behaviour, for its known-answer test */
__global__ void compactVerticesRefKernel(float *outVertices, uint64_t *outKeys, uint32_t *indexRemap,
                                         uint32_t *firstExternal, const uint32_t *vertexUnique,
                                         const float4 *inVertices, const uint64_t *inKeys,
                                         uint64_t minExternalKey, uint64_t keyOffset, uint64_t n)
{
    const uint64_t gid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n)
        return;
    const uint32_t u = vertexUnique[gid];
    const float4 v = inVertices[gid];
    const uint64_t key = inKeys[gid];
    const uint64_t nextKey = inKeys[gid + 1];
    const bool ext = key >= minExternalKey;
    if (key != nextKey)
    {
        outVertices[3 * (uint64_t) u + 0] = v.x;
        outVertices[3 * (uint64_t) u + 1] = v.y;
        outVertices[3 * (uint64_t) u + 2] = v.z;
        if (ext)
        {
            outKeys[u] = (key & (KEY_EXTERNAL_FLAG - 1)) + keyOffset;
            if (u == 0)
                *firstExternal = 0;
        }
        else if (nextKey >= minExternalKey)
            *firstExternal = u + 1;
    }
    indexRemap[__float_as_uint(v.w)] = u;
}

__global__ void computeKeyTestKernel(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t tx, uint32_t ty, uint32_t tz,
                                     uint64_t *out)
{
    /* computeKey, kernels/marching.cl:148-154 */
    uint64_t key = ((uint64_t) cz << (2 * KEY_AXIS_BITS)) | ((uint64_t) cy << KEY_AXIS_BITS) | (uint64_t) cx;
    if (cx == 0 || cy == 0 || cx == tx || cy == ty || cz == tz)
        key |= KEY_EXTERNAL_FLAG;
    *out = key;
}

/* ------------------------------------------------------------------ lattice weld
 *
 * Sort-free replacement for generateElements + sortVertices + countUnique + compactVertices + reindex,
 * used when the whole batch lies in the resident field (one swathe per bucket, the MI355X default).
 *
 * Every welded vertex sits on a grid edge whose endpoints have different signs, i.e. on one point of
 * the HALF LATTICE (x2, y2, z2) = 2*cell + keyTable offset -- exactly the fixed-point coordinates the
 * reference packs into its vertex key (kernels/marching.cl:148-154,252).  Sorting by key therefore
 * equals enumerating half-lattice points in (z2, y2, x2) order, and a welded vertex's index is its
 * rank in that order within its class:
 *     class 0  internal                        (first in the output)
 *     class 1  unflagged, z2 == 2*zMax         (external by "key >= minExternalKey", src/marching.cpp:593)
 *     class 2  flagged (x2 == 0, y2 == 0, x2 == top.x, y2 == top.y or z2 == top.z; marching.cl:151-152)
 * which is the order the reference's 64-bit sort produces (flag bit above z above y above x).
 * A vertex exists on an edge iff some OCCUPIED cell of the batch contains the edge and sees different
 * signs at its ends (every sign-changing edge of a cell is used by one of its tetrahedra, so this is
 * exactly the set of vertices the per-cell tables emit).
 *
 * Rows are (z2, y2); each row is a bit mask over x2 plus per-word prefix counts, so an index is
 * rowStart[class] + wordPrefix + popcount -- three small L2-resident reads instead of a global sort.
 */
/* One 64-point word of a row, 16 bytes, everything a lookup needs in ONE load:
 *   mask    existence bits of the row's MAIN class (all points of a class-2 row; a class-0/1 row without its two
 *           class-2 columns x2 == 0 and x2 == top.x)
 *   prefix  output index of the first main-class vertex of this word (row-relative until latticePatchKernel
 *           adds the row's start)
 *   flag    class-0/1 rows: output index of the row's first class-2 vertex, bit 31 = a vertex exists at x2 == 0,
 *           bit 30 = one exists at x2 == top.x (latticeMaskKernel sets each bit in the word that holds the column;
 *           latticePatchKernel copies both into every word of the row and adds the index) */
struct LatWord
{
    uint64_t mask;
    uint32_t prefix;
    uint32_t flag;
};
#define LAT_FLAG_X0 0x80000000u
#define LAT_FLAG_TOP 0x40000000u
#define LAT_FLAG_INDEX 0x3FFFFFFFu

struct Lattice
{
    LatWord *words;          /* [rows][nw] */
    U3 *rowCounts;           /* [rows] vertices per class; exclusive-scanned in place into per-class row starts */
    const U3 *totals;        /* device: class totals after the scan */
    uint32_t nw;             /* 64-bit words per row */
    uint32_t rowsPerLayer;   /* 2H - 1 */
    uint32_t topx, topy;     /* 2(W-1), 2(H-1) */
    uint32_t z2First;        /* 2 * zTop: first layer, also top.z */
    uint32_t z2Last;         /* 2 * zMax: last layer */
    uint32_t cw, ch;         /* cells per row / rows of cells */

    __device__ __forceinline__ uint32_t rowClass(uint32_t y2, uint32_t z2) const
    {
        if (y2 == 0 || y2 == topy || z2 == z2First)
            return 2;
        return z2 == z2Last ? 1 : 0;
    }
    /* bits of word w that belong to columns x2 == 0 and x2 == top.x (class 2 inside class 0/1 rows) */
    __device__ __forceinline__ uint64_t columnMask(uint32_t w) const
    {
        uint64_t m = w == 0 ? 1ull : 0ull;
        if (w == (topx >> 6))
            m |= 1ull << (topx & 63);
        return m;
    }
};

/* does a cell with this code (0 = not occupied) see different signs at its local corners a and b? */
__device__ __forceinline__ uint32_t edgeBit(uint32_t code, uint32_t a, uint32_t b)
{
    return ((code >> a) ^ (code >> b)) & 1u;
}

/*
 * Existence masks.  One wave per row of CORNERS (fixed y, z), lane = corner x.  A corner owns the seven
 * edges towards +x, +y, +z, +xy, +xz, +yz, +xyz, i.e. the half-lattice points of four rows:
 *   (z2, y2) = (2z, 2y): +x at x2 = 2x+1          (2z, 2y+1): +y at 2x, +xy at 2x+1
 *              (2z+1, 2y): +z at 2x, +xz at 2x+1  (2z+1, 2y+1): +yz at 2x, +xyz at 2x+1
 * An edge carries a vertex iff one of the occupied cells containing it sees a sign change along it; those
 * cells are among the eight around the corner, whose code bytes are four loads per lane (cell x of the four
 * adjacent cell rows) plus a lane shift for cell x-1.  The seven bits of 64 corners become the rows' words
 * (even/odd x2 interleaved) through two lane permutations and one ballot per word.
 */
struct LatticeMaskArgs
{
    Lattice L;
    CodeView C;
    const U3 *cellRowCounts;
    uint32_t zCellFirst, zCellLast, H, numCornerRows;
};

__global__ __launch_bounds__(256) void latticeMaskKernel(Lanes<LatticeMaskArgs> lanes, const uint64_t *edgeLut)
{
    __shared__ uint64_t sLut[256];
    sLut[threadIdx.x] = edgeLut[threadIdx.x];
    __syncthreads();
    const LatticeMaskArgs A = lanes.a[blockIdx.y];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t cr = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cr >= A.numCornerRows)
        return;
    const Lattice L = A.L;            /* a copy: the fields live in scalar registers, not behind kernel-argument loads */
    const CodeView C = A.C;
    const U3 *const cellRowCounts = A.cellRowCounts;
    const uint32_t zCellFirst = A.zCellFirst, zCellLast = A.zCellLast, H = A.H;
    const uint32_t y = cr % H, z = cr / H + zCellFirst;
    const uint32_t W = L.cw + 1;
    /* adjacent cell rows (y - dy, z - dz); a row outside the batch contributes no occupied cell */
    bool rowOk[2][2];
    const uint8_t *rowPtr[2][2];
#pragma unroll
    for (int dy = 0; dy < 2; dy++)
#pragma unroll
        for (int dz = 0; dz < 2; dz++)
        {
            const int cy = (int) y - dy, cz = (int) z - dz;
            rowOk[dy][dz] = cy >= 0 && cy < (int) L.ch && cz >= (int) zCellFirst && cz < (int) zCellLast;
            rowPtr[dy][dz] = C.codes + ((uint64_t) ((uint32_t) cz - C.z0) * C.ch + (uint32_t) cy) * C.cw;
        }
    /* the four half-lattice rows of this corner row: h = py | pz << 1 */
    uint32_t rowId[4], rowCls[4], running[4], nFlag[4];
    bool rowExists[4];
#pragma unroll
    for (int h = 0; h < 4; h++)
    {
        const uint32_t y2 = 2 * y + (h & 1), z2 = 2 * z + (h >> 1);
        rowExists[h] = y2 <= L.topy && z2 <= L.z2Last;
        rowId[h] = (z2 - L.z2First) * L.rowsPerLayer + y2;
        rowCls[h] = L.rowClass(y2, z2);
        running[h] = nFlag[h] = 0;
    }
    /* (the first code bytes are requested before the rows' totals are looked at: one latency, not two) */
    uint32_t next[2][2];
#pragma unroll
    for (int dy = 0; dy < 2; dy++)
#pragma unroll
        for (int dz = 0; dz < 2; dz++)
            next[dy][dz] = (rowOk[dy][dz] && lane < L.cw) ? rowPtr[dy][dz][lane] : 0u;
    /* Surface-like data leaves most of a bucket empty: if none of the adjacent cell rows holds an occupied cell (their
     * totals come from cellCodeKernel) the four rows are all zeros, written by 16-byte stores of the first lanes. */
    uint32_t occupied = 0;
#pragma unroll
    for (int dy = 0; dy < 2; dy++)
#pragma unroll
        for (int dz = 0; dz < 2; dz++)
            if (rowOk[dy][dz])
                occupied += cellRowCounts[(uint64_t) ((uint32_t) ((int) z - dz) - C.z0) * C.ch + (uint32_t) ((int) y - dy)].a;
    if (occupied == 0)
    {
#pragma unroll
        for (int h = 0; h < 4; h++)
        {
            if (!rowExists[h])
                continue;
            for (uint32_t w = lane; w < L.nw; w += 64)
                L.words[(uint64_t) rowId[h] * L.nw + w] = LatWord{0ull, 0u, 0u};
            if (lane == 0)
                L.rowCounts[rowId[h]] = U3{0u, 0u, 0u};
        }
        return;
    }
    uint32_t prevLeft = 0;
    for (uint32_t x0 = 0; x0 < W; x0 += 64)
    {
        /* What the cell at (x - ox, y - dy, z - dz) gives the corner's seven points depends on its code byte alone: a table
         * (makeEdgeLut) holds, per code, the bits of each of the four (dy, dz) as the corner's own cell (ox = 0, bytes 0-3)
         * and as the cell to its left (ox = 1, bytes 4-7).  The left cell of lane i is lane i - 1's own, so its bits come
         * with one lane shift; 64 corners further the carry is lane 63's. */
        uint32_t own = 0, left = 0;
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
#pragma unroll
            for (int dz = 0; dz < 2; dz++)
            {
                const uint32_t c = next[dy][dz];
                const uint32_t xn = x0 + 64 + lane;
                next[dy][dz] = (rowOk[dy][dz] && xn < L.cw) ? rowPtr[dy][dz][xn] : 0u;
                const uint64_t v = sLut[c];
                const int r = dy | (dz << 1);
                own |= ((uint32_t) v >> (8 * r)) & 0xFFu;
                left |= ((uint32_t) (v >> 32) >> (8 * r)) & 0xFFu;
            }
        const uint32_t up = waveShiftUp1(left);
        /* bit 0 +x, 1 +xy, 2 +xz, 3 +xyz (odd x2 of rows h = 0..3), 5 +y, 6 +z, 7 +yz (even x2 of rows 1..3) */
        const uint32_t pk = own | (lane == 0 ? prevLeft : up);
        prevLeft = readLane(left, 63);
        /* Interleave (even x2 = 2x, odd x2 = 2x+1) by lane permutation instead of bit spreading: lane i of word
         * `half` takes corner 32*half + i/2 and its even (i even) or odd (i odd) point, so one ballot per row and
         * half IS the 64-bit word. */
        const uint32_t from[2] = {(uint32_t) __shfl(pk, lane >> 1, 64), (uint32_t) __shfl(pk, 32 + (lane >> 1), 64)};
        const uint32_t sel = (lane & 1) ? 0u : 4u;
#pragma unroll
        for (int h = 0; h < 4; h++)
        {
            if (!rowExists[h])
                continue;
#pragma unroll
            for (int half = 0; half < 2; half++)
            {
                const uint32_t w = (x0 >> 6) * 2 + half;
                if (w >= L.nw)
                    continue;
                const uint64_t bits = __ballot((from[half] >> (sel + h)) & 1u);
                const uint64_t cm = rowCls[h] == 2 ? 0ull : L.columnMask(w);
                const uint64_t col = bits & cm;
                const uint64_t topBit = w == (L.topx >> 6) ? 1ull << (L.topx & 63) : 0ull;
                const uint32_t colBits = ((w == 0 && (col & 1ull)) ? LAT_FLAG_X0 : 0u) | ((col & topBit) ? LAT_FLAG_TOP : 0u);
                if (lane == 0)
                    L.words[(uint64_t) rowId[h] * L.nw + w] = LatWord{bits & ~cm, running[h], colBits};
                running[h] += (uint32_t) __popcll(bits & ~cm);
                nFlag[h] += (uint32_t) __popcll(col);
            }
        }
    }
    if (lane == 0)
    {
#pragma unroll
        for (int h = 0; h < 4; h++)
            if (rowExists[h])
            {
                U3 cnt{0u, 0u, 0u};
                if (rowCls[h] == 0) { cnt.a = running[h]; cnt.c = nFlag[h]; }
                else if (rowCls[h] == 1) { cnt.b = running[h]; cnt.c = nFlag[h]; }
                else cnt.c = running[h];
                L.rowCounts[rowId[h]] = cnt;
            }
    }
}

/*
 * The same masks, one THREAD per 64-bit word (32 corners) instead of one wave per row of corners.  The wave-per-row kernel
 * above spends ~730 vector and ~460 scalar instructions per row of 171 corners -- seven ballots per 64 corners, each followed
 * by scalar mask arithmetic and a 16-byte store from one lane -- and is bound by issuing them (VALU 61 %, one scalar unit per
 * CU 58 % busy; profiles/r05b_sq_stage_kernels.csv): 0.95 ms per step for 0.35 GB.  Here a lane walks its word's 32 corners
 * with bit operations of its own: per corner four table lookups (what the cell (x, y - dy, z - dz) gives the corner as its
 * own cell, low byte, and as the cell to its left, high byte; bits 2h + 1 / 2h = the odd / even point of lattice row h),
 * three ORs and four two-bit inserts -- ~22 vector instructions per corner and LANE, i.e. a third of an instruction per
 * corner and wave.  The lanes of a row of corners (nw consecutive ones; 64 / nw rows per wave) exchange their popcounts for
 * the words' prefixes, and every lattice row leaves as one run of 16-byte stores.  Words, counts and flags are the ones
 * the wave-per-row kernel writes, bit for bit (MLSGPU_HIP_LATTICE_MASK_ROWS=1 selects it for A/B).
 */
__global__ __launch_bounds__(256) void latticeMaskWordKernel(Lanes<LatticeMaskArgs> lanes, const uint16_t *edgeLut16)
{
    __shared__ uint16_t sLut[4][256];
    for (uint32_t i = threadIdx.x; i < 1024; i += 256)
        (&sLut[0][0])[i] = edgeLut16[i];
    __syncthreads();
    const LatticeMaskArgs A = lanes.a[blockIdx.y];
    const Lattice L = A.L;
    const CodeView C = A.C;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t nw = L.nw, rowsPerWave = 64u / nw;
    const uint32_t sub = lane / nw, w = lane - sub * nw;
    const uint32_t cr = (blockIdx.x * 4 + (threadIdx.x >> 6)) * rowsPerWave + sub;
    const bool active = sub < rowsPerWave && cr < A.numCornerRows;
    const uint32_t H = A.H, zCellFirst = A.zCellFirst, zCellLast = A.zCellLast;
    const uint32_t crc = active ? cr : 0u;
    const uint32_t y = crc % H, z = crc / H + zCellFirst;
    const uint32_t cw = L.cw;
    const uint32_t x0 = 32u * w;                 /* the word's first corner */

    /* the word's 4 x 32 code bytes (cell x of the four adjacent rows of cells), and the byte before them; the bytes behind a
     * row's end are the next row's (or the buffer's pad) and are masked away */
    uint32_t keep[8];
    {
        const uint32_t have = x0 < cw ? min(cw - x0, 32u) : 0u;
#pragma unroll
        for (int j = 0; j < 8; j++)
        {
            const uint32_t n = have > 4u * j ? min(have - 4u * j, 4u) : 0u;      /* bytes of dword j that exist */
            keep[j] = n >= 4u ? 0xFFFFFFFFu : (1u << (8u * n)) - 1u;
        }
    }
    uint32_t cd[4][8], before[4];
    uint32_t occupied = 0;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        const int dy = r & 1, dz = r >> 1;
        const int cy = (int) y - dy, cz = (int) z - dz;
        const bool ok = active && cy >= 0 && cy < (int) L.ch && cz >= (int) zCellFirst && cz < (int) zCellLast;
        const uint64_t rowIndex = ok ? (uint64_t) ((uint32_t) cz - C.z0) * C.ch + (uint32_t) cy : 0ull;
        const uint8_t *const rowBase = C.codes + rowIndex * C.cw;
        const uint8_t *p = rowBase + (x0 < cw ? x0 : 0u);
        if (ok)
            occupied += A.cellRowCounts[rowIndex].a;
        uint4 lo = make_uint4(0, 0, 0, 0), hi = make_uint4(0, 0, 0, 0);
        if (ok && x0 < cw)
        {
            __builtin_memcpy(&lo, p, 16);
            __builtin_memcpy(&hi, p + 16, 16);
        }
        cd[r][0] = lo.x & keep[0]; cd[r][1] = lo.y & keep[1]; cd[r][2] = lo.z & keep[2]; cd[r][3] = lo.w & keep[3];
        cd[r][4] = hi.x & keep[4]; cd[r][5] = hi.y & keep[5]; cd[r][6] = hi.z & keep[6]; cd[r][7] = hi.w & keep[7];
        before[r] = (ok && x0 > 0 && x0 - 1 < cw) ? (uint32_t) rowBase[x0 - 1] : 0u;     /* (a word may begin AT the row's end) */
    }

    /* the four lattice rows of the corner row: h = py | pz << 1 */
    uint64_t bits[4] = {0, 0, 0, 0};
    if (occupied != 0)
    {
        uint32_t lo32[4] = {0, 0, 0, 0}, hi32[4] = {0, 0, 0, 0};
        /* what the cell left of the word's first corner gives it */
        uint32_t leftPrev = ((uint32_t) sLut[0][before[0]] | sLut[1][before[1]] | sLut[2][before[2]] | sLut[3][before[3]]) >> 8;
#pragma unroll
        for (int i = 0; i < 32; i++)
        {
            const uint32_t e = (uint32_t) sLut[0][(cd[0][i >> 2] >> (8 * (i & 3))) & 0xFFu]
                | sLut[1][(cd[1][i >> 2] >> (8 * (i & 3))) & 0xFFu]
                | sLut[2][(cd[2][i >> 2] >> (8 * (i & 3))) & 0xFFu]
                | sLut[3][(cd[3][i >> 2] >> (8 * (i & 3))) & 0xFFu];
            const uint32_t pk = (e & 0xFFu) | leftPrev;
            leftPrev = e >> 8;
#pragma unroll
            for (int h = 0; h < 4; h++)
            {
                const uint32_t pair = (pk >> (2 * h)) & 3u;
                if (i < 16)
                    lo32[h] |= pair << (2 * i);
                else
                    hi32[h] |= pair << (2 * (i - 16));
            }
        }
#pragma unroll
        for (int h = 0; h < 4; h++)
            bits[h] = (uint64_t) lo32[h] | (uint64_t) hi32[h] << 32;
    }

    /* per lattice row: the word, its bits' count and the column bits; prefixes over the row's words */
    uint32_t rowId[4], rowCls[4], cnt[4], colCnt[4], colBits[4];
    bool rowExists[4];
    uint64_t mainBits[4];
#pragma unroll
    for (int h = 0; h < 4; h++)
    {
        const uint32_t y2 = 2 * y + (h & 1), z2 = 2 * z + (h >> 1);
        rowExists[h] = active && y2 <= L.topy && z2 <= L.z2Last;
        rowId[h] = (z2 - L.z2First) * L.rowsPerLayer + y2;
        rowCls[h] = L.rowClass(y2, z2);
        /* points beyond the row's end cannot exist (their cells do not), so the word needs no trimming */
        const uint64_t cm = rowCls[h] == 2 ? 0ull : L.columnMask(w);
        const uint64_t col = bits[h] & cm;
        const uint64_t topBit = w == (L.topx >> 6) ? 1ull << (L.topx & 63) : 0ull;
        colBits[h] = ((w == 0 && (col & 1ull)) ? LAT_FLAG_X0 : 0u) | ((col & topBit) ? LAT_FLAG_TOP : 0u);
        mainBits[h] = bits[h] & ~cm;
        cnt[h] = (uint32_t) __popcll(mainBits[h]);
        colCnt[h] = (uint32_t) __popcll(col);
    }
    /* inclusive sums over the lanes of the row of corners (w' <= w): counts up to 64 * nw <= 512 in 16-bit fields */
    uint32_t pa = cnt[0] | cnt[1] << 16, pb = cnt[2] | cnt[3] << 16, pc = colCnt[0] | colCnt[1] << 8 | colCnt[2] << 16 | colCnt[3] << 24;
    uint32_t ia = pa, ib = pb, ic = pc;
    for (uint32_t d = 1; d < nw; d++)
    {
        const uint32_t ta = (uint32_t) __shfl_up((int) pa, d, 64), tb = (uint32_t) __shfl_up((int) pb, d, 64),
                       tc = (uint32_t) __shfl_up((int) pc, d, 64);
        if (w >= d)
        {
            ia += ta;
            ib += tb;
            ic += tc;
        }
    }
    const uint32_t incl[4] = {ia & 0xFFFFu, ia >> 16, ib & 0xFFFFu, ib >> 16};
#pragma unroll
    for (int h = 0; h < 4; h++)
    {
        if (!rowExists[h])
            continue;
        L.words[(uint64_t) rowId[h] * nw + w] = LatWord{mainBits[h], incl[h] - cnt[h], colBits[h]};
        if (w == nw - 1)
        {
            const uint32_t total = incl[h], nFlag = (ic >> (8 * h)) & 0xFFu;
            U3 c{0u, 0u, 0u};
            if (rowCls[h] == 0) { c.a = total; c.c = nFlag; }
            else if (rowCls[h] == 1) { c.b = total; c.c = nFlag; }
            else c.c = total;
            L.rowCounts[rowId[h]] = c;
        }
    }
}

/* After the scan of the row counts, one thread per word: make the word's prefix an absolute output index and
 * give every word of a class-0/1 row the row's class-2 start and its two column bits. */
struct LatticePatchArgs
{
    Lattice L;
    uint32_t numWords;
};

__global__ __launch_bounds__(256) void latticePatchKernel(Lanes<LatticePatchArgs> lanes)
{
    const LatticePatchArgs A = lanes.a[blockIdx.y];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.numWords)
        return;
    const Lattice L = A.L;            /* a copy: the fields live in scalar registers, not behind kernel-argument loads */
    const uint32_t row = i / L.nw;
    const uint32_t z2 = row / L.rowsPerLayer + L.z2First, y2 = row % L.rowsPerLayer;
    const uint32_t rc = L.rowClass(y2, z2);
    const U3 rs = L.rowCounts[row];
    const U3 tot = *L.totals;
    const uint32_t off2 = tot.a + tot.b;
    const uint32_t mainBase = rc == 0 ? rs.a : (rc == 1 ? tot.a + rs.b : off2 + rs.c);
    /* bits 31 / 30 of word 0 / the top word keep their value through this kernel, so rows that straddle blocks are safe */
    const uint32_t x0 = L.words[(uint64_t) row * L.nw].flag & LAT_FLAG_X0;
    const uint32_t top = L.words[(uint64_t) row * L.nw + (L.topx >> 6)].flag & LAT_FLAG_TOP;
    L.words[i].prefix += mainBase;
    L.words[i].flag = (off2 + rs.c) | x0 | top;
}

/* 1.0f / x, correctly rounded, in three vector instructions where the compiler's division takes twelve: v_rcp_f32 and one
 * Newton step in fused arithmetic give the correctly rounded quotient for EVERY x with a biased exponent of 1 .. 252 --
 * checked exhaustively on gfx950 over all 2^32 inputs (tools/microbench/rcp_exact.hip: the only differences are zeros and
 * denormals, |x| >= 2^126, infinities and NaNs) -- and those take the division.  latticeVertices spent a third of an
 * iteration's vector instructions in it. */
__device__ __forceinline__ float exactRcp(float x)
{
    const uint32_t e = (__float_as_uint(x) >> 23) & 0xFFu;
    if (e - 1u < 252u)
    {
        const float r = __builtin_amdgcn_rcpf(x);
        return fmaf(r, fmaf(-x, r, 1.0f), r);
    }
    return 1.0f / x;
}

/* Positions (interp, kernels/marching.cl:130-138) and external keys of the existing points.  One wave per QUAD of lattice
 * rows -- (z2, y2) = (2 cz + pz, 2 cy + py): the four rows whose points hang off the corner row (cy, cz) -- lane = x2 within
 * a word.  The kernel was bound by memory latency times resident waves, not by bytes or instructions, so a wave carries as
 * much independent work as its registers allow: the four rows share the values of the corner row (five field loads per four
 * points instead of eight), the lattice words of a layer's two rows (8 words of each per trip) arrive with ONE coalesced load
 * -- lanes 0-31 hold row 2 cy's dwords, lanes 32-63 row 2 cy + 1's -- and are read out of it lane by lane into scalar
 * registers, and all of a trip's loads are requested before the first point is computed (a trip is 512 points: 255 cells).
 * Measured on cfg3, two buckets per launch: one row per wave 144 us, a pair 107, the quad 101 -- 517 MB of traffic at 5.1 TB/s. */
struct LatticeVerticesArgs
{
    Lattice L;
    FieldView F;
    float *outVertices;
    uint64_t *outKeys;
    uint32_t gox, goy, goz;
    uint64_t keyOffset;
    VertexTransform X;
    uint32_t numPairs;       /* quads: ceil(layers / 2) x ceil(rowsPerLayer / 2) */
};

/* XF: the scale / bias of ScaleBiasFilter for every bucket of the launch (1), for none (0), or as each bucket says (2: both
 * results computed and selected -- nine vector instructions of an iteration's ~40 instead of three) */
template<int XF>
__global__ __launch_bounds__(256) void latticeVerticesKernel(Lanes<LatticeVerticesArgs> lanes)
{
    const LatticeVerticesArgs A = lanes.a[blockIdx.y];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t quadId = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (quadId >= A.numPairs)
        return;
    const Lattice L = A.L;            /* a copy: the fields live in scalar registers, not behind kernel-argument loads */
    const FieldView F = A.F;
    float *const outVertices = A.outVertices;
    uint64_t *const outKeys = A.outKeys;
    const uint32_t gox = A.gox, goy = A.goy, goz = A.goz;
    const uint64_t keyOffset = A.keyOffset;
    const VertexTransform X = A.X;
    const uint32_t pairsPerLayer = (L.rowsPerLayer + 1) / 2;
    const uint32_t layers = L.z2Last - L.z2First + 1;
    const uint32_t slab = quadId / pairsPerLayer, cy = quadId % pairsPerLayer;      /* slab: the layers z2 = 2 cz and 2 cz + 1 */
    const uint32_t cz = (L.z2First >> 1) + slab;
    const bool hasY = 2 * cy + 1 < L.rowsPerLayer;      /* a layer's last row (y2 = top.y) is alone ... */
    const bool hasZ = 2 * slab + 1 < layers;            /* ... and so is the last layer */
    /* endpoint A of every point of the quad is a corner of the row (cy, cz); endpoint B is A + (px, py, pz) */
    const float *fieldB[2][2];
#pragma unroll
    for (int pz = 0; pz < 2; pz++)
#pragma unroll
        for (int py = 0; py < 2; py++)
            fieldB[pz][py] = F.field + (uint64_t) (cy + (hasY ? py : 0) + F.zStride * (cz + (hasZ ? pz : 0)) + (uint32_t) F.zBias) * F.pitch;
    const float *const field0 = fieldB[0][0];
    const uint32_t px = lane & 1;
    const uint32_t rowDwords = 4 * L.nw;
    const uint32_t *wordsOf[2];      /* per layer: row y2 = 2 cy's dwords; row 2 cy + 1's follow */
#pragma unroll
    for (int pz = 0; pz < 2; pz++)
        wordsOf[pz] = (const uint32_t *) (L.words + (uint64_t) ((2 * slab + (hasZ ? pz : 0)) * L.rowsPerLayer + 2 * cy) * L.nw);
    /* interp (kernels/marching.cl:130-138), scale / bias, the vertex and -- for an external one -- its key.  Endpoint A =
     * owner corner, B = A + (px, py, pz): A has the lower local corner id in every cell */
    auto emit = [&](uint32_t x2, uint32_t cx, uint32_t ex, uint32_t ey, uint32_t ez, uint32_t idx, float isoA, float isoB, bool withKey)
    {
        const float inv = exactRcp(isoA - isoB);
        const float t = isoA * inv;
        float vx = fmaf(t, (float) ex, (float) (cx + gox));
        float vy = fmaf(t, (float) ey, (float) (cy + goy));
        float vz = fmaf(t, (float) ez, (float) (cz + goz));
        if (XF == 2 ? (bool) X.enabled : XF == 1)
        {
            vx = fmaf(vx, X.scale, X.bx);
            vy = fmaf(vy, X.scale, X.by);
            vz = fmaf(vz, X.scale, X.bz);
        }
        outVertices[3 * (uint64_t) idx + 0] = vx;
        outVertices[3 * (uint64_t) idx + 1] = vy;
        outVertices[3 * (uint64_t) idx + 2] = vz;
        if (withKey)
            outKeys[idx] = (((uint64_t) (2 * cz + ez) << (2 * KEY_AXIS_BITS)) | ((uint64_t) (2 * cy + ey) << KEY_AXIS_BITS) | (uint64_t) x2) + keyOffset;
    };
    constexpr int G = 8;
    for (uint32_t w0 = 0; w0 < L.nw; w0 += G)
    {
        /* dword (lane % 32) of the trip's eight words of row y2 = 2 cy (lanes 0-31) and 2 cy + 1 (lanes 32-63), clamped */
        const uint32_t dw = min(4 * w0 + (lane & 31u), rowDwords - 1) + ((lane >> 5) != 0 && hasY ? rowDwords : 0u);
        const uint32_t latDword[2] = {wordsOf[0][dw], wordsOf[1][dw]};
        float iso0[G], iso1[2][2][G];
#pragma unroll
        for (int j = 0; j < G; j++)
        {
            /* every lane loads (clamped inside the row), so the loads do not wait on the masks; checking the words for
             * emptiness first was measured: -7 % on surface-like data, +8 % on the noise cloud (exposed latency) */
            const uint32_t w = min(w0 + j, L.nw - 1);
            const uint32_t cx = min(w * 32 + (lane >> 1), L.cw - px);
            iso0[j] = field0[cx];
#pragma unroll
            for (int pz = 0; pz < 2; pz++)
#pragma unroll
                for (int py = 0; py < 2; py++)
                    iso1[pz][py][j] = fieldB[pz][py][cx + px];
        }
#pragma unroll
        for (int j = 0; j < G; j++)
        {
            if (w0 + j >= L.nw)
                break;
            const uint32_t x2 = (w0 + j) * 64 + lane;
            const uint32_t cx = x2 >> 1;
#pragma unroll
            for (int pz = 0; pz < 2; pz++)
            {
                if (pz == 1 && !hasZ)
                    break;
#pragma unroll
                for (int py = 0; py < 2; py++)
                {
                    if (py == 1 && !hasY)
                        break;
                    const uint32_t y2 = 2 * cy + py, z2 = 2 * cz + pz;
                    const uint32_t cls = L.rowClass(y2, z2);
                    const uint32_t at = 32 * py + 4 * j;
                    const uint64_t exists = (uint64_t) readLane(latDword[pz], at) | (uint64_t) readLane(latDword[pz], at + 1) << 32;
                    const uint32_t prefix = readLane(latDword[pz], at + 2);
                    if (!((exists >> lane) & 1))
                        continue;
                    /* (a class-0/1 row's two class-2 points, x2 = 0 and x2 = top.x, are not in the mask: see below) */
                    const uint32_t idx = prefix + (uint32_t) __popcll(exists & ((1ull << lane) - 1));
                    emit(x2, cx, px, (uint32_t) py, (uint32_t) pz, idx, iso0[j], iso1[pz][py][j], cls != 0);
                }
            }
        }
    }
    /* The class-2 points inside class-0/1 rows -- x2 = 0 and x2 = top.x of each of the quad's four rows, where the row's flag
     * word says they exist (the same in every word of the row, latticePatchKernel) -- once per wave, a lane per candidate,
     * instead of two comparisons and their scalar bookkeeping in each of the loop's iterations (24 per wave on cfg3: the
     * kernel issued as many scalar as vector instructions, 63 % of the CU's scalar unit). */
    if (lane < 8)
    {
        const uint32_t r = lane >> 1, py = r & 1u, pz = r >> 1, top = lane & 1u;
        const uint32_t y2 = 2 * cy + py, z2 = 2 * cz + pz;
        if ((py == 0 || hasY) && (pz == 0 || hasZ) && L.rowClass(y2, z2) != 2)
        {
            const uint32_t flag = (pz ? wordsOf[1] : wordsOf[0])[(py ? rowDwords : 0u) + 3u];
            if (flag & (top ? LAT_FLAG_TOP : LAT_FLAG_X0))
            {
                const uint32_t x2 = top ? L.topx : 0u, cx = x2 >> 1;            /* top.x is even: px = 0 */
                const uint32_t idx = (flag & LAT_FLAG_INDEX) + (top ? flag >> 31 : 0u);
                const float *const rowB = F.field + (uint64_t) (cy + py + F.zStride * (cz + pz) + (uint32_t) F.zBias) * F.pitch;
                emit(x2, cx, 0u, py, pz, idx, field0[cx], rowB[cx], true);
            }
        }
    }
}

/* One thread per occupied cell: look up the welded index of each of the cell's vertices, then emit its
 * triangles (the index half of generateElements + reindex).
 *
 * The cell's 19 possible vertices lie in 9 lattice rows (y2, z2) = (2y + 0..2, 2z + 0..2) at x2 = 2x + 0..2, and
 * x2 = 2x + 2 still ranks inside the word of 2x (a full-word popcount when it is the next word's bit 0), so the
 * lookup is nine independent 16-byte loads issued together, then 19 unrolled edge tests -- vertex slots follow
 * edge ids (makeTables), so no per-code key table is needed.
 *
 * A block's cells are consecutive in the compacted list, so their index ranges are one contiguous span of the
 * output: it is assembled in LDS and written with fully coalesced stores. */
struct LatticeTrianglesArgs
{
    Lattice L;
    DevTables T;
    const uint2 *cells, *viStart;
    uint32_t *indices;
    const U3 *batchTotals;
};

__global__ __launch_bounds__(256) void latticeTrianglesKernel(Lanes<LatticeTrianglesArgs> lanes)
{
    __shared__ uint32_t sIdx[256][MAX_CELL_VERTICES];
    __shared__ uint16_t sRef[256 * MAX_CELL_INDICES];   /* thread * 13 + vertex slot: 18 KB instead of 36 KB of indices */
    __shared__ uint32_t sSpan;
    const LatticeTrianglesArgs A = lanes.a[blockIdx.y];
    const Lattice L = A.L;            /* a copy: the fields live in scalar registers, not behind kernel-argument loads */
    const DevTables T = A.T;
    const uint2 *const cells = A.cells, *const viStart = A.viStart;
    uint32_t *const indices = A.indices;
    const uint32_t numCells = A.batchTotals->a;         /* grid covers the host's count; the device value rules */
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x * blockDim.x >= numCells)
        return;
    const uint32_t blockBase = viStart[blockIdx.x * blockDim.x].y;
    if (gid < numCells)
    {
        const uint2 cell = cells[gid];
        const uint32_t x = cell.x & 0xFFFFu, y = cell.x >> 16, z = cell.y & 0xFFFFu;
        const uint32_t code = cell.y >> 16;                  /* packed by compactRowCellsKernel */
        const uint32_t local = viStart[gid].y - blockBase;
        /* the whole record at once (index count in word 3, index bytes in words 4..12), together with the nine words */
        const uint4 *rec4 = (const uint4 *) T.rec + code * 4;
        const uint4 r0 = rec4[0], r1 = rec4[1], r2 = rec4[2];
        const uint32_t r12 = T.rec[code * 16 + 12];
        const uint32_t ni = r0.w >> 8;
        /* the nine rows' words */
        const uint32_t wIdx = (2 * x) >> 6, sh = (2 * x) & 63;
        LatWord wd[9];
        bool rowFlagged[9];
#pragma unroll
        for (int r = 0; r < 9; r++)
        {
            const uint32_t y2 = 2 * y + (r % 3), z2 = 2 * z + (r / 3);
            const uint32_t row = (z2 - L.z2First) * L.rowsPerLayer + y2;
            wd[r] = L.words[(uint64_t) row * L.nw + wIdx];
            rowFlagged[r] = y2 == 0 || y2 == L.topy || z2 == L.z2First;
        }
        /* bits of the word below x2 = 2x + px */
        uint64_t below[3];
        below[0] = (1ull << sh) - 1;
        below[1] = (2ull << sh) - 1;
        below[2] = sh == 62 ? ~0ull : (4ull << sh) - 1;
        const bool atX0 = x == 0, atTop = 2 * x + 2 == L.topx;
        uint32_t slot = 0;
#pragma unroll
        for (int e = 0; e < NUM_EDGES; e++)
        {
            const int a = edgeIndices[e][0], b = edgeIndices[e][1];
            const int px = (a & 1) + (b & 1), py = ((a >> 1) & 1) + ((b >> 1) & 1), pz = ((a >> 2) & 1) + ((b >> 2) & 1);
            const int r = py + 3 * pz;
            if (((code >> a) ^ (code >> b)) & 1u)
            {
                uint32_t idx = wd[r].prefix + (uint32_t) __popcll(wd[r].mask & below[px]);
                if (px == 0 && atX0 && !rowFlagged[r])
                    idx = wd[r].flag & LAT_FLAG_INDEX;
                if (px == 2 && atTop && !rowFlagged[r])
                    idx = (wd[r].flag & LAT_FLAG_INDEX) + (wd[r].flag >> 31);
                sIdx[threadIdx.x][slot++] = idx;
            }
        }
        /* a reference is the word index of the vertex's welded index inside sIdx: one LDS read resolves it */
        const uint32_t first = threadIdx.x * MAX_CELL_VERTICES;
        const uint32_t iws[9] = {r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r12};
#pragma unroll
        for (uint32_t q = 0; q < 9; q++)
        {
            if (4 * q >= ni)
                break;
            const uint32_t iw = iws[q];
            const uint32_t left = ni - 4 * q;
            sRef[local + 4 * q] = (uint16_t) (first + (iw & 0xFF));
            if (left > 1) sRef[local + 4 * q + 1] = (uint16_t) (first + ((iw >> 8) & 0xFF));
            if (left > 2) sRef[local + 4 * q + 2] = (uint16_t) (first + ((iw >> 16) & 0xFF));
            if (left > 3) sRef[local + 4 * q + 3] = (uint16_t) (first + (iw >> 24));
        }
        if (gid == numCells - 1 || threadIdx.x == blockDim.x - 1)
            sSpan = local + ni;
    }
    __syncthreads();
    const uint32_t span = sSpan;
    /* four positions per thread and trip, so that the two dependent LDS reads of a position overlap the others' */
    const uint32_t *flat = &sIdx[0][0];
    for (uint32_t k0 = 0; k0 < span; k0 += 4 * 256)
    {
        uint32_t ref[4], val[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const uint32_t k = k0 + 256 * u + threadIdx.x;
            ref[u] = k < span ? sRef[k] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            val[u] = flat[ref[u]];
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const uint32_t k = k0 + 256 * u + threadIdx.x;
            if (k < span)
                indices[blockBase + k] = val[u];
        }
    }
}

/* The same emission without the compacted cell list: one wave per ROW of cells, lane = cell x, 64 cells at a time.  The
 * code bytes are one coalesced read, the row's first index slot comes from the scan of the row totals and a cell's slot
 * from a wave scan of the per-code index counts -- so the (cell, first slot) records that compactRowCellsKernel wrote and
 * latticeTrianglesKernel read back (16 bytes per occupied cell each way, and a dependent load in front of everything
 * else) do not exist.  The nine lattice rows are the same for the whole wave.  Index order is unchanged: rows in (z, y)
 * order, cells by x, a cell's indices in table order.  Each wave stages its own span in LDS; no workgroup barrier. */
struct LatticeTrianglesRowArgs
{
    Lattice L;
    CodeView C;
    DevTables T;
    const U3 *rowCounts, *rowStarts;
    uint32_t zFirst;
    uint32_t *indices;
    uint32_t numRows;
    uint32_t *trash;         /* a word nobody reads: where the stores of a chunk without indices go */
};

/* Software-pipelined over the chunks of a row: nothing chunk n + 1 needs is requested after chunk n's stores.  Vector loads
 * and stores retire in order (one counter, vmcnt), so a load issued behind a chunk's stores is usable only when all of them
 * have reached the L2.  Here the loads of chunk n + 1 (its code records; the code bytes of chunk n + 2) are issued BEFORE the
 * stores of chunk n, and every chunk issues a number of stores the compiler can count (three per quarter of the wave, a fourth
 * for a quarter of more than 192 indices; one of more than 256 drains), so the wait in front of chunk n + 1 is `vmcnt(12)`:
 * chunk n's stores stay in flight.  (A path through the loop without stores -- an empty chunk skipped, a predicated store --
 * would make that wait a full one again, which is why positions past a quarter's end store its last index once more and an
 * empty quarter stores to a word nobody reads.)  The
 * lattice words of the row's nine lattice rows are staged once in LDS (dynamic: 4 x 9 x nw x 16 bytes) instead of nine
 * 16-byte loads per cell; a vertex index is its row's word prefix plus the mask bits below its x2, counted once per row
 * (not per edge); the references go out four or eight bytes at a time.  Measured on cfg3 (two buckets per launch): 216 -> 195 us,
 * 3.16 -> 2.90 ms per step (profiles/NOTES_r04.md section 9). */
__global__ __launch_bounds__(256) void latticeTrianglesRowKernel(Lanes<LatticeTrianglesRowArgs> lanes)
{
    const LatticeTrianglesRowArgs A = lanes.a[blockIdx.y];
    const Lattice L = A.L;
    const CodeView C = A.C;
    const DevTables T = A.T;
    const U3 *const rowCounts = A.rowCounts, *const rowStarts = A.rowStarts;
    const uint32_t zFirst = A.zFirst, numRows = A.numRows;
    uint32_t *const indices = A.indices, *const trash = A.trash;
    __shared__ uint32_t sIdx[4][64][MAX_CELL_VERTICES];
    __shared__ uint8_t sRef[4][64 * MAX_CELL_INDICES];
    extern __shared__ uint4 sLat[];                 /* [4][9][nw] */
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);
    if (r >= numRows)
        return;
    /* everything the row needs that does not depend on another load is requested at once: the row's totals, its first
     * code bytes and the lattice words (an empty row returns with those requests in flight; the rows route is the dense one) */
    const uint32_t y = r % L.ch, z = r / L.ch + zFirst;
    const uint8_t *codeRow = C.codes + ((uint64_t) (z - C.z0) * C.ch + y) * C.cw;
    const uint32_t nw = L.nw, cw = L.cw;
    const uint32_t occupied = rowCounts[r].a;
    uint32_t indexBase = rowStarts[r].c;
    uint32_t c0 = lane < cw ? codeRow[lane] : 0u;
    uint32_t c1 = lane + 64 < cw ? codeRow[lane + 64] : 0u;
    uint4 *const myLat = sLat + wv * 9 * nw;
    bool rowFlagged[9];
#pragma unroll
    for (int q = 0; q < 9; q++)
    {
        const uint32_t y2 = 2 * y + (q % 3), z2 = 2 * z + (q / 3);
        rowFlagged[q] = y2 == 0 || y2 == L.topy || z2 == L.z2First;
    }
    /* the 9 x nw lattice words, 64 per request: entry e is word e % nw of row e / nw (nw <= 64: the division is a multiply) */
    const uint32_t entries = 9 * nw, recip = (65536u + nw - 1) / nw;
    auto latWord = [&](const uint32_t e) -> const uint4 *
    {
        const uint32_t q = (e * recip) >> 16, w = e - q * nw;
        const uint32_t qz = (q * 11u) >> 5, qy = q - 3u * qz;
        return (const uint4 *) (L.words + (uint64_t) ((2 * z + qz - L.z2First) * L.rowsPerLayer + 2 * y + qy) * nw + w);
    };
    const uint32_t e0 = min(lane, entries - 1), e1 = min(lane + 64, entries - 1);
    const uint4 w0 = *latWord(e0), w1 = *latWord(e1);
    if (__builtin_amdgcn_readfirstlane(occupied) == 0)
        return;
    const uint4 *rec4 = (const uint4 *) T.rec;
    uint4 n1 = rec4[c0 * 4 + 1], n2 = rec4[c0 * 4 + 2], n3 = rec4[c0 * 4 + 3];
    myLat[e0] = w0;
    myLat[e1] = w1;
    for (uint32_t e = 128 + lane; e < entries; e += 64)
        myLat[e] = *latWord(e);
    uint32_t (*myIdx)[MAX_CELL_VERTICES] = sIdx[wv];
    uint8_t *myRef = sRef[wv];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    /* a class-0/1 row's flag word is the same in all of its words (latticePatchKernel) */
    uint32_t rowFlag[9];
#pragma unroll
    for (int q = 0; q < 9; q++)
        rowFlag[q] = __builtin_amdgcn_readfirstlane(myLat[q * nw].w);
    __builtin_amdgcn_s_waitcnt(0x0F70);            /* vmcnt(0): the loop is entered with nothing in flight, so that the waits
                                                    * inside it are the steady state's, not the first chunk's */
    for (uint32_t x0 = 0; x0 < cw; x0 += 64)
    {
        const uint32_t x = x0 + lane;
        const uint32_t code = c0;
        const uint4 r1 = n1, r2 = n2, r3 = n3;
        /* the next chunk's records and the code bytes of the one after it */
        c0 = c1;
        c1 = x + 128 < cw ? codeRow[x + 128] : 0u;
        n1 = rec4[c0 * 4 + 1];
        n2 = rec4[c0 * 4 + 2];
        n3 = rec4[c0 * 4 + 3];
        const uint32_t ni = r3.w >> 8;             /* code 0 (not occupied, or past the row's end): an all-zero record */
        const uint32_t incl = waveInclusiveScan(ni);
        const uint32_t span = readLane(incl, 63);
        const uint32_t local = incl - ni;
        if (code != 0)
        {
            const uint32_t wIdx = (2 * x) >> 6, sh = (2 * x) & 63;
            const uint64_t below0 = (1ull << sh) - 1, below1 = (2ull << sh) - 1;
            const bool atX0 = x == 0, atTop = 2 * x + 2 == L.topx;
            uint32_t base[9], two[9];
#pragma unroll
            for (int q = 0; q < 9; q++)
            {
                const bool throughCorners = (q % 3) != 1 && (q / 3) != 1;
                const uint4 wd = myLat[q * nw + wIdx];
                const uint64_t mask = (uint64_t) wd.x | (uint64_t) wd.y << 32;
                base[q] = wd.z + (uint32_t) __popcll(mask & (throughCorners ? below1 : below0));
                two[q] = throughCorners ? 0u : (uint32_t) (mask >> sh) & 3u;
            }
            const uint32_t used = r3.z;
            uint32_t *dst = myIdx[lane];
#pragma unroll
            for (int e = 0; e < NUM_EDGES; e++)
            {
                const int a = edgeIndices[e][0], b = edgeIndices[e][1];
                const int px = (a & 1) + (b & 1), py = ((a >> 1) & 1) + ((b >> 1) & 1), pz = ((a >> 2) & 1) + ((b >> 2) & 1);
                const int q = py + 3 * pz;
                const bool throughCorners = (q % 3) != 1 && (q / 3) != 1;
                if (used & (1u << e))
                {
                    uint32_t idx = base[q];
                    if (!throughCorners && px == 1)
                        idx += two[q] & 1u;
                    if (!throughCorners && px == 2)
                        idx += (uint32_t) __popc(two[q]);
                    if (px == 0 && atX0 && !rowFlagged[q])
                        idx = rowFlag[q] & LAT_FLAG_INDEX;
                    if (px == 2 && atTop && !rowFlagged[q])
                        idx = (rowFlag[q] & LAT_FLAG_INDEX) + (rowFlag[q] >> 31);
                    *dst++ = idx;
                }
            }
            const uint32_t first4 = (lane & 15u) * (MAX_CELL_VERTICES * 0x01010101u);
            const uint32_t iws[9] = {r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x};
            uint8_t *const ref = myRef + local;
            const uint32_t whole = ni >> 2;
            /* two words per store where two are whole (an unaligned 8-byte LDS store costs what a 4-byte one does) */
#pragma unroll
            for (uint32_t q = 0; q < 8; q += 2)
            {
                if (q + 1 < whole)
                {
                    const uint64_t v = (uint64_t) (iws[q] + first4) | (uint64_t) (iws[q + 1] + first4) << 32;
                    __builtin_memcpy(ref + 4 * q, &v, 8);
                }
                else if (q < whole)
                {
                    const uint32_t v = iws[q] + first4;
                    __builtin_memcpy(ref + 4 * q, &v, 4);
                }
            }
            if (8 < whole)
            {
                const uint32_t v = iws[8] + first4;
                __builtin_memcpy(ref + 32, &v, 4);
            }
            const uint32_t rest = ni & 3u, tail = r3.y + first4;
            if (rest > 0) ref[4 * whole] = (uint8_t) tail;
            if (rest > 1) ref[4 * whole + 1] = (uint8_t) (tail >> 8);
            if (rest > 2) ref[4 * whole + 2] = (uint8_t) (tail >> 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        /* The write-out, one QUARTER of the wave (16 cells, whose 16 x 13 slots a reference byte can address) at a time: a
         * quarter's positions are the interval between two wave-uniform bounds, so a position costs an add, a clamp and the
         * two LDS reads.  Every lane stores in the three trips every quarter makes (192 positions; 16 cells hold 189 on the
         * uniform cloud): a position past the quarter's end is its last once more (same value, same address), an empty
         * quarter stores to a word nobody reads -- so that the number of stores does not depend on the data. */
        const uint32_t bound[5] = {0u, readLane(incl, 15), readLane(incl, 31), readLane(incl, 47), span};
        uint32_t *const out = indices + indexBase;
        bool drain = false;
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const uint32_t lo = bound[i], hi = bound[i + 1];
            const uint32_t *const slots = &myIdx[16 * i][0];
            const uint32_t last = hi != lo ? hi - 1 : lo;
            uint32_t *const dstq = hi != lo ? out : trash - lo;
            const uint32_t refMask = hi != lo ? 0xFFu : 0u;      /* an empty quarter's reference byte is stale: slot 0 then */
            uint32_t k[3], val[3];
#pragma unroll
            for (int u = 0; u < 3; u++)
            {
                k[u] = min(lo + 64 * u + lane, last);
                val[u] = myRef[k[u]] & refMask;
            }
#pragma unroll
            for (int u = 0; u < 3; u++)
                val[u] = slots[val[u]];
#pragma unroll
            for (int u = 0; u < 3; u++)
                dstq[k[u]] = val[u];
            if (hi - lo > 192)
            {
                const uint32_t k3 = min(lo + 192 + lane, last);
                dstq[k3] = slots[myRef[k3]];
            }
            if (hi - lo > 256)
            {
#pragma unroll 1
                for (uint32_t k0 = lo + 256 + lane; k0 < hi; k0 += 64)
                    out[k0] = slots[myRef[k0]];
                drain = true;
            }
        }
        if (drain)
            __builtin_amdgcn_s_waitcnt(0x0F70);    /* vmcnt(0): the count of those stores is not known at compile time */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        indexBase += span;
    }
}

uint32_t bitsFor(uint32_t maxValue)
{
    uint32_t b = 1;
    while ((maxValue >> b) != 0)
        b++;
    return b;
}

/* latticeMaskKernel's table.  A corner (x, y, z) owns the lattice points towards +x, +y, +z, +xy, +xz, +yz, +xyz; the point on
 * an edge exists iff an occupied cell containing the edge sees different signs at its ends.  The cells are the eight around
 * the corner, (x - ox, y - dy, z - dz), in which the corner is local corner a = ox | dy << 1 | dz << 2.  Byte dy | dz << 1 of an
 * entry: what the cell gives as the corner's own cell (ox = 0); byte 4 + (dy | dz << 1): as the cell to its left (ox = 1),
 * which only holds the edges without an x component.  Bits: 0 +x, 1 +xy, 2 +xz, 3 +xyz, 5 +y, 6 +z, 7 +yz. */
std::vector<uint64_t> makeEdgeLut()
{
    std::vector<uint64_t> lut(256, 0);
    const int dir[7] = {1, 3, 5, 7, 2, 4, 6};          /* the edge's direction as a corner offset: +x, +xy, +xz, +xyz, +y, +z, +yz */
    const int bit[7] = {0, 1, 2, 3, 5, 6, 7};
    for (uint32_t code = 1; code < 255; code++)
        for (int ox = 0; ox < 2; ox++)
            for (int dy = 0; dy < 2; dy++)
                for (int dz = 0; dz < 2; dz++)
                {
                    const int a = ox | (dy << 1) | (dz << 2);
                    uint32_t bits = 0;
                    for (int e = 0; e < 7; e++)
                        if ((a & dir[e]) == 0)          /* the other end, a + dir, is a corner of the same cell */
                            bits |= (((code >> a) ^ (code >> (a | dir[e]))) & 1u) << bit[e];
                    lut[code] |= (uint64_t) bits << (8 * ((dy | (dz << 1)) + 4 * ox));
                }
    return lut;
}

/* latticeMaskWordKernel's tables, [r = dy | dz << 1][code]: low byte = what the cell gives the corner as its own cell, high
 * byte = as the cell to its left; bit 2h + 1 = the odd point of lattice row h (+x, +xy, +xz, +xyz), bit 2h = its even point
 * (h = 1 .. 3: +y, +z, +yz) -- a rearrangement of makeEdgeLut's bytes */
std::vector<uint16_t> makeEdgeLut16()
{
    const std::vector<uint64_t> lut = makeEdgeLut();
    auto remap = [](uint32_t old)
    {
        uint32_t v = 0;
        for (int h = 0; h < 4; h++)
        {
            v |= ((old >> h) & 1u) << (2 * h + 1);
            if (h > 0)
                v |= ((old >> (4 + h)) & 1u) << (2 * h);
        }
        return v;
    };
    std::vector<uint16_t> out(4 * 256, 0);
    for (uint32_t code = 0; code < 256; code++)
        for (int r = 0; r < 4; r++)
        {
            const uint32_t own = (uint32_t) (lut[code] >> (8 * r)) & 0xFFu, left = (uint32_t) (lut[code] >> (32 + 8 * r)) & 0xFFu;
            out[r * 256 + code] = (uint16_t) (remap(own) | remap(left) << 8);
        }
    return out;
}

struct Readback
{
    U3 totals;              /* occupied cells, vertices, indices of the (sub-)swathe */
    uint32_t numWelded;
    uint32_t firstExternal;
    U3 classTotals;         /* lattice weld: welded vertices per class (internal, unflagged top, flagged) */
    U3 batchTotals;         /* lattice weld: occupied cells / vertices / indices of the batch */
};

} // namespace

struct mlsgpu_marching
{
    mlsgpu_ctx *ctx = nullptr;
    uint32_t maxWidth = 0, maxHeight = 0, maxDepth = 0, maxSwathe = 0;
    uint32_t imageWidth = 0, imageHeight = 0, zStride = 0;
    uint64_t fieldRows = 0;
    uint64_t vertexSpace = 0, indexSpace = 0, swatheCells = 0;
    bool wideKeys = false;
    HostTables tables;

    float *dField = nullptr;
    uchar2 *dCount = nullptr;
    ushort2 *dStart = nullptr;
    uint8_t *dData = nullptr;
    uint32_t *dKey = nullptr;
    uint32_t *dCodeRec = nullptr;
    uint64_t *dEdgeLut = nullptr;           /* latticeMaskKernel: per code byte, the lattice points a cell gives a corner */
    uint16_t *dEdgeLut16 = nullptr;         /* latticeMaskWordKernel: the same, per (dy, dz) */
    uint2 *dCells = nullptr, *dViStart = nullptr, *dHistogram = nullptr;
    U3 *dTileSums3 = nullptr;
    float4 *dVertices = nullptr;
    void *dKeysA = nullptr, *dKeysB = nullptr;
    uint32_t *dValsA = nullptr, *dValsB = nullptr;
    uint32_t *dIndices = nullptr, *dIndexRemap = nullptr;
    float *dWelded = nullptr;
    uint64_t *dWeldedKeys = nullptr;
    uint32_t *dHist = nullptr, *dTileSums = nullptr;
    Readback *dReadback = nullptr;          /* device words */
    Readback *hReadback = nullptr;          /* pinned */
    HostMailbox box;                        /* the totals a host decision waits for (swathe totals, welded counts) */
    uint2 *hHistogram = nullptr;            /* pinned, maxDepth entries (viReadback in the reference) */

    /* lattice weld (single-swathe buckets) */
    uint8_t *dCellCode = nullptr;
    U3 *dRowCounts = nullptr, *dRowStarts = nullptr;   /* per row of cells of the swathe: (occupied, vertices, indices) */
    LatWord *dLatWords = nullptr;
    U3 *dLatRows = nullptr;
    uint64_t latRowsMax = 0;
    uint32_t latWords = 0;
    bool legacyBuffers = true;              /* false: every bucket fits one swathe, sort path never needed */

    uint64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    VertexTransform transform = {1.0f, 0.0f, 0.0f, 0.0f, 0};

    /* per-generate state */
    bool direct = false;                    /* this generate() call uses the lattice weld */
    uint32_t codeZ0 = 0;                    /* first cell slice held in dCellCode */
    uint32_t bufferedCells = 0;             /* occupied cells accounted since the last ship-out (direct mode) */
    const mlsgpu_generator *generator = nullptr;
    mlsgpu_output_fn output = nullptr;
    void *outputUser = nullptr;
    uint32_t keyOffset[3] = {0, 0, 0};
    KeyLayout layout = {1, 1, 1};

    FieldView view(const mlsgpu_swathe &sw) const { return FieldView{dField, imageWidth, sw.zStride, sw.zBias}; }
    CodeView codeView(const mlsgpu_swathe &sw) const { return CodeView{dCellCode, sw.width - 1, sw.height - 1, codeZ0}; }
    DevTables devTables() const { return DevTables{dCount, dStart, dData, dKey, dCodeRec}; }

    int generateCells(const mlsgpu_swathe &sw, U3 *totals);
    int sliceHistogram(const mlsgpu_swathe &sw);
    int computeCodes(const mlsgpu_swathe &sw);
    int shipOut(const mlsgpu_swathe &sw, const uint32_t sizes[2], uint32_t zTop, uint32_t zMax);
    int shipOutSorted(const uint32_t sizes[2], uint32_t zMax, mlsgpu_mesh *mesh);
    int shipOutDone(const uint32_t sizes[2], const mlsgpu_mesh &mesh);
    CellCodeArgs cellCodeArgs(const mlsgpu_swathe &sw);
    int addSlices(const mlsgpu_swathe &sw, uint32_t offsets[2], uint32_t &zTop, uint32_t *shipOuts);
    template<typename K> int weld(uint32_t nv, uint32_t zMax);
};

namespace
{

uint64_t marchingSizes(uint32_t maxWidth, uint32_t maxHeight, uint32_t maxDepth, uint32_t maxSwathe,
                       uint64_t meshMemory, const uint32_t alignment[3],
                       uint32_t *imageWidth, uint32_t *imageHeight, uint32_t *swathe,
                       uint64_t *vertexSpace, uint64_t *indexSpace, uint64_t *swatheCells, uint64_t *fieldRows)
{
    /* src/marching.cpp:273-284 */
    *imageWidth = roundUp(maxWidth, alignment[0]);
    *imageHeight = roundUp(maxHeight, alignment[1]);
    *swathe = std::min(maxSwathe, maxDepth) / alignment[2] * alignment[2];
    const uint64_t sliceCells = (uint64_t) (maxWidth - 1) * (maxHeight - 1);
    *swatheCells = sliceCells * *swathe;
    const uint64_t meshCells = meshMemory / MLSGPU_MARCHING_MAX_CELL_BYTES;
    *vertexSpace = meshCells * MAX_CELL_VERTICES;
    *indexSpace = meshCells * MAX_CELL_INDICES;
    /* image height imageHeight * (maxSwathe + 1) (src/marching.cpp:380-381), plus the slack a
     * generator may write when the last swathe is padded up to its Z alignment (src/marching.h:236-237) */
    *fieldRows = (uint64_t) *imageHeight * (*swathe + 1 + alignment[2]);
    const uint64_t vs = *vertexSpace, is = *indexSpace, sc = *swatheCells;
    const bool legacy = *swathe < maxDepth;         /* some bucket may need more than one swathe */
    const uint64_t latRows = (uint64_t) (2 * *swathe + 1) * (2 * maxHeight - 1);
    const uint64_t latWords = (2 * maxWidth - 1 + 63) / 64;
    uint64_t bytes = *fieldRows * *imageWidth * 4;
    bytes += sc * 16 + sc;                          /* cells + viStart + cell codes */
    bytes += (uint64_t) scanTiles(std::max(sc, latRows)) * 12 + 12;    /* tile sums (U3) */
    bytes += latRows * latWords * 16 + latRows * 28;                   /* lattice words, row counts, row info */
    bytes += (uint64_t) *swathe * (maxHeight - 1) * 24;                /* per cell-row counts and starts */
    if (legacy)
    {
        bytes += vs * 16;                           /* unwelded vertices */
        bytes += vs * 8 * 2 + vs * 4 * 2;           /* sort keys + values, ping-pong */
        bytes += vs * 4;                            /* indexRemap */
        bytes += sortHistElems(vs) * 4 + (uint64_t) scanTiles(std::max(sortHistElems(vs), vs)) * 4;
    }
    bytes += is * 4;                                /* indices */
    bytes += vs * 12 + vs * 8;                      /* welded vertices + keys */
    bytes += (uint64_t) maxDepth * 8 + 512 + 1028 + 8192 + 2432 * 4;
    return bytes;
}

} // namespace

MLSGPU_API uint64_t mlsgpu_hip_marching_resource_usage(uint32_t maxWidth, uint32_t maxHeight, uint32_t maxDepth,
                                                       uint32_t maxSwathe, uint64_t meshMemory, const uint32_t alignment[3])
{
    uint32_t iw, ih, sw;
    uint64_t vs, is, sc, fr;
    return marchingSizes(maxWidth, maxHeight, maxDepth, maxSwathe, meshMemory, alignment, &iw, &ih, &sw, &vs, &is, &sc, &fr);
}

MLSGPU_API int mlsgpu_hip_marching_create(mlsgpu_ctx *ctx, uint32_t maxWidth, uint32_t maxHeight, uint32_t maxDepth,
                                          uint32_t maxSwathe, uint64_t meshMemory, const uint32_t alignment[3],
                                          mlsgpu_marching **out)
{
    REQUIRE(ctx != nullptr && out != nullptr && alignment != nullptr, MLSGPU_ERR_INVALID);
    /* src/marching.cpp:359-363 */
    REQUIRE(2 <= maxWidth && maxWidth <= MLSGPU_MARCHING_MAX_DIMENSION, MLSGPU_ERR_INVALID);
    REQUIRE(2 <= maxHeight && maxHeight <= MLSGPU_MARCHING_MAX_DIMENSION, MLSGPU_ERR_INVALID);
    REQUIRE(2 <= maxDepth && maxDepth <= MLSGPU_MARCHING_MAX_DIMENSION, MLSGPU_ERR_INVALID);
    REQUIRE(alignment[0] >= 1 && alignment[1] >= 1 && alignment[2] >= 1, MLSGPU_ERR_INVALID);
    REQUIRE(alignment[2] <= maxSwathe, MLSGPU_ERR_INVALID);
    REQUIRE(meshMemory >= (uint64_t) (maxWidth - 1) * (maxHeight - 1) * MLSGPU_MARCHING_MAX_CELL_BYTES, MLSGPU_ERR_INVALID);
    HIP_CHECK(hipSetDevice(ctx->device));

    mlsgpu_marching *m = new mlsgpu_marching;
    m->ctx = ctx;
    m->maxWidth = maxWidth; m->maxHeight = maxHeight; m->maxDepth = maxDepth;
    marchingSizes(maxWidth, maxHeight, maxDepth, maxSwathe, meshMemory, alignment, &m->imageWidth, &m->imageHeight,
                  &m->maxSwathe, &m->vertexSpace, &m->indexSpace, &m->swatheCells, &m->fieldRows);
    m->zStride = m->imageHeight;
    if (m->vertexSpace >= (uint64_t(1) << 32) || m->indexSpace >= (uint64_t(1) << 32))
    {
        delete m;
        return setError(MLSGPU_ERR_LENGTH, "Marching: meshMemory gives more than 2^32 vertices or indices");
    }
    if (m->swatheCells >= (uint64_t(1) << 32))
    {
        delete m;
        return setError(MLSGPU_ERR_LENGTH, "Marching: more than 2^32 cells per swathe");
    }
    makeTables(m->tables);
    assert(m->tables.data.size() == 8192 && m->tables.key.size() == 2432 * 3);    /* src/marching.cpp:248-251 */

    int rc = MLSGPU_OK;
    auto alloc = [&](void **p, uint64_t bytes) {
        if (rc == MLSGPU_OK && hipMalloc(p, bytes ? bytes : 4) != hipSuccess)
            rc = setError(MLSGPU_ERR_NOMEM, "Marching: cannot allocate %llu bytes", (unsigned long long) bytes);
    };
    const uint64_t vs = m->vertexSpace, is = m->indexSpace, sc = m->swatheCells;
    alloc((void **) &m->dField, m->fieldRows * m->imageWidth * 4);
    alloc((void **) &m->dCount, 512);
    alloc((void **) &m->dStart, 257 * 4);
    alloc((void **) &m->dData, 8192);
    alloc((void **) &m->dKey, 2432 * 4);
    alloc((void **) &m->dCodeRec, 256 * 16 * 4);
    alloc((void **) &m->dEdgeLut, 256 * 8);
    alloc((void **) &m->dEdgeLut16, 4 * 256 * 2);
    alloc((void **) &m->dCells, sc * 8);
    alloc((void **) &m->dViStart, sc * 8);
    alloc((void **) &m->dHistogram, (uint64_t) maxDepth * 8);
    m->legacyBuffers = m->maxSwathe < maxDepth;
    m->latRowsMax = (uint64_t) (2 * m->maxSwathe + 1) * (2 * maxHeight - 1);
    m->latWords = (2 * maxWidth - 1 + 63) / 64;
    alloc((void **) &m->dTileSums3, ((uint64_t) scanTiles(std::max(sc, m->latRowsMax)) + 1) * sizeof(U3));
    alloc((void **) &m->dCellCode, sc + 64);        /* latticeMaskWordKernel reads whole 32-byte words: up to 31 bytes behind the last row */
    alloc((void **) &m->dRowCounts, ((uint64_t) m->maxSwathe * (maxHeight - 1) + 1) * sizeof(U3));
    alloc((void **) &m->dRowStarts, ((uint64_t) m->maxSwathe * (maxHeight - 1) + 1) * sizeof(U3));
    alloc((void **) &m->dLatWords, m->latRowsMax * m->latWords * sizeof(LatWord));
    alloc((void **) &m->dLatRows, m->latRowsMax * sizeof(U3));
    if (m->legacyBuffers)
    {
        alloc((void **) &m->dVertices, vs * 16);
        alloc(&m->dKeysA, (vs + 1) * 8);
        alloc(&m->dKeysB, (vs + 1) * 8);
        alloc((void **) &m->dValsA, (vs + 1) * 4);
        alloc((void **) &m->dValsB, (vs + 1) * 4);
        alloc((void **) &m->dIndexRemap, vs * 4);
        alloc((void **) &m->dHist, (sortHistElems(vs) + 1) * 4);
        alloc((void **) &m->dTileSums, ((uint64_t) scanTiles(std::max(sortHistElems(vs), vs)) + 1) * 4);
    }
    alloc((void **) &m->dIndices, is * 4);
    alloc((void **) &m->dWelded, vs * 12);
    alloc((void **) &m->dWeldedKeys, vs * 8);
    /* + the word latticeTrianglesRowKernel's idle stores hit, + the gate of rowTotalsKernel (zero between launches) */
    alloc((void **) &m->dReadback, sizeof(Readback) + 128);
    if (rc == MLSGPU_OK && hipMemset(m->dReadback, 0, sizeof(Readback) + 128) != hipSuccess)
        rc = setError(MLSGPU_ERR_HIP, "Marching: cannot clear the read-back words");
    if (rc == MLSGPU_OK && hipHostMalloc((void **) &m->hReadback, sizeof(Readback)) != hipSuccess)
        rc = setError(MLSGPU_ERR_NOMEM, "Marching: cannot allocate pinned readback");
    if (rc == MLSGPU_OK && hipHostMalloc((void **) &m->hHistogram, (uint64_t) maxDepth * 8) != hipSuccess)
        rc = setError(MLSGPU_ERR_NOMEM, "Marching: cannot allocate pinned histogram");
    if (rc == MLSGPU_OK)
        rc = m->box.create();
    if (rc != MLSGPU_OK)
    {
        mlsgpu_hip_marching_destroy(m);
        return rc;
    }
    /* upload the tables */
    std::vector<uint32_t> packedKey(2432);
    for (int i = 0; i < 2432; i++)
        packedKey[i] = m->tables.key[3 * i] | (m->tables.key[3 * i + 1] << 8) | (m->tables.key[3 * i + 2] << 16);
    std::vector<uint32_t> codeRec(256 * 16, 0u);
    for (int c = 0; c < 256; c++)
    {
        const uint32_t sv = m->tables.start[c][0], si = m->tables.start[c][1];
        const uint32_t nv = m->tables.count[c][0], ni = m->tables.count[c][1];
        uint64_t klo = 0;
        uint32_t khi = 0;
        for (uint32_t i = 0; i < nv; i++)
        {
            const uint32_t *k3 = &m->tables.key[3 * (sv + i)];
            const uint64_t k6 = k3[0] | (k3[1] << 2) | (k3[2] << 4);
            if (i < 10) klo |= k6 << (6 * i); else khi |= (uint32_t) k6 << (6 * (i - 10));
        }
        uint32_t *rec = &codeRec[c * 16];
        rec[0] = (uint32_t) klo; rec[1] = (uint32_t) (klo >> 32); rec[2] = khi; rec[3] = nv | (ni << 8);
        for (uint32_t i = 0; i < ni; i++)
            rec[4 + i / 4] |= (uint32_t) m->tables.data[si + i] << (8 * (i % 4));
        rec[13] = (ni & 3u) != 0 ? rec[4 + ni / 4] : 0u;
        if (nv != 0)
            for (uint32_t e = 0; e < NUM_EDGES; e++)
                rec[14] |= (((uint32_t) c >> edgeIndices[e][0] ^ (uint32_t) c >> edgeIndices[e][1]) & 1u) << e;
        rec[15] = rec[3];
    }
    hipError_t e = hipMemcpy(m->dCount, m->tables.count, 512, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->dCodeRec, codeRec.data(), codeRec.size() * 4, hipMemcpyHostToDevice);
    const std::vector<uint64_t> edgeLut = makeEdgeLut();
    if (e == hipSuccess) e = hipMemcpy(m->dEdgeLut, edgeLut.data(), edgeLut.size() * 8, hipMemcpyHostToDevice);
    const std::vector<uint16_t> edgeLut16 = makeEdgeLut16();
    if (e == hipSuccess) e = hipMemcpy(m->dEdgeLut16, edgeLut16.data(), edgeLut16.size() * 2, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->dStart, m->tables.start, 257 * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->dData, m->tables.data.data(), 8192, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->dKey, packedKey.data(), 2432 * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess)
    {
        mlsgpu_hip_marching_destroy(m);
        return setError(MLSGPU_ERR_HIP, "Marching: table upload failed: %s", hipGetErrorString(e));
    }
    *out = m;
    return MLSGPU_OK;
}

MLSGPU_API void mlsgpu_hip_marching_destroy(mlsgpu_marching *m)
{
    if (!m)
        return;
    hipSetDevice(m->ctx->device);
    hipFree(m->dField); hipFree(m->dCount); hipFree(m->dStart); hipFree(m->dData); hipFree(m->dKey); hipFree(m->dCodeRec); hipFree(m->dEdgeLut); hipFree(m->dEdgeLut16);
    hipFree(m->dCells); hipFree(m->dViStart); hipFree(m->dHistogram); hipFree(m->dTileSums3);
    hipFree(m->dVertices); hipFree(m->dKeysA); hipFree(m->dKeysB); hipFree(m->dValsA); hipFree(m->dValsB);
    hipFree(m->dIndices); hipFree(m->dIndexRemap); hipFree(m->dWelded); hipFree(m->dWeldedKeys);
    hipFree(m->dHist); hipFree(m->dTileSums); hipFree(m->dReadback);
    hipFree(m->dCellCode); hipFree(m->dRowCounts); hipFree(m->dRowStarts);
    hipFree(m->dLatWords); hipFree(m->dLatRows);
    if (m->hReadback) hipHostFree(m->hReadback);
    if (m->hHistogram) hipHostFree(m->hHistogram);
    m->box.destroy();
    delete m;
}

/* generateCells, src/marching.cpp:500-551: classify the cells of the swathe and return the totals.
 * Leaves the scanned tile sums in dTileSums3 for the compaction pass. */
int mlsgpu_marching::generateCells(const mlsgpu_swathe &sw, U3 *totals)
{
    const CellRange R{sw.width - 1, sw.height - 1, sw.zFirst};
    if (direct)
    {
        /* lattice weld: only the totals are needed now -- reduce the row totals of the (sub-)swathe */
        const uint64_t rows = (uint64_t) R.ch * (sw.zLast - sw.zFirst);
        const U3 *first = dRowCounts + (uint64_t) (sw.zFirst - codeZ0) * R.ch;
        PROPAGATE((scanPhase1<U3, ArrayIn<U3> >(ctx, "kernel.marching.genOccupied.time", ArrayIn<U3>{first},
                                                 R.cw > 0 ? rows : 0, U3{0, 0, 0}, dTileSums3, &dReadback->totals)));
    }
    else
    {
        const uint64_t n = (uint64_t) R.cw * R.ch * (sw.zLast - sw.zFirst);
        ClassifyIn in{codeView(sw), R, dCount};
        PROPAGATE((scanPhase1<U3, ClassifyIn>(ctx, "kernel.marching.genOccupied.time", in, n, U3{0, 0, 0},
                                              dTileSums3, &dReadback->totals)));
    }
    int pend = -1;
    if (ctx->timing) pend = ctx->beginTiming(ctx->statId("kernel.marching.readback.time"));
    PROPAGATE(box.publish(ctx->stream, &dReadback->totals, 3));
    if (pend >= 0) ctx->endTiming(pend);
    PROPAGATE(box.wait(ctx->stream));                    /* the reference's queue.finish(), :548 */
    std::memcpy(&hReadback->totals, box.payload(), sizeof(U3));
    *totals = hReadback->totals;
    return MLSGPU_OK;
}

int mlsgpu_marching::sliceHistogram(const mlsgpu_swathe &sw)
{
    const uint32_t slices = sw.zLast - sw.zFirst;
    LAUNCH(ctx, "kernel.marching.genOccupied.time", sliceHistogramKernel, dim3(slices), dim3(64),
           (const U3 *) dRowCounts, sw.height - 1, sw.zFirst, codeZ0, dHistogram);
    HIP_CHECK(hipMemcpyAsync(hHistogram + sw.zFirst, dHistogram + sw.zFirst, (size_t) slices * 8,
                             hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MLSGPU_OK;
}

template<typename K>
int mlsgpu_marching::weld(uint32_t nv, uint32_t zMax)
{
    K *keysA = static_cast<K *>(dKeysA), *keysB = static_cast<K *>(dKeysB);
    SortResult<K> sorted;
    PROPAGATE(radixSort<K>(ctx, "kernel.marching.sortVertices.time", keysA, dValsA, keysB, dValsB, nv, layout.bits(),
                           true, dHist, dTileSums, &sorted));
    const uint64_t keyOffsetL = ((uint64_t) keyOffset[2] << (2 * KEY_AXIS_BITS + 1))
        | ((uint64_t) keyOffset[1] << (KEY_AXIS_BITS + 1))
        | ((uint64_t) keyOffset[0] << 1);                                   /* src/marching.cpp:594-597 */
    UniqueIn<K> in{sorted.keys, nv};
    CompactVerticesOut<K> outF{sorted.keys, sorted.vals, dVertices, dWelded, dWeldedKeys, dIndexRemap,
                               &dReadback->firstExternal, layout, 2 * zMax, keyOffsetL, nv, transform};
    return exclusiveScan<uint32_t>(ctx, "kernel.marching.compactVertices.time", in, outF, nv, 0u, dTileSums,
                                   &dReadback->numWelded);
}

/* cellCodeKernel over the cells of one top-level swathe */
CellCodeArgs mlsgpu_marching::cellCodeArgs(const mlsgpu_swathe &sw)
{
    const uint32_t cw = sw.width - 1, ch = sw.height - 1;
    const uint32_t rows = cw > 0 ? ch * (sw.zLast - sw.zFirst) : 0u;
    codeZ0 = sw.zFirst;
    return CellCodeArgs{dCellCode, dRowCounts, view(sw), cw, ch, sw.zFirst, rows, (const uchar2 *) dCount};
}

/* ... for the buckets of a batch (one launch); a single bucket is a batch of one */
static int computeCodesLanes(mlsgpu_marching *const *ms, const mlsgpu_swathe *sws, uint32_t count)
{
    mlsgpu_ctx *ctx = ms[0]->ctx;
    Lanes<CellCodeArgs> L;
    uint32_t maxRows = 0;
    for (uint32_t k = 0; k < MAX_LANES; k++)
    {
        if (k < count)
        {
            L.a[k] = ms[k]->cellCodeArgs(sws[k]);
            maxRows = std::max(maxRows, L.a[k].numRows);
        }
        else
            L.a[k] = L.a[0];
    }
    if (maxRows > 0)
    {
        /* a wave per 2 x 2 group of rows: the grid covers the lane with the most groups */
        uint32_t maxGroups = 0;
        for (uint32_t k = 0; k < count; k++)
            if (L.a[k].ch > 0)
                maxGroups = std::max(maxGroups, (L.a[k].ch + 1) / 2 * ((L.a[k].numRows / L.a[k].ch + 1) / 2));
        LAUNCH(ctx, "kernel.marching.genOccupied.time", cellCodeKernel, dim3(divUp(maxGroups, 4), count), dim3(256), L);
    }
    return MLSGPU_OK;
}

int mlsgpu_marching::computeCodes(const mlsgpu_swathe &sw)
{
    mlsgpu_marching *self = this;
    return computeCodesLanes(&self, &sw, 1);
}

/* shipOut, src/marching.cpp:553-625, sort-based: works on the unwelded buffers filled by generateElements */
int mlsgpu_marching::shipOutSorted(const uint32_t sizes[2], uint32_t zMax, mlsgpu_mesh *mesh)
{
    const uint32_t nv = sizes[0], ni = sizes[1];
    if (wideKeys)
        PROPAGATE(weld<uint64_t>(nv, zMax));
    else
        PROPAGATE(weld<uint32_t>(nv, zMax));
    if (ni > 0)
        LAUNCH(ctx, "kernel.marching.reindex.time", reindexKernel, dim3(divUp(ni, 256)), dim3(256),
               dIndices, (const uint32_t *) dIndexRemap, ni);
    HIP_CHECK(hipMemcpyAsync(&hReadback->numWelded, &dReadback->numWelded, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));        /* the reference's queue.finish(), :617 */
    mesh->numVertices = hReadback->numWelded;
    mesh->numInternalVertices = hReadback->firstExternal;
    return MLSGPU_OK;
}

/* one bucket of a batched ship-out */
struct ShipLane
{
    mlsgpu_marching *m;
    mlsgpu_swathe sw;
    uint32_t sizes[2];
    uint32_t zTop, zMax;
    mlsgpu_mesh mesh;           /* out: sizes filled in */
};

/* shipOut without a sort: the lattice weld over cells z in [zTop, zMax) of the resident swathe -- for every bucket of a
 * batch with ONE set of launches (blockIdx.y = bucket) and one read-back of all the welded counts */
static int shipOutLatticeLanes(ShipLane *lanes, uint32_t count)
{
    REQUIRE(count >= 1 && count <= MAX_LANES, MLSGPU_ERR_INVALID);
    mlsgpu_ctx *ctx = lanes[0].m->ctx;
    Lattice Ls[MAX_LANES];
    CodeView Cs[MAX_LANES];
    uint32_t numRows[MAX_LANES], cornerRows[MAX_LANES], cellRows[MAX_LANES];
    const U3 *firstRow[MAX_LANES];
    uint64_t keyOffsetL[MAX_LANES];
    for (uint32_t k = 0; k < count; k++)
    {
        mlsgpu_marching *m = lanes[k].m;
        const mlsgpu_swathe &sw = lanes[k].sw;
        const uint32_t zTop = lanes[k].zTop, zMax = lanes[k].zMax;
        const uint32_t W = sw.width, H = sw.height;
        Lattice &L = Ls[k];
        L.words = m->dLatWords;
        L.rowCounts = m->dLatRows;
        L.totals = &m->dReadback->classTotals;
        L.nw = (2 * W - 1 + 63) / 64;
        L.rowsPerLayer = 2 * H - 1;
        L.topx = 2 * (W - 1);
        L.topy = 2 * (H - 1);
        L.z2First = 2 * zTop;
        L.z2Last = 2 * zMax;
        L.cw = W - 1;
        L.ch = H - 1;
        numRows[k] = (2 * (zMax - zTop) + 1) * L.rowsPerLayer;
        REQUIRE(numRows[k] <= m->latRowsMax && L.nw <= m->latWords, MLSGPU_ERR_LENGTH);
        Cs[k] = m->codeView(sw);
        cornerRows[k] = (zMax - zTop + 1) * H;
        cellRows[k] = (zMax - zTop) * L.ch;
        firstRow[k] = m->dRowCounts + (uint64_t) (zTop - m->codeZ0) * L.ch;
        keyOffsetL[k] = ((uint64_t) m->keyOffset[2] << (2 * KEY_AXIS_BITS + 1))
            | ((uint64_t) m->keyOffset[1] << (KEY_AXIS_BITS + 1))
            | ((uint64_t) m->keyOffset[0] << 1);                                /* src/marching.cpp:594-597 */
    }
    auto lane = [&](uint32_t k) { return k < count ? k : 0u; };
    /* existence masks */
    {
        Lanes<LatticeMaskArgs> A;
        uint32_t most = 0;
        for (uint32_t j = 0; j < MAX_LANES; j++)
        {
            const uint32_t k = lane(j);
            A.a[j] = LatticeMaskArgs{Ls[k], Cs[k], (const U3 *) lanes[k].m->dRowCounts, lanes[k].zTop, lanes[k].zMax,
                                     lanes[k].sw.height, j < count ? cornerRows[k] : 0u};
            most = std::max(most, A.a[j].numCornerRows);
        }
        static const bool byRowsEnv = getenv("MLSGPU_HIP_LATTICE_MASK_ROWS") != nullptr && atoi(getenv("MLSGPU_HIP_LATTICE_MASK_ROWS")) != 0;
        bool byRows = byRowsEnv;
        for (uint32_t j = 0; j < count; j++)
            byRows = byRows || A.a[j].L.nw > 64;        /* (rows wider than 2 048 corners: more words than a wave has lanes) */
        if (byRows)
            LAUNCH(ctx, "kernel.marching.countUniqueVertices.time", latticeMaskKernel, dim3(divUp(most, 4), count), dim3(256), A,
                   (const uint64_t *) lanes[0].m->dEdgeLut);
        else
        {
            /* a thread per word: 64 / nw rows of corners per wave, four waves per workgroup */
            uint32_t groups = 1;
            for (uint32_t j = 0; j < count; j++)
                groups = std::max(groups, divUp(A.a[j].numCornerRows, 4 * (64u / A.a[j].L.nw)));
            LAUNCH(ctx, "kernel.marching.countUniqueVertices.time", latticeMaskWordKernel, dim3(groups, count), dim3(256), A,
                   (const uint16_t *) lanes[0].m->dEdgeLut16);
        }
    }
    typedef ScanJob<U3, ArrayIn<U3>, ArrayIn<U3>, ArrayOut<U3> > RowJob;
    {
        RowJob jobs[MAX_LANES];
        for (uint32_t k = 0; k < count; k++)
        {
            mlsgpu_marching *m = lanes[k].m;
            jobs[k] = RowJob{ArrayIn<U3>{m->dLatRows}, ArrayIn<U3>{m->dLatRows}, ArrayOut<U3>{m->dLatRows}, numRows[k],
                             U3{0, 0, 0}, m->dTileSums3, &m->dReadback->classTotals, nullptr};
        }
        PROPAGATE((exclusiveScanBatch<U3, ArrayIn<U3>, ArrayIn<U3>, ArrayOut<U3> >(ctx, "kernel.marching.scanUint.time", jobs, count)));
    }
    {
        Lanes<LatticePatchArgs> A;
        uint32_t most = 0;
        for (uint32_t j = 0; j < MAX_LANES; j++)
        {
            const uint32_t k = lane(j);
            A.a[j] = LatticePatchArgs{Ls[k], j < count ? numRows[k] * Ls[k].nw : 0u};
            most = std::max(most, A.a[j].numWords);
        }
        LAUNCH(ctx, "kernel.marching.scanUint.time", latticePatchKernel, dim3(divUp(most, 256), count), dim3(256), A);
    }
    {
        Lanes<LatticeVerticesArgs> A;
        uint32_t most = 0;
        for (uint32_t j = 0; j < MAX_LANES; j++)
        {
            const uint32_t k = lane(j);
            mlsgpu_marching *m = lanes[k].m;
            A.a[j] = LatticeVerticesArgs{Ls[k], m->view(lanes[k].sw), m->dWelded, m->dWeldedKeys, m->keyOffset[0], m->keyOffset[1],
                                         m->keyOffset[2], keyOffsetL[k], m->transform,
                                         j < count ? (numRows[k] / Ls[k].rowsPerLayer + 1) / 2 * ((Ls[k].rowsPerLayer + 1) / 2) : 0u};
            most = std::max(most, A.a[j].numPairs);
        }
        uint32_t enabled = 0;
        for (uint32_t k = 0; k < count; k++)
            enabled += lanes[k].m->transform.enabled ? 1u : 0u;
        if (enabled == count)
            LAUNCH(ctx, "kernel.marching.compactVertices.time", latticeVerticesKernel<1>, dim3(divUp(most, 4), count), dim3(256), A);
        else if (enabled == 0)
            LAUNCH(ctx, "kernel.marching.compactVertices.time", latticeVerticesKernel<0>, dim3(divUp(most, 4), count), dim3(256), A);
        else
            LAUNCH(ctx, "kernel.marching.compactVertices.time", latticeVerticesKernel<2>, dim3(divUp(most, 4), count), dim3(256), A);
    }
    /* first cell / index slot of every row of cells of the batch, the compacted cells, then the triangles */
    {
        RowJob jobs[MAX_LANES];
        for (uint32_t k = 0; k < count; k++)
        {
            mlsgpu_marching *m = lanes[k].m;
            jobs[k] = RowJob{ArrayIn<U3>{firstRow[k]}, ArrayIn<U3>{firstRow[k]}, ArrayOut<U3>{m->dRowStarts}, cellRows[k],
                             U3{0, 0, 0}, m->dTileSums3, &m->dReadback->batchTotals, nullptr};
        }
        PROPAGATE((exclusiveScanBatch<U3, ArrayIn<U3>, ArrayIn<U3>, ArrayOut<U3> >(ctx, "kernel.marching.scanElements.time", jobs, count)));
    }
    /* Two routes to the same index list, chosen per bucket.  By ROWS (one wave per row of cells, no compacted cell list): no
     * compactRowCells launch and 32 bytes per occupied cell less traffic, but a wave per row whatever it holds -- the
     * faster one when most cells are occupied (cfg3 noise cloud, 85 %: 4.2 -> 3.6 ms per step).  By CELLS (compact, then
     * one thread per occupied cell): time follows the surface, 0.70 against 1.31 ms per step on the shells cloud
     * (~6 % occupied).  MLSGPU_HIP_TRIANGLES_BY_CELLS=0/1 forces one. */
    {
        const char *const routeEnv = getenv("MLSGPU_HIP_TRIANGLES_BY_CELLS");     /* read per ship-out: tests flip it */
        /* rows when at least a quarter of the cells are occupied (the rows kernel walks every cell, the cells route pays a
         * compaction first): the slabs of the 1024^3 cloud, a third to a half occupied, 10.83 ms per step by cells and 10.60 by
         * rows; the shells cloud (under a sixth) 0.51 against 0.90 ms of emission */
        const uint32_t rowsFactor = 4;
        uint32_t byCells[MAX_LANES], byRows[MAX_LANES], nc = 0, nr = 0;
        for (uint32_t k = 0; k < count; k++)
        {
            const uint32_t cellsInBatch = lanes[k].m->bufferedCells;
            if (cellRows[k] == 0 || cellsInBatch == 0)
                continue;
            /* (the rows kernel stages 9 x nw lattice words per wave in LDS and divides by nw with a 16-bit reciprocal:
             * a lattice wider than 64 words -- 2047 cells -- takes the cells route whatever its density) */
            const bool cellsRoute = Ls[k].nw > 64
                                    || (routeEnv != nullptr ? routeEnv[0] != '0'
                                                            : (uint64_t) cellsInBatch * rowsFactor < (uint64_t) cellRows[k] * Ls[k].cw);
            if (cellsRoute) byCells[nc++] = k; else byRows[nr++] = k;
        }
        if (nc > 0)
        {
            Lanes<CompactRowCellsArgs> A;
            Lanes<LatticeTrianglesArgs> B;
            uint32_t mostRows = 0, mostCells = 0;
            for (uint32_t j = 0; j < MAX_LANES; j++)
            {
                const uint32_t k = byCells[j < nc ? j : 0];
                mlsgpu_marching *m = lanes[k].m;
                A.a[j] = CompactRowCellsArgs{(const uint8_t *) m->dCellCode, Ls[k].cw, Ls[k].ch, m->codeZ0, lanes[k].zTop,
                                             (const U3 *) m->dRowStarts, (const uchar2 *) m->dCount, m->dCells, m->dViStart,
                                             j < nc ? cellRows[k] : 0u};
                B.a[j] = LatticeTrianglesArgs{Ls[k], m->devTables(), (const uint2 *) m->dCells, (const uint2 *) m->dViStart,
                                              m->dIndices, (const U3 *) &m->dReadback->batchTotals};
                mostRows = std::max(mostRows, A.a[j].numRows);
                if (j < nc)
                    mostCells = std::max(mostCells, m->bufferedCells);
            }
            LAUNCH(ctx, "kernel.marching.scanElements.time", compactRowCellsKernel, dim3(divUp(mostRows, 4), nc), dim3(256), A);
            LAUNCH(ctx, "kernel.marching.generateElements.time", latticeTrianglesKernel, dim3(divUp(mostCells, 256), nc), dim3(256), B);
        }
        if (nr > 0)
        {
            Lanes<LatticeTrianglesRowArgs> A;
            uint32_t most = 0;
            for (uint32_t j = 0; j < MAX_LANES; j++)
            {
                const uint32_t k = byRows[j < nr ? j : 0];
                mlsgpu_marching *m = lanes[k].m;
                A.a[j] = LatticeTrianglesRowArgs{Ls[k], Cs[k], m->devTables(), firstRow[k], (const U3 *) m->dRowStarts,
                                                 lanes[k].zTop, m->dIndices, j < nr ? cellRows[k] : 0u,
                                                 (uint32_t *) (m->dReadback + 1)};
                most = std::max(most, A.a[j].numRows);
            }
            uint32_t nwMax = 0;
            for (uint32_t j = 0; j < nr; j++)
                nwMax = std::max(nwMax, A.a[j].L.nw);
            /* dynamic LDS: the nine lattice rows of each of the four waves' rows of cells */
            LAUNCH_LDS(ctx, "kernel.marching.generateElements.time", latticeTrianglesRowKernel, dim3(divUp(most, 4), nr), dim3(256),
                       4 * 9 * nwMax * 16, A);
        }
    }
    /* the welded counts of every bucket, one publication (classTotals and batchTotals are adjacent) */
    HostMailbox &box = lanes[0].m->box;
    const void *srcs[MAX_LANES];
    for (uint32_t k = 0; k < count; k++)
        srcs[k] = &lanes[k].m->dReadback->classTotals;
    PROPAGATE(box.publishGather(ctx->stream, srcs, count, 6));
    PROPAGATE(box.wait(ctx->stream));
    for (uint32_t k = 0; k < count; k++)
    {
        mlsgpu_marching *m = lanes[k].m;
        std::memcpy(&m->hReadback->classTotals, box.payload() + 6 * k, 2 * sizeof(U3));
        const U3 ct = m->hReadback->classTotals, bt = m->hReadback->batchTotals;
        if (bt.a != m->bufferedCells || bt.b != lanes[k].sizes[0] || bt.c != lanes[k].sizes[1])
            return setError(MLSGPU_ERR_INVALID, "lattice weld: batch accounting mismatch (%u/%u cells, %u/%u vertices, %u/%u indices)",
                            bt.a, m->bufferedCells, bt.b, lanes[k].sizes[0], bt.c, lanes[k].sizes[1]);
        mlsgpu_mesh &mesh = lanes[k].mesh;
        mesh.dVertices = m->dWelded;
        mesh.dTriangles = m->dIndices;
        mesh.dVertexKeys = m->dWeldedKeys;
        mesh.numTriangles = lanes[k].sizes[1] / 3;
        mesh.numVertices = (uint64_t) ct.a + ct.b + ct.c;
        mesh.numInternalVertices = ct.a;
    }
    return MLSGPU_OK;
}

/* the tail of shipOut (src/marching.cpp:619-624): counters, then the output functor */
int mlsgpu_marching::shipOutDone(const uint32_t sizes[2], const mlsgpu_mesh &mesh)
{
    bufferedCells = 0;
    counters[1]++;
    counters[4] += sizes[0];
    counters[5] += sizes[1];
    counters[6] += mesh.numVertices;
    counters[7] += mesh.numVertices - mesh.numInternalVertices;
    if (output != nullptr)
    {
        const int rc = output(outputUser, ctx->stream, &mesh);
        if (rc != 0)
            return setError(MLSGPU_ERR_CALLBACK, "output functor failed with %d", rc);
    }
    return MLSGPU_OK;
}

/* shipOut, src/marching.cpp:553-625 */
int mlsgpu_marching::shipOut(const mlsgpu_swathe &sw, const uint32_t sizes[2], uint32_t zTop, uint32_t zMax)
{
    if (direct)
    {
        ShipLane lane{this, sw, {sizes[0], sizes[1]}, zTop, zMax, mlsgpu_mesh()};
        PROPAGATE(shipOutLatticeLanes(&lane, 1));
        return shipOutDone(sizes, lane.mesh);
    }
    mlsgpu_mesh mesh;
    mesh.dVertices = dWelded;
    mesh.dTriangles = dIndices;
    mesh.dVertexKeys = dWeldedKeys;
    mesh.numTriangles = sizes[1] / 3;
    PROPAGATE(shipOutSorted(sizes, zMax, &mesh));
    return shipOutDone(sizes, mesh);
}

/* addSlices, src/marching.cpp:627-743 */
int mlsgpu_marching::addSlices(const mlsgpu_swathe &swathe, uint32_t offsets[2], uint32_t &zTop, uint32_t *shipOuts)
{
    uint32_t top[3] = {2 * (swathe.width - 1), 2 * (swathe.height - 1), 2 * zTop};
    U3 totals;
    PROPAGATE(generateCells(swathe, &totals));
    const uint32_t compacted = totals.a;
    if (compacted > 0)
    {
        uint32_t counts[2] = {totals.b, totals.c};
        if (counts[0] > vertexSpace || counts[1] > indexSpace)
        {
            counters[0]++;
            ctx->addValue("marching.overflow", 1.0);                /* overflowStat, src/marching.cpp:655 */
            /* Swathe is too big on its own: split it into maximal runs of slices using the
             * per-slice histogram and recurse (:652-701). */
            PROPAGATE(sliceHistogram(swathe));
            std::vector<uint2> hist(hHistogram + swathe.zFirst, hHistogram + swathe.zLast);
            auto H = [&](uint32_t z, int j) -> uint64_t { return j == 0 ? hist[z - swathe.zFirst].x : hist[z - swathe.zFirst].y; };
            uint32_t subFirst = swathe.zFirst;
            while (subFirst < swathe.zLast)
            {
                uint32_t subLast = subFirst;
                uint64_t c0 = 0, c1 = 0;
                while (subLast < swathe.zLast
                       && offsets[0] + c0 + H(subLast, 0) <= vertexSpace
                       && offsets[1] + c1 + H(subLast, 1) <= indexSpace)
                {
                    c0 += H(subLast, 0);
                    c1 += H(subLast, 1);
                    subLast++;
                }
                if (subFirst == subLast)
                {
                    while (subLast < swathe.zLast
                           && c0 + H(subLast, 0) <= vertexSpace
                           && c1 + H(subLast, 1) <= indexSpace)
                    {
                        c0 += H(subLast, 0);
                        c1 += H(subLast, 1);
                        subLast++;
                    }
                }
                if (subLast <= subFirst)
                    return setError(MLSGPU_ERR_LENGTH, "Marching: a single slice exceeds the mesh memory");
                mlsgpu_swathe sub = swathe;
                sub.zFirst = subFirst;
                sub.zLast = subLast;
                PROPAGATE(addSlices(sub, offsets, zTop, shipOuts));
                subFirst = subLast;
            }
        }
        else
        {
            if ((uint64_t) offsets[0] + counts[0] > vertexSpace || (uint64_t) offsets[1] + counts[1] > indexSpace)
            {
                /* fits, but only after flushing what is buffered (:705-719) */
                PROPAGATE(shipOut(swathe, offsets, zTop, swathe.zFirst));
                (*shipOuts)++;
                offsets[0] = offsets[1] = 0;
                zTop = swathe.zFirst;
                top[2] = 2 * swathe.zFirst;
            }
            if (direct)
            {
                /* lattice weld: nothing to generate now -- the field stays resident, the ship-out works from it */
                bufferedCells += compacted;
            }
            else
            {
                /* scanElements + generateElements (:721-731): compaction pass, then one thread per cell */
                const CellRange R{swathe.width - 1, swathe.height - 1, swathe.zFirst};
                const uint64_t n = (uint64_t) R.cw * R.ch * (swathe.zLast - swathe.zFirst);
                ClassifyIn in{codeView(swathe), R, dCount};
                CompactCellsOut outF{R, dCells, dViStart, offsets[0], offsets[1]};
                PROPAGATE((scanPhase2<U3, ClassifyIn, CompactCellsOut>(ctx, "kernel.marching.scanElements.time", in, outF, n,
                                                                       (const U3 *) dTileSums3)));
                const dim3 grid(divUp(compacted, 256)), block(256);
                if (wideKeys)
                    LAUNCH(ctx, "kernel.marching.generateElements.time", (generateElementsKernel<uint64_t>), grid, block,
                           dVertices, static_cast<uint64_t *>(dKeysA), dIndices, (const uint2 *) dViStart, (const uint2 *) dCells,
                           view(swathe), devTables(), keyOffset[0], keyOffset[1], keyOffset[2], top[0], top[1], top[2],
                           layout, compacted);
                else
                    LAUNCH(ctx, "kernel.marching.generateElements.time", (generateElementsKernel<uint32_t>), grid, block,
                           dVertices, static_cast<uint32_t *>(dKeysA), dIndices, (const uint2 *) dViStart, (const uint2 *) dCells,
                           view(swathe), devTables(), keyOffset[0], keyOffset[1], keyOffset[2], top[0], top[1], top[2],
                           layout, compacted);
            }
            offsets[0] += counts[0];
            offsets[1] += counts[1];
            counters[3] += compacted;
        }
    }
    counters[2] += compacted > 0;
    ctx->addValue("marching.slices.nonempty", compacted > 0 ? 1.0 : 0.0);      /* nonemptyStat, src/marching.cpp:739 */
    return MLSGPU_OK;
}

/* the argument checks and per-call state of Marching::generate (src/marching.cpp:745-786) */
static int beginGenerate(mlsgpu_marching *m, const mlsgpu_generator *generator, mlsgpu_output_fn output, void *outputUser,
                         const uint32_t size[3], const uint32_t keyOffset[3], mlsgpu_swathe *swathe)
{
    REQUIRE(m != nullptr && generator != nullptr && generator->enqueue != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(size != nullptr && keyOffset != nullptr, MLSGPU_ERR_INVALID);
    swathe->width = size[0];
    swathe->height = size[1];
    swathe->zStride = m->zStride;
    const uint32_t depth = size[2];
    REQUIRE(1u <= swathe->width && swathe->width <= m->maxWidth, MLSGPU_ERR_LENGTH);
    REQUIRE(1u <= swathe->height && swathe->height <= m->maxHeight, MLSGPU_ERR_LENGTH);
    REQUIRE(1u <= depth && depth <= m->maxDepth, MLSGPU_ERR_LENGTH);
    /* global coordinates must fit the 20.1 key fields (MAX_GLOBAL_DIMENSION, src/marching.h:145-150) */
    for (int i = 0; i < 3; i++)
        REQUIRE((uint64_t) keyOffset[i] + size[i] <= (1u << 20) - 1, MLSGPU_ERR_LENGTH);
    m->generator = generator;
    m->output = output;
    m->outputUser = outputUser;
    for (int i = 0; i < 3; i++)
        m->keyOffset[i] = keyOffset[i];
    /* the generate-time key layout: only as many bits as this bucket's doubled local coordinates need */
    m->layout.bx = bitsFor(2 * (swathe->width - 1));
    m->layout.by = bitsFor(2 * (swathe->height - 1));
    m->layout.bz = bitsFor(2 * (depth - 1));
    m->wideKeys = m->layout.bits() > 32;
    /* One swathe covers the bucket: weld from the resident field without sorting.  Otherwise (or when forced
     * with MLSGPU_HIP_WELD=sort) the reference's generate + sort + compact structure is used. */
    m->direct = depth <= m->maxSwathe;
    {
        const char *force = getenv("MLSGPU_HIP_WELD");
        if (force != nullptr && std::strcmp(force, "sort") == 0 && m->legacyBuffers)
            m->direct = false;
    }
    if (!m->direct && !m->legacyBuffers)
        return setError(MLSGPU_ERR_LENGTH, "Marching: depth %u needs several swathes but was created for one", depth);
    m->bufferedCells = 0;
    return MLSGPU_OK;
}

static int callGenerator(const mlsgpu_generator *generator, mlsgpu_marching *m, const mlsgpu_swathe *swathe)
{
    const int rc = generator->enqueue(generator->user, m->ctx->stream, m->dField, m->imageWidth, swathe);
    if (rc != 0)
        return rc > 0 && rc <= MLSGPU_ERR_CALLBACK ? rc : setError(MLSGPU_ERR_CALLBACK, "generator failed with %d", rc);
    return MLSGPU_OK;
}

/* Marching::generate, src/marching.cpp:745-824 */
MLSGPU_API int mlsgpu_hip_marching_generate(mlsgpu_marching *m, const mlsgpu_generator *generator,
                                            mlsgpu_output_fn output, void *outputUser,
                                            const uint32_t size[3], const uint32_t keyOffset[3])
{
    mlsgpu_swathe swathe;
    PROPAGATE(beginGenerate(m, generator, output, outputUser, size, keyOffset, &swathe));
    const uint32_t depth = size[2];
    mlsgpu_ctx *ctx = m->ctx;
    HIP_CHECK(hipSetDevice(ctx->device));

    uint32_t offsets[2] = {0, 0};
    uint32_t zTop = 0;
    uint32_t shipOuts = 0;
    const uint64_t shipOutsBefore = m->counters[1];
    for (uint32_t z = 0; z < depth; z += m->maxSwathe)
    {
        swathe.zFirst = z;
        swathe.zLast = std::min(depth, z + m->maxSwathe) - 1;
        swathe.zBias = (1 - (int32_t) z) * (int32_t) swathe.zStride;
        if (z != 0)
            PROPAGATE(mlsgpu_hip_marching_copy_slice(m, m->dField, m->imageWidth, m->maxSwathe, 0,
                                                     swathe.width, swathe.height, swathe.zStride));
        PROPAGATE(callGenerator(generator, m, &swathe));
        if (z > 0)
            swathe.zFirst--;
        PROPAGATE(m->computeCodes(swathe));
        PROPAGATE(m->addSlices(swathe, offsets, zTop, &shipOuts));
    }
    if (offsets[0] > 0)
    {
        PROPAGATE(m->shipOut(swathe, offsets, zTop, depth - 1));
        shipOuts++;
    }
    ctx->addValue("marching.shipouts", (double) (m->counters[1] - shipOutsBefore));     /* shipoutsStat, src/marching.cpp:822 */
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MLSGPU_OK;
}

/*
 * Marching::generate for the buckets of a batch (the SubItems of one WorkItem, which the reference's worker walks one by
 * one, src/workers.cpp:232-286), in lock-step: one Marching object per bucket (own field, lattice and mesh arena), every
 * kernel launched once for the whole batch, the swathe totals and the welded counts of all buckets read back together --
 * three host decisions per BATCH.  Per bucket the results are those of mlsgpu_hip_marching_generate, bit for bit: the same
 * kernels run on the same data, only the launch is shared.  Meshes are handed to `output` bucket by bucket, in order.
 *
 * The shared launches cover buckets that take the lattice weld (one swathe spans the bucket) and whose swathe fits the mesh
 * memory; a bucket that overflows is finished by the sequential path behind the batch (the reference's slice splitting),
 * and when any bucket needs several swathes the whole batch is processed one bucket at a time.
 */
namespace
{
struct BatchThunk
{
    mlsgpu_batch_output_fn fn;
    void *user;
    uint32_t index;
};
int batchThunkOutput(void *user, void *stream, const mlsgpu_mesh *mesh)
{
    const BatchThunk *t = static_cast<const BatchThunk *>(user);
    return t->fn != nullptr ? t->fn(t->user, t->index, stream, mesh) : 0;
}
} // namespace

extern "C" mlsgpu_mls *mlsgpu_hip_mls_of_generator(const mlsgpu_generator *gen);

MLSGPU_API int mlsgpu_hip_marching_generate_batch(mlsgpu_marching *const *ms, const mlsgpu_generator *generators, uint32_t count,
                                                  mlsgpu_batch_output_fn output, void *outputUser,
                                                  const uint32_t *sizes, const uint32_t *keyOffsets)
{
    REQUIRE(ms != nullptr && generators != nullptr && sizes != nullptr && keyOffsets != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(count >= 1 && count <= MLSGPU_MAX_BATCH, MLSGPU_ERR_LENGTH);
    BatchThunk thunks[MAX_LANES];
    mlsgpu_swathe sws[MAX_LANES];
    for (uint32_t k = 0; k < count; k++)
    {
        REQUIRE(ms[k] != nullptr && ms[k]->ctx == ms[0]->ctx, MLSGPU_ERR_INVALID);
        for (uint32_t j = 0; j < k; j++)
            REQUIRE(ms[j] != ms[k], MLSGPU_ERR_INVALID);
        thunks[k] = BatchThunk{output, outputUser, k};
    }
    mlsgpu_ctx *ctx = ms[0]->ctx;
    bool lockStep = count > 1;
    for (uint32_t k = 0; k < count; k++)
    {
        PROPAGATE(beginGenerate(ms[k], &generators[k], batchThunkOutput, &thunks[k], sizes + 3 * k, keyOffsets + 3 * k, &sws[k]));
        lockStep = lockStep && ms[k]->direct;
    }
    if (!lockStep)
    {
        for (uint32_t k = 0; k < count; k++)
            PROPAGATE(mlsgpu_hip_marching_generate(ms[k], &generators[k], batchThunkOutput, &thunks[k], sizes + 3 * k, keyOffsets + 3 * k));
        return MLSGPU_OK;
    }
    HIP_CHECK(hipSetDevice(ctx->device));
    /* every bucket is one swathe: slices [0, depth - 1], stored from row block 1 (src/marching.cpp:787-802) */
    mlsgpu_mls *functors[MAX_LANES];
    float *fields[MAX_LANES];
    uint64_t pitches[MAX_LANES];
    bool allMls = true;
    for (uint32_t k = 0; k < count; k++)
    {
        sws[k].zFirst = 0;
        sws[k].zLast = sizes[3 * k + 2] - 1;
        sws[k].zBias = (int32_t) sws[k].zStride;
        functors[k] = mlsgpu_hip_mls_of_generator(&generators[k]);
        allMls = allMls && functors[k] != nullptr;
        fields[k] = ms[k]->dField;
        pitches[k] = ms[k]->imageWidth;
    }
    if (allMls)
    {
        /* MlsFunctors: processCorners of the whole batch is one launch */
        const int rc = mlsgpu_hip_mls_enqueue_batch(functors, fields, pitches, nullptr, sws, count);
        if (rc != MLSGPU_OK)
            return rc;
    }
    else
        for (uint32_t k = 0; k < count; k++)
            PROPAGATE(callGenerator(&generators[k], ms[k], &sws[k]));
    PROPAGATE(computeCodesLanes(ms, sws, count));
    /* generateCells (src/marching.cpp:500-551): the swathe totals of every bucket, one read-back */
    {
        Lanes<RowTotalsArgs> rt;
        for (uint32_t k = 0; k < MAX_LANES; k++)
        {
            mlsgpu_marching *m = ms[k < count ? k : 0];
            const mlsgpu_swathe &sw = sws[k < count ? k : 0];
            const uint32_t cw = sw.width - 1, ch = sw.height - 1;
            const uint64_t rows = cw > 0 ? (uint64_t) ch * (sw.zLast - sw.zFirst) : 0;
            rt.a[k] = RowTotalsArgs{m->dRowCounts, rows, &m->dReadback->totals};
        }
        uint32_t *const gate = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(ms[0]->dReadback) + sizeof(Readback) + 64);
        const uint32_t seq = ms[0]->box.reserve();
        LAUNCH(ctx, "kernel.marching.genOccupied.time", rowTotalsKernel, dim3(count), dim3(1024), rt, count, gate, ms[0]->box.dev, seq);
        PROPAGATE(ms[0]->box.wait(ctx->stream));                 /* the reference's queue.finish(), :548 */
    }
    /* addSlices (src/marching.cpp:627-743) for a bucket that is one swathe with nothing buffered: it either fits the mesh
     * memory and waits for the ship-out at the end of the bucket, or it has to be split */
    ShipLane ship[MAX_LANES];
    uint32_t shipOf[MAX_LANES], numShip = 0;
    bool split[MAX_LANES];
    uint64_t shipOutsBefore[MAX_LANES];
    for (uint32_t k = 0; k < count; k++)
        shipOutsBefore[k] = ms[k]->counters[1];
    for (uint32_t k = 0; k < count; k++)
    {
        mlsgpu_marching *m = ms[k];
        std::memcpy(&m->hReadback->totals, ms[0]->box.payload() + 3 * k, sizeof(U3));
        const U3 totals = m->hReadback->totals;
        split[k] = false;
        shipOf[k] = MAX_LANES;
        if (totals.a == 0)
        {
            ctx->addValue("marching.slices.nonempty", 0.0);
            continue;
        }
        if (totals.b > m->vertexSpace || totals.c > m->indexSpace)
        {
            split[k] = true;            /* (addSlices below accounts for it) */
            continue;
        }
        m->bufferedCells += totals.a;
        m->counters[3] += totals.a;
        m->counters[2] += 1;
        ctx->addValue("marching.slices.nonempty", 1.0);
        shipOf[k] = numShip;
        ship[numShip++] = ShipLane{m, sws[k], {totals.b, totals.c}, 0u, sizes[3 * k + 2] - 1, mlsgpu_mesh()};
    }
    if (numShip > 0)
        PROPAGATE(shipOutLatticeLanes(ship, numShip));
    /* results bucket by bucket, in order; a bucket that has to be split goes through the sequential path here */
    for (uint32_t k = 0; k < count; k++)
    {
        mlsgpu_marching *m = ms[k];
        if (shipOf[k] != MAX_LANES)
            PROPAGATE(m->shipOutDone(ship[shipOf[k]].sizes, ship[shipOf[k]].mesh));
        else if (split[k])
        {
            uint32_t offsets[2] = {0, 0}, zTop = 0, shipOuts = 0;
            PROPAGATE(m->addSlices(sws[k], offsets, zTop, &shipOuts));
            if (offsets[0] > 0)
                PROPAGATE(m->shipOut(sws[k], offsets, zTop, sizes[3 * k + 2] - 1));
        }
        ctx->addValue("marching.shipouts", (double) (m->counters[1] - shipOutsBefore[k]));
    }
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_marching_set_vertex_transform(mlsgpu_marching *m, int enabled, float scale,
                                                        float bx, float by, float bz)
{
    REQUIRE(m != nullptr, MLSGPU_ERR_INVALID);
    m->transform = VertexTransform{scale, bx, by, bz, enabled ? 1 : 0};
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_marching_counters(const mlsgpu_marching *m, uint64_t out[8])
{
    REQUIRE(m != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    std::memcpy(out, m->counters, sizeof(m->counters));
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_marching_tables(const mlsgpu_marching *m, uint8_t *count, uint16_t *start, uint8_t *data, uint32_t *key)
{
    REQUIRE(m != nullptr, MLSGPU_ERR_INVALID);
    /* read back from the DEVICE copies so that the test sees what the kernels see */
    HIP_CHECK(hipSetDevice(m->ctx->device));
    HIP_CHECK(hipMemcpy(count, m->dCount, 512, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(start, m->dStart, 257 * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(data, m->dData, 8192, hipMemcpyDeviceToHost));
    std::vector<uint32_t> packed(2432);
    HIP_CHECK(hipMemcpy(packed.data(), m->dKey, 2432 * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < 2432; i++)
    {
        key[3 * i] = packed[i] & 0xFF;
        key[3 * i + 1] = (packed[i] >> 8) & 0xFF;
        key[3 * i + 2] = packed[i] >> 16;
    }
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_marching_copy_slice(mlsgpu_marching *m, float *dField, uint64_t pitch, uint32_t src, uint32_t trg,
                                              uint32_t width, uint32_t height, uint32_t zStride)
{
    REQUIRE(m != nullptr && dField != nullptr, MLSGPU_ERR_INVALID);
    if (width == 0 || height == 0)
        return MLSGPU_OK;
    LAUNCH(m->ctx, "kernel.marching.copySlice.time", copySliceKernel, dim3(divUp((uint64_t) width * height, 256)), dim3(256),
           dField, pitch, src * zStride, trg * zStride, width, height);
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_compact_vertices(mlsgpu_ctx *ctx, float *dOutVertices, uint64_t *dOutKeys, uint32_t *dIndexRemap,
                                           uint32_t *dFirstExternal, const uint32_t *dVertexUnique,
                                           const float *dInVertices4, const uint64_t *dInKeys,
                                           uint64_t minExternalKey, uint64_t keyOffset, uint64_t n)
{
    REQUIRE(ctx != nullptr, MLSGPU_ERR_INVALID);
    if (n == 0)
        return MLSGPU_OK;
    HIP_CHECK(hipSetDevice(ctx->device));
    LAUNCH(ctx, "kernel.marching.compactVertices.time", compactVerticesRefKernel, dim3(divUp(n, 64)), dim3(64),
           dOutVertices, dOutKeys, dIndexRemap, dFirstExternal, dVertexUnique,
           reinterpret_cast<const float4 *>(dInVertices4), dInKeys, minExternalKey, keyOffset, n);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_test_compute_key(mlsgpu_ctx *ctx, const uint32_t c[3], const uint32_t top[3], uint64_t *out)
{
    REQUIRE(ctx != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    HIP_CHECK(hipSetDevice(ctx->device));
    uint64_t *d = nullptr;
    HIP_CHECK(hipMalloc(&d, 8));
    hipLaunchKernelGGL(computeKeyTestKernel, dim3(1), dim3(1), 0, ctx->stream, c[0], c[1], c[2], top[0], top[1], top[2], d);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(out, d, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    hipFree(d);
    return MLSGPU_OK;
}

/* ------------------------------------------------------------------ mesh plumbing */

MLSGPU_API uint64_t mlsgpu_hip_mesh_host_bytes(const mlsgpu_mesh *mesh)
{
    /* MeshSizes::getHostBytes, src/mesh.h:75-80 */
    return 12 * mesh->numVertices + 12 * mesh->numTriangles + 8 * (mesh->numVertices - mesh->numInternalVertices);
}

MLSGPU_API int mlsgpu_hip_mesh_read(mlsgpu_ctx *ctx, const mlsgpu_mesh *mesh, void *hostBlob, int async)
{
    REQUIRE(ctx != nullptr && mesh != nullptr && hostBlob != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(mesh->numInternalVertices <= mesh->numVertices, MLSGPU_ERR_INVALID);             /* src/mesh.cpp:69 */
    REQUIRE(reinterpret_cast<uintptr_t>(hostBlob) % 8 == 0, MLSGPU_ERR_INVALID);             /* src/mesh.cpp:55 */
    HIP_CHECK(hipSetDevice(ctx->device));
    const uint64_t numExt = mesh->numVertices - mesh->numInternalVertices;
    /* HostKeyMesh layout, src/mesh.cpp:51-60 */
    uint64_t *hKeys = static_cast<uint64_t *>(hostBlob);
    float *hVerts = reinterpret_cast<float *>(hKeys + numExt);
    uint32_t *hTris = reinterpret_cast<uint32_t *>(hVerts + 3 * mesh->numVertices);
    int pend = -1;
    if (ctx->timing) pend = ctx->beginTiming(ctx->statId("device.read"));
    if (mesh->numTriangles)
        HIP_CHECK(hipMemcpyAsync(hTris, mesh->dTriangles, mesh->numTriangles * 12, hipMemcpyDeviceToHost, ctx->stream));
    if (numExt)
        HIP_CHECK(hipMemcpyAsync(hKeys, mesh->dVertexKeys + mesh->numInternalVertices, numExt * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (mesh->numVertices)
        HIP_CHECK(hipMemcpyAsync(hVerts, mesh->dVertices, mesh->numVertices * 12, hipMemcpyDeviceToHost, ctx->stream));
    if (pend >= 0) ctx->endTiming(pend);
    if (!async)
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MLSGPU_OK;
}

/* Measurement aid: sum over the 32-bit words w[i] of an array of w[i] * (2 i + 1), modulo 2^64 -- sensitive to
 * values and to their order; partial sums per wave, one atomic per wave. */
namespace
{
__global__ __launch_bounds__(256) void checksumKernel(const uint32_t *words, uint64_t n, unsigned long long *out)
{
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
        acc += (unsigned long long) words[i] * (2 * i + 1);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        acc += __shfl_xor(acc, d, 64);
    if (laneId() == 0 && acc != 0)
        atomicAdd(out, acc);
}
} // namespace

MLSGPU_API int mlsgpu_hip_mesh_checksum(mlsgpu_ctx *ctx, const mlsgpu_mesh *mesh, uint64_t out[3])
{
    REQUIRE(ctx != nullptr && mesh != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(mesh->numInternalVertices <= mesh->numVertices, MLSGPU_ERR_INVALID);
    HIP_CHECK(hipSetDevice(ctx->device));
    auto it = ctx->scratchCache.find("mesh.checksum");
    if (it == ctx->scratchCache.end())
    {
        void *d = nullptr;
        HIP_CHECK(hipMalloc(&d, 24));
        it = ctx->scratchCache.emplace("mesh.checksum", std::shared_ptr<void>(d, [](void *p) { hipFree(p); })).first;
    }
    unsigned long long *d = static_cast<unsigned long long *>(it->second.get());
    HIP_CHECK(hipMemsetAsync(d, 0, 24, ctx->stream));
    const uint64_t numExt = mesh->numVertices - mesh->numInternalVertices;
    const struct { const void *p; uint64_t words; } parts[3] = {
        {mesh->dVertices, 3 * mesh->numVertices}, {mesh->dTriangles, 3 * mesh->numTriangles},
        {mesh->dVertexKeys + mesh->numInternalVertices, 2 * numExt}};
    for (int k = 0; k < 3; k++)
        if (parts[k].words > 0)
            hipLaunchKernelGGL(checksumKernel, dim3((uint32_t) std::min<uint64_t>(divUp(parts[k].words, 256), 4096)), dim3(256), 0,
                               ctx->stream, static_cast<const uint32_t *>(parts[k].p), parts[k].words, d + k);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(out, d, 24, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_scale_bias(mlsgpu_ctx *ctx, const mlsgpu_mesh *mesh, float scale, float bx, float by, float bz)
{
    REQUIRE(ctx != nullptr && mesh != nullptr, MLSGPU_ERR_INVALID);
    if (mesh->numVertices == 0)
        return MLSGPU_OK;           /* empty mesh is legal, test/test_mesh_filter.cpp:340-361 */
    HIP_CHECK(hipSetDevice(ctx->device));
    const uint64_t n = mesh->numVertices * 3;
    LAUNCH(ctx, "kernel.scaleBias.time", scaleBiasKernel, dim3(divUp(n, 256)), dim3(256), mesh->dVertices, n, scale, bx, by, bz);
    return MLSGPU_OK;
}
