/*
 * Host mesh sink: the weld OOCMesher performs on the reference's single mesher thread (src/mesher.cpp:220-469),
 * kept in memory.  This is the north_star's "welding stays on host" route and the cross-GPU welder: ship-outs of any
 * device reach it through the bucket farm's pinned circular buffer (mlsgpu_hip_farm_set_host_output).
 *
 * add() does what OOCMesher::add does per block (:370-469): union-find over two edges per triangle gives the block's
 * local components ("clumps", computeLocalComponents :220-236, updateGlobalClumps :238-281), and every external key
 * that has been seen before merges the two clumps and un-counts the shared vertex (updateClumpKeyMap :286-311).
 * finalize() is the part of MesherBase::write that precedes the file output (:763-852): component sizes, the prune
 * threshold uint64(total * threshold) with the `>=` keep test (getStatistics :491-536), and one mesh per chunk in
 * which a key occurs once (externalRemap :538-567).  Differences from the reference, all invisible up to the
 * isomorphism its own tests compare by (test/test_mesher.cpp:401-460): no temporary files and no reorder buffer
 * (host memory is the arena), output order is (chunk by first arrival, block arrival, order inside the block) exactly
 * as the device sink's (mesher.hip), so the two sinks can be compared element for element.
 *
 * Host code only; it lives in the HIP library because the farm's mesher thread calls it.
 */
#include "common.hpp"

#include <algorithm>
#include <mutex>
#include <unordered_map>

using namespace mlsgpu;

namespace
{

struct Block
{
    uint32_t chunk;             /* dense chunk index */
    uint64_t vBase, nv, nInternal;
    uint64_t tBase, nt;
    uint64_t eBase;
};

struct Clump                    /* OOCMesher::Clump, src/mesher.h:~395: a union-find node with vertex / triangle counts */
{
    int64_t parent = -1;        /* -1: root */
    uint64_t vertices = 0, triangles = 0;
};

} // namespace

struct mlsgpu_host_mesher
{
    std::mutex mutex;
    double pruneThreshold = 0.0;
    std::vector<float> vertices;            /* 3 per vertex, arrival order */
    std::vector<uint32_t> triangles;        /* 3 per triangle, block-local indices */
    std::vector<uint64_t> extKeys;
    std::vector<uint32_t> clumpOf;          /* per vertex: its clump */
    std::vector<Block> blocks;
    std::vector<uint64_t> chunkIds;
    std::unordered_map<uint64_t, uint32_t> chunkIndex;
    std::vector<Clump> clumps;
    std::unordered_map<uint64_t, uint32_t> clumpIdMap;      /* key -> a clump that holds the vertex (clumpIdMap) */
    std::vector<int32_t> uf;                /* scratch: the block's union-find */
    bool finalized = false;

    std::vector<float> outVertices;
    std::vector<uint32_t> outTriangles;
    std::vector<uint64_t> chunkVStart, chunkTStart;
    std::vector<uint32_t> outChunks;
    uint64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    uint32_t clumpRoot(uint32_t c)
    {
        uint32_t r = c;
        while (clumps[r].parent >= 0)
            r = (uint32_t) clumps[r].parent;
        while (clumps[c].parent >= 0)       /* path compression */
        {
            const uint32_t next = (uint32_t) clumps[c].parent;
            clumps[c].parent = r;
            c = next;
        }
        return r;
    }
};

MLSGPU_API int mlsgpu_hip_host_mesher_create(mlsgpu_host_mesher **out)
{
    REQUIRE(out != nullptr, MLSGPU_ERR_INVALID);
    *out = new mlsgpu_host_mesher;
    return MLSGPU_OK;
}

MLSGPU_API void mlsgpu_hip_host_mesher_destroy(mlsgpu_host_mesher *m) { delete m; }

MLSGPU_API int mlsgpu_hip_host_mesher_set_prune_threshold(mlsgpu_host_mesher *m, double threshold)
{
    REQUIRE(m != nullptr && threshold >= 0.0 && threshold <= 1.0, MLSGPU_ERR_INVALID);
    m->pruneThreshold = threshold;
    return MLSGPU_OK;
}

static int32_t ufRoot(std::vector<int32_t> &uf, int32_t v)
{
    int32_t r = v;
    while (uf[r] >= 0)
        r = uf[r];
    while (uf[v] >= 0)
    {
        const int32_t next = uf[v];
        uf[v] = r;
        v = next;
    }
    return r;
}

MLSGPU_API int mlsgpu_hip_host_mesher_add(mlsgpu_host_mesher *m, uint64_t chunkId, const mlsgpu_host_mesh *mesh)
{
    REQUIRE(m != nullptr && mesh != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(mesh->numInternalVertices <= mesh->numVertices, MLSGPU_ERR_INVALID);
    REQUIRE(mesh->numVertices < (uint64_t(1) << 31), MLSGPU_ERR_LENGTH);
    const uint64_t nv = mesh->numVertices, nt = mesh->numTriangles, ni = mesh->numInternalVertices, ne = nv - ni;
    REQUIRE((nv == 0 || mesh->vertices != nullptr) && (nt == 0 || mesh->triangles != nullptr)
            && (ne == 0 || mesh->vertexKeys != nullptr), MLSGPU_ERR_INVALID);
    for (uint64_t t = 0; t < 3 * nt; t++)       /* before anything is appended: a bad mesh leaves the sink unchanged */
        REQUIRE(mesh->triangles[t] < nv, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    /* computeLocalComponents, src/mesher.cpp:220-236: union by size, negative value = -(size) at a root.  It only reads
     * the incoming mesh, so it runs -- with the "too many connected components" check (:252) -- before anything is appended:
     * a mesh that cannot be taken leaves the sink unchanged. */
    std::vector<int32_t> &uf = m->uf;
    uf.assign(nv, -1);
    for (uint64_t t = 0; t < nt; t++)
    {
        const uint32_t *tri = mesh->triangles + 3 * t;
        for (int e = 0; e < 2; e++)
        {
            int32_t a = ufRoot(uf, (int32_t) tri[e]), c = ufRoot(uf, (int32_t) tri[e + 1]);
            if (a == c)
                continue;
            if (uf[a] > uf[c])          /* a is the smaller tree */
                std::swap(a, c);
            uf[a] += uf[c];
            uf[c] = a;
        }
    }
    uint64_t roots = 0;
    for (uint64_t i = 0; i < nv; i++)
        roots += uf[i] < 0;
    REQUIRE(m->clumps.size() + roots < 0x7FFFFFFFu, MLSGPU_ERR_LENGTH);
    m->finalized = false;
    Block b;
    auto it = m->chunkIndex.find(chunkId);
    if (it == m->chunkIndex.end())
    {
        b.chunk = (uint32_t) m->chunkIds.size();
        m->chunkIndex.emplace(chunkId, b.chunk);
        m->chunkIds.push_back(chunkId);
    }
    else
        b.chunk = it->second;
    b.vBase = m->vertices.size() / 3;
    b.nv = nv;
    b.nInternal = ni;
    b.tBase = m->triangles.size() / 3;
    b.nt = nt;
    b.eBase = m->extKeys.size();
    m->vertices.insert(m->vertices.end(), mesh->vertices, mesh->vertices + 3 * nv);
    m->triangles.insert(m->triangles.end(), mesh->triangles, mesh->triangles + 3 * nt);
    m->extKeys.insert(m->extKeys.end(), mesh->vertexKeys, mesh->vertexKeys + ne);

    /* updateGlobalClumps, :238-281 */
    const uint64_t cBase = m->clumpOf.size();
    m->clumpOf.resize(cBase + nv);
    for (uint64_t i = 0; i < nv; i++)
        if (uf[i] < 0)
        {
            m->clumpOf[cBase + i] = (uint32_t) m->clumps.size();
            Clump c;
            c.vertices = (uint64_t) -(int64_t) uf[i];
            m->clumps.push_back(c);
        }
    for (uint64_t i = 0; i < nv; i++)
        if (uf[i] >= 0)
            m->clumpOf[cBase + i] = m->clumpOf[cBase + ufRoot(uf, (int32_t) i)];
    for (uint64_t t = 0; t < nt; t++)
        m->clumps[m->clumpOf[cBase + mesh->triangles[3 * t]]].triangles++;
    /* updateClumpKeyMap, :286-311 */
    for (uint64_t i = 0; i < ne; i++)
    {
        const uint32_t cid = m->clumpOf[cBase + ni + i];
        auto added = m->clumpIdMap.emplace(mesh->vertexKeys[i], cid);
        if (!added.second)
        {
            uint32_t a = m->clumpRoot(cid), c = m->clumpRoot(added.first->second);
            if (a != c)
            {
                if (m->clumps[a].vertices < m->clumps[c].vertices)
                    std::swap(a, c);
                m->clumps[c].parent = a;
                m->clumps[a].vertices += m->clumps[c].vertices;
                m->clumps[a].triangles += m->clumps[c].triangles;
            }
            m->clumps[a].vertices--;        /* both counted the common vertex */
        }
    }
    m->blocks.push_back(b);
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_host_mesher_farm_output(void *mesher, int device, uint64_t chunkId, const mlsgpu_host_mesh *mesh)
{
    (void) device;
    return mlsgpu_hip_host_mesher_add(static_cast<mlsgpu_host_mesher *>(mesher), chunkId, mesh);
}

/* keepClump == nullptr: the prune rule on this mesher's own counts (getStatistics, src/mesher.cpp:491-536);
 * otherwise keepClump[root clump] decides (the caller merged the clumps of several meshers, see
 * mlsgpu_hip_host_mesher_boundary) */
static int finalizeWith(mlsgpu_host_mesher *m, const uint8_t *keepClump, uint32_t *numChunks)
{
    const uint64_t nv = m->vertices.size() / 3, nt = m->triangles.size() / 3;
    const uint32_t nc = (uint32_t) m->chunkIds.size();
    uint64_t total = 0, components = 0;
    for (size_t c = 0; c < m->clumps.size(); c++)
        if (m->clumps[c].parent < 0)
        {
            total += m->clumps[c].vertices;
            components++;
        }
    const uint64_t threshold = (uint64_t) ((double) total * m->pruneThreshold);
    auto kept = [&](uint32_t root) { return keepClump ? keepClump[root] != 0 : m->clumps[root].vertices >= threshold; };
    uint64_t keptComponents = 0, keptVertices = 0, keptTriangles = 0;
    for (size_t c = 0; c < m->clumps.size(); c++)
        if (m->clumps[c].parent < 0 && kept((uint32_t) c))
        {
            keptComponents++;
            keptVertices += m->clumps[c].vertices;
            keptTriangles += m->clumps[c].triangles;
        }
    /* blocks in (chunk by first arrival, arrival) order; a key is emitted once per chunk (externalRemap, :538-567) */
    std::vector<uint32_t> order(m->blocks.size());
    for (uint32_t i = 0; i < order.size(); i++)
        order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return m->blocks[a].chunk < m->blocks[b].chunk; });
    m->outVertices.clear();
    m->outTriangles.clear();
    m->outVertices.reserve(3 * nv);
    m->outTriangles.reserve(3 * nt);
    m->chunkVStart.assign(nc + 1, 0);
    m->chunkTStart.assign(nc + 1, 0);
    m->outChunks.clear();
    std::vector<uint32_t> remap;
    std::unordered_map<uint64_t, uint32_t> chunkKeys;       /* key -> index inside the current chunk's output */
    size_t k = 0;
    for (uint32_t c = 0; c < nc; c++)
    {
        m->chunkVStart[c] = m->outVertices.size() / 3;
        m->chunkTStart[c] = m->outTriangles.size() / 3;
        chunkKeys.clear();
        for (; k < order.size() && m->blocks[order[k]].chunk == c; k++)
        {
            const Block &b = m->blocks[order[k]];
            remap.assign(b.nv, 0xFFFFFFFFu);
            const uint64_t first = m->chunkVStart[c];
            for (uint64_t i = 0; i < b.nv; i++)
            {
                const uint32_t root = m->clumpRoot(m->clumpOf[b.vBase + i]);
                if (!kept(root))
                    continue;
                if (i >= b.nInternal)
                {
                    auto added = chunkKeys.emplace(m->extKeys[b.eBase + (i - b.nInternal)], 0u);
                    if (!added.second)
                    {
                        remap[i] = added.first->second;
                        continue;
                    }
                    added.first->second = (uint32_t) (m->outVertices.size() / 3 - first);
                }
                remap[i] = (uint32_t) (m->outVertices.size() / 3 - first);
                const float *v = &m->vertices[3 * (b.vBase + i)];
                m->outVertices.insert(m->outVertices.end(), v, v + 3);
            }
            for (uint64_t t = 0; t < b.nt; t++)
            {
                const uint32_t *tri = &m->triangles[3 * (b.tBase + t)];
                if (remap[tri[0]] == 0xFFFFFFFFu)
                    continue;               /* the whole clump was pruned */
                for (int j = 0; j < 3; j++)
                    m->outTriangles.push_back(remap[tri[j]]);
            }
        }
    }
    m->chunkVStart[nc] = m->outVertices.size() / 3;
    m->chunkTStart[nc] = m->outTriangles.size() / 3;
    for (uint32_t c = 0; c < nc; c++)
        if (m->chunkTStart[c + 1] > m->chunkTStart[c])      /* no output for a chunk without triangles, :820 */
            m->outChunks.push_back(c);
    m->stats[0] = total;
    m->stats[1] = threshold;
    m->stats[2] = components;
    m->stats[3] = keptComponents;
    m->stats[4] = keptVertices;
    m->stats[5] = keptTriangles;
    m->stats[6] = nv;
    m->stats[7] = nt;
    m->finalized = true;
    if (numChunks)
        *numChunks = (uint32_t) m->outChunks.size();
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_host_mesher_finalize(mlsgpu_host_mesher *m, uint32_t *numChunks)
{
    REQUIRE(m != nullptr, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    return finalizeWith(m, nullptr, numChunks);
}

/* ---- several meshers, one job (one process per GPU: every rank welds its own buckets; components that cross rank
 *      boundaries and the prune threshold need the other ranks' clumps).  boundary() exports what the merge needs:
 *      every external key this mesher has seen with the ROOT clump that holds its vertex, and the vertex / triangle
 *      counts of every root clump.  The caller unites clumps that share a key across meshers (a vertex seen by r
 *      meshers was counted r times), applies the prune rule to the merged counts and hands the verdict back to
 *      finalize_with().  mlsgpu_amd/dist_sink.py does this over torch.distributed. ---- */
MLSGPU_API int mlsgpu_hip_host_mesher_boundary(mlsgpu_host_mesher *m, uint64_t *numKeys, uint64_t *numClumps)
{
    REQUIRE(m != nullptr && numKeys != nullptr && numClumps != nullptr, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    *numKeys = m->clumpIdMap.size();
    *numClumps = m->clumps.size();
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_host_mesher_boundary_read(mlsgpu_host_mesher *m, uint64_t *keys, uint32_t *keyClump,
                                                    uint64_t *clumpVertices, uint64_t *clumpTriangles)
{
    REQUIRE(m != nullptr && keys != nullptr && keyClump != nullptr && clumpVertices != nullptr && clumpTriangles != nullptr,
            MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    std::vector<std::pair<uint64_t, uint32_t> > sorted(m->clumpIdMap.begin(), m->clumpIdMap.end());
    std::sort(sorted.begin(), sorted.end());        /* by key: the same order on every rank, whatever the hash map did */
    for (size_t i = 0; i < sorted.size(); i++)
    {
        keys[i] = sorted[i].first;
        keyClump[i] = m->clumpRoot(sorted[i].second);
    }
    for (size_t c = 0; c < m->clumps.size(); c++)
    {
        const bool root = m->clumps[c].parent < 0;
        clumpVertices[c] = root ? m->clumps[c].vertices : 0;
        clumpTriangles[c] = root ? m->clumps[c].triangles : 0;
    }
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_host_mesher_finalize_with(mlsgpu_host_mesher *m, const uint8_t *keepClump, uint64_t numClumps,
                                                    uint32_t *numChunks)
{
    REQUIRE(m != nullptr && keepClump != nullptr, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    REQUIRE(numClumps == m->clumps.size(), MLSGPU_ERR_LENGTH);
    return finalizeWith(m, keepClump, numChunks);
}

MLSGPU_API int mlsgpu_hip_host_mesher_chunk(mlsgpu_host_mesher *m, uint32_t i, uint64_t *chunkId, uint64_t *numVertices,
                                            uint64_t *numTriangles, const float **vertices, const uint32_t **triangles)
{
    REQUIRE(m != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(m->finalized && i < m->outChunks.size(), MLSGPU_ERR_INVALID);
    const uint32_t c = m->outChunks[i];
    if (chunkId) *chunkId = m->chunkIds[c];
    if (numVertices) *numVertices = m->chunkVStart[c + 1] - m->chunkVStart[c];
    if (numTriangles) *numTriangles = m->chunkTStart[c + 1] - m->chunkTStart[c];
    if (vertices) *vertices = m->outVertices.data() + 3 * m->chunkVStart[c];
    if (triangles) *triangles = m->outTriangles.data() + 3 * m->chunkTStart[c];
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_host_mesher_stats(mlsgpu_host_mesher *m, uint64_t out[8])
{
    REQUIRE(m != nullptr && out != nullptr && m->finalized, MLSGPU_ERR_INVALID);
    std::copy(m->stats, m->stats + 8, out);
    return MLSGPU_OK;
}
