/*
 * Context, memory, statistics and the per-bucket worker of the HIP path.
 *
 * mlsgpu_worker restates DeviceWorkerGroupBase::Worker (src/workers.cpp:207-286): one stream, one
 * SplatTreeCL, one MlsFunctor, one Marching and a ScaleBiasFilter in front of the output functor.
 */
#include "common.hpp"

#include <algorithm>
#include <atomic>
#include <mutex>

using namespace mlsgpu;

namespace mlsgpu
{

static thread_local char lastError[512] = "";

int setError(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(lastError, sizeof(lastError), fmt, ap);
    va_end(ap);
    return code;
}

} // namespace mlsgpu

MLSGPU_API const char *mlsgpu_hip_last_error(void) { return mlsgpu::lastError; }

/* ------------------------------------------------------------------ context */

int mlsgpu_ctx::statId(const char *name)
{
    auto it = statIds.find(name);
    if (it != statIds.end())
        return it->second;
    const int id = (int) statNames.size();
    statNames.push_back(name);
    stats.push_back(Stat());
    statIds[name] = id;
    return id;
}

int mlsgpu_ctx::beginTiming(int id)
{
    PendingTiming p;
    p.nameId = id;
    hipEvent_t ev[2];
    for (int i = 0; i < 2; i++)
    {
        if (!eventPool.empty())
        {
            ev[i] = eventPool.back();
            eventPool.pop_back();
        }
        else if (hipEventCreate(&ev[i]) != hipSuccess)
            return -1;
    }
    p.start = ev[0];
    p.stop = ev[1];
    hipEventRecord(p.start, stream);
    pending.push_back(p);
    return (int) pending.size() - 1;
}

void mlsgpu_ctx::endTiming(int idx)
{
    hipEventRecord(pending[idx].stop, stream);
}

int mlsgpu_ctx::resolveTimings()
{
    HIP_CHECK(hipStreamSynchronize(stream));
    for (const PendingTiming &p : pending)
    {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess)
        {
            stats[p.nameId].totalMs += ms;
            stats[p.nameId].launches++;
        }
        eventPool.push_back(p.start);
        eventPool.push_back(p.stop);
    }
    pending.clear();
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_device_count(int *count)
{
    REQUIRE(count != nullptr, MLSGPU_ERR_INVALID);
    HIP_CHECK(hipGetDeviceCount(count));
    return MLSGPU_OK;
}

namespace
{
__global__ void mailboxPublishKernel(const uint32_t *src, uint32_t words, uint32_t *box, uint32_t seq)
{
    /* one wave: a word per lane, then lane 0 publishes behind a system-scope fence (the stores of one wave instruction are
     * ordered before it) */
    if (threadIdx.x < words)
        __hip_atomic_store(box + 1 + threadIdx.x, src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    if (threadIdx.x == 0)
        __hip_atomic_store(box, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void mailboxGatherKernel(Lanes<const uint32_t *> srcs, uint32_t count, uint32_t wordsEach, uint32_t *box, uint32_t seq)
{
    /* one wave: word t comes from source t / wordsEach */
    const uint32_t t = threadIdx.x;
    if (t < count * wordsEach)
        __hip_atomic_store(box + 1 + t, srcs.a[t / wordsEach][t % wordsEach], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    if (t == 0)
        __hip_atomic_store(box, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
} // namespace

int mlsgpu::HostMailbox::create()
{
    if (host != nullptr)
        return MLSGPU_OK;
    if (hipHostMalloc((void **) &host, (WORDS + 1) * 4, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
        return setError(MLSGPU_ERR_NOMEM, "cannot allocate the pinned read-back words");
    std::memset(host, 0, (WORDS + 1) * 4);
    if (hipHostGetDevicePointer((void **) &dev, host, 0) != hipSuccess)
    {
        hipHostFree(host);
        host = nullptr;
        return setError(MLSGPU_ERR_HIP, "cannot map the pinned read-back words");
    }
    seq = 0;
    return MLSGPU_OK;
}

void mlsgpu::HostMailbox::destroy()
{
    if (host != nullptr)
        hipHostFree(host);
    host = dev = nullptr;
}

int mlsgpu::HostMailbox::publish(hipStream_t stream, const void *src, uint32_t words)
{
    REQUIRE(host != nullptr && words <= WORDS, MLSGPU_ERR_INVALID);
    reserve();
    hipLaunchKernelGGL(mailboxPublishKernel, dim3(1), dim3(64), 0, stream, static_cast<const uint32_t *>(src), words, dev, seq);
    HIP_CHECK(hipGetLastError());
    return MLSGPU_OK;
}

int mlsgpu::HostMailbox::publishGather(hipStream_t stream, const void *const *srcs, uint32_t count, uint32_t wordsEach)
{
    REQUIRE(host != nullptr && count >= 1 && count <= MAX_LANES && count * wordsEach <= WORDS, MLSGPU_ERR_INVALID);
    Lanes<const uint32_t *> s;
    for (uint32_t k = 0; k < MAX_LANES; k++)
        s.a[k] = static_cast<const uint32_t *>(srcs[k < count ? k : 0]);
    reserve();
    hipLaunchKernelGGL(mailboxGatherKernel, dim3(1), dim3(64), 0, stream, s, count, wordsEach, dev, seq);
    HIP_CHECK(hipGetLastError());
    return MLSGPU_OK;
}

int mlsgpu::HostMailbox::wait(hipStream_t stream)
{
    /* poll; after a while make sure the stream is still healthy (a failed kernel would never publish) */
    uint64_t spins = 0;
    while (__atomic_load_n(host, __ATOMIC_ACQUIRE) != seq)
    {
        __builtin_ia32_pause();
        if (++spins == (1ull << 22))
        {
            spins = 0;
            HIP_CHECK(hipStreamSynchronize(stream));
            if (__atomic_load_n(host, __ATOMIC_ACQUIRE) != seq)
                return setError(MLSGPU_ERR_HIP, "a device read-back never arrived");
        }
    }
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_ctx_create(int device, void *stream, mlsgpu_ctx **out)
{
    REQUIRE(out != nullptr, MLSGPU_ERR_INVALID);
    int count = 0;
    HIP_CHECK(hipGetDeviceCount(&count));
    REQUIRE(device >= 0 && device < count, MLSGPU_ERR_INVALID);
    HIP_CHECK(hipSetDevice(device));
    static std::atomic<uint64_t> nextSerial{1};
    mlsgpu_ctx *ctx = new mlsgpu_ctx;
    ctx->device = device;
    ctx->serial = nextSerial++;
    if (stream != nullptr)
    {
        ctx->stream = static_cast<hipStream_t>(stream);
        ctx->ownStream = false;
    }
    else
    {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess)
        {
            delete ctx;
            return setError(MLSGPU_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
        }
        ctx->ownStream = true;
    }
    *out = ctx;
    return MLSGPU_OK;
}

int mlsgpu_ctx::scanFlags(uint32_t **flags, uint32_t *epoch, uint32_t **tickets, uint32_t *bases, uint32_t gridX, uint32_t count)
{
    const size_t flagWords = (size_t) MLSGPU_MAX_BATCH * 1024;      /* MAX_LANES x SCAN_ONEPASS_MAX_TILES */
    const size_t ticketWords = (size_t) MLSGPU_MAX_BATCH * 32;      /* MAX_LANES x SCAN_TICKET_STRIDE */
    if (dScanFlags == nullptr)
    {
        HIP_CHECK(hipSetDevice(device));
        HIP_CHECK(hipMalloc((void **) &dScanFlags, (flagWords + ticketWords) * sizeof(uint32_t)));
        HIP_CHECK(hipMemsetAsync(dScanFlags, 0, (flagWords + ticketWords) * sizeof(uint32_t), stream));
    }
    if (++scanEpoch == 0)
    {
        /* wrapped: a flag left by the launch of 2^32 launches ago would match again.  Fresh flags hold 0, which is never an epoch. */
        HIP_CHECK(hipMemsetAsync(dScanFlags, 0, flagWords * sizeof(uint32_t), stream));
        scanEpoch = 1;
    }
    *flags = dScanFlags;
    *epoch = scanEpoch;
    *tickets = dScanFlags + flagWords;
    for (uint32_t k = 0; k < MLSGPU_MAX_BATCH; k++)
    {
        bases[k] = scanTicketBase[k];
        if (k < count)
            scanTicketBase[k] += gridX;         /* every workgroup of the lane draws one (wraps with the counter) */
    }
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_ctx_release_scratch(mlsgpu_ctx *ctx)
{
    REQUIRE(ctx != nullptr, MLSGPU_ERR_INVALID);
    HIP_CHECK(hipSetDevice(ctx->device));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->scratchCache.clear();
    return MLSGPU_OK;
}

MLSGPU_API void mlsgpu_hip_ctx_destroy(mlsgpu_ctx *ctx)
{
    if (!ctx)
        return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    ctx->scratchCache.clear();
    hipFree(ctx->dScanFlags);
    for (const PendingTiming &p : ctx->pending)
    {
        hipEventDestroy(p.start);
        hipEventDestroy(p.stop);
    }
    for (hipEvent_t e : ctx->eventPool)
        hipEventDestroy(e);
    if (ctx->ownStream)
        hipStreamDestroy(ctx->stream);
    delete ctx;
}

MLSGPU_API void *mlsgpu_hip_ctx_stream(mlsgpu_ctx *ctx) { return ctx ? ctx->stream : nullptr; }

MLSGPU_API int mlsgpu_hip_ctx_synchronize(mlsgpu_ctx *ctx)
{
    REQUIRE(ctx != nullptr, MLSGPU_ERR_INVALID);
    HIP_CHECK(hipSetDevice(ctx->device));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_malloc(mlsgpu_ctx *ctx, size_t bytes, void **dptr)
{
    REQUIRE(ctx != nullptr && dptr != nullptr, MLSGPU_ERR_INVALID);
    HIP_CHECK(hipSetDevice(ctx->device));
    HIP_CHECK(hipMalloc(dptr, bytes ? bytes : 4));
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_free(mlsgpu_ctx *ctx, void *dptr)
{
    REQUIRE(ctx != nullptr, MLSGPU_ERR_INVALID);
    HIP_CHECK(hipSetDevice(ctx->device));
    HIP_CHECK(hipFree(dptr));
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_host_alloc(size_t bytes, void **hptr)
{
    REQUIRE(hptr != nullptr, MLSGPU_ERR_INVALID);
    HIP_CHECK(hipHostMalloc(hptr, bytes ? bytes : 8));
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_host_free(void *hptr)
{
    HIP_CHECK(hipHostFree(hptr));
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_memcpy_h2d(mlsgpu_ctx *ctx, void *dst, const void *src, size_t bytes, int async)
{
    REQUIRE(ctx != nullptr, MLSGPU_ERR_INVALID);
    if (bytes == 0)
        return MLSGPU_OK;
    HIP_CHECK(hipSetDevice(ctx->device));
    int pend = -1;
    if (ctx->timing) pend = ctx->beginTiming(ctx->statId("copy.write"));     /* src/workers.cpp:356-361 */
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (pend >= 0) ctx->endTiming(pend);
    if (!async)
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_memcpy_d2h(mlsgpu_ctx *ctx, void *dst, const void *src, size_t bytes, int async)
{
    REQUIRE(ctx != nullptr, MLSGPU_ERR_INVALID);
    if (bytes == 0)
        return MLSGPU_OK;
    HIP_CHECK(hipSetDevice(ctx->device));
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (!async)
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_memcpy_d2d(mlsgpu_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    REQUIRE(ctx != nullptr, MLSGPU_ERR_INVALID);
    if (bytes == 0)
        return MLSGPU_OK;
    HIP_CHECK(hipSetDevice(ctx->device));
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_memset(mlsgpu_ctx *ctx, void *dst, int value, size_t bytes)
{
    REQUIRE(ctx != nullptr, MLSGPU_ERR_INVALID);
    if (bytes == 0)
        return MLSGPU_OK;
    HIP_CHECK(hipSetDevice(ctx->device));
    HIP_CHECK(hipMemsetAsync(dst, value, bytes, ctx->stream));
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_ctx_set_timing(mlsgpu_ctx *ctx, int enabled)
{
    REQUIRE(ctx != nullptr, MLSGPU_ERR_INVALID);
    if (!enabled && ctx->timing)
        PROPAGATE(ctx->resolveTimings());
    ctx->timing = enabled != 0;
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_ctx_get_stat(mlsgpu_ctx *ctx, const char *name, double *totalMs, uint64_t *launches)
{
    REQUIRE(ctx != nullptr && name != nullptr, MLSGPU_ERR_INVALID);
    HIP_CHECK(hipSetDevice(ctx->device));
    PROPAGATE(ctx->resolveTimings());
    auto it = ctx->statIds.find(name);
    double ms = 0.0;
    uint64_t n = 0;
    if (it != ctx->statIds.end())
    {
        ms = ctx->stats[it->second].totalMs;
        n = ctx->stats[it->second].launches;
    }
    if (totalMs) *totalMs = ms;
    if (launches) *launches = n;
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_ctx_reset_stats(mlsgpu_ctx *ctx)
{
    REQUIRE(ctx != nullptr, MLSGPU_ERR_INVALID);
    HIP_CHECK(hipSetDevice(ctx->device));
    PROPAGATE(ctx->resolveTimings());
    for (Stat &s : ctx->stats)
        s = Stat();
    return MLSGPU_OK;
}

MLSGPU_API size_t mlsgpu_hip_ctx_dump_stats(mlsgpu_ctx *ctx, char *buf, size_t bufSize)
{
    if (!ctx)
        return 0;
    hipSetDevice(ctx->device);
    ctx->resolveTimings();
    std::string s;
    char line[256];
    for (size_t i = 0; i < ctx->statNames.size(); i++)
    {
        snprintf(line, sizeof(line), "%s %.6f %llu\n", ctx->statNames[i].c_str(), ctx->stats[i].totalMs,
                 (unsigned long long) ctx->stats[i].launches);
        s += line;
    }
    if (buf && bufSize)
    {
        const size_t n = std::min(bufSize - 1, s.size());
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return s.size() + 1;
}

/* ------------------------------------------------------------------ worker */

MLSGPU_API uint32_t mlsgpu_hip_compute_max_swathe(uint32_t yMax, uint32_t y, uint32_t yAlign, uint32_t zAlign)
{
    /* DeviceWorkerGroupBase::computeMaxSwathe, src/workers.cpp:169-182 */
    y = roundUp(y, yAlign);
    if (yMax < y)
        return zAlign;
    uint32_t chunks = (yMax - y) / (y * zAlign);
    if (chunks == 0)
        chunks = 1;
    return chunks * zAlign;
}

/* one bucket's worth of objects; a worker has one LANE per bucket it can take through the path in lock-step */
struct WorkerLane
{
    mlsgpu_tree *tree = nullptr;
    mlsgpu_mls *mls = nullptr;
    mlsgpu_marching *marching = nullptr;
};

struct mlsgpu_worker
{
    mlsgpu_ctx *ctx = nullptr;
    mlsgpu_worker_config cfg;
    std::vector<WorkerLane> lanes;          /* lanes[0] is the worker of the reference */
    bool keepSplats = false;
    uint32_t marchingGroup = 2;     /* buckets per processCorners / marching launch; 0: all the lanes (see set_marching_group) */
    mlsgpu_output_fn userOutput = nullptr;
    void *userOutputData = nullptr;
    mlsgpu_batch_output_fn batchOutput = nullptr;
    void *batchOutputData = nullptr;
    uint32_t batchBase = 0;                 /* index of the first bucket of the group that is being processed */
    uint32_t batchCompleted = 0;            /* leading buckets of the last batch call with all their meshes delivered */
};

static void resolveConfig(mlsgpu_worker_config &c)
{
    if (c.levels == 0) c.levels = 6;                    /* src/mlsgpu_core.cpp:110-111 */
    if (c.subsampling == 0) c.subsampling = 3;
    if (c.maxCells == 0) c.maxCells = (1u << (c.levels + c.subsampling - 1)) - 1;   /* :601-602 */
    if (c.meshMemory == 0)
        c.meshMemory = (uint64_t) c.maxCells * c.maxCells * 2 * MLSGPU_MARCHING_MAX_CELL_BYTES;   /* :359-370 */
    if (c.maxSwathe == 0)
        c.maxSwathe = roundUp(c.maxCells + 1, 8);       /* whole bucket: HBM buffers have no 8192-row limit */
    if (c.gridSpacing == 0.0f) c.gridSpacing = 1.0f;
}

MLSGPU_API uint64_t mlsgpu_hip_worker_resource_usage(const mlsgpu_worker_config *cfgIn)
{
    /* DeviceWorkerGroup::resourceUsage, src/workers.cpp:184-205 (per worker lane, without the item pool) */
    mlsgpu_worker_config c = *cfgIn;
    resolveConfig(c);
    const uint32_t wgs[3] = {8, 8, 8};
    const uint32_t block = c.maxCells + 1;
    return mlsgpu_hip_marching_resource_usage(block, block, roundUp(block, 8), c.maxSwathe, c.meshMemory, wgs)
        + mlsgpu_hip_tree_resource_usage(c.levels, c.maxBucketSplats);
}

MLSGPU_API uint64_t mlsgpu_hip_worker_resource_usage_lanes(const mlsgpu_worker_config *cfg, uint32_t lanes)
{
    return mlsgpu_hip_worker_resource_usage(cfg) * std::max(lanes, 1u);
}

static void destroyLane(WorkerLane &l)
{
    mlsgpu_hip_marching_destroy(l.marching);
    mlsgpu_hip_mls_destroy(l.mls);
    mlsgpu_hip_tree_destroy(l.tree);
    l = WorkerLane();
}

/* the objects DeviceWorkerGroupBase::Worker's constructor makes (src/workers.cpp:207-225) */
static int createLane(mlsgpu_worker *w, WorkerLane *out)
{
    const mlsgpu_worker_config &c = w->cfg;
    mlsgpu_ctx *ctx = w->ctx;
    WorkerLane l;
    const uint32_t wgs[3] = {8, 8, 8};
    const uint32_t block = c.maxCells + 1;
    int rc = mlsgpu_hip_tree_create(ctx, c.levels, c.maxBucketSplats, &l.tree);
    if (rc == MLSGPU_OK) rc = mlsgpu_hip_mls_create(ctx, c.shape, &l.mls);
    if (rc == MLSGPU_OK) rc = mlsgpu_hip_mls_set_boundary_limit(l.mls, c.boundaryLimit);
    /* depth padded to the MLS block size so that the default maxSwathe (whole bucket) is not rounded DOWN
     * to a multiple of 8 below the bucket depth (src/marching.cpp:373), which would cost a second swathe */
    if (rc == MLSGPU_OK) rc = mlsgpu_hip_marching_create(ctx, block, block, roundUp(block, 8), c.maxSwathe, c.meshMemory, wgs, &l.marching);
    if (rc == MLSGPU_OK)
        rc = mlsgpu_hip_marching_set_vertex_transform(l.marching, 1, c.gridSpacing, c.gridOrigin[0], c.gridOrigin[1], c.gridOrigin[2]);
    if (rc == MLSGPU_OK) rc = mlsgpu_hip_tree_set_mutate(l.tree, w->keepSplats ? 0 : 1);
    if (rc != MLSGPU_OK)
    {
        const std::string text = mlsgpu::lastError;     /* the destructors below do not touch it, but be explicit */
        destroyLane(l);
        return setError(rc, "%s", text.c_str());
    }
    *out = l;
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_worker_create(mlsgpu_ctx *ctx, const mlsgpu_worker_config *cfgIn, mlsgpu_worker **out)
{
    REQUIRE(ctx != nullptr && cfgIn != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    mlsgpu_worker *w = new mlsgpu_worker;
    w->ctx = ctx;
    w->cfg = *cfgIn;
    resolveConfig(w->cfg);
    const mlsgpu_worker_config &c = w->cfg;
    int rc = MLSGPU_OK;
    /* option constraints of src/mlsgpu_core.cpp:411-442 */
    if (!(c.subsampling >= 3 && c.levels >= 1 && c.levels <= 10 && c.subsampling + c.levels <= 14))
        rc = setError(MLSGPU_ERR_INVALID, "worker: need subsampling >= 3, 1 <= levels <= 10, subsampling + levels <= 14");
    if (rc == MLSGPU_OK && c.maxCells + 1 > (1u << (c.levels + c.subsampling - 1)))
        rc = setError(MLSGPU_ERR_LENGTH, "worker: maxCells + 1 exceeds the octree side 2^(levels+subsampling-1)");
    WorkerLane lane;
    if (rc == MLSGPU_OK) rc = createLane(w, &lane);
    if (rc != MLSGPU_OK)
    {
        mlsgpu_hip_worker_destroy(w);
        return rc;
    }
    w->lanes.push_back(lane);
    *out = w;
    return MLSGPU_OK;
}

MLSGPU_API void mlsgpu_hip_worker_destroy(mlsgpu_worker *w)
{
    if (!w)
        return;
    for (WorkerLane &l : w->lanes)
        destroyLane(l);
    delete w;
}

/* Room for `lanes` buckets in lock-step (mlsgpu_hip_worker_process_batch): every lane has its own tree, field, lattice and
 * mesh arena -- mlsgpu_hip_worker_resource_usage bytes each; 288 GB of HBM hold many.  Lanes are only added. */
MLSGPU_API int mlsgpu_hip_worker_set_batch(mlsgpu_worker *w, uint32_t lanes)
{
    REQUIRE(w != nullptr && !w->lanes.empty(), MLSGPU_ERR_INVALID);
    REQUIRE(lanes >= 1 && lanes <= MLSGPU_MAX_BATCH, MLSGPU_ERR_LENGTH);
    HIP_CHECK(hipSetDevice(w->ctx->device));
    const size_t before = w->lanes.size();
    while (w->lanes.size() < lanes)
    {
        WorkerLane l;
        const int rc = createLane(w, &l);
        if (rc != MLSGPU_OK)
        {
            /* all or nothing: a caller that falls back to fewer lanes (the farm's workers do, on MLSGPU_ERR_NOMEM) gets the
             * memory of the lanes this call managed to create back, and no stale HIP error is left pending */
            const std::string text = mlsgpu::lastError;
            while (w->lanes.size() > before)
            {
                destroyLane(w->lanes.back());
                w->lanes.pop_back();
            }
            (void) hipGetLastError();
            return setError(rc, "%s", text.c_str());
        }
        w->lanes.push_back(l);
    }
    return MLSGPU_OK;
}

MLSGPU_API uint32_t mlsgpu_hip_worker_batch(const mlsgpu_worker *w) { return w ? (uint32_t) w->lanes.size() : 0; }

/* How many of the lanes' buckets share one set of processCorners / marching launches (0: all of them).  The octree build
 * always takes all lanes together: its kernels are mid-sized (tens of microseconds per bucket) and gain from every bucket
 * added to a launch, while the large kernels behind it lose the Infinity Cache between producer and consumer when more than
 * two buckets' fields and lattices are in flight (profiles/NOTES_r04.md).  Default 2. */
MLSGPU_API int mlsgpu_hip_worker_set_marching_group(mlsgpu_worker *w, uint32_t buckets)
{
    REQUIRE(w != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(buckets <= MLSGPU_MAX_BATCH, MLSGPU_ERR_LENGTH);
    w->marchingGroup = buckets;
    return MLSGPU_OK;
}

MLSGPU_API uint32_t mlsgpu_hip_worker_marching_group(const mlsgpu_worker *w) { return w ? w->marchingGroup : 0; }

/* MeshFilterChain::operator() with the one ScaleBiasFilter the worker installs
 * (src/workers.cpp:226-230, src/mesh_filter.cpp:45-66): filter, then the user's output functor.
 * (Scale / bias is folded into Marching's vertex emission, mlsgpu_hip_marching_set_vertex_transform.) */
static int workerOutput(void *user, void *stream, const mlsgpu_mesh *mesh)
{
    mlsgpu_worker *w = static_cast<mlsgpu_worker *>(user);
    if (w->userOutput)
        return w->userOutput(w->userOutputData, stream, mesh);
    return 0;
}

static int workerBatchOutput(void *user, uint32_t index, void *stream, const mlsgpu_mesh *mesh)
{
    mlsgpu_worker *w = static_cast<mlsgpu_worker *>(user);
    if (w->batchOutput)
        return w->batchOutput(w->batchOutputData, w->batchBase + index, stream, mesh);
    return 0;
}

MLSGPU_API int mlsgpu_hip_worker_process(mlsgpu_worker *w, mlsgpu_splat *dSplats, uint64_t firstSplat, uint64_t numSplats,
                                         const int32_t lowExtent[3], const uint32_t numVertices[3],
                                         mlsgpu_output_fn output, void *outputUser)
{
    REQUIRE(w != nullptr && dSplats != nullptr && lowExtent != nullptr && numVertices != nullptr, MLSGPU_ERR_INVALID);
    /* src/workers.cpp:237-261 */
    uint32_t keyOffset[3], size[3], expanded[3];
    for (int i = 0; i < 3; i++)
    {
        REQUIRE(lowExtent[i] >= 0, MLSGPU_ERR_INVALID);     /* keyOffset is cl_uint in the reference */
        keyOffset[i] = (uint32_t) lowExtent[i];
        size[i] = numVertices[i];
        expanded[i] = roundUp(size[i], 8);
    }
    WorkerLane &l = w->lanes[0];
    w->userOutput = output;
    w->userOutputData = outputUser;
    int pend = -1;
    if (w->ctx->timing) pend = w->ctx->beginTiming(w->ctx->statId("device.compute"));
    /* a failing step skips the rest, never the end of the timing region or the release of the borrowed splats */
    mlsgpu_generator gen;
    int rc = mlsgpu_hip_tree_build(l.tree, dSplats, firstSplat, numSplats, expanded, lowExtent, w->cfg.subsampling);
    if (rc == MLSGPU_OK) rc = mlsgpu_hip_mls_set(l.mls, lowExtent, l.tree, w->cfg.subsampling);
    if (rc == MLSGPU_OK) rc = mlsgpu_hip_mls_generator(l.mls, &gen);
    if (rc == MLSGPU_OK) rc = mlsgpu_hip_marching_generate(l.marching, &gen, workerOutput, w, size, keyOffset);
    if (pend >= 0) w->ctx->endTiming(pend);
    mlsgpu_hip_tree_clear_splats(l.tree);
    return rc;
}

/*
 * The SubItems of a WorkItem (src/workers.h:148-181; the reference's worker walks them one by one, src/workers.cpp:232-286)
 * through the path in groups of as many buckets as the worker has lanes (mlsgpu_hip_worker_set_batch): per group ONE set of
 * launches -- octree build, processCorners, marching -- with a bucket dimension, and three host decisions instead of three
 * per bucket.  Every bucket's meshes are those of mlsgpu_hip_worker_process, bit for bit; `output` gets them bucket by
 * bucket, in order, with the bucket's index.
 */
MLSGPU_API int mlsgpu_hip_worker_process_batch(mlsgpu_worker *w, mlsgpu_splat *dSplats, const mlsgpu_subitem *items,
                                               uint32_t numItems, mlsgpu_batch_output_fn output, void *outputUser)
{
    REQUIRE(w != nullptr && (items != nullptr || numItems == 0), MLSGPU_ERR_INVALID);
    for (uint32_t i = 0; i < numItems; i++)
    {
        REQUIRE(dSplats != nullptr || items[i].dSplats != nullptr, MLSGPU_ERR_INVALID);
        for (int a = 0; a < 3; a++)
            REQUIRE(items[i].lowExtent[a] >= 0, MLSGPU_ERR_INVALID);     /* keyOffset is cl_uint in the reference */
    }
    w->batchOutput = output;
    w->batchOutputData = outputUser;
    w->batchCompleted = 0;
    const uint32_t width = (uint32_t) w->lanes.size();
    int rc = MLSGPU_OK;
    for (uint32_t base = 0; base < numItems && rc == MLSGPU_OK; base += width)
    {
        const uint32_t count = std::min(width, numItems - base);
        w->batchBase = base;
        mlsgpu_tree *trees[MLSGPU_MAX_BATCH];
        mlsgpu_marching *marchings[MLSGPU_MAX_BATCH];
        mlsgpu_tree_build builds[MLSGPU_MAX_BATCH];
        mlsgpu_generator gens[MLSGPU_MAX_BATCH];
        uint32_t sizes[3 * MLSGPU_MAX_BATCH], keyOffsets[3 * MLSGPU_MAX_BATCH];
        for (uint32_t k = 0; k < count; k++)
        {
            const mlsgpu_subitem &it = items[base + k];
            WorkerLane &l = w->lanes[k];
            trees[k] = l.tree;
            marchings[k] = l.marching;
            builds[k].dSplats = it.dSplats != nullptr ? it.dSplats : dSplats;
            builds[k].firstSplat = it.firstSplat;
            builds[k].numSplats = it.numSplats;
            for (int a = 0; a < 3; a++)
            {
                /* src/workers.cpp:237-261 */
                keyOffsets[3 * k + a] = (uint32_t) it.lowExtent[a];
                sizes[3 * k + a] = it.numVertices[a];
                builds[k].size[a] = roundUp(it.numVertices[a], 8);
                builds[k].offset[a] = it.lowExtent[a];
            }
            if (k > 0 && rc == MLSGPU_OK)
                rc = mlsgpu_hip_mls_copy_settings(l.mls, w->lanes[0].mls);
        }
        /* a failing step skips the rest of the batch, never the end of the timing region or the release of the borrowed
         * splats; batchCompleted tells the caller how many leading buckets had delivered all their meshes by then */
        int pend = -1;
        if (w->ctx->timing) pend = w->ctx->beginTiming(w->ctx->statId("device.compute"));
        if (rc == MLSGPU_OK)
            rc = mlsgpu_hip_tree_build_batch(trees, builds, count, w->cfg.subsampling);
        for (uint32_t k = 0; k < count && rc == MLSGPU_OK; k++)
        {
            rc = mlsgpu_hip_mls_set(w->lanes[k].mls, items[base + k].lowExtent, w->lanes[k].tree, w->cfg.subsampling);
            if (rc == MLSGPU_OK)
                rc = mlsgpu_hip_mls_generator(w->lanes[k].mls, &gens[k]);
        }
        /* the octree of all `count` buckets in one set of launches, processCorners and marching a GROUP of buckets at a time
         * (mlsgpu_hip_worker_set_marching_group): as many consecutive buckets as hold the corners of `marchingGroup` buckets
         * of the worker's full size -- two full buckets, or the eight small ones of a partition that cost what two do */
        const uint64_t side = (uint64_t) w->cfg.maxCells + 1;
        const uint64_t budget = w->marchingGroup != 0 ? (uint64_t) w->marchingGroup * side * side * side : UINT64_MAX;
        for (uint32_t k0 = 0, n = 0; k0 < count && rc == MLSGPU_OK; k0 += n)
        {
            uint64_t corners = 0;
            for (n = 0; k0 + n < count; n++)
            {
                const uint32_t *nv = sizes + 3 * (k0 + n);
                corners += (uint64_t) nv[0] * nv[1] * nv[2];
                if (n > 0 && corners > budget)
                    break;
            }
            w->batchBase = base + k0;
            rc = mlsgpu_hip_marching_generate_batch(marchings + k0, gens + k0, n, workerBatchOutput, w, sizes + 3 * k0,
                                                    keyOffsets + 3 * k0);
            if (rc == MLSGPU_OK)
                w->batchCompleted = base + k0 + n;
        }
        if (pend >= 0) w->ctx->endTiming(pend);
        for (uint32_t k = 0; k < count; k++)
            mlsgpu_hip_tree_clear_splats(w->lanes[k].tree);
    }
    return rc;
}

/* Of the last mlsgpu_hip_worker_process_batch call's items, how many leading ones had delivered all their meshes when it
 * returned (numItems after a success): what a caller that accounts per bucket needs after a failure in mid-batch. */
MLSGPU_API uint32_t mlsgpu_hip_worker_batch_completed(const mlsgpu_worker *w) { return w ? w->batchCompleted : 0; }

/* DeviceWorkerGroup's item buffer is rewritten by every H2D copy, so nothing downstream reads the splats the build has
 * mutated; a caller whose splats are RESIDENT (several passes over the same buffer) keeps them intact with this: the tree
 * is built without the in-place radius -> 1/radius^2 and processCorners takes the reciprocal while staging.  Same field,
 * bit for bit. */
MLSGPU_API int mlsgpu_hip_worker_set_keep_splats(mlsgpu_worker *w, int keep)
{
    REQUIRE(w != nullptr, MLSGPU_ERR_INVALID);
    w->keepSplats = keep != 0;
    for (WorkerLane &l : w->lanes)
        PROPAGATE(mlsgpu_hip_tree_set_mutate(l.tree, keep ? 0 : 1));
    return MLSGPU_OK;
}

/* the objects of lane 0: the reference's worker */
MLSGPU_API mlsgpu_tree *mlsgpu_hip_worker_tree(mlsgpu_worker *w) { return w ? w->lanes[0].tree : nullptr; }
MLSGPU_API mlsgpu_mls *mlsgpu_hip_worker_mls(mlsgpu_worker *w) { return w ? w->lanes[0].mls : nullptr; }
MLSGPU_API mlsgpu_marching *mlsgpu_hip_worker_marching(mlsgpu_worker *w) { return w ? w->lanes[0].marching : nullptr; }
MLSGPU_API mlsgpu_tree *mlsgpu_hip_worker_lane_tree(mlsgpu_worker *w, uint32_t lane)
{
    return w && lane < w->lanes.size() ? w->lanes[lane].tree : nullptr;
}
MLSGPU_API mlsgpu_marching *mlsgpu_hip_worker_lane_marching(mlsgpu_worker *w, uint32_t lane)
{
    return w && lane < w->lanes.size() ? w->lanes[lane].marching : nullptr;
}
