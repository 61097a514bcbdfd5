/*
 * Bucket farm: the reference's CopyGroup + per-device DeviceWorkerGroup (src/workers.{h,cpp}) re-expressed
 * with std::thread, HIP streams and events.  See include/mlsgpu_hip.h for the contract.
 *
 * Threads: the caller's thread does what the reference's "copy" thread does (stage, pick a device, enqueue
 * the H2D copy); `workersPerDevice` threads per GPU do what "device.N" threads do.  Ordering between the copy
 * stream and a worker's stream is a hipEvent (WorkItem::copyEvent in the reference).
 */
#include "common.hpp"

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

using namespace mlsgpu;

namespace
{

struct SubItem                   /* DeviceWorkerGroupBase::SubItem, src/workers.h:148-160 */
{
    uint64_t chunkId;
    int32_t low[3];
    uint32_t numVertices[3];
    uint64_t firstSplat, numSplats;
};

struct WorkItem                  /* DeviceWorkerGroupBase::WorkItem, src/workers.h:165-181 */
{
    std::vector<SubItem> subItems;
    mlsgpu_splat *dSplats = nullptr;
    hipEvent_t copyEvent = nullptr;
    uint64_t numSplats = 0;
};

struct DeviceGroup;

struct Farm;

struct DeviceGroup               /* DeviceWorkerGroup, src/workers.h:214-350 */
{
    Farm *farm = nullptr;
    int device = 0;
    uint32_t index = 0;
    hipStream_t copyStream = nullptr;
    mlsgpu_ctx *copyCtx = nullptr;       /* the copy stream as a context, for device-side loads */
    std::vector<std::unique_ptr<WorkItem> > items;
    std::deque<WorkItem *> pool;         /* itemPool */
    std::deque<WorkItem *> queue;        /* pushed, not yet taken by a worker */
    uint64_t unallocated = 0;
    uint64_t bucketsDone = 0;
    std::vector<std::thread> threads;
};

struct Farm
{
    mlsgpu_farm_config cfg;
    mlsgpu_farm_output_fn output = nullptr;
    void *user = nullptr;
    std::vector<std::unique_ptr<DeviceGroup> > groups;
    std::mutex mutex;                    /* guards pools, queues, counters, error */
    std::condition_variable popCond;     /* an item went back to some pool (popCondition in the reference) */
    std::condition_variable queueCond;   /* an item was pushed / stopping */
    std::condition_variable idleCond;    /* a bucket finished */
    bool stopping = false;
    int error = MLSGPU_OK;
    std::string errorText;
    uint64_t inFlightItems = 0;

    /* staging (CopyGroupBase::Worker: pinned + bufferedItems + bufferedSplats) */
    mlsgpu_splat *pinned[2] = {nullptr, nullptr};
    hipEvent_t pinnedFree[2] = {nullptr, nullptr};   /* the copy out of this buffer has completed */
    bool pinnedBusy[2] = {false, false};
    int cur = 0;
    std::vector<SubItem> bufferedItems;
    uint64_t bufferedSplats = 0;
    uint64_t acquired = 0;          /* splats handed out by acquire and not pushed yet */

    uint64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    void fail(int code, const char *text)
    {
        std::lock_guard<std::mutex> l(mutex);
        if (error == MLSGPU_OK)
        {
            error = code;
            errorText = text;
        }
        stopping = true;
        queueCond.notify_all();
        popCond.notify_all();
        idleCond.notify_all();
    }
};

} // namespace

struct mlsgpu_farm : Farm {};

namespace
{

struct OutputThunk
{
    Farm *farm;
    DeviceGroup *group;
    mlsgpu_ctx *ctx;
    uint64_t chunkId;
};

int outputThunk(void *user, void *stream, const mlsgpu_mesh *mesh)
{
    (void) stream;
    OutputThunk *t = static_cast<OutputThunk *>(user);
    {
        std::lock_guard<std::mutex> l(t->farm->mutex);
        t->farm->stats[4]++;
        t->farm->stats[5] += mesh->numVertices;
        t->farm->stats[6] += mesh->numTriangles;
        t->farm->stats[7] += mesh->numVertices - mesh->numInternalVertices;
    }
    if (t->farm->output != nullptr)
        return t->farm->output(t->farm->user, t->group->device, t->chunkId, t->ctx, mesh);
    return 0;
}

/* DeviceWorkerGroupBase::Worker::operator(), src/workers.cpp:232-286 */
void workerMain(Farm *farm, DeviceGroup *g)
{
    mlsgpu_ctx *ctx = nullptr;
    mlsgpu_worker *worker = nullptr;
    int rc = mlsgpu_hip_ctx_create(g->device, nullptr, &ctx);
    if (rc == MLSGPU_OK)
        rc = mlsgpu_hip_worker_create(ctx, &farm->cfg.worker, &worker);
    if (rc != MLSGPU_OK)
    {
        farm->fail(rc, mlsgpu_hip_last_error());
        if (ctx) mlsgpu_hip_ctx_destroy(ctx);
        return;
    }
    for (;;)
    {
        WorkItem *item = nullptr;
        {
            std::unique_lock<std::mutex> l(farm->mutex);
            farm->queueCond.wait(l, [&] { return farm->stopping || !g->queue.empty(); });
            if (g->queue.empty())
                break;
            item = g->queue.front();
            g->queue.pop_front();
        }
        /* wait[0] = work.copyEvent (src/workers.cpp:268) */
        hipError_t e = hipStreamWaitEvent(static_cast<hipStream_t>(mlsgpu_hip_ctx_stream(ctx)), item->copyEvent, 0);
        if (e != hipSuccess)
            farm->fail(MLSGPU_ERR_HIP, hipGetErrorString(e));
        for (size_t i = 0; i < item->subItems.size() && e == hipSuccess; i++)
        {
            const SubItem &sub = item->subItems[i];
            OutputThunk thunk = {farm, g, ctx, sub.chunkId};
            rc = mlsgpu_hip_worker_process(worker, item->dSplats, sub.firstSplat, sub.numSplats, sub.low, sub.numVertices,
                                           outputThunk, &thunk);
            if (rc != MLSGPU_OK)
            {
                farm->fail(rc, mlsgpu_hip_last_error());
                break;
            }
            std::lock_guard<std::mutex> l(farm->mutex);
            g->unallocated += sub.numSplats;         /* src/workers.cpp:281-284 */
            g->bucketsDone++;
        }
        {
            /* freeItem, src/workers.cpp:148-161 */
            std::lock_guard<std::mutex> l(farm->mutex);
            item->subItems.clear();
            g->pool.push_back(item);
            farm->inFlightItems--;
            farm->popCond.notify_all();
            farm->idleCond.notify_all();
        }
    }
    mlsgpu_hip_worker_destroy(worker);
    mlsgpu_hip_ctx_destroy(ctx);
}

/* CopyGroupBase::Worker::flush, src/workers.cpp:315-375 */
int flushBatch(Farm *f)
{
    if (f->bufferedItems.empty())
        return MLSGPU_OK;
    DeviceGroup *out = nullptr;
    WorkItem *item = nullptr;
    {
        std::unique_lock<std::mutex> l(f->mutex);
        for (;;)
        {
            if (f->error != MLSGPU_OK)
                return setError(f->error, "%s", f->errorText.c_str());
            /* among the devices that can take an item now, the one with the most unallocated capacity */
            uint64_t best = 0;
            for (auto &g : f->groups)
                if (!g->pool.empty() && g->unallocated >= best)
                {
                    best = g->unallocated;
                    out = g.get();
                }
            if (out != nullptr)
                break;
            f->popCond.wait(l);
        }
        item = out->pool.front();                   /* DeviceWorkerGroup::get, src/workers.cpp:135-146 */
        out->pool.pop_front();
        out->unallocated -= f->bufferedSplats;
        f->inFlightItems++;
    }
    item->subItems.swap(f->bufferedItems);
    item->numSplats = f->bufferedSplats;
    HIP_CHECK(hipSetDevice(out->device));
    HIP_CHECK(hipMemcpyAsync(item->dSplats, f->pinned[f->cur], f->bufferedSplats * sizeof(mlsgpu_splat),
                             hipMemcpyHostToDevice, out->copyStream));
    HIP_CHECK(hipEventRecord(item->copyEvent, out->copyStream));
    /* the staging buffer is free again once this copy is done; the other buffer is filled meanwhile */
    HIP_CHECK(hipEventRecord(f->pinnedFree[f->cur], out->copyStream));
    f->pinnedBusy[f->cur] = true;
    {
        std::lock_guard<std::mutex> l(f->mutex);
        f->stats[2] += f->bufferedSplats * sizeof(mlsgpu_splat);
        f->stats[3]++;
        out->queue.push_back(item);                 /* DeviceWorkerGroup::push */
    }
    f->queueCond.notify_all();
    f->bufferedSplats = 0;
    f->cur ^= 1;
    if (f->pinnedBusy[f->cur])
    {
        HIP_CHECK(hipEventSynchronize(f->pinnedFree[f->cur]));     /* copyEvent.wait(), src/workers.cpp:367-372 */
        f->pinnedBusy[f->cur] = false;
    }
    return MLSGPU_OK;
}

} // namespace

MLSGPU_API int mlsgpu_hip_farm_create(const mlsgpu_farm_config *cfg, mlsgpu_farm_output_fn output, void *user,
                                      mlsgpu_farm **out)
{
    REQUIRE(cfg != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(cfg->numDevices >= 1 && cfg->numDevices <= 16, MLSGPU_ERR_INVALID);
    REQUIRE(cfg->worker.maxBucketSplats >= 1, MLSGPU_ERR_INVALID);
    int count = 0;
    HIP_CHECK(hipGetDeviceCount(&count));
    mlsgpu_farm *f = new mlsgpu_farm;
    f->cfg = *cfg;
    if (f->cfg.workersPerDevice == 0) f->cfg.workersPerDevice = 1;
    if (f->cfg.spare == 0) f->cfg.spare = 1;
    f->output = output;
    f->user = user;
    const uint64_t cap = cfg->worker.maxBucketSplats;
    int rc = MLSGPU_OK;
    for (int b = 0; b < 2 && rc == MLSGPU_OK; b++)
    {
        if (hipHostMalloc((void **) &f->pinned[b], cap * sizeof(mlsgpu_splat)) != hipSuccess
            || hipEventCreateWithFlags(&f->pinnedFree[b], hipEventDisableTiming) != hipSuccess)
            rc = setError(MLSGPU_ERR_NOMEM, "farm: cannot allocate pinned staging of %llu splats", (unsigned long long) cap);
    }
    for (uint32_t d = 0; d < cfg->numDevices && rc == MLSGPU_OK; d++)
    {
        const int dev = cfg->devices ? cfg->devices[d] : (int) d;
        if (dev < 0 || dev >= count)
        {
            rc = setError(MLSGPU_ERR_INVALID, "farm: device %d of %d", dev, count);
            break;
        }
        std::unique_ptr<DeviceGroup> g(new DeviceGroup);
        g->farm = f;
        g->device = dev;
        g->index = d;
        if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&g->copyStream, hipStreamNonBlocking) != hipSuccess)
            rc = setError(MLSGPU_ERR_HIP, "farm: cannot create the copy stream on device %d", dev);
        if (rc == MLSGPU_OK)
            rc = mlsgpu_hip_ctx_create(dev, g->copyStream, &g->copyCtx);
        const uint32_t nItems = f->cfg.workersPerDevice + f->cfg.spare;
        for (uint32_t i = 0; i < nItems && rc == MLSGPU_OK; i++)
        {
            std::unique_ptr<WorkItem> item(new WorkItem);
            if (hipMalloc((void **) &item->dSplats, cap * sizeof(mlsgpu_splat)) != hipSuccess
                || hipEventCreateWithFlags(&item->copyEvent, hipEventDisableTiming) != hipSuccess)
                rc = setError(MLSGPU_ERR_NOMEM, "farm: cannot allocate a device item of %llu splats", (unsigned long long) cap);
            g->pool.push_back(item.get());
            g->items.push_back(std::move(item));
        }
        g->unallocated = cap * nItems;              /* src/workers.cpp:116 */
        f->groups.push_back(std::move(g));
    }
    if (rc != MLSGPU_OK)
    {
        mlsgpu_hip_farm_destroy(f);
        return rc;
    }
    for (auto &g : f->groups)
        for (uint32_t w = 0; w < f->cfg.workersPerDevice; w++)
            g->threads.push_back(std::thread(workerMain, static_cast<Farm *>(f), g.get()));
    *out = f;
    return MLSGPU_OK;
}

MLSGPU_API void mlsgpu_hip_farm_destroy(mlsgpu_farm *f)
{
    if (!f)
        return;
    {
        std::lock_guard<std::mutex> l(f->mutex);
        f->stopping = true;
    }
    f->queueCond.notify_all();
    for (auto &g : f->groups)
    {
        for (auto &t : g->threads)
            t.join();
        hipSetDevice(g->device);
        for (auto &item : g->items)
        {
            hipFree(item->dSplats);
            if (item->copyEvent) hipEventDestroy(item->copyEvent);
        }
        if (g->copyCtx) mlsgpu_hip_ctx_destroy(g->copyCtx);
        if (g->copyStream) hipStreamDestroy(g->copyStream);
    }
    for (int b = 0; b < 2; b++)
    {
        if (f->pinned[b]) hipHostFree(f->pinned[b]);
        if (f->pinnedFree[b]) hipEventDestroy(f->pinnedFree[b]);
    }
    delete f;
}

/* CopyGroupBase::Worker::operator(), src/workers.cpp:377-418 */
/* CopyGroup::get (src/workers.h:281-349 `get`): room for one bucket in the pinned staging buffer */
MLSGPU_API int mlsgpu_hip_farm_acquire(mlsgpu_farm *f, uint64_t numSplats, mlsgpu_splat **out)
{
    REQUIRE(f != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(numSplats <= f->cfg.worker.maxBucketSplats, MLSGPU_ERR_LENGTH);
    if (f->bufferedSplats + numSplats > f->cfg.worker.maxBucketSplats)
        PROPAGATE(flushBatch(f));
    f->acquired = numSplats;
    *out = f->pinned[f->cur] + f->bufferedSplats;
    return MLSGPU_OK;
}

/* CopyGroup::push: the bucket written into the acquired space joins the current batch */
MLSGPU_API int mlsgpu_hip_farm_push(mlsgpu_farm *f, uint64_t numSplats, const int32_t lowExtent[3],
                                    const uint32_t numVertices[3], uint64_t chunkId)
{
    REQUIRE(f != nullptr && lowExtent != nullptr && numVertices != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(numSplats <= f->acquired, MLSGPU_ERR_LENGTH);
    f->acquired = 0;
    SubItem sub;
    sub.chunkId = chunkId;
    for (int i = 0; i < 3; i++)
    {
        sub.low[i] = lowExtent[i];
        sub.numVertices[i] = numVertices[i];
    }
    sub.firstSplat = f->bufferedSplats;
    sub.numSplats = numSplats;
    f->bufferedItems.push_back(sub);
    f->bufferedSplats += numSplats;
    f->stats[0]++;
    f->stats[1] += numSplats;
    return MLSGPU_OK;
}

/* one host thread moves ~10 GB/s; a bucket of 2 M splats (64 MB) is split over a few so that staging keeps up
 * with the PCIe link */
static void parallelCopy(void *dst, const void *src, size_t bytes, unsigned threads)
{
    const size_t minChunk = size_t(4) << 20;
    if (threads <= 1 || bytes < 2 * minChunk)
    {
        std::memcpy(dst, src, bytes);
        return;
    }
    const size_t parts = std::min<size_t>(threads, bytes / minChunk);
    const size_t chunk = (bytes / parts + 4095) & ~size_t(4095);
    std::vector<std::thread> pool;
    for (size_t p = 1; p < parts; p++)
    {
        const size_t off = p * chunk;
        if (off >= bytes)
            break;
        pool.emplace_back([=] { std::memcpy((char *) dst + off, (const char *) src + off, std::min(chunk, bytes - off)); });
    }
    std::memcpy(dst, src, std::min(chunk, bytes));
    for (std::thread &t : pool)
        t.join();
}

MLSGPU_API int mlsgpu_hip_farm_submit(mlsgpu_farm *f, const mlsgpu_splat *hSplats, uint64_t numSplats,
                                      const int32_t lowExtent[3], const uint32_t numVertices[3], uint64_t chunkId)
{
    REQUIRE(f != nullptr && lowExtent != nullptr && numVertices != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(numSplats == 0 || hSplats != nullptr, MLSGPU_ERR_INVALID);
    mlsgpu_splat *dst = nullptr;
    PROPAGATE(mlsgpu_hip_farm_acquire(f, numSplats, &dst));
    parallelCopy(dst, hSplats, numSplats * sizeof(mlsgpu_splat), f->cfg.copyThreads == 0 ? 4u : f->cfg.copyThreads);
    return mlsgpu_hip_farm_push(f, numSplats, lowExtent, numVertices, chunkId);
}

/* A bucket whose splats are already on `device` (the device bucketer's callback): the item is filled by the gather +
 * transform kernel of mlsgpu_hip_bucket_load on the group's copy stream instead of a host-to-device copy. */
MLSGPU_API int mlsgpu_hip_farm_submit_device(mlsgpu_farm *f, int device, const mlsgpu_splat *dSplats, const uint32_t *dIds,
                                             uint64_t numSplats, const mlsgpu_grid *fullGrid, const int32_t lowExtent[3],
                                             const uint32_t numVertices[3], uint64_t chunkId)
{
    REQUIRE(f != nullptr && fullGrid != nullptr && lowExtent != nullptr && numVertices != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(numSplats == 0 || dSplats != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(numSplats <= f->cfg.worker.maxBucketSplats, MLSGPU_ERR_LENGTH);
    PROPAGATE(flushBatch(f));           /* host buckets submitted earlier keep their place in the order */
    DeviceGroup *out = nullptr;
    WorkItem *item = nullptr;
    {
        std::unique_lock<std::mutex> l(f->mutex);
        bool known = false;
        for (auto &g : f->groups)
            known = known || g->device == device;
        REQUIRE(known, MLSGPU_ERR_INVALID);
        for (;;)
        {
            if (f->error != MLSGPU_OK)
                return setError(f->error, "%s", f->errorText.c_str());
            uint64_t best = 0;
            for (auto &g : f->groups)
                if (g->device == device && !g->pool.empty() && g->unallocated >= best)
                {
                    best = g->unallocated;
                    out = g.get();
                }
            if (out != nullptr)
                break;
            f->popCond.wait(l);
        }
        item = out->pool.front();
        out->pool.pop_front();
        out->unallocated -= numSplats;
        f->inFlightItems++;
    }
    SubItem sub;
    sub.chunkId = chunkId;
    for (int i = 0; i < 3; i++)
    {
        sub.low[i] = lowExtent[i];
        sub.numVertices[i] = numVertices[i];
    }
    sub.firstSplat = 0;
    sub.numSplats = numSplats;
    item->subItems.assign(1, sub);
    item->numSplats = numSplats;
    int rc = mlsgpu_hip_bucket_load(out->copyCtx, dSplats, dIds, numSplats, fullGrid, item->dSplats);
    if (rc == MLSGPU_OK && hipEventRecord(item->copyEvent, out->copyStream) != hipSuccess)
        rc = setError(MLSGPU_ERR_HIP, "farm: cannot record the load event");
    /* the id list belongs to the caller (the bucketer reuses it after its callback returns) */
    if (rc == MLSGPU_OK && hipStreamSynchronize(out->copyStream) != hipSuccess)
        rc = setError(MLSGPU_ERR_HIP, "farm: the device-side load failed");
    {
        std::lock_guard<std::mutex> l(f->mutex);
        if (rc != MLSGPU_OK)
        {
            item->subItems.clear();
            out->pool.push_back(item);
            out->unallocated += numSplats;
            f->inFlightItems--;
        }
        else
        {
            f->stats[0]++;
            f->stats[1] += numSplats;
            f->stats[3]++;
            out->queue.push_back(item);
        }
    }
    if (rc == MLSGPU_OK)
        f->queueCond.notify_all();
    return rc;
}

MLSGPU_API int mlsgpu_hip_farm_finish(mlsgpu_farm *f)
{
    REQUIRE(f != nullptr, MLSGPU_ERR_INVALID);
    PROPAGATE(flushBatch(f));
    std::unique_lock<std::mutex> l(f->mutex);
    f->idleCond.wait(l, [&] { return f->inFlightItems == 0 || f->error != MLSGPU_OK; });
    if (f->error != MLSGPU_OK)
        return setError(f->error, "%s", f->errorText.c_str());
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_farm_stats(mlsgpu_farm *f, uint64_t out[24])
{
    REQUIRE(f != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> l(f->mutex);
    std::memset(out, 0, 24 * sizeof(uint64_t));
    for (int i = 0; i < 8; i++)
        out[i] = f->stats[i];
    for (size_t d = 0; d < f->groups.size() && d < 16; d++)
        out[8 + d] = f->groups[d]->bucketsDone;
    return MLSGPU_OK;
}

MLSGPU_API void mlsgpu_hip_transform_splats(mlsgpu_splat *s, uint64_t n, const float reference[3], float spacing,
                                            const int32_t low[3])
{
    const float invSpacing = 1.0f / spacing;
    for (uint64_t i = 0; i < n; i++)
    {
        for (int a = 0; a < 3; a++)
            s[i].position[a] = (s[i].position[a] - reference[a]) * invSpacing - (float) low[a];
        s[i].radius *= invSpacing;
    }
}
