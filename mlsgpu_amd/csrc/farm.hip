/*
 * Bucket farm: the reference's CopyGroup + per-device DeviceWorkerGroup (src/workers.{h,cpp}) re-expressed
 * with std::thread, HIP streams and events.  See include/mlsgpu_hip.h for the contract.
 *
 * Threads: the caller's thread does what the reference's "copy" thread does (stage, pick a device, enqueue
 * the H2D copy); `workersPerDevice` threads per GPU do what "device.N" threads do.  Ordering between the copy
 * stream and a worker's stream is a hipEvent (WorkItem::copyEvent in the reference).
 */
#include "common.hpp"
#include "placement.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
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
    hipEvent_t copyStart = nullptr;  /* timed pair around the item's last host-to-device copy: read when the item is next */
    hipEvent_t copyStop = nullptr;   /* taken from the pool (the copy is long done then) into the copy side's busy time */
    bool copyTimed = false;
    uint64_t numSplats = 0;
};

struct DeviceGroup;

struct Farm;

struct DeviceGroup               /* DeviceWorkerGroup, src/workers.h:214-350 */
{
    Farm *farm = nullptr;
    int device = 0;
    uint32_t index = 0;
    int node = -1;                       /* NUMA node the GPU hangs off (-1: unknown or a one-node machine) */
    uint32_t side = 0;                   /* the copy side (staging ring + copy threads of that node) that serves it */
    hipStream_t copyStream = nullptr;
    mlsgpu_ctx *copyCtx = nullptr;       /* the copy stream as a context, for device-side loads */
    std::vector<std::unique_ptr<WorkItem> > items;
    std::deque<WorkItem *> pool;         /* itemPool */
    std::deque<WorkItem *> queue;        /* pushed, not yet taken by a worker */
    uint64_t unallocated = 0;
    uint64_t bucketsDone = 0;
    double clock[4] = {0, 0, 0, 0};      /* this group's share of the farm's worker clock: sets, buckets, waited, worked */
    std::vector<std::thread> threads;
    std::vector<hipEvent_t> eventPool;   /* read-back events of this device (guarded by the farm's mutex) */
    /* Buckets of a cloud resident HERE on their way to another GPU's item: gathered into one buffer of a small ring,
     * then peer-copied on the target's copy stream.  `loaded` (this device) orders the copy behind the gather, `busy` (the
     * target item's copy event) orders the ring slot's next gather behind the copy -- both waits happen on the GPUs, so
     * several leaves are in flight at once; the host only waits for its own gather (the id list is the caller's). */
    struct PeerSlot
    {
        mlsgpu_splat *ptr = nullptr;
        hipEvent_t loaded = nullptr;
        hipEvent_t busy = nullptr;
    };
    std::vector<PeerSlot> peerRing;
    size_t peerCur = 0;
};

/* One ship-out on its way to the host: OutputGeneratorBuilder::Functor's MesherGroup::WorkItem, src/workers.h:488-509 */
struct HostSlot
{
    uint64_t offset = 0, bytes = 0;      /* the slot's bytes in the ring, padding at the wrap included in `bytes` */
    int device = 0;
    uint64_t chunkId = 0;
    mlsgpu_host_mesh mesh;
    hipEvent_t done = nullptr;
    DeviceGroup *group = nullptr;
    bool ready = false;                  /* the reads and `done` have been enqueued */
};

struct Farm
{
    mlsgpu_farm_config cfg;
    mlsgpu_farm_output_fn output = nullptr;
    void *user = nullptr;
    std::vector<std::unique_ptr<DeviceGroup> > groups;
    std::mutex mutex;                    /* guards pools, queues, counters, error, the host ring */
    std::condition_variable popCond;     /* an item went back to some pool (popCondition in the reference) */
    std::condition_variable queueCond;   /* an item was pushed / stopping */
    std::condition_variable idleCond;    /* a bucket finished / a host mesh was consumed */
    bool stopping = false;
    int error = MLSGPU_OK;
    std::string errorText;
    uint64_t inFlightItems = 0;
    uint64_t inFlightMax = 0;            /* high-water mark of inFlightItems */
    std::atomic<uint32_t> batch{1};      /* buckets of an item a worker takes through the path in lock-step */

    /* staging (CopyGroupBase::Worker: pinned + bufferedItems + bufferedSplats).  A ring of numDevices + 1 (at least
     * two) portable pinned buffers: while one is being filled, one copy per GPU can be in flight, each on its own
     * device's copy stream (the reference has one buffer and waits for its copy, src/workers.cpp:367-372).  A buffer
     * is free again once the copy event of the item it was sent to has completed; that event was created on the
     * item's device, so no event is ever recorded on another device's stream. */
    struct Staging
    {
        mlsgpu_splat *ptr = nullptr;
        hipEvent_t busy = nullptr;       /* WorkItem::copyEvent of the last copy out of this buffer, or null */
    };
    /* One COPY SIDE per NUMA node that has a GPU of the farm (placement.hpp): its ring of pinned buffers is allocated on
     * that node, its copy threads run there, and it feeds that node's GPUs -- the DMA engines read local memory and the
     * socket interconnect carries only what the caller's source array forces over it.  A one-socket machine has one. */
    struct CopySide
    {
        int node = -1;
        std::vector<Staging> staging;
        size_t cur = 0;
        std::vector<DeviceGroup *> groups;
        placement::CopyPool pool;
        int stagingNode = -1;            /* where the first staging buffer's memory really is (move_pages query) */
    };
    std::vector<std::unique_ptr<CopySide> > sides;
    CopySide *active = nullptr;          /* the side whose current buffer holds the batch being assembled */
    std::vector<SubItem> bufferedItems;
    uint64_t bufferedSplats = 0;
    uint64_t acquired = 0;          /* splats handed out by acquire and not pushed yet */

    uint64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    /* the copy side's clock (seconds): filling staging, waiting for a staging buffer, waiting for a device item, the
     * host-to-device copies themselves (timed event pairs), first submit .. last flush; [5] copies, [6] batches sent to a
     * GPU of another side because no own one had a free item, [7] inside the runtime's enqueue calls */
    double copyClock[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    /* the device workers' clock: [0] sets of launches (batches, or single buckets), [1] buckets in them, [2] seconds the
     * workers waited for an item, [3] seconds they spent processing (summed over the workers) */
    double workerClock[4] = {0, 0, 0, 0};
    std::chrono::steady_clock::time_point firstSubmit, lastFlush;
    bool clockRunning = false;

    /* host output: ship-outs are read back asynchronously on the worker's stream into a pinned circular buffer
     * (MesherGroup::meshBuffer, --mem-mesh) and consumed in arrival order by ONE mesher thread
     * (MesherGroup, src/workers.cpp:47-85). */
    bool hostOutput = false;
    mlsgpu_farm_landing_fn landingFn = nullptr;     /* set: ship-outs land in memory the consumer hands out, not in the ring */
    mlsgpu_farm_host_output_fn hostFn = nullptr;
    void *hostUser = nullptr;
    char *ring = nullptr;
    int ringNode = -1;                   /* where the ring's memory really is (move_pages query) */
    uint64_t ringBytes = 0, ringHead = 0, ringUsed = 0;   /* in use: the ringUsed bytes that end at ringHead */
    std::deque<HostSlot *> hostQueue;
    std::condition_variable hostCond;    /* a slot became ready / stopping */
    std::condition_variable ringCond;    /* ring space was released */
    std::thread mesherThread;
    uint32_t liveWorkers = 0;            /* the mesher thread outlives every thread that can still queue a slot */
    uint64_t hostStats[4] = {0, 0, 0, 0};   /* meshes, bytes read back, waits for ring space, largest mesh */

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
        hostCond.notify_all();
        ringCond.notify_all();
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

/* The ship-out goes to the host: room in the circular buffer (blocking while it is full, as
 * CircularBuffer::allocate does), enqueueReadMesh on the worker's own stream, an event behind the reads.  The
 * worker does not wait: Marching's next kernels are behind the reads in stream order, and the mesher thread waits
 * for the event (MesherGroup's verticesEvent / vertexKeysEvent / trianglesEvent). */
int hostReadBack(OutputThunk *t, const mlsgpu_mesh *mesh)
{
    Farm *f = t->farm;
    const uint64_t need = (mlsgpu_hip_mesh_host_bytes(mesh) + 63) & ~uint64_t(63);
    char *landing = nullptr;
    if (f->landingFn != nullptr)
    {
        /* the consumer's own memory (mlsgpu_hip_farm_set_host_landing): nothing to wait for, nothing to give back */
        void *p = nullptr;
        if (f->landingFn(f->hostUser, need, &p) != 0 || p == nullptr)
        {
            f->fail(MLSGPU_ERR_CALLBACK, "farm: the landing allocator of the host output failed");
            return setError(MLSGPU_ERR_CALLBACK, "farm: the landing allocator of the host output failed");
        }
        landing = static_cast<char *>(p);
    }
    else if (need > f->ringBytes)
    {
        /* the first error is the one finish() reports (Marching only says "the output functor failed") */
        char text[160];
        snprintf(text, sizeof(text), "farm: a ship-out of %llu bytes does not fit the host mesh buffer of %llu bytes",
                 (unsigned long long) need, (unsigned long long) f->ringBytes);
        f->fail(MLSGPU_ERR_LENGTH, text);
        return setError(MLSGPU_ERR_LENGTH, "%s", text);
    }
    std::unique_ptr<HostSlot> slot(new HostSlot);
    HostSlot *s = slot.get();
    {
        std::unique_lock<std::mutex> l(f->mutex);
        bool waited = false;
        for (;;)
        {
            if (f->error != MLSGPU_OK)
                return setError(f->error, "%s", f->errorText.c_str());
            if (landing != nullptr)
            {
                s->offset = 0;
                s->bytes = 0;           /* no ring space to release */
                break;
            }
            /* contiguous room at the head, or after wrapping (the skipped tail end is charged to this slot) */
            const uint64_t toEnd = f->ringBytes - f->ringHead;
            const uint64_t pad = need <= toEnd ? 0 : toEnd;
            if (f->ringUsed + pad + need <= f->ringBytes)
            {
                s->offset = pad ? 0 : f->ringHead;
                s->bytes = pad + need;
                f->ringHead = (s->offset + need) % f->ringBytes;
                f->ringUsed += s->bytes;
                break;
            }
            if (!waited)
                f->hostStats[2]++;
            waited = true;
            f->ringCond.wait(l);
        }
        if (!t->group->eventPool.empty())
        {
            s->done = t->group->eventPool.back();
            t->group->eventPool.pop_back();
        }
        s->device = t->group->device;
        s->group = t->group;
        s->chunkId = t->chunkId;
        f->hostQueue.push_back(slot.release());         /* arrival order = allocation order = release order */
        f->hostStats[0]++;
        f->hostStats[1] += need;
        f->hostStats[3] = std::max(f->hostStats[3], need);
    }
    int rc = MLSGPU_OK;
    hipStream_t stream = static_cast<hipStream_t>(mlsgpu_hip_ctx_stream(t->ctx));
    /* the event must belong to the worker's device whatever an output functor before this one left current */
    if (hipSetDevice(t->group->device) != hipSuccess)
        rc = setError(MLSGPU_ERR_HIP, "farm: cannot select device %d", t->group->device);
    if (rc == MLSGPU_OK && s->done == nullptr && hipEventCreateWithFlags(&s->done, hipEventDisableTiming) != hipSuccess)
        rc = setError(MLSGPU_ERR_HIP, "farm: cannot create a read-back event");
    char *blob = landing != nullptr ? landing : f->ring + s->offset;
    const uint64_t ne = mesh->numVertices - mesh->numInternalVertices;
    s->mesh.vertexKeys = reinterpret_cast<const uint64_t *>(blob);
    s->mesh.vertices = reinterpret_cast<const float *>(blob + 8 * ne);
    s->mesh.triangles = reinterpret_cast<const uint32_t *>(blob + 8 * ne + 12 * mesh->numVertices);
    s->mesh.numVertices = mesh->numVertices;
    s->mesh.numTriangles = mesh->numTriangles;
    s->mesh.numInternalVertices = mesh->numInternalVertices;
    if (rc == MLSGPU_OK)
        rc = mlsgpu_hip_mesh_read(t->ctx, mesh, blob, 1);
    if (rc == MLSGPU_OK && hipEventRecord(s->done, stream) != hipSuccess)
        rc = setError(MLSGPU_ERR_HIP, "farm: cannot record a read-back event");
    if (rc != MLSGPU_OK)
    {
        /* the slot stays queued (the ring is released in order); the mesher thread drops it once the farm has failed */
        const std::string text = mlsgpu_hip_last_error();
        f->fail(rc, text.c_str());
        hipStreamSynchronize(stream);                    /* nothing may still be writing into the ring */
    }
    {
        std::lock_guard<std::mutex> l(f->mutex);
        s->ready = true;
    }
    f->hostCond.notify_all();
    return rc;
}

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
        PROPAGATE(t->farm->output(t->farm->user, t->group->device, t->chunkId, t->ctx, mesh));
    if (t->farm->hostOutput)
        return hostReadBack(t, mesh);
    return 0;
}

/* MesherGroupBase::Worker::operator(), src/workers.cpp:47-52: wait for the reads, hand the mesh to the consumer,
 * give the memory back to the circular buffer */
void mesherMain(Farm *f)
{
    for (;;)
    {
        HostSlot *s = nullptr;
        {
            std::unique_lock<std::mutex> l(f->mutex);
            f->hostCond.wait(l, [&] { return (!f->hostQueue.empty() && f->hostQueue.front()->ready)
                                             || (f->stopping && f->liveWorkers == 0 && f->hostQueue.empty()); });
            if (f->hostQueue.empty())
                break;
            s = f->hostQueue.front();
        }
        int rc = MLSGPU_OK;
        bool failed;
        {
            std::lock_guard<std::mutex> l(f->mutex);
            failed = f->error != MLSGPU_OK;
        }
        if (s->done != nullptr && hipEventSynchronize(s->done) != hipSuccess && !failed)
        {
            rc = MLSGPU_ERR_HIP;
            f->fail(rc, "farm: waiting for a mesh read-back failed");
            failed = true;
        }
        if (!failed && f->hostFn != nullptr)
        {
            rc = f->hostFn(f->hostUser, s->device, s->chunkId, &s->mesh);
            if (rc != MLSGPU_OK)
                f->fail(MLSGPU_ERR_CALLBACK, "farm: the host output functor reported an error");
        }
        {
            std::lock_guard<std::mutex> l(f->mutex);
            f->hostQueue.pop_front();
            f->ringUsed -= s->bytes;                    /* slots are released in allocation order */
            if (f->ringUsed == 0)
                f->ringHead = 0;
            if (s->done != nullptr)
                s->group->eventPool.push_back(s->done);
        }
        delete s;
        f->ringCond.notify_all();
        f->idleCond.notify_all();
    }
}

double secondsSince(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

struct BatchThunk
{
    Farm *farm;
    DeviceGroup *group;
    mlsgpu_ctx *ctx;
    std::vector<uint64_t> chunkIds;         /* of the batch's buckets, in order */
};

/* the output functor of a batch: the mesh belongs to bucket `index` */
int batchOutputThunk(void *user, uint32_t index, void *stream, const mlsgpu_mesh *mesh)
{
    BatchThunk *b = static_cast<BatchThunk *>(user);
    OutputThunk t = {b->farm, b->group, b->ctx, b->chunkIds[index]};
    return outputThunk(&t, stream, mesh);
}

/* DeviceWorkerGroupBase::Worker::operator(), src/workers.cpp:232-286 */
void workerMain(Farm *farm, DeviceGroup *g)
{
    /* a device worker's host side -- launches, mailbox polls in pinned memory, the output functor -- next to its GPU */
    placement::bindThisThread(placement::cpusOfNode(g->node));
    mlsgpu_ctx *ctx = nullptr;
    mlsgpu_worker *worker = nullptr;
    int rc = mlsgpu_hip_ctx_create(g->device, nullptr, &ctx);
    if (rc == MLSGPU_OK)
        rc = mlsgpu_hip_worker_create(ctx, &farm->cfg.worker, &worker);
    /* a device item is the farm's own copy, so the tree could turn its radii into 1 / r^2 in place (kernels/octree.cl:193);
     * taking 1 / r^2 when a splat is staged instead spares the strided 4-byte stores of the build: cfg5 565.5-566.5 -> 555-558 ms
     * per pass, same meshes */
    if (rc == MLSGPU_OK)
        rc = mlsgpu_hip_worker_set_keep_splats(worker, 1);
    if (rc != MLSGPU_OK)
    {
        farm->fail(rc, mlsgpu_hip_last_error());
        if (ctx) mlsgpu_hip_ctx_destroy(ctx);
        ctx = nullptr;
    }
    bool lanesRefused = false;          /* the device had no room for the lanes of a batch: bucket by bucket from then on */
    for (;;)
    {
        /* One item, or -- with lanes -- as many queued items as fit a batch: device-side leaves arrive one bucket per item
         * (mlsgpu_hip_farm_submit_device), and the buckets of several such items share one set of launches. */
        std::vector<WorkItem *> taken;
        const uint32_t lanes = farm->batch.load();
        const auto tIdle = std::chrono::steady_clock::now();
        {
            std::unique_lock<std::mutex> l(farm->mutex);
            farm->queueCond.wait(l, [&] { return farm->stopping || !g->queue.empty(); });
            const double waited = secondsSince(tIdle);
            farm->workerClock[2] += waited;
            g->clock[2] += waited;
            if (g->queue.empty())
                break;
            size_t subs = 0;
            do
            {
                taken.push_back(g->queue.front());
                subs += g->queue.front()->subItems.size();
                g->queue.pop_front();
            } while (lanes > 1 && !g->queue.empty() && subs + g->queue.front()->subItems.size() <= lanes);
        }
        /* after a failure the queued items are only handed back, so that finish() and destroy() never wait for them */
        bool run = ctx != nullptr;
        {
            std::lock_guard<std::mutex> l(farm->mutex);
            run = run && farm->error == MLSGPU_OK;
        }
        /* wait[0] = work.copyEvent (src/workers.cpp:268) */
        hipError_t e = hipSuccess;
        for (WorkItem *item : taken)
            if (run && e == hipSuccess)
            {
                e = hipStreamWaitEvent(static_cast<hipStream_t>(mlsgpu_hip_ctx_stream(ctx)), item->copyEvent, 0);
                if (e != hipSuccess)
                    farm->fail(MLSGPU_ERR_HIP, hipGetErrorString(e));
            }
        std::vector<size_t> processed(taken.size(), 0);
        size_t numSubs = 0;
        for (WorkItem *item : taken)
            numSubs += item->subItems.size();
        const auto tWork = std::chrono::steady_clock::now();
        /* the lanes' buffers (a tree, a field, a lattice and a mesh arena each) are allocated on first use; a device that
         * cannot hold them keeps the worker on the bucket-by-bucket path instead of failing the farm */
        bool batched = run && e == hipSuccess && lanes > 1 && numSubs > 1 && !lanesRefused;
        if (batched && mlsgpu_hip_worker_batch(worker) < lanes)
        {
            rc = mlsgpu_hip_worker_set_batch(worker, lanes);
            if (rc == MLSGPU_ERR_NOMEM)
                lanesRefused = true, batched = false;
            else if (rc != MLSGPU_OK)
                farm->fail(rc, mlsgpu_hip_last_error()), batched = false, run = false;
        }
        if (batched)
        {
            /* the SubItems `lanes` at a time through ONE set of launches (mlsgpu_hip_worker_process_batch); meshes still
             * arrive bucket by bucket, in order */
            rc = MLSGPU_OK;
            std::vector<mlsgpu_subitem> subs;
            BatchThunk thunk = {farm, g, ctx, {}};
            for (WorkItem *item : taken)
                for (const SubItem &sub : item->subItems)
                {
                    mlsgpu_subitem s;
                    s.firstSplat = sub.firstSplat;
                    s.numSplats = sub.numSplats;
                    for (int a = 0; a < 3; a++)
                    {
                        s.lowExtent[a] = sub.low[a];
                        s.numVertices[a] = sub.numVertices[a];
                    }
                    s.dSplats = item->dSplats;
                    subs.push_back(s);
                    thunk.chunkIds.push_back(sub.chunkId);
                }
            if (rc == MLSGPU_OK)
                rc = mlsgpu_hip_worker_process_batch(worker, nullptr, subs.data(), (uint32_t) subs.size(), batchOutputThunk, &thunk);
            if (rc != MLSGPU_OK)
                farm->fail(rc, mlsgpu_hip_last_error());
            /* per bucket, as src/workers.cpp:281-284: after a failure in mid-batch the buckets that had delivered all their
             * meshes count as done, the others go back unprocessed below */
            size_t done = rc == MLSGPU_OK ? subs.size() : mlsgpu_hip_worker_batch_completed(worker);
            std::lock_guard<std::mutex> l(farm->mutex);
            for (size_t t = 0; t < taken.size() && done > 0; t++)
                for (const SubItem &sub : taken[t]->subItems)
                {
                    if (done == 0)
                        break;
                    done--;
                    processed[t]++;
                    g->unallocated += sub.numSplats;
                    g->bucketsDone++;
                }
        }
        else
            for (size_t t = 0; t < taken.size(); t++)
            {
                WorkItem *item = taken[t];
                for (size_t i = 0; run && i < item->subItems.size() && e == hipSuccess; i++)
                {
                    const SubItem &sub = item->subItems[i];
                    OutputThunk thunk = {farm, g, ctx, sub.chunkId};
                    rc = mlsgpu_hip_worker_process(worker, item->dSplats, sub.firstSplat, sub.numSplats, sub.low, sub.numVertices,
                                                   outputThunk, &thunk);
                    if (rc != MLSGPU_OK)
                    {
                        farm->fail(rc, mlsgpu_hip_last_error());
                        run = false;
                        break;
                    }
                    processed[t] = i + 1;
                    std::lock_guard<std::mutex> l(farm->mutex);
                    g->unallocated += sub.numSplats;         /* src/workers.cpp:281-284 */
                    g->bucketsDone++;
                }
            }
        {
            std::lock_guard<std::mutex> l(farm->mutex);
            const double worked = secondsSince(tWork);
            farm->workerClock[0] += 1;
            farm->workerClock[1] += (double) numSubs;
            farm->workerClock[3] += worked;
            g->clock[0] += 1;
            g->clock[1] += (double) numSubs;
            g->clock[3] += worked;
        }
        for (size_t t = 0; t < taken.size(); t++)
        {
            WorkItem *item = taken[t];
            if (processed[t] < item->subItems.size())
            {
                /* a failed or skipped item (the farm has failed: it refuses further work and can only be finished and
                 * destroyed): the copy that filled it may still be running -- it must not be when the item is handed out
                 * again -- and the splats of the buckets that were not processed go back to the group's capacity */
                hipStreamSynchronize(g->copyStream);
                std::lock_guard<std::mutex> l(farm->mutex);
                for (size_t i = processed[t]; i < item->subItems.size(); i++)
                    g->unallocated += item->subItems[i].numSplats;
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
    }
    if (ctx != nullptr)
    {
        mlsgpu_hip_ctx_synchronize(ctx);             /* read-backs still in flight use this stream */
        mlsgpu_hip_worker_destroy(worker);
        mlsgpu_hip_ctx_destroy(ctx);
    }
    {
        std::lock_guard<std::mutex> l(farm->mutex);
        farm->liveWorkers--;
    }
    farm->hostCond.notify_all();
}

/* a failure between taking an item from a pool and queueing it: the item goes back, the accounting is undone and the
 * farm is marked failed, so that finish() reports the error instead of waiting for an item that will never be done */
int abandonItem(Farm *f, DeviceGroup *g, WorkItem *item, uint64_t splats, int rc)
{
    const std::string text = mlsgpu_hip_last_error();
    {
        std::lock_guard<std::mutex> l(f->mutex);
        item->subItems.clear();
        g->pool.push_back(item);
        g->unallocated += splats;
        f->inFlightItems--;
    }
    f->fail(rc, text.c_str());
    return setError(rc, "%s", text.c_str());
}

/* among the groups that can take an item now, the one with the most unallocated capacity (flush, src/workers.cpp:
 * 320-351); `prefer` >= 0 breaks ties in favour of that device; `side` non-null restricts the choice to the GPUs that copy
 * side serves.  Called with the mutex held; null if none. */
DeviceGroup *pickGroup(Farm *f, int prefer, const Farm::CopySide *side = nullptr)
{
    DeviceGroup *out = nullptr;
    uint64_t best = 0;
    for (auto &g : f->groups)
        if (!g->pool.empty() && (side == nullptr || f->sides[g->side].get() == side)
            && (out == nullptr || g->unallocated > best || (g->unallocated == best && g->device == prefer && out->device != prefer)))
        {
            best = g->unallocated;
            out = g.get();
        }
    return out;
}

/* the side the next batch is staged on: the one of the group the batch would go to if it were flushed now -- the most
 * unallocated capacity, a group that can take an item before one that cannot.  Called with the mutex held. */
Farm::CopySide *pickSide(Farm *f)
{
    if (f->sides.size() == 1)
        return f->sides[0].get();
    DeviceGroup *out = pickGroup(f, -1);
    if (out == nullptr)
        for (auto &g : f->groups)
            if (out == nullptr || g->unallocated > out->unallocated)
                out = g.get();
    return f->sides[out->side].get();
}

/* CopyGroupBase::Worker::flush, src/workers.cpp:315-375 */
int flushBatch(Farm *f)
{
    if (f->bufferedItems.empty())
        return MLSGPU_OK;
    DeviceGuard restore;
    DeviceGroup *out = nullptr;
    WorkItem *item = nullptr;
    Farm::CopySide *side = f->active;
    const auto tWait = std::chrono::steady_clock::now();
    {
        std::unique_lock<std::mutex> l(f->mutex);
        for (;;)
        {
            if (f->error != MLSGPU_OK)
                return setError(f->error, "%s", f->errorText.c_str());
            /* a GPU of the side the batch was staged on; when none of them can take an item and another side's can, the
             * batch goes there (the reference's greedy choice never waits while some device is free) */
            out = pickGroup(f, -1, side);
            if (out == nullptr && (out = pickGroup(f, -1)) != nullptr)
                f->copyClock[6] += 1;
            if (out != nullptr)
                break;
            f->popCond.wait(l);
        }
        item = out->pool.front();                   /* DeviceWorkerGroup::get, src/workers.cpp:135-146 */
        out->pool.pop_front();
        out->unallocated -= f->bufferedSplats;
        f->inFlightItems++;
        f->inFlightMax = std::max(f->inFlightMax, f->inFlightItems);
    }
    f->copyClock[2] += secondsSince(tWait);
    const uint64_t splats = f->bufferedSplats;
    item->subItems.swap(f->bufferedItems);
    item->numSplats = splats;
    f->bufferedItems.clear();
    f->bufferedSplats = 0;
    Farm::Staging &st = side->staging[side->cur];
    const auto tEnqueue = std::chrono::steady_clock::now();
    hipError_t e = hipSetDevice(out->device);
    if (e == hipSuccess && item->copyTimed)
    {
        /* the item's previous copy ended long ago (a worker has processed the item since): its duration */
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, item->copyStart, item->copyStop) == hipSuccess)
            f->copyClock[3] += ms * 1e-3;
        item->copyTimed = false;
        (void) hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipEventRecord(item->copyStart, out->copyStream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(item->dSplats, st.ptr, splats * sizeof(mlsgpu_splat), hipMemcpyHostToDevice, out->copyStream);
    if (e == hipSuccess)
        e = hipEventRecord(item->copyStop, out->copyStream);
    if (e == hipSuccess)
        e = hipEventRecord(item->copyEvent, out->copyStream);
    if (e != hipSuccess)
    {
        setError(MLSGPU_ERR_HIP, "farm: host-to-device copy to device %d failed: %s", out->device, hipGetErrorString(e));
        hipStreamSynchronize(out->copyStream);
        return abandonItem(f, out, item, splats, MLSGPU_ERR_HIP);
    }
    item->copyTimed = true;
    f->copyClock[7] += secondsSince(tEnqueue);
    st.busy = item->copyEvent;          /* this buffer is free again once that copy is done */
    {
        std::lock_guard<std::mutex> l(f->mutex);
        f->stats[2] += splats * sizeof(mlsgpu_splat);
        f->stats[3]++;
        f->copyClock[5] += 1;
        out->queue.push_back(item);                 /* DeviceWorkerGroup::push */
    }
    f->queueCond.notify_all();
    f->lastFlush = std::chrono::steady_clock::now();
    side->cur = (side->cur + 1) % side->staging.size();
    f->active = nullptr;                /* the next bucket chooses its side (and waits for that side's next buffer) */
    return MLSGPU_OK;
}

/* the buffer of `side` the next batch is assembled in: it may still be the source of an older copy */
int claimStaging(Farm *f, Farm::CopySide *side)
{
    Farm::Staging &next = side->staging[side->cur];
    if (next.busy != nullptr)
    {
        const auto t0 = std::chrono::steady_clock::now();
        const hipError_t e = hipEventSynchronize(next.busy);         /* copyEvent.wait(), src/workers.cpp:367-372 */
        f->copyClock[1] += secondsSince(t0);
        next.busy = nullptr;
        if (e != hipSuccess)
        {
            f->fail(MLSGPU_ERR_HIP, hipGetErrorString(e));
            return setError(MLSGPU_ERR_HIP, "farm: waiting for a staging buffer failed: %s", hipGetErrorString(e));
        }
    }
    f->active = side;
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
        g->node = mlsgpu_hip_device_node(dev);
        /* everything below belongs to `dev`: streams, events and items are created with it current */
        if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&g->copyStream, hipStreamNonBlocking) != hipSuccess)
            rc = setError(MLSGPU_ERR_HIP, "farm: cannot create the copy stream on device %d", dev);
        if (rc == MLSGPU_OK)
            rc = mlsgpu_hip_ctx_create(dev, g->copyStream, &g->copyCtx);
        const uint32_t nItems = f->cfg.workersPerDevice + f->cfg.spare;
        for (uint32_t i = 0; i < nItems && rc == MLSGPU_OK; i++)
        {
            std::unique_ptr<WorkItem> item(new WorkItem);
            if (hipMalloc((void **) &item->dSplats, cap * sizeof(mlsgpu_splat)) != hipSuccess
                || hipEventCreateWithFlags(&item->copyEvent, hipEventDisableTiming) != hipSuccess
                || hipEventCreate(&item->copyStart) != hipSuccess || hipEventCreate(&item->copyStop) != hipSuccess)
                rc = setError(MLSGPU_ERR_NOMEM, "farm: cannot allocate a device item of %llu splats", (unsigned long long) cap);
            g->pool.push_back(item.get());
            g->items.push_back(std::move(item));
        }
        g->unallocated = cap * nItems;              /* src/workers.cpp:116 */
        f->groups.push_back(std::move(g));
    }
    /* ---- copy sides: one per NUMA node with a GPU of the farm.  A side's pinned ring is allocated with one of its GPUs
     * current, on a thread bound to the node (the runtime places pinned memory next to the current device; the binding
     * covers a runtime that places it next to the caller); depth = its GPUs + 2, so that while one buffer is being filled
     * one copy per GPU can be in flight and one more is queued behind them; its copy threads are bound to the node. ---- */
    if (rc == MLSGPU_OK)
    {
        std::vector<int> nodes, sideOf, nodeOfSide;
        for (auto &g : f->groups)
            nodes.push_back(g->node);
        placement::planSides(nodes.data(), nodes.size(), placement::numNodes(), sideOf, nodeOfSide);
        for (size_t k = 0; k < nodeOfSide.size(); k++)
        {
            f->sides.emplace_back(new Farm::CopySide);
            f->sides.back()->node = nodeOfSide[k];
        }
        for (size_t i = 0; i < f->groups.size(); i++)
        {
            f->groups[i]->side = (uint32_t) sideOf[i];
            f->sides[(size_t) sideOf[i]]->groups.push_back(f->groups[i].get());
        }
        const uint32_t copyThreads = f->cfg.copyThreads == 0 ? 4u : f->cfg.copyThreads;
        for (auto &sp : f->sides)
        {
            Farm::CopySide *side = sp.get();
            const std::vector<int> cpus = placement::cpusOfNode(side->node);
            const uint32_t depth = f->cfg.stagingBuffers != 0 ? std::max(2u, f->cfg.stagingBuffers)
                                                              : (uint32_t) side->groups.size() + 2;
            side->staging.resize(depth);
            int arc = MLSGPU_OK;
            std::thread allocator([&] {
                placement::bindThisThread(cpus);
                if (hipSetDevice(side->groups[0]->device) != hipSuccess)
                    arc = MLSGPU_ERR_HIP;
                for (size_t b = 0; b < side->staging.size() && arc == MLSGPU_OK; b++)
                    if (hipHostMalloc((void **) &side->staging[b].ptr, cap * sizeof(mlsgpu_splat), hipHostMallocPortable) != hipSuccess)
                        arc = MLSGPU_ERR_NOMEM;
                    else
                        std::memset(side->staging[b].ptr, 0, std::min<uint64_t>(cap * sizeof(mlsgpu_splat), 4096));
            });
            allocator.join();
            if (arc != MLSGPU_OK)
            {
                rc = setError(arc, "farm: cannot allocate pinned staging of %llu splats x %u", (unsigned long long) cap, depth);
                break;
            }
            side->stagingNode = placement::nodeOfAddress(side->staging[0].ptr);
            side->pool.start(copyThreads, cpus);
        }
    }
    if (rc != MLSGPU_OK)
    {
        mlsgpu_hip_farm_destroy(f);
        return rc;
    }
    for (auto &a : f->groups)
        for (auto &b : f->groups)
            if (a->device < b->device)
                enablePeerAccess(a->device, b->device);
    f->liveWorkers = (uint32_t) f->groups.size() * f->cfg.workersPerDevice;
    for (auto &g : f->groups)
        for (uint32_t w = 0; w < f->cfg.workersPerDevice; w++)
            g->threads.push_back(std::thread(workerMain, static_cast<Farm *>(f), g.get()));
    *out = f;
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_farm_set_host_output(mlsgpu_farm *f, uint64_t ringBytes, mlsgpu_farm_host_output_fn fn, void *user)
{
    REQUIRE(f != nullptr && ringBytes >= 4096, MLSGPU_ERR_INVALID);
    if (f->hostOutput)
    {
        /* between jobs (everything submitted has been finished): the next job's meshes go to another consumer; the ring
         * keeps its size */
        std::lock_guard<std::mutex> l(f->mutex);
        REQUIRE(f->inFlightItems == 0 && f->hostQueue.empty() && f->bufferedItems.empty(), MLSGPU_ERR_INVALID);
        f->hostFn = fn;
        f->hostUser = user;
        f->landingFn = nullptr;         /* back to the ring (of the size it was created with) */
        return MLSGPU_OK;
    }
    REQUIRE(f->stats[0] == 0, MLSGPU_ERR_INVALID);          /* before the first bucket */
    ringBytes = (ringBytes + 63) & ~uint64_t(63);
    {
        /* ONE ring and ONE mesher thread, as the reference (src/workers.cpp:47-85): next to the first GPU */
        const std::vector<int> cpus = placement::cpusOfNode(f->groups[0]->node);
        hipError_t e = hipSuccess;
        std::thread allocator([&] {
            placement::bindThisThread(cpus);
            e = hipSetDevice(f->groups[0]->device);
            if (e == hipSuccess)
                e = hipHostMalloc((void **) &f->ring, ringBytes, hipHostMallocPortable);
        });
        allocator.join();
        if (e != hipSuccess)
            return setError(MLSGPU_ERR_NOMEM, "farm: cannot allocate %llu bytes of pinned mesh buffer", (unsigned long long) ringBytes);
        f->ring[0] = 0;
        f->ringNode = placement::nodeOfAddress(f->ring);
        if (getenv("MLSGPU_HIP_FARM_TRACE") != nullptr)
        {
            std::string where;
            for (int k = 0; k < 32; k++)
                where += std::to_string(placement::nodeOfAddress(f->ring + (ringBytes / 32) * k)) + " ";
            fprintf(stderr, "mlsgpu_hip farm: the ring's %llu MB, node of every 32nd part: %s\n", (unsigned long long) (ringBytes >> 20), where.c_str());
        }
    }
    f->ringBytes = ringBytes;
    f->hostFn = fn;
    f->hostUser = user;
    f->mesherThread = std::thread([f] {
        placement::bindThisThread(placement::cpusOfNode(f->groups[0]->node));
        mesherMain(static_cast<Farm *>(f));
    });
    f->hostOutput = true;
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_farm_set_host_landing(mlsgpu_farm *f, mlsgpu_farm_landing_fn landing, mlsgpu_farm_host_output_fn fn, void *user)
{
    REQUIRE(f != nullptr && landing != nullptr, MLSGPU_ERR_INVALID);
    /* the ring route's machinery (slot queue, events, ONE mesher thread in ship-out order) with a token ring: no ship-out
     * ever waits for room */
    if (!f->hostOutput)
        PROPAGATE(mlsgpu_hip_farm_set_host_output(f, 4096, fn, user));
    std::lock_guard<std::mutex> l(f->mutex);
    REQUIRE(f->hostQueue.empty(), MLSGPU_ERR_INVALID);
    f->landingFn = landing;
    f->hostFn = fn;
    f->hostUser = user;
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
    f->hostCond.notify_all();
    for (auto &g : f->groups)
        for (auto &t : g->threads)
            t.join();
    f->hostCond.notify_all();
    if (f->mesherThread.joinable())
        f->mesherThread.join();
    for (auto &g : f->groups)
    {
        hipSetDevice(g->device);
        for (auto &item : g->items)
        {
            hipFree(item->dSplats);
            if (item->copyEvent) hipEventDestroy(item->copyEvent);
            if (item->copyStart) hipEventDestroy(item->copyStart);
            if (item->copyStop) hipEventDestroy(item->copyStop);
        }
        for (hipEvent_t e : g->eventPool)
            hipEventDestroy(e);
        for (auto &ps : g->peerRing)
        {
            hipFree(ps.ptr);
            if (ps.loaded) hipEventDestroy(ps.loaded);
        }
        if (g->copyCtx) mlsgpu_hip_ctx_destroy(g->copyCtx);
        if (g->copyStream) hipStreamDestroy(g->copyStream);
    }
    for (auto &side : f->sides)
    {
        side->pool.stop();
        for (auto &st : side->staging)
            if (st.ptr) hipHostFree(st.ptr);
    }
    if (f->ring) hipHostFree(f->ring);
    delete f;
}

/* CopyGroupBase::Worker::operator(), src/workers.cpp:377-418 */
/* CopyGroup::get (src/workers.h:281-349 `get`): room for one bucket in the pinned staging buffer */
MLSGPU_API int mlsgpu_hip_farm_acquire(mlsgpu_farm *f, uint64_t numSplats, mlsgpu_splat **out)
{
    REQUIRE(f != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(numSplats <= f->cfg.worker.maxBucketSplats, MLSGPU_ERR_LENGTH);
    if (!f->clockRunning)
    {
        f->firstSubmit = std::chrono::steady_clock::now();
        f->clockRunning = true;
    }
    if (f->bufferedSplats + numSplats > f->cfg.worker.maxBucketSplats)
        PROPAGATE(flushBatch(f));
    if (f->active == nullptr)
    {
        Farm::CopySide *side;
        {
            std::lock_guard<std::mutex> l(f->mutex);
            side = pickSide(f);
        }
        PROPAGATE(claimStaging(f, side));
    }
    f->acquired = numSplats;
    *out = f->active->staging[f->active->cur].ptr + f->bufferedSplats;
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

MLSGPU_API int mlsgpu_hip_farm_submit(mlsgpu_farm *f, const mlsgpu_splat *hSplats, uint64_t numSplats,
                                      const int32_t lowExtent[3], const uint32_t numVertices[3], uint64_t chunkId)
{
    REQUIRE(f != nullptr && lowExtent != nullptr && numVertices != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(numSplats == 0 || hSplats != nullptr, MLSGPU_ERR_INVALID);
    mlsgpu_splat *dst = nullptr;
    PROPAGATE(mlsgpu_hip_farm_acquire(f, numSplats, &dst));
    /* one host thread moves ~10 GB/s: a bucket of 2 M splats (64 MB) is split over the side's copy threads so that
     * staging keeps up with the PCIe link */
    const auto t0 = std::chrono::steady_clock::now();
    f->active->pool.copy(dst, hSplats, numSplats * sizeof(mlsgpu_splat));
    f->copyClock[0] += secondsSince(t0);
    return mlsgpu_hip_farm_push(f, numSplats, lowExtent, numVertices, chunkId);
}

/* A bucket whose splats are already on GPU `device` (the device bucketer's callback): the item is filled by the gather +
 * transform kernel of mlsgpu_hip_bucket_load instead of a host-to-device copy.  The bucket goes to the group with the
 * most unallocated capacity, like a host bucket (flush, src/workers.cpp:320-351), `device`'s own group on a tie.  On
 * the same GPU the kernel writes straight into the item; for another GPU it writes into a scratch buffer on `device`
 * and the item is filled by a peer copy on the target's copy stream (xGMI), so one resident cloud feeds every GPU. */
MLSGPU_API int mlsgpu_hip_farm_submit_device(mlsgpu_farm *f, int device, const mlsgpu_splat *dSplats, const uint32_t *dIds,
                                             uint64_t numSplats, const mlsgpu_grid *fullGrid, const int32_t lowExtent[3],
                                             const uint32_t numVertices[3], uint64_t chunkId)
{
    void *consumed = nullptr;
    PROPAGATE(mlsgpu_hip_farm_submit_device_async(f, device, dSplats, dIds, numSplats, fullGrid, lowExtent, numVertices, chunkId,
                                                  &consumed));
    /* the caller's id list is free again once the gather has run */
    if (consumed != nullptr && hipEventSynchronize(static_cast<hipEvent_t>(consumed)) != hipSuccess)
    {
        f->fail(MLSGPU_ERR_HIP, "farm: the device-side load failed");
        return setError(MLSGPU_ERR_HIP, "farm: the device-side load failed");
    }
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_farm_submit_device_async(mlsgpu_farm *f, int device, const mlsgpu_splat *dSplats, const uint32_t *dIds,
                                                   uint64_t numSplats, const mlsgpu_grid *fullGrid, const int32_t lowExtent[3],
                                                   const uint32_t numVertices[3], uint64_t chunkId, void **consumed)
{
    REQUIRE(consumed != nullptr, MLSGPU_ERR_INVALID);
    *consumed = nullptr;
    REQUIRE(f != nullptr && fullGrid != nullptr && lowExtent != nullptr && numVertices != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(numSplats == 0 || dSplats != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(numSplats <= f->cfg.worker.maxBucketSplats, MLSGPU_ERR_LENGTH);
    PROPAGATE(flushBatch(f));           /* host buckets submitted earlier keep their place in the order */
    DeviceGroup *out = nullptr, *src = nullptr;
    WorkItem *item = nullptr;
    {
        std::unique_lock<std::mutex> l(f->mutex);
        for (auto &g : f->groups)
            if (g->device == device && src == nullptr)
                src = g.get();
        REQUIRE(src != nullptr, MLSGPU_ERR_INVALID);
        for (;;)
        {
            if (f->error != MLSGPU_OK)
                return setError(f->error, "%s", f->errorText.c_str());
            out = pickGroup(f, device);
            if (out != nullptr)
                break;
            f->popCond.wait(l);
        }
        item = out->pool.front();
        out->pool.pop_front();
        out->unallocated -= numSplats;
        f->inFlightItems++;
        f->inFlightMax = std::max(f->inFlightMax, f->inFlightItems);
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
    /* tests: the peer route on one GPU.  Said once on stderr, because left set by accident it sends every device-side
     * bucket through the scratch ring */
    static const bool forcePeer = [] {
        const bool on = getenv("MLSGPU_HIP_FARM_FORCE_PEER") != nullptr;
        if (on)
            fprintf(stderr, "mlsgpu_hip: MLSGPU_HIP_FARM_FORCE_PEER is set: device-side buckets take the peer route (test hook)\n");
        return on;
    }();
    DeviceGuard restore;
    int rc = MLSGPU_OK;
    if (out->device == device && !forcePeer)
    {
        rc = mlsgpu_hip_bucket_load(out->copyCtx, dSplats, dIds, numSplats, fullGrid, item->dSplats);
        if (rc == MLSGPU_OK && hipEventRecord(item->copyEvent, out->copyStream) != hipSuccess)
            rc = setError(MLSGPU_ERR_HIP, "farm: cannot record the load event");
        /* the id list belongs to the caller: it is free again when this event has fired */
        *consumed = item->copyEvent;
    }
    else
    {
        /* gather + transform where the cloud lives, then device-to-device over the fabric */
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess && src->peerRing.empty())
        {
            src->peerRing.resize(std::max<size_t>(2, f->groups.size()));
            for (auto &ps : src->peerRing)
                if (e == hipSuccess)
                {
                    e = hipMalloc((void **) &ps.ptr, f->cfg.worker.maxBucketSplats * sizeof(mlsgpu_splat));
                    if (e == hipSuccess)
                        e = hipEventCreateWithFlags(&ps.loaded, hipEventDisableTiming);
                }
            if (e != hipSuccess)
                rc = setError(MLSGPU_ERR_NOMEM, "farm: cannot allocate the peer scratch ring on device %d", device);
        }
        DeviceGroup::PeerSlot *ps = nullptr;
        if (rc == MLSGPU_OK && e == hipSuccess)
        {
            ps = &src->peerRing[src->peerCur];
            src->peerCur = (src->peerCur + 1) % src->peerRing.size();
            /* the slot's previous bucket must have left it: ordered on the GPU, not on the host */
            if (ps->busy != nullptr)
                e = hipStreamWaitEvent(src->copyStream, ps->busy, 0);
            ps->busy = nullptr;
        }
        if (rc == MLSGPU_OK && e == hipSuccess)
            rc = mlsgpu_hip_bucket_load(src->copyCtx, dSplats, dIds, numSplats, fullGrid, ps->ptr);
        if (rc == MLSGPU_OK && e == hipSuccess)
            e = hipEventRecord(ps->loaded, src->copyStream);
        if (rc == MLSGPU_OK && e == hipSuccess)
            e = hipSetDevice(out->device);
        if (rc == MLSGPU_OK && e == hipSuccess)
            e = hipStreamWaitEvent(out->copyStream, ps->loaded, 0);
        if (rc == MLSGPU_OK && e == hipSuccess && numSplats > 0)
            e = hipMemcpyPeerAsync(item->dSplats, out->device, ps->ptr, device, numSplats * sizeof(mlsgpu_splat),
                                   out->copyStream);
        if (rc == MLSGPU_OK && e == hipSuccess)
            e = hipEventRecord(item->copyEvent, out->copyStream);
        if (rc == MLSGPU_OK && e == hipSuccess)
        {
            ps->busy = item->copyEvent;
            /* the caller's id list is free again once the gather has run; the peer copy goes on without the host */
            *consumed = ps->loaded;
        }
        if (rc == MLSGPU_OK && e != hipSuccess)
        {
            rc = setError(MLSGPU_ERR_HIP, "farm: peer route %d -> %d failed: %s", device, out->device, hipGetErrorString(e));
            hipStreamSynchronize(src->copyStream);
            hipStreamSynchronize(out->copyStream);
            if (ps != nullptr)
                ps->busy = nullptr;
        }
    }
    if (rc != MLSGPU_OK)
        return abandonItem(f, out, item, numSplats, rc);
    {
        std::lock_guard<std::mutex> l(f->mutex);
        f->stats[0]++;
        f->stats[1] += numSplats;
        f->stats[3]++;
        out->queue.push_back(item);
    }
    f->queueCond.notify_all();
    return rc;
}

MLSGPU_API int mlsgpu_hip_farm_finish(mlsgpu_farm *f)
{
    REQUIRE(f != nullptr, MLSGPU_ERR_INVALID);
    PROPAGATE(flushBatch(f));
    std::unique_lock<std::mutex> l(f->mutex);
    f->idleCond.wait(l, [&] { return (f->inFlightItems == 0 && f->hostQueue.empty()) || f->error != MLSGPU_OK; });
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

MLSGPU_API int mlsgpu_hip_farm_host_stats(mlsgpu_farm *f, uint64_t out[4])
{
    REQUIRE(f != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> l(f->mutex);
    for (int i = 0; i < 4; i++)
        out[i] = f->hostStats[i];
    return MLSGPU_OK;
}

/* Buckets of a device item the workers take through the path in lock-step (1 .. MLSGPU_MAX_BATCH; default 1: bucket by
 * bucket, the reference's loop).  Takes effect with the next item a worker picks up; every worker then holds `lanes`
 * sets of per-bucket buffers (mlsgpu_hip_worker_set_batch). */
MLSGPU_API int mlsgpu_hip_farm_set_batch(mlsgpu_farm *f, uint32_t lanes)
{
    REQUIRE(f != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(lanes >= 1 && lanes <= MLSGPU_MAX_BATCH, MLSGPU_ERR_LENGTH);
    f->batch.store(lanes);
    return MLSGPU_OK;
}

/* most device items that were in flight (taken from a pool, not yet handed back by a worker) at the same time */
MLSGPU_API int mlsgpu_hip_farm_in_flight_max(mlsgpu_farm *f, uint64_t *out)
{
    REQUIRE(f != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> l(f->mutex);
    *out = f->inFlightMax;
    return MLSGPU_OK;
}

/* ---- placement (placement.hpp) ---- */

MLSGPU_API int mlsgpu_hip_device_node(int device)
{
    const int forced = placement::overriddenDeviceNode(device);
    if (forced != -2)
        return forced;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int) sizeof(bdf), device) != hipSuccess)
    {
        (void) hipGetLastError();
        return -1;
    }
    return placement::pciNode(placement::sysfsRoot(), bdf);
}

MLSGPU_API int mlsgpu_hip_topology(uint32_t *numNodes, uint32_t cpusPerNode[16])
{
    REQUIRE(numNodes != nullptr, MLSGPU_ERR_INVALID);
    const std::vector<std::vector<int> > nodes = placement::readNodes(placement::sysfsRoot());
    *numNodes = (uint32_t) nodes.size();
    for (size_t i = 0; cpusPerNode != nullptr && i < 16; i++)
        cpusPerNode[i] = i < nodes.size() ? (uint32_t) nodes[i].size() : 0;
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_plan_copy_sides(const int32_t *deviceNodes, uint32_t numDevices, uint32_t numNodes,
                                          int32_t *sideOfDevice, int32_t *nodeOfSide, uint32_t *numSides)
{
    REQUIRE(deviceNodes != nullptr && sideOfDevice != nullptr && nodeOfSide != nullptr && numSides != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(numDevices >= 1 && numDevices <= 16, MLSGPU_ERR_INVALID);
    std::vector<int> sideOf, nodeOf;
    placement::planSides(deviceNodes, numDevices, numNodes, sideOf, nodeOf);
    for (uint32_t i = 0; i < numDevices; i++)
        sideOfDevice[i] = sideOf[i];
    for (size_t k = 0; k < nodeOf.size(); k++)
        nodeOfSide[k] = nodeOf[k];
    *numSides = (uint32_t) nodeOf.size();
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_bind_thread_to_node(int node)
{
    return placement::bindThisThread(placement::cpusOfNode(node)) ? 1 : 0;
}

/* test hook: `rounds` copies of varying sizes up to `bytes` through a CopyPool of `threads` threads bound to `node`'s CPUs,
 * each compared with its source; returns the number of mismatching copies */
MLSGPU_API int mlsgpu_hip_test_copy_pool(uint32_t threads, uint32_t rounds, uint64_t bytes, int node)
{
    placement::CopyPool pool;
    pool.start(threads, placement::cpusOfNode(node));
    std::vector<unsigned char> src(bytes), dst(bytes);
    int bad = 0;
    uint64_t x = 0x9E3779B97F4A7C15ull;
    if (rounds == 0)
    {
        /* one copy of exactly `bytes` bytes, every byte distinct from the destination's fill */
        for (uint64_t i = 0; i < bytes; i++)
            src[i] = (unsigned char) (1 + i % 151);
        std::fill(dst.begin(), dst.end(), (unsigned char) 0xA5);
        pool.copy(dst.data(), src.data(), bytes);
        return std::memcmp(dst.data(), src.data(), bytes) != 0;
    }
    for (uint32_t r = 0; r < rounds; r++)
    {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const uint64_t n = bytes == 0 ? 0 : (r % 3 == 0 ? bytes : x % (bytes + 1));
        for (uint64_t i = 0; i < n; i += 509)
            src[i] = (unsigned char) (x + i + r);
        std::fill(dst.begin(), dst.begin() + (long) n, (unsigned char) 0xA5);
        pool.copy(dst.data(), src.data(), n);
        bad += std::memcmp(dst.data(), src.data(), n) != 0;
    }
    return bad;
}

MLSGPU_API int mlsgpu_hip_farm_placement(mlsgpu_farm *f, int32_t out[100])
{
    REQUIRE(f != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    for (int i = 0; i < 100; i++)
        out[i] = -1;
    out[0] = (int32_t) f->sides.size();
    out[1] = f->ringNode;
    out[2] = (int32_t) placement::numNodes();
    out[3] = (int32_t) f->groups.size();
    for (size_t d = 0; d < f->groups.size() && d < 16; d++)
    {
        out[4 + 3 * d] = f->groups[d]->device;
        out[5 + 3 * d] = f->groups[d]->node;
        out[6 + 3 * d] = (int32_t) f->groups[d]->side;
    }
    for (size_t k = 0; k < f->sides.size() && k < 12; k++)
    {
        out[52 + 4 * k] = f->sides[k]->node;
        out[53 + 4 * k] = f->sides[k]->stagingNode;
        out[54 + 4 * k] = (int32_t) f->sides[k]->staging.size();
        out[55 + 4 * k] = (int32_t) f->sides[k]->pool.threads();
    }
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_farm_worker_clock(mlsgpu_farm *f, double out[4])
{
    REQUIRE(f != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> l(f->mutex);
    for (int i = 0; i < 4; i++)
        out[i] = f->workerClock[i];
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_farm_group_clock(mlsgpu_farm *f, uint32_t group, double out[4])
{
    REQUIRE(f != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(group < f->groups.size(), MLSGPU_ERR_LENGTH);
    std::lock_guard<std::mutex> l(f->mutex);
    for (int i = 0; i < 4; i++)
        out[i] = f->groups[group]->clock[i];
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_farm_copy_clock(mlsgpu_farm *f, double out[8])
{
    REQUIRE(f != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    /* the copies still attached to items are added without being consumed: the caller has finished the farm */
    /* the event handles are snapshotted under the mutex and queried after it is released (the workers contend for it), each
     * on its own device */
    struct Timed { int device; hipEvent_t start, stop; };
    std::vector<Timed> timed;
    {
        std::lock_guard<std::mutex> l(f->mutex);
        for (int i = 0; i < 8; i++)
            out[i] = f->copyClock[i];
        for (auto &g : f->groups)
            for (auto &item : g->items)
                if (item->copyTimed)
                    timed.push_back(Timed{g->device, item->copyStart, item->copyStop});
        out[4] = f->clockRunning ? std::chrono::duration<double>(f->lastFlush - f->firstSubmit).count() : 0.0;
    }
    int current = -1;
    (void) hipGetDevice(&current);
    for (const Timed &t : timed)
    {
        (void) hipSetDevice(t.device);
        float ms = 0.0f;
        if (hipEventQuery(t.stop) == hipSuccess && hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess)
            out[3] += ms * 1e-3;
    }
    if (current >= 0)
        (void) hipSetDevice(current);
    (void) hipGetLastError();
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
