/*
 * Header-only C++ mirror of the reference's device-path classes over the C-ABI of
 * include/mlsgpu_hip.h.  Names, argument meaning and error behaviour follow the reference so that
 * code written against SplatTreeCL / MlsFunctor / Marching / DeviceWorkerGroup
 * (src/splat_tree_cl.h, src/mls.h, src/marching.h, src/mesh.h, src/mesh_filter.h, src/workers.h)
 * ports by replacing cl::Context+cl::Device with mlsgpu::hip::Context, cl::CommandQueue with the
 * context's stream, cl::Buffer with device pointers and cl::Event chains with stream order.
 *
 * Errors: MLSGPU_ERR_LENGTH -> std::length_error, MLSGPU_ERR_INVALID -> std::invalid_argument
 * (the reference's MLSGPU_ASSERT classes, src/errors.h:41-42), anything else -> mlsgpu::hip::Error
 * (the reference's cl::Error).
 */
#ifndef MLSGPU_AMD_HOST_HPP
#define MLSGPU_AMD_HOST_HPP

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mlsgpu_hip.h"

namespace mlsgpu
{
namespace hip
{

class Error : public std::runtime_error
{
public:
    int code;
    Error(int code, const std::string &what) : std::runtime_error(what), code(code) {}
};

inline void check(int rc)
{
    if (rc == MLSGPU_OK)
        return;
    const std::string msg = mlsgpu_hip_last_error();
    switch (rc)
    {
    case MLSGPU_ERR_LENGTH: throw std::length_error(msg);
    case MLSGPU_ERR_INVALID: throw std::invalid_argument(msg);
    default: throw Error(rc, msg);
    }
}

/// One device + one in-order stream: what a worker's cl::Context + cl::CommandQueue are in the reference.
class Context
{
    mlsgpu_ctx *h;
    Context(const Context &);
    Context &operator=(const Context &);
public:
    explicit Context(int device, void *stream = NULL) : h(NULL) { check(mlsgpu_hip_ctx_create(device, stream, &h)); }
    ~Context() { mlsgpu_hip_ctx_destroy(h); }
    mlsgpu_ctx *get() const { return h; }
    void *stream() const { return mlsgpu_hip_ctx_stream(h); }
    void finish() const { check(mlsgpu_hip_ctx_synchronize(h)); }            // cl::CommandQueue::finish
    void releaseScratch() const { check(mlsgpu_hip_ctx_release_scratch(h)); }   // what Bucket::bucket keeps between calls
    void setTiming(bool on) const { check(mlsgpu_hip_ctx_set_timing(h, on)); } // --statistics-cl
};

/// cl::Buffer
template<typename T>
class Buffer
{
    const Context *ctx;
    T *ptr;
    std::size_t count;
    Buffer(const Buffer &);
    Buffer &operator=(const Buffer &);
public:
    Buffer(const Context &ctx, std::size_t count) : ctx(&ctx), ptr(NULL), count(count)
    {
        void *p = NULL;
        check(mlsgpu_hip_malloc(ctx.get(), count * sizeof(T), &p));
        ptr = static_cast<T *>(p);
    }
    ~Buffer() { mlsgpu_hip_free(ctx->get(), ptr); }
    T *get() const { return ptr; }
    std::size_t size() const { return count; }
    void write(const T *src, std::size_t n, std::size_t first = 0, bool blocking = true) const
    {
        check(mlsgpu_hip_memcpy_h2d(ctx->get(), ptr + first, src, n * sizeof(T), !blocking));
    }
    void read(T *dst, std::size_t n, std::size_t first = 0, bool blocking = true) const
    {
        check(mlsgpu_hip_memcpy_d2h(ctx->get(), dst, ptr + first, n * sizeof(T), !blocking));
    }
};

typedef mlsgpu_splat Splat;                 // src/splat.h:40-46
enum MlsShape { MLS_SHAPE_SPHERE = MLSGPU_SHAPE_SPHERE, MLS_SHAPE_PLANE = MLSGPU_SHAPE_PLANE };  // src/mls.h:47-51

/// SplatTreeCL, src/splat_tree_cl.h:216-296
class SplatTreeCL
{
    mlsgpu_tree *h;
    SplatTreeCL(const SplatTreeCL &);
    SplatTreeCL &operator=(const SplatTreeCL &);
public:
    typedef std::int32_t command_type;
    enum { MAX_LEVELS = MLSGPU_TREE_MAX_LEVELS };
    static std::uint64_t resourceUsage(std::size_t maxLevels, std::size_t maxSplats)
    {
        return mlsgpu_hip_tree_resource_usage(maxLevels, maxSplats);
    }
    SplatTreeCL(const Context &ctx, std::size_t maxLevels, std::size_t maxSplats) : h(NULL)
    {
        check(mlsgpu_hip_tree_create(ctx.get(), maxLevels, maxSplats, &h));
    }
    ~SplatTreeCL() { mlsgpu_hip_tree_destroy(h); }
    /// src/splat_tree_cl.cpp:269-335.  Ordering is the context's stream instead of cl::Event lists.
    void enqueueBuild(Splat *splats, std::size_t firstSplat, std::size_t numSplats,
                      const std::uint32_t size[3], const std::int32_t offset[3], unsigned int subsamplingShift)
    {
        check(mlsgpu_hip_tree_build(h, splats, firstSplat, numSplats, size, offset, subsamplingShift));
    }
    void clearSplats() { mlsgpu_hip_tree_clear_splats(h); }
    /* false: enqueueBuild leaves the splats untouched (no radius -> 1/radius^2); MlsFunctor::set follows the tree */
    void setMutate(bool mutate) { check(mlsgpu_hip_tree_set_mutate(h, mutate ? 1 : 0)); }
    const Splat *getSplats() const { return mlsgpu_hip_tree_splats(h); }
    const command_type *getCommands() const { return mlsgpu_hip_tree_commands(h); }
    const command_type *getStart() const { return mlsgpu_hip_tree_start(h); }
    std::size_t getNumLevels() const { return mlsgpu_hip_tree_num_levels(h); }
    mlsgpu_tree *get() const { return h; }
};

/// MeshSizes / DeviceKeyMesh / HostKeyMesh, src/mesh.h:40-179
struct DeviceKeyMesh : public mlsgpu_mesh
{
    std::size_t numExternalVertices() const { return numVertices - numInternalVertices; }
    std::size_t getHostBytes() const { return mlsgpu_hip_mesh_host_bytes(this); }
};

struct HostKeyMesh
{
    std::uint64_t *vertexKeys;      // numExternalVertices entries: vertexKeys[i] belongs to vertices[i + numInternal]
    float (*vertices)[3];
    std::uint32_t (*triangles)[3];
    std::size_t numVertices, numTriangles, numInternalVertices;

    /// Construct over an existing 8-byte aligned pool of getHostBytes() bytes (src/mesh.cpp:51-60).
    HostKeyMesh(void *ptr, const DeviceKeyMesh &sizes)
        : numVertices(sizes.numVertices), numTriangles(sizes.numTriangles), numInternalVertices(sizes.numInternalVertices)
    {
        if (reinterpret_cast<std::uintptr_t>(ptr) % 8 != 0)
            throw std::invalid_argument("HostKeyMesh: pointer is not 8-byte aligned");
        vertexKeys = static_cast<std::uint64_t *>(ptr);
        vertices = reinterpret_cast<float (*)[3]>(vertexKeys + (numVertices - numInternalVertices));
        triangles = reinterpret_cast<std::uint32_t (*)[3]>(vertices + numVertices);
    }
};

/// enqueueReadMesh, src/mesh.cpp:62-102: all three arrays, asynchronous on the context's stream.
inline void enqueueReadMesh(const Context &ctx, const DeviceKeyMesh &dMesh, HostKeyMesh &hMesh)
{
    check(mlsgpu_hip_mesh_read(ctx.get(), &dMesh, hMesh.vertexKeys, 1));
}

/// Marching, src/marching.h:204-608
class Marching
{
    mlsgpu_marching *h;
    Marching(const Marching &);
    Marching &operator=(const Marching &);
public:
    enum { MAX_CELL_BYTES = MLSGPU_MARCHING_MAX_CELL_BYTES, MAX_DIMENSION = MLSGPU_MARCHING_MAX_DIMENSION };
    typedef mlsgpu_swathe Swathe;

    /// Marching::Generator, src/marching.h:204-253
    class Generator
    {
    public:
        virtual ~Generator() {}
        virtual const std::uint32_t *alignment() const = 0;
        /// Enqueue work on `stream` that fills slices swathe.zFirst..zLast of `distance`
        /// (corner (x,y,z) at distance[(y + z*zStride + zBias) * pitch + x]).
        virtual void enqueue(void *stream, float *distance, std::size_t pitch, const Swathe &swathe) = 0;
    };

    /// Marching::OutputFunctor, src/marching.h:516-546
    typedef std::function<void(void *stream, const DeviceKeyMesh &mesh)> OutputFunctor;

    static std::uint64_t resourceUsage(std::uint32_t maxWidth, std::uint32_t maxHeight, std::uint32_t maxDepth,
                                       std::uint32_t maxSwathe, std::size_t meshMemory, const std::uint32_t alignment[3])
    {
        return mlsgpu_hip_marching_resource_usage(maxWidth, maxHeight, maxDepth, maxSwathe, meshMemory, alignment);
    }

    Marching(const Context &ctx, std::uint32_t maxWidth, std::uint32_t maxHeight, std::uint32_t maxDepth,
             std::uint32_t maxSwathe, std::size_t meshMemory, const std::uint32_t alignment[3]) : h(NULL)
    {
        check(mlsgpu_hip_marching_create(ctx.get(), maxWidth, maxHeight, maxDepth, maxSwathe, meshMemory, alignment, &h));
    }
    ~Marching() { mlsgpu_hip_marching_destroy(h); }

    /// src/marching.cpp:745-824.  Blocks until the bucket is finished, like the reference.
    void generate(Generator &generator, const OutputFunctor &output,
                  const std::uint32_t size[3], const std::uint32_t keyOffset[3])
    {
        Thunk thunk = {&generator, &output, std::exception_ptr()};
        mlsgpu_generator g;
        const std::uint32_t *a = generator.alignment();
        g.alignment[0] = a[0]; g.alignment[1] = a[1]; g.alignment[2] = a[2];
        g.enqueue = &Marching::enqueueThunk;
        g.user = &thunk;
        const int rc = mlsgpu_hip_marching_generate(h, &g, &Marching::outputThunk, &thunk, size, keyOffset);
        if (thunk.error)
            std::rethrow_exception(thunk.error);
        check(rc);
    }
    mlsgpu_marching *get() const { return h; }

private:
    struct Thunk
    {
        Generator *generator;
        const OutputFunctor *output;
        std::exception_ptr error;
    };
    static int enqueueThunk(void *user, void *stream, float *field, std::uint64_t pitch, const mlsgpu_swathe *sw)
    {
        Thunk *t = static_cast<Thunk *>(user);
        try { t->generator->enqueue(stream, field, pitch, *sw); return 0; }
        catch (...) { t->error = std::current_exception(); return MLSGPU_ERR_CALLBACK; }
    }
    static int outputThunk(void *user, void *stream, const mlsgpu_mesh *mesh)
    {
        Thunk *t = static_cast<Thunk *>(user);
        try { (*t->output)(stream, *static_cast<const DeviceKeyMesh *>(mesh)); return 0; }
        catch (...) { t->error = std::current_exception(); return MLSGPU_ERR_CALLBACK; }
    }
};

/// MlsFunctor, src/mls.h:79-170
class MlsFunctor : public Marching::Generator
{
    mlsgpu_mls *h;
    std::uint32_t wgs_[3];
public:
    static const int subsamplingMin = 3;
    MlsFunctor(const Context &ctx, MlsShape shape) : h(NULL)
    {
        check(mlsgpu_hip_mls_create(ctx.get(), shape, &h));
        wgs_[0] = wgs_[1] = wgs_[2] = 8;      // MlsFunctor::wgs, src/mls.cpp:53
    }
    ~MlsFunctor() { mlsgpu_hip_mls_destroy(h); }
    void set(const std::int32_t offset[3], const SplatTreeCL &tree, unsigned int subsamplingShift)
    {
        check(mlsgpu_hip_mls_set(h, offset, tree.get(), subsamplingShift));
    }
    void setBoundaryLimit(float limit) { check(mlsgpu_hip_mls_set_boundary_limit(h, limit)); }
    /* with explicit buffers: the splats still hold the radius, not 1/radius^2 */
    void setRawRadius(bool raw) { check(mlsgpu_hip_mls_set_raw_radius(h, raw ? 1 : 0)); }
    virtual const std::uint32_t *alignment() const { return wgs_; }
    virtual void enqueue(void *, float *distance, std::size_t pitch, const Marching::Swathe &swathe)
    {
        check(mlsgpu_hip_mls_enqueue(h, distance, pitch, UINT64_MAX, &swathe));
    }
    mlsgpu_mls *get() const { return h; }
};

/// ScaleBiasFilter, src/mesh_filter.h:117-168 (in place; returns its input like the reference's outMesh = inMesh)
class ScaleBiasFilter
{
    const Context *ctx;
    float scale, bias[3];
public:
    explicit ScaleBiasFilter(const Context &ctx) : ctx(&ctx), scale(1.0f) { bias[0] = bias[1] = bias[2] = 0.0f; }
    void setScaleBias(float s, float x, float y, float z) { scale = s; bias[0] = x; bias[1] = y; bias[2] = z; }
    void operator()(void *, const DeviceKeyMesh &inMesh, DeviceKeyMesh &outMesh) const
    {
        check(mlsgpu_hip_scale_bias(ctx->get(), &inMesh, scale, bias[0], bias[1], bias[2]));
        outMesh = inMesh;
    }
};

/// MeshFilterChain, src/mesh_filter.h:69-115
class MeshFilterChain
{
public:
    typedef std::function<void(void *stream, const DeviceKeyMesh &in, DeviceKeyMesh &out)> Filter;
    void addFilter(const Filter &f) { filters.push_back(f); }
    void setOutput(const Marching::OutputFunctor &o) { output = o; }
    void operator()(void *stream, const DeviceKeyMesh &mesh) const      // src/mesh_filter.cpp:45-66
    {
        DeviceKeyMesh a = mesh, b;
        for (std::size_t i = 0; i < filters.size(); i++)
        {
            filters[i](stream, a, b);
            a = b;
        }
        output(stream, a);
    }
private:
    std::vector<Filter> filters;
    Marching::OutputFunctor output;
};

/// A bucket grid: low extents + number of vertices per axis (the part of Grid the worker reads,
/// src/workers.cpp:237-261) and the full grid's spacing / origin for ScaleBiasFilter (:227-230).
struct BucketGrid
{
    std::int32_t low[3];
    std::uint32_t numVertices[3];
};

/**
 * DeviceWorkerGroup, src/workers.h:214-350 / src/workers.cpp:87-286, re-expressed with std::thread:
 * `numWorkers` threads, each owning a Context (stream) and an mlsgpu_worker (tree + MlsFunctor +
 * Marching + ScaleBias), fed from a pool of numWorkers + spare device splat buffers.  The producer calls
 * get() for an item, fills item->splats (its own H2D copy on item->copyStream or via write()), lists
 * the buckets in item->subItems and push()es it.
 */
class DeviceWorkerGroup
{
public:
    struct SubItem              // src/workers.h:148-160
    {
        std::uint64_t chunkId;
        BucketGrid grid;
        std::size_t firstSplat, numSplats, progressSplats;
    };
    struct WorkItem             // src/workers.h:165-181
    {
        std::vector<SubItem> subItems;
        std::unique_ptr<Buffer<Splat> > splats;
        std::size_t numSplats() const
        {
            std::size_t n = 0;
            for (std::size_t i = 0; i < subItems.size(); i++) n += subItems[i].numSplats;
            return n;
        }
    };
    /// OutputGenerator, src/workers.h:225: makes the output functor for one chunk.
    typedef std::function<Marching::OutputFunctor(std::uint64_t chunkId)> OutputGenerator;

    DeviceWorkerGroup(std::size_t numWorkers, std::size_t spare, const OutputGenerator &outputGenerator, int device,
                      std::size_t maxBucketSplats, std::uint32_t maxCells, std::size_t meshMemory,
                      int levels, int subsampling, float boundaryLimit, MlsShape shape)
        : outputGenerator(outputGenerator), device(device), maxItemSplats(maxBucketSplats), stopping(false),
          unallocated_(0), itemCtx(device)
    {
        std::memset(&cfg, 0, sizeof(cfg));
        cfg.maxBucketSplats = maxBucketSplats;
        cfg.maxCells = maxCells;
        cfg.meshMemory = meshMemory;
        cfg.levels = levels;
        cfg.subsampling = subsampling;
        cfg.boundaryLimit = boundaryLimit;
        cfg.shape = shape;
        cfg.gridSpacing = 1.0f;
        for (std::size_t i = 0; i < numWorkers + spare; i++)
        {
            std::shared_ptr<WorkItem> item(new WorkItem);
            item->splats.reset(new Buffer<Splat>(itemCtx, maxItemSplats));
            itemPool.push_back(item);
        }
        unallocated_ = maxItemSplats * (numWorkers + spare);
        this->numWorkers = numWorkers;
    }
    ~DeviceWorkerGroup() { if (!threads.empty()) stop(); }

    /// src/workers.cpp:184-205.  `lanes`: what setBatch() will be given -- every worker then holds that many sets of tree,
    /// field, lattice and mesh arena.
    static std::uint64_t resourceUsage(std::size_t numWorkers, std::size_t spare, std::size_t maxBucketSplats,
                                       std::uint32_t maxCells, std::size_t meshMemory, int levels, std::uint32_t lanes = 1)
    {
        mlsgpu_worker_config c;
        std::memset(&c, 0, sizeof(c));
        c.maxBucketSplats = maxBucketSplats; c.maxCells = maxCells; c.meshMemory = meshMemory; c.levels = levels;
        return mlsgpu_hip_worker_resource_usage_lanes(&c, lanes) * numWorkers
            + maxBucketSplats * sizeof(Splat) * (numWorkers + spare);
    }

    /// start(fullGrid): spacing and world position of vertex (0,0,0) feed ScaleBiasFilter (src/workers.cpp:124-128,227-230)
    void start(float gridSpacing, const float gridOrigin[3])
    {
        cfg.gridSpacing = gridSpacing;
        for (int i = 0; i < 3; i++) cfg.gridOrigin[i] = gridOrigin[i];
        for (std::size_t i = 0; i < numWorkers; i++)
            threads.push_back(std::thread(&DeviceWorkerGroup::run, this));
    }
    /// Buckets of a work item the workers take through the device path in lock-step (1 .. MLSGPU_MAX_BATCH; default 1 =
    /// the reference's loop over the SubItems, src/workers.cpp:232-286).  Before start(); the outputs do not change.
    void setBatch(std::uint32_t lanes) { batch = lanes; }
    /// of a batch's buckets, how many share one set of processCorners / marching launches (0: all; the octree takes the batch)
    void setMarchingGroup(std::uint32_t buckets) { marchingGroup = buckets; }
    bool canGet() { std::lock_guard<std::mutex> l(mutex); return !itemPool.empty(); }
    std::shared_ptr<WorkItem> get(std::size_t numSplats)                    // src/workers.cpp:135-146
    {
        std::unique_lock<std::mutex> l(mutex);
        poolCond.wait(l, [this] { return !itemPool.empty(); });
        std::shared_ptr<WorkItem> item = itemPool.front();
        itemPool.pop_front();
        unallocated_ -= numSplats;
        return item;
    }
    void push(const std::shared_ptr<WorkItem> &item)
    {
        itemCtx.finish();        // the producer's writes through item->splats are complete
        { std::lock_guard<std::mutex> l(mutex); queue.push_back(item); }
        queueCond.notify_one();
    }
    std::size_t unallocated() { std::lock_guard<std::mutex> l(mutex); return unallocated_; }   // src/workers.cpp:163-167
    std::size_t getMaxItemSplats() const { return maxItemSplats; }
    void stop()
    {
        { std::lock_guard<std::mutex> l(mutex); stopping = true; }
        queueCond.notify_all();
        for (std::size_t i = 0; i < threads.size(); i++) threads[i].join();
        threads.clear();
        if (workerError)
            std::rethrow_exception(workerError);
    }

private:
    void freeItem(const std::shared_ptr<WorkItem> &item)                    // src/workers.cpp:148-161
    {
        item->subItems.clear();
        { std::lock_guard<std::mutex> l(mutex); itemPool.push_back(item); }
        poolCond.notify_one();
    }
    struct OutputUser { const MeshFilterChain *chain; std::exception_ptr error; };
    struct BatchUser { std::vector<MeshFilterChain> chains; std::exception_ptr error; };
    static int batchThunk(void *user, std::uint32_t index, void *stream, const mlsgpu_mesh *mesh)
    {
        BatchUser *u = static_cast<BatchUser *>(user);
        try { u->chains[index](stream, *static_cast<const DeviceKeyMesh *>(mesh)); return 0; }
        catch (...) { u->error = std::current_exception(); return MLSGPU_ERR_CALLBACK; }
    }
    static int outputThunk(void *user, void *stream, const mlsgpu_mesh *mesh)
    {
        OutputUser *u = static_cast<OutputUser *>(user);
        try { (*u->chain)(stream, *static_cast<const DeviceKeyMesh *>(mesh)); return 0; }
        catch (...) { u->error = std::current_exception(); return MLSGPU_ERR_CALLBACK; }
    }
    void run()                                                              // Worker::operator(), src/workers.cpp:232-286
    {
        try
        {
            Context ctx(device);
            mlsgpu_worker *w = NULL;
            check(mlsgpu_hip_worker_create(ctx.get(), &cfg, &w));
            std::shared_ptr<mlsgpu_worker> guard(w, mlsgpu_hip_worker_destroy);
            if (batch > 1)
                check(mlsgpu_hip_worker_set_batch(w, batch));
            check(mlsgpu_hip_worker_set_marching_group(w, marchingGroup));
            for (;;)
            {
                std::shared_ptr<WorkItem> item;
                {
                    std::unique_lock<std::mutex> l(mutex);
                    queueCond.wait(l, [this] { return stopping || !queue.empty(); });
                    if (queue.empty())
                        return;
                    item = queue.front();
                    queue.pop_front();
                }
                if (batch > 1 && item->subItems.size() > 1)
                {
                    // the SubItems `batch` at a time through one set of launches (mlsgpu_hip_worker_process_batch)
                    BatchUser user;
                    std::vector<mlsgpu_subitem> subs(item->subItems.size());
                    user.chains.resize(subs.size());
                    std::size_t splats = 0;
                    for (std::size_t i = 0; i < subs.size(); i++)
                    {
                        const SubItem &sub = item->subItems[i];
                        user.chains[i].setOutput(outputGenerator(sub.chunkId));
                        subs[i].firstSplat = sub.firstSplat;
                        subs[i].numSplats = sub.numSplats;
                        for (int a = 0; a < 3; a++)
                        {
                            subs[i].lowExtent[a] = sub.grid.low[a];
                            subs[i].numVertices[a] = sub.grid.numVertices[a];
                        }
                        subs[i].dSplats = NULL;
                        splats += sub.numSplats;
                    }
                    const int rc = mlsgpu_hip_worker_process_batch(w, item->splats->get(), subs.data(), (std::uint32_t) subs.size(),
                                                                   &batchThunk, &user);
                    if (user.error)
                        std::rethrow_exception(user.error);
                    check(rc);
                    std::lock_guard<std::mutex> l(mutex);
                    unallocated_ += splats;
                }
                else
                for (std::size_t i = 0; i < item->subItems.size(); i++)
                {
                    const SubItem &sub = item->subItems[i];
                    MeshFilterChain chain;      // scale/bias is applied inside the worker; the chain carries the output
                    chain.setOutput(outputGenerator(sub.chunkId));
                    OutputUser user = {&chain, std::exception_ptr()};
                    const int rc = mlsgpu_hip_worker_process(w, item->splats->get(), sub.firstSplat, sub.numSplats,
                                                             sub.grid.low, sub.grid.numVertices, &outputThunk, &user);
                    if (user.error)
                        std::rethrow_exception(user.error);
                    check(rc);
                    std::lock_guard<std::mutex> l(mutex);
                    unallocated_ += sub.numSplats;
                }
                freeItem(item);
            }
        }
        catch (...)
        {
            // the reference prints and exits (src/worker_group.h:286-290); here the error surfaces from stop()
            std::lock_guard<std::mutex> l(mutex);
            if (!workerError)
                workerError = std::current_exception();
            stopping = true;
            queueCond.notify_all();
        }
    }

    OutputGenerator outputGenerator;
    int device;
    mlsgpu_worker_config cfg;
    std::size_t maxItemSplats, numWorkers;
    std::uint32_t batch = 1;
    std::uint32_t marchingGroup = 2;
    bool stopping;
    std::size_t unallocated_;
    Context itemCtx;
    std::mutex mutex;
    std::condition_variable queueCond, poolCond;
    std::deque<std::shared_ptr<WorkItem> > queue, itemPool;
    std::vector<std::thread> threads;
    std::exception_ptr workerError;
};

/**
 * MesherBase / OOCMesher (src/mesher.h:203-420) for meshes that stay on the device: functor() gives the
 * Marching::OutputFunctor that appends a ship-out, write() welds, prunes and writes one PLY per non-empty chunk.
 * One pass (numPasses() == 1).  The namer maps a chunk id to a file name (TrivialNamer / ChunkNamer, :136-182).
 */
class DeviceMesher
{
    mlsgpu_mesher *h;
    const Context *ctx;
    DeviceMesher(const DeviceMesher &);
    DeviceMesher &operator=(const DeviceMesher &);
public:
    typedef std::function<std::string(std::uint64_t chunkId)> Namer;

    explicit DeviceMesher(const Context &ctx) : h(NULL), ctx(&ctx) { check(mlsgpu_hip_mesher_create(ctx.get(), &h)); }
    ~DeviceMesher() { mlsgpu_hip_mesher_destroy(h); }
    mlsgpu_mesher *get() const { return h; }
    unsigned int numPasses() const { return 1; }
    void setPruneThreshold(double threshold) { check(mlsgpu_hip_mesher_set_prune_threshold(h, threshold)); }
    void reserve(std::uint64_t vertices, std::uint64_t triangles, std::uint64_t external)
    {
        check(mlsgpu_hip_mesher_reserve(h, vertices, triangles, external));
    }
    /// The output functor of one chunk for the worker that owns `from` (thread safe across workers).
    Marching::OutputFunctor functor(const Context &from, std::uint64_t chunkId)
    {
        mlsgpu_mesher *mesher = h;
        mlsgpu_ctx *fromCtx = from.get();
        return [mesher, fromCtx, chunkId](void *, const DeviceKeyMesh &mesh) { check(mlsgpu_hip_mesher_add(mesher, fromCtx, chunkId, &mesh)); };
    }
    /// pinned host memory a write may hold (two buffers of half this; 0 = 64 MiB), whatever the size of the mesh
    std::uint64_t writeBufferBytes = 0;
    /// MesherBase::write: returns the number of files written.
    std::size_t write(const Namer &namer, const std::vector<std::string> &comments = std::vector<std::string>())
    {
        std::uint32_t chunks = 0;
        check(mlsgpu_hip_mesher_finalize(h, &chunks));
        std::vector<const char *> cstr;
        for (const std::string &c : comments)
            cstr.push_back(c.c_str());
        for (std::uint32_t i = 0; i < chunks; i++)
        {
            std::uint64_t id, nv, nt;
            const float *dV;
            const std::uint32_t *dT;
            check(mlsgpu_hip_mesher_chunk(h, i, &id, &nv, &nt, &dV, &dT));
            // straight from HBM through a bounded pinned buffer (the reference's asynchronous writer, src/async_io.h)
            check(mlsgpu_hip_mesher_write_ply(h, i, namer(id).c_str(), cstr.empty() ? NULL : cstr.data(),
                                              (std::uint32_t) cstr.size(), writeBufferBytes));
        }
        return chunks;
    }
    void getStatistics(std::uint64_t out[8]) const { check(mlsgpu_hip_mesher_stats(h, out)); }
};

/**
 * OOCMesher (src/mesher.h:331-420, src/mesher.cpp:220-469) in host memory: the reference's weld, run where the reference
 * runs it.  add() is MesherBase::InputFunctor for a HostKeyMesh; meshes of any GPU may be added (it is what
 * BucketFarm::setHostOutput feeds from its mesher thread).  write() welds, prunes and writes one PLY per non-empty
 * chunk, like DeviceMesher::write.
 */
class OOCMesher
{
    mlsgpu_host_mesher *h;
    OOCMesher(const OOCMesher &);
    OOCMesher &operator=(const OOCMesher &);
public:
    typedef std::function<std::string(std::uint64_t chunkId)> Namer;
    OOCMesher() : h(NULL) { check(mlsgpu_hip_host_mesher_create(&h)); }
    ~OOCMesher() { mlsgpu_hip_host_mesher_destroy(h); }
    mlsgpu_host_mesher *get() const { return h; }
    void setPruneThreshold(double t) { check(mlsgpu_hip_host_mesher_set_prune_threshold(h, t)); }
    /// threads of the welder (0 = default); before the first add
    void setThreads(std::uint32_t threads) { check(mlsgpu_hip_host_mesher_set_threads(h, threads)); }
    /// the welder's threads on one NUMA node (the one its meshes arrive on); -1 = unbound.  Before the first add.
    void setNode(int node) { check(mlsgpu_hip_host_mesher_set_node(h, node)); }
    int node() const { return mlsgpu_hip_host_mesher_node(h); }
    /// the reference's --tmp-dir (TmpWriterWorkerGroup, src/mesher.cpp:404-419): blocks live in temporary files there and
    /// leave memory once welded, beyond `residentBytes` of them.  Before the first add.
    void setTmpDir(const std::string &dir, std::uint64_t residentBytes = 0)
    {
        check(mlsgpu_hip_host_mesher_set_tmp_dir(h, dir.c_str(), residentBytes));
    }
    void add(std::uint64_t chunkId, const mlsgpu_host_mesh &mesh) { check(mlsgpu_hip_host_mesher_add(h, chunkId, &mesh)); }
    std::size_t write(const Namer &namer, const std::vector<std::string> &comments = std::vector<std::string>())
    {
        std::uint32_t chunks = 0;
        check(mlsgpu_hip_host_mesher_finalize(h, &chunks));
        std::vector<const char *> cstr;
        for (const std::string &c : comments)
            cstr.push_back(c.c_str());
        for (std::uint32_t i = 0; i < chunks; i++)
        {
            std::uint64_t id, nv, nt;
            const float *v;
            const std::uint32_t *t;
            check(mlsgpu_hip_host_mesher_chunk(h, i, &id, &nv, &nt, &v, &t));
            check(mlsgpu_hip_write_ply(namer(id).c_str(), v, nv, t, nt, cstr.empty() ? NULL : cstr.data(),
                                       (std::uint32_t) cstr.size()));
        }
        return chunks;
    }
    void getStatistics(std::uint64_t out[8]) const { check(mlsgpu_hip_host_mesher_stats(h, out)); }
};

/**
 * SplatSet::FileSet (src/splat_set.h:383-700): several PLY files read as one splat sequence; load() is the bounded-memory
 * route into HBM (reader threads + pinned quarters + H2D overlap).
 */
class FileSet
{
    mlsgpu_fileset *h;
    FileSet(const FileSet &);
    FileSet &operator=(const FileSet &);
public:
    FileSet(float smooth, float maxRadius) : h(NULL) { check(mlsgpu_hip_fileset_create(smooth, maxRadius, &h)); }
    ~FileSet() { mlsgpu_hip_fileset_destroy(h); }
    void addFile(const std::string &path) { check(mlsgpu_hip_fileset_add_file(h, path.c_str())); }
    void setBufferSize(std::uint64_t bytes) { check(mlsgpu_hip_fileset_set_buffer_size(h, bytes)); }
    std::uint64_t maxSplats() const { return mlsgpu_hip_fileset_num_splats(h); }
    mlsgpu_fileset *get() const { return h; }
    /// FastBlobSet::getBoundingGrid for a set that need not fit the device: one pass over the files
    mlsgpu_grid boundingGrid(const Context &ctx, float spacing, std::uint32_t bucketSize, std::uint64_t chunkSplats,
                             unsigned readerThreads = 0)
    {
        mlsgpu_grid g;
        check(mlsgpu_hip_fileset_bounding_grid(h, ctx.get(), spacing, bucketSize, chunkSplats, readerThreads, &g));
        return g;
    }
    void read(std::uint64_t first, std::uint64_t count, Splat *out) { check(mlsgpu_hip_fileset_read(h, first, count, out)); }
    void load(const Context &ctx, const Buffer<Splat> &out, std::uint64_t first, std::uint64_t count, unsigned readerThreads = 0)
    {
        check(mlsgpu_hip_fileset_load(h, ctx.get(), first, count, out.get(), readerThreads));
    }
};

/**
 * CopyGroup + one DeviceWorkerGroup per GPU inside the library (mlsgpu_hip_farm_*; src/workers.h:214-438,
 * src/mlsgpu_core.cpp:704-741).  Ship-outs go to a DeviceMesher (appended in HBM, by peer copy from other GPUs) or, with
 * setHostOutput, through the pinned circular buffer to an OOCMesher on the farm's mesher thread (the reference's route,
 * src/workers.h:488-509).
 */
class BucketFarm
{
    mlsgpu_farm *h;
    BucketFarm(const BucketFarm &);
    BucketFarm &operator=(const BucketFarm &);
public:
    BucketFarm(const std::vector<std::int32_t> &devices, const mlsgpu_worker_config &worker, unsigned workersPerDevice,
               unsigned spare, DeviceMesher *deviceSink = NULL) : h(NULL)
    {
        mlsgpu_farm_config cfg;
        std::memset(&cfg, 0, sizeof(cfg));
        cfg.numDevices = (std::uint32_t) devices.size();
        cfg.devices = devices.data();
        cfg.workersPerDevice = workersPerDevice;
        cfg.spare = spare;
        cfg.worker = worker;
        check(mlsgpu_hip_farm_create(&cfg, deviceSink ? &mlsgpu_hip_mesher_farm_output : NULL,
                                     deviceSink ? deviceSink->get() : NULL, &h));
    }
    ~BucketFarm() { mlsgpu_hip_farm_destroy(h); }
    mlsgpu_farm *get() const { return h; }
    void setHostOutput(std::uint64_t ringBytes, OOCMesher &mesher)
    {
        check(mlsgpu_hip_farm_set_host_output(h, ringBytes, &mlsgpu_hip_host_mesher_farm_output, mesher.get()));
        // a welder that has not been placed joins the ring's NUMA node (ignored once its threads exist)
        std::int32_t where[100];
        if (mesher.node() < 0 && mlsgpu_hip_farm_placement(h, where) == MLSGPU_OK && where[1] >= 0)
            (void) mlsgpu_hip_host_mesher_set_node(mesher.get(), where[1]);
    }
    Splat *acquire(std::uint64_t numSplats)                                 // CopyGroup::get
    {
        Splat *p = NULL;
        check(mlsgpu_hip_farm_acquire(h, numSplats, &p));
        return p;
    }
    void push(std::uint64_t numSplats, const BucketGrid &grid, std::uint64_t chunkId)     // CopyGroup::push
    {
        check(mlsgpu_hip_farm_push(h, numSplats, grid.low, grid.numVertices, chunkId));
    }
    void submitDevice(int device, const Buffer<Splat> &cloud, const mlsgpu_bucket &bin, const mlsgpu_grid &fullGrid,
                      std::uint64_t chunkId)
    {
        BucketGrid g;
        for (int i = 0; i < 3; i++)
        {
            g.low[i] = bin.extents[2 * i] - fullGrid.extents[2 * i];
            g.numVertices[i] = (std::uint32_t) (bin.extents[2 * i + 1] - bin.extents[2 * i] + 1);
        }
        // a resident cloud: only enqueued -- the bucketer orders the reuse of the bin's id list behind the gather on the GPU
        // (mlsgpu_bucket::consumed), so the leaves of a level reach the farm back to back and its workers batch them
        if (bin.consumed != NULL)
            check(mlsgpu_hip_farm_submit_device_async(h, device, cloud.get(), bin.dIds, bin.numSplats, &fullGrid, g.low,
                                                      g.numVertices, chunkId, bin.consumed));
        else
            check(mlsgpu_hip_farm_submit_device(h, device, cloud.get(), bin.dIds, bin.numSplats, &fullGrid, g.low, g.numVertices,
                                                chunkId));
    }
    /// the same for a bin of Bucket::bucketStream: its splats are in the batch that is resident during the callback
    void submitDevice(int device, const mlsgpu_bucket &bin, const mlsgpu_grid &fullGrid, std::uint64_t chunkId)
    {
        BucketGrid g;
        for (int i = 0; i < 3; i++)
        {
            g.low[i] = bin.extents[2 * i] - fullGrid.extents[2 * i];
            g.numVertices[i] = (std::uint32_t) (bin.extents[2 * i + 1] - bin.extents[2 * i] + 1);
        }
        check(mlsgpu_hip_farm_submit_device(h, device, bin.dSplats, bin.dIds, bin.numSplats, &fullGrid, g.low, g.numVertices,
                                            chunkId));
    }
    void finish() { check(mlsgpu_hip_farm_finish(h)); }
};

/**
 * Bucket::bucket (src/bucket.h:116-180) for a cloud that is resident on the device, and the device half of
 * BucketLoader (src/bucket_loader.cpp:77-102).  The processor receives the bucket's grid, recursion state and its
 * splat ids (device memory, ascending, valid during the call) where the reference's receives a splat subset.
 */
namespace Bucket
{

typedef mlsgpu_grid Grid;                   // src/grid.h: reference, spacing, per-axis [first, second) extents
typedef mlsgpu_bucket Bin;                  // extents, Recursion::chunk / depth, numSplats, dIds

/// Bucket::DensityError, src/bucket.h:52-65
class DensityError : public std::runtime_error
{
    std::uint64_t cellSplats;
public:
    explicit DensityError(std::uint64_t cellSplats)
        : std::runtime_error("Too many splats covering one cell"), cellSplats(cellSplats) {}
    std::uint64_t getCellSplats() const { return cellSplats; }
};

typedef std::function<void(const Bin &)> Processor;

namespace detail
{
struct Thunk
{
    const Processor *process;
    std::exception_ptr error;
    static int call(void *user, mlsgpu_ctx *, const mlsgpu_bucket *b)
    {
        Thunk *self = static_cast<Thunk *>(user);
        try
        {
            (*self->process)(*b);
            return 0;
        }
        catch (...)
        {
            self->error = std::current_exception();
            return 1;
        }
    }
};
} // namespace detail

inline void bucket(const Context &ctx, const Buffer<Splat> &splats, std::uint64_t numSplats, const Grid &region,
                   std::uint64_t maxSplats, std::uint32_t maxCells, std::uint32_t chunkCells, std::uint32_t microCells,
                   std::uint64_t maxSplit, const Processor &process)
{
    mlsgpu_bucket_params p;
    p.maxSplats = maxSplats;
    p.maxCells = maxCells;
    p.chunkCells = chunkCells;
    p.microCells = microCells;
    p.maxSplit = maxSplit;
    detail::Thunk thunk{&process, std::exception_ptr()};
    std::uint64_t cellSplats = 0;
    const int rc = mlsgpu_hip_bucket(ctx.get(), splats.get(), numSplats, &region, &p, &detail::Thunk::call, &thunk, &cellSplats);
    if (thunk.error)
        std::rethrow_exception(thunk.error);
    if (rc == MLSGPU_ERR_DENSITY)
        throw DensityError(cellSplats);
    check(rc);
}

/// Bucket::bucket over a FileSet that need NOT fit the device (mlsgpu_hip_bucket_stream): the files are streamed through a
/// chunk buffer, once to count and once per batch of top-level regions that fit `budgetSplats`; bin.dSplats is the batch
/// the ids index, valid during the call.  Same bins as bucket() makes of the set resident.
inline void bucketStream(const Context &ctx, FileSet &files, const Grid &region, std::uint64_t maxSplats, std::uint32_t maxCells,
                         std::uint32_t chunkCells, std::uint32_t microCells, std::uint64_t maxSplit, std::uint64_t budgetSplats,
                         std::uint64_t chunkSplats, unsigned readerThreads, const Processor &process, std::uint64_t stats[4] = NULL)
{
    mlsgpu_bucket_params p;
    p.maxSplats = maxSplats;
    p.maxCells = maxCells;
    p.chunkCells = chunkCells;
    p.microCells = microCells;
    p.maxSplit = maxSplit;
    detail::Thunk thunk{&process, std::exception_ptr()};
    std::uint64_t cellSplats = 0;
    const int rc = mlsgpu_hip_bucket_stream(ctx.get(), files.get(), &region, &p, budgetSplats, chunkSplats, readerThreads,
                                            &detail::Thunk::call, &thunk, &cellSplats, stats);
    if (thunk.error)
        std::rethrow_exception(thunk.error);
    if (rc == MLSGPU_ERR_DENSITY)
        throw DensityError(cellSplats);
    check(rc);
}

/// Gathers a bin's splats into `out` in the full grid's vertex coordinates and returns the sub-grid the worker needs
/// (BucketLoader::operator(), src/bucket_loader.cpp:77-102).
inline BucketGrid load(const Context &ctx, const Buffer<Splat> &splats, const Bin &bin, const Grid &fullGrid,
                       const Buffer<Splat> &out)
{
    check(mlsgpu_hip_bucket_load(ctx.get(), splats.get(), bin.dIds, bin.numSplats, &fullGrid, out.get()));
    BucketGrid g;
    for (int i = 0; i < 3; i++)
    {
        g.low[i] = bin.extents[2 * i] - fullGrid.extents[2 * i];
        g.numVertices[i] = (std::uint32_t) (bin.extents[2 * i + 1] - bin.extents[2 * i] + 1);
    }
    return g;
}

} // namespace Bucket

} // namespace hip
} // namespace mlsgpu

#endif /* MLSGPU_AMD_HOST_HPP */
