/*
 * Host mesh sink: the weld OOCMesher performs on the host (src/mesher.cpp:220-469).  This is the
 * north_star's "welding stays on host" route and the cross-GPU welder: ship-outs of any device reach it through the
 * bucket farm's pinned circular buffer (mlsgpu_hip_farm_set_host_output).
 *
 * What the reference does per block on its ONE mesher thread (OOCMesher::add, :370-469) -- union-find over two edges per
 * triangle for the block's local components ("clumps", computeLocalComponents :220-236, updateGlobalClumps :238-281),
 * and for every external key seen before a merge of the two clumps and one vertex un-counted (updateClumpKeyMap :286-311)
 * -- is here a TASK per block on a pool of threads (the reference parallelises its heavy rewrite with OpenMP,
 * src/mesher.cpp:597-600; with eight GPUs behind one welder a single thread is the whole job's limit, as its manual says,
 * doc/mlsgpu-user-manual.xml:508-511).  add() copies the block out of the caller's memory (the farm's ring slot is free
 * again when it returns), checks its indices and queues the task; nothing a task does depends on another block:
 *   - the block's components and their vertex / triangle counts are local;
 *   - the key map is SHARDED by key hash (a lock per shard): the first clump seen for a key, every later occurrence as a
 *     (clump, clump) record -- one union and one un-counted vertex each, applied in finalize -- and per (key, chunk) the
 *     first occurrence in ARRIVAL order, which is the vertex externalRemap (:538-567) keeps in that chunk's file.
 * Clump ids, union records and owners are expressed in (arrival number, index inside the block), so the result does not
 * depend on which thread ran what when.  finalize() is the part of MesherBase::write that precedes the file output
 * (:763-852): the unions, component sizes, the prune threshold uint64(total * threshold) with the `>=` keep test
 * (getStatistics :491-536), and one mesh per chunk in which a key occurs once, built by the pool block by block (three
 * passes: what is emitted, vertex ranks and copies, aliases and triangles).  Differences from the reference, all invisible
 * up to the isomorphism its own tests compare by (test/test_mesher.cpp:401-460): no writer / reader threads of our own and no
 * reorder buffer (the arena is host memory, or -- mlsgpu_hip_host_mesher_set_tmp_dir, the reference's --tmp-dir -- mappings of
 * temporary files that the kernel writes out and drops once a block is welded: a mesh larger than host memory has a path),
 * output order is (chunk by first arrival, block arrival, order inside the block) exactly as the device sink's (mesher.hip),
 * so the two sinks can be compared element for element.
 *
 * Host code only; it lives in the HIP library because the farm's mesher thread calls it.
 */
#include "common.hpp"
#include "placement.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_map>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

using namespace mlsgpu;

namespace
{

/* ---------------------------------------------------------------- memory
 *
 * A job moves hundreds of MB per step through the welder (the shells cloud of BASELINE configs[2]: 17 M vertices and 35 M
 * triangles, 0.6 GB in and as much out), and freshly mapped memory costs a page fault per 4 KB -- about 1 GB/s per thread,
 * several times the copies themselves.  Everything large therefore comes from slabs (mmap, transparent huge pages asked
 * for) that a bump allocator hands out and that go back to a process-wide cache when the mesher is destroyed: the next
 * job's welder writes into memory that is already mapped. */
struct Slab
{
    char *base = nullptr;
    size_t cap = 0;
    bool landing = false;       /* a slab ship-outs land in: registered with the HIP runtime (page-locked) where that works */
    bool registered = false;
    bool file = false;          /* a mapping of an unlinked temporary file (mlsgpu_hip_host_mesher_set_tmp_dir): never cached */
};

/* Bounded-memory mode: the slab is a shared mapping of a temporary file that has no name any more.  What the welder writes
 * there is the kernel's to write back and to drop from memory, and comes back with a page fault when the output passes
 * read it -- OOCMesher's temporary vertex / triangle files (src/mesher.cpp:404-419, 763-852) without a reader and a writer
 * of our own.  Returns an empty slab when the directory does not take the file. */
std::atomic<uint64_t> tmpBytesMapped{0};

Slab takeFileSlab(size_t atLeast, const std::string &dir)
{
    Slab s;
    const size_t cap = (atLeast + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);
    int fd = open(dir.c_str(), O_TMPFILE | O_RDWR | O_CLOEXEC, 0600);
    if (fd < 0)
    {
        /* a file system without O_TMPFILE: a named file, unlinked at once */
        std::string name = dir + "/mlsgpu-welder-XXXXXX";
        fd = mkstemp(&name[0]);
        if (fd < 0)
            return s;
        unlink(name.c_str());
    }
    if (ftruncate(fd, (off_t) cap) != 0)
    {
        close(fd);
        return s;
    }
    void *p = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);                  /* the mapping keeps the file */
    if (p == MAP_FAILED)
        return s;
    s.base = static_cast<char *>(p);
    s.cap = cap;
    s.file = true;
    tmpBytesMapped += cap;
    return s;
}

class SlabCache
{
public:
    static SlabCache &instance()
    {
        static SlabCache c;
        return c;
    }
    Slab take(size_t atLeast, bool landing = false)
    {
        {
            std::lock_guard<std::mutex> l(mutex);
            for (size_t i = 0; i < free.size(); i++)
                if (free[i].cap >= atLeast && free[i].landing == landing)
                {
                    Slab s = free[i];
                    free.erase(free.begin() + (long) i);
                    held -= s.cap;
                    return s;
                }
        }
        Slab s;
        s.cap = (atLeast + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);
        freshSlabs++;
        freshBytes += s.cap;
        void *p = mmap(nullptr, s.cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED)
            return Slab();
        (void) madvise(p, s.cap, MADV_HUGEPAGE);
        s.base = static_cast<char *>(p);
        s.landing = landing;
        if (landing)
        {
            /* page-locked, so that a device-to-host copy lands here at the link's rate; a process without a GPU (the CPU tests)
             * keeps the slab unregistered: the route works, only slower */
            std::memset(p, 0, s.cap);           /* (touched here: on the caller's NUMA node, before the pages are locked) */
            s.registered = hipHostRegister(p, s.cap, hipHostRegisterPortable) == hipSuccess;
            if (!s.registered)
                (void) hipGetLastError();
        }
        return s;
    }
    static void unmap(Slab &s)
    {
        if (s.registered)
            (void) hipHostUnregister(s.base);
        munmap(s.base, s.cap);
    }
    void give(Slab s)
    {
        if (s.base == nullptr)
            return;
        if (s.file)
        {
            munmap(s.base, s.cap);      /* the file goes with its last mapping */
            return;
        }
        {
            std::lock_guard<std::mutex> l(mutex);
            if (held + s.cap <= limit)
            {
                free.push_back(s);
                held += s.cap;
                return;
            }
        }
        unmap(s);
    }
    ~SlabCache()
    {
        for (Slab &s : free)
            munmap(s.base, s.cap);      /* (process exit: the HIP runtime may be gone already, nothing is unregistered) */
    }
    /* how much mapped memory stays with the process between jobs; what is held beyond the new limit goes back now */
    size_t setLimit(size_t bytes)
    {
        std::vector<Slab> drop;
        {
            std::lock_guard<std::mutex> l(mutex);
            limit = bytes;
            while (held > limit && !free.empty())
            {
                drop.push_back(free.back());
                held -= free.back().cap;
                free.pop_back();
            }
        }
        size_t released = 0;
        for (Slab &s : drop)
        {
            released += s.cap;
            unmap(s);
        }
        return released;
    }
    size_t getLimit()
    {
        std::lock_guard<std::mutex> l(mutex);
        return limit;
    }
    std::atomic<uint64_t> freshSlabs{0}, freshBytes{0};     /* mappings made because no kept slab was large enough */

private:
    SlabCache()
    {
        /* MLSGPU_HIP_WELDER_CACHE_MB: the limit of a process that cannot call mlsgpu_hip_host_mesher_trim_cache */
        if (const char *e = getenv("MLSGPU_HIP_WELDER_CACHE_MB"))
            limit = (size_t) strtoull(e, nullptr, 10) << 20;
    }
    std::mutex mutex;
    std::vector<Slab> free;
    size_t held = 0;
    size_t limit = size_t(16) << 30;            /* mapped memory kept between jobs */
};

class Arena
{
public:
    explicit Arena(bool landing_ = false) : landing(landing_) {}
    ~Arena() { release(); }
    /* every slab from now on is a mapping of a temporary file in `dir` (and no landing slab is page-locked) */
    void setTmpDir(const std::string &dir)
    {
        std::lock_guard<std::mutex> l(mutex);
        tmpDir = dir;
    }
    /* bytes of the arena's file-backed slabs, and how many of them are in memory right now (mincore) */
    void fileUsage(uint64_t &mapped, uint64_t &resident)
    {
        std::lock_guard<std::mutex> l(mutex);
        const size_t page = (size_t) sysconf(_SC_PAGESIZE);
        std::vector<unsigned char> vec;
        for (const Slab &s : slabs)
            if (s.file)
            {
                mapped += s.cap;
                vec.resize(s.cap / page);
                if (mincore(s.base, s.cap, vec.data()) == 0)
                    for (unsigned char c : vec)
                        resident += (c & 1u) ? page : 0;
            }
    }
    bool contains(const void *p, size_t bytes)
    {
        std::lock_guard<std::mutex> l(mutex);
        for (const Slab &s : slabs)
            if (static_cast<const char *>(p) >= s.base && static_cast<const char *>(p) + bytes <= s.base + s.cap)
                return true;
        return false;
    }
    bool allRegistered()
    {
        std::lock_guard<std::mutex> l(mutex);
        for (const Slab &s : slabs)
            if (!s.registered)
                return false;
        return true;
    }
    void release()
    {
        for (Slab &s : slabs)
            SlabCache::instance().give(s);
        slabs.clear();
        used = 0;
    }
    /* 64-byte aligned, uninitialised; nullptr when the system has no memory left */
    void *alloc(size_t bytes)
    {
        bytes = (bytes + 63) & ~size_t(63);
        std::lock_guard<std::mutex> l(mutex);
        if (slabs.empty() || used + bytes > slabs.back().cap)
        {
            Slab s = tmpDir.empty() ? SlabCache::instance().take(std::max(bytes, size_t(256) << 20), landing)
                                    : takeFileSlab(std::max(bytes, size_t(256) << 20), tmpDir);
            if (s.base == nullptr)
                return nullptr;
            slabs.push_back(s);
            used = 0;
        }
        void *p = slabs.back().base + used;
        used += bytes;
        return p;
    }
    template<typename T>
    T *array(size_t n) { return static_cast<T *>(alloc(std::max<size_t>(n, 1) * sizeof(T))); }
    /* trace: where the arena's slabs are and how much of each the kernel backs with huge pages (/proc/self/smaps) */
    std::string describe()
    {
        std::lock_guard<std::mutex> l(mutex);
        std::string out;
        for (const Slab &s : slabs)
        {
            unsigned long huge = 0;
            FILE *f = fopen("/proc/self/smaps", "r");
            if (f != nullptr)
            {
                char line[256];
                bool mine = false;
                while (fgets(line, sizeof(line), f) != nullptr)
                {
                    unsigned long lo, hi;
                    if (sscanf(line, "%lx-%lx ", &lo, &hi) == 2 && strchr(line, '-') != nullptr && line[0] != ' ' && strstr(line, "kB") == nullptr)
                        mine = lo <= (unsigned long) (uintptr_t) s.base && (unsigned long) (uintptr_t) s.base < hi;
                    else if (mine && strncmp(line, "AnonHugePages:", 14) == 0)
                    {
                        unsigned long kb = 0;
                        sscanf(line + 14, "%lu", &kb);
                        huge += kb;
                    }
                }
                fclose(f);
            }
            char text[96];
            snprintf(text, sizeof(text), " [%zu MB node %d/%d huge %lu MB]", s.cap >> 20, placement::nodeOfAddress(s.base),
                     placement::nodeOfAddress(s.base + s.cap / 2), huge >> 10);
            out += text;
        }
        return out;
    }

private:
    std::mutex mutex;
    std::vector<Slab> slabs;
    size_t used = 0;
    bool landing = false;
    std::string tmpDir;
};

/* ---------------------------------------------------------------- a small pool: tasks and parallel loops */

class Pool
{
public:
    /* `cpus`: the CPUs the pool's threads are bound to (empty: wherever the scheduler puts them) */
    explicit Pool(unsigned threads, const std::vector<int> &cpus = std::vector<int>())
    {
        for (unsigned i = 0; i < threads; i++)
            workers.emplace_back([this, cpus] {
                placement::bindThisThread(cpus);
                run();
            });
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> l(mutex);
            stopping = true;
        }
        wake.notify_all();
        for (std::thread &t : workers)
            t.join();
    }
    unsigned size() const { return (unsigned) workers.size(); }
    void submit(std::function<void()> fn)
    {
        {
            std::lock_guard<std::mutex> l(mutex);
            queue.push_back(std::move(fn));
            pending++;
        }
        wake.notify_one();
    }
    void drain()
    {
        std::unique_lock<std::mutex> l(mutex);
        idle.wait(l, [this] { return pending == 0; });
    }
    /* fn(i) for i in [0, n), dynamically scheduled; returns when all are done.  Must not be called from a pool thread. */
    void parallelFor(size_t n, const std::function<void(size_t)> &fn)
    {
        if (n == 0)
            return;
        if (workers.empty() || n == 1)
        {
            for (size_t i = 0; i < n; i++)
                fn(i);
            return;
        }
        std::atomic<size_t> next{0};
        const size_t lanes = std::min<size_t>(workers.size(), n);
        for (size_t t = 0; t < lanes; t++)
            submit([&next, n, &fn] {
                for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1))
                    fn(i);
            });
        drain();
    }

    /* The same for a caller that must not wait for unrelated tasks (add() copies a block while older blocks' tasks are
     * queued): the pieces go to the FRONT of the queue, the caller works on them too and waits for these pieces only. */
    void parallelForNow(size_t n, const std::function<void(size_t)> &fn)
    {
        if (n == 0)
            return;
        struct Shared
        {
            std::atomic<size_t> next{0}, done{0};
        };
        auto sh = std::make_shared<Shared>();
        const size_t helpers = std::min<size_t>(workers.size(), n > 1 ? n - 1 : 0);
        const std::function<void(size_t)> *fnp = &fn;
        auto body = [sh, n, fnp] {
            for (size_t i = sh->next.fetch_add(1); i < n; i = sh->next.fetch_add(1))
            {
                (*fnp)(i);          /* i < n: the caller is still waiting for this piece, fn is alive */
                sh->done.fetch_add(1);
            }
        };
        {
            std::lock_guard<std::mutex> l(mutex);
            for (size_t t = 0; t < helpers; t++)
            {
                queue.push_front(body);     /* a helper that arrives late finds no piece left and never touches fn */
                pending++;
            }
        }
        for (size_t t = 0; t < helpers; t++)
            wake.notify_one();              /* as many threads as there are pieces for, not the whole pool */
        body();
        while (sh->done.load() < n)
            std::this_thread::yield();
    }

private:
    void run()
    {
        for (;;)
        {
            std::function<void()> fn;
            {
                std::unique_lock<std::mutex> l(mutex);
                wake.wait(l, [this] { return stopping || !queue.empty(); });
                if (queue.empty())
                    return;
                fn = std::move(queue.front());
                queue.pop_front();
            }
            fn();
            {
                std::lock_guard<std::mutex> l(mutex);
                pending--;
                if (pending == 0)
                    idle.notify_all();
            }
        }
    }
    std::vector<std::thread> workers;
    std::deque<std::function<void()> > queue;
    std::mutex mutex;
    std::condition_variable wake, idle;
    size_t pending = 0;
    bool stopping = false;
};

/* ---------------------------------------------------------------- open-addressing tables (one per shard) */

static inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

/* a vertex of the job: block (arrival number) and index inside it -- or a clump: block and local clump number */
struct Ref
{
    uint32_t seq, idx;
};
static inline bool refLess(Ref a, Ref b) { return a.seq != b.seq ? a.seq < b.seq : a.idx < b.idx; }

const uint64_t EMPTY_KEY = ~uint64_t(0);    /* never a vertex key: the fields are 21 bits each */

/* key -> the first clump seen holding the vertex (clumpIdMap, src/mesher.h) */
struct ClumpTable
{
    struct Slot { uint64_t key; Ref clump; };
    std::vector<Slot> slots;
    size_t used = 0;
    ClumpTable() : slots(1024, Slot{EMPTY_KEY, Ref{0, 0}}) {}
    void grow()
    {
        std::vector<Slot> old;
        old.swap(slots);
        slots.assign(old.size() * 2, Slot{EMPTY_KEY, Ref{0, 0}});
        for (const Slot &s : old)
            if (s.key != EMPTY_KEY)
                *find(s.key) = s;
    }
    Slot *find(uint64_t key)        /* the key's slot, or the empty one it belongs in */
    {
        const size_t mask = slots.size() - 1;
        size_t i = (size_t) (mix64(key) >> 7) & mask;
        while (slots[i].key != EMPTY_KEY && slots[i].key != key)
            i = (i + 1) & mask;
        return &slots[i];
    }
    /* true: the key was new */
    bool insert(uint64_t key, Ref clump, Ref *existing)
    {
        if ((used + 1) * 10 > slots.size() * 7)
            grow();
        Slot *s = find(key);
        if (s->key == key)
        {
            *existing = s->clump;
            return false;
        }
        s->key = key;
        s->clump = clump;
        used++;
        return true;
    }
};

/* (key, chunk) -> the key's first vertex of that chunk in arrival order: the one externalRemap keeps (src/mesher.cpp:538-567) */
struct OwnerTable
{
    struct Slot { uint64_t key; uint32_t chunk; Ref owner; };
    std::vector<Slot> slots;
    size_t used = 0;
    OwnerTable() : slots(1024, Slot{EMPTY_KEY, 0, Ref{0, 0}}) {}
    static size_t hash(uint64_t key, uint32_t chunk) { return (size_t) (mix64(key ^ (uint64_t(chunk) * 0x9E3779B97F4A7C15ULL)) >> 7); }
    Slot *find(uint64_t key, uint32_t chunk)
    {
        const size_t mask = slots.size() - 1;
        size_t i = hash(key, chunk) & mask;
        while (slots[i].key != EMPTY_KEY && !(slots[i].key == key && slots[i].chunk == chunk))
            i = (i + 1) & mask;
        return &slots[i];
    }
    void grow()
    {
        std::vector<Slot> old;
        old.swap(slots);
        slots.assign(old.size() * 2, Slot{EMPTY_KEY, 0, Ref{0, 0}});
        for (const Slot &s : old)
            if (s.key != EMPTY_KEY)
                *find(s.key, s.chunk) = s;
    }
    void note(uint64_t key, uint32_t chunk, Ref vertex)
    {
        if ((used + 1) * 10 > slots.size() * 7)
            grow();
        Slot *s = find(key, chunk);
        if (s->key == EMPTY_KEY)
        {
            *s = Slot{key, chunk, vertex};
            used++;
        }
        else if (refLess(vertex, s->owner))
            s->owner = vertex;
    }
    Ref owner(uint64_t key, uint32_t chunk) { return find(key, chunk)->owner; }
};

struct UnionRecord      /* one later occurrence of a key: unite the clumps, un-count one vertex (updateClumpKeyMap) */
{
    Ref a, b;
};

enum { NUM_SHARDS = 64 };

struct Shard
{
    std::mutex mutex;
    ClumpTable clumps;
    OwnerTable owners;
    std::vector<UnionRecord> unions;
};

struct Block
{
    uint32_t seq = 0;           /* arrival number */
    uint32_t chunk = 0;         /* dense chunk index */
    uint64_t nv = 0, nInternal = 0, nt = 0;
    /* in the mesher's arena */
    float *vertices = nullptr;
    uint32_t *triangles = nullptr;          /* block-local indices */
    uint64_t *keys = nullptr;               /* of the external vertices */
    uint32_t *clumpOf = nullptr;            /* per vertex: local clump number; filled by the block's task */
    uint32_t *remap = nullptr;              /* finalize's scratch, kept for the next finalize */
    bool checkIndices = false;              /* a landed block: its triangle indices have not been looked at yet */
    std::vector<uint64_t> clumpVertices, clumpTriangles;    /* per local clump */
    /* finalize */
    uint64_t clumpBase = 0;                 /* global id of local clump 0 */
    uint64_t emitted = 0, keptTriangles = 0, vOut = 0, tOut = 0;
};

} // namespace

struct mlsgpu_host_mesher
{
    std::mutex mutex;                       /* add / finalize / boundary from different threads */
    double pruneThreshold = 0.0;
    std::vector<std::unique_ptr<Block> > blocks;            /* in arrival order: blocks[i]->seq == i */
    std::vector<uint64_t> chunkIds;
    std::unordered_map<uint64_t, uint32_t> chunkIndex;
    Shard shards[NUM_SHARDS];
    std::unique_ptr<Pool> pool;
    unsigned threads = 0;
    int node = -1;                          /* NUMA node the pool's threads are bound to (-1: none) */
    /* MLSGPU_HIP_WELDER_TRACE=1: where a job's time went, one line on stderr when the welder is destroyed */
    struct Trace
    {
        bool on = getenv("MLSGPU_HIP_WELDER_TRACE") != nullptr;
        std::chrono::steady_clock::time_point firstAdd, lastAdd;
        double addS = 0, copyS = 0, drainS = 0, resolveS = 0, passesS = 0;
        uint64_t adds = 0, pieces = 0;
        std::atomic<uint64_t> callerPieces{0}, taskNs{0};
        uint64_t slabs0 = 0, bytes0 = 0;
    } trace;
    std::mutex errorMutex;
    int taskError = MLSGPU_OK;
    std::string taskErrorText;
    bool finalized = false;

    /* resolved in finalize / boundary */
    std::vector<int64_t> parent;            /* per global clump: -1 = root */
    std::vector<uint64_t> compVertices, compTriangles;      /* valid at roots */
    uint64_t numKeys = 0;

    /* bounded-memory mode (mlsgpu_hip_host_mesher_set_tmp_dir): the arenas are mappings of temporary files, and a block whose
     * task is done is handed to the kernel to write out and drop once more than tmpResident bytes of blocks have arrived */
    std::string tmpDir;
    uint64_t tmpResident = 0;
    std::atomic<uint64_t> blockBytes{0}, pagedOut{0};
    Arena arena;                            /* blocks, scratch and outputs */
    Arena landed{true};                     /* ship-outs that arrive in place (mlsgpu_hip_host_mesher_landing): page-locked slabs */
    float *outVertices = nullptr;
    uint32_t *outTriangles = nullptr;
    uint64_t outVCap = 0, outTCap = 0;
    std::vector<uint64_t> chunkVStart, chunkTStart;
    std::vector<uint32_t> outChunks;
    uint64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    Pool &getPool()
    {
        if (!pool)
        {
            unsigned n = threads;
            if (n == 0)
            {
                const char *e = getenv("MLSGPU_HIP_HOST_MESHER_THREADS");
                n = e != nullptr ? (unsigned) std::max(1, atoi(e)) : std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
            }
            pool.reset(new Pool(n, placement::cpusOfNode(node)));
            threads = n;
        }
        return *pool;
    }
    void fail(int code, const char *text)
    {
        std::lock_guard<std::mutex> l(errorMutex);
        if (taskError == MLSGPU_OK)
        {
            taskError = code;
            taskErrorText = text;
        }
    }
    uint64_t rootOf(uint64_t c)
    {
        uint64_t r = c;
        while (parent[r] >= 0)
            r = (uint64_t) parent[r];
        while (parent[c] >= 0)       /* path compression */
        {
            const uint64_t next = (uint64_t) parent[c];
            parent[c] = (int64_t) r;
            c = next;
        }
        return r;
    }
    void processBlock(Block *b);
    void pageOut(Block *b);
    int resolve();
    int finalizeWith(const uint8_t *keepClump, uint32_t *numChunks);
};

/* Bounded-memory mode: the block's arrays are not looked at again before the output passes.  Beyond the resident budget
 * they are given to the kernel to write to the temporary file and drop (whole pages inside the arrays; MADV_PAGEOUT is a
 * request: a kernel without it, or a file system in memory, keeps the pages and the job is merely not bounded). */
void mlsgpu_host_mesher::pageOut(Block *b)
{
    const uint64_t bytes = 3 * b->nv * sizeof(float) + 3 * b->nt * sizeof(uint32_t) + (b->nv - b->nInternal) * sizeof(uint64_t);
    if (blockBytes.fetch_add(bytes) + bytes <= tmpResident)
        return;
    const uintptr_t page = (uintptr_t) sysconf(_SC_PAGESIZE);
    auto drop = [&](void *p, size_t n) {
        const uintptr_t lo = ((uintptr_t) p + page - 1) & ~(page - 1), hi = ((uintptr_t) p + n) & ~(page - 1);
        if (hi > lo)
        {
#ifdef MADV_PAGEOUT
            if (madvise((void *) lo, hi - lo, MADV_PAGEOUT) == 0)
                pagedOut += hi - lo;
            else
#endif
                (void) msync((void *) lo, hi - lo, MS_ASYNC);       /* at least clean: the kernel can drop the pages when it wants the memory */
        }
    };
    drop(b->vertices, 3 * b->nv * sizeof(float));
    drop(b->triangles, 3 * b->nt * sizeof(uint32_t));
    drop(b->keys, (b->nv - b->nInternal) * sizeof(uint64_t));
}

/* the block's task: computeLocalComponents + updateGlobalClumps + updateClumpKeyMap (src/mesher.cpp:220-311) */
void mlsgpu_host_mesher::processBlock(Block *b)
{
    const uint64_t nv = b->nv, nt = b->nt, ni = b->nInternal, ne = nv - ni;
    /* Local components by a union-find that several threads work on at once (the pieces go to the front of the pool's queue:
     * whoever is idle helps, and behind the job's last mesh everybody is): parents only ever point to SMALLER indices and a
     * hook re-parents roots only (compare-and-swap), so an ancestor stays an ancestor whatever the other threads do, a
     * shortened path is always valid, and a finished component's root is its smallest vertex whatever the schedule was.
     * One thread used to take ~35 ms for a bucket's 1.3 M triangles -- the tail of every job. */
    uint32_t *const parent = arena.array<uint32_t>(nv);
    b->clumpOf = arena.array<uint32_t>(nv);
    if ((parent == nullptr || b->clumpOf == nullptr) && nv != 0)
    {
        fail(MLSGPU_ERR_NOMEM, "host mesher: out of memory");
        return;
    }
    const uint32_t *tris = b->triangles;
    const uint64_t SLICE = 1 << 16;
    Pool &P = getPool();
    if (b->checkIndices)
    {
        /* what add()'s copy checks on the way: no triangle index beyond the block's vertices (the union-find below indexes
         * with them) */
        std::atomic<uint32_t> bad{0};
        const uint64_t words = 3 * nt, CSLICE = uint64_t(1) << 20;
        P.parallelForNow((size_t) ((words + CSLICE - 1) / CSLICE), [&](size_t s) {
            uint32_t most = 0;
            for (uint64_t k = s * CSLICE, e = std::min(words, k + CSLICE); k < e; k++)
                most = tris[k] > most ? tris[k] : most;
            if (most >= nv)
                bad.store(1);
        });
        b->checkIndices = false;
        if (bad.load() != 0)
        {
            fail(MLSGPU_ERR_INVALID, "host mesher: a landed mesh has a triangle index beyond its vertices");
            b->nt = 0;      /* the block contributes nothing; finalize reports the error */
            b->nv = b->nInternal = 0;
            return;
        }
    }
    auto load = [parent](uint32_t v) { return __atomic_load_n(&parent[v], __ATOMIC_RELAXED); };
    auto find = [parent, &load](uint32_t v) {
        for (;;)
        {
            const uint32_t p = load(v);
            if (p == v)
                return v;
            const uint32_t g = load(p);
            if (g == p)
                return p;
            __atomic_store_n(&parent[v], g, __ATOMIC_RELAXED);      /* path halving */
            v = g;
        }
    };
    for (uint64_t i = 0; i < nv; i++)
        parent[i] = (uint32_t) i;
    /* a few large pieces: at most PIECES - 1 helpers per block (threads of another socket make the shared array slow) */
    static const uint64_t PIECES = getenv("MLSGPU_HIP_HOST_MESHER_PIECES") ? (uint64_t) atoi(getenv("MLSGPU_HIP_HOST_MESHER_PIECES")) : 4;
    const uint64_t USLICE = std::max<uint64_t>(SLICE, (nt + PIECES - 1) / std::max<uint64_t>(PIECES, 1));
    P.parallelForNow((size_t) ((nt + USLICE - 1) / USLICE), [&](size_t s) {
        for (uint64_t t = s * USLICE, e = std::min(nt, t + USLICE); t < e; t++)
        {
            const uint32_t *tri = tris + 3 * t;
            for (int k = 0; k < 2; k++)         /* the third edge is redundant, src/mesher.cpp:231-234 */
            {
                uint32_t x = tri[k], y = tri[k + 1];
                for (;;)
                {
                    x = find(x);
                    y = find(y);
                    if (x == y)
                        break;
                    if (x < y)
                        std::swap(x, y);
                    uint32_t expected = x;      /* hook the larger root under the smaller; retry if it was re-parented */
                    if (__atomic_compare_exchange_n(&parent[x], &expected, y, false, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED))
                        break;
                }
            }
        }
    });
    /* every vertex straight under its root */
    for (uint64_t i = 0; i < nv; i++)
    {
        uint32_t r = parent[i];
        while (parent[r] != r)
            r = parent[r];
        b->clumpOf[i] = r;                  /* the root for now */
    }
    /* clumps numbered by their smallest vertex, ascending */
    uint32_t numLocal = 0;
    for (uint64_t i = 0; i < nv; i++)
        if (b->clumpOf[i] == (uint32_t) i)
        {
            parent[i] = numLocal++;             /* the forest is not needed any more: the root's clump number */
            b->clumpVertices.push_back(0);
        }
    for (uint64_t i = 0; i < nv; i++)
    {
        const uint32_t c = parent[b->clumpOf[i]];
        b->clumpVertices[c]++;
        b->clumpOf[i] = c;
    }
    b->clumpTriangles.assign(numLocal, 0);
    for (uint64_t t = 0; t < nt; t++)
        b->clumpTriangles[b->clumpOf[tris[3 * t]]]++;
    /* the external keys, shard by shard */
    for (uint64_t i = 0; i < ne; i++)
    {
        const uint64_t key = b->keys[i];
        if (key == EMPTY_KEY)
        {
            fail(MLSGPU_ERR_INVALID, "host mesher: an external vertex key is all ones");
            return;
        }
        const Ref clump{b->seq, b->clumpOf[ni + i]};
        Shard &s = shards[mix64(key) >> 58];
        std::lock_guard<std::mutex> l(s.mutex);
        Ref first;
        if (!s.clumps.insert(key, clump, &first))
            s.unions.push_back(UnionRecord{clump, first});
        s.owners.note(key, b->chunk, Ref{b->seq, (uint32_t) (ni + i)});
    }
}

/* The welders' memory comes from mapped slabs that are kept, up to a limit (16 GiB, or MLSGPU_HIP_WELDER_CACHE_MB), when a
 * welder is destroyed: a fresh mapping faults at ~1 GB/s per thread, a kept one is reused at once (but is NOT zero-filled:
 * Arena memory is uninitialised).  This sets the limit and gives what is held beyond it back to the system; 0 = keep
 * nothing.  Returns the bytes released. */
MLSGPU_API uint64_t mlsgpu_hip_host_mesher_trim_cache(uint64_t keepBytes)
{
    return SlabCache::instance().setLimit((size_t) keepBytes);
}

MLSGPU_API int mlsgpu_hip_host_mesher_create(mlsgpu_host_mesher **out)
{
    REQUIRE(out != nullptr, MLSGPU_ERR_INVALID);
    *out = new mlsgpu_host_mesher;
    return MLSGPU_OK;
}

MLSGPU_API void mlsgpu_hip_host_mesher_destroy(mlsgpu_host_mesher *m)
{
    if (m == nullptr)
        return;
    if (m->pool)
        m->pool->drain();
    if (m->trace.on && m->trace.adds > 0)
        fprintf(stderr, "mlsgpu_hip welder: %llu adds in %.1f ms (first add .. last add returned), add() %.1f ms of which copies %.1f "
                        "(%llu of %llu pieces by the caller), block tasks %.1f ms of thread time; finalize: drain %.1f, resolve %.1f, "
                        "passes %.1f ms; fresh slabs %llu (%.0f MB)\n",
                (unsigned long long) m->trace.adds,
                std::chrono::duration<double>(m->trace.lastAdd - m->trace.firstAdd).count() * 1e3, m->trace.addS * 1e3,
                m->trace.copyS * 1e3, (unsigned long long) m->trace.callerPieces.load(), (unsigned long long) m->trace.pieces,
                m->trace.taskNs.load() * 1e-6, m->trace.drainS * 1e3, (m->trace.resolveS - m->trace.drainS) * 1e3,
                m->trace.passesS * 1e3, (unsigned long long) (SlabCache::instance().freshSlabs.load() - m->trace.slabs0),
                (SlabCache::instance().freshBytes.load() - m->trace.bytes0) / 1048576.0);
    if (m->trace.on && m->trace.adds > 0)
        fprintf(stderr, "mlsgpu_hip welder slabs:%s\n", m->arena.describe().c_str());
    delete m;
}

MLSGPU_API int mlsgpu_hip_host_mesher_set_prune_threshold(mlsgpu_host_mesher *m, double threshold)
{
    REQUIRE(m != nullptr && threshold >= 0.0 && threshold <= 1.0, MLSGPU_ERR_INVALID);
    m->pruneThreshold = threshold;
    return MLSGPU_OK;
}

/* Bounded-memory mode, OOCMesher's temporary files (src/mesher.cpp:404-419 writes every block's vertices and triangles to
 * them, :763-852 reads them back clump by clump): every slab of the welder -- blocks, scratch, the output arrays -- becomes a
 * mapping of a nameless temporary file in `dir`, and once more than `residentBytes` of blocks have arrived a block is
 * written out and dropped from memory as soon as its task is done.  The job's resident memory is then the per-vertex and
 * per-clump bookkeeping plus what the kernel chooses to keep, not the mesh.  The output passes fault the blocks back in
 * arrival order (the order they were written in).  Results are the in-memory mode's, element for element.  Ship-outs that
 * land in place (mlsgpu_hip_host_mesher_landing) land in pageable file-backed memory: not page-locked, slower, still
 * in place.  Before the first add; an unusable directory is MLSGPU_ERR_INVALID.  NULL or "" = back to memory. */
MLSGPU_API int mlsgpu_hip_host_mesher_set_tmp_dir(mlsgpu_host_mesher *m, const char *dir, uint64_t residentBytes)
{
    REQUIRE(m != nullptr, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    REQUIRE(m->blocks.empty(), MLSGPU_ERR_INVALID);
    const std::string d = dir != nullptr ? dir : "";
    if (!d.empty())
    {
        Slab probe = takeFileSlab(1, d);
        if (probe.base == nullptr)
            return setError(MLSGPU_ERR_INVALID, "host mesher: cannot create a temporary file in the given directory");
        probe.base[0] = 1;
        tmpBytesMapped -= probe.cap;
        munmap(probe.base, probe.cap);
    }
    m->tmpDir = d;
    m->tmpResident = residentBytes;
    m->arena.setTmpDir(d);
    m->landed.setTmpDir(d);
    return MLSGPU_OK;
}

/* out[0] = bytes of temporary files mapped by this welder, out[1] = how many of them are in memory right now, out[2] = bytes
 * of blocks handed to the kernel to write out and drop */
MLSGPU_API int mlsgpu_hip_host_mesher_tmp_usage(mlsgpu_host_mesher *m, uint64_t out[3])
{
    REQUIRE(m != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    out[0] = out[1] = 0;
    m->arena.fileUsage(out[0], out[1]);
    m->landed.fileUsage(out[0], out[1]);
    out[2] = m->pagedOut.load();
    return MLSGPU_OK;
}

/* Threads of the welder (block tasks and the output passes); 0 = the default: MLSGPU_HIP_HOST_MESHER_THREADS, else
 * min(32, hardware threads).  Before the first add. */
MLSGPU_API int mlsgpu_hip_host_mesher_set_threads(mlsgpu_host_mesher *m, uint32_t threads)
{
    REQUIRE(m != nullptr && threads <= 1024, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    REQUIRE(!m->pool, MLSGPU_ERR_INVALID);
    m->threads = threads;
    return MLSGPU_OK;
}

/* The welder's threads on ONE NUMA node: the one the meshes arrive on (the farm's read-back ring is next to its first GPU,
 * mlsgpu_hip_farm_placement out[1]).  A block's pieces are then welded by threads that share a socket's caches -- a block
 * whose pieces were spread over two sockets took the pass from 39 to 54-65 ms (profiles/NOTES_r04.md 9.9).  -1 = unbound
 * (the default).  Before the first add. */
MLSGPU_API int mlsgpu_hip_host_mesher_set_node(mlsgpu_host_mesher *m, int node)
{
    REQUIRE(m != nullptr, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    REQUIRE(!m->pool, MLSGPU_ERR_INVALID);
    m->node = node;
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_host_mesher_node(mlsgpu_host_mesher *m) { return m ? m->node : -1; }

MLSGPU_API uint32_t mlsgpu_hip_host_mesher_threads(mlsgpu_host_mesher *m)
{
    if (m == nullptr)
        return 0;
    std::lock_guard<std::mutex> lock(m->mutex);
    return m->getPool().size();
}

/* the common tail of add / add_landed: the block joins the job and its task is queued (the mesher's mutex is held) */
static int enlistBlock(mlsgpu_host_mesher *m, uint64_t chunkId, std::unique_ptr<Block> &b)
{
    REQUIRE(m->blocks.size() < 0xFFFFFFFFu, MLSGPU_ERR_LENGTH);
    m->finalized = false;
    auto it = m->chunkIndex.find(chunkId);
    if (it == m->chunkIndex.end())
    {
        b->chunk = (uint32_t) m->chunkIds.size();
        m->chunkIndex.emplace(chunkId, b->chunk);
        m->chunkIds.push_back(chunkId);
    }
    else
        b->chunk = it->second;
    b->seq = (uint32_t) m->blocks.size();
    Block *raw = b.get();
    m->blocks.push_back(std::move(b));
    m->getPool().submit([m, raw] {
        const auto t0 = std::chrono::steady_clock::now();
        m->processBlock(raw);
        if (!m->tmpDir.empty())
            m->pageOut(raw);
        if (m->trace.on)
            m->trace.taskNs += (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    });
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_host_mesher_add(mlsgpu_host_mesher *m, uint64_t chunkId, const mlsgpu_host_mesh *mesh)
{
    REQUIRE(m != nullptr && mesh != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(mesh->numInternalVertices <= mesh->numVertices, MLSGPU_ERR_INVALID);
    REQUIRE(mesh->numVertices < (uint64_t(1) << 31), MLSGPU_ERR_LENGTH);
    const uint64_t nv = mesh->numVertices, nt = mesh->numTriangles, ni = mesh->numInternalVertices, ne = nv - ni;
    REQUIRE((nv == 0 || mesh->vertices != nullptr) && (nt == 0 || mesh->triangles != nullptr)
            && (ne == 0 || mesh->vertexKeys != nullptr), MLSGPU_ERR_INVALID);
    std::unique_ptr<Block> b(new Block);
    b->nv = nv;
    b->nInternal = ni;
    b->nt = nt;
    const auto tAdd0 = std::chrono::steady_clock::now();
    if (m->trace.on && m->trace.adds == 0)
    {
        m->trace.firstAdd = tAdd0;
        m->trace.slabs0 = SlabCache::instance().freshSlabs.load();
        m->trace.bytes0 = SlabCache::instance().freshBytes.load();
    }
    /* the caller's memory (a slot of the farm's ring) is free again when this call returns */
    b->vertices = m->arena.array<float>(3 * nv);
    b->triangles = m->arena.array<uint32_t>(3 * nt);
    b->keys = m->arena.array<uint64_t>(ne);
    if (b->vertices == nullptr || b->triangles == nullptr || b->keys == nullptr)
        return setError(MLSGPU_ERR_NOMEM, "host mesher: out of memory");
    std::lock_guard<std::mutex> lock(m->mutex);
    {
        /* a block of tens of MB leaves the caller's memory on several threads at once */
        struct Piece { char *dst; const char *src; size_t bytes; bool indices; };
        std::vector<Piece> pieces;
        auto cut = [&](void *dst, const void *src, size_t bytes, bool indices) {
            const size_t step = size_t(4) << 20;
            for (size_t o = 0; o < bytes; o += step)
                pieces.push_back(Piece{static_cast<char *>(dst) + o, static_cast<const char *>(src) + o, std::min(step, bytes - o),
                                       indices});
        };
        cut(b->vertices, mesh->vertices, 3 * nv * sizeof(float), false);
        cut(b->triangles, mesh->triangles, 3 * nt * sizeof(uint32_t), true);
        cut(b->keys, mesh->vertexKeys, ne * sizeof(uint64_t), false);
        std::atomic<uint32_t> badIndex{0};
        const std::thread::id caller = std::this_thread::get_id();
        const auto tCopy0 = std::chrono::steady_clock::now();
        auto copyPiece = [&](size_t i) {
            const Piece &p = pieces[i];
            if (m->trace.on && std::this_thread::get_id() == caller)
                m->trace.callerPieces++;
            std::memcpy(p.dst, p.src, p.bytes);
            if (p.indices)
            {
                /* the check of the triangle indices rides on the copy */
                const uint32_t *t = reinterpret_cast<const uint32_t *>(p.dst);
                uint32_t most = 0;
                for (size_t k = 0; k < p.bytes / 4; k++)
                    most = t[k] > most ? t[k] : most;
                if (most >= nv)
                    badIndex.store(1);
            }
        };
        if (pieces.size() <= 2)
            for (size_t i = 0; i < pieces.size(); i++)
                copyPiece(i);
        else
            m->getPool().parallelForNow(pieces.size(), copyPiece);
        m->trace.copyS += std::chrono::duration<double>(std::chrono::steady_clock::now() - tCopy0).count();
        m->trace.pieces += pieces.size();
        /* a bad mesh leaves the sink unchanged (its copy stays behind in the arena, unused) */
        REQUIRE(badIndex.load() == 0, MLSGPU_ERR_INVALID);
    }
    PROPAGATE(enlistBlock(m, chunkId, b));
    m->trace.adds++;
    m->trace.lastAdd = std::chrono::steady_clock::now();
    m->trace.addS += std::chrono::duration<double>(m->trace.lastAdd - tAdd0).count();
    return MLSGPU_OK;
}

/* Room for a ship-out that arrives IN PLACE: `bytes` of the welder's own memory, page-locked where the process has a GPU,
 * valid until the welder is destroyed.  The farm's read-back lands there (mlsgpu_hip_farm_set_host_landing) and
 * mlsgpu_hip_host_mesher_add_landed adopts it: the copy out of the farm's ring that mlsgpu_hip_host_mesher_add makes -- a
 * third of the welder's pass on a good day, two thirds on a box whose GPU hangs off the other socket -- does not exist.
 * (The reference's mesher reads its CircularBuffer allocation in place too, src/workers.h:488-509, src/mesher.cpp:447-469,
 * and then writes the block to a temporary file; here the landing memory IS the block's home.) */
MLSGPU_API int mlsgpu_hip_host_mesher_landing(mlsgpu_host_mesher *m, uint64_t bytes, void **out)
{
    REQUIRE(m != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    void *p = m->landed.alloc((size_t) std::max<uint64_t>(bytes, 1));
    if (p == nullptr)
        return setError(MLSGPU_ERR_NOMEM, "host mesher: out of memory");
    *out = p;
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_host_mesher_landing_pinned(mlsgpu_host_mesher *m)
{
    return m != nullptr && m->landed.allRegistered() ? 1 : 0;
}

/* add() for a mesh that lies in memory mlsgpu_hip_host_mesher_landing returned: adopted, not copied.  The check of the
 * triangle indices (which rides on add()'s copy) is the first thing the block's task does; a bad mesh makes finalize fail
 * with MLSGPU_ERR_INVALID instead of this call. */
MLSGPU_API int mlsgpu_hip_host_mesher_add_landed(mlsgpu_host_mesher *m, uint64_t chunkId, const mlsgpu_host_mesh *mesh)
{
    REQUIRE(m != nullptr && mesh != nullptr, MLSGPU_ERR_INVALID);
    REQUIRE(mesh->numInternalVertices <= mesh->numVertices, MLSGPU_ERR_INVALID);
    REQUIRE(mesh->numVertices < (uint64_t(1) << 31), MLSGPU_ERR_LENGTH);
    const uint64_t nv = mesh->numVertices, nt = mesh->numTriangles, ni = mesh->numInternalVertices, ne = nv - ni;
    REQUIRE((nv == 0 || m->landed.contains(mesh->vertices, 12 * nv)) && (nt == 0 || m->landed.contains(mesh->triangles, 12 * nt))
            && (ne == 0 || m->landed.contains(mesh->vertexKeys, 8 * ne)), MLSGPU_ERR_INVALID);
    std::unique_ptr<Block> b(new Block);
    b->nv = nv;
    b->nInternal = ni;
    b->nt = nt;
    b->checkIndices = true;
    const auto tAdd0 = std::chrono::steady_clock::now();
    if (m->trace.on && m->trace.adds == 0)
    {
        m->trace.firstAdd = tAdd0;
        m->trace.slabs0 = SlabCache::instance().freshSlabs.load();
        m->trace.bytes0 = SlabCache::instance().freshBytes.load();
    }
    b->vertices = const_cast<float *>(mesh->vertices);
    b->triangles = const_cast<uint32_t *>(mesh->triangles);
    b->keys = const_cast<uint64_t *>(mesh->vertexKeys);
    std::lock_guard<std::mutex> lock(m->mutex);
    PROPAGATE(enlistBlock(m, chunkId, b));
    m->trace.adds++;
    m->trace.lastAdd = std::chrono::steady_clock::now();
    m->trace.addS += std::chrono::duration<double>(m->trace.lastAdd - tAdd0).count();
    return MLSGPU_OK;
}

/* the pair a farm takes (mlsgpu_hip_farm_set_host_landing): `user` is the mlsgpu_host_mesher */
MLSGPU_API int mlsgpu_hip_host_mesher_farm_landing(void *mesher, uint64_t bytes, void **out)
{
    return mlsgpu_hip_host_mesher_landing(static_cast<mlsgpu_host_mesher *>(mesher), bytes, out);
}

MLSGPU_API int mlsgpu_hip_host_mesher_farm_output_landed(void *mesher, int device, uint64_t chunkId, const mlsgpu_host_mesh *mesh)
{
    (void) device;
    return mlsgpu_hip_host_mesher_add_landed(static_cast<mlsgpu_host_mesher *>(mesher), chunkId, mesh);
}

MLSGPU_API int mlsgpu_hip_host_mesher_farm_output(void *mesher, int device, uint64_t chunkId, const mlsgpu_host_mesh *mesh)
{
    (void) device;
    return mlsgpu_hip_host_mesher_add(static_cast<mlsgpu_host_mesher *>(mesher), chunkId, mesh);
}

/* every block's task has run: global clump numbers (block by block in arrival order), the unions the key map recorded, and
 * the components' vertex / triangle counts -- a vertex seen k times was counted k times, each record takes one back */
int mlsgpu_host_mesher::resolve()
{
    const auto tDrain0 = std::chrono::steady_clock::now();
    getPool().drain();
    trace.drainS += std::chrono::duration<double>(std::chrono::steady_clock::now() - tDrain0).count();
    {
        std::lock_guard<std::mutex> l(errorMutex);
        if (taskError != MLSGPU_OK)
            return setError(taskError, "%s", taskErrorText.c_str());
    }
    uint64_t total = 0;
    for (auto &b : blocks)
    {
        b->clumpBase = total;
        total += b->clumpVertices.size();
    }
    REQUIRE(total < 0x7FFFFFFFu, MLSGPU_ERR_LENGTH);            /* "too many connected components", src/mesher.cpp:252 */
    parent.assign(total, -1);
    compVertices.resize(total);
    compTriangles.resize(total);
    for (auto &b : blocks)
        for (size_t c = 0; c < b->clumpVertices.size(); c++)
        {
            compVertices[b->clumpBase + c] = b->clumpVertices[c];
            compTriangles[b->clumpBase + c] = b->clumpTriangles[c];
        }
    numKeys = 0;
    for (Shard &s : shards)
    {
        numKeys += s.clumps.used;
        for (const UnionRecord &u : s.unions)
        {
            uint64_t a = rootOf(blocks[u.a.seq]->clumpBase + u.a.idx), c = rootOf(blocks[u.b.seq]->clumpBase + u.b.idx);
            if (a != c)
            {
                if (compVertices[a] < compVertices[c])
                    std::swap(a, c);
                parent[c] = (int64_t) a;
                compVertices[a] += compVertices[c];
                compTriangles[a] += compTriangles[c];
            }
            compVertices[a]--;          /* both counted the common vertex */
        }
    }
    /* Which member ended up as a component's root depends on the order the tasks reached the key map; the component's
     * LOWEST clump number becomes its root, so that ids (boundary export, verdicts) are the same whatever the threads did.
     * Every chain is one step afterwards: the output passes only read. */
    std::vector<uint64_t> lowest(total, ~uint64_t(0));
    for (uint64_t c = 0; c < total; c++)
    {
        const uint64_t r = rootOf(c);
        if (lowest[r] == ~uint64_t(0))
            lowest[r] = c;              /* ascending c: the first member seen is the lowest */
    }
    std::vector<int64_t> np(total);
    std::vector<uint64_t> nv(total, 0), nt(total, 0);
    for (uint64_t c = 0; c < total; c++)
    {
        const uint64_t r = rootOf(c), low = lowest[r];
        np[c] = c == low ? -1 : (int64_t) low;
        if (c == low)
        {
            nv[c] = compVertices[r];
            nt[c] = compTriangles[r];
        }
    }
    parent.swap(np);
    compVertices.swap(nv);
    compTriangles.swap(nt);
    return MLSGPU_OK;
}

/* keepClump == nullptr: the prune rule on this mesher's own counts (getStatistics, src/mesher.cpp:491-536);
 * otherwise keepClump[root clump] decides (the caller merged the clumps of several meshers, see
 * mlsgpu_hip_host_mesher_boundary) */
int mlsgpu_host_mesher::finalizeWith(const uint8_t *keepClump, uint32_t *numChunks)
{
    const auto tRes0 = std::chrono::steady_clock::now();
    PROPAGATE(resolve());
    const auto tPass0 = std::chrono::steady_clock::now();
    trace.resolveS += std::chrono::duration<double>(tPass0 - tRes0).count();
    const uint32_t nc = (uint32_t) chunkIds.size();
    const uint64_t numClumps = parent.size();
    uint64_t total = 0, components = 0;
    for (uint64_t c = 0; c < numClumps; c++)
        if (parent[c] < 0)
        {
            total += compVertices[c];
            components++;
        }
    const uint64_t threshold = (uint64_t) ((double) total * pruneThreshold);
    std::vector<uint8_t> keptRoot(numClumps, 0);
    uint64_t keptComponents = 0, keptVertices = 0, keptTriangles = 0;
    for (uint64_t c = 0; c < numClumps; c++)
        if (parent[c] < 0 && (keepClump ? keepClump[c] != 0 : compVertices[c] >= threshold))
        {
            keptRoot[c] = 1;
            keptComponents++;
            keptVertices += compVertices[c];
            keptTriangles += compTriangles[c];
        }
    auto keptClump = [&](uint64_t c) { return keptRoot[parent[c] < 0 ? c : (uint64_t) parent[c]] != 0; };
    /* blocks in (chunk by first arrival, arrival) order; a key is emitted once per chunk (externalRemap, :538-567) */
    std::vector<uint32_t> order(blocks.size());
    for (uint32_t i = 0; i < order.size(); i++)
        order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return blocks[a]->chunk < blocks[b]->chunk; });
    Pool &P = getPool();
    const uint32_t ALIAS = 0xFFFFFFFEu, PRUNED = 0xFFFFFFFFu;
    for (auto &b : blocks)
        if (b->remap == nullptr)
        {
            b->remap = arena.array<uint32_t>(b->nv);
            REQUIRE(b->remap != nullptr, MLSGPU_ERR_NOMEM);
        }
    /* The passes below run over SLICES of the blocks' vertex and triangle arrays, not over blocks: a job of a few dozen large
     * blocks (27 buckets: 0.65 M vertices and 1.3 M triangles each) keeps every thread of the pool busy. */
    const uint64_t SLICE = 1 << 16;
    struct Piece
    {
        uint32_t block;
        uint64_t begin, end, count, out;
    };
    std::vector<Piece> vPieces, tPieces;
    std::vector<size_t> vFirst(blocks.size() + 1, 0), tFirst(blocks.size() + 1, 0);
    for (size_t bi = 0; bi < blocks.size(); bi++)
    {
        const Block &b = *blocks[bi];
        vFirst[bi] = vPieces.size();
        tFirst[bi] = tPieces.size();
        for (uint64_t i = 0; i < b.nv; i += SLICE)
            vPieces.push_back(Piece{(uint32_t) bi, i, std::min(i + SLICE, b.nv), 0, 0});
        for (uint64_t t = 0; t < b.nt; t += SLICE)
            tPieces.push_back(Piece{(uint32_t) bi, t, std::min(t + SLICE, b.nt), 0, 0});
    }
    vFirst[blocks.size()] = vPieces.size();
    tFirst[blocks.size()] = tPieces.size();
    /* pass 1: which vertices a block emits (kept, and for an external vertex: the key's first of its chunk) ... */
    P.parallelFor(vPieces.size(), [&](size_t pi) {
        Piece &pc = vPieces[pi];
        Block &b = *blocks[pc.block];
        uint64_t emitted = 0;
        for (uint64_t i = pc.begin; i < pc.end; i++)
        {
            b.remap[i] = PRUNED;
            if (!keptClump(b.clumpBase + b.clumpOf[i]))
                continue;
            if (i >= b.nInternal)
            {
                const uint64_t key = b.keys[i - b.nInternal];
                const Ref o = shards[mix64(key) >> 58].owners.owner(key, b.chunk);
                if (o.seq != b.seq || o.idx != (uint32_t) i)
                {
                    b.remap[i] = ALIAS;
                    continue;
                }
            }
            b.remap[i] = 0;
            emitted++;
        }
        pc.count = emitted;
    });
    /* ... and how many of its triangles stay (a triangle's clump is its first corner's) */
    P.parallelFor(tPieces.size(), [&](size_t pi) {
        Piece &pc = tPieces[pi];
        const Block &b = *blocks[pc.block];
        uint64_t kt = 0;
        for (uint64_t t = pc.begin; t < pc.end; t++)
            kt += b.remap[b.triangles[3 * t]] != PRUNED;
        pc.count = kt;
    });
    chunkVStart.assign(nc + 1, 0);
    chunkTStart.assign(nc + 1, 0);
    uint64_t vAll = 0, tAll = 0;
    {
        size_t k = 0;
        for (uint32_t c = 0; c < nc; c++)
        {
            chunkVStart[c] = vAll;
            chunkTStart[c] = tAll;
            for (; k < order.size() && blocks[order[k]]->chunk == c; k++)
            {
                const uint32_t bi = order[k];
                Block &b = *blocks[bi];
                b.vOut = vAll;
                b.tOut = tAll;
                for (size_t pi = vFirst[bi]; pi < vFirst[bi + 1]; pi++)
                {
                    vPieces[pi].out = vAll;
                    vAll += vPieces[pi].count;
                }
                for (size_t pi = tFirst[bi]; pi < tFirst[bi + 1]; pi++)
                {
                    tPieces[pi].out = tAll;
                    tAll += tPieces[pi].count;
                }
                b.emitted = vAll - b.vOut;
                b.keptTriangles = tAll - b.tOut;
            }
        }
        chunkVStart[nc] = vAll;
        chunkTStart[nc] = tAll;
    }
    if (vAll > outVCap || outVertices == nullptr)
    {
        outVertices = arena.array<float>(3 * vAll);
        outVCap = vAll;
    }
    if (tAll > outTCap || outTriangles == nullptr)
    {
        outTriangles = arena.array<uint32_t>(3 * tAll);
        outTCap = tAll;
    }
    REQUIRE(outVertices != nullptr && outTriangles != nullptr, MLSGPU_ERR_NOMEM);
    /* pass 2: output indices (relative to the chunk's first vertex) and the vertex copies */
    P.parallelFor(vPieces.size(), [&](size_t pi) {
        const Piece &pc = vPieces[pi];
        Block &b = *blocks[pc.block];
        uint64_t at = pc.out;
        const uint64_t first = chunkVStart[b.chunk];
        for (uint64_t i = pc.begin; i < pc.end; i++)
            if (b.remap[i] == 0)
            {
                b.remap[i] = (uint32_t) (at - first);
                const float *v = &b.vertices[3 * i];
                outVertices[3 * at] = v[0];
                outVertices[3 * at + 1] = v[1];
                outVertices[3 * at + 2] = v[2];
                at++;
            }
    });
    /* pass 3: a later occurrence of a key points at the chunk's copy (an owner is never an alias: written in pass 2) ... */
    P.parallelFor(vPieces.size(), [&](size_t pi) {
        const Piece &pc = vPieces[pi];
        Block &b = *blocks[pc.block];
        for (uint64_t i = std::max(pc.begin, b.nInternal); i < pc.end; i++)
            if (b.remap[i] == ALIAS)
            {
                const uint64_t key = b.keys[i - b.nInternal];
                const Ref o = shards[mix64(key) >> 58].owners.owner(key, b.chunk);
                b.remap[i] = blocks[o.seq]->remap[o.idx];
            }
    });
    /* ... then the triangles */
    P.parallelFor(tPieces.size(), [&](size_t pi) {
        const Piece &pc = tPieces[pi];
        const Block &b = *blocks[pc.block];
        uint64_t at = 3 * pc.out;
        for (uint64_t t = pc.begin; t < pc.end; t++)
        {
            const uint32_t *tri = &b.triangles[3 * t];
            if (b.remap[tri[0]] == PRUNED)
                continue;               /* the whole clump was pruned */
            outTriangles[at] = b.remap[tri[0]];
            outTriangles[at + 1] = b.remap[tri[1]];
            outTriangles[at + 2] = b.remap[tri[2]];
            at += 3;
        }
    });
    uint64_t nvAdded = 0, ntAdded = 0;
    for (auto &b : blocks)
    {
        nvAdded += b->nv;
        ntAdded += b->nt;
    }
    outChunks.clear();
    for (uint32_t c = 0; c < nc; c++)
        if (chunkTStart[c + 1] > chunkTStart[c])      /* no output for a chunk without triangles, :820 */
            outChunks.push_back(c);
    stats[0] = total;
    stats[1] = threshold;
    stats[2] = components;
    stats[3] = keptComponents;
    stats[4] = keptVertices;
    stats[5] = keptTriangles;
    stats[6] = nvAdded;
    stats[7] = ntAdded;
    finalized = true;
    trace.passesS += std::chrono::duration<double>(std::chrono::steady_clock::now() - tPass0).count();
    if (numChunks)
        *numChunks = (uint32_t) outChunks.size();
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_host_mesher_finalize(mlsgpu_host_mesher *m, uint32_t *numChunks)
{
    REQUIRE(m != nullptr, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    return m->finalizeWith(nullptr, numChunks);
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
    PROPAGATE(m->resolve());
    *numKeys = m->numKeys;
    *numClumps = m->parent.size();
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_host_mesher_boundary_read(mlsgpu_host_mesher *m, uint64_t *keys, uint32_t *keyClump,
                                                    uint64_t *clumpVertices, uint64_t *clumpTriangles)
{
    REQUIRE(m != nullptr && keys != nullptr && keyClump != nullptr && clumpVertices != nullptr && clumpTriangles != nullptr,
            MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    PROPAGATE(m->resolve());
    std::vector<std::pair<uint64_t, uint32_t> > sorted;
    sorted.reserve(m->numKeys);
    for (Shard &s : m->shards)
        for (const ClumpTable::Slot &slot : s.clumps.slots)
            if (slot.key != EMPTY_KEY)
                sorted.emplace_back(slot.key, (uint32_t) m->rootOf(m->blocks[slot.clump.seq]->clumpBase + slot.clump.idx));
    std::sort(sorted.begin(), sorted.end());        /* by key: the same order on every rank, whatever the hash map did */
    for (size_t i = 0; i < sorted.size(); i++)
    {
        keys[i] = sorted[i].first;
        keyClump[i] = sorted[i].second;
    }
    for (size_t c = 0; c < m->parent.size(); c++)
    {
        const bool root = m->parent[c] < 0;
        clumpVertices[c] = root ? m->compVertices[c] : 0;
        clumpTriangles[c] = root ? m->compTriangles[c] : 0;
    }
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_host_mesher_finalize_with(mlsgpu_host_mesher *m, const uint8_t *keepClump, uint64_t numClumps,
                                                    uint32_t *numChunks)
{
    REQUIRE(m != nullptr && keepClump != nullptr, MLSGPU_ERR_INVALID);
    std::lock_guard<std::mutex> lock(m->mutex);
    PROPAGATE(m->resolve());
    REQUIRE(numClumps == m->parent.size(), MLSGPU_ERR_LENGTH);
    return m->finalizeWith(keepClump, numChunks);
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
    if (vertices) *vertices = m->outVertices + 3 * m->chunkVStart[c];
    if (triangles) *triangles = m->outTriangles + 3 * m->chunkTStart[c];
    return MLSGPU_OK;
}

MLSGPU_API int mlsgpu_hip_host_mesher_stats(mlsgpu_host_mesher *m, uint64_t out[8])
{
    REQUIRE(m != nullptr && out != nullptr && m->finalized, MLSGPU_ERR_INVALID);
    std::copy(m->stats, m->stats + 8, out);
    return MLSGPU_OK;
}
