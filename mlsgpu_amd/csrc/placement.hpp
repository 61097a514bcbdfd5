/*
 * Where host threads and host memory live relative to the GPUs (host side only).
 *
 * The reference has no notion of placement: its copy thread, device threads and mesher thread run wherever the scheduler
 * puts them (src/workers.cpp:320-351 picks a device by free capacity only), which was fine behind 1-6 GB GPUs on one
 * socket.  An 8 x MI355X node has two sockets, four GPUs behind each, ~62 GB/s per GPU link: pinned staging that the DMA
 * engine reads across the socket interconnect, or copy threads filling it from the other socket, share that interconnect
 * with every other GPU's traffic.  So: a GPU's socket is read from sysfs (hipDeviceGetPCIBusId ->
 * /sys/bus/pci/devices/<bdf>/numa_node), and what serves that GPU -- staging, read-back ring, copy threads, device worker
 * threads, welder threads -- is placed on it.
 *
 * Everything here degrades to "do nothing" on a one-node machine or when sysfs does not say.
 * MLSGPU_HIP_SYSFS_ROOT (default /sys) and MLSGPU_HIP_DEVICE_NODES ("0,0,1,1": node of device 0, 1, ...) let a test
 * describe a machine that is not there.
 */
#ifndef MLSGPU_AMD_PLACEMENT_HPP
#define MLSGPU_AMD_PLACEMENT_HPP

#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace mlsgpu
{
namespace placement
{

inline std::string sysfsRoot()
{
    const char *e = getenv("MLSGPU_HIP_SYSFS_ROOT");
    return e != nullptr && *e ? std::string(e) : std::string("/sys");
}

/* "0-3,8,10-11" -> {0,1,2,3,8,10,11} */
inline std::vector<int> parseCpuList(const std::string &text)
{
    std::vector<int> out;
    size_t i = 0;
    while (i < text.size())
    {
        while (i < text.size() && (text[i] < '0' || text[i] > '9'))
            i++;
        if (i >= text.size())
            break;
        long lo = strtol(text.c_str() + i, nullptr, 10), hi = lo;
        while (i < text.size() && text[i] >= '0' && text[i] <= '9')
            i++;
        if (i < text.size() && text[i] == '-')
        {
            i++;
            hi = strtol(text.c_str() + i, nullptr, 10);
            while (i < text.size() && text[i] >= '0' && text[i] <= '9')
                i++;
        }
        for (long c = lo; c <= hi && c - lo < 4096; c++)
            out.push_back((int) c);
    }
    return out;
}

/* cpus[node] for the nodes 0 .. n-1 that <root>/devices/system/node/ lists (a node without CPUs keeps an empty list) */
inline std::vector<std::vector<int> > readNodes(const std::string &root)
{
    std::vector<std::vector<int> > cpus;
    for (int node = 0; node < 64; node++)
    {
        std::ifstream in(root + "/devices/system/node/node" + std::to_string(node) + "/cpulist");
        if (!in)
            break;
        std::string line;
        std::getline(in, line);
        cpus.push_back(parseCpuList(line));
    }
    return cpus;
}

/* the NUMA node a PCI function hangs off, or -1 (sysfs says -1 on one-node machines) */
inline int pciNode(const std::string &root, const std::string &bdf)
{
    std::string name = bdf;
    std::transform(name.begin(), name.end(), name.begin(), [](unsigned char c) { return (char) tolower(c); });
    std::ifstream in(root + "/bus/pci/devices/" + name + "/numa_node");
    int node = -1;
    if (in)
        in >> node;
    return node;
}

/* MLSGPU_HIP_DEVICE_NODES="0,0,1,1": the node of device ordinal i (a test's machine); -2 = not given */
inline int overriddenDeviceNode(int device)
{
    const char *e = getenv("MLSGPU_HIP_DEVICE_NODES");
    if (e == nullptr || !*e)
        return -2;
    const std::vector<int> list = [&] {
        std::vector<int> v;
        const char *p = e;
        while (*p)
        {
            char *end = nullptr;
            const long x = strtol(p, &end, 10);
            if (end == p)
                break;
            v.push_back((int) x);
            p = *end ? end + 1 : end;
        }
        return v;
    }();
    return device >= 0 && (size_t) device < list.size() ? list[(size_t) device] : -1;
}

/*
 * The plan: which copy side serves which device.  One side per NUMA node that has a device of the farm (devices whose
 * node is unknown share side 0's); sides are numbered in order of first appearance, so device 0's side is side 0.
 * sideOfDevice[i] for i < n; nodeOfSide gets one entry per side.  Pure: a test calls it with a machine of its own.
 */
inline void planSides(const int *deviceNodes, size_t n, size_t numNodes, std::vector<int> &sideOfDevice, std::vector<int> &nodeOfSide)
{
    sideOfDevice.assign(n, 0);
    nodeOfSide.clear();
    for (size_t i = 0; i < n; i++)
    {
        const int node = numNodes > 1 && deviceNodes[i] >= 0 && (size_t) deviceNodes[i] < numNodes ? deviceNodes[i] : -1;
        size_t s = 0;
        while (s < nodeOfSide.size() && nodeOfSide[s] != node)
            s++;
        if (s == nodeOfSide.size())
        {
            if (node < 0 && !nodeOfSide.empty())
                s = 0;                          /* unknown: with the first side */
            else
                nodeOfSide.push_back(node);
        }
        sideOfDevice[i] = (int) s;
    }
    if (nodeOfSide.empty())
        nodeOfSide.push_back(-1);
}

/* binds the CALLING thread to a node's CPUs; false (and no change) when the node is unknown or has no CPUs here */
inline bool bindThisThread(const std::vector<int> &cpus)
{
    if (cpus.empty())
        return false;
    cpu_set_t set;
    CPU_ZERO(&set);
    int usable = 0;
    for (int c : cpus)
        if (c >= 0 && c < CPU_SETSIZE)
        {
            CPU_SET(c, &set);
            usable++;
        }
    return usable > 0 && sched_setaffinity(0, sizeof(set), &set) == 0;
}

inline std::vector<int> cpusOfNode(int node)
{
    if (node < 0)
        return std::vector<int>();
    static const std::vector<std::vector<int> > nodes = readNodes(sysfsRoot());
    if (nodes.size() < 2 || (size_t) node >= nodes.size())
        return std::vector<int>();              /* one node: nothing to choose */
    return nodes[(size_t) node];
}

inline size_t numNodes()
{
    static const size_t n = readNodes(sysfsRoot()).size();
    return n;
}

/* which node the page at `p` is on (move_pages with no target: a query), or -1 */
inline int nodeOfAddress(const void *p)
{
#ifdef SYS_move_pages
    void *page = (void *) ((uintptr_t) p & ~(uintptr_t) 4095);
    int status = -1;
    if (syscall(SYS_move_pages, 0, 1UL, &page, nullptr, &status, 0) == 0)
        return status;
#endif
    return -1;
}

/*
 * A memcpy spread over a few persistent threads (bound to the CPUs of a node if given): one bucket's 64 MB into pinned
 * staging at the rate the PCIe link drains it, without a thread creation per bucket.
 */
class CopyPool
{
public:
    CopyPool() = default;
    CopyPool(const CopyPool &) = delete;
    CopyPool &operator=(const CopyPool &) = delete;
    ~CopyPool() { stop(); }

    /* `threads` includes the caller of copy(); helpers = threads - 1 */
    void start(unsigned threads, const std::vector<int> &cpus)
    {
        stop();
        bound = cpus;
        quit = false;
        const unsigned helpers = threads > 1 ? threads - 1 : 0;
        for (unsigned i = 0; i < helpers; i++)
            pool.emplace_back([this] { helper(); });
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> l(mutex);
            quit = true;
        }
        wake.notify_all();
        for (std::thread &t : pool)
            t.join();
        pool.clear();
    }
    unsigned threads() const { return (unsigned) pool.size() + 1; }

    void copy(void *dst, const void *src, size_t bytes)
    {
        const size_t minChunk = size_t(2) << 20;
        const size_t parts = std::min<size_t>(pool.size() + 1, std::max<size_t>(1, bytes / minChunk));
        if (parts <= 1)
        {
            std::memcpy(dst, src, bytes);
            return;
        }
        Job j;
        j.dst = static_cast<char *>(dst);
        j.src = static_cast<const char *>(src);
        j.bytes = bytes;
        j.chunk = ((bytes + parts - 1) / parts + 4095) & ~size_t(4095);   /* parts * chunk >= bytes: the CEILING, rounded up to pages */
        j.parts = parts;
        launch(j);
    }

    /* fn(part) for part in [0, parts) on the pool's threads and the caller; returns when every part is done.  One caller at
     * a time. */
    void run(size_t parts, const std::function<void(size_t)> &fn)
    {
        if (parts == 0)
            return;
        if (parts == 1 || pool.empty())
        {
            for (size_t i = 0; i < parts; i++)
                fn(i);
            return;
        }
        Job j;
        j.parts = parts;
        j.fn = &fn;
        launch(j);
    }

private:
    struct Job
    {
        char *dst = nullptr;
        const char *src = nullptr;
        size_t bytes = 0, chunk = 0, parts = 0;
        const std::function<void(size_t)> *fn = nullptr;    /* null: the memcpy of part p */
    };

    void launch(const Job &j)
    {
        const size_t parts = j.parts;
        uint64_t mine;
        {
            std::lock_guard<std::mutex> l(mutex);
            job = j;
            pendingParts = parts;
            mine = ++generation;
            ticket.store(mine << 32);               /* part 0 of this generation is the next to be taken */
        }
        wake.notify_all();
        work(j, mine);                              /* the caller copies too, until no part is left to take ... */
        std::unique_lock<std::mutex> l(mutex);
        done.wait(l, [this] { return pendingParts == 0; });     /* ... then waits for the parts the helpers took */
    }

    /* Takes parts of generation `gen` until none is left.  The ticket holds (generation, next part): a helper that is late
     * leaving the previous job cannot take -- or lose -- a part of the next one. */
    void work(const Job &j, uint64_t gen)
    {
        for (;;)
        {
            uint64_t t = ticket.load();
            size_t p;
            for (;;)
            {
                p = (size_t) (t & 0xFFFFFFFFu);
                if ((t >> 32) != gen || p >= j.parts)
                    return;
                if (ticket.compare_exchange_weak(t, t + 1))
                    break;
            }
            if (j.fn != nullptr)
                (*j.fn)(p);
            else
            {
                const size_t off = p * j.chunk;
                if (off < j.bytes)
                    std::memcpy(j.dst + off, j.src + off, std::min(j.chunk, j.bytes - off));
            }
            std::lock_guard<std::mutex> l(mutex);
            if (--pendingParts == 0)
                done.notify_all();
        }
    }
    void helper()
    {
        bindThisThread(bound);
        uint64_t seen = 0;
        for (;;)
        {
            Job j;
            {
                std::unique_lock<std::mutex> l(mutex);
                wake.wait(l, [&] { return quit || generation != seen; });
                if (quit)
                    return;
                seen = generation;
                j = job;
            }
            work(j, seen);
        }
    }

    std::vector<std::thread> pool;
    std::vector<int> bound;
    std::mutex mutex;
    std::condition_variable wake, done;
    bool quit = false;
    uint64_t generation = 0;
    Job job;
    size_t pendingParts = 0;
    std::atomic<uint64_t> ticket{0};
};

} // namespace placement
} // namespace mlsgpu

#endif
