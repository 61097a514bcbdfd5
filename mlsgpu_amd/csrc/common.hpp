/* Shared host/device plumbing of the HIP path: context, error reporting, launch timing, wave/block scans. */
#ifndef MLSGPU_AMD_COMMON_HPP
#define MLSGPU_AMD_COMMON_HPP

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mlsgpu_hip.h"

#define MLSGPU_API extern "C" __attribute__((visibility("default")))

namespace mlsgpu
{

/* ---------------------------------------------------------------- errors */

int setError(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_CHECK(expr)                                                                         \
    do {                                                                                        \
        hipError_t err__ = (expr);                                                              \
        if (err__ != hipSuccess)                                                                \
            return ::mlsgpu::setError(err__ == hipErrorOutOfMemory ? MLSGPU_ERR_NOMEM : MLSGPU_ERR_HIP, \
                                      "%s failed: %s (%s:%d)", #expr, hipGetErrorString(err__), \
                                      __FILE__, __LINE__);                                      \
    } while (0)

/* MLSGPU_ASSERT of src/errors.h:41-42: argument checks that raise length_error / invalid_argument */
#define REQUIRE(cond, code)                                                                     \
    do {                                                                                        \
        if (!(cond))                                                                            \
            return ::mlsgpu::setError(code, "requirement failed: %s (%s:%d)", #cond, __FILE__, __LINE__); \
    } while (0)

#define PROPAGATE(expr)                 \
    do {                                \
        int rc__ = (expr);              \
        if (rc__ != MLSGPU_OK)          \
            return rc__;                \
    } while (0)

/* Restores the calling thread's current device on every exit path.  Entry points that switch devices (peer appends, the
 * farm's dispatch to another GPU's copy stream) are called from worker threads in the middle of their own device's work;
 * anything device-implicit they do afterwards (event creation, hipMalloc) must still target their own GPU. */
struct DeviceGuard
{
    int saved = -1;
    DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
    ~DeviceGuard() { if (saved >= 0) (void) hipSetDevice(saved); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

/* Direct copies between two GPUs (xGMI) need peer access enabled in both directions; harmless when it already is, and
 * when the devices cannot reach each other the runtime stages the copy through the host. */
static inline void enablePeerAccess(int a, int b)
{
    if (a == b)
        return;
    DeviceGuard restore;
    for (int k = 0; k < 2; k++)
    {
        const int from = k ? b : a, to = k ? a : b;
        int can = 0;
        if (hipSetDevice(from) == hipSuccess && hipDeviceCanAccessPeer(&can, from, to) == hipSuccess && can)
            (void) hipDeviceEnablePeerAccess(to, 0);
        (void) hipGetLastError();       /* "already enabled" is not an error */
    }
}

/* ---------------------------------------------------------------- context */

struct Stat
{
    double totalMs = 0.0;
    uint64_t launches = 0;
};

struct PendingTiming
{
    int nameId;
    hipEvent_t start, stop;
};

} // namespace mlsgpu

struct mlsgpu_ctx
{
    int device = 0;
    uint64_t serial = 0;              /* unique per context: call sites cache stat ids against it, not the address */
    hipStream_t stream = nullptr;
    bool ownStream = false;
    bool timing = false;
    std::vector<std::string> statNames;
    std::map<std::string, int> statIds;
    std::vector<mlsgpu::Stat> stats;
    std::vector<mlsgpu::PendingTiming> pending;
    std::vector<hipEvent_t> eventPool;
    /* device scratch that entry points without an object of their own (mlsgpu_hip_bucket) keep between calls;
     * released with the context */
    std::map<std::string, std::shared_ptr<void> > scratchCache;

    /* one-launch scans (primitives.hpp): a flag word per (lane, tile), never cleared -- a launch's flags carry its epoch */
    uint32_t *dScanFlags = nullptr;
    uint32_t scanEpoch = 0;
    uint32_t scanTicketBase[MLSGPU_MAX_BATCH] = {};     /* what lane k's ticket counter reads when the next launch begins */
    /* flags + epoch of the next one-launch scan, the lanes' ticket counters and what they read now; a launch of `gridX`
     * workgroups for each of `count` lanes is accounted for */
    int scanFlags(uint32_t **flags, uint32_t *epoch, uint32_t **tickets, uint32_t *bases, uint32_t gridX, uint32_t count);

    int statId(const char *name);
    /* a sample of one of the reference's non-timer statistics (Statistics::Counter / Variable: marching.overflow,
     * marching.slices.nonempty, marching.shipouts; src/marching.cpp:350-352): the sum in the "ms" column, the samples in the
     * launch column of mlsgpu_hip_ctx_get_stat / _dump_stats */
    void addValue(const char *name, double value)
    {
        mlsgpu::Stat &st = stats[(size_t) statId(name)];
        st.totalMs += value;
        st.launches += 1;
    }
    int beginTiming(int id);          /* returns index into pending or -1 */
    void endTiming(int pendingIdx);
    int resolveTimings();             /* synchronises the stream */
};

namespace mlsgpu
{

/*
 * Small device -> host read-backs on the critical path (the octree's entry count, Marching's swathe totals and welded
 * counts: 4 to 24 bytes each, three per bucket, each followed by a host decision).  A hipMemcpyAsync of a few bytes plus
 * hipStreamSynchronize costs a copy-engine submission and a signal wait; here a one-thread kernel stores the words straight
 * into mapped, coherent pinned memory, a sequence number behind a system-scope fence, and the host polls that word.  The
 * stream is not drained: what the host enqueues next lands behind the kernel that published.
 */
struct HostMailbox
{
    enum { WORDS = 64 };            /* six words per bucket of a batch (MAX_LANES buckets) and room to spare */
    uint32_t *host = nullptr;       /* [0] = sequence number of the last publication, [1 .. WORDS] = payload */
    uint32_t *dev = nullptr;        /* the same memory as the device sees it */
    uint32_t seq = 0;

    int create();
    void destroy();
    /* for a kernel that publishes by itself (payload words first, then a system-scope fence, then the sequence number into
     * dev[0]): the sequence number it has to write; wait() then waits for that one */
    uint32_t reserve()
    {
        seq++;
        if (seq == 0)
            seq = 1;
        return seq;
    }
    /* enqueue: copy `words` 32-bit words from device memory `src` to the mailbox (on `stream`) */
    int publish(hipStream_t stream, const void *src, uint32_t words);
    /* the same for a batch: `wordsEach` words from each of `count` device addresses, laid out one source after the other */
    int publishGather(hipStream_t stream, const void *const *srcs, uint32_t count, uint32_t wordsEach);
    /* block until the last publication has landed; the payload is host[1 ..] */
    int wait(hipStream_t stream);
    const uint32_t *payload() const { return host + 1; }
};

/* Launch a kernel on the context's stream, timing it under a reference stat name when enabled. */
#define LAUNCH(ctx, statName, kernel, grid, block, ...) LAUNCH_LDS(ctx, statName, kernel, grid, block, 0, __VA_ARGS__)

/* the same with `ldsBytes` of dynamic LDS on top of the kernel's static allocation */
#define LAUNCH_LDS(ctx, statName, kernel, grid, block, ldsBytes, ...)                         \
    do {                                                                                      \
        static thread_local int statId__ = -1;                                                \
        static thread_local uint64_t statCtx__ = 0;                                           \
        int pend__ = -1;                                                                      \
        if ((ctx)->timing) {                                                                  \
            if (statCtx__ != (ctx)->serial) { statId__ = (ctx)->statId(statName); statCtx__ = (ctx)->serial; } \
            pend__ = (ctx)->beginTiming(statId__);                                            \
        }                                                                                     \
        hipLaunchKernelGGL(kernel, grid, block, ldsBytes, (ctx)->stream, __VA_ARGS__);        \
        if (pend__ >= 0) (ctx)->endTiming(pend__);                                            \
        HIP_CHECK(hipGetLastError());                                                         \
    } while (0)

/*
 * Batches.  A device worker may take several buckets (the SubItems of a WorkItem, src/workers.h:148-181) through the path
 * in lock-step: every kernel has a bucket dimension -- blockIdx.y selects the LANE, whose arguments are one element of an
 * array passed by value in the kernel-argument segment (read through the scalar cache, no upload of a parameter block),
 * grid.x covers the largest lane and the workgroups beyond a smaller lane's own extent leave at once.  A single bucket is a
 * batch of one: same kernels, same results.
 * A kernel COPIES its lane's element (`const Args A = lanes.a[blockIdx.y];`): the fields then live in scalar registers as
 * those of a by-value parameter do.  Through a reference the compiler re-loads a field wherever it is used under a condition
 * -- latticeMask, bound by scalar / vector issue, ran 18 % slower with ~100 kernel-argument loads in its loop.
 */
enum { MAX_LANES = MLSGPU_MAX_BATCH };
template<typename A>
struct Lanes
{
    A a[MAX_LANES];
};

static inline uint32_t divUp(uint64_t a, uint64_t b) { return (uint32_t) ((a + b - 1) / b); }
static inline uint32_t roundUp(uint32_t a, uint32_t b) { return (a + b - 1) / b * b; }

/* ---------------------------------------------------------------- device helpers */

#ifdef __HIPCC__

#define WAVE 64

__device__ __forceinline__ uint32_t laneId()
{
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

/* number of set bits of `mask` strictly below this lane */
__device__ __forceinline__ uint32_t popcBelow(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
}

/* Cross-lane steps as DPP modifiers of the vector instruction itself (gfx9: row_shr, row_bcast15 / 31, wave_shr) instead
 * of ds_bpermute round trips through the LDS crossbar: a scan is six dependent VALU instructions, not six LDS latencies.
 * update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl = false): lanes without a source lane, or outside the masks,
 * get `old`. */
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_WAVE_SHR1 0x138
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143

/* inclusive prefix sum across the 64 lanes of a wave */
__device__ __forceinline__ uint32_t waveInclusiveScan(uint32_t v)
{
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, DPP_ROW_SHR(1), 0xf, 0xf, false);
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, DPP_ROW_SHR(2), 0xf, 0xf, false);
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, DPP_ROW_SHR(4), 0xf, 0xf, false);
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, DPP_ROW_SHR(8), 0xf, 0xf, false);
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, DPP_ROW_BCAST15, 0xa, 0xf, false);   /* rows 1, 3 += row 0 / 2 */
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, DPP_ROW_BCAST31, 0xc, 0xf, false);   /* rows 2, 3 += rows 0 + 1 */
    return v;
}

/* The reductions below read lane 63 of the scan, and the shifts use the gfx9 wave_shr / wave_shl DPP controls: they need a
 * FULL, CONVERGENT wave64 (every caller in this library is one; an inactive lane 63 would hand back stale register
 * contents, where a ds_bpermute version reads zero from inactive lanes). */
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "the wave helpers are written for gfx950 (wave64, gfx9 DPP controls)"
#endif

/* the sum over the wave, in every lane */
__device__ __forceinline__ uint32_t waveSum(uint32_t v)
{
    return (uint32_t) __builtin_amdgcn_readlane((int) waveInclusiveScan(v), 63);
}

/* maximum over the wave, in every lane (wave-uniform) */
__device__ __forceinline__ uint32_t waveMax(uint32_t v)
{
    v = max(v, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, DPP_ROW_SHR(1), 0xf, 0xf, false));
    v = max(v, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, DPP_ROW_SHR(2), 0xf, 0xf, false));
    v = max(v, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, DPP_ROW_SHR(4), 0xf, 0xf, false));
    v = max(v, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, DPP_ROW_SHR(8), 0xf, 0xf, false));
    v = max(v, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, DPP_ROW_BCAST15, 0xa, 0xf, false));
    v = max(v, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, DPP_ROW_BCAST31, 0xc, 0xf, false));
    return (uint32_t) __builtin_amdgcn_readlane((int) v, 63);
}

/* value of lane - 1; lane 0 gets zero */
__device__ __forceinline__ uint32_t waveShiftUp1(uint32_t v)
{
    return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, DPP_WAVE_SHR1, 0xf, 0xf, false);
}

/* value of lane + 1; lane 63 gets zero */
__device__ __forceinline__ uint32_t waveShiftDown1(uint32_t v)
{
    return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

__device__ __forceinline__ uint32_t readLane(uint32_t v, int lane)
{
    return __builtin_amdgcn_readlane(v, lane);
}

#endif /* __HIPCC__ */

} // namespace mlsgpu

#endif
