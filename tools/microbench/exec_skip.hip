// Does a SIMD-32 skip the pass of a wave64 vector instruction whose 32 lanes are all switched off?  The drain of processCorners
// runs its last iterations with one or two lanes left: if an empty half cost nothing, packing a sub-block's busy corners into
// one half would pay.  A loop of independent v_fma_f32 under four EXEC masks, eight waves per SIMD, every CU.
// Build and run ON THE GPU BOX:
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/exec_skip tools/microbench/exec_skip.hip && /tmp/exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define FMA8 asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t" \
                          "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9" \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c))

/* every iteration: 32 instructions under maskA, then 32 under maskB */
__global__ __launch_bounds__(512) void fmaKernel(float *out, uint64_t maskA, uint64_t maskB, uint32_t iters)
{
    const uint32_t lane = threadIdx.x & 63;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    const float m = 1.0001f, c = 0.5f;
    const bool inA = (maskA >> lane) & 1, inB = (maskB >> lane) & 1;
    for (uint32_t i = 0; i < iters; i++)
    {
        if (inA)
        {
            FMA8; FMA8; FMA8; FMA8;
        }
        if (inB)
        {
            FMA8; FMA8; FMA8; FMA8;
        }
    }
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 12345.678f)
        out[threadIdx.x] = s;
}

int main()
{
    float *out;
    CHECK(hipMalloc((void **) &out, 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const uint32_t iters = 4096, wgs = 256 * 4;      /* four workgroups of eight waves per CU: eight waves per SIMD */
    const uint64_t all = ~0ull;
    auto low = [](int n) { return n >= 64 ? ~0ull : (1ull << n) - 1; };
    const struct { const char *name; uint64_t a, b; } cases[] = {
        {"all 64 lanes", all, all}, {"lanes 0-31", low(32), low(32)}, {"lanes 32-63", ~low(32), ~low(32)}, {"even lanes", 0x5555555555555555ull, 0x5555555555555555ull},
        {"lanes 0-15", low(16), low(16)}, {"lanes 0-14", low(15), low(15)}, {"lanes 0-13", low(14), low(14)}, {"lanes 0-12", low(13), low(13)},
        {"lanes 0-11", low(12), low(12)}, {"lanes 0-10", low(11), low(11)}, {"lanes 0-9", low(10), low(10)}, {"lanes 0-8", low(9), low(9)},
        {"lanes 0-7", low(8), low(8)}, {"lanes 0-3", low(4), low(4)}, {"lane 0 only", 1, 1}, {"lane 63 only", 1ull << 63, 1ull << 63},
        {"every 8th lane", 0x0101010101010101ull, 0x0101010101010101ull}, {"every 4th lane", 0x1111111111111111ull, 0x1111111111111111ull},
        {"8 lanes + 1", 0x0101010101010101ull | 2, 0x0101010101010101ull | 2},
        {"all, then lane 0", all, 1}, {"all, then lanes 0-7", all, low(8)}, {"lanes 0-15, then lane 0", low(16), 1}};
    for (int rep = 0; rep < 2; rep++)
        for (const auto &cs : cases)
        {
            fmaKernel<<<wgs, 512>>>(out, cs.a, cs.b, 16);
            CHECK(hipEventRecord(e0));
            fmaKernel<<<wgs, 512>>>(out, cs.a, cs.b, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            /* per SIMD: 8 waves x iters x 64 instructions */
            const double instr = 8.0 * iters * 64.0;
            printf("%-24s %8.3f ms  %.2f cycles per wave instruction and SIMD at 2.4 GHz\n", cs.name, ms, ms * 1e-3 * 2.4e9 / instr);
        }
    return 0;
}
