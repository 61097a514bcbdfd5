// The store pattern of latticeTrianglesRow in isolation: one wave per "row", a row = R consecutive dwords of the output (rows are
// back to back), written front to back in 256-byte instructions (64 lanes x 4 bytes).  How much of the streaming store rate is
// left when the rows -- hence every store instruction -- are not aligned to the 128-byte lines, and when a wave writes its row in
// bursts of `burst` instructions with `gap` dependent LDS round trips in between (the cell side of a chunk)?
// Build and run ON THE GPU BOX:
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/rowstore_bw tools/microbench/rowstore_bw.hip && /tmp/rowstore_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template<int WIDTH>     /* dwords per lane and store instruction: 1 or 4 */
__global__ __launch_bounds__(256) void rowStoreKernel(uint32_t *dst, uint32_t rows, uint32_t R, uint32_t burst, uint32_t gap, uint32_t ldsPad)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t r = blockIdx.x * 4 + wv;
    if (r >= rows)
        return;
    uint32_t *const row = dst + (uint64_t) r * R;
    uint32_t v = r;
    lds[threadIdx.x] = threadIdx.x;
    uint32_t k = 0, inBurst = 0;
    while (k < R)
    {
        if (WIDTH == 1)
        {
            const uint32_t p = k + lane;
            row[p < R ? p : R - 1] = v;
            k += 64;
        }
        else
        {
            const uint32_t p = k + 4 * lane;
            if (p + 3 < R)
                *(uint4 *) (row + p) = make_uint4(v, v, v, v);     /* (needs R % 4 == 0 and 16-byte aligned rows to be legal: only then used) */
            k += 256;
        }
        if (++inBurst == burst)
        {
            inBurst = 0;
            /* `gap` dependent LDS round trips: what the cell side of the next chunk costs a wave at least */
            uint32_t a = (v + lane) & 255u;
            for (uint32_t g = 0; g < gap; g++)
                a = lds[a] & 255u;
            v += a;
        }
    }
    if (v == 0xFFFFFFFFu)
        dst[0] = ldsPad;
}

int main()
{
    const uint64_t bytes = 2ull << 30;
    uint32_t *a;
    CHECK(hipMalloc((void **) &a, bytes + (1 << 20)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct Case { const char *name; uint32_t R; int width; uint32_t burst, gap, ldsBytes; };
    const Case cases[] = {
        {"rows of 3264 dwords (aligned), 4 B/lane", 3264, 1, 1u << 30, 0, 1024},
        {"rows of 3271 dwords (unaligned), 4 B/lane", 3271, 1, 1u << 30, 0, 1024},
        {"rows of 3264 dwords (aligned), 16 B/lane", 3264, 4, 1u << 30, 0, 1024},
        {"rows of 3268 dwords (16-byte aligned only), 16 B/lane", 3268, 4, 1u << 30, 0, 1024},
        {"the same, 6 WGs/CU", 3268, 4, 1u << 30, 0, 26 * 1024},
        {"the same, 6 WGs/CU, bursts of 1 + 6 LDS trips", 3268, 4, 1, 6, 26 * 1024},
        {"unaligned, 6 WGs/CU (26 KB of LDS)", 3271, 1, 1u << 30, 0, 26 * 1024},
        {"unaligned, 6 WGs/CU, bursts of 13 + 8 LDS trips", 3271, 1, 13, 8, 26 * 1024},
        {"unaligned, 6 WGs/CU, bursts of 13 + 24 LDS trips", 3271, 1, 13, 24, 26 * 1024},
        {"unaligned, 6 WGs/CU, bursts of 3 + 2 LDS trips", 3271, 1, 3, 2, 26 * 1024},
        {"unaligned, 6 WGs/CU, bursts of 3 + 6 LDS trips", 3271, 1, 3, 6, 26 * 1024},
        {"aligned, 6 WGs/CU, bursts of 3 + 6 LDS trips", 3264, 1, 3, 6, 26 * 1024},
    };
    for (const Case &c : cases)
    {
        const uint32_t rows = (uint32_t) (bytes / 4 / c.R);
        const dim3 grid((rows + 3) / 4), block(256);
        float ms;
        const int reps = 5;
        auto launch = [&] {
            if (c.width == 1)
                hipLaunchKernelGGL(rowStoreKernel<1>, grid, block, c.ldsBytes, 0, a, rows, c.R, c.burst, c.gap, 0u);
            else
                hipLaunchKernelGGL(rowStoreKernel<4>, grid, block, c.ldsBytes, 0, a, rows, c.R, c.burst, c.gap, 0u);
        };
        launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++)
            launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-52s: %8.1f GB/s\n", c.name, (double) rows * c.R * 4 * reps / (ms * 1e-3) / 1e9);
    }
    return 0;
}
