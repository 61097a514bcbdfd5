// Streaming bandwidth of one MI355X as this path's kernels see it: pure store, pure load (sum), copy.
// Build and run ON THE GPU BOX:
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_bw tools/microbench/stream_bw.hip && /tmp/stream_bw
// Prints GB/s per pattern for 16-byte and 4-byte accesses; the write rate is what bounds latticeTriangles / latticeVertices.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template<typename T>
__global__ __launch_bounds__(256) void storeKernel(T *dst, uint64_t n, T v)
{
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t) gridDim.x * 256)
        dst[i] = v;
}

template<typename T>
__global__ __launch_bounds__(256) void copyKernel(T *dst, const T *src, uint64_t n)
{
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t) gridDim.x * 256)
        dst[i] = src[i];
}

__global__ __launch_bounds__(256) void loadKernel(const uint4 *src, uint64_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t) gridDim.x * 256)
    {
        const uint4 v = src[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u)
        *out = acc;
}

int main()
{
    const uint64_t bytes = 4ull << 30;
    void *a, *b;
    uint32_t *out;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMalloc((void **) &out, 4));
    CHECK(hipMemset(a, 1, bytes));
    CHECK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grids[] = {2048, 8192, 32768, 131072};
    for (int g : grids)
    {
        float ms;
        const int reps = 5;
#define TIME(name, bytesMoved, launch)                                                              \
        launch; CHECK(hipDeviceSynchronize());                                                      \
        CHECK(hipEventRecord(e0));                                                                  \
        for (int r = 0; r < reps; r++) { launch; }                                                  \
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));                                  \
        CHECK(hipEventElapsedTime(&ms, e0, e1));                                                    \
        printf("%-28s grid %6d: %8.1f GB/s\n", name, g, (double) (bytesMoved) * reps / (ms * 1e-3) / 1e9);
        TIME("store 16 B / thread", bytes, (storeKernel<uint4><<<dim3(g), dim3(256)>>>((uint4 *) a, bytes / 16, make_uint4(1, 2, 3, 4))))
        TIME("store 4 B / thread", bytes, (storeKernel<uint32_t><<<dim3(g), dim3(256)>>>((uint32_t *) a, bytes / 4, 7u)))
        TIME("load 16 B / thread", bytes, (loadKernel<<<dim3(g), dim3(256)>>>((const uint4 *) a, bytes / 16, out)))
        TIME("copy 16 B (read + write)", 2 * bytes, (copyKernel<uint4><<<dim3(g), dim3(256)>>>((uint4 *) b, (const uint4 *) a, bytes / 16)))
        TIME("copy 4 B (read + write)", 2 * bytes, (copyKernel<uint32_t><<<dim3(g), dim3(256)>>>((uint32_t *) b, (const uint32_t *) a, bytes / 4)))
    }
    float ms;
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) CHECK(hipMemsetAsync(a, 0, bytes, 0));
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s             : %8.1f GB/s\n", "hipMemsetAsync", (double) bytes * 5 / (ms * 1e-3) / 1e9);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) CHECK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0));
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s             : %8.1f GB/s\n", "hipMemcpyAsync d2d (r + w)", (double) 2 * bytes * 5 / (ms * 1e-3) / 1e9);
    return 0;
}
