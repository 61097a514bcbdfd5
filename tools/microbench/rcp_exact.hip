// Exhaustive check (all 2^32 float bit patterns): where does a short reciprocal -- v_rcp_f32 and one Newton step in fused
// arithmetic -- equal the correctly rounded 1.0f / x that -fhip-fp32-correctly-rounded-divide-sqrt emits (12 instructions)?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o rcp_exact rcp_exact.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float shortRcp(float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = fmaf(-x, r, 1.0f);          /* exact residual of r */
    return fmaf(r, e, r);
}

__device__ __forceinline__ float shortRcp2(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    r = fmaf(r, fmaf(-x, r, 1.0f), r);
    return fmaf(r, fmaf(-x, r, 1.0f), r);
}

__global__ void check(unsigned long long *bad, unsigned int *firstBad, int variant)
{
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride)
    {
        const float x = __uint_as_float((uint32_t) i);
        const float want = 1.0f / x;
        const float got = variant == 0 ? shortRcp(x) : shortRcp2(x);
        const uint32_t w = __float_as_uint(want), g = __float_as_uint(got);
        const bool same = w == g || (want != want && got != got);
        if (!same)
        {
            const uint32_t expo = ((uint32_t) i >> 23) & 0xFF;
            atomicAdd(&bad[expo], 1ull);
            atomicMin(&firstBad[expo], (uint32_t) i & 0x7FFFFFFFu);
        }
    }
}

int main()
{
    unsigned long long *dBad;
    unsigned int *dFirst;
    hipMalloc(&dBad, 256 * 8);
    hipMalloc(&dFirst, 256 * 4);
    for (int variant = 0; variant < 2; variant++)
    {
        hipMemset(dBad, 0, 256 * 8);
        hipMemset(dFirst, 0xFF, 256 * 4);
        hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, dBad, dFirst, variant);
        unsigned long long bad[256];
        unsigned int first[256];
        hipMemcpy(bad, dBad, sizeof(bad), hipMemcpyDeviceToHost);
        hipMemcpy(first, dFirst, sizeof(first), hipMemcpyDeviceToHost);
        unsigned long long total = 0;
        int lo = -1, hi = -1;
        for (int e = 0; e < 256; e++)
            if (bad[e])
            {
                total += bad[e];
                if (lo < 0) lo = e;
                hi = e;
            }
        printf("variant %d (%s): %llu of 2^32 inputs differ from 1.0f / x", variant, variant == 0 ? "rcp + 1 Newton step" : "rcp + 2 Newton steps", total);
        if (total)
        {
            printf("; biased exponents with differences:");
            for (int e = 0; e < 256; e++)
                if (bad[e])
                    printf(" %d(%llu, first 0x%08x)", e, bad[e], first[e]);
        }
        printf("\n");
    }
    return 0;
}
