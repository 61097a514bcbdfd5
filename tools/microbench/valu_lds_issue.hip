/*
 * Issue-rate microbenchmark for the questions DESIGN.md section 4.1 turns on (gfx950):
 *   - how many cycles does one SIMD need per wave64 vector instruction (plain and packed f32), at 1 / 2 / 4 / 8 waves
 *     per SIMD?
 *   - how many cycles does the CU's LDS need per ds_read_b128 with ONE address for the whole wave (the broadcast read of
 *     a staged splat), per ds_write_b8 (the hit-list append) and per random 16-byte gather (the drain)?
 *   - do the two overlap when one wave issues both?
 * Build: hipcc --offload-arch=gfx950 -O3 -o valu_lds_issue valu_lds_issue.hip ; run without arguments.
 * Output: one line per (test, waves per SIMD): SIMD cycles per wave-instruction (wall: hipEvent time x 2.4 GHz is NOT
 * used; s_memtime stamps of every wave give the in-kernel cycle count).
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { T_FMA, T_PKFMA, T_MIX, T_LDS_BCAST, T_LDS_WRITE8, T_LDS_GATHER, T_TESTLOOP, T_TESTLOOP_NOLDS, T_CMPADD, T_COUNT };
static const char *NAMES[T_COUNT] = {"v_fma_f32 x64", "v_pk_fma_f32 x64", "v_fma + v_pk_fma alternating x64",
                                      "ds_read_b128 one address x16", "ds_write_b8 per-lane x16", "ds_read_b128 random gather x16",
                                      "test body (bcast read + 9 valu + write_b8) x8", "test body without LDS x8",
                                      "v_cmp + v_addc pairs x32"};

template<int T>
__global__ __launch_bounds__(256) void bench(int iters, unsigned long long *stamps, float *sink, const int *perm)
{
    __shared__ float4 lds[512];
    __shared__ unsigned char lists[256 * 20];
    const int tid = threadIdx.x;
    for (int i = tid; i < 512; i += 256)
        lds[i] = make_float4(i * 0.25f, i * 0.5f, i * 0.125f, 1.0f / (1 + i));
    __syncthreads();
    float a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3, a4 = tid * 0.5f, a5 = tid * 0.25f, a6 = 3, a7 = 4;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    const float b = 1.0000001f, c = 1e-9f;
    const f32x2 pb = {b, b}, pc = {c, c};
    typedef __attribute__((address_space(3))) unsigned char LdsByte;
    LdsByte *tail = (LdsByte *) &lists[tid * 20];
    LdsByte *const base = tail;
    const int gidx = perm[tid] & 511;
    float4 acc = make_float4(0, 0, 0, 0);
    const float cx = tid & 3, cy = (tid >> 2) & 3, cz = tid >> 4;
    const f32x2 cxy = {cx, cy};

    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++)
    {
        if (T == T_FMA)
        {
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(b), "v"(c));
            }
        }
        else if (T == T_PKFMA)
        {
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pb), "v"(pc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(pb), "v"(pc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(pb), "v"(pc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(pb), "v"(pc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p4) : "v"(pb), "v"(pc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p5) : "v"(pb), "v"(pc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p6) : "v"(pb), "v"(pc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p7) : "v"(pb), "v"(pc));
            }
        }
        else if (T == T_MIX)
        {
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(pb), "v"(pc));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(pb), "v"(pc));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p5) : "v"(pb), "v"(pc));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(b), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p7) : "v"(pb), "v"(pc));
            }
        }
        else if (T == T_LDS_BCAST)
        {
#pragma unroll
            for (int k = 0; k < 16; k++)
            {
                const float4 v = lds[(it * 16 + k) & 511];     /* one address per wave: a broadcast */
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        else if (T == T_LDS_WRITE8)
        {
#pragma unroll
            for (int k = 0; k < 16; k++)
            {
                *tail = (unsigned char) (it + k);
                tail += ((it + k + tid) & 3) == 0 ? 1 : 0;
                if (tail >= base + 16) tail = base;
                asm volatile("" : "+v"(tail));
            }
        }
        else if (T == T_LDS_GATHER)
        {
#pragma unroll
            for (int k = 0; k < 16; k++)
            {
                const float4 v = lds[(gidx + (it * 16 + k) * 37) & 511];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        else if (T == T_TESTLOOP || T == T_TESTLOOP_NOLDS)
        {
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                float4 a;
                if (T == T_TESTLOOP)
                    a = lds[(it * 8 + k) & 511];
                else
                {
                    a = make_float4(a0, a1, a2, a3);
                    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w));
                }
                const f32x2 pxy = f32x2{a.x, a.y} - cxy;
                const float pz = a.z - cz;
                const float pp = fmaf(pxy.x, pxy.x, fmaf(pxy.y, pxy.y, pz * pz));
                const float d = pp * a.w;
                if (T == T_TESTLOOP)
                    *tail = (unsigned char) k;
                tail += d < 0.99f ? 1 : 0;
                if (k == 7 && tail >= base + 8) tail = base;
                asm volatile("" : "+v"(tail));
            }
        }
        else if (T == T_CMPADD)
        {
#pragma unroll
            for (int k = 0; k < 32; k++)
            {
                tail += a0 < (float) (it + k) ? 1 : 0;
                asm volatile("" : "+v"(tail));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((tid & 63) == 0)
    {
        stamps[2 * (blockIdx.x * 4 + (tid >> 6))] = t0;
        stamps[2 * (blockIdx.x * 4 + (tid >> 6)) + 1] = t1;
    }
    sink[blockIdx.x * 256 + tid] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
        p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y + acc.x + acc.y + acc.z + acc.w + (float) (tail - base);
}

template<int T>
static void run(int wavesPerSimd, int instrPerIter)
{
    const int blocks = 256 * wavesPerSimd;      /* 256 CUs, one 256-thread block = one wave per SIMD */
    const int iters = 4000;
    unsigned long long *dStamps;
    float *dSink;
    int *dPerm;
    CHECK(hipMalloc(&dStamps, blocks * 8 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&dSink, blocks * 256 * sizeof(float)));
    std::vector<int> perm(256);
    for (int i = 0; i < 256; i++) perm[i] = rand();
    CHECK(hipMalloc(&dPerm, 256 * 4));
    CHECK(hipMemcpy(dPerm, perm.data(), 256 * 4, hipMemcpyHostToDevice));
    /* occupancy: LDS 16 KB + 13 KB per block -> at most 5 per CU by LDS; pad so that exactly wavesPerSimd blocks fit */
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++)
    {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<T>, dim3(blocks), dim3(256), 0, 0, iters, dStamps, dSink, (const int *) dPerm);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
    }
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(blocks * 8);
    CHECK(hipMemcpy(st.data(), dStamps, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> per;
    for (int i = 0; i < blocks * 4; i++)
        per.push_back((double) (st[2 * i + 1] - st[2 * i]));
    std::sort(per.begin(), per.end());
    const double med = per[per.size() / 2];
    const double n = (double) iters * instrPerIter;
    /* s_memtime ticks at 100 MHz on this part if it reads the constant-rate counter; report both interpretations */
    printf("%-48s waves/SIMD %d: median wave time %.0f ticks; per instr per wave %.3f ticks; per SIMD (divide by waves) %.3f; "
           "wall %.3f ms => %.2f cycles@2.4GHz per instr per SIMD\n",
           NAMES[T], wavesPerSimd, med, med / n, med / n / wavesPerSimd, ms, ms * 1e-3 * 2.4e9 / (n * wavesPerSimd));
    hipFree(dStamps); hipFree(dSink); hipFree(dPerm);
}

int main()
{
    for (int w : {1, 2, 4, 8})
    {
        run<T_FMA>(w, 64);
        run<T_PKFMA>(w, 64);
        run<T_MIX>(w, 64);
        run<T_CMPADD>(w, 64);
        run<T_LDS_BCAST>(w, 16);
        run<T_LDS_WRITE8>(w, 16);
        run<T_LDS_GATHER>(w, 16);
        run<T_TESTLOOP>(w, 8);
        run<T_TESTLOOP_NOLDS>(w, 8);
    }
    return 0;
}
