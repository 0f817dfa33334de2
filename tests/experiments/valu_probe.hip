// Issue rate / dependent latency of v_fma_f32 and v_pk_fma_f32 on gfx950, one or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int CHAINS, bool PK>
__global__ void probe(float *out, unsigned long long *stamps, int iters)
{
    v2f a[CHAINS]; 
    for (int c = 0; c < CHAINS; ++c) a[c] = v2f{threadIdx.x * 1e-3f + c, 0.5f + c};
    const v2f b = v2f{0.999f, 0.998f}, d = v2f{0.01f, 0.02f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i)
    {
        #pragma unroll
        for (int r = 0; r < 8; ++r)
            #pragma unroll
            for (int c = 0; c < CHAINS; ++c)
            {
                if (PK) a[c] = __builtin_elementwise_fma(a[c], b, d);
                else    a[c].x = fmaf(a[c].x, b.x, d.x);
            }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int c = 0; c < CHAINS; ++c) s += a[c].x + a[c].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int CHAINS, bool PK>
void run(const char *name, int blocks, int threads)
{
    float *out; unsigned long long *st;
    hipMalloc(&out, blocks * threads * sizeof(float)); hipMalloc(&st, blocks * 8 * sizeof(unsigned long long));
    const int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<CHAINS, PK>), dim3(blocks), dim3(threads), 0, 0, out, st, iters);
    hipDeviceSynchronize();
    const int waves = blocks * threads / 64;
    std::vector<unsigned long long> h(waves);
    hipMemcpy(h.data(), st, waves * sizeof(h[0]), hipMemcpyDeviceToHost);
    double cyc = 0; for (auto v : h) cyc += v;
    printf("%-28s waves/SIMD %.0f: %.2f cycles per instruction per wave\n", name, waves / 1024.0, cyc / waves / (iters * 8.0 * CHAINS));
    hipFree(out); hipFree(st);
}
int main()
{
    for (int wps : {1, 2, 4})
    {
        const int blocks = 256 * wps;       // 256-thread blocks: one wave on each SIMD of a CU per block
        run<1, false>("v_fma_f32 1 chain", blocks, 256);
        run<2, false>("v_fma_f32 2 chains", blocks, 256);
        run<4, false>("v_fma_f32 4 chains", blocks, 256);
        run<1, true>("v_pk_fma_f32 1 chain", blocks, 256);
        run<2, true>("v_pk_fma_f32 2 chains", blocks, 256);
        run<4, true>("v_pk_fma_f32 4 chains", blocks, 256);
    }
    return 0;
}
