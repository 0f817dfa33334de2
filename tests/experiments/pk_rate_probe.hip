// How many shader clocks does a SIMD of gfx950 spend per wave64 VALU instruction?  v_fma_f32 against v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32,
// 1, 2 and 4 waves per SIMD, sixteen independent accumulators per wave (no dependent issue).  Round 5 experiment: the wave-resident
// transforms (fft_wave.h) measured 7.5 clocks per (mostly packed) instruction whatever the number of waves.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tests/experiments/pk_rate_probe.hip -o tests/experiments/pk_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void spin(float *out, unsigned long long *stamps, int iters)
{
    v2f a[16], b = v2f{1.0001f, 0.9999f}, c = v2f{0.5f, 0.25f};
    for (int i = 0; i < 16; ++i) a[i] = v2f{threadIdx.x * 1e-3f + i, 1.0f + i};
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it)
    {
        #pragma unroll
        for (int i = 0; i < 16; ++i)
        {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(a[i]) : "v"(b), "v"(c));
            if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(a[i]) : "s"(b));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main()
{
    float *out; unsigned long long *st;
    const int cus = 256, iters = 20000;
    (void)hipMalloc(&out, cus * 8 * 4 * 64 * sizeof(float)); (void)hipMalloc(&st, cus * 2 * 2 * sizeof(unsigned long long));
    std::vector<unsigned long long> h(cus * 2 * 2);
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32 with op_sel / neg", "v_pk_mul_f32 with an SGPR pair"};
    for (int kind = 0; kind < 6; ++kind)
        for (int waves : {1, 2, 4, 8})                      // per SIMD: workgroups of 4 * min(waves, 4) waves, waves / 4 of them per CU
        {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep)
            {
                (void)hipEventRecord(e0, nullptr);
                const dim3 g(cus * (waves > 4 ? waves / 4 : 1)), blk(64 * 4 * (waves > 4 ? 4 : waves));
                switch (kind)
                {
                    case 0: hipLaunchKernelGGL(spin<0>, g, blk, 0, 0, out, st, iters); break;
                    case 1: hipLaunchKernelGGL(spin<1>, g, blk, 0, 0, out, st, iters); break;
                    case 2: hipLaunchKernelGGL(spin<2>, g, blk, 0, 0, out, st, iters); break;
                    case 3: hipLaunchKernelGGL(spin<3>, g, blk, 0, 0, out, st, iters); break;
                    case 4: hipLaunchKernelGGL(spin<4>, g, blk, 0, 0, out, st, iters); break;
                    default: hipLaunchKernelGGL(spin<5>, g, blk, 0, 0, out, st, iters); break;
                }
                (void)hipEventRecord(e1, nullptr);
                (void)hipDeviceSynchronize();
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            (void)hipMemcpy(h.data(), st, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
            double cyc = 0, wall = 0;
            const int wgs = cus * (waves > 4 ? waves / 4 : 1);
            for (int b = 0; b < wgs; ++b) { cyc += double(h[2 * b]); wall += double(h[2 * b + 1]); }
            cyc /= (wgs / cus); wall /= (wgs / cus);
            printf("%-34s %d wave(s) per SIMD: %.2f clocks of the SIMD per instruction (%.3f GHz); the launch took %.3f ms = %.2f ns of a SIMD per instruction\n",
                   names[kind], waves, cyc / cus / (double(iters) * 16.0 * waves), cyc / wall / 10.0, ms, double(ms) * 1e6 / (double(iters) * 16.0 * waves));
        }
    return 0;
}
