// Measures the shader clock the part holds under (a) a long VALU-bound loop and (b) short bursty launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(float *out, unsigned long long *stamps, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
    unsigned long long t0 = __builtin_readcyclecounter();           // s_memtime
    unsigned long long r0 = wall_clock64();                          // 100 MHz
    for (int i = 0; i < iters; ++i) { a = fmaf(a, b, c); c = fmaf(c, b, a); }
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long r1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + c;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
int main()
{
    float *out; unsigned long long *st;
    const int blocks = 1024;
    hipMalloc(&out, blocks * 64 * sizeof(float)); hipMalloc(&st, blocks * 2 * sizeof(unsigned long long));
    std::vector<unsigned long long> h(blocks * 2);
    for (int iters : {2000, 20000, 2000000})
    {
        for (int rep = 0; rep < 200; ++rep) hipLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, 0, out, st, iters);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), st, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0;
        for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; wall += h[2 * b + 1]; }
        printf("iters %8d: %.0f shader cycles in %.2f us -> %.3f GHz; %.2f cycles per dependent fma pair\n", iters, cyc / blocks,
               wall / blocks / 100.0, (cyc / blocks) / (wall / blocks / 100.0) / 1e3, (cyc / blocks) / iters);
    }
    return 0;
}
