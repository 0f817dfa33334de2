// Cost of the in-LDS real FFT pair (forward + inverse) in isolation: one workgroup per CU, R repetitions.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I lsp-dsp-units_amd/csrc tests/experiments/fft_probe.hip -o tests/experiments/fft_probe
#include "fft_device.h"
#include <cmath>
#include <cstdio>
#include <vector>
using namespace mi_fft;


template <int LOGM>
__global__ __launch_bounds__(plan<LOGM>::T)
void probe(float2 *data, const float2 *__restrict__ tw, int reps, unsigned long long *cycles)
{
    constexpr int M = plan<LOGM>::N, T = plan<LOGM>::T;
    __shared__ float2 buf[M], scr[M];
    const int tid = threadIdx.x;
    real_fft<LOGM> rf;
    rf.load(tw, TWN, tid);
    rf.prepare();
    for (int k = tid; k < M; k += T)
        buf[k] = data[size_t(blockIdx.x) * M + k];
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r)
    {
        rf.forward(buf, scr, tid);
        rf.inverse(buf, scr, tid);
        for (int k = tid; k < M; k += T)
            buf[k] = make_float2(buf[k].x * (0.5f / M), buf[k].y * (0.5f / M));
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    for (int k = tid; k < M; k += T)
        data[size_t(blockIdx.x) * M + k] = buf[k];
    if (tid == 0)
        cycles[blockIdx.x] = t1 - t0;
}

template <int LOGM>
void run(int blocks)
{
    constexpr int M = 1 << LOGM;
    std::vector<float2> tw(TWN), h(size_t(blocks) * M);
    for (int j = 0; j < TWN; ++j)
        tw[j] = make_float2(float(cos(-2.0 * M_PI * j / TWN)), float(sin(-2.0 * M_PI * j / TWN)));
    for (size_t i = 0; i < h.size(); ++i)
        h[i] = make_float2(float((i * 7919) % 1000) * 1e-3f - 0.5f, float((i * 104729) % 1000) * 1e-3f - 0.5f);
    float2 *d, *dtw; unsigned long long *dc;
    (void)hipMalloc(&d, h.size() * sizeof(float2)); (void)hipMalloc(&dtw, TWN * sizeof(float2)); (void)hipMalloc(&dc, blocks * 8);
    (void)hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice);
    (void)hipMemcpy(dtw, tw.data(), TWN * sizeof(float2), hipMemcpyHostToDevice);
    const int reps = 20;
    hipLaunchKernelGGL((probe<LOGM>), dim3(blocks), dim3(plan<LOGM>::T), 0, 0, d, dtw, reps, dc);
    (void)hipDeviceSynchronize();
    std::vector<float2> out(h.size());
    std::vector<unsigned long long> c(blocks);
    (void)hipMemcpy(out.data(), d, h.size() * sizeof(float2), hipMemcpyDeviceToHost);
    (void)hipMemcpy(c.data(), dc, blocks * 8, hipMemcpyDeviceToHost);
    double err = 0, cyc = 0;
    for (size_t i = 0; i < h.size(); ++i) err = fmax(err, fmax(fabs(out[i].x - h[i].x), fabs(out[i].y - h[i].y)));
    for (auto v : c) cyc += double(v);
    printf("LOGM %2d, %4d blocks x %4d threads: %.0f cycles per forward+inverse pair (round-trip error %.2e)\n",
           LOGM, blocks, plan<LOGM>::T, cyc / blocks / reps, err);
    (void)hipFree(d); (void)hipFree(dtw); (void)hipFree(dc);
}

int main()
{
    run<12>(256);
    run<12>(512);
    run<11>(1024);
    run<10>(1024);
    run<8>(1024);
    return 0;
}
