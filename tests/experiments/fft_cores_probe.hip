// Steady-state throughput of the two transform cores at their common interface (a real transform pair through LDS:
// real_fft::forward + ::inverse), the device filled with workgroups: radix-8 (fft_device.h, T = M / 8, two LDS buffers) against
// radix-16 (fft16.h, T = M / 16, one buffer -- twice the workgroups fit a CU's LDS).  Round 5's question: the launches built on
// the radix-8 core are bound by their passes through LDS (profiles/r05_experiments/fft_kernels_issue_and_occupancy.txt); does the
// core with a third fewer exchanges win where the work around the transform is small?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I lsp-dsp-units_amd/csrc tests/experiments/fft_cores_probe.hip -o tests/experiments/fft_cores_probe
#include "fft_device.h"
#include "fft16.h"
#include <cmath>
#include <cstdio>
#include <vector>
using mi_fft::TWN;

template <int LOGM, bool R16> using core = mi_fft16::fsel<LOGM, R16>;

template <int LOGM, bool R16>
__global__ __launch_bounds__((core<LOGM, R16>::T))
void probe(float2 *data, const float2 *__restrict__ tw, int reps)
{
    using PL = mi_fft16::fsel<LOGM, R16>;
    constexpr int M = PL::N, T = PL::T;
    __shared__ float2 lds_[PL::LDS];
    float2 *const buf = lds_, *const scr = lds_ + PL::SCR;
    const int tid = threadIdx.x;
    typename PL::real rf;
    rf.load(tw, TWN, tid);
    rf.prepare();
    for (int k = tid; k < M; k += T)
        buf[k] = data[size_t(blockIdx.x) * M + k];
    __syncthreads();
    for (int r = 0; r < reps; ++r)
    {
        rf.forward(buf, scr, tid);
        rf.inverse(buf, scr, tid);
        for (int k = tid; k < M; k += T)
            buf[k] = make_float2(buf[k].x * (0.5f / M), buf[k].y * (0.5f / M));
        __syncthreads();
    }
    for (int k = tid; k < M; k += T)
        data[size_t(blockIdx.x) * M + k] = buf[k];
}

template <int LOGM, bool R16>
void run(int blocks, int reps, const float2 *dtw)
{
    using PL = mi_fft16::fsel<LOGM, R16>;
    constexpr int M = 1 << LOGM;
    std::vector<float2> h(size_t(blocks) * M), out(h.size());
    for (size_t i = 0; i < h.size(); ++i)
        h[i] = make_float2(float((i * 7919) % 1000) * 1e-3f - 0.5f, float((i * 104729) % 1000) * 1e-3f - 0.5f);
    float2 *d;
    (void)hipMalloc(&d, h.size() * sizeof(float2));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep)
    {
        (void)hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice);
        (void)hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL((probe<LOGM, R16>), dim3(blocks), dim3(PL::T), 0, 0, d, dtw, reps);
        (void)hipEventRecord(e1, nullptr);
        (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        best = fminf(best, ms);
    }
    (void)hipMemcpy(out.data(), d, h.size() * sizeof(float2), hipMemcpyDeviceToHost);
    double err = 0;
    for (size_t i = 0; i < h.size(); ++i) err = fmax(err, fmax(fabs(out[i].x - h[i].x), fabs(out[i].y - h[i].y)));
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe<LOGM, R16>, PL::T, 0);
    printf("%5d-point complex (real pair of %5d), %-8s T %4d, LDS %6zu B, %d workgroups (%2d waves) per CU: %7.2f ns per transform pair chip-wide, "
           "round trip error %.1e\n", M, 2 * M, R16 ? "radix-16" : "radix-8", PL::T, sizeof(float2) * PL::LDS, occ, occ * PL::T / 64,
           double(best) * 1e6 / (double(blocks) * reps), err);
    (void)hipFree(d);
}

int main()
{
    const size_t total = size_t(TWN) + 2 * size_t(mi_fft16::table16_total());
    std::vector<float2> tw(total);
    for (int j = 0; j < TWN; ++j)
        tw[j] = make_float2(float(cos(-2.0 * M_PI * j / TWN)), float(sin(-2.0 * M_PI * j / TWN)));
    mi_fft16::table16_build(reinterpret_cast<float *>(tw.data() + TWN));
    float2 *dtw;
    (void)hipMalloc(&dtw, total * sizeof(float2));
    (void)hipMemcpy(dtw, tw.data(), total * sizeof(float2), hipMemcpyHostToDevice);
    const int blocks = 4096, reps = 40;
    run<10, false>(blocks, reps, dtw); run<10, true>(blocks, reps, dtw);
    run<11, false>(blocks, reps, dtw); run<11, true>(blocks, reps, dtw);
    run<12, false>(blocks, reps, dtw); run<12, true>(blocks, reps, dtw);
    return 0;
}
