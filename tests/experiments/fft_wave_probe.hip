// A transform core WITHOUT workgroup-wide LDS passes (round 5 experiment, priced in profiles/r05_INDEX.md): one WAVE owns a
// 4096-point complex transform, 64 points per lane.
//   1. lane l holds x[l + 64 j], j < 64, and runs a 64-point transform over j in registers (radix-4 decimation in frequency, fully
//      unrolled: the twiddles of those stages are compile-time constants, the digit-reversed output order is a renaming of registers);
//   2. Y[l][k2] *= W_4096^(l k2): the lane's own 64 twiddles as products of two sets of eight kept in registers,
//      W^(l (8 a + b)) = W^(8 l a) W^(l b);
//   3. ONE exchange through a wave-private LDS area -- lane k2 collects Y[l][k2] from every lane l -- real parts first, then
//      imaginary parts (64 x 65 floats: the values a lane has written are dead, what it reads takes their registers), no
//      s_barrier: a wave's LDS accesses complete in order; the values are read into digit-reversed registers, which is free;
//   4. a second 64-point transform in registers over l, by decimation in TIME (digit-reversed in, natural out): lane k2 holds
//      X[k2 + 64 k1] in register k1 -- natural order both sides, not one register move.
// Against the radix-8 / radix-16 cores (tests/experiments/fft_cores_probe.hip): a third of their LDS bytes per transform, no barriers,
// the same butterflies.  The probe runs pairs forward + inverse with the device full and checks one forward transform against a
// double-precision DFT on the host.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I lsp-dsp-units_amd/csrc -I include tests/experiments/fft_wave_probe.hip -o tests/experiments/fft_wave_probe
#include "fft_wave.h"
#include <cmath>
#include <cstdio>
#include <vector>
using namespace mi_fftw;

namespace
{
    constexpr int WAVES = 8;

    __global__ __launch_bounds__(64 * WAVES, 2)
    void probe(float2 *data, const float2 *__restrict__ tw, int reps, int check)
    {
        __shared__ float areas[WAVES][AREA];
        __shared__ float2 pl[8 * R];
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        fill_table(pl, tw, tid);
        v2f Q[8];
        load_lane_twiddles(Q, tw, lane);
        __syncthreads();
        float2 *seq = data + (size_t(blockIdx.x) * WAVES + wv) * N + lane;
        v2f x[R];
        #pragma unroll
        for (int j = 0; j < R; ++j)
            x[j] = mi_fft::ld2(seq + 64 * j);
        if (check)                                          // one forward transform
            fft4096<false>(x, pl, Q, areas[wv], lane);
        else
            for (int r = 0; r < reps; ++r)
            {
                fft4096<false>(x, pl, Q, areas[wv], lane);
                #pragma unroll
                for (int k1 = 0; k1 < R; ++k1)
                    x[k1] = x[k1] * (1.0f / N);
                fft4096<true>(x, pl, Q, areas[wv], lane);
            }
        #pragma unroll
        for (int j = 0; j < R; ++j)
            mi_fft::st2(seq + 64 * j, x[j]);
    }
}

namespace
{
    // a real frame of 8192 per wave: forward, split-filter-merge, inverse (what a block of the FIR equalizer is)
    __global__ __launch_bounds__(64 * WAVES, 2)
    void probe_conv(float2 *data, const float4 *__restrict__ ab, const float2 *__restrict__ tw, int reps)
    {
        __shared__ float areas[WAVES][AREA];
        __shared__ float2 pl[8 * R];
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        fill_table(pl, tw, tid);
        v2f Q[8];
        load_lane_twiddles(Q, tw, lane);
        __syncthreads();
        const __amdgpu_buffer_rsrc_t tab = table_of(ab);
        float2 *seq = data + (size_t(blockIdx.x) * WAVES + wv) * N + lane;      // (x[2 n], x[2 n + 1]), n = lane + 64 j
        v2f x[R];
        #pragma unroll
        for (int j = 0; j < R; ++j)
            x[j] = mi_fft::ld2(seq + 64 * j);
        for (int r = 0; r < reps; ++r)
        {
            float4 q[2 * AHEAD];
            fft4096<false>(x, pl, Q, areas[wv], lane, [&]() { table_ahead(q, tab, lane); });
            split_filter_merge(x, q, tab, lane);
            fft4096<true>(x, pl, Q, areas[wv], lane);
        }
        #pragma unroll
        for (int j = 0; j < R; ++j)
            mi_fft::st2(seq + 64 * j, x[j]);
    }
}

int main()
{
    std::vector<float2> tw(mi_fft::TWN);                    // the device table of the library: exp(-2 pi i j / TWN)
    for (int m = 0; m < mi_fft::TWN; ++m)
        tw[m] = make_float2(float(cos(-2.0 * PI * m / mi_fft::TWN)), float(sin(-2.0 * PI * m / mi_fft::TWN)));
    float2 *dtw;
    (void)hipMalloc(&dtw, tw.size() * sizeof(float2));
    (void)hipMemcpy(dtw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice);
    const int blocks = 512, reps = 40;                       // 512 x 8 = 4096 transforms in flight over the launch, as the cores' probe
    std::vector<float2> h(size_t(blocks) * WAVES * N), out(h.size());
    for (size_t i = 0; i < h.size(); ++i)
        h[i] = make_float2(float((i * 7919) % 1000) * 1e-3f - 0.5f, float((i * 104729) % 1000) * 1e-3f - 0.5f);
    float2 *d;
    (void)hipMalloc(&d, h.size() * sizeof(float2));
    // 1. one forward transform against the DFT (first sequence)
    (void)hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(64 * WAVES), 0, 0, d, dtw, 1, 1);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(out.data(), d, size_t(N) * sizeof(float2), hipMemcpyDeviceToHost);
    double worst = 0, peak = 0;
    for (int k = 0; k < N; k += 37)
    {
        double re = 0, im = 0;
        for (int n = 0; n < N; ++n)
        {
            const double a = -2.0 * PI * double((size_t(k) * n) % N) / double(N);
            re += h[n].x * cos(a) - h[n].y * sin(a);
            im += h[n].x * sin(a) + h[n].y * cos(a);
        }
        worst = fmax(worst, fmax(fabs(out[k].x - re), fabs(out[k].y - im)));
        peak = fmax(peak, fmax(fabs(re), fabs(im)));
    }
    printf("forward transform against a double-precision DFT (every 37th bin): worst error %.2e of the peak %.2f\n", worst / peak, peak);
    // 2. throughput: pairs forward + inverse
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep)
    {
        (void)hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice);
        (void)hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(64 * WAVES), 0, 0, d, dtw, reps, 0);
        (void)hipEventRecord(e1, nullptr);
        (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        best = fminf(best, ms);
    }
    (void)hipMemcpy(out.data(), d, h.size() * sizeof(float2), hipMemcpyDeviceToHost);
    double err = 0;
    for (size_t i = 0; i < h.size(); ++i) err = fmax(err, fmax(fabs(out[i].x - h[i].x), fabs(out[i].y - h[i].y)));
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe, 64 * WAVES, 0);
    printf("4096-point complex, one wave per transform (64 points per lane): %d workgroups (%d waves) per CU: %.2f ns per transform PAIR chip-wide "
           "(complex pairs: no real split / merge), round trip error after %d pairs %.1e\n", occ, occ * WAVES,
           double(best) * 1e6 / (double(blocks) * WAVES * reps), reps, err);
    // 3. the real pair with the response in between: one frame against a direct circular convolution, then the rate
    {
        const int L = 2 * N;                                 // real points
        std::vector<double> hr(L), cs(L), sn(L);
        for (int m = 0; m < L; ++m)
        {
            hr[m] = (m < 3000) ? exp(-m / 400.0) * cos(0.37 * m + 0.1 * (m % 7)) : 0.0;
            cs[m] = cos(2.0 * PI * m / L); sn[m] = sin(2.0 * PI * m / L);
        }
        std::vector<double> Hre(N + 1), Him(N + 1);
        for (int k = 0; k <= N; ++k)
        {
            double re = 0, im = 0;
            for (int m = 0; m < 3000; ++m)
            {
                const int i = int((size_t(k) * m) % L);
                re += hr[m] * cs[i]; im -= hr[m] * sn[i];
            }
            Hre[k] = re; Him[k] = im;
        }
        std::vector<float4> ab(N), ident(N);
        for (int k = 0; k < N; ++k)
        {
            // S, D = (H[k] +- conj H[N - k]) / 2
            const double sr = 0.5 * (Hre[k] + Hre[N - k]), si = 0.5 * (Him[k] - Him[N - k]);
            const double dr = 0.5 * (Hre[k] - Hre[N - k]), di = 0.5 * (Him[k] + Him[N - k]);
            const double wr = cos(PI * k / N), wi = -sin(PI * k / N);
            const double ar = sr + dr * wi, ai = si + di * wi, br = -di * wr, bi = dr * wr;      // beta = i D Re W
            const int lane = k & 63, r = k >> 6;
            ab[r * R + lane] = make_float4(float(ar / N), float(ai / N), float(br / N), float(bi / N));
            ident[r * R + lane] = make_float4(1.0f / N, 0.0f, 0.0f, 0.0f);
        }
        float4 *dab, *dident;
        (void)hipMalloc(&dab, N * sizeof(float4)); (void)hipMalloc(&dident, N * sizeof(float4));
        (void)hipMemcpy(dab, ab.data(), N * sizeof(float4), hipMemcpyHostToDevice);
        (void)hipMemcpy(dident, ident.data(), N * sizeof(float4), hipMemcpyHostToDevice);
        (void)hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe_conv, dim3(blocks), dim3(64 * WAVES), 0, 0, d, dab, dtw, 1);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(out.data(), d, size_t(N) * sizeof(float2), hipMemcpyDeviceToHost);
        const float *xr = reinterpret_cast<const float *>(h.data()), *yr = reinterpret_cast<const float *>(out.data());
        double werr = 0, wpeak = 0;
        for (int n = 0; n < L; n += 61)
        {
            double acc = 0;
            for (int m = 0; m < 3000; ++m)
                acc += hr[m] * xr[(n - m + L) % L];
            werr = fmax(werr, fabs(acc - yr[n]));
            wpeak = fmax(wpeak, fabs(acc));
        }
        printf("real frame of %d x response (forward, split-filter-merge in one step, inverse) against the circular convolution "
               "(every 61st sample): worst error %.2e of the peak %.2f\n", L, werr / wpeak, wpeak);
        float bestc = 1e30f;
        for (int rep = 0; rep < 4; ++rep)
        {
            (void)hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice);
            (void)hipEventRecord(e0, nullptr);
            hipLaunchKernelGGL(probe_conv, dim3(blocks), dim3(64 * WAVES), 0, 0, d, dident, dtw, reps);
            (void)hipEventRecord(e1, nullptr);
            (void)hipDeviceSynchronize();
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            bestc = fminf(bestc, ms);
        }
        (void)hipMemcpy(out.data(), d, h.size() * sizeof(float2), hipMemcpyDeviceToHost);
        double errc = 0;
        for (size_t i = 0; i < h.size(); ++i) errc = fmax(errc, fmax(fabs(out[i].x - h[i].x), fabs(out[i].y - h[i].y)));
        printf("real pair of %d (forward, split-filter-merge from a 64 KiB table in L2, inverse), one wave per frame: %.2f ns per pair chip-wide, "
               "round trip error after %d pairs %.1e   [fft_cores_probe, same work through LDS: 15.12 radix-8, 13.44 radix-16]\n",
               L, double(bestc) * 1e6 / (double(blocks) * WAVES * reps), reps, errc);
    }
    return 0;
}
