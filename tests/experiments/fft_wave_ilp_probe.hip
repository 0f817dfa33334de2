// Does a wave that has its SIMD to itself issue faster with TWO independent transforms in its instruction stream?  (round 5 experiment)
// One wave per SIMD (four per workgroup, 512 registers): (a) one 4096-point transform pair after the other, (b) two sequences per wave,
// their register stages side by side in the same basic blocks (the exchanges one after the other through the wave's one LDS area).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -I lsp-dsp-units_amd/csrc -I include tests/experiments/fft_wave_ilp_probe.hip -o tests/experiments/fft_wave_ilp_probe
#include "fft_wave.h"
#include <cmath>
#include <cstdio>
#include <vector>
using namespace mi_fftw;

namespace
{
    constexpr int WAVES = 4;

    template <bool INV>
    __device__ __forceinline__ void twiddle(v2f (&x)[R], const float2 *plq, int lane)
    {
        v2f Q[8];
        #pragma unroll
        for (int b = 0; b < 8; ++b)
            Q[b] = mi_fft::ld2(plq + (8 + b) * R + lane);
        #pragma unroll
        for (int a = 0; a < 8; ++a)
        {
            const v2f Pa = mi_fft::ld2(plq + a * R + lane);
            #pragma unroll
            for (int b = 0; b < 8; ++b)
            {
                if (a == 0 && b == 0)
                    continue;
                const v2f w = (a == 0) ? Q[b] : (b == 0) ? Pa : pmul<false>(Pa, Q[b]);
                const int k2 = 8 * a + b;
                x[rev4_6(k2)] = INV ? pmul<true>(w, x[rev4_6(k2)]) : pmul<false>(w, x[rev4_6(k2)]);
            }
        }
    }

    template <bool INV>
    __device__ __forceinline__ void fft4096_two(v2f (&x)[R], v2f (&y)[R], const float2 *plq, float *area, int lane)
    {
        fft64_dif<INV>(x);
        fft64_dif<INV>(y);
        twiddle<INV>(x, plq, lane);
        twiddle<INV>(y, plq, lane);
        exchange(x, area, lane);
        exchange(y, area, lane);
        fft64_dit<INV>(x);
        fft64_dit<INV>(y);
    }

    template <int TWO>
    __global__ __launch_bounds__(64 * WAVES, 1)
    void probe(float2 *data, const float2 *__restrict__ tw, int reps, unsigned long long *stamps)
    {
        __shared__ float areas[WAVES][AREA];
        __shared__ float2 pl[16 * R];
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        fill_table_pq(pl, tw, tid, 64 * WAVES);
        __syncthreads();
        float2 *seq = data + (size_t(blockIdx.x) * WAVES + wv) * 2 * N + lane;
        v2f x[R], y[R];
        #pragma unroll
        for (int j = 0; j < R; ++j)
        {
            x[j] = mi_fft::ld2(seq + 64 * j);
            y[j] = mi_fft::ld2(seq + N + 64 * j);
        }
        const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
        for (int r = 0; r < reps; ++r)
        {
            if (TWO)
            {
                fft4096_two<false>(x, y, pl, areas[wv], lane);
                #pragma unroll
                for (int k = 0; k < R; ++k) { x[k] = x[k] * (1.0f / N); y[k] = y[k] * (1.0f / N); }
                fft4096_two<true>(x, y, pl, areas[wv], lane);
            }
            else
            {
                fft4096_t<false>(x, pl, areas[wv], lane);
                #pragma unroll
                for (int k = 0; k < R; ++k) x[k] = x[k] * (1.0f / N);
                fft4096_t<true>(x, pl, areas[wv], lane);
                fft4096_t<false>(y, pl, areas[wv], lane);
                #pragma unroll
                for (int k = 0; k < R; ++k) y[k] = y[k] * (1.0f / N);
                fft4096_t<true>(y, pl, areas[wv], lane);
            }
        }
        if (tid == 0) { stamps[2 * blockIdx.x] = __builtin_readcyclecounter() - t0; stamps[2 * blockIdx.x + 1] = wall_clock64() - r0; }
        #pragma unroll
        for (int j = 0; j < R; ++j)
        {
            mi_fft::st2(seq + 64 * j, x[j]);
            mi_fft::st2(seq + N + 64 * j, y[j]);
        }
    }
}

int main()
{
    std::vector<float2> tw(mi_fft::TWN);
    for (int m = 0; m < mi_fft::TWN; ++m)
        tw[m] = make_float2(float(cos(-2.0 * PI * m / mi_fft::TWN)), float(sin(-2.0 * PI * m / mi_fft::TWN)));
    float2 *dtw;
    (void)hipMalloc(&dtw, tw.size() * sizeof(float2));
    (void)hipMemcpy(dtw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice);
    const int blocks = 512, reps = 20;
    std::vector<float2> h(size_t(blocks) * WAVES * 2 * N), out(h.size());
    for (size_t i = 0; i < h.size(); ++i)
        h[i] = make_float2(float((i * 7919) % 1000) * 1e-3f - 0.5f, float((i * 104729) % 1000) * 1e-3f - 0.5f);
    float2 *d;
    (void)hipMalloc(&d, h.size() * sizeof(float2));
    unsigned long long *dst;
    (void)hipMalloc(&dst, blocks * 2 * sizeof(unsigned long long));
    std::vector<unsigned long long> st(blocks * 2);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int two = 0; two < 2; ++two)
    {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep)
        {
            (void)hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice);
            (void)hipEventRecord(e0, nullptr);
            if (two) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(64 * WAVES), 0, 0, d, dtw, reps, dst);
            else     hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(64 * WAVES), 0, 0, d, dtw, reps, dst);
            (void)hipEventRecord(e1, nullptr);
            (void)hipDeviceSynchronize();
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            best = fminf(best, ms);
        }
        (void)hipMemcpy(out.data(), d, h.size() * sizeof(float2), hipMemcpyDeviceToHost);
        double err = 0;
        for (size_t i = 0; i < h.size(); ++i) err = fmax(err, fmax(fabs(out[i].x - h[i].x), fabs(out[i].y - h[i].y)));
        // a wave does 2 * reps pairs; blocks / 256 workgroups per CU in turn
        (void)hipMemcpy(st.data(), dst, st.size() * sizeof(st[0]), hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0;
        for (int b = 0; b < blocks; ++b) { cyc += double(st[2 * b]); wall += double(st[2 * b + 1]); }
        printf("   shader clock inside the loop: %.3f GHz; %.0f shader cycles per transform pair and wave\n", cyc / wall / 10.0, cyc / blocks / (2.0 * reps));
        printf("%s: %.2f us per transform PAIR and wave (one wave per SIMD), round trip error %.1e\n",
               two ? "two sequences side by side" : "one sequence after the other",
               double(best) * 1e3 / (2.0 * reps * (blocks / 256.0)), err);
    }
    return 0;
}
