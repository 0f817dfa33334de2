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
#include "fft_device.h"
#include "mi_common.h"
#include <cmath>
#include <cstdio>
#include <vector>
using mi_fft::v2f;
using mi_fft::padd_i;
using mi_fft::pmul;

namespace
{
    constexpr int N = 4096, R = 64, WAVES = 8, PITCH = 65;                 // PITCH: floats per row of the exchange area

    constexpr int rev4_6(int k)                             // base-4 digit reversal of a 6-bit index
    {
        return ((k & 3) << 4) | (k & 12) | ((k >> 4) & 3);
    }

    constexpr double PI = 3.14159265358979323846;

    // (wx + i wy) b with the constant in a pair of SGPRs: mi_fft::pmul wants its operands in VGPRs, and 108 constants hoisted out
    // of the loop as VGPR pairs are what spilled.  The empty volatile asm pins the s_mov's next to their use.
    __device__ __forceinline__ v2f pmul_c(float wx, float wy, v2f b)
    {
        v2f w{wx, wy}, t, r;
        asm volatile("" : "+s"(w));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "s"(w), "v"(b));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "s"(w), "v"(b), "v"(t));
        return r;
    }

    // 64-point transform over the registers, in place; out[k] = x[rev4_6(k)] afterwards.  INV: e^{+i}.
    template <bool INV>
    __device__ __forceinline__ void fft64(v2f (&x)[R])
    {
        #pragma unroll
        for (int len = 64; len >= 4; len /= 4)
        {
            const int q = len / 4;
            #pragma unroll
            for (int g = 0; g < R; g += len)
                #pragma unroll
                for (int j = 0; j < q; ++j)
                {
                    const v2f a0 = x[g + j], a1 = x[g + j + q], a2 = x[g + j + 2 * q], a3 = x[g + j + 3 * q];
                    const v2f t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = a1 - a3;
                    v2f y0 = t0 + t2, y2 = t0 - t2;
                    v2f y1 = INV ? padd_i<true>(t1, d) : padd_i<false>(t1, d);      // t1 -+ i d
                    v2f y3 = INV ? padd_i<false>(t1, d) : padd_i<true>(t1, d);
                    if (j > 0)
                    {
                        const double a = (INV ? 2.0 : -2.0) * PI * double(j) / double(len);
                        y1 = pmul_c(float(__builtin_cos(a)), float(__builtin_sin(a)), y1);
                        y2 = pmul_c(float(__builtin_cos(2 * a)), float(__builtin_sin(2 * a)), y2);
                        y3 = pmul_c(float(__builtin_cos(3 * a)), float(__builtin_sin(3 * a)), y3);
                    }
                    x[g + j] = y0; x[g + j + q] = y1; x[g + j + 2 * q] = y2; x[g + j + 3 * q] = y3;
                }
        }
    }

    // the same transform by decimation in time: in: x[rev4_6(n)] = in[n]; out: natural order.
    template <bool INV>
    __device__ __forceinline__ void fft64_dit(v2f (&x)[R])
    {
        #pragma unroll
        for (int len = 4; len <= 64; len *= 4)
        {
            const int q = len / 4;
            #pragma unroll
            for (int g = 0; g < R; g += len)
                #pragma unroll
                for (int j = 0; j < q; ++j)
                {
                    v2f a0 = x[g + j], a1 = x[g + j + q], a2 = x[g + j + 2 * q], a3 = x[g + j + 3 * q];
                    if (j > 0)
                    {
                        const double a = (INV ? 2.0 : -2.0) * PI * double(j) / double(len);
                        a1 = pmul_c(float(__builtin_cos(a)), float(__builtin_sin(a)), a1);
                        a2 = pmul_c(float(__builtin_cos(2 * a)), float(__builtin_sin(2 * a)), a2);
                        a3 = pmul_c(float(__builtin_cos(3 * a)), float(__builtin_sin(3 * a)), a3);
                    }
                    const v2f t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = a1 - a3;
                    x[g + j] = t0 + t2;
                    x[g + j + q] = INV ? padd_i<true>(t1, d) : padd_i<false>(t1, d);
                    x[g + j + 2 * q] = t0 - t2;
                    x[g + j + 3 * q] = INV ? padd_i<false>(t1, d) : padd_i<true>(t1, d);
                }
        }
    }

    // the exchange: in: lane l holds x[rev4_6(k2)] = Y[l][k2]; out: lane k2 holds x[rev4_6(l)] = Y[l][k2] (the order the second
    // transform wants: which register a value is read into is free).  area: 64 x PITCH floats of this wave.
    __device__ __forceinline__ void exchange(v2f (&x)[R], float *area, int lane)
    {
        asm volatile("" : "+v"(lane));                      // (the sixteen bases the reads want are computed here, not kept across
                                                            //  the loop in registers this kernel does not have)
        float *row = area + lane * PITCH;                   // row = source lane, column = k2
        #pragma unroll
        for (int k2 = 0; k2 < R; ++k2)
            row[k2] = x[rev4_6(k2)].x;
        __builtin_amdgcn_wave_barrier();
        #pragma unroll
        for (int l = 0; l < R; ++l)
            x[rev4_6(l)].x = area[l * PITCH + lane];        // (same wave: the writes above are performed first)
        __builtin_amdgcn_wave_barrier();
        #pragma unroll
        for (int k2 = 0; k2 < R; ++k2)
            row[k2] = x[rev4_6(k2)].y;
        __builtin_amdgcn_wave_barrier();
        #pragma unroll
        for (int l = 0; l < R; ++l)
            x[rev4_6(l)].y = area[l * PITCH + lane];
        __builtin_amdgcn_wave_barrier();
    }

    // a whole transform of the wave, natural order both sides: x[j] = in[lane + 64 j]  ->  x[k1] = out[lane + 64 k1].
    // Q[b] = W_4096^(l b) in registers, P[a] = W_4096^(8 l a) from the workgroup's table pl[a][l] (this lane's l), a, b < 8.
    struct no_hook { __device__ __forceinline__ void operator()() const { } };
    template <bool INV, class HOOK = no_hook>
    __device__ __forceinline__ void fft4096(v2f (&x)[R], const float2 *pl, const v2f (&Q)[8], float *area, int lane, HOOK mid = HOOK())
    {
        fft64<INV>(x);
        #pragma unroll
        for (int a = 0; a < 8; ++a)
        {
            const v2f Pa = mi_fft::ld2(pl + a * R + lane);
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
        exchange(x, area, lane);
        mid();                                              // (loads the step after this transform wants: in flight over its second half)
        fft64_dit<INV>(x);
    }

    __global__ __launch_bounds__(64 * WAVES, 2)
    void probe(float2 *data, const float2 *__restrict__ tw /* W_4096^m, m < 4096 */, int reps, int check)
    {
        __shared__ float areas[WAVES][R * PITCH];
        __shared__ float2 pl[8 * R];
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        pl[tid] = tw[(8 * (tid & 63) * (tid >> 6)) & (N - 1)];
        v2f Q[8];
        #pragma unroll
        for (int b = 0; b < 8; ++b)
            Q[b] = mi_fft::ld2(tw + lane * b);
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
    // alpha x + beta conj(c) in four packed instructions
    __device__ __forceinline__ v2f fused_bin(float4 ab, v2f x, v2f c)
    {
        const v2f al{ab.x, ab.y}, be{ab.z, ab.w};
        v2f t;
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(al), "v"(x));
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(t) : "v"(al), "v"(x));
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "+v"(t) : "v"(be), "v"(c));
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(t) : "v"(be), "v"(c));
        return t;
    }

    __device__ __forceinline__ v2f from_partner(int addr, v2f v)
    {
        return v2f{__int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v.x))),
                   __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v.y)))};
    }

    // The packed spectrum of a real frame of 8192 (bin k = lane + 64 r in register r), times a real response's spectrum, packed
    // again -- split, product and merge of the workgroup cores as ONE step per bin:
    //     Zy[k] = alpha[k] Z[k] + beta[k] conj(Z[N - k]),
    //     alpha = S + D Im W,  beta = i D Re W,  S, D = (H[k] +- conj H[N - k]) / 2,  W = e^{-i pi k / N}   (k = 0: H[0], H[N] real)
    // (tables per response, made where the response is parsed).  The partner N - k sits in lane 64 - lane, register 63 - r; lane 0
    // pairs with itself, register (64 - r) & 63 -- the one select per value.
    constexpr int AHEAD = 4;                                // iterations of split_filter_merge whose table rows are in flight
    // row r of the table for this lane: ONE lane offset in a VGPR, the row in the scalar offset (64-bit row pointers are what spilled)
    __device__ __forceinline__ float4 table_row(__amdgpu_buffer_rsrc_t tab, int lane16, int r)
    {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 d = __builtin_amdgcn_raw_buffer_load_b128(tab, lane16, r * R * int(sizeof(float4)), 0);
        return make_float4(__uint_as_float(d.x), __uint_as_float(d.y), __uint_as_float(d.z), __uint_as_float(d.w));
    }
    __device__ __forceinline__ void table_ahead(float4 (&q)[2 * AHEAD], __amdgpu_buffer_rsrc_t tab, int lane)
    {
        #pragma unroll
        for (int r = 0; r < AHEAD; ++r)
        {
            q[2 * r] = table_row(tab, lane * 16, r);
            q[2 * r + 1] = table_row(tab, lane * 16, 63 - r);
        }
    }
    __device__ __forceinline__ void split_filter_merge(v2f (&x)[R], float4 (&q)[2 * AHEAD], __amdgpu_buffer_rsrc_t tab /* [r][lane] */, int lane)
    {
        const int paddr = ((64 - lane) & 63) * 4;
        const bool l0 = lane == 0;
        v2f saved = x[0];
        #pragma unroll
        for (int r = 0; r < R / 2; ++r)
        {
            const int r2 = 63 - r, s = r % AHEAD;
            const float4 ab1 = q[2 * s], ab2 = q[2 * s + 1];
            if (r + AHEAD < R / 2)
            {
                q[2 * s] = table_row(tab, lane * 16, r + AHEAD);
                q[2 * s + 1] = table_row(tab, lane * 16, r2 - AHEAD);
            }
            const v2f t1 = from_partner(paddr, x[r2]), t2 = from_partner(paddr, x[r]);
            const v2f own1 = (r == 0) ? x[0] : saved, own2 = x[r + 1];
            const v2f c1 = v2f{l0 ? own1.x : t1.x, l0 ? own1.y : t1.y}, c2 = v2f{l0 ? own2.x : t2.x, l0 ? own2.y : t2.y};
            saved = x[r2];
            x[r] = fused_bin(ab1, x[r], c1);
            x[r2] = fused_bin(ab2, x[r2], c2);
        }
    }

    // a real frame of 8192 per wave: forward, split-filter-merge, inverse (what a block of the FIR equalizer is)
    __global__ __launch_bounds__(64 * WAVES, 2)
    void probe_conv(float2 *data, const float4 *__restrict__ ab, const float2 *__restrict__ tw, int reps)
    {
        __shared__ float areas[WAVES][R * PITCH];
        __shared__ float2 pl[8 * R];
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        pl[tid] = tw[(8 * (tid & 63) * (tid >> 6)) & (N - 1)];
        v2f Q[8];
        #pragma unroll
        for (int b = 0; b < 8; ++b)
            Q[b] = mi_fft::ld2(tw + lane * b);
        __syncthreads();
        const __amdgpu_buffer_rsrc_t tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(ab), 0, int(N * sizeof(float4)), mi::BUFFER_DWORD3);
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
    std::vector<float2> tw(N);
    for (int m = 0; m < N; ++m)
        tw[m] = make_float2(float(cos(-2.0 * PI * m / N)), float(sin(-2.0 * PI * m / N)));
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
