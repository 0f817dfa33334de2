// A 4096-point complex transform that lives in ONE wavefront (gfx950): 64 points per lane, no workgroup-wide LDS passes, no
// s_barrier.  Round 5: tests/experiments/fft_wave_probe.hip, profiles/r05_experiments/fft_wave_probe.txt -- 10.75 ns per real pair
// of 8192 chip-wide against the 15.12 ns of the radix-8 LDS core (fft_device.h) the FIR equalizer's frames were built on.
//
//   1. lane l holds x[l + 64 j], j < 64, and runs a 64-point transform over j in registers (radix-4, decimation in frequency, fully
//      unrolled: the twiddles of those stages are compile-time constants in SGPR pairs, the digit-reversed order of the results is
//      a renaming of registers);
//   2. Y[l][k2] *= W_4096^(l k2): the lane's own 64 twiddles as products of two sets of eight, W^(l (8 a + b)) = W^(8 l a) W^(l b)
//      (the first set from a 4 KiB table of the workgroup, the second in registers);
//   3. ONE exchange through a wave-private LDS area of 64 x 65 floats -- lane k2 collects Y[l][k2] from every lane l -- real parts,
//      then imaginary parts: what a lane has written is dead, what it reads takes those registers.  A wave's LDS accesses complete
//      in order: no barrier.  The values are read straight into digit-reversed registers;
//   4. a second 64-point transform over l, by decimation in TIME (digit-reversed in, natural out): lane k2 holds X[k2 + 64 k1] in
//      register k1.  Natural order both sides, not one register move.
//
// Budget: 128 VGPRs of data, kernels built on it are compiled for two waves per SIMD (__launch_bounds__(64 W, 2)); 8 waves of a
// workgroup take 8 x 16640 B of exchange areas + the 4 KiB table.
//
// split_filter_merge: the packed spectrum of a real frame of 8192 times a real response's spectrum, packed again -- real_split, the
// product and real_merge of fft_device.h as ONE step per bin:
//     Zy[k] = alpha[k] Z[k] + beta[k] conj(Z[N - k]),   alpha = (S + D Im W) / N,  beta = i D Re W / N,
//     S, D = (H[k] +- conj H[N - k]) / 2,  W = e^{-i pi k / N}      (k = 0: H[0] and H[N], both real; k = N / 2: beta = 0)
// so that the unnormalised inverse transform of Zy is the packed result itself.  (alpha, beta) come from a table per response, 16 B
// per bin in the natural order of the bins, made where the response is parsed (convolver.hip, conv_wave_table_kernel).
#pragma once
#include "fft_device.h"
#include "mi_common.h"

#ifndef MI_FFTW_PAIRS_AHEAD
#define MI_FFTW_PAIRS_AHEAD 4
#endif

namespace mi_fftw
{
    using mi_fft::v2f;
    using mi_fft::padd_i;
    using mi_fft::pmul;

    constexpr int N = 4096, R = 64, PITCH = 65;             // PITCH: floats per row of a wave's exchange area
    constexpr int AREA = R * PITCH;                         // floats of a wave's exchange area
    constexpr double PI = 3.14159265358979323846;

    constexpr int rev4_6(int k)                             // base-4 digit reversal of a 6-bit index
    {
        return ((k & 3) << 4) | (k & 12) | ((k >> 4) & 3);
    }

    // (wx + i wy) b with the constant in a pair of SGPRs: mi_fft::pmul wants its operands in VGPRs, and the 108 constants of the two
    // directions hoisted out of a loop as VGPR pairs spill.  The empty volatile asm pins the s_mov's next to their use.
    __device__ __forceinline__ v2f pmul_c(float wx, float wy, v2f b)
    {
        v2f w{wx, wy}, t, r;
        asm volatile("" : "+s"(w));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "s"(w), "v"(b));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "s"(w), "v"(b), "v"(t));
        return r;
    }

    // 64-point transform over the registers, in place; natural order in, out[k] = x[rev4_6(k)].  INV: e^{+i}.
    template <bool INV>
    __device__ __forceinline__ void fft64_dif(v2f (&x)[R])
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
    // transform wants: which register a value is read into is free).  area: AREA floats of this wave.
    __device__ __forceinline__ void exchange(v2f (&x)[R], float *area, int lane)
    {
        asm volatile("" : "+v"(lane));                      // (the sixteen bases the reads want are computed here, not kept across
                                                            //  a loop in registers such a kernel does not have)
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

    // the twiddles between the two transforms: pl[a][l] = W_4096^(8 l a), a < 8 (512 float2 of the workgroup, filled by
    // fill_table), Q[b] = W_4096^(l b), b < 8 (load_lane_twiddles); tw: the device table exp(-2 pi i j / TWN)
    __device__ __forceinline__ void fill_table(float2 *pl, const float2 *__restrict__ tw, int tid /* < 512: every entry once */)
    {
        pl[tid] = tw[((8 * (tid & 63) * (tid >> 6)) & (N - 1)) * (mi_fft::TWN / N)];
    }
    __device__ __forceinline__ void load_lane_twiddles(v2f (&Q)[8], const float2 *__restrict__ tw, int lane)
    {
        #pragma unroll
        for (int b = 0; b < 8; ++b)
            Q[b] = mi_fft::ld2(tw + lane * b * (mi_fft::TWN / N));
    }

    // both sets in the workgroup's table (16 rows, 8 KiB: rows 0 .. 7 = P, rows 8 .. 15 = Q): fft4096_t reads the lane's Q at the
    // twiddle step instead of holding sixteen registers through the transforms (kernels with other things to keep: stft_wave_*)
    __device__ __forceinline__ void fill_table_pq(float2 *plq /* 16 R */, const float2 *__restrict__ tw, int tid, int threads)
    {
        for (int i = tid; i < 16 * R; i += threads)
        {
            const int l = i & 63, row = i >> 6;
            const int m = (row < 8) ? ((8 * l * row) & (N - 1)) : l * (row - 8);
            plq[i] = tw[m * (mi_fft::TWN / N)];
        }
    }

    struct no_hook { __device__ __forceinline__ void operator()() const { } };

    // a whole transform of the wave, natural order both sides: x[j] = in[lane + 64 j]  ->  x[k1] = out[lane + 64 k1], unnormalised.
    // mid(): called between the exchange and the second transform (loads the step after this one wants: in flight over its second half)
    template <bool INV, class HOOK = no_hook>
    __device__ __forceinline__ void fft4096(v2f (&x)[R], const float2 *pl, const v2f (&Q)[8], float *area, int lane, HOOK mid = HOOK())
    {
        fft64_dif<INV>(x);
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
        mid();
        fft64_dit<INV>(x);
    }

    template <bool INV, class HOOK = no_hook>
    __device__ __forceinline__ void fft4096_t(v2f (&x)[R], const float2 *plq, float *area, int lane, HOOK mid = HOOK())
    {
        fft64_dif<INV>(x);
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
        exchange(x, area, lane);
        mid();
        fft64_dit<INV>(x);
    }

    // ---- split, product and merge in one step -------------------------------------------------------------------------------
    // alpha x + beta conj(c) in four packed instructions; ab = (alpha.re, alpha.im, beta.re, beta.im)
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

    // (__float_as_int, not __builtin_bit_cast of the vector's element: that form lost half of the permutes to the optimiser)
    __device__ __forceinline__ v2f from_partner(int addr, v2f v)
    {
        return v2f{__int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v.x))),
                   __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v.y)))};
    }

    constexpr int AHEAD = 4;                                // iterations of split_filter_merge whose table rows are in flight
    // bins lane + 64 r of the table for this lane: ONE lane offset in a VGPR, the row in the scalar offset (64-bit row pointers spill)
    __device__ __forceinline__ float4 table_row(__amdgpu_buffer_rsrc_t tab, int lane16, int r)
    {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 d = __builtin_amdgcn_raw_buffer_load_b128(tab, lane16, r * R * int(sizeof(float4)), 0);
        return make_float4(__uint_as_float(d.x), __uint_as_float(d.y), __uint_as_float(d.z), __uint_as_float(d.w));
    }
    __device__ __forceinline__ __amdgpu_buffer_rsrc_t table_of(const float4 *tab /* N entries */)
    {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(tab), 0, int(N * sizeof(float4)), mi::BUFFER_DWORD3);
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
    // x: bin k = lane + 64 r in register r.  The partner N - k sits in lane 64 - lane, register 63 - r; lane 0 pairs with itself,
    // register (64 - r) & 63 -- the one select per value.  Registers r and 63 - r are updated together: lane 0's partner of
    // register r (r >= 1) is the OLD value of register 64 - r, kept from the iteration before.
    __device__ __forceinline__ void split_filter_merge(v2f (&x)[R], float4 (&q)[2 * AHEAD] /* from table_ahead */,
                                                       __amdgpu_buffer_rsrc_t tab, int lane)
    {
        const int paddr = ((64 - lane) & 63) * 4;
        const bool l0 = lane == 0;
        v2f saved = x[0];
        // the partners' values are asked for PAIRS_AHEAD iterations early (registers r + 1 .. and .. 62 - r are still the old
        // ones): with two waves on a SIMD the round trip of a ds_bpermute per iteration was what this step waited for
        constexpr int PAIRS_AHEAD = MI_FFTW_PAIRS_AHEAD;
        v2f n1[PAIRS_AHEAD], n2[PAIRS_AHEAD];
        #pragma unroll
        for (int r = 0; r < PAIRS_AHEAD; ++r)
        {
            n1[r] = from_partner(paddr, x[63 - r]);
            n2[r] = from_partner(paddr, x[r]);
        }
        #pragma unroll
        for (int r = 0; r < R / 2; ++r)
        {
            const int r2 = 63 - r, s = r % AHEAD, sp = r % PAIRS_AHEAD;
            const float4 ab1 = q[2 * s], ab2 = q[2 * s + 1];
            if (r + AHEAD < R / 2)
            {
                q[2 * s] = table_row(tab, lane * 16, r + AHEAD);
                q[2 * s + 1] = table_row(tab, lane * 16, r2 - AHEAD);
            }
            const v2f t1 = n1[sp], t2 = n2[sp];
            if (r + PAIRS_AHEAD < R / 2)
            {
                n1[sp] = from_partner(paddr, x[r2 - PAIRS_AHEAD]);
                n2[sp] = from_partner(paddr, x[r + PAIRS_AHEAD]);
            }
            const v2f own1 = (r == 0) ? x[0] : saved, own2 = x[r + 1];
            const v2f c1 = v2f{l0 ? own1.x : t1.x, l0 ? own1.y : t1.y}, c2 = v2f{l0 ? own2.x : t2.x, l0 ? own2.y : t2.y};
            saved = x[r2];
            x[r] = fused_bin(ab1, x[r], c1);
            x[r2] = fused_bin(ab2, x[r2], c2);
        }
    }
} // namespace mi_fftw
