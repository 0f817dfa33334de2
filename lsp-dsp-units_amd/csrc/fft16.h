// In-LDS FFT core, second generation: sixteen points per thread, radix-16 passes, affine LDS addressing.
//
// fft_device.h gives every thread ONE radix-8 butterfly per pass (N/8 threads, four passes at N = 2048 or 4096) and
// bank-swizzles the LDS image with an XOR, so every LDS access pays address arithmetic and the transform pair of a
// 4096-point frame costs about 880 instructions per thread, three quarters of them not arithmetic (round-2 review).
// Here a thread owns SIXTEEN points of the sequence for the whole transform:
//
//   * N = 2^LOGN points, T = N / 16 threads; thread t holds x[t + u T], u = 0 .. 15 ("strided layout") on entry AND on
//     exit of a transform -- Stockham's autosort puts the result back in natural order, so a transform is registers ->
//     registers and only the exchanges BETWEEN the passes go through LDS;
//   * passes: radix 16 as often as it fits (LOGN / 4 times), then one pass of radix 2^(LOGN mod 4) -- 16 x 16 x 8 for 2048
//     points, 16 x 16 x 16 for 4096: three passes and two exchanges instead of four and four.  In the strided layout the
//     inputs of a thread's butterfly of EVERY pass are exactly the sixteen values it holds (butterfly b of a radix-R pass
//     takes slots b + k 16/R), so no pass moves data before it computes;
//   * the exchange image is padded by two complex values per 32 (pitch 34): every address of every pass is
//     base(t) + constant -- one VGPR of address per phase, everything else in the instructions' offset fields -- the
//     strided reads are conflict free (32 consecutive lanes = 256 consecutive bytes) and the first pass writes its
//     sixteen consecutive outputs as eight ds_write_b128;
//   * twiddles W^(s p m), m = 1 .. 15, come as eight 16-byte loads per twiddled pass from a table built on the host in
//     double precision (appended to the device twiddle table, see table16_* below) -- no powers derived in the kernel.
//
// Arithmetic per 2048-point transform: 2 x (81 + 30) + 2 x 29 = 280 packed instructions per thread x 128 threads = 35.8 k
// lane-instructions against 56 k before; LDS instructions per transform 2 x (8..16 + 16) per thread.
//
// Conventions are fft_device.h's: forward unnormalised with e^{-jwn}, inverse unnormalised too.
// Everything that is not a barrier is written as __host__ __device__ "phase" functions of one thread, so that
// tests/cpp/fft16_host.cpp can run the same index arithmetic for all T threads of a workgroup in lock step on the CPU.
#pragma once

#include "fft_device.h"

namespace mi_fft16
{
    using mi_fft::v2f;
    using mi_fft::pmul;
    using mi_fft::padd_i;
    using mi_fft::dft4;
    using mi_fft::dft8;

#if defined(__HIPCC__)
    #define MI_HD __host__ __device__ __forceinline__
#else
    #define MI_HD inline
#endif

    constexpr int MIN_LOG = 10, MAX_LOG = 13;               // 1024 .. 8192 complex points (real frames of 2048 .. 16384)

    template <int LOGN>
    struct plan16
    {
        static_assert(LOGN >= MIN_LOG && LOGN <= MAX_LOG, "fft16 covers 1024 .. 8192 points; fft_device.h has the rest");
        static constexpr int N    = 1 << LOGN;
        static constexpr int T    = N / 16;                                 // threads of the workgroup
        static constexpr int N16  = LOGN / 4;                               // radix-16 passes
        static constexpr int RL   = 1 << (LOGN % 4);                        // radix of the pass after them (1: none)
        static constexpr int NP   = N16 + ((RL > 1) ? 1 : 0);               // passes
        static constexpr int NTW  = NP - 1;                                 // passes that apply twiddles (all but the last)
        static constexpr int PITCH = T + T / 16;                            // complex cells between slots u and u + 1 of the image
        static constexpr int LDS  = N + N / 16;                             // complex cells of the padded exchange image
        static_assert(15 * PITCH * 8 < 65536, "slot offsets must fit the 16-bit offset field of ds_read / ds_write");
    };

    // ---- twiddle table of the twiddled passes ---------------------------------------------------------------------------
    // Pass i (radix 16, stride s = 16^i) multiplies output m of the butterfly of thread t by W_N^(e m), e = t & ~(s - 1)
    // = s p.  Table of pass i: float4 [8][T >> 4i], entry [m2][p] = (W^(e 2 m2), W^(e (2 m2 + 1))).
    // All sizes live behind the TWN entries of the device twiddle table (mi::fft_twiddles): float4 index
    // table16_offset(LOGN) from there.
    constexpr int table16_pass_count(int logn, int i) { return ((1 << logn) / 16) >> (4 * i); }
    constexpr int table16_passes(int logn)            { return logn / 4 + ((logn % 4) ? 1 : 0) - 1; }
    constexpr int table16_size(int logn)                                    // float4 entries of one size
    {
        int n = 0;
        for (int i = 0; i < table16_passes(logn); ++i)
            n += 8 * table16_pass_count(logn, i);
        return n;
    }
    constexpr int table16_offset(int logn)                                  // float4 entries in front of this size's table
    {
        int n = 0;
        for (int l = MIN_LOG; l < logn; ++l)
            n += table16_size(l);
        return n;
    }
    constexpr int table16_total() { return table16_offset(MAX_LOG + 1); }

    // host: fill `dst` (table16_total() float4 = 4 floats each) -- double precision source
    inline void table16_build(float *dst)
    {
        for (int logn = MIN_LOG; logn <= MAX_LOG; ++logn)
        {
            const int n = 1 << logn;
            float *base = dst + 4 * size_t(table16_offset(logn));
            for (int i = 0; i < table16_passes(logn); ++i)
            {
                const int cnt = table16_pass_count(logn, i), s = 1 << (4 * i);
                for (int m2 = 0; m2 < 8; ++m2)
                    for (int p = 0; p < cnt; ++p)
                    {
                        const int e = s * p;
                        for (int h = 0; h < 2; ++h)
                        {
                            const long long k = (long long)(e) * (2 * m2 + h) % n;
                            const double a = -2.0 * 3.14159265358979323846 * double(k) / double(n);
                            base[4 * (m2 * cnt + p) + 2 * h]     = float(__builtin_cos(a));
                            base[4 * (m2 * cnt + p) + 2 * h + 1] = float(__builtin_sin(a));
                        }
                    }
                base += 4 * size_t(8 * cnt);
            }
        }
    }

    template <int LOGN>
    struct tw16
    {
        float4 w[(plan16<LOGN>::NTW > 0) ? plan16<LOGN>::NTW : 1][8];
    };

    // `table`: this size's table (device twiddle table + TWN float2 + table16_offset(LOGN) float4)
    template <int LOGN>
    MI_HD void load_tw16(tw16<LOGN> &r, const float4 *__restrict__ table, int t)
    {
        using P = plan16<LOGN>;
        int off = 0;
        #pragma unroll
        for (int i = 0; i < P::NTW; ++i)
        {
            const int cnt = P::T >> (4 * i), idx = t >> (4 * i);
            #pragma unroll
            for (int m2 = 0; m2 < 8; ++m2)
                r.w[i][m2] = table[off + m2 * cnt + idx];
            off += 8 * cnt;
        }
    }

    // ---- butterflies ----------------------------------------------------------------------------------------------------
    // 16-point DFT in registers, natural order in and out, as 4 x 4: k = k0 + 4 k1, m = m0 + 4 m1,
    //   W16^(mk) = W4^(m0 k1) W16^(m0 k0) W4^(m1 k0):  DFT4 over k1, twiddle, DFT4 over k0.
    template <bool INV>
    MI_HD void dft16(v2f (&x)[16])
    {
        constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f, H = 0.70710678118654752440f;
        #pragma unroll
        for (int k0 = 0; k0 < 4; ++k0)
            dft4<INV>(x[k0], x[k0 + 4], x[k0 + 8], x[k0 + 12]);            // x[k0 + 4 m0] = u_k0[m0]
        // x[k0 + 4 m0] *= W16^(m0 k0)   (forward: e^{-2 pi i e / 16}; INV: the conjugate)
        const v2f w1 = v2f{C1, -S1}, w3 = v2f{S1, -C1};
        const v2f hh = v2f{H, H}, nh = v2f{-H, -H};
        auto rot8  = [&](v2f a) -> v2f { return padd_i<INV>(a, a) * hh; };            // W16^2: a (1 -+ i) / sqrt2
        auto rot24 = [&](v2f a) -> v2f { return padd_i<!INV>(a, a) * nh; };           // W16^6: -a (1 +- i) / sqrt2
        auto rot4  = [&](v2f a) -> v2f { return INV ? v2f{-a.y, a.x} : v2f{a.y, -a.x}; };   // W16^4: a (-+ i)
        x[1 + 4]  = pmul<INV>(w1, x[1 + 4]);                               // e = 1
        x[2 + 4]  = rot8(x[2 + 4]);                                        // e = 2
        x[3 + 4]  = pmul<INV>(w3, x[3 + 4]);                               // e = 3
        x[1 + 8]  = rot8(x[1 + 8]);                                        // e = 2
        x[2 + 8]  = rot4(x[2 + 8]);                                        // e = 4
        x[3 + 8]  = rot24(x[3 + 8]);                                       // e = 6
        x[1 + 12] = pmul<INV>(w3, x[1 + 12]);                              // e = 3
        x[2 + 12] = rot24(x[2 + 12]);                                      // e = 6
        x[3 + 12] = -pmul<INV>(w1, x[3 + 12]);                             // e = 9: W16^9 = -W16^1
        v2f y[16];
        #pragma unroll
        for (int m0 = 0; m0 < 4; ++m0)
        {
            dft4<INV>(x[4 * m0], x[4 * m0 + 1], x[4 * m0 + 2], x[4 * m0 + 3]);     // x[4 m0 + m1] = y[m0 + 4 m1]
            #pragma unroll
            for (int m1 = 0; m1 < 4; ++m1)
                y[m0 + 4 * m1] = x[4 * m0 + m1];
        }
        #pragma unroll
        for (int m = 0; m < 16; ++m)
            x[m] = y[m];
    }

    template <bool INV>
    MI_HD void dft2(v2f &a, v2f &b) { const v2f s = a + b, d = a - b; a = s; b = d; }

    // ---- passes (one thread; the values stay in the strided layout) -------------------------------------------------------
    // pass I < NTW: radix 16 + twiddles.  Output m of the butterfly is left in slot m; it belongs at image position
    // q + s (16 p + m) (exchange_store below).
    template <int LOGN, bool INV, int I>
    MI_HD void pass_twiddled(v2f (&x)[16], const tw16<LOGN> &tw)
    {
        dft16<INV>(x);
        #pragma unroll
        for (int m2 = 0; m2 < 8; ++m2)
        {
            const float4 w = tw.w[I][m2];
            if (m2 > 0)
                x[2 * m2] = pmul<INV>(v2f{w.x, w.y}, x[2 * m2]);
            x[2 * m2 + 1] = pmul<INV>(v2f{w.z, w.w}, x[2 * m2 + 1]);
        }
    }

    // last pass: radix R = 16 / B, B butterflies; butterfly b takes slots b + k B and leaves output m in slot b + m B:
    // natural order in the strided layout, no exchange behind it
    template <int LOGN, bool INV>
    MI_HD void pass_last(v2f (&x)[16])
    {
        constexpr int R = (plan16<LOGN>::RL > 1) ? plan16<LOGN>::RL : 16, B = 16 / R;
        if (R == 16)
            dft16<INV>(x);
        else if (R == 8)
        {
            #pragma unroll
            for (int b = 0; b < B; ++b)
            {
                v2f v[8];
                #pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = x[b + k * B];
                dft8<INV>(v);
                #pragma unroll
                for (int k = 0; k < 8; ++k) x[b + k * B] = v[k];
            }
        }
        else if (R == 4)
        {
            #pragma unroll
            for (int b = 0; b < B; ++b)
                dft4<INV>(x[b], x[b + B], x[b + 2 * B], x[b + 3 * B]);
        }
        else
        {
            #pragma unroll
            for (int b = 0; b < B; ++b)
                dft2<INV>(x[b], x[b + B]);
        }
    }

    // ---- the exchange image -----------------------------------------------------------------------------------------------
    // cell of sequence position pos: pos + 2 (pos >> 5).  All three index forms below are base(t) + constant(slot).
    MI_HD int image_cell(int pos) { return pos + 2 * (pos >> 5); }

    // after pass I: slot m of thread t -> position q + s (16 p + m), s = 16^I, q = t mod s, p = t div s
    template <int LOGN, int I>
    MI_HD void exchange_store(float2 *img, const v2f (&x)[16], int t)
    {
        constexpr int s = 1 << (4 * I);
        if (I == 0)
        {
            // 16 t + m: sixteen consecutive cells, 16-byte aligned pairs
            float4 *p = reinterpret_cast<float4 *>(img + 16 * t + 2 * (t >> 1));
            #pragma unroll
            for (int m2 = 0; m2 < 8; ++m2)
                p[m2] = make_float4(x[2 * m2].x, x[2 * m2].y, x[2 * m2 + 1].x, x[2 * m2 + 1].y);
        }
        else if (I == 1)
        {
            // q + 256 p + 16 m, q < 16: cell = q + 272 p + 16 m + 2 (m >> 1)
            float2 *p = img + (t & 15) + 272 * (t >> 4);
            #pragma unroll
            for (int m = 0; m < 16; ++m)
                p[16 * m + 2 * (m >> 1)] = make_float2(x[m].x, x[m].y);
        }
        else
        {
            // q + 4096 p + 256 m, q < 256: cell = q + 2 (q >> 5) + 4352 p + 272 m
            static_assert(I <= 2, "three twiddled passes at most (8192 points)");
            const int q = t & (s - 1);
            float2 *p = img + q + 2 * (q >> 5) + 4352 * (t >> 8);
            #pragma unroll
            for (int m = 0; m < 16; ++m)
                p[272 * m] = make_float2(x[m].x, x[m].y);
        }
    }

    // strided layout out of the image: slot u of thread t <- position t + u T
    template <int LOGN>
    MI_HD void exchange_load(const float2 *img, v2f (&x)[16], int t)
    {
        constexpr int PITCH = plan16<LOGN>::PITCH;
        const float2 *p = img + t + 2 * (t >> 5);
        #pragma unroll
        for (int u = 0; u < 16; ++u)
        {
            const float2 v = p[u * PITCH];
            x[u] = v2f{v.x, v.y};
        }
    }

    // strided layout <-> an UNPADDED natural-order buffer (the interface of fft_device.h's transforms)
    template <int LOGN>
    MI_HD void natural_load(const float2 *buf, v2f (&x)[16], int t)
    {
        constexpr int T = plan16<LOGN>::T;
        #pragma unroll
        for (int u = 0; u < 16; ++u)
        {
            const float2 v = buf[t + u * T];
            x[u] = v2f{v.x, v.y};
        }
    }
    template <int LOGN>
    MI_HD void natural_store(float2 *buf, const v2f (&x)[16], int t)
    {
        constexpr int T = plan16<LOGN>::T;
        #pragma unroll
        for (int u = 0; u < 16; ++u)
            buf[t + u * T] = make_float2(x[u].x, x[u].y);
    }

#if defined(__HIPCC__)
    // ---- a whole transform, registers -> registers (all T threads of the workgroup; synchronises) ------------------------
    // img: plan16::LDS complex cells of LDS that nobody else touches during the call.  On entry no thread of the
    // workgroup may still be reading or writing img (the caller's last barrier covers that); on exit img is free again.
    template <int LOGN, bool INV>
    __device__ __forceinline__ void fft16_regs(v2f (&x)[16], float2 *img, const tw16<LOGN> &tw, int t)
    {
        using P = plan16<LOGN>;
        if (P::NTW >= 1)
        {
            pass_twiddled<LOGN, INV, 0>(x, tw);
            exchange_store<LOGN, 0>(img, x, t);
            __syncthreads();
            exchange_load<LOGN>(img, x, t);
        }
        if (P::NTW >= 2)
        {
            pass_twiddled<LOGN, INV, (P::NTW >= 2) ? 1 : 0>(x, tw);
            __syncthreads();                                    // everybody has taken the first image
            exchange_store<LOGN, (P::NTW >= 2) ? 1 : 0>(img, x, t);
            __syncthreads();
            exchange_load<LOGN>(img, x, t);
        }
        if (P::NTW >= 3)
        {
            pass_twiddled<LOGN, INV, (P::NTW >= 3) ? 2 : 0>(x, tw);
            __syncthreads();
            exchange_store<LOGN, (P::NTW >= 3) ? 2 : 0>(img, x, t);
            __syncthreads();
            exchange_load<LOGN>(img, x, t);
        }
        pass_last<LOGN, INV>(x);
        if (P::NTW >= 1)
            __syncthreads();                                    // the last image has been taken: img is free
    }

    // ---- drop-in for fft_device.h's real_fft: natural-order LDS buffer in, natural-order LDS buffer out ---------------
    // buf must hold plan16::LDS complex cells (the first N are the sequence / the image, the rest is exchange padding):
    // the strided layout is read out of buf before the first exchange image goes into the same cells.
    template <int LOGM>
    struct real_fft16
    {
        using P = plan16<LOGM>;
        tw16<LOGM>             ft;

        static constexpr int ITER = (P::N / 2 + P::T - 1) / P::T;
        float2 rw[ITER];

        // tw: the device twiddle table (TWN float2 entries, then the radix-16 tables)
        __device__ __forceinline__ void load(const float2 *__restrict__ tw, int twn, int tid)
        {
            constexpr int M = P::N;
            load_tw16<LOGM>(ft, reinterpret_cast<const float4 *>(tw + twn) + table16_offset(LOGM), tid);
            #pragma unroll
            for (int i = 0; i < ITER; ++i)                      // e^{-i pi k / M} of the pairs (k, M - k), k = tid + i T
                rw[i] = tw[((tid + i * P::T) & (M / 2 - 1)) * (twn / (2 * M))];
        }
        __device__ __forceinline__ void prepare() { }

        // Z in buf (natural order) -> image of the real sequence, in place
        __device__ __forceinline__ void split(float2 *buf, int tid) const
        {
            constexpr int M = P::N, T = P::T;
            #pragma unroll
            for (int i = 0; i < ITER; ++i)
            {
                const int k = tid + i * T;
                if (k == 0)
                {
                    const float2 z0 = buf[0];
                    buf[0] = make_float2(z0.x + z0.y, z0.x - z0.y);
                    buf[M / 2] = mi_fft::cconj(buf[M / 2]);
                }
                else
                {
                    const float2 zk = buf[k], zm = buf[M - k];
                    const float2 w  = rw[i];
                    const float2 e  = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                    const float2 o  = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
                    const float2 wo = mi_fft::cmul(w, o);
                    buf[k]     = make_float2(e.x + wo.y, e.y - wo.x);
                    buf[M - k] = make_float2(e.x - wo.y, -(e.y + wo.x));
                }
            }
            __syncthreads();
        }
        __device__ __forceinline__ void merge(float2 *buf, int tid) const
        {
            constexpr int M = P::N, T = P::T;
            #pragma unroll
            for (int i = 0; i < ITER; ++i)
            {
                const int k = tid + i * T;
                if (k == 0)
                {
                    const float2 y0 = buf[0];
                    buf[0] = make_float2(y0.x + y0.y, y0.x - y0.y);
                    const float2 y = buf[M / 2];
                    buf[M / 2] = make_float2(2.0f * y.x, -2.0f * y.y);
                }
                else
                {
                    const float2 xk = buf[k], xm = buf[M - k];
                    const float2 w  = mi_fft::cconj(rw[i]);
                    const float2 e  = make_float2(xk.x + xm.x, xk.y - xm.y);
                    const float2 o  = make_float2(xk.x - xm.x, xk.y + xm.y);
                    const float2 wo = mi_fft::cmul(w, o);
                    buf[k]     = make_float2(e.x - wo.y, e.y + wo.x);
                    buf[M - k] = make_float2(e.x + wo.y, -(e.y - wo.x));
                }
            }
            __syncthreads();
        }

        template <bool INV>
        __device__ __forceinline__ void transform(float2 *buf, int tid) const
        {
            v2f x[16];
            natural_load<LOGM>(buf, x, tid);
            __syncthreads();                                    // everybody holds its points: buf becomes the exchange image
            fft16_regs<LOGM, INV>(x, buf, ft, tid);
            natural_store<LOGM>(buf, x, tid);
            __syncthreads();
        }
        // packed samples z[n] = x[2n] + i x[2n+1] in buf -> image in buf (synchronised on entry by the caller, on exit here)
        __device__ __forceinline__ void forward(float2 *buf, float2 * /*scr: unused*/, int tid) const
        {
            transform<false>(buf, tid);
            split(buf, tid);
        }
        // image in buf -> 2M real samples (times 2M), left in buf as (x[2n], x[2n+1])
        __device__ __forceinline__ void inverse(float2 *buf, float2 * /*scr*/, int tid) const
        {
            merge(buf, tid);
            transform<true>(buf, tid);
        }
    };

    // ---- which core a kernel gets for a 2^LOGM-point complex transform -------------------------------------------------------
    // fsel<LOGM>::T threads, fsel<LOGM>::LDS complex cells of LDS (buf = lds, scr = lds + fsel::SCR for the radix-8 core; the
    // radix-16 core only uses buf, N + N/16 cells of it), fsel<LOGM>::real with the common interface
    //   load(tw, twn, tid); prepare(); forward(buf, scr, tid); inverse(buf, scr, tid)
    // Round 3 measured the radix-16 core as a drop-in (same kernels, half the threads): analyzer 12.1 -> 18.2 us, stft hop
    // 17.5 -> 18.8 us, equalizer FIR step 10.6 -> 12.5 us, splitter 36.5 -> 41.7 us (profiles/r03_experiments/
    // fft_radix16_core.txt).  Its transform is 36 % fewer arithmetic instructions, but a bank of 1024 channels is then 2048
    // waves -- two per SIMD instead of four -- and these kernels are bound by how many waves a SIMD has to pick from, not by
    // the instruction count.  It stays selectable (-DMI_FFT_RADIX16) and tested on the host (tests/cpp/fft16_host.cpp).
#if defined(MI_FFT_RADIX16)
    constexpr bool RADIX16_DEFAULT = true;
#else
    constexpr bool RADIX16_DEFAULT = false;
#endif
    template <int LOGM, bool R16 = (RADIX16_DEFAULT && LOGM >= MIN_LOG && LOGM <= MAX_LOG)>
    struct fsel;
    template <int LOGM>
    struct fsel<LOGM, true>
    {
        static constexpr int N = plan16<LOGM>::N, T = plan16<LOGM>::T, LDS = plan16<LOGM>::LDS, SCR = N;
        static constexpr bool radix16 = true;
        typedef real_fft16<LOGM> real;
    };
    template <int LOGM>
    struct fsel<LOGM, false>
    {
        // buf and scr of the radix-8 core: N cells of sequence plus the padding of its intermediate layouts each
        static constexpr int N = mi_fft::plan<LOGM>::N, T = mi_fft::plan<LOGM>::T, SCR = mi_fft::plan<LOGM>::BUF, LDS = 2 * SCR;
        static constexpr bool radix16 = false;
        typedef mi_fft::real_fft<LOGM> real;
    };
#endif // __HIPCC__
} // namespace mi_fft16
