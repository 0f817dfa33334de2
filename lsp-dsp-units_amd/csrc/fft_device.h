// In-LDS Stockham FFT for one workgroup (device code shared by the convolver, equalizer and spectral kernels).
//
// The transform is the autosort radix-4 decimation-in-frequency Stockham scheme with a final radix-2
// pass when log2(N) is odd.  A workgroup of T threads owns one N-point complex sequence held in a single
// LDS buffer: in every pass each thread pulls its butterflies' inputs into registers, the workgroup
// synchronises, and the outputs are written back permuted (read index j + k*N/4 is conflict free; the write
// index q + s*(4p + k) is the autosort permutation).  No bit reversal, no second buffer.
//
// Conventions match the reference's dsp::packed_direct_fft / packed_reverse_fft: forward
// is unnormalised with e^{-jwn}; the inverse here is ALSO unnormalised -- callers fold the 1/N into the
// pass that follows it (window, overlap-add, ...), which is where the reference's 1/N ends up as well.
//
// Twiddles come from a table tw[j] = exp(-2*pi*i*j / TWN), TWN >= N a power of two (built on the host in
// double precision); W_N^k = tw[k * (TWN / N)].
#pragma once

#include <hip/hip_runtime.h>

namespace mi_fft
{
    __device__ __forceinline__ float2 cmul(float2 a, float2 b)
    {
        return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
    }

    __device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
    __device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
    __device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

    // Number of threads that cooperate on one N = 2^LOGN point transform.
    template <int LOGN>
    struct plan
    {
        static constexpr int N   = 1 << LOGN;
        static constexpr int T   = (N / 4 > 64) ? ((N / 4 > 1024) ? 1024 : N / 4) : 64;   // one butterfly per thread
        static constexpr int BPT = (N / 4 + T - 1) / T;         // radix-4 butterflies per thread and pass
    };

    // buf: N complex points in LDS.  All T threads of the workgroup must call this (it synchronises).
    // On entry the caller must have synchronised after filling buf; on exit buf is complete and synchronised.
    template <int LOGN, bool INVERSE>
    __device__ void fft_lds(float2 *buf, const float2 *__restrict__ tw, int tw_stride /* TWN / N */, int tid)
    {
        using P = plan<LOGN>;
        constexpr int N = P::N, T = P::T, BPT = P::BPT, Q = N / 4;

        int s = 1;
        #pragma unroll
        for (int pass = 0; pass < LOGN / 2; ++pass, s <<= 2)
        {
            float2 v[BPT][4];
            #pragma unroll
            for (int b = 0; b < BPT; ++b)
            {
                const int j = tid + b * T;
                if (j < Q)
                {
                    v[b][0] = buf[j];
                    v[b][1] = buf[j + Q];
                    v[b][2] = buf[j + 2 * Q];
                    v[b][3] = buf[j + 3 * Q];
                }
            }
            __syncthreads();
            #pragma unroll
            for (int b = 0; b < BPT; ++b)
            {
                const int j = tid + b * T;
                if (j < Q)
                {
                    const int p = j / s, q = j - p * s;          // s is a power of four: shifts
                    const float2 a = v[b][0], bb = v[b][1], c = v[b][2], d = v[b][3];
                    const float2 apc = cadd(a, c), amc = csub(a, c), bpd = cadd(bb, d), bmd = csub(bb, d);
                    // forward: -i*(b-d) ; inverse: +i*(b-d)
                    const float2 jb = INVERSE ? make_float2(-bmd.y, bmd.x) : make_float2(bmd.y, -bmd.x);
                    const int o = q + 4 * s * p;
                    const int ti = p * s * tw_stride;
                    // one table read per butterfly; w^2 and w^3 by multiplication (2 roundings, ~1e-7)
                    float2 w1 = tw[ti];
                    if (INVERSE)
                        w1 = cconj(w1);
                    const float2 w2 = cmul(w1, w1), w3 = cmul(w2, w1);
                    buf[o]         = cadd(apc, bpd);
                    buf[o + s]     = cmul(w1, cadd(amc, jb));
                    buf[o + 2 * s] = cmul(w2, csub(apc, bpd));
                    buf[o + 3 * s] = cmul(w3, csub(amc, jb));
                }
            }
            __syncthreads();
        }
        if (LOGN & 1)           // one radix-2 pass left: pairs (j, j + N/2), stride N/2, no twiddle
        {
            constexpr int H = N / 2;
            constexpr int PPT = (H + T - 1) / T;
            float2 a[PPT], b2[PPT];
            #pragma unroll
            for (int b = 0; b < PPT; ++b)
            {
                const int j = tid + b * T;
                if (j < H)
                {
                    a[b]  = buf[j];
                    b2[b] = buf[j + H];
                }
            }
            __syncthreads();
            #pragma unroll
            for (int b = 0; b < PPT; ++b)
            {
                const int j = tid + b * T;
                if (j < H)
                {
                    buf[j]     = cadd(a[b], b2[b]);
                    buf[j + H] = csub(a[b], b2[b]);
                }
            }
            __syncthreads();
        }
    }

    // ---- real sequences of length 2M through an M-point complex transform -------------------------------
    //
    // Packed half spectrum ("image") of a real 2M-point sequence: M complex values,
    //   img[0] = (X[0], X[M])  (both real),  img[k] = X[k], 0 < k < M.
    // rtw[k] = exp(-i*pi*k/M) = tw2m[k] with tw2m the 2M-point table.

    // Z (the M-point transform of z[n] = x[2n] + i x[2n+1], in buf) -> image, in place.
    template <int LOGM>
    __device__ void real_split(float2 *buf, const float2 *__restrict__ tw2m, int tw_stride /* TWN / 2M */, int tid)
    {
        using P = plan<LOGM>;
        constexpr int M = P::N, T = P::T;
        // pairs (k, M-k), k = 1 .. M/2 - 1; k = 0 and k = M/2 are their own partners
        for (int k = tid; k <= M / 2; k += T)
        {
            if (k == 0)
            {
                const float2 z0 = buf[0];
                buf[0] = make_float2(z0.x + z0.y, z0.x - z0.y);
            }
            else if (k == M / 2)
                buf[k] = cconj(buf[k]);
            else
            {
                const float2 zk = buf[k], zm = buf[M - k];
                const float2 w  = tw2m[k * tw_stride];                    // e^{-i pi k / M}
                const float2 e  = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));   // (Zk + conj Zm)/2
                const float2 o  = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));   // (Zk - conj Zm)/2
                // X[k]   = e - i w o ;  X[M-k] = conj(e) - i (-conj w) conj(o) = conj(e + i w o) ... written out:
                const float2 wo = cmul(w, o);
                buf[k]     = make_float2(e.x + wo.y, e.y - wo.x);         // e - i*wo
                buf[M - k] = make_float2(e.x - wo.y, -(e.y + wo.x));      // conj(e + i*wo)
            }
        }
        __syncthreads();
    }

    // image (in buf) -> Z such that the UNNORMALISED inverse M-point transform of Z gives
    // z[n] = M * (x[2n] + i x[2n+1]) * ... : with the factors used here  ifft_unnorm(Z)[n] = 2M * (x[2n] + i x[2n+1])
    // where x is the real 2M-point sequence whose UNNORMALISED forward transform is the image.
    template <int LOGM>
    __device__ void real_merge(float2 *buf, const float2 *__restrict__ tw2m, int tw_stride, int tid)
    {
        using P = plan<LOGM>;
        constexpr int M = P::N, T = P::T;
        for (int k = tid; k <= M / 2; k += T)
        {
            if (k == 0)
            {
                const float2 y0 = buf[0];                                 // (X0, XM)
                buf[0] = make_float2(y0.x + y0.y, y0.x - y0.y);
            }
            else if (k == M / 2)
            {
                const float2 y = buf[k];
                buf[k] = make_float2(2.0f * y.x, -2.0f * y.y);
            }
            else
            {
                const float2 xk = buf[k], xm = buf[M - k];
                const float2 w  = cconj(tw2m[k * tw_stride]);             // e^{+i pi k / M}
                const float2 e  = make_float2(xk.x + xm.x, xk.y - xm.y);  // Xk + conj Xm
                const float2 o  = make_float2(xk.x - xm.x, xk.y + xm.y);  // Xk - conj Xm
                const float2 wo = cmul(w, o);
                buf[k]     = make_float2(e.x - wo.y, e.y + wo.x);         // e + i*wo
                buf[M - k] = make_float2(e.x + wo.y, -(e.y - wo.x));      // conj(e - i*wo)
            }
        }
        __syncthreads();
    }
} // namespace mi_fft
