// In-LDS Stockham FFT for one workgroup (device code shared by the convolver, equalizer and spectral kernels).
//
// The transform is the autosort decimation-in-frequency Stockham scheme with radix-8 passes followed by up to two
// radix-4 passes.  A workgroup of T threads owns one N-point complex sequence held in a single
// pair of LDS buffers: in every pass each thread pulls its butterflies' inputs from one buffer and writes the outputs
// permuted into the other (read index j + k*N/4 is conflict free; the write index q + s*(4p + k) is the autosort
// permutation, bank-swizzled in the intermediate passes).  No bit reversal, one barrier per pass.
//
// Conventions match the reference's dsp::packed_direct_fft / packed_reverse_fft: forward
// is unnormalised with e^{-jwn}; the inverse here is ALSO unnormalised -- callers fold the 1/N into the
// pass that follows it (window, overlap-add, ...), which is where the reference's 1/N ends up as well.
//
// Twiddles come from a table tw[j] = exp(-2*pi*i*j / TWN), TWN >= N a power of two (built on the host in
// double precision); W_N^k = tw[k * (TWN / N)].  A thread fetches the few values it needs once per kernel, before
// the samples, and keeps them in registers for every transform of the kernel.
#pragma once

#include <hip/hip_runtime.h>

namespace mi_fft
{
    // length of the shared twiddle table exp(-2 pi i j / TWN) (mi::fft_twiddles): the largest real transform is 2^14 points
    constexpr int TWN = 16384;

    __device__ __forceinline__ float2 cmul(float2 a, float2 b)
    {
        return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
    }

    __device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
    __device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
    __device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

    // Packed-fp32 forms for the butterflies: a complex value is one 64-bit register pair, add/sub are one
    // v_pk_add_f32, a complex product is v_pk_mul_f32 + v_pk_fma_f32 (half swaps and sign flips fold into the
    // instructions' op_sel / neg modifiers).  The transforms are VALU-issue bound, so instruction count is what matters.
    typedef float v2f __attribute__((ext_vector_type(2)));
    // complex a * b (CONJ_A: conj(a) * b).  op_sel picks the half that feeds the low lane, op_sel_hi the high lane:
    //   t = (a.x b.x, a.x b.y);   r = (t.x -+ a.y b.y, t.y +- a.y b.x)
    // (host build: the same arithmetic in plain C++ -- tests/cpp/fft16_host.cpp runs the index math of fft16.h on the CPU)
    template <bool CONJ_A>
    __host__ __device__ __forceinline__ v2f pmul(v2f a, v2f b)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        v2f t, r;
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
        if (CONJ_A)
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
        else
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
        return r;
#else
        const float ay = CONJ_A ? -a.y : a.y;
        return v2f{a.x * b.x - ay * b.y, a.x * b.y + ay * b.x};
#endif
    }
    // a + i b (PLUS_I) or a - i b:  (a.x -+ b.y, a.y +- b.x) in one v_pk_add_f32
    template <bool PLUS_I>
    __host__ __device__ __forceinline__ v2f padd_i(v2f a, v2f b)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        v2f r;
        if (PLUS_I)
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
        else
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
        return r;
#else
        return PLUS_I ? v2f{a.x - b.y, a.y + b.x} : v2f{a.x + b.y, a.y - b.x};
#endif
    }
    // a + conj(b), a - conj(b), and the two conjugated +-i forms of the real split / merge, one v_pk_add_f32 each
    __host__ __device__ __forceinline__ v2f padd_cj(v2f a, v2f b)           // (a.x + b.x, a.y - b.y)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        v2f r;
        asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
        return r;
#else
        return v2f{a.x + b.x, a.y - b.y};
#endif
    }
    __host__ __device__ __forceinline__ v2f psub_cj(v2f a, v2f b)           // (a.x - b.x, a.y + b.y)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        v2f r;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
        return r;
#else
        return v2f{a.x - b.x, a.y + b.y};
#endif
    }
    // conj(a + i b) (PLUS_I) or conj(a - i b):  (a.x -+ b.y, -(a.y +- b.x))
    template <bool PLUS_I>
    __host__ __device__ __forceinline__ v2f pconj_add_i(v2f a, v2f b)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        v2f r;
        if (PLUS_I)
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
        else
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
        return r;
#else
        return PLUS_I ? v2f{a.x - b.y, -(a.y + b.x)} : v2f{a.x + b.y, -(a.y - b.x)};
#endif
    }
    __device__ __forceinline__ v2f ld2(const float2 *p) { return *reinterpret_cast<const v2f *>(p); }
    __device__ __forceinline__ void st2(float2 *p, v2f v) { *reinterpret_cast<v2f *>(p) = v; }

    // Pass plan of one N = 2^LOGN point transform: radix-8 passes first, then radix-4 passes (3a + 2b = LOGN with b <= 2:
    // every LOGN >= 2 except 3 decomposes; LOGN == 3 is a single radix-8 pass).  One radix-8 butterfly per thread.
    template <int LOGN>
    struct plan
    {
        static constexpr int N   = 1 << LOGN;
        // radix-4 passes.  256 points run as four of them: every lane of the one wave has a butterfly in every pass (8 x 8 x 4
        // leaves half the wave idle in its two radix-8 passes; the convolver's 256-sample block kernel: 172.3 -> 168.6 us
        // per 4096 samples)
        static constexpr int N4  = (LOGN == 8) ? 4 : (LOGN % 3 == 0) ? 0 : (LOGN % 3 == 2) ? 1 : (LOGN >= 4 ? 2 : 0);
        static constexpr int N8  = (LOGN - 2 * N4) / 3;                                               // radix-8 passes
        static constexpr int NP  = N8 + N4;
        static constexpr int TB  = (N8 > 0) ? N / 8 : N / 4;                         // butterflies of the widest pass
        static constexpr int T   = (TB > 64) ? ((TB > 1024) ? 1024 : TB) : 64;
        static constexpr int BPT8 = (N / 8 + T - 1) / T;        // radix-8 butterflies per thread and pass
        static constexpr int BPT4 = (N / 4 + T - 1) / T;        // radix-4 butterflies per thread and pass
        static_assert(LOGN >= 2 && 3 * N8 + 2 * N4 == LOGN, "unsupported transform size");
        // radix of pass i and the stride (product of the radices before it)
        static constexpr int radix(int i)  { return (i < N8) ? 8 : 4; }
        static constexpr int stride(int i) { return (i <= N8) ? (1 << (3 * i)) : (1 << (3 * N8 + 2 * (i - N8))); }
        // Layout of the buffer a pass WRITES (see fft_lds): a pass that is not the last and whose stride is below 64 leaves s
        // cells of padding behind every R s cells; everything else is in natural order.  Cells a buffer needs:
        static constexpr bool padded(int i) { return i < NP - 1 && stride(i) < 64; }
        static constexpr int pad_cells()
        {
            int worst = 0;
            for (int i = 0; i < NP; ++i)
                if (padded(i) && N / radix(i) > worst)
                    worst = N / radix(i);
            return worst;
        }
        static constexpr int BUF = N + pad_cells();             // cells of buf and of scr
    };

    // Bank layout of the intermediate buffers (round 3; the XOR swizzle of rounds 1-2 cost three to four VALU instructions
    // per LDS access -- more address arithmetic than butterfly arithmetic in every pass).  The autosort write index of a
    // radix-R pass with stride s is q + R s p + s m for lane j = q + s p: runs of s consecutive lanes R s cells apart -- for
    // s < 64 a many-way bank conflict.  The buffer such a pass writes gets s cells of padding behind every R s cells:
    //     cell(pos) = pos + s (pos div R s)
    // so that the runs of a 16-lane write group tile the banks (pitch (R + 1) s cells: 18 dwords at R = 8, s = 1; 144 = 16
    // mod 32 at s = 8), and BOTH sides stay base(lane) + constant(slot):
    //     writer:  cell(q + R s p + s m)  = q + (R + 1) s p + s m                     (q + s m < R s)
    //     reader:  cell(j + k Q')         = j + s (j div R s) + k (Q' + Q' / R)       (Q' a multiple of R s)
    // The reader's 32-lane groups stay inside one R s block for s >= 32 and cross at most three pads for s = 1 (a 2-way
    // conflict on six banks of one pass's reads).  Passes with s >= 64 write 64 consecutive lanes: natural order.
    // Twiddles of every pass for the butterflies this thread owns: W^(m p s), m = 1 .. radix-1, fetched (m = 1) once
    // per kernel -- ideally long before the transform: the table lives in global memory -- and reused by the forward and
    // the inverse transform.
    template <int LOGN>
    struct fft_tw
    {
        static constexpr int MAXB = (plan<LOGN>::BPT8 > plan<LOGN>::BPT4) ? plan<LOGN>::BPT8 : plan<LOGN>::BPT4;
        v2f w[plan<LOGN>::NP][MAXB][7];
    };

    // Two steps, so that the table reads can be issued first thing in a kernel and their latency hidden behind the
    // kernel's other loads: load_fft_tw() only requests W^(p s); finish_fft_tw() derives the higher powers.
    template <int LOGN>
    __device__ __forceinline__ void load_fft_tw(fft_tw<LOGN> &r, const float2 *__restrict__ tw, int tw_stride /* TWN / N */, int tid)
    {
        using P = plan<LOGN>;
        constexpr int T = P::T, N = P::N;
        #pragma unroll
        for (int pass = 0; pass < P::NP; ++pass)
        {
            const int R = P::radix(pass), s = P::stride(pass), Q = N / R;
            const int BPT = (R == 8) ? P::BPT8 : P::BPT4;
            #pragma unroll
            for (int b = 0; b < BPT; ++b)
            {
                const int j = (tid + b * T) & (Q - 1);
                if (s < Q)                                          // s == Q: p = 0, the twiddles are 1 and unused
                    r.w[pass][b][0] = ld2(tw + (j & ~(s - 1)) * tw_stride);     // W_N^(p s), p = j / s
            }
        }
    }

    template <int LOGN>
    __device__ __forceinline__ void finish_fft_tw(fft_tw<LOGN> &r)
    {
        using P = plan<LOGN>;
        #pragma unroll
        for (int pass = 0; pass < P::NP; ++pass)
        {
            const int R = P::radix(pass), s = P::stride(pass), Q = P::N / R;
            const int BPT = (R == 8) ? P::BPT8 : P::BPT4;
            #pragma unroll
            for (int b = 0; b < BPT; ++b)
                if (s < Q)
                {
                    // one table value per butterfly; the higher powers by multiplication (a few roundings, ~1e-7)
                    const v2f w1 = r.w[pass][b][0];
                    const v2f w2 = pmul<false>(w1, w1);
                    const v2f w3 = pmul<false>(w2, w1);
                    r.w[pass][b][1] = w2;
                    r.w[pass][b][2] = w3;
                    if (R == 8)
                    {
                        const v2f w4 = pmul<false>(w2, w2);
                        r.w[pass][b][3] = w4;
                        r.w[pass][b][4] = pmul<false>(w4, w1);
                        r.w[pass][b][5] = pmul<false>(w3, w3);
                        r.w[pass][b][6] = pmul<false>(w4, w3);
                    }
                }
        }
    }

    // 4-point DFT in registers, outputs in natural order (forward: e^{-j}, INVERSE: e^{+j})
    template <bool INVERSE>
    __host__ __device__ __forceinline__ void dft4(v2f &x0, v2f &x1, v2f &x2, v2f &x3)
    {
        const v2f apc = x0 + x2, amc = x0 - x2, bpd = x1 + x3, bmd = x1 - x3;
        x0 = apc + bpd;
        x1 = padd_i<INVERSE>(amc, bmd);          // forward: (a-c) - i (b-d)
        x2 = apc - bpd;
        x3 = padd_i<!INVERSE>(amc, bmd);
    }

    // 8-point DFT in registers, outputs in natural order: even outputs = DFT4 of the sums, odd outputs = DFT4 of the
    // differences rotated by W8^k
    template <bool INVERSE>
    __host__ __device__ __forceinline__ void dft8(v2f (&x)[8])
    {
        constexpr float H = 0.70710678118654752f;
        v2f a0 = x[0] + x[4], a1 = x[1] + x[5], a2 = x[2] + x[6], a3 = x[3] + x[7];
        v2f b0 = x[0] - x[4], b1 = x[1] - x[5], b2 = x[2] - x[6], b3 = x[3] - x[7];
        // W8^1 = (1 -+ i)/sqrt2, W8^2 = -+i, W8^3 = (-1 -+ i)/sqrt2   (upper sign: forward)
        b1 = padd_i<INVERSE>(b1, b1) * v2f{H, H};                // b1 (1 -+ i) / sqrt2
        const v2f t3 = padd_i<!INVERSE>(b3, b3) * v2f{H, H};     // b3 (1 +- i) / sqrt2, negated below
        b3 = -t3;
        b2 = INVERSE ? v2f{-b2.y, b2.x} : v2f{b2.y, -b2.x};      // b2 * (-+ i)
        dft4<INVERSE>(a0, a1, a2, a3);
        dft4<INVERSE>(b0, b1, b2, b3);
        x[0] = a0; x[2] = a1; x[4] = a2; x[6] = a3;
        x[1] = b0; x[3] = b1; x[5] = b2; x[7] = b3;
    }

    // buf: N complex points in LDS in natural order, scr: N more.  All T threads of the workgroup must call this (it
    // synchronises).  On entry the caller must have synchronised after filling buf; on exit the transform is in buf,
    // natural order, synchronised.  One barrier per pass: the passes ping-pong between the two buffers (an odd number of
    // passes starts with one in-place pass).
    // REG_IN:  the first pass takes its inputs from `io` instead of buf: io[i] = x[tid + i T], i < N / T -- the natural
    //          distribution of a sequence over the workgroup, which is exactly what thread tid's butterflies of the first
    //          pass read (j + k Q with Q a multiple of T), so a kernel that has just loaded (and windowed) its frame hands it
    //          over in registers and saves the LDS write, the barrier and the first pass's LDS reads;
    // REG_OUT: the last pass leaves its outputs in `io` (same distribution: output m of butterfly j is X[j + m Q]) instead
    //          of writing them to buf -- for a consumer that works element-wise (window, overlap-add, store).
    // On entry with REG_IN nobody may still be using buf or scr (a barrier of the caller covers that); on exit with REG_OUT
    // buf and scr are free again after the caller's next barrier.
    template <int LOGN, bool INVERSE, bool REG_IN = false, bool REG_OUT = false>
    __device__ void fft_lds(float2 *buf, float2 *scr, const fft_tw<LOGN> &tws, int tid, v2f *io = nullptr)
    {
        using P = plan<LOGN>;
        constexpr int N = P::N, T = P::T, NP = P::NP;
        static_assert(!(REG_IN || REG_OUT) || (T == P::TB), "register hand-over needs a whole butterfly of the widest pass per thread (512 .. 8192 points)");

        float2 *src = buf;
        float2 *dst = (NP & 1) ? buf : scr;
        #pragma unroll
        for (int pass = 0; pass < NP; ++pass)
        {
            const bool first = (pass == 0), last = (pass == NP - 1);
            const int R = P::radix(pass), s = P::stride(pass), Q = N / R;
            // layout of the buffer this pass reads (written by the pass before) and of the one it writes
            const bool rd_pad = !first && P::padded(pass - 1), wr_pad = P::padded(pass);
            const int ps = first ? 1 : P::stride(pass - 1), pR = first ? 8 : P::radix(pass - 1);
            const int rd_step = rd_pad ? Q + Q / pR : Q;
            auto rd_base = [&](int j) -> int { return rd_pad ? j + ps * (j / (pR * ps)) : j; };
            if (R == 8)
            {
                constexpr int BPT = P::BPT8;
                v2f v[BPT][8];
                #pragma unroll
                for (int b = 0; b < BPT; ++b)
                {
                    const int j = tid + b * T;
                    if (j < Q)
                    {
                        const int rb = rd_base(j);                       // cell(j + k Q) = rb + k rd_step
                        #pragma unroll
                        for (int k = 0; k < 8; ++k)
                            v[b][k] = (first && REG_IN) ? io[b + k * (Q / T)]          // x[j + k Q] = x[tid + (b + k Q/T) T]
                                                        : ld2(src + rb + k * rd_step);
                    }
                }
                if (dst == src && !(first && REG_IN))
                    __syncthreads();
                #pragma unroll
                for (int b = 0; b < BPT; ++b)
                {
                    const int j = tid + b * T;
                    if (j < Q)
                    {
                        const int q = j & (s - 1);
                        const int o = q + 8 * (j - q);                   // q + R s p
                        dft8<INVERSE>(v[b]);
                        #pragma unroll
                        for (int m = 0; m < 8; ++m)
                        {
                            v2f r = v[b][m];
                            if (m > 0 && s < Q)                          // s == Q: p = 0, all twiddles are 1
                                r = pmul<INVERSE>(tws.w[pass][b][m - 1], r);
                            if (last && REG_OUT)
                                io[b + m * (Q / T)] = r;                 // X[j + m Q] (last pass: s = Q, o = j)
                            else
                                st2(dst + (wr_pad ? q + 9 * (j - q) : o) + m * s, r);   // padded: q + (R + 1) s p + s m
                        }
                    }
                }
            }
            else
            {
                constexpr int BPT = P::BPT4;
                v2f v[BPT][4];
                #pragma unroll
                for (int b = 0; b < BPT; ++b)
                {
                    const int j = tid + b * T;
                    if (j < Q)
                    {
                        const int rb = rd_base(j);
                        #pragma unroll
                        for (int k = 0; k < 4; ++k)
                            v[b][k] = (first && REG_IN) ? io[b + k * (Q / T)]
                                                        : ld2(src + rb + k * rd_step);
                    }
                }
                if (dst == src && !(first && REG_IN))
                    __syncthreads();
                #pragma unroll
                for (int b = 0; b < BPT; ++b)
                {
                    const int j = tid + b * T;
                    if (j < Q)
                    {
                        const int q = j & (s - 1);
                        const int o = q + 4 * (j - q);
                        dft4<INVERSE>(v[b][0], v[b][1], v[b][2], v[b][3]);
                        #pragma unroll
                        for (int m = 0; m < 4; ++m)
                        {
                            v2f r = v[b][m];
                            if (m > 0 && s < Q)
                                r = pmul<INVERSE>(tws.w[pass][b][m - 1], r);
                            if (last && REG_OUT)
                                io[b + m * (Q / T)] = r;
                            else
                                st2(dst + (wr_pad ? q + 5 * (j - q) : o) + m * s, r);
                        }
                    }
                }
            }
            if (!(last && REG_OUT))
                __syncthreads();
            src = dst;
            dst = (src == buf) ? scr : buf;
        }
    }

    // ---- real sequences of length 2M through an M-point complex transform -------------------------------
    //
    // Packed half spectrum ("image") of a real 2M-point sequence: M complex values,
    //   img[0] = (X[0], X[M])  (both real),  img[k] = X[k], 0 < k < M.
    // rtw[k] = exp(-i*pi*k/M) = tw2m[k] with tw2m the 2M-point table.

    // The split/merge twiddles e^{-i pi k / M} of the pairs (k, M-k) this thread owns, k = tid + i*T < M/2.
    template <int LOGM>
    struct real_tw
    {
        static constexpr int ITER = (plan<LOGM>::N / 2 + plan<LOGM>::T - 1) / plan<LOGM>::T;
        float2 w[ITER];
    };

    template <int LOGM>
    __device__ __forceinline__ void load_real_tw(real_tw<LOGM> &r, const float2 *__restrict__ tw2m, int tw_stride /* TWN / 2M */, int tid)
    {
        constexpr int M = plan<LOGM>::N, T = plan<LOGM>::T;
        #pragma unroll
        for (int i = 0; i < real_tw<LOGM>::ITER; ++i)
            r.w[i] = tw2m[((tid + i * T) & (M / 2 - 1)) * tw_stride];
    }

    // Z (the M-point transform of z[n] = x[2n] + i x[2n+1], in buf) -> image, in place.
    template <int LOGM>
    __device__ void real_split(float2 *buf, const real_tw<LOGM> &rt, int tid)
    {
        using P = plan<LOGM>;
        constexpr int M = P::N, T = P::T;
        // pairs (k, M-k), k = 1 .. M/2 - 1; k = 0 and k = M/2 are their own partners
        #pragma unroll
        for (int i = 0; i < real_tw<LOGM>::ITER; ++i)
        {
            const int k = tid + i * T;
            if (k >= M / 2)
                continue;
            if (k == 0)
            {
                const float2 z0 = buf[0];
                buf[0] = make_float2(z0.x + z0.y, z0.x - z0.y);
                buf[M / 2] = cconj(buf[M / 2]);
            }
            else
            {
                // packed: e2 = Zk + conj Zm, o2 = Zk - conj Zm, wo = (w / 2) o2;  X_k = e2 / 2 - i wo,  X_(M-k) = conj(e2 / 2 + i wo)
                const v2f zk = ld2(buf + k), zm = ld2(buf + M - k);
                const v2f wh = v2f{0.5f * rt.w[i].x, 0.5f * rt.w[i].y};   // e^{-i pi k / M} / 2
                const v2f eh = padd_cj(zk, zm) * v2f{0.5f, 0.5f};
                const v2f wo = pmul<false>(wh, psub_cj(zk, zm));
                st2(buf + k, padd_i<false>(eh, wo));
                st2(buf + M - k, pconj_add_i<true>(eh, wo));
            }
        }
        __syncthreads();
    }

    // image (in buf) -> Z such that the UNNORMALISED inverse M-point transform of Z gives
    // ifft_unnorm(Z)[n] = 2M * (x[2n] + i x[2n+1]) where x is the real 2M-point sequence whose UNNORMALISED forward
    // transform is the image.
    template <int LOGM>
    __device__ void real_merge(float2 *buf, const real_tw<LOGM> &rt, int tid)
    {
        using P = plan<LOGM>;
        constexpr int M = P::N, T = P::T;
        #pragma unroll
        for (int i = 0; i < real_tw<LOGM>::ITER; ++i)
        {
            const int k = tid + i * T;
            if (k >= M / 2)
                continue;
            if (k == 0)
            {
                const float2 y0 = buf[0];                                 // (X0, XM)
                buf[0] = make_float2(y0.x + y0.y, y0.x - y0.y);
                const float2 y = buf[M / 2];
                buf[M / 2] = make_float2(2.0f * y.x, -2.0f * y.y);
            }
            else
            {
                // packed: e = Xk + conj Xm, o = Xk - conj Xm, wo = conj(w) o;  Z_k = e + i wo,  Z_(M-k) = conj(e - i wo)
                const v2f xk = ld2(buf + k), xm = ld2(buf + M - k);
                const v2f w  = v2f{rt.w[i].x, rt.w[i].y};
                const v2f e  = padd_cj(xk, xm);
                const v2f wo = pmul<true>(w, psub_cj(xk, xm));
                st2(buf + k, padd_i<true>(e, wo));
                st2(buf + M - k, pconj_add_i<false>(e, wo));
            }
        }
        __syncthreads();
    }

    // real_split, a function of the image's bins, real_merge -- in ONE pass over the buffer: thread by thread the pair
    // (Z_k, Z_(M-k)) becomes the image's (X_k, X_(M-k)), filt(k, X_k) and filt(M - k, X_(M-k)) give the bins that go back
    // (k = 0: the packed (DC, Nyquist) bin; k = M / 2 with it), and the merged pair is stored where the pair came from.  The
    // same arithmetic as the three passes, two barriers and two trips through LDS less.  Synchronises behind the stores.
    // (`arrive` runs in every thread in front of the closing barrier: what a caller must have waited for before the
    // workgroup as a whole moves on)
    template <int LOGM, class F, class A>
    __device__ __forceinline__ void real_split_filter_merge(float2 *buf, const real_tw<LOGM> &rt, int tid, F filt, A arrive)
    {
        using P = plan<LOGM>;
        constexpr int M = P::N, T = P::T;
        #pragma unroll
        for (int i = 0; i < real_tw<LOGM>::ITER; ++i)
        {
            const int k = tid + i * T;
            if (k >= M / 2)
                continue;
            if (k == 0)
            {
                const float2 z0 = buf[0];
                const float2 y0 = filt(i, 0, make_float2(z0.x + z0.y, z0.x - z0.y), false);
                const float2 yh = filt(i, M / 2, cconj(buf[M / 2]), true);
                buf[0] = make_float2(y0.x + y0.y, y0.x - y0.y);
                buf[M / 2] = make_float2(2.0f * yh.x, -2.0f * yh.y);
            }
            else
            {
                const v2f zk = ld2(buf + k), zm = ld2(buf + M - k);
                const v2f wh = v2f{0.5f * rt.w[i].x, 0.5f * rt.w[i].y};
                const v2f eh = padd_cj(zk, zm) * v2f{0.5f, 0.5f};
                const v2f ws = pmul<false>(wh, psub_cj(zk, zm));
                const v2f xk = padd_i<false>(eh, ws), xm = pconj_add_i<true>(eh, ws);
                const float2 yk = filt(i, k, make_float2(xk.x, xk.y), false), ym = filt(i, M - k, make_float2(xm.x, xm.y), true);
                const v2f w  = v2f{rt.w[i].x, rt.w[i].y};
                const v2f e  = padd_cj(v2f{yk.x, yk.y}, v2f{ym.x, ym.y});
                const v2f wo = pmul<true>(w, psub_cj(v2f{yk.x, yk.y}, v2f{ym.x, ym.y}));
                st2(buf + k, padd_i<true>(e, wo));
                st2(buf + M - k, pconj_add_i<false>(e, wo));
            }
        }
        arrive();
        __syncthreads();
    }

    // A 2M-point real transform pair through M-point complex transforms: twiddles in registers, two LDS buffers.
    template <int LOGM>
    struct real_fft
    {
        fft_tw<LOGM>  ft;
        real_tw<LOGM> rt;

        __device__ __forceinline__ void load(const float2 *__restrict__ tw, int twn, int tid)
        {
            constexpr int M = plan<LOGM>::N;
            load_fft_tw<LOGM>(ft, tw, twn / M, tid);
            load_real_tw<LOGM>(rt, tw, twn / (2 * M), tid);
        }
        // after the kernel's other loads have been issued, before the first transform
        __device__ __forceinline__ void prepare() { finish_fft_tw<LOGM>(ft); }
        // packed samples z[n] = x[2n] + i x[2n+1] in buf -> image in buf
        __device__ __forceinline__ void forward(float2 *buf, float2 *scr, int tid) const
        {
            fft_lds<LOGM, false>(buf, scr, ft, tid);
            real_split<LOGM>(buf, rt, tid);
        }
        // image in buf -> 2M real samples (times 2M), left in buf as (x[2n], x[2n+1])
        __device__ __forceinline__ void inverse(float2 *buf, float2 *scr, int tid) const
        {
            real_merge<LOGM>(buf, rt, tid);
            fft_lds<LOGM, true>(buf, scr, ft, tid);
        }

        // The same with the samples handed over in registers (512 .. 8192-point transforms: fft_lds REG_IN / REG_OUT):
        // io[i] = pair tid + i T of the packed sequence, i < M / T.
        static constexpr bool REGS = (plan<LOGM>::T == plan<LOGM>::TB);
        static constexpr int  KPT  = plan<LOGM>::N / plan<LOGM>::T;
        __device__ __forceinline__ void forward_regs(v2f *io, float2 *buf, float2 *scr, int tid) const
        {
            fft_lds<LOGM, false, true, false>(buf, scr, ft, tid, io);
            real_split<LOGM>(buf, rt, tid);
        }
        __device__ __forceinline__ void inverse_regs(v2f *io, float2 *buf, float2 *scr, int tid) const
        {
            real_merge<LOGM>(buf, rt, tid);
            fft_lds<LOGM, true, false, true>(buf, scr, ft, tid, io);
        }
        // Z (the M-point transform of the packed sequence, in buf) -> image, times REAL gains per bin (g[k], k = 0 .. M: a
        // Hermitian-symmetric mask), -> the Z' whose unnormalised inverse is 2M times the masked sequence: real_split, the
        // multiplication and real_merge in ONE pass over the pairs (k, M - k) -- one LDS round trip and one barrier instead
        // of three of each.  With A = e - i w o (= X_k) and Bc = e + i w o (= conj X_(M-k)) of real_split:
        //     e' = g_k A + g_m Bc,  o' = g_k A - g_m Bc,  Z'_k = e' + i conj(w) o',  Z'_(M-k) = conj(e' - i conj(w) o')
        // The same in two steps, for several masks on one spectrum: the thread's pairs are read once into registers
        // (zk[i] = Z_k, zm[i] = Z_(M-k), k = tid + i T < M/2; the thread of k = 0 holds Z_0 and Z_(M/2)) ...
        static constexpr int PAIRS = real_tw<LOGM>::ITER;
        __device__ __forceinline__ void pairs_load(const float2 *buf, float2 (&zk)[PAIRS], float2 (&zm)[PAIRS], int tid) const
        {
            constexpr int M = plan<LOGM>::N, T = plan<LOGM>::T;
            #pragma unroll
            for (int i = 0; i < PAIRS; ++i)
            {
                const int k = tid + i * T;
                zk[i] = (k < M / 2) ? buf[k] : make_float2(0.0f, 0.0f);
                zm[i] = (k < M / 2) ? buf[(k == 0) ? M / 2 : M - k] : make_float2(0.0f, 0.0f);
            }
        }
        // ... and every mask writes its Z' (see mask_pairs) from them; gain(k), k = 0 .. M.  Synchronises behind the stores.
        template <typename G>
        __device__ __forceinline__ void pairs_mask_store(float2 *buf, const float2 (&zk)[PAIRS], const float2 (&zm)[PAIRS], G gain, int tid) const
        {
            constexpr int M = plan<LOGM>::N, T = plan<LOGM>::T;
            #pragma unroll
            for (int i = 0; i < PAIRS; ++i)
            {
                const int k = tid + i * T;
                if (k >= M / 2)
                    continue;
                if (k == 0)
                {
                    const float x0 = (zk[i].x + zk[i].y) * gain(0), xm = (zk[i].x - zk[i].y) * gain(M);
                    buf[0] = make_float2(x0 + xm, x0 - xm);
                    const float gh = 2.0f * gain(M / 2);
                    buf[M / 2] = make_float2(gh * zm[i].x, gh * zm[i].y);
                }
                else
                {
                    // packed (the halves of real_split go into the gains): A = g_k X_k, Bc = g_m conj X_(M-k)
                    const v2f w  = v2f{rt.w[i].x, rt.w[i].y};
                    const float gk = 0.5f * gain(k), gm = 0.5f * gain(M - k);
                    const v2f Zk = v2f{zk[i].x, zk[i].y}, Zm = v2f{zm[i].x, zm[i].y};
                    const v2f e  = padd_cj(Zk, Zm);
                    const v2f wo = pmul<false>(w, psub_cj(Zk, Zm));
                    const v2f A  = padd_i<false>(e, wo) * v2f{gk, gk};
                    const v2f Bc = padd_i<true>(e, wo) * v2f{gm, gm};
                    const v2f e2 = A + Bc;
                    const v2f wq = pmul<true>(w, A - Bc);
                    st2(buf + k, padd_i<true>(e2, wq));
                    st2(buf + M - k, pconj_add_i<false>(e2, wq));
                }
            }
            __syncthreads();
        }
        template <class GAINS /* a pointer to the M + 1 gains, in whatever address space the caller knows them to be */>
        __device__ __forceinline__ void mask_pairs(float2 *buf, GAINS g, int tid) const
        {
            constexpr int M = plan<LOGM>::N, T = plan<LOGM>::T;
            #pragma unroll
            for (int i = 0; i < real_tw<LOGM>::ITER; ++i)
            {
                const int k = tid + i * T;
                if (k >= M / 2)
                    continue;
                if (k == 0)
                {
                    const float2 z0 = buf[0];
                    const float x0 = (z0.x + z0.y) * g[0], xm = (z0.x - z0.y) * g[M];
                    buf[0] = make_float2(x0 + xm, x0 - xm);
                    const float2 zh = buf[M / 2];
                    const float gh = 2.0f * g[M / 2];
                    buf[M / 2] = make_float2(gh * zh.x, gh * zh.y);
                }
                else
                {
                    const v2f Zk = ld2(buf + k), Zm = ld2(buf + M - k);
                    const v2f w  = v2f{rt.w[i].x, rt.w[i].y};
                    const float gk = 0.5f * g[k], gm = 0.5f * g[M - k];
                    const v2f e  = padd_cj(Zk, Zm);
                    const v2f wo = pmul<false>(w, psub_cj(Zk, Zm));
                    const v2f A  = padd_i<false>(e, wo) * v2f{gk, gk};
                    const v2f Bc = padd_i<true>(e, wo) * v2f{gm, gm};
                    const v2f e2 = A + Bc;
                    const v2f wq = pmul<true>(w, A - Bc);
                    st2(buf + k, padd_i<true>(e2, wq));
                    st2(buf + M - k, pconj_add_i<false>(e2, wq));
                }
            }
            __syncthreads();
        }
    };
} // namespace mi_fft
