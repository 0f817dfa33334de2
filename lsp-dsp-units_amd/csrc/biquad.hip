// Biquad cascade bank for gfx950: the GPU side of lsp::dspu::FilterBank::process
// (reference: src/main/filters/FilterBank.cpp:256-291, which calls
// dsp::biquad_process_x8/x4/x2/x1 of lsp-dsp-lib once per packed bank).
//
// Why this is not "one channel per lane, serial in time": a block of N samples
// through one section is a chain of 2N dependent FMAs; at N = 4096 that chain
// alone is longer than the whole HBM budget of the block.  The kernel therefore
// cuts every channel's block into chunks of L samples, one chunk per lane, and
// runs every section in three steps (state s = {d0,d1}, s' = A s + B x):
//
//   1. zero-state response of the chunk's end state:  z = sum_k A^(L-1-k) B x[k]
//      -> two dot products with per-section tables p[],q[];
//   2. prefix over chunks  E_t = P E_(t-1) + z_t,  P = A^L, done in two levels so
//      that no data-dependent lane shuffles are needed:
//        a. inclusive scan inside each row of 16 lanes with DPP row_shr 1,2,4,8
//           and the uniform matrices P, P^2, P^4, P^8;
//        b. the four row totals of a wave are chained with P^16 (uniform math),
//           waves are chained through one LDS word pair and one barrier;
//        c. every lane adds P^(i+1) C_row, i = lane % 16, C_row = state entering
//           its row (a 16-entry matrix table, the same for all rows);
//      the state the channel carried in from the previous call enters at chunk 0;
//   3. the exact TDF-II recurrence over the chunk, started from E_(t-1):
//         y = b0 x + d0;  d0 = (b1 x + d1) + a1 y;  d1 = b2 x + a2 y
//      -- the reference's own per-sample arithmetic; only the chunk start state
//      carries the (float32 round-off sized) difference of steps 1-2.
//
// All sections of a channel run back to back on samples held in registers, so
// HBM sees each sample once in and once out (8 B per channel-sample).
// A workgroup owns one channel: it loads the block with coalesced 16-B loads,
// transposes it through a padded LDS tile so each lane gets its L consecutive
// samples (conflict-free ds_read_b128: row pitch L+4 dwords, (L+4)/4 odd), and
// stores the result the same way back.  The per-section tables of the channel
// are staged in LDS (8 sections at a time) and read as broadcasts.
#include "mi_common.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace
{
    template <int L, int NT>
    struct geom
    {
        static constexpr int TAB    = 72 + 2 * L;           // floats per (channel, section)
        static constexpr int PITCH  = L + 4;                // LDS dwords per chunk
        static constexpr int BLOCK  = L * NT;               // samples per launch and channel
        static constexpr int NW     = NT / 64;              // waves per workgroup
        static constexpr int SG     = 8;                    // sections whose tables are staged at once
        static_assert(((PITCH / 4) & 1) == 1, "LDS pitch must be an odd number of 16-B slots");
        static_assert((TAB % 4) == 0, "table rows stay 16-B aligned");
        static_assert(NT % 64 == 0, "whole waves");
    };

    // Table row of one section:
    //   [0..4]   b0 b1 b2 a1 a2            [5..7] unused
    //   [8+4i..] P^(i+1) row-major, i = 0..15   (P = A^L; P,P^2,P^4,P^8 drive the row scan, P^16 the chain)
    //   [72..]   p[L], q[L]
    template <int CTRL>
    __device__ __forceinline__ float dpp_zero(float v)
    {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
    }

    __device__ __forceinline__ float lane_value(float v, int lane)
    {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
    }

    template <int L, int NT, bool ALIGNED, bool FULL>
    __global__ __launch_bounds__(NT)
    void biquad_bank_kernel(float *out, const float *in, size_t out_stride, size_t in_stride,
                            int cnt, const float *__restrict__ tab, float *state,
                            const uint32_t *__restrict__ nsec, int max_sec)
    {
        using G = geom<L, NT>;
        constexpr int TAB   = G::TAB;
        constexpr int PITCH = G::PITCH;
        constexpr int NW    = G::NW;
        constexpr int SG    = G::SG;

        constexpr int TQ    = SG * TAB / 4;                 // float4 per staged table group
        constexpr int TPT   = (TQ + NT - 1) / NT;           // float4 per thread and group

        __shared__ __attribute__((aligned(16))) float sx[NT * PITCH];
        __shared__ __attribute__((aligned(16))) float stab[SG * TAB];
        __shared__ float2 sstate[SG];
        __shared__ float2 stot[2][NW];

        const int ch    = blockIdx.x;
        const int t     = threadIdx.x;
        const int lane  = t & 63;
        const int l16   = t & 15;
        const int row   = lane >> 4;
        const int wave  = t >> 6;
        const int ns    = int(nsec[ch]);
        const float *xin = in + size_t(ch) * in_stride;
        float *yout      = out + size_t(ch) * out_stride;

        // tables and carried state of the first section group: issued before the samples so that
        // their latency hides behind the block load
        float4 tpre[TPT];
        float2 spre = make_float2(0.0f, 0.0f);
        {
            const int group = (ns < SG) ? ns : SG;
            const float4 *src = reinterpret_cast<const float4 *>(tab + size_t(ch) * max_sec * TAB);
            #pragma unroll
            for (int j = 0; j < TPT; ++j)
            {
                const int i = t + j * NT;
                tpre[j] = (i < group * (TAB / 4)) ? src[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            if (t < group)
                spre = reinterpret_cast<const float2 *>(state + size_t(ch) * max_sec * 2)[t];
        }

        // ---- coalesced load, transposed through LDS -------------------------------------
        float x[L];
        #pragma unroll
        for (int k = 0; k < L / 4; ++k)
        {
            const int i = 4 * (k * NT + t);
            float4 v;
            if (ALIGNED && (FULL || i + 4 <= cnt))
                v = *reinterpret_cast<const float4 *>(xin + i);
            else
            {
                v.x = (i + 0 < cnt) ? xin[i + 0] : 0.0f;
                v.y = (i + 1 < cnt) ? xin[i + 1] : 0.0f;
                v.z = (i + 2 < cnt) ? xin[i + 2] : 0.0f;
                v.w = (i + 3 < cnt) ? xin[i + 3] : 0.0f;
            }
            *reinterpret_cast<float4 *>(&sx[i + (i / L) * 4]) = v;
        }
        __syncthreads();
        #pragma unroll
        for (int k = 0; k < L / 4; ++k)
        {
            const float4 v = *reinterpret_cast<const float4 *>(&sx[t * PITCH + 4 * k]);
            x[4 * k + 0] = v.x; x[4 * k + 1] = v.y; x[4 * k + 2] = v.z; x[4 * k + 3] = v.w;
        }

        // Lane that owns the last valid sample of the block, and how many it owns.
        const int t_last = FULL ? (NT - 1) : ((cnt - 1) / L);
        const int m_last = FULL ? L : (cnt - t_last * L);

        // ---- sections, strictly in series (FilterBank.cpp:267-290) ----------------------
        for (int s0 = 0; s0 < ns; s0 += SG)
        {
            const int group = (ns - s0 < SG) ? (ns - s0) : SG;
            if (s0 > 0)
            {
                __syncthreads();                        // everybody is done with the previous tables
                const float4 *src = reinterpret_cast<const float4 *>(tab + (size_t(ch) * max_sec + s0) * TAB);
                #pragma unroll
                for (int j = 0; j < TPT; ++j)
                {
                    const int i = t + j * NT;
                    tpre[j] = (i < group * (TAB / 4)) ? src[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
                if (t < group)
                    spre = reinterpret_cast<const float2 *>(state + (size_t(ch) * max_sec + s0) * 2)[t];
            }
            #pragma unroll
            for (int j = 0; j < TPT; ++j)
            {
                const int i = t + j * NT;
                if (i < TQ)
                    reinterpret_cast<float4 *>(stab)[i] = tpre[j];
            }
            if (t < SG)
                sstate[t] = spre;
            __syncthreads();

            for (int si = 0; si < group; ++si)
            {
                const int s     = s0 + si;
                const float *T  = stab + si * TAB;
                float *st       = state + (size_t(ch) * max_sec + s) * 2;
                const float4 cf = *reinterpret_cast<const float4 *>(T);
                const float b0 = cf.x, b1 = cf.y, b2 = cf.z, a1 = cf.w, a2 = T[4];
                const float2 cs = sstate[si];           // state carried in from the previous call
                const float c0 = cs.x, c1 = cs.y;

                // 1. end state of the chunk for zero start state
                float z0 = 0.0f, z1 = 0.0f, w0 = 0.0f, w1 = 0.0f;
                #pragma unroll
                for (int k = 0; k < L; k += 4)
                {
                    const float4 p = *reinterpret_cast<const float4 *>(T + 72 + k);
                    const float4 q = *reinterpret_cast<const float4 *>(T + 72 + L + k);
                    z0 = fmaf(p.x, x[k + 0], z0); w0 = fmaf(q.x, x[k + 0], w0);
                    z1 = fmaf(p.y, x[k + 1], z1); w1 = fmaf(q.y, x[k + 1], w1);
                    z0 = fmaf(p.z, x[k + 2], z0); w0 = fmaf(q.z, x[k + 2], w0);
                    z1 = fmaf(p.w, x[k + 3], z1); w1 = fmaf(q.w, x[k + 3], w1);
                }
                float z = z0 + z1, w = w0 + w1;
                const float4 P1  = *reinterpret_cast<const float4 *>(T + 8 + 4 * 0);
                if (t == 0)
                {
                    z = fmaf(P1.x, c0, fmaf(P1.y, c1, z));
                    w = fmaf(P1.z, c0, fmaf(P1.w, c1, w));
                }

                // 2a. inclusive scan inside rows of 16 lanes
                {
                    const float4 P2 = *reinterpret_cast<const float4 *>(T + 8 + 4 * 1);
                    const float4 P4 = *reinterpret_cast<const float4 *>(T + 8 + 4 * 3);
                    const float4 P8 = *reinterpret_cast<const float4 *>(T + 8 + 4 * 7);
                    float zs, ws;
                    zs = dpp_zero<0x111>(z); ws = dpp_zero<0x111>(w);
                    z = fmaf(P1.x, zs, fmaf(P1.y, ws, z)); w = fmaf(P1.z, zs, fmaf(P1.w, ws, w));
                    zs = dpp_zero<0x112>(z); ws = dpp_zero<0x112>(w);
                    z = fmaf(P2.x, zs, fmaf(P2.y, ws, z)); w = fmaf(P2.z, zs, fmaf(P2.w, ws, w));
                    zs = dpp_zero<0x114>(z); ws = dpp_zero<0x114>(w);
                    z = fmaf(P4.x, zs, fmaf(P4.y, ws, z)); w = fmaf(P4.z, zs, fmaf(P4.w, ws, w));
                    zs = dpp_zero<0x118>(z); ws = dpp_zero<0x118>(w);
                    z = fmaf(P8.x, zs, fmaf(P8.y, ws, z)); w = fmaf(P8.z, zs, fmaf(P8.w, ws, w));
                }

                // 2b. chain the row totals (uniform per wave), waves one after another
                const float4 P16 = *reinterpret_cast<const float4 *>(T + 8 + 4 * 15);
                const float t0x = lane_value(z, 15), t0y = lane_value(w, 15);
                const float t1x = lane_value(z, 31), t1y = lane_value(w, 31);
                const float t2x = lane_value(z, 47), t2y = lane_value(w, 47);
                const float t3x = lane_value(z, 63), t3y = lane_value(w, 63);
                float cinx = 0.0f, ciny = 0.0f;         // state entering this wave (carry is already in lane 0)
                float c1x, c1y, c2x, c2y, c3x, c3y;
                #pragma unroll
                for (int wv = 0; wv < NW; ++wv)
                {
                    if (wave == wv)
                    {
                        c1x = fmaf(P16.x, cinx, fmaf(P16.y, ciny, t0x)); c1y = fmaf(P16.z, cinx, fmaf(P16.w, ciny, t0y));
                        c2x = fmaf(P16.x, c1x, fmaf(P16.y, c1y, t1x));   c2y = fmaf(P16.z, c1x, fmaf(P16.w, c1y, t1y));
                        c3x = fmaf(P16.x, c2x, fmaf(P16.y, c2y, t2x));   c3y = fmaf(P16.z, c2x, fmaf(P16.w, c2y, t2y));
                        if (wv + 1 < NW && lane == 0)
                        {
                            const float ex = fmaf(P16.x, c3x, fmaf(P16.y, c3y, t3x));
                            const float ey = fmaf(P16.z, c3x, fmaf(P16.w, c3y, t3y));
                            stot[s & 1][wv] = make_float2(ex, ey);
                        }
                    }
                    if (wv + 1 < NW)
                    {
                        __syncthreads();
                        if (wave == wv + 1)
                        {
                            const float2 e = stot[s & 1][wv];
                            cinx = e.x;
                            ciny = e.y;
                        }
                    }
                }

                // 2c. state entering the lane's row, pushed through the lane's own power of P
                const float crx = (row == 0) ? cinx : (row == 1) ? c1x : (row == 2) ? c2x : c3x;
                const float cry = (row == 0) ? ciny : (row == 1) ? c1y : (row == 2) ? c2y : c3y;
                const float4 PL = *reinterpret_cast<const float4 *>(T + 8 + 4 * l16);
                z = fmaf(PL.x, crx, fmaf(PL.y, cry, z));
                w = fmaf(PL.z, crx, fmaf(PL.w, cry, w));

                // start state of the chunk = end state of the previous chunk
                float d0 = dpp_zero<0x111>(z);
                float d1 = dpp_zero<0x111>(w);
                if (l16 == 0)
                {
                    d0 = crx;
                    d1 = cry;
                }
                if (t == 0)
                {
                    d0 = c0;
                    d1 = c1;
                }

                // 3. exact recurrence over the chunk
                float f0 = d0, f1 = d1;
                #pragma unroll
                for (int k = 0; k < L; ++k)
                {
                    const float xx = x[k];
                    const float y  = fmaf(b0, xx, d0);
                    const float tt = fmaf(b1, xx, d1);
                    d0   = fmaf(a1, y, tt);
                    d1   = fmaf(a2, y, b2 * xx);
                    x[k] = y;
                    if (!FULL && (k + 1 == m_last))
                    {
                        f0 = d0;
                        f1 = d1;
                    }
                }
                if (FULL)
                {
                    f0 = d0;
                    f1 = d1;
                }
                if (t == t_last)
                {
                    st[0] = f0;
                    st[1] = f1;
                }
            }
        }

        // ---- transposed back through LDS, coalesced store -------------------------------
        __syncthreads();
        #pragma unroll
        for (int k = 0; k < L / 4; ++k)
            *reinterpret_cast<float4 *>(&sx[t * PITCH + 4 * k]) =
                make_float4(x[4 * k + 0], x[4 * k + 1], x[4 * k + 2], x[4 * k + 3]);
        __syncthreads();
        #pragma unroll
        for (int k = 0; k < L / 4; ++k)
        {
            const int i = 4 * (k * NT + t);
            const float4 v = *reinterpret_cast<const float4 *>(&sx[i + (i / L) * 4]);
            if (ALIGNED && (FULL || i + 4 <= cnt))
                *reinterpret_cast<float4 *>(yout + i) = v;
            else
            {
                if (i + 0 < cnt) yout[i + 0] = v.x;
                if (i + 1 < cnt) yout[i + 1] = v.y;
                if (i + 2 < cnt) yout[i + 2] = v.z;
                if (i + 3 < cnt) yout[i + 3] = v.w;
            }
        }
    }

    // ------------------------------------------------------------------------------------------
    // Packed variant for long blocks: ONE wave per channel, every lane owns TWO chunks (chunk t of
    // the first half of the block and chunk t of the second half) and runs them as the two halves
    // of v_pk_fma_f32 operands.  A lone wave issues one VALU instruction per 4 cycles, which is
    // half the SIMD's rate for plain v_fma_f32 but the full rate for packed fp32, so this shape
    // gets full VALU throughput at one wave per SIMD, needs no workgroup barrier inside the
    // section loop, and leaves the whole 512-register file to the wave.
    // ------------------------------------------------------------------------------------------
    typedef float v2f __attribute__((ext_vector_type(2)));

    __device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
    __device__ __forceinline__ v2f splat(float a) { return v2f{a, a}; }

    template <int CTRL>
    __device__ __forceinline__ v2f dpp_zero2(v2f v)
    {
        return v2f{dpp_zero<CTRL>(v.x), dpp_zero<CTRL>(v.y)};
    }

    template <int L, bool ALIGNED, bool FULL>
    __global__ __launch_bounds__(64, 1)
    void biquad_bank_kernel_pk(float *out, const float *in, size_t out_stride, size_t in_stride,
                               int cnt, const float *__restrict__ tab, float *state,
                               const uint32_t *__restrict__ nsec, int max_sec)
    {
        constexpr int NT    = 64;
        constexpr int NC    = 128;                          // chunks per block
        constexpr int TAB   = 72 + 2 * L;
        constexpr int PITCH = L + 4;
        constexpr int SG    = 8;
        constexpr int TQ    = SG * TAB / 4;
        constexpr int TPT   = (TQ + NT - 1) / NT;

        __shared__ __attribute__((aligned(16))) float sx[NC * PITCH];
        __shared__ __attribute__((aligned(16))) float stab[SG * TAB];
        __shared__ float2 sstate[SG];

        const int ch    = blockIdx.x;
        const int t     = threadIdx.x;
        const int l16   = t & 15;
        const int row   = t >> 4;
        const int ns    = int(nsec[ch]);
        const float *xin = in + size_t(ch) * in_stride;
        float *yout      = out + size_t(ch) * out_stride;

        float4 tpre[TPT];
        float2 spre = make_float2(0.0f, 0.0f);
        {
            const int group = (ns < SG) ? ns : SG;
            const float4 *src = reinterpret_cast<const float4 *>(tab + size_t(ch) * max_sec * TAB);
            #pragma unroll
            for (int j = 0; j < TPT; ++j)
            {
                const int i = t + j * NT;
                tpre[j] = (i < group * (TAB / 4)) ? src[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            if (t < group)
                spre = reinterpret_cast<const float2 *>(state + size_t(ch) * max_sec * 2)[t];
        }

        // ---- coalesced load, transposed through LDS -------------------------------------
        #pragma unroll
        for (int k = 0; k < NC * L / 4 / NT; ++k)
        {
            const int i = 4 * (k * NT + t);
            float4 v;
            if (ALIGNED && (FULL || i + 4 <= cnt))
                v = *reinterpret_cast<const float4 *>(xin + i);
            else
            {
                v.x = (i + 0 < cnt) ? xin[i + 0] : 0.0f;
                v.y = (i + 1 < cnt) ? xin[i + 1] : 0.0f;
                v.z = (i + 2 < cnt) ? xin[i + 2] : 0.0f;
                v.w = (i + 3 < cnt) ? xin[i + 3] : 0.0f;
            }
            *reinterpret_cast<float4 *>(&sx[i + (i / L) * 4]) = v;
        }
        __syncthreads();
        v2f x[L];                                            // .x: chunk t, .y: chunk t + 64
        #pragma unroll
        for (int k = 0; k < L / 4; ++k)
        {
            const float4 a = *reinterpret_cast<const float4 *>(&sx[t * PITCH + 4 * k]);
            const float4 b = *reinterpret_cast<const float4 *>(&sx[(t + 64) * PITCH + 4 * k]);
            x[4 * k + 0] = v2f{a.x, b.x}; x[4 * k + 1] = v2f{a.y, b.y};
            x[4 * k + 2] = v2f{a.z, b.z}; x[4 * k + 3] = v2f{a.w, b.w};
        }

        // chunk (0..127) that owns the last valid sample, and how many samples it owns
        const int c_last = FULL ? (NC - 1) : ((cnt - 1) / L);
        const int m_last = FULL ? L : (cnt - c_last * L);

        for (int s0 = 0; s0 < ns; s0 += SG)
        {
            const int group = (ns - s0 < SG) ? (ns - s0) : SG;
            if (s0 > 0)
            {
                __syncthreads();
                const float4 *src = reinterpret_cast<const float4 *>(tab + (size_t(ch) * max_sec + s0) * TAB);
                #pragma unroll
                for (int j = 0; j < TPT; ++j)
                {
                    const int i = t + j * NT;
                    tpre[j] = (i < group * (TAB / 4)) ? src[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
                if (t < group)
                    spre = reinterpret_cast<const float2 *>(state + (size_t(ch) * max_sec + s0) * 2)[t];
            }
            #pragma unroll
            for (int j = 0; j < TPT; ++j)
            {
                const int i = t + j * NT;
                if (i < TQ)
                    reinterpret_cast<float4 *>(stab)[i] = tpre[j];
            }
            if (t < SG)
                sstate[t] = spre;
            __syncthreads();

            // Section tables are read from LDS as broadcasts (all lanes, same address).
            constexpr int TAB4 = TAB / 4;
            float4 ta[TAB4];
            float4 pla;
            float2 csa;
            auto prefetch = [&](float4 (&tt)[TAB4], float4 &pl, float2 &cs, int si)
            {
                const float *T = stab + si * TAB;
                #pragma unroll
                for (int j = 0; j < TAB4; ++j)
                    tt[j] = reinterpret_cast<const float4 *>(T)[j];
                pl = *reinterpret_cast<const float4 *>(T + 8 + 4 * l16);
                cs = sstate[si];
            };
            auto body = [&](const float4 (&tt)[TAB4], const float4 &PL, const float2 &cs, int s)
            {
                float *st       = state + (size_t(ch) * max_sec + s) * 2;
                const float4 cf = tt[0];
                const v2f b0 = splat(cf.x), b1 = splat(cf.y), b2 = splat(cf.z), a1 = splat(cf.w), a2 = splat(tt[1].x);
                const float c0 = cs.x, c1 = cs.y;

                // 1. end state of both chunks for zero start state
                v2f z0 = splat(0.0f), z1 = splat(0.0f), w0 = splat(0.0f), w1 = splat(0.0f);
                #pragma unroll
                for (int k = 0; k < L; k += 4)
                {
                    const float4 p = tt[18 + k / 4];
                    const float4 q = tt[18 + L / 4 + k / 4];
                    z0 = pk_fma(splat(p.x), x[k + 0], z0); w0 = pk_fma(splat(q.x), x[k + 0], w0);
                    z1 = pk_fma(splat(p.y), x[k + 1], z1); w1 = pk_fma(splat(q.y), x[k + 1], w1);
                    z0 = pk_fma(splat(p.z), x[k + 2], z0); w0 = pk_fma(splat(q.z), x[k + 2], w0);
                    z1 = pk_fma(splat(p.w), x[k + 3], z1); w1 = pk_fma(splat(q.w), x[k + 3], w1);
                }
                v2f z = z0 + z1, w = w0 + w1;
                const float4 P1 = tt[2 + 0];
                if (t == 0)
                {
                    z.x = fmaf(P1.x, c0, fmaf(P1.y, c1, z.x));
                    w.x = fmaf(P1.z, c0, fmaf(P1.w, c1, w.x));
                }

                // 2a. inclusive scan inside rows of 16 lanes (both halves at once)
                {
                    const float4 P2 = tt[2 + 1];
                    const float4 P4 = tt[2 + 3];
                    const float4 P8 = tt[2 + 7];
                    v2f zs, ws;
                    zs = dpp_zero2<0x111>(z); ws = dpp_zero2<0x111>(w);
                    z = pk_fma(splat(P1.x), zs, pk_fma(splat(P1.y), ws, z)); w = pk_fma(splat(P1.z), zs, pk_fma(splat(P1.w), ws, w));
                    zs = dpp_zero2<0x112>(z); ws = dpp_zero2<0x112>(w);
                    z = pk_fma(splat(P2.x), zs, pk_fma(splat(P2.y), ws, z)); w = pk_fma(splat(P2.z), zs, pk_fma(splat(P2.w), ws, w));
                    zs = dpp_zero2<0x114>(z); ws = dpp_zero2<0x114>(w);
                    z = pk_fma(splat(P4.x), zs, pk_fma(splat(P4.y), ws, z)); w = pk_fma(splat(P4.z), zs, pk_fma(splat(P4.w), ws, w));
                    zs = dpp_zero2<0x118>(z); ws = dpp_zero2<0x118>(w);
                    z = pk_fma(splat(P8.x), zs, pk_fma(splat(P8.y), ws, z)); w = pk_fma(splat(P8.z), zs, pk_fma(splat(P8.w), ws, w));
                }

                // 2b. chain the eight row totals: rows 0..3 = first half, 4..7 = second half
                const float4 P16 = tt[2 + 15];
                float cx[8], cy[8];
                cx[0] = 0.0f; cy[0] = 0.0f;                 // the carry is already inside lane 0
                #pragma unroll
                for (int r = 0; r < 7; ++r)
                {
                    const int ln = 16 * (r & 3) + 15;
                    const float tx = (r < 4) ? lane_value(z.x, ln) : lane_value(z.y, ln);
                    const float ty = (r < 4) ? lane_value(w.x, ln) : lane_value(w.y, ln);
                    cx[r + 1] = fmaf(P16.x, cx[r], fmaf(P16.y, cy[r], tx));
                    cy[r + 1] = fmaf(P16.z, cx[r], fmaf(P16.w, cy[r], ty));
                }

                // 2c. state entering the lane's rows, pushed through the lane's own power of P
                // (selects written as a flat chain so they stay v_cndmask, not branches)
                const bool r1 = (row == 1), r2 = (row == 2), r3 = (row == 3);
                v2f crx = v2f{cx[0], cx[4]}, cry = v2f{cy[0], cy[4]};
                crx.x = r1 ? cx[1] : crx.x; cry.x = r1 ? cy[1] : cry.x; crx.y = r1 ? cx[5] : crx.y; cry.y = r1 ? cy[5] : cry.y;
                crx.x = r2 ? cx[2] : crx.x; cry.x = r2 ? cy[2] : cry.x; crx.y = r2 ? cx[6] : crx.y; cry.y = r2 ? cy[6] : cry.y;
                crx.x = r3 ? cx[3] : crx.x; cry.x = r3 ? cy[3] : cry.x; crx.y = r3 ? cx[7] : crx.y; cry.y = r3 ? cy[7] : cry.y;
                z = pk_fma(splat(PL.x), crx, pk_fma(splat(PL.y), cry, z));
                w = pk_fma(splat(PL.z), crx, pk_fma(splat(PL.w), cry, w));

                // start state of each chunk = end state of the chunk before it
                v2f d0 = dpp_zero2<0x111>(z);
                v2f d1 = dpp_zero2<0x111>(w);
                if (l16 == 0)
                {
                    d0 = crx;
                    d1 = cry;
                }
                if (t == 0)
                {
                    d0.x = c0;
                    d1.x = c1;
                }

                // 3. exact recurrence over both chunks
                v2f f0 = d0, f1 = d1;
                #pragma unroll
                for (int k = 0; k < L; ++k)
                {
                    const v2f xx = x[k];
                    const v2f y  = pk_fma(b0, xx, d0);
                    const v2f tt2 = pk_fma(b1, xx, d1);
                    d0   = pk_fma(a1, y, tt2);
                    d1   = pk_fma(a2, y, b2 * xx);
                    x[k] = y;
                    if (!FULL && (k + 1 == m_last))
                    {
                        f0 = d0;
                        f1 = d1;
                    }
                }
                if (FULL)
                {
                    f0 = d0;
                    f1 = d1;
                }
                if (t == (c_last & 63))
                {
                    st[0] = (c_last < 64) ? f0.x : f0.y;
                    st[1] = (c_last < 64) ? f1.x : f1.y;
                }
            };

            for (int si = 0; si < group; ++si)
            {
                prefetch(ta, pla, csa, si);
                body(ta, pla, csa, s0 + si);
            }
        }

        // ---- transposed back through LDS, coalesced store -------------------------------
        __syncthreads();
        #pragma unroll
        for (int k = 0; k < L / 4; ++k)
        {
            *reinterpret_cast<float4 *>(&sx[t * PITCH + 4 * k]) =
                make_float4(x[4 * k + 0].x, x[4 * k + 1].x, x[4 * k + 2].x, x[4 * k + 3].x);
            *reinterpret_cast<float4 *>(&sx[(t + 64) * PITCH + 4 * k]) =
                make_float4(x[4 * k + 0].y, x[4 * k + 1].y, x[4 * k + 2].y, x[4 * k + 3].y);
        }
        __syncthreads();
        #pragma unroll
        for (int k = 0; k < NC * L / 4 / NT; ++k)
        {
            const int i = 4 * (k * NT + t);
            const float4 v = *reinterpret_cast<const float4 *>(&sx[i + (i / L) * 4]);
            if (ALIGNED && (FULL || i + 4 <= cnt))
                *reinterpret_cast<float4 *>(yout + i) = v;
            else
            {
                if (i + 0 < cnt) yout[i + 0] = v.x;
                if (i + 1 < cnt) yout[i + 1] = v.y;
                if (i + 2 < cnt) yout[i + 2] = v.z;
                if (i + 3 < cnt) yout[i + 3] = v.w;
            }
        }
    }

    __global__ void impulse_kernel(float *out, size_t stride, size_t samples, uint32_t channels)
    {
        // FilterBank.cpp:316-318: zero the buffer, out[0] = 1
        const size_t total = size_t(channels) * samples;
        for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
             i += size_t(gridDim.x) * blockDim.x)
        {
            const size_t c = i / samples, k = i - c * samples;
            out[c * stride + k] = (k == 0) ? 1.0f : 0.0f;
        }
    }

    // ---- host-side tables -------------------------------------------------------------------
    struct mat2 { double a, b, c, d; };
    inline mat2 mul(const mat2 &x, const mat2 &y)
    {
        return { x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d,
                 x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d };
    }

    template <int L, int NT>
    void fill_row(float *row, const float *q /* b0 b1 b2 a1 a2 */)
    {
        const double b0 = q[0], b1 = q[1], b2 = q[2], a1 = q[3], a2 = q[4];
        for (int i = 0; i < 8; ++i)
            row[i] = (i < 5) ? q[i] : 0.0f;
        // s' = A s + B x  with  A = [a1 1; a2 0],  B = [b1 + a1 b0, b2 + a2 b0]
        const mat2 A = { a1, 1.0, a2, 0.0 };
        double v0 = b1 + a1 * b0, v1 = b2 + a2 * b0;
        float *p = row + 72, *qq = p + L;
        for (int k = L - 1; k >= 0; --k)        // p[k],q[k] = A^(L-1-k) B
        {
            p[k]  = float(v0);
            qq[k] = float(v1);
            const double n0 = A.a * v0 + A.b * v1, n1 = A.c * v0 + A.d * v1;
            v0 = n0;
            v1 = n1;
        }
        mat2 P = { 1.0, 0.0, 0.0, 1.0 };
        for (int k = 0; k < L; ++k)
            P = mul(P, A);
        mat2 Pi = P;                             // P^(i+1)
        for (int i = 0; i < 16; ++i)
        {
            float *m = row + 8 + 4 * i;
            m[0] = float(Pi.a); m[1] = float(Pi.b); m[2] = float(Pi.c); m[3] = float(Pi.d);
            Pi = mul(Pi, P);
        }
    }

    using big   = geom<32, 128>;    // blocks of up to 4096 samples per launch
    using small = geom<8, 64>;      // blocks of up to 512 samples per launch
} // namespace

struct mi_biquad_bank
{
    uint32_t                channels    = 0;
    uint32_t                max_sec     = 0;
    std::vector<uint32_t>   nsec;           // FilterBank::nItems per channel
    std::vector<int64_t>    last_nsec;      // FilterBank::nLastItems (-1 after init)
    std::vector<float>      coef;           // [channels][max_sec][5]
    std::vector<uint8_t>    dirty;          // tables of the channel need a rebuild
    std::vector<uint8_t>    clear;          // delay memory of the channel must be cleared
    bool                    pending     = false;
    std::vector<float>      h_big, h_small; // host images of the device tables
    float                  *d_big       = nullptr;
    float                  *d_small     = nullptr;
    float                  *d_state     = nullptr;
    float                  *d_backup    = nullptr;
    uint32_t               *d_nsec      = nullptr;
};

namespace
{
    template <int L, int NT>
    hipError_t launch(mi_biquad_bank *b, float *out, const float *in, size_t out_stride,
                      size_t in_stride, int cnt, bool aligned, const float *tab, hipStream_t st)
    {
        const dim3 grid(b->channels), block(NT);
        const bool full = (cnt == L * NT);
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        mi::take_profile_events(&ev0, &ev1);
        #define MI_LAUNCH(A, F)                                                                   \
            hipExtLaunchKernelGGL((biquad_bank_kernel<L, NT, A, F>), grid, block, 0, st, ev0, ev1, 0, out, in, \
                               out_stride, in_stride, cnt, tab, b->d_state, b->d_nsec, int(b->max_sec))
        if (aligned) { if (full) MI_LAUNCH(true, true); else MI_LAUNCH(true, false); }
        else         { if (full) MI_LAUNCH(false, true); else MI_LAUNCH(false, false); }
        #undef MI_LAUNCH
        return hipGetLastError();
    }

    template <int L>
    hipError_t launch_pk(mi_biquad_bank *b, float *out, const float *in, size_t out_stride,
                         size_t in_stride, int cnt, bool aligned, const float *tab, hipStream_t st)
    {
        const dim3 grid(b->channels), block(64);
        const bool full = (cnt == L * 128);
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        mi::take_profile_events(&ev0, &ev1);
        #define MI_LAUNCH(A, F)                                                                   \
            hipExtLaunchKernelGGL((biquad_bank_kernel_pk<L, A, F>), grid, block, 0, st, ev0, ev1, 0, out, in, \
                               out_stride, in_stride, cnt, tab, b->d_state, b->d_nsec, int(b->max_sec))
        if (aligned) { if (full) MI_LAUNCH(true, true); else MI_LAUNCH(true, false); }
        else         { if (full) MI_LAUNCH(false, true); else MI_LAUNCH(false, false); }
        #undef MI_LAUNCH
        return hipGetLastError();
    }

    int commit(mi_biquad_bank *b, hipStream_t st)
    {
        if (!b->pending)
            return MI_OK;
        const size_t row_big = size_t(b->max_sec) * big::TAB, row_small = size_t(b->max_sec) * small::TAB;
        size_t n_dirty = 0;
        for (uint32_t c = 0; c < b->channels; ++c)
        {
            if (!b->dirty[c])
                continue;
            ++n_dirty;
            for (uint32_t s = 0; s < b->nsec[c]; ++s)
            {
                const float *q = &b->coef[(size_t(c) * b->max_sec + s) * 5];
                fill_row<32, 128>(&b->h_big[c * row_big + size_t(s) * big::TAB], q);
                fill_row<8, 64>(&b->h_small[c * row_small + size_t(s) * small::TAB], q);
            }
        }
        if (n_dirty > 0)
        {
            if (n_dirty * 4 >= b->channels)     // mostly dirty: one transfer per table
            {
                MI_HIP_CHECK(hipMemcpyAsync(b->d_big, b->h_big.data(), b->h_big.size() * sizeof(float),
                                            hipMemcpyHostToDevice, st));
                MI_HIP_CHECK(hipMemcpyAsync(b->d_small, b->h_small.data(), b->h_small.size() * sizeof(float),
                                            hipMemcpyHostToDevice, st));
            }
            else
            {
                for (uint32_t c = 0; c < b->channels; ++c)
                {
                    if (!b->dirty[c] || b->nsec[c] == 0)
                        continue;
                    MI_HIP_CHECK(hipMemcpyAsync(b->d_big + c * row_big, &b->h_big[c * row_big],
                                                size_t(b->nsec[c]) * big::TAB * sizeof(float),
                                                hipMemcpyHostToDevice, st));
                    MI_HIP_CHECK(hipMemcpyAsync(b->d_small + c * row_small, &b->h_small[c * row_small],
                                                size_t(b->nsec[c]) * small::TAB * sizeof(float),
                                                hipMemcpyHostToDevice, st));
                }
            }
            MI_HIP_CHECK(hipMemcpyAsync(b->d_nsec, b->nsec.data(), b->channels * sizeof(uint32_t),
                                        hipMemcpyHostToDevice, st));
        }
        // delay memory clears (FilterBank.cpp:233-235)
        uint32_t c = 0;
        while (c < b->channels)
        {
            if (!b->clear[c]) { ++c; continue; }
            uint32_t e = c;
            while (e < b->channels && b->clear[e]) ++e;
            MI_HIP_CHECK(hipMemsetAsync(b->d_state + size_t(c) * b->max_sec * 2, 0,
                                        size_t(e - c) * b->max_sec * 2 * sizeof(float), st));
            c = e;
        }
        // pageable host memory: the runtime has consumed the sources when the calls return
        std::fill(b->dirty.begin(), b->dirty.end(), uint8_t(0));
        std::fill(b->clear.begin(), b->clear.end(), uint8_t(0));
        b->pending = false;
        return MI_OK;
    }
} // namespace

extern "C" {

int mi_biquad_bank_create(mi_biquad_bank_t **bank, uint32_t channels, uint32_t max_sections)
{
    MI_REQUIRE(bank != nullptr, MI_EINVAL, "mi_biquad_bank_create: NULL result pointer");
    *bank = nullptr;
    MI_REQUIRE(channels > 0, MI_EINVAL, "mi_biquad_bank_create: channels must be > 0");
    if (max_sections == 0)
        max_sections = 1;               // FilterBank::init(0) still allocates 3 banks (FilterBank.cpp:67)
    MI_REQUIRE(mi_dspu_device_count() > 0, MI_ENODEV, "no HIP device available (there is no CPU fallback)");

    mi_biquad_bank *b = new (std::nothrow) mi_biquad_bank();
    MI_REQUIRE(b != nullptr, MI_ENOMEM, "mi_biquad_bank_create: out of host memory");
    b->channels = channels;
    b->max_sec  = max_sections;
    const size_t cs = size_t(channels) * max_sections;
    try
    {
        b->nsec.assign(channels, 0);
        b->last_nsec.assign(channels, -1);
        b->coef.assign(cs * 5, 0.0f);
        b->dirty.assign(channels, 0);
        b->clear.assign(channels, 0);
        b->h_big.assign(cs * big::TAB, 0.0f);
        b->h_small.assign(cs * small::TAB, 0.0f);
    }
    catch (...)
    {
        delete b;
        return mi::fail(MI_ENOMEM, "mi_biquad_bank_create: out of host memory");
    }
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_big), cs * big::TAB * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_small), cs * small::TAB * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_state), cs * 2 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_backup), cs * 2 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_nsec), channels * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(b->d_state, 0, cs * 2 * sizeof(float));
    if (e == hipSuccess) e = hipMemset(b->d_nsec, 0, channels * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(b->d_big, 0, cs * big::TAB * sizeof(float));
    if (e == hipSuccess) e = hipMemset(b->d_small, 0, cs * small::TAB * sizeof(float));
    if (e != hipSuccess)
    {
        mi_biquad_bank_destroy(b);
        return mi::fail(e == hipErrorOutOfMemory ? MI_ENOMEM : MI_EHIP,
                        "mi_biquad_bank_create: %s", hipGetErrorString(e));
    }
    *bank = b;
    return MI_OK;
}

int mi_biquad_bank_destroy(mi_biquad_bank_t *b)
{
    if (b == nullptr)
        return MI_OK;
    (void)hipFree(b->d_big);
    (void)hipFree(b->d_small);
    (void)hipFree(b->d_state);
    (void)hipFree(b->d_backup);
    (void)hipFree(b->d_nsec);
    delete b;
    return MI_OK;
}

int mi_biquad_bank_set_chains(mi_biquad_bank_t *b, uint32_t channel,
                              const mi_biquad_x1_t *chains, uint32_t count, int clear)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_biquad_bank_set_chains: NULL bank");
    MI_REQUIRE(channel < b->channels, MI_EINVAL, "mi_biquad_bank_set_chains: channel %u out of range", channel);
    MI_REQUIRE(count == 0 || chains != nullptr, MI_EINVAL, "mi_biquad_bank_set_chains: NULL chains");
    float *dst = &b->coef[size_t(channel) * b->max_sec * 5];
    for (uint32_t i = 0; i < count; ++i)
    {
        // add_chain() beyond the capacity hands out the last slot again (FilterBank.cpp:94-99)
        const uint32_t slot = (i < b->max_sec) ? i : b->max_sec - 1;
        dst[slot * 5 + 0] = chains[i].b0;
        dst[slot * 5 + 1] = chains[i].b1;
        dst[slot * 5 + 2] = chains[i].b2;
        dst[slot * 5 + 3] = chains[i].a1;
        dst[slot * 5 + 4] = chains[i].a2;
    }
    const uint32_t items = (count < b->max_sec) ? count : b->max_sec;
    b->nsec[channel]  = items;
    b->dirty[channel] = 1;
    if (clear || int64_t(items) != b->last_nsec[channel])     // FilterBank.cpp:233-235
        b->clear[channel] = 1;
    b->last_nsec[channel] = items;
    b->pending = true;
    return MI_OK;
}

int mi_biquad_bank_set_all_chains(mi_biquad_bank_t *b, const mi_biquad_x1_t *chains, uint32_t count, int clear)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_biquad_bank_set_all_chains: NULL bank");
    for (uint32_t c = 0; c < b->channels; ++c)
    {
        const int r = mi_biquad_bank_set_chains(b, c, chains + size_t(c) * count, count, clear);
        if (r != MI_OK)
            return r;
    }
    return MI_OK;
}

int mi_biquad_bank_size(const mi_biquad_bank_t *b, uint32_t channel, uint32_t *count)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_biquad_bank_size: NULL bank");
    MI_REQUIRE(channel < b->channels && count != nullptr, MI_EINVAL, "mi_biquad_bank_size: bad argument");
    *count = b->nsec[channel];
    return MI_OK;
}

int mi_biquad_bank_commit(mi_biquad_bank_t *b, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_biquad_bank_commit: NULL bank");
    return commit(b, mi::as_stream(stream));
}

int mi_biquad_bank_reset(mi_biquad_bank_t *b, uint32_t channel, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_biquad_bank_reset: NULL bank");
    if (channel == UINT32_MAX)
        std::fill(b->clear.begin(), b->clear.end(), uint8_t(1));
    else
    {
        MI_REQUIRE(channel < b->channels, MI_EINVAL, "mi_biquad_bank_reset: channel %u out of range", channel);
        b->clear[channel] = 1;
    }
    b->pending = true;
    return commit(b, mi::as_stream(stream));
}

int mi_biquad_bank_process(mi_biquad_bank_t *b, float *out, const float *in, size_t samples,
                           size_t out_stride, size_t in_stride, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_biquad_bank_process: NULL bank");
    if (samples == 0)
        return MI_OK;
    MI_REQUIRE(out != nullptr && in != nullptr, MI_EINVAL, "mi_biquad_bank_process: NULL buffer");
    MI_REQUIRE(out_stride >= samples && in_stride >= samples, MI_EINVAL,
               "mi_biquad_bank_process: stride shorter than the block");
    hipStream_t st = mi::as_stream(stream);
    int r = commit(b, st);
    if (r != MI_OK)
        return r;

    const bool aligned = ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(in)) % 16 == 0) &&
                         (out_stride % 4 == 0) && (in_stride % 4 == 0);
    size_t done = 0;
    while (done < samples)
    {
        const size_t left = samples - done;
        hipError_t e;
        size_t step;
        if (left > size_t(small::BLOCK))
        {
            step = (left < size_t(big::BLOCK)) ? left : size_t(big::BLOCK);
            static const bool two_wave = (getenv("MI_BIQUAD_TWO_WAVE") != nullptr);   // A/B knob for profiling
            e = two_wave ? launch<32, 128>(b, out + done, in + done, out_stride, in_stride, int(step),
                                           aligned && (done % 4 == 0), b->d_big, st)
                         : launch_pk<32>(b, out + done, in + done, out_stride, in_stride, int(step),
                                         aligned && (done % 4 == 0), b->d_big, st);
        }
        else
        {
            step = left;
            e = launch<8, 64>(b, out + done, in + done, out_stride, in_stride, int(step),
                              aligned && (done % 4 == 0), b->d_small, st);
        }
        MI_HIP_CHECK(e);
        done += step;
    }
    return MI_OK;
}

int mi_biquad_bank_impulse_response(mi_biquad_bank_t *b, float *out, size_t samples, size_t out_stride, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_biquad_bank_impulse_response: NULL bank");
    if (samples == 0)
        return MI_OK;
    MI_REQUIRE(out != nullptr && out_stride >= samples, MI_EINVAL, "mi_biquad_bank_impulse_response: bad buffer");
    hipStream_t st = mi::as_stream(stream);
    int r = commit(b, st);
    if (r != MI_OK)
        return r;
    const size_t bytes = size_t(b->channels) * b->max_sec * 2 * sizeof(float);
    MI_HIP_CHECK(hipMemcpyAsync(b->d_backup, b->d_state, bytes, hipMemcpyDeviceToDevice, st));
    MI_HIP_CHECK(hipMemsetAsync(b->d_state, 0, bytes, st));
    hipLaunchKernelGGL(impulse_kernel, dim3(1024), dim3(256), 0, st, out, out_stride, samples, b->channels);
    MI_HIP_CHECK(hipGetLastError());
    r = mi_biquad_bank_process(b, out, out, samples, out_stride, out_stride, stream);
    if (r != MI_OK)
        return r;
    MI_HIP_CHECK(hipMemcpyAsync(b->d_state, b->d_backup, bytes, hipMemcpyDeviceToDevice, st));
    return MI_OK;
}

int mi_biquad_section_tables(const mi_biquad_x1_t *chain, int variant, float *table, uint32_t *geometry)
{
    MI_REQUIRE(chain != nullptr && geometry != nullptr, MI_EINVAL, "mi_biquad_section_tables: bad argument");
    MI_REQUIRE(variant == 0 || variant == 1, MI_EINVAL, "mi_biquad_section_tables: variant must be 0 or 1");
    const float q[5] = { chain->b0, chain->b1, chain->b2, chain->a1, chain->a2 };
    if (variant == 0)
    {
        geometry[0] = 32; geometry[1] = 128; geometry[2] = 16; geometry[3] = big::TAB;
        if (table != nullptr)
            fill_row<32, 128>(table, q);
    }
    else
    {
        geometry[0] = 8; geometry[1] = 64; geometry[2] = 16; geometry[3] = small::TAB;
        if (table != nullptr)
            fill_row<8, 64>(table, q);
    }
    return MI_OK;
}

int mi_biquad_bank_get_state(mi_biquad_bank_t *b, float *host_state, void *stream)
{
    MI_REQUIRE(b != nullptr && host_state != nullptr, MI_EINVAL, "mi_biquad_bank_get_state: bad argument");
    hipStream_t st = mi::as_stream(stream);
    int r = commit(b, st);
    if (r != MI_OK)
        return r;
    MI_HIP_CHECK(hipMemcpyAsync(host_state, b->d_state, size_t(b->channels) * b->max_sec * 2 * sizeof(float),
                                hipMemcpyDeviceToHost, st));
    MI_HIP_CHECK(hipStreamSynchronize(st));
    return MI_OK;
}

int mi_biquad_bank_set_state(mi_biquad_bank_t *b, const float *host_state, void *stream)
{
    MI_REQUIRE(b != nullptr && host_state != nullptr, MI_EINVAL, "mi_biquad_bank_set_state: bad argument");
    hipStream_t st = mi::as_stream(stream);
    int r = commit(b, st);
    if (r != MI_OK)
        return r;
    MI_HIP_CHECK(hipMemcpyAsync(b->d_state, host_state, size_t(b->channels) * b->max_sec * 2 * sizeof(float),
                                hipMemcpyHostToDevice, st));
    MI_HIP_CHECK(hipStreamSynchronize(st));
    return MI_OK;
}

} // extern "C"
