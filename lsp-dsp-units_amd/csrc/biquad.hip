// Biquad cascade bank for gfx950: the GPU side of lsp::dspu::FilterBank::process
// (reference: src/main/filters/FilterBank.cpp:256-291, which calls
// dsp::biquad_process_x8/x4/x2/x1 of lsp-dsp-lib once per packed bank).
//
// Why this is not "one channel per lane, serial in time": a block of N samples
// through one section is a chain of 2N dependent FMAs; at N = 4096 that chain
// alone is longer than the whole HBM budget of the block.  The kernel therefore
// makes the recurrence parallel in time.  Per section, state s = {d0,d1},
// s' = A s + B x, and every lane owns a PAIR of adjacent chunks of L samples:
//
//   1. zero-state end state of each chunk:  (z,w) = sum_k A^(L-1-k) B x[k]
//      -> dot products with the per-section table (p[k],q[k]);
//   2. the pair's end state for a zero start:  e = P zwA + zwB,  P = A^L;
//      the state carried in from the samples before the wave enters at lane 0
//      (e_0 += P^2 c); then an inclusive scan over the 64 pairs of the wave,
//      E_t = P^2 E_(t-1) + e_t, entirely with DPP:
//        a. inside each row of 16 lanes: row_shr 1,2,4,8 with P^2, P^4, P^8, P^16;
//        b. row_bcast:15 into rows 1 and 3 with the lane's own (P^2)^(i+1),
//           i = lane % 16 (16-entry per-lane table);
//        c. row_bcast:31 into rows 2 and 3 (row 3 through one more P^32);
//   3. start state of the lane's first chunk = E_(t-1) (wave_shr:1), of its
//      second chunk P S + zwA; then the EXACT TDF-II recurrence of the reference
//      over both chunks at once, as the two halves of v_pk_fma_f32 operands:
//         y = b0 x + d0;  d0 = (b1 x + d1) + a1 y;  d1 = b2 x + a2 y
//      Only the chunk start states carry the (float32 round-off sized)
//      difference of steps 1-2.
//
// A wave owns a sub-block of 64 x 2L samples of one channel; NW waves of a
// workgroup cover NW consecutive sub-blocks (a super-block) and hand the state
// from wave to wave through LDS, one workgroup barrier per section.  With 1024
// channels x 4096 samples that is 2 waves on every SIMD of the chip.
// All sections run back to back on samples held in registers (HBM sees each
// sample once in, once out).  The uniform part of a section's table lives in
// SGPRs (s_load, no LDS staging, no VGPRs); the loads of section s+1 are issued
// while section s is still computing.  Longer calls walk super-block after
// super-block; the loads of the next one are in flight during the sections and
// the write-through (sc1) stores drain behind them.  Loads are coalesced 16-B
// rows, transposed through a padded LDS tile private to the wave (pitch 2L+4
// dwords, an odd number of 16-B slots: conflict-free ds_read_b128).
#include "mi_common.h"
#include "ilufs_device.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace
{
    template <int L>
    struct geom
    {
        static constexpr int W      = 2 * L;                // samples per lane and sub-block
        static constexpr int TAB    = 96 + 2 * L;           // floats per (channel, section)
        static constexpr int PITCH  = W + 4;                // LDS dwords per lane
        static constexpr int BLOCK  = 64 * W;               // samples per sub-block
        static constexpr int SG     = 128;                  // sections whose carried state is kept in LDS at once
        static_assert(((PITCH / 4) & 1) == 1, "LDS pitch must be an odd number of 16-B slots");
        static_assert((TAB % 16) == 0, "table rows stay 64-B aligned (s_load_dwordx16)");
    };

    // Table row of one section (matrices column-major: m00 m10 m01 m11, so that a column is a register pair):
    //   [0..7]     b0 b1 b2 a1 a2 0 0 0
    //   [8..15]    P = A^L, P^2
    //   [16..31]   P^4, P^8, P^16, P^32
    //   [32..]     (p[k], q[k]) k < L : weights of sample k in the chunk's zero-state end state
    //   [32+2L..]  (P^2)^(i+1), i = 0..15 (per-lane operand of the scan)
    constexpr int TAB_PQ = 32;
    typedef float v2f  __attribute__((ext_vector_type(2)));
    typedef float v8f  __attribute__((ext_vector_type(8)));
    typedef float v16f __attribute__((ext_vector_type(16)));

    __device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
    __device__ __forceinline__ v2f splat(float a) { return v2f{a, a}; }
    // a + M v for a column-major 2x2 matrix given as its two columns
    __device__ __forceinline__ v2f mat_fma(v2f c0, v2f c1, v2f v, v2f a) { return pk_fma(c0, splat(v.x), pk_fma(c1, splat(v.y), a)); }

    template <int CTRL, int ROW_MASK>
    __device__ __forceinline__ float dpp_or(float old, float v)      // lanes without a source / masked rows keep `old`
    {
        return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
    }
    template <int CTRL, int ROW_MASK>
    __device__ __forceinline__ v2f dpp_or(v2f old, v2f v) { return v2f{dpp_or<CTRL, ROW_MASK>(old.x, v.x), dpp_or<CTRL, ROW_MASK>(old.y, v.y)}; }

    template <int CTRL>
    __device__ __forceinline__ v2f dpp_zero(v2f v)                    // lanes without a source read 0 (bound_ctrl)
    {
        return v2f{__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.x), CTRL, 0xf, 0xf, true)),
                   __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.y), CTRL, 0xf, 0xf, true))};
    }

    constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
    constexpr int DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

    using mi::BUFFER_DWORD3;
    using mi::CPOL_SC1;
    using mi::u32x4;
    using mi::CPOL_NT_SC1;
    template <int POL = CPOL_NT_SC1>
    __device__ __forceinline__ void store_through(__amdgpu_buffer_rsrc_t rsrc, int dword_index, float4 v)
    {
        mi::wt_store<POL>(rsrc, dword_index * 4, v);
    }

#ifndef MI_ABLATE
#define MI_ABLATE 0                 // timing experiments only (tests/experiments/biquad_phase_probe.hip): drop one phase
#endif
#ifdef MI_BIQUAD_PROBE
    // phase timestamps of lane 0 (tests/experiments/biquad_phase_probe.hip): [block][slot] = {100 MHz wall clock, shader cycles}
    __device__ unsigned long long g_probe[4096 * 16 * 2];
    #define MI_PROBE(slot) do { __builtin_amdgcn_sched_barrier(0); if (t == 0 && (slot) < 16) { \
        const unsigned pw_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); \
        g_probe[(pw_ * 16 + (slot)) * 2] = wall_clock64(); \
        g_probe[(pw_ * 16 + (slot)) * 2 + 1] = __builtin_readcyclecounter(); } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
    #define MI_PROBE(slot) do { } while (0)
#endif
    // A chain of cascades on one block held in registers (library-internal: the Crossover's split points, and any
    // caller of several banks on the same samples): stage k runs bank k's sections either IN PLACE on the travelling
    // signal or on a BRANCH of it (the travelling signal goes on unchanged), and writes its result to `out` if there is
    // one.  Crossover.cpp:451-498 is  band k = LPF_k(src) [branch],  src = HPF_k(src) [in place]  per split point: one
    // read of the source and one write per band instead of a read and a write per filter.
    constexpr int CHAIN_MAX = 14;
    struct chain_stage
    {
        const float    *tab;            // the bank's table of this launch variant
        float          *state;
        const uint32_t *nsec;
        float          *out;            // NULL: nothing is written (an in-place stage in the middle of the chain)
        size_t          out_stride;
        int             max_sec;
        int             branch;
    };
    struct chain_args
    {
        int         stages;
        chain_stage st[CHAIN_MAX];
    };

    // Epilogue of the meters (library-internal): the filtered samples are not stored; the sum of their squares is added
    // to sums[channel * 4 + s] for the up to four consecutive segments [0, e0), [e0, e1), [e1, e2), [e2, n) of the call
    // (ILUFSMeter.cpp:372-384 accumulates the squares of the weighted signal per quarter of a gating block).
    struct sumsq_args
    {
        float      *sums;
        int         e0, e1, e2;         // segment ends, e0 <= e1 <= e2 <= n; unused ones = n
    };

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "biquad.hip relies on gfx950 (CDNA4) behaviour: wave64 DPP row/bank semantics and arrival-counting s_barrier"
#endif
    // ROWS > 1 (SUMSQ only): the workgroup runs ROWS consecutive rows side by side, NW waves each -- for the integrated meter,
    // whose rows then leave their sums of squares in LDS (sq.sums points there) for the same workgroup's bookkeeping.  The
    // barriers are the workgroup's: the rows must have the same number of sections and none of them may be switched off.
    template <int L, int NW, bool ALIGNED, bool CHAIN, bool SUMSQ = false, int ROWS = 1>
    __device__ __forceinline__
    void biquad_body(float *out, const float *in, size_t out_stride, size_t in_stride,
                     int n /* multiple of L */, const float *__restrict__ tab, float *state,
                     const uint32_t *__restrict__ nsec, int max_sec, const chain_args &chain,
                     const sumsq_args &sq = sumsq_args(), const bool sums_local = false /* sq.sums: [ROWS][4] in LDS */,
                     const bool out_reread = false /* the next launch reads `out`: plain write-through, not non-temporal */)
    {
        using G = geom<L>;
        constexpr int W = G::W, TAB = G::TAB, PITCH = G::PITCH, SB = G::BLOCK, SG = G::SG;
        constexpr int LPT = W / 4;                          // float4 per lane and sub-block
        constexpr int NT = 64 * NW;
        constexpr int XW = NW;                              // waves that share one section of one super-block
        constexpr int TAB_QL = TAB_PQ + 2 * L;

        static_assert(ROWS == 1 || (SUMSQ && !CHAIN), "rows side by side: the meters' form only");
        __shared__ __attribute__((aligned(16))) float sx_rows[ROWS][NW * 64 * PITCH];
        __shared__ float2 sstate_rows[ROWS][2][SG];         // state carried between super-blocks, by parity
        __shared__ float2 xchg_rows[ROWS][2][SG][NW];   // end state of every wave's sub-block
        // (one row per workgroup: `slot` is the constant 0 and everything below is what it was)
        const int slot = (ROWS > 1) ? __builtin_amdgcn_readfirstlane(int(threadIdx.x) / NT) : 0;
        float *const sx_all = sx_rows[slot];
        float2 (*const sstate)[SG] = sstate_rows[slot];
        float2 (*const xchg)[SG][NW] = xchg_rows[slot];

        const int ch   = int(blockIdx.x) * ROWS + slot;
        const int tid  = int(threadIdx.x) - slot * NT;
        const int t    = tid & 63;                          // lane
        const int wv   = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int l16  = t & 15;
        // stage accessors: the one bank of the plain kernel, or stage k of the chain (all wave-uniform)
        const int nst  = CHAIN ? chain.stages : 1;
        auto stage_ns   = [&](int k) -> int { const int v = CHAIN ? int(chain.st[k].nsec[ch]) : int(nsec[ch]); return v; };
        auto stage_tab  = [&](int k) -> const float * {
            return CHAIN ? chain.st[k].tab + size_t(ch) * chain.st[k].max_sec * TAB : tab + size_t(ch) * max_sec * TAB; };
        auto stage_mem  = [&](int k) -> float * {
            return CHAIN ? chain.st[k].state + size_t(ch) * chain.st[k].max_sec * 2 : state + size_t(ch) * max_sec * 2; };
        const int ns   = stage_ns(0);
        if (!CHAIN && ns < 0)                               // row switched off: state kept, output not written
            return;
        float *sx = sx_all + wv * 64 * PITCH;               // this wave's private transpose tile
        const bool lane0 = (t == 0), row3 = (t >= 48);
        const float *ctab = stage_tab(0);                   // this channel's table rows (uniform address)
        // Buffer descriptors over the channel's n samples: reads past the end return 0, writes past the end are
        // dropped, so the tile rows need no bounds branches and the compiler's vmcnt bookkeeping stays exact.
        const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(in + size_t(ch) * in_stride), 0, n * 4, BUFFER_DWORD3);
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
            out + size_t(ch) * out_stride, 0, n * 4, BUFFER_DWORD3);
        MI_PROBE(0);
        // The two waves that share a SIMD belong to workgroups 256 apart in dispatch order (census of a 1024 x 128-thread
        // launch on this part, tests/experiments/census.hip).  The earlier one gets issue priority: it reaches its stores
        // while its partner is still in the sections, so the write-back of one overlaps the arithmetic of the other
        // (about 0.8 us of a 13.9 us launch at 1024 channels; placement is never relied on for correctness).
        if (((blockIdx.x >> 8) & 1) == 0)
            __builtin_amdgcn_s_setprio(1);

        // ---- helpers ---------------------------------------------------------------------------
        float4 ld[LPT];
        auto issue_loads = [&](int base)                    // coalesced rows of the wave's sub-block -> registers
        {
            #pragma unroll
            for (int k = 0; k < LPT; ++k)
            {
                const int o = (base + 4 * (k * 64 + t)) * 4;
                if (ALIGNED)
                {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(irsrc, o, 0, 0);
                    ld[k] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                }
                else
                    ld[k] = make_float4(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(irsrc, o, 0, 0)),
                                        __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(irsrc, o + 4, 0, 0)),
                                        __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(irsrc, o + 8, 0, 0)),
                                        __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(irsrc, o + 12, 0, 0)));
            }
        };
        auto load_state = [&](float *mem, int s0, int group, int par, int lds0)   // carried state of a section group -> LDS
        {
            for (int i = tid; i < group; i += NT)
                sstate[par][lds0 + i] = reinterpret_cast<const float2 *>(mem + size_t(s0) * 2)[i];
        };
        auto flush_state = [&](float *mem, int s0, int group, int par, int lds0)
        {
            for (int i = tid; i < group; i += NT)
                reinterpret_cast<float2 *>(mem + size_t(s0) * 2)[i] = sstate[par][lds0 + i];
        };

        v2f x[L];                                            // .x: first chunk of the lane, .y: second chunk

        // One section's table.  Everything but `ql` is wave-uniform and sits in SGPRs (scalar loads); `ql` is the lane's
        // own (P^2)^(lane%16+1).  The three parts are reloaded for the next section as soon as this section is done with
        // them, so the loads are in flight underneath the rest of the section.
        struct sectab
        {
            v16f pq[L / 8];                                  // (p[k], q[k]) pairs
            v8f  m0;                                         // P, P^2
            v16f m1;                                         // P^4, P^8, P^16, P^32
            float4 cf;                                       // b0 b1 b2 a1   (no dead lanes in any scalar load: the
            float  a2;                                       //  compiler would recycle them and stall on the load)
            float4 ql;
        };
        // The wave-uniform parts are read through the CONSTANT address space from an address made uniform with
        // readfirstlane: scalar loads into SGPRs.  The plain kernel got those anyway (its table is a __restrict__ kernel
        // argument); the chain kernel, whose table pointers come out of the argument struct by a loop index, did not -- the
        // compiler could neither prove the address uniform nor the rows unclobbered by the band stores, and fetched every
        // row with vector loads into VGPRs: 29 vector-memory instructions and about 100 extra VALU per section (SQ
        // counters: 388 VMEM and 3432 VALU per wave for 12 sections).
        typedef const __attribute__((address_space(4))) float cfloat;
        auto uniform_row = [&](const float *T) -> cfloat * {
            const uint64_t v = reinterpret_cast<uint64_t>(T);
            const uint32_t lo = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v))));
            const uint32_t hi = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v >> 32))));
            return reinterpret_cast<cfloat *>((uint64_t(hi) << 32) | lo);
        };
        typedef const __attribute__((address_space(4))) v16f cv16f;
        typedef const __attribute__((address_space(4))) v8f cv8f;
        auto load_pq = [&](sectab &r, const float *T)
        {
            cfloat *U = uniform_row(T);
            #pragma unroll
            for (int j = 0; j < L / 8; ++j)
                r.pq[j] = *reinterpret_cast<cv16f *>(U + TAB_PQ + 16 * j);
        };
        auto load_mats = [&](sectab &r, const float *T)
        {
            cfloat *U = uniform_row(T);
            r.m0 = *reinterpret_cast<cv8f *>(U + 8);
            r.m1 = *reinterpret_cast<cv16f *>(U + 16);
            r.ql = *reinterpret_cast<const float4 *>(T + TAB_QL + 4 * l16);
        };
        auto load_coefs = [&](sectab &r, const float *T)
        {
            cfloat *U = uniform_row(T);
            r.cf = make_float4(U[0], U[1], U[2], U[3]);
            r.a2 = U[4];
        };

        // One section over the lane's two chunks.  par: parity of the super-block.  `saver` marks the lane holding
        // the last sample of the super-block (in its first chunk if !save_hi) -- it saves the state for what follows.
        // `tb` holds this section's table on entry and the table at `Tnext` on exit.
        auto section = [&](sectab &tb, const float *Tnext, int si, int par, bool saver, bool save_hi)
        {
            // 1. zero-state end states of both chunks: zwA = (z, w) of the first, zwB of the second
            v2f a0 = splat(0.0f), a1_ = splat(0.0f), b0_ = splat(0.0f), b1_ = splat(0.0f);
            #pragma unroll
            for (int k = 0; k < L; k += 2)
            {
                const v16f &r = tb.pq[k / 8];
                const v2f pq0 = v2f{r[(2 * k) % 16], r[(2 * k + 1) % 16]};
                const v2f pq1 = v2f{r[(2 * k + 2) % 16], r[(2 * k + 3) % 16]};
                a0  = pk_fma(pq0, splat(x[k].x), a0);      b0_ = pk_fma(pq0, splat(x[k].y), b0_);
                a1_ = pk_fma(pq1, splat(x[k + 1].x), a1_); b1_ = pk_fma(pq1, splat(x[k + 1].y), b1_);
            }
            const v2f zwA = a0 + a1_, zwB = b0_ + b1_;

            // 2. end state of the pair for a zero start
            const v2f Pc0 = v2f{tb.m0[0], tb.m0[1]}, Pc1 = v2f{tb.m0[2], tb.m0[3]};
            v2f e = mat_fma(Pc0, Pc1, zwA, zwB);

            // state entering this wave's sub-block: carried over for wave 0, else the end state of the wave before.
            // Hand-off by counted barriers: wave w passes w barriers before it reads its predecessor's end state and
            // NW - 1 - w after it has published its own, i.e. every wave executes exactly NW - 1 s_barrier instructions
            // per section, but at different program points.  That is legal on gfx9 / CDNA (gfx950 is the only target of
            // this file, see the guard above the kernel): s_barrier counts ARRIVALS of the workgroup's waves, whatever
            // their program counter is, and the compiler is told nothing else (no convergent-region assumption is made
            // across the loop: the trip counts are wave-uniform).  On an architecture with split or named barriers this
            // hand-off has to be rewritten (NW - 1 uniform barrier sites with predicated work).
            if (XW > 1)
                for (int v = 0; v < wv; ++v)
                    __syncthreads();
            const float2 cs = (XW > 1 && wv > 0) ? xchg[par][si][wv - 1] : sstate[par][si];
            const v2f cvec = lane0 ? v2f{cs.x, cs.y} : splat(0.0f);
            e = mat_fma(v2f{tb.m0[4], tb.m0[5]}, v2f{tb.m0[6], tb.m0[7]}, cvec, e);

            // 2a. inclusive scan over the pairs inside rows of 16 lanes: E += (P^2)^d E(lane - d)
            const v2f zero = splat(0.0f);
            e = mat_fma(v2f{tb.m0[4], tb.m0[5]}, v2f{tb.m0[6], tb.m0[7]}, dpp_zero<DPP_ROW_SHR1>(e), e);
            e = mat_fma(v2f{tb.m1[0], tb.m1[1]}, v2f{tb.m1[2], tb.m1[3]}, dpp_zero<DPP_ROW_SHR2>(e), e);
            e = mat_fma(v2f{tb.m1[4], tb.m1[5]}, v2f{tb.m1[6], tb.m1[7]}, dpp_zero<DPP_ROW_SHR4>(e), e);
            e = mat_fma(v2f{tb.m1[8], tb.m1[9]}, v2f{tb.m1[10], tb.m1[11]}, dpp_zero<DPP_ROW_SHR8>(e), e);
            // 2b. rows 1 and 3 take in the row before them
            const v2f QLc0 = v2f{tb.ql.x, tb.ql.y}, QLc1 = v2f{tb.ql.z, tb.ql.w};
            e = mat_fma(QLc0, QLc1, dpp_or<DPP_ROW_BCAST15, 0xa>(zero, e), e);
            // 2c. rows 2 and 3 take in rows 0-1 (lane 31), row 3 across one more row of 16 pairs
            {
                const v2f s  = dpp_or<DPP_ROW_BCAST31, 0xc>(zero, e);
                const v2f s2 = mat_fma(v2f{tb.m1[12], tb.m1[13]}, v2f{tb.m1[14], tb.m1[15]}, s, zero);
                e = mat_fma(QLc0, QLc1, row3 ? s2 : s, e);
            }
            if (XW > 1)
            {
                if (t == 63)
                    xchg[par][si][wv] = make_float2(e.x, e.y);
                for (int v = wv; v < XW - 1; ++v)
                    __syncthreads();
            }

            // 3. start states: first chunk = end of the previous pair, second chunk = P S + zwA
            const v2f S  = dpp_or<DPP_WAVE_SHR1, 0xf>(cvec, e);
            const v2f SB = mat_fma(Pc0, Pc1, S, zwA);
            v2f d0 = v2f{S.x, SB.x}, d1 = v2f{S.y, SB.y};
            __builtin_amdgcn_sched_barrier(0);              // the next section's table streams in underneath the recurrence,
            load_pq(tb, Tnext);                             // into the registers this section no longer needs
            load_mats(tb, Tnext);
            __builtin_amdgcn_sched_barrier(0);

            // exact recurrence over both chunks
            const v2f b0 = splat(tb.cf.x), b1 = splat(tb.cf.y), b2 = splat(tb.cf.z), a1 = splat(tb.cf.w), a2 = splat(tb.a2);
            #pragma unroll
            for (int k = 0; k < L; ++k)
            {
                const v2f xx = x[k];
                const v2f tq = pk_fma(b1, xx, d1);
                const v2f u  = b2 * xx;
                const v2f y  = pk_fma(b0, xx, d0);
                d0   = pk_fma(a1, y, tq);
                d1   = pk_fma(a2, y, u);
                x[k] = y;
            }
            __builtin_amdgcn_sched_barrier(0);
            load_coefs(tb, Tnext);
            if (saver)                                       // n is a multiple of L: the call ends with a chunk
                sstate[par ^ 1][si] = save_hi ? make_float2(d0.y, d1.y) : make_float2(d0.x, d1.x);
        };

        // ---- super-blocks of NW sub-blocks --------------------------------------------------------
        constexpr int SUPER = NW * SB;
        const int nsup = (n + SUPER - 1) / SUPER;
        const int off  = wv * SB;                            // this wave's sub-block inside the super-block
        sectab tb;

        // x (registers) -> the wave's tile (transposed) -> coalesced write-through store at `base` of a block's output
        auto store_block = [&](const __amdgpu_buffer_rsrc_t &dst, int base)
        {
            #pragma unroll
            for (int k = 0; k < L / 4; ++k)
            {
                *reinterpret_cast<float4 *>(&sx[t * PITCH + 4 * k]) =
                    make_float4(x[4 * k + 0].x, x[4 * k + 1].x, x[4 * k + 2].x, x[4 * k + 3].x);
                *reinterpret_cast<float4 *>(&sx[t * PITCH + L + 4 * k]) =
                    make_float4(x[4 * k + 0].y, x[4 * k + 1].y, x[4 * k + 2].y, x[4 * k + 3].y);
            }
            __builtin_amdgcn_wave_barrier();
            #pragma unroll
            for (int k = 0; k < LPT; ++k)
            {
                const int i = 4 * (k * 64 + t);
                const float4 v = *reinterpret_cast<const float4 *>(&sx[i + (i / W) * 4]);
                constexpr int POL = SUMSQ ? CPOL_SC1 : CPOL_NT_SC1;
                if (ALIGNED && out_reread)
                    store_through<CPOL_SC1>(dst, base + i, v);
                else if (ALIGNED)
                    store_through<POL>(dst, base + i, v);
                else
                {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.x), dst, (base + i) * 4, 0, POL);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.y), dst, (base + i) * 4 + 4, 0, POL);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.z), dst, (base + i) * 4 + 8, 0, POL);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.w), dst, (base + i) * 4 + 12, 0, POL);
                }
            }
            __builtin_amdgcn_wave_barrier();
        };

        // SUMSQ: x (registers) -> the segments' sums of squares.  A chunk lies inside one segment unless one of the (at most
        // three) segment ends of the call falls into it: only those lanes walk their chunk sample by sample.
        float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        auto seg_of = [&](int i) -> int { return int(i >= sq.e0) + int(i >= sq.e1) + int(i >= sq.e2); };
        auto add_chunk = [&](float q, int c0, bool hi)
        {
            if (c0 >= n)                                    // a chunk past the end of the call (its loads returned zeros)
                return;
            const int sa = seg_of(c0), sb = seg_of(c0 + L - 1);
            if (sa == sb)
            {
                #pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] += (sa == j) ? q : 0.0f;
                return;
            }
            #pragma unroll
            for (int k = 0; k < L; ++k)
            {
                const float y = hi ? x[k].y : x[k].x;
                const int sk = seg_of(c0 + k);
                #pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = (sk == j) ? fmaf(y, y, acc[j]) : acc[j];
            }
        };
        auto add_squares = [&](int base)
        {
            v2f q = splat(0.0f);
            #pragma unroll
            for (int k = 0; k < L; ++k)
                q = pk_fma(x[k], x[k], q);
            add_chunk(q.x, base + t * W, false);
            add_chunk(q.y, base + t * W + L, true);
        };


        // (a two-role form -- the sections split between the two waves, the tiles pipelined -- was built and measured in round 3:
        // 15.3 against 12.7 us, profiles/r03_experiments/biquad_two_roles.txt; removed in round 6)


        if (!CHAIN)
        {
            if (ns <= SG)
                load_state(stage_mem(0), 0, ns, 0, 0);
            if (ns > 0)                                     // scalar loads: they do not queue behind the samples
            {
                load_pq(tb, ctab);
                load_mats(tb, ctab);
                load_coefs(tb, ctab);
            }
            issue_loads(off);
            if (ns <= SG)
                __syncthreads();
        }
        else
        {
            // every stage's carried state sits in LDS for the whole launch (the host sends only chains whose sections
            // fit: sum of the stages' section counts <= SG, every stage with at least one section)
            int lds0 = 0;
            for (int k = 0; k < nst; ++k)
            {
                const int nk = stage_ns(k);
                load_state(stage_mem(k), 0, nk, 0, lds0);
                lds0 += nk;
            }
            load_pq(tb, ctab);
            load_mats(tb, ctab);
            load_coefs(tb, ctab);
            issue_loads(off);
            __syncthreads();
        }

        for (int sp = 0; sp < nsup; ++sp)
        {
            const int par  = sp & 1;
            const int base = sp * SUPER + off;
            const int left = n - sp * SUPER;                 // samples of the call from this super-block on

            // registers -> own LDS tile (transposed), then immediately start the loads of the next super-block.
            // The tile is private to the wave and a wave's LDS accesses complete in program order: no barrier.
            #pragma unroll
            for (int k = 0; k < LPT; ++k)
            {
                const int i = 4 * (k * 64 + t);
                *reinterpret_cast<float4 *>(&sx[i + (i / W) * 4]) = ld[k];
            }
            __builtin_amdgcn_wave_barrier();
            MI_PROBE(1 + 4 * sp);
            if (sp + 1 < nsup)
                issue_loads(base + SUPER);
            #pragma unroll
            for (int k = 0; k < L / 4; ++k)
            {
                const float4 a = *reinterpret_cast<const float4 *>(&sx[t * PITCH + 4 * k]);
                const float4 b = *reinterpret_cast<const float4 *>(&sx[t * PITCH + L + 4 * k]);
                x[4 * k + 0] = v2f{a.x, b.x}; x[4 * k + 1] = v2f{a.y, b.y};
                x[4 * k + 2] = v2f{a.z, b.z}; x[4 * k + 3] = v2f{a.w, b.w};
            }

            // the lane that owns the last sample of the super-block
            const int last   = ((left < SUPER) ? left : SUPER) - 1;
            const int w_last = last / SB;
            const int t_last = (last - w_last * SB) / W;
            const bool save_hi = (last - w_last * SB - t_last * W) >= L;
            const bool saver   = (wv == w_last) && (t == t_last);
            if (!CHAIN)
            {
                for (int s0 = 0; s0 < ns; s0 += SG)
                {
                    const int group = (ns - s0 < SG) ? (ns - s0) : SG;
                    if (ns > SG)                             // more sections than state slots: one group at a time
                    {
                        __syncthreads();
                        load_state(stage_mem(0), s0, group, par, 0);
                        __syncthreads();
                    }
                    for (int si = 0; si < group; ++si)
                    {
                        const int snext = (s0 + si + 1 < ns) ? s0 + si + 1 : 0;
                        section(tb, ctab + size_t(snext) * TAB, si, par, saver, save_hi);
                    }
                    if (ns > SG)
                    {
                        __syncthreads();
                        flush_state(stage_mem(0), s0, group, par ^ 1, 0);
                    }
                }
                MI_PROBE(2 + 4 * sp);
                if (SUMSQ)
                    add_squares(base);
                else
                    store_block(orsrc, base);
            }
            else
            {
                // Every stage has at least one section for every channel (the host sends nothing else), so the table
                // that follows a stage's last section is simply the next stage's first one.  What a stage needs (its
                // section count, table, output) is fetched one stage ahead: those scalar loads then complete underneath
                // the sections of the stage before instead of stalling the wave at every stage boundary.
                struct stage_info { int ns; const float *T; float *out; size_t out_stride; int branch; };
                auto fetch = [&](int k) -> stage_info {
                    stage_info q;
                    q.ns = stage_ns(k);
                    q.T = stage_tab(k);
                    q.out = chain.st[k].out;
                    q.out_stride = chain.st[k].out_stride;
                    q.branch = chain.st[k].branch;
                    return q;
                };
                v2f xs[L];                                   // the travelling signal while a branch is computed
                int lds0 = 0;
                stage_info cur = fetch(0);
                for (int k = 0; k < nst; ++k)
                {
                    const stage_info nxt = fetch((k + 1 < nst) ? k + 1 : 0);
                    if (cur.branch)
                    {
                        #pragma unroll
                        for (int i = 0; i < L; ++i)
                            xs[i] = x[i];
                    }
                    for (int si = 0; si < cur.ns; ++si)
                        section(tb, (si + 1 < cur.ns) ? cur.T + size_t(si + 1) * TAB : nxt.T, lds0 + si, par, saver, save_hi);
                    lds0 += cur.ns;
                    if (cur.out != nullptr)
                    {
                        const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(
                            cur.out + size_t(ch) * cur.out_stride, 0, n * 4, BUFFER_DWORD3);
                        store_block(dst, base);
                    }
                    if (cur.branch)
                    {
                        #pragma unroll
                        for (int i = 0; i < L; ++i)
                            x[i] = xs[i];
                    }
                    cur = nxt;
                }
            }
            if (NW > 1)
                __syncthreads();                             // the saved state is visible before wave 0 reads it again
            MI_PROBE(3 + 4 * sp);
        }
        if (!CHAIN)
        {
            if (ns > 0 && ns <= SG)
            {
                __syncthreads();
                flush_state(stage_mem(0), 0, ns, nsup & 1, 0);
            }
        }
        else
        {
            __syncthreads();
            int lds0 = 0;
            for (int k = 0; k < nst; ++k)
            {
                const int nk = stage_ns(k);
                flush_state(stage_mem(k), 0, nk, nsup & 1, lds0);
                lds0 += nk;
            }
        }
        if (SUMSQ)
        {
            __shared__ float red_rows[ROWS][NW][4];
            float (*const red)[4] = red_rows[slot];
            #pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                float v = acc[j];
                #pragma unroll
                for (int d = 32; d > 0; d >>= 1)
                    v += __shfl_xor(v, d);
                if (lane0)
                    red[wv][j] = v;
            }
            __syncthreads();
            if (tid < 4)
            {
                float v = red[0][tid];
                for (int w = 1; w < NW; ++w)
                    v += red[w][tid];
                if (ROWS > 1 || sums_local)
                    sq.sums[slot * 4 + tid] = v;            // LDS of this workgroup: its bookkeeping follows
                else
                    // one addition per cell and launch (the order of the launches is the stream's): an atomic without a
                    // return value does not hold the workgroup up for the round trip a read-modify-write would
                    atomicAdd(&sq.sums[size_t(ch) * 4 + tid], v);
            }
        }
        MI_PROBE(15);
    }

    template <int L, int NW, bool ALIGNED>
    __global__ __launch_bounds__(64 * NW, (NW > 1) ? 2 : 1)
    void biquad_bank_kernel(float *out, const float *in, size_t out_stride, size_t in_stride,
                            int n /* multiple of L */, const float *__restrict__ tab, float *state,
                            const uint32_t *__restrict__ nsec, int max_sec, int out_reread)
    {
        biquad_body<L, NW, ALIGNED, false>(out, in, out_stride, in_stride, n, tab, state, nsec, max_sec, chain_args(), sumsq_args(), false,
                                           out_reread != 0);
    }

    template <int L, int NW, bool ALIGNED>
    __global__ __launch_bounds__(64 * NW, (NW > 1) ? 2 : 1)
    void biquad_sumsq_kernel(const float *in, size_t in_stride, int n /* multiple of L */, const float *__restrict__ tab,
                             float *state, const uint32_t *__restrict__ nsec, int max_sec, const sumsq_args sq)
    {
        biquad_body<L, NW, ALIGNED, false, true>(nullptr, in, 0, in_stride, n, tab, state, nsec, max_sec, chain_args(), sq);
    }

    // The same with the integrated loudness meter's bookkeeping riding on the launch (ilufs_device.h): a workgroup that has
    // left its row's sums of squares counts itself in at its meter; the one that finds all the other rows of the meter
    // already counted does what ilufs_call_kernel would do in a launch of its own.  The sums and the count are agent-scope
    // atomics, performed at the memory side; the adding threads wait for theirs (s_waitcnt vmcnt(0): an atomic without a
    // return value completes like a store) before the barrier behind which thread 0 counts the row in, and the reader
    // takes the sums with agent-scope loads.  No fence: an agent-scope release writes the XCD's L2 back, once per workgroup.
    // This ordering is gfx950 BEHAVIOUR, not the HIP memory model: memory-side atomics are acknowledged through vmcnt, and
    // relaxed agent-scope accesses of other workgroups see them in the order of those acknowledgements
    // (MI355X_MICROARCH.md, "valid hand-off forms").  The file refuses other targets (see the #error above), the written-out
    // wait in front of the count is checked in the compiler's output (tests/test_isa_checks.py), the two-launch form is
    // one environment variable away (MI_ILUFS_TWO_LAUNCHES) and compared bit for bit in tests/test_ilufs_gpu.py, and the
    // count is zeroed again by mi_ilufs_bank_clear.
    template <int L, int NW, bool ALIGNED>
    __global__ __launch_bounds__(64 * NW, (NW > 1) ? 2 : 1)
    void biquad_sumsq_ilufs_kernel(const float *in, size_t in_stride, int n /* multiple of L */, const float *__restrict__ tab,
                                   float *state, const uint32_t *__restrict__ nsec, int max_sec, const sumsq_args sq,
                                   const mi_meters::ilufs_epilogue ep)
    {
        biquad_body<L, NW, ALIGNED, false, true>(nullptr, in, 0, in_stride, n, tab, state, nsec, max_sec, chain_args(), sq);
        __shared__ uint32_t s_last;
        __shared__ float s_sum[4];
        __shared__ uint32_t s_cnt[4];
        __shared__ float s_val;
        __shared__ float s_chan[2 * 64 * NW];
        const uint32_t meter = blockIdx.x / ep.channels;
        // what the meter's bookkeeping reads of EARLIER calls is asked for by every row's workgroup, underneath the wait for
        // its own additions: the one that turns out to be the last has it in registers by then (ilufs_device.h)
        const mi_meters::ilufs_early<64 * NW> early = mi_meters::ilufs_ask<64 * NW>(meter, ep.block, ep.cfg, ep.channels, ep.st, ep.hist,
                                                                                   ep.size, ep.ms_int);
        // The output run of the call's FIRST piece is the value held since the last call -- known now, to every row of the
        // meter: each writes its share of the run here, in the shadow of the wait below, instead of the last one writing
        // all of it on the tail of the launch (16 KB per meter and 4096-sample call).
        const mi_meters::ilufs_piece first = ep.pieces.p[0];
        const bool fill_early = ep.out != nullptr && ep.pieces.count > 0 && first.n > 0;
        if (fill_early)
        {
            const uint32_t row = blockIdx.x - meter * ep.channels;
            const uint32_t share = (((first.n + ep.channels - 1) / ep.channels) + 3u) & ~3u;
            const uint32_t a = (row * share < first.n) ? row * share : first.n, b = (a + share < first.n) ? a + share : first.n;
            if (b > a)
                mi_meters::ilufs_fill<64 * NW>(ep.out + size_t(meter) * ep.out_stride + first.offset + a, b - a, early.me.loudness * ep.gain);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this row's additions are performed before it is counted in
        __syncthreads();
        if (threadIdx.x == 0)
        {
            const uint32_t before = __hip_atomic_fetch_add(&ep.arrived[meter], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (before + 1 == ep.channels) ? 1u : 0u;
            if (s_last)
                __hip_atomic_store(&ep.arrived[meter], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // for the next launch
        }
        __syncthreads();
        if (!s_last)
            return;
        mi_meters::ilufs_call_body<64 * NW, true>(meter, ep.block, sq.sums, ep.pieces, ep.cfg, ep.channels, ep.out, ep.out_stride,
                                            ep.st, ep.gain, ep.hist, ep.size, ep.ms_int, ep.avg, s_sum, s_cnt, s_val, s_chan, early, fill_early);
    }

    // The stereo meter in ONE workgroup: its two rows run the weighting filter side by side (biquad_body ROWS = 2), leave
    // their sums of squares in LDS, and the workgroup -- 256 threads, one per virtual thread of the bookkeeping -- goes
    // straight on with the meter's pieces.  No hand-over through memory at all: the riding form above pays three dependent
    // trips to the memory side on the tail of its launch (additions performed, count fetched, the other row's sums fetched).
    // Host: two channels per meter, every row enabled, the same number of sections in every row.
    // (ROWS = 1: the mono meter, whose one row needs no hand-over either -- 128 threads stand in for the 256 virtual ones)
    template <int L, int NW, bool ALIGNED, int ROWS = 2>
    __global__ __launch_bounds__(ROWS * 64 * NW, 1)
    void biquad_sumsq_ilufs_pair_kernel(const float *in, size_t in_stride, int n /* multiple of L */, const float *__restrict__ tab,
                                        float *state, const uint32_t *__restrict__ nsec, int max_sec, const sumsq_args sq,
                                        const mi_meters::ilufs_epilogue ep)
    {
        constexpr int TT = ROWS * 64 * NW;
        static_assert(TT == 128 || TT == mi_meters::VTH, "two or four real waves for the 256 virtual threads");
        __shared__ float s_seg[ROWS * 4];
        __shared__ float s_sum[4];
        __shared__ uint32_t s_cnt[4];
        __shared__ float s_val;
        __shared__ float s_chan[2 * TT];
        const uint32_t meter = blockIdx.x;
        // what the bookkeeping reads of earlier calls is asked for now and arrives underneath the filter
        const mi_meters::ilufs_early<TT> early = mi_meters::ilufs_ask<TT>(meter, ep.block, ep.cfg, uint32_t(ROWS), ep.st, ep.hist, ep.size, ep.ms_int);
        sumsq_args local = sq;
        local.sums = s_seg;
        biquad_body<L, NW, ALIGNED, false, true, ROWS>(nullptr, in, 0, in_stride, n, tab, state, nsec, max_sec, chain_args(), local, true);
        __syncthreads();
        mi_meters::ilufs_call_body<TT, false, true>(meter, ep.block, s_seg, ep.pieces, ep.cfg, uint32_t(ROWS), ep.out, ep.out_stride,
                                                    ep.st, ep.gain, ep.hist, ep.size, ep.ms_int, ep.avg, s_sum, s_cnt, s_val, s_chan, early);
    }

    template <int L, int NW, bool ALIGNED>
    __global__ __launch_bounds__(64 * NW, (NW > 1) ? 2 : 1)
    void biquad_chain_kernel(const float *in, size_t in_stride, int n /* multiple of L */, const chain_args chain)
    {
        biquad_body<L, NW, ALIGNED, true>(nullptr, in, 0, in_stride, n, nullptr, nullptr, nullptr, 0, chain);
    }

    // ---- several consecutive blocks in ONE launch -------------------------------------------------------------------
    // FilterBank::process is called block after block (FilterBank.cpp:256-291); a launch per block is one round of work on
    // this chip -- every wave loads, computes and stores at the same time, and the memory phases add to the arithmetic.
    // Here a channel's workgroup walks a STREAM of sub-blocks: the 2048-sample sub-blocks of block 0, then those of block
    // 1, ... (each block a buffer of its own, the pointers in the kernel arguments).  Wave w takes sub-blocks w, w + NW,
    // w + 2 NW, ...; the loads of its next sub-block fly underneath the sections of the current one and the stores drain
    // behind them, so in the steady state some waves of a SIMD compute while others wait on memory.
    //
    // The state travels from sub-block g - 1 to sub-block g per section, through one LDS cell per (section, wave):
    // {d0, d1, seq}.  The producer writes the state, then seq = g (a wave's LDS accesses are performed in program order);
    // the consumer reads seq, then the state, and repeats both until seq == g.  No s_barrier: a wave never waits for
    // anything but the one value it needs, so NW = 4 pipelines (with counted barriers every wave would wait for every
    // other wave's scan).  A cell cannot be overwritten early: its producer publishes (section s, sub-block g + NW) only
    // after the chain g+1 ... g+NW-1 of section s, whose first link is the consumer's own publication, made after its read.
    //
    // Bit for bit the same as `blocks` launches of biquad_bank_kernel<16, 2>: the same table, the same scan and recurrence
    // per sub-block, and the same KIND of state at every hand-over -- the scan's end state of lane 63 from an even sub-block
    // of a block to the odd one behind it (there: wave 0 to wave 1 of a super-block), the recurrence's own end state after
    // an odd sub-block and at the end of a block (there: the state saved for the next super-block / the next call).
#ifdef MI_BIQUAD_PROBE
    // per-wave account of the stream kernel (tests/experiments/biquad_stream_probe.hip): g_probe[wave][slot] =
    //   0 entry (100 MHz wall clock)  1 exit  2 shader cycles in all  3 waiting for the tile (loads)  4 sections
    //   5 transposition + store issue  6 turns of the hand-over wait loop  7 first tile ready (wall)  8 first store issued (wall)
    #define MI_STREAM_PROBE_BEGIN() unsigned long long pc_[4] = {0, 0, 0, 0}, pt_ = __builtin_readcyclecounter(), pw0_ = wall_clock64(), \
        pw1_ = 0, pw2_ = 0; const unsigned long long pc0_ = pt_; unsigned spins_ = 0
    #define MI_STREAM_PROBE(slot) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_readcyclecounter(); \
        pc_[slot] += now_ - pt_; pt_ = now_; if ((slot) == 1 && pw1_ == 0) pw1_ = wall_clock64(); if ((slot) == 3 && pw2_ == 0) pw2_ = wall_clock64(); \
        __builtin_amdgcn_sched_barrier(0); } while (0)
    #define MI_STREAM_PROBE_SPIN() (++spins_)
    #define MI_STREAM_PROBE_ITER(it) do { if (t == 0 && (it) < 20) g_probe[(size_t(blockIdx.x) * NW + wv) * 32 + 11 + (it)] = \
        (wall_clock64() & 0xffffffffull) | ((__builtin_readcyclecounter() & 0xffffffffull) << 32); } while (0)
    #define MI_STREAM_PROBE_END() do { if (t == 0) { unsigned long long *p_ = g_probe + (size_t(blockIdx.x) * NW + wv) * 32; \
        p_[0] = pw0_; p_[1] = wall_clock64(); p_[2] = __builtin_readcyclecounter() - pc0_; p_[3] = pc_[1]; p_[4] = pc_[2]; p_[5] = pc_[3] + pc_[0]; \
        p_[6] = spins_; p_[7] = pw1_; p_[8] = pw2_; \
        p_[9] = __builtin_amdgcn_s_getreg((31 << 11) | 4); p_[10] = __builtin_amdgcn_s_getreg((31 << 11) | 20); } } while (0)
#else
    #define MI_STREAM_PROBE_BEGIN() do { } while (0)
    #define MI_STREAM_PROBE(slot) do { } while (0)
    #define MI_STREAM_PROBE_SPIN() do { } while (0)
    #define MI_STREAM_PROBE_ITER(it) do { } while (0)
    #define MI_STREAM_PROBE_END() do { } while (0)
#endif
#ifndef MI_STREAM_ROTATE
#define MI_STREAM_ROTATE 1                      // 0: experiments only (the arbiter's oldest-first order decides)
#endif
    constexpr int STREAM_MAX_BLOCKS = 128;      // 2 KiB of pointers in the kernel arguments
    constexpr int STREAM_SG         = 32;       // sections with a hand-over cell
    struct stream_args
    {
        int             blocks;
        float          *out[STREAM_MAX_BLOCKS];
        const float    *in[STREAM_MAX_BLOCKS];
    };
    // ... and a CHAIN of banks on the blocks of such a run (chain_stage above; the Crossover's plan): stage k's output of block
    // i, if it has one, is out[i * outs + slot_k].  All outputs share one row stride.
    constexpr int STREAM_CHAIN_BLOCKS = 64;     // 0.5 KiB of input pointers
    constexpr int STREAM_CHAIN_PTRS   = 320;    // 2.5 KiB of output pointers: blocks * outs at most
    struct stream_chain_stage
    {
        const float    *tab;
        float          *state;
        const uint32_t *nsec;
        int             max_sec;
        int             slot;           // -1: nothing is written
        int             branch;
    };
    struct stream_chain_args
    {
        int                 blocks, stages, outs;
        stream_chain_stage  st[CHAIN_MAX];
        const float        *in[STREAM_CHAIN_BLOCKS];
        float              *out[STREAM_CHAIN_PTRS];
    };
    struct stream_cell { float d0, d1; uint32_t seq, pad; };
    typedef volatile __attribute__((address_space(3))) stream_cell lds_cell;    // ds_read / ds_write, in program order

    // QLDS: the per-lane operand of the scan ((P^2)^(lane % 16 + 1), 16 bytes per lane and section) waits in LDS for the whole
    // launch instead of being fetched from the table section by section.  As a vector load it shared the wave's in-order
    // counter with the prefetched rows of the next sub-block and the stores of the last one: the wait in front of the first
    // section's scan was a wait for all of those (s_waitcnt vmcnt(0): the loop over the sections is one piece of code and
    // cannot count differently for its first turn) -- a memory latency per sub-block in the middle of the arithmetic.
    // `cap`: sections the dynamic LDS has room for (cells, and 256 bytes per section for QLDS).
    template <int NW, bool CHAIN, bool QLDS>
    __device__ __forceinline__
    void biquad_stream_body(const stream_args *pa, const stream_chain_args *pc, size_t out_stride, size_t in_stride,
                            int n /* multiple of 16 */, const float *__restrict__ tab, float *state,
                            const uint32_t *__restrict__ nsec, int max_sec, int cap)
    {
        using G = geom<16>;
        constexpr int L = 16, W = G::W, TAB = G::TAB, PITCH = G::PITCH, SB = G::BLOCK;
        constexpr int LPT = W / 4;
        constexpr int TAB_QL = TAB_PQ + 2 * L;
        __shared__ __attribute__((aligned(16))) float sx_all[NW * 64 * PITCH];
        extern __shared__ float4 stream_dyn[];              // cells [cap][NW], then (QLDS) the scan operands [cap][16]
        stream_cell *const cell = reinterpret_cast<stream_cell *>(stream_dyn);
        const float4 *const sql = stream_dyn + cap * NW;

        const int ch  = int(blockIdx.x);
        const int tid = int(threadIdx.x);
        const int t   = tid & 63;
        const int wv  = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int l16 = t & 15;
        float *const sx = sx_all + wv * 64 * PITCH;
        const bool lane0 = (t == 0), row3 = (t >= 48);
        // stage accessors: the one bank of the plain kernel, or stage k of the chain (all wave-uniform)
        const int nst = CHAIN ? pc->stages : 1;
        auto stage_ns  = [&](int k) -> int { return CHAIN ? int(pc->st[k].nsec[ch]) : int(nsec[ch]); };
        auto stage_tab = [&](int k) -> const float * {
            return CHAIN ? pc->st[k].tab + size_t(ch) * pc->st[k].max_sec * TAB : tab + size_t(ch) * max_sec * TAB; };
        auto stage_mem = [&](int k) -> float * {
            return CHAIN ? pc->st[k].state + size_t(ch) * pc->st[k].max_sec * 2 : state + size_t(ch) * max_sec * 2; };
        const float *ctab = stage_tab(0);
        float2 *const mem0 = reinterpret_cast<float2 *>(stage_mem(0));
        const int spb   = (n + SB - 1) / SB;                 // sub-blocks of a block
        const int total = (CHAIN ? pc->blocks : pa->blocks) * spb;
        const int pred  = (wv + NW - 1) % NW;

        typedef const __attribute__((address_space(4))) float cfloat;
        typedef const __attribute__((address_space(4))) v16f cv16f;
        typedef const __attribute__((address_space(4))) v8f cv8f;
        auto uniform_row = [&](const float *T) -> cfloat * {
            const uint64_t v = reinterpret_cast<uint64_t>(T);
            const uint32_t lo = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v))));
            const uint32_t hi = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v >> 32))));
            return reinterpret_cast<cfloat *>((uint64_t(hi) << 32) | lo);
        };
        struct sectab
        {
            v16f pq[L / 8];
            v8f  m0;
            v16f m1;
            float4 cf;
            float  a2;
            float4 ql;
        };
        auto load_pq = [&](sectab &r, const float *T)
        {
            cfloat *U = uniform_row(T);
            #pragma unroll
            for (int j = 0; j < L / 8; ++j)
                r.pq[j] = *reinterpret_cast<cv16f *>(U + TAB_PQ + 16 * j);
        };
        auto load_mats = [&](sectab &r, const float *T, int qi /* the section's place among the launch's sections */)
        {
            cfloat *U = uniform_row(T);
            r.m0 = *reinterpret_cast<cv8f *>(U + 8);
            r.m1 = *reinterpret_cast<cv16f *>(U + 16);
            r.ql = QLDS ? sql[qi * 16 + l16] : *reinterpret_cast<const float4 *>(T + TAB_QL + 4 * l16);
        };
        auto load_coefs = [&](sectab &r, const float *T)
        {
            cfloat *U = uniform_row(T);
            r.cf = make_float4(U[0], U[1], U[2], U[3]);
            r.a2 = U[4];
        };

        float4 ld[LPT];
        auto issue_loads = [&](int g, bool there = true)     // coalesced rows of sub-block g -> registers (!there: zeros, no traffic)
        {
            const int k = g / spb, j = g - k * spb;
            const float *blk = CHAIN ? pc->in[k] : pa->in[k];
            const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(blk + size_t(ch) * in_stride), 0, there ? n * 4 : 0, BUFFER_DWORD3);
            #pragma unroll
            for (int q = 0; q < LPT; ++q)
            {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(src, (j * SB + 4 * (q * 64 + t)) * 4, 0, 0);
                ld[q] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
        };

        // the first tile's loads go out before anything else is asked of memory (section count, filter memory, tables):
        // those arrive underneath them
        if (wv < total)
            issue_loads(wv);
        // The rows are waited for at the top of every sub-block, with the stores of the sub-block before issued in between: the
        // wait may leave those eight stores in flight -- if the compiler's count of what follows the loads is the same on the way
        // INTO the loop as round it (it settles for the smaller of the two, and on the way in nothing followed the loads: the
        // wave then waited for its own stores to be acknowledged at the top of every sub-block).  Eight stores into a window of
        // no bytes behind the first request: dropped by the address check, counted by the compiler.
        if (QLDS)
        {
            const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc(state, 0, 0, BUFFER_DWORD3);
            #pragma unroll
            for (int q = 0; q < LPT; ++q)
                store_through(none, 4 * (q * 64 + t), make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        }
        const int ns  = stage_ns(0);
        if (!CHAIN && ns < 0)                                // row switched off (the chain's rows never are: the host sees to it)
            return;
        // the cells: nothing published yet (seq 0), and the memory the call starts from as "the state behind sub-block -1"
        // (the chain: the stages' sections one behind the other -- the host sends only chains whose sections have a cell each)
        {
            int lds0 = 0;
            for (int k = 0; k < nst; ++k)
            {
                const int nk = (k == 0) ? ns : stage_ns(k);
                const float *mk = stage_mem(k);
                for (int i = tid; i < nk * NW; i += 64 * NW)
                {
                    const int si = i / NW, w = i - si * NW;
                    stream_cell c = { 0.0f, 0.0f, 0u, 0u };
                    if (w == NW - 1)
                    {
                        const float2 s = reinterpret_cast<const float2 *>(mk)[si];
                        c.d0 = s.x;
                        c.d1 = s.y;
                    }
                    cell[(lds0 + si) * NW + w] = c;
                }
                if (QLDS)
                {
                    const float *tk = stage_tab(k);
                    for (int i = tid; i < nk * 16; i += 64 * NW)
                        stream_dyn[cap * NW + lds0 * 16 + i] = *reinterpret_cast<const float4 *>(tk + size_t(i >> 4) * TAB + TAB_QL + 4 * (i & 15));
                }
                lds0 += nk;
            }
        }
        __syncthreads();

        v2f x[L];
        sectab tb;
        if (CHAIN || ns > 0)
        {
            load_pq(tb, ctab);
            load_mats(tb, ctab, 0);
            load_coefs(tb, ctab);
        }
        MI_STREAM_PROBE_BEGIN();

        for (int g = wv; g < total; g += NW)
        {
            MI_STREAM_PROBE(0);
            const int k = g / spb, j = g - k * spb;
            const int base  = j * SB;
            const int valid = (n - base < SB) ? n - base : SB;       // multiple of L
            const int last  = valid - 1;
            const int t_last = last / W;
            const bool save_hi = (last - t_last * W) >= L;
            const bool saver   = (t == t_last);
            const bool exact_out = (j & 1) || (j == spb - 1);        // how this sub-block hands its state on
            const bool final_sb  = (g == total - 1);
            const uint32_t want = uint32_t(g), mine = uint32_t(g + 1);

            // The tile holds a lane's two chunks INTERLEAVED (first, second, first, second, ...): a 16-byte read is two
            // register pairs {first chunk's sample, second chunk's sample} as the packed arithmetic wants them, no moves.
            // The row pieces go in (and come out) with a stride of two dwords instead.
            #pragma unroll
            for (int q = 0; q < LPT; ++q)
            {
                const int i = 4 * (q * 64 + t);              // the piece's first sample in the sub-block
                float *d = &sx[(i / W) * PITCH + 2 * (i % L) + ((i % W) / L)];
                d[0] = ld[q].x; d[2] = ld[q].y; d[4] = ld[q].z; d[6] = ld[q].w;
            }
            __builtin_amdgcn_wave_barrier();
            MI_STREAM_PROBE(1);
            if (!CHAIN && g + NW < total)                    // (the chain asks later: see below)
                issue_loads(g + NW);
            #pragma unroll
            for (int q = 0; q < L / 2; ++q)
            {
                const float4 v = *reinterpret_cast<const float4 *>(&sx[t * PITCH + 4 * q]);
                x[2 * q] = v2f{v.x, v.y};
                x[2 * q + 1] = v2f{v.z, v.w};
            }

            // One section over the sub-block: `tb` holds its table on entry and the table at Tnext on exit; ci: its hand-over
            // cell; mem2: where its state goes at the end of the launch; turn: what the issue priority goes round with.
            auto section = [&](const float *Tnext, int qnext, int ci, float2 *mem2, int turn)
            {
#if MI_STREAM_ROTATE
                // The instruction arbiter of a SIMD serves its OLDEST wave first: of the four workgroups that share a CU --
                // on this part workgroups b, b + 256, b + 512, b + 768 of a 1024-channel launch, in the order of their
                // entry (census in tests/experiments/biquad_stream_probe.hip) -- the first runs ahead and the last falls
                // behind (exits 480 .. 850 us apart in a 125-block launch), and the launch ends with the stragglers alone
                // on their SIMDs.  So the issue priority goes round with a wave's own progress, every four sections, each
                // workgroup of a CU starting at a different level: exits 720 .. 750 us, the launch 3 - 6 % shorter.  (Keeping
                // the pace through progress marks in global memory balances perfectly and costs a factor 1.8: the agent-scope
                // marks alone do, profiles/r04_experiments/biquad_stream_pace.txt; priorities switched by the 100 MHz
                // clock instead of the progress: the same balance, no shorter.)
                if ((turn & 3) == 0)
                    switch (((turn >> 2) + (g / NW) + int(blockIdx.x >> 8)) & 3)
                    {
                        case 0: __builtin_amdgcn_s_setprio(3); break;
                        case 1: __builtin_amdgcn_s_setprio(2); break;
                        case 2: __builtin_amdgcn_s_setprio(1); break;
                        default: __builtin_amdgcn_s_setprio(0); break;
                    }
#endif
                lds_cell *const from = (lds_cell *)&cell[ci * NW + pred];
                lds_cell *const to   = (lds_cell *)&cell[ci * NW + wv];
                // asked for now, looked at after the dot products
                uint32_t got = from->seq;
                float c0 = from->d0, c1 = from->d1;

                // 1. zero-state end states of both chunks
                v2f a0 = splat(0.0f), a1_ = splat(0.0f), b0_ = splat(0.0f), b1_ = splat(0.0f);
                #pragma unroll
                for (int q = 0; q < L; q += 2)
                {
                    const v16f &r = tb.pq[q / 8];
                    const v2f pq0 = v2f{r[(2 * q) % 16], r[(2 * q + 1) % 16]};
                    const v2f pq1 = v2f{r[(2 * q + 2) % 16], r[(2 * q + 3) % 16]};
                    a0  = pk_fma(pq0, splat(x[q].x), a0);      b0_ = pk_fma(pq0, splat(x[q].y), b0_);
                    a1_ = pk_fma(pq1, splat(x[q + 1].x), a1_); b1_ = pk_fma(pq1, splat(x[q + 1].y), b1_);
                }
                const v2f zwA = a0 + a1_, zwB = b0_ + b1_;

                // 2. end state of the pair for a zero start, the state entering the sub-block at lane 0
                const v2f Pc0 = v2f{tb.m0[0], tb.m0[1]}, Pc1 = v2f{tb.m0[2], tb.m0[3]};
                v2f e = mat_fma(Pc0, Pc1, zwA, zwB);
                asm volatile("" : "+v"(e));                  // the dot products stay in front of the wait (no sinking behind the loop)
                while (got != want)
                {
                    MI_STREAM_PROBE_SPIN();
                    __builtin_amdgcn_s_sleep(1);
                    got = from->seq;
                    c0 = from->d0;
                    c1 = from->d1;
                }
                const v2f cvec = lane0 ? v2f{c0, c1} : splat(0.0f);
                e = mat_fma(v2f{tb.m0[4], tb.m0[5]}, v2f{tb.m0[6], tb.m0[7]}, cvec, e);

                const v2f zero = splat(0.0f);
                e = mat_fma(v2f{tb.m0[4], tb.m0[5]}, v2f{tb.m0[6], tb.m0[7]}, dpp_zero<DPP_ROW_SHR1>(e), e);
                e = mat_fma(v2f{tb.m1[0], tb.m1[1]}, v2f{tb.m1[2], tb.m1[3]}, dpp_zero<DPP_ROW_SHR2>(e), e);
                e = mat_fma(v2f{tb.m1[4], tb.m1[5]}, v2f{tb.m1[6], tb.m1[7]}, dpp_zero<DPP_ROW_SHR4>(e), e);
                e = mat_fma(v2f{tb.m1[8], tb.m1[9]}, v2f{tb.m1[10], tb.m1[11]}, dpp_zero<DPP_ROW_SHR8>(e), e);
                const v2f QLc0 = v2f{tb.ql.x, tb.ql.y}, QLc1 = v2f{tb.ql.z, tb.ql.w};
                e = mat_fma(QLc0, QLc1, dpp_or<DPP_ROW_BCAST15, 0xa>(zero, e), e);
                {
                    const v2f s  = dpp_or<DPP_ROW_BCAST31, 0xc>(zero, e);
                    const v2f s2 = mat_fma(v2f{tb.m1[12], tb.m1[13]}, v2f{tb.m1[14], tb.m1[15]}, s, zero);
                    e = mat_fma(QLc0, QLc1, row3 ? s2 : s, e);
                }
                if (!exact_out && t == 63)                   // the scan's end state goes on (a full sub-block)
                {
                    to->d0 = e.x;
                    to->d1 = e.y;
                    to->seq = mine;
                }

                // 3. start states, the next section's table underneath the recurrence
                const v2f S  = dpp_or<DPP_WAVE_SHR1, 0xf>(cvec, e);
                const v2f SB2 = mat_fma(Pc0, Pc1, S, zwA);
                v2f d0 = v2f{S.x, SB2.x}, d1 = v2f{S.y, SB2.y};
                __builtin_amdgcn_sched_barrier(0);
                load_pq(tb, Tnext);
                load_mats(tb, Tnext, qnext);
                __builtin_amdgcn_sched_barrier(0);

                const v2f b0 = splat(tb.cf.x), b1 = splat(tb.cf.y), b2 = splat(tb.cf.z), a1 = splat(tb.cf.w), a2 = splat(tb.a2);
                #pragma unroll
                for (int q = 0; q < L; ++q)
                {
                    const v2f xx = x[q];
                    const v2f tq = pk_fma(b1, xx, d1);
                    const v2f u  = b2 * xx;
                    const v2f y  = pk_fma(b0, xx, d0);
                    d0   = pk_fma(a1, y, tq);
                    d1   = pk_fma(a2, y, u);
                    x[q] = y;
                }
                __builtin_amdgcn_sched_barrier(0);
                load_coefs(tb, Tnext);
                if (saver && (exact_out || final_sb))
                {
                    const float s0 = save_hi ? d0.y : d0.x, s1 = save_hi ? d1.y : d1.x;
                    if (exact_out)
                    {
                        to->d0 = s0;
                        to->d1 = s1;
                        to->seq = mine;
                    }
                    if (final_sb)                            // the memory the next call starts from
                        *mem2 = make_float2(s0, s1);
                }
            };
            // x (registers) -> the wave's tile (transposed) -> coalesced write-through store
            auto store_tile = [&](float *rows)
            {
                const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(rows + size_t(ch) * out_stride, 0, n * 4, BUFFER_DWORD3);
                #pragma unroll
                for (int q = 0; q < L / 2; ++q)
                    *reinterpret_cast<float4 *>(&sx[t * PITCH + 4 * q]) = make_float4(x[2 * q].x, x[2 * q].y, x[2 * q + 1].x, x[2 * q + 1].y);
                __builtin_amdgcn_wave_barrier();
                #pragma unroll
                for (int q = 0; q < LPT; ++q)
                {
                    const int i = 4 * (q * 64 + t);
                    const float *d = &sx[(i / W) * PITCH + 2 * (i % L) + ((i % W) / L)];
                    store_through(dst, base + i, make_float4(d[0], d[2], d[4], d[6]));
                }
                __builtin_amdgcn_wave_barrier();
            };

            if constexpr (!CHAIN)
            {
                for (int si = 0; si < ns; ++si)
                {
                    const int snext = (si + 1 < ns) ? si + 1 : 0;
                    section(ctab + size_t(snext) * TAB, snext, si, mem0 + si, si);
                }
                MI_STREAM_PROBE(2);
                store_tile(pa->out[k]);
            }
            else
            {
                // Every stage has at least one section on every channel (the host sends nothing else): the table that follows
                // a stage's last section is the next stage's first one.  What a stage needs is fetched one stage ahead, so that
                // the scalar loads complete underneath the sections of the stage before.
                struct stage_info { int ns; const float *T; float2 *mem2; int slot, branch; };
                auto fetch = [&](int q) -> stage_info {
                    stage_info v;
                    v.ns = stage_ns(q);
                    v.T = stage_tab(q);
                    v.mem2 = reinterpret_cast<float2 *>(stage_mem(q));
                    v.slot = pc->st[q].slot;
                    v.branch = pc->st[q].branch;
                    return v;
                };
                // A branch keeps a copy of the travelling signal (32 registers) and the next sub-block's rows wait in 32 more:
                // the two never at the same time -- the rows are asked for in front of the LAST stage (which the host sends
                // as an in-place one: the Crossover's last high-pass), whose sections cover the latency of memory, so the
                // kernel stays within the 128 registers of four waves per SIMD.
                auto run_stage = [&](auto may_branch, const stage_info &cur, const float *Tn, int lds0, int qn /* Tn's place */)
                {
                    constexpr bool BR = decltype(may_branch)::value;
                    v2f xs[L];                               // the travelling signal while a branch is computed
                    if (BR && cur.branch)
                    {
                        #pragma unroll
                        for (int i = 0; i < L; ++i)
                            xs[i] = x[i];
                    }
                    for (int si = 0; si < cur.ns; ++si)
                        section((si + 1 < cur.ns) ? cur.T + size_t(si + 1) * TAB : Tn, (si + 1 < cur.ns) ? lds0 + si + 1 : qn,
                                lds0 + si, cur.mem2 + si, lds0 + si);
                    if (cur.slot >= 0)
                        store_tile(pc->out[k * pc->outs + cur.slot]);
                    if (BR && cur.branch)
                    {
                        #pragma unroll
                        for (int i = 0; i < L; ++i)
                            x[i] = xs[i];
                    }
                };
                int lds0 = 0;
                stage_info cur = fetch(0);
                for (int q = 0; q + 1 < nst; ++q)
                {
                    const stage_info nxt = fetch(q + 1);
                    run_stage(std::true_type(), cur, nxt.T, lds0, lds0 + cur.ns);
                    lds0 += cur.ns;
                    cur = nxt;
                }
                // (asked for unconditionally -- behind the last sub-block from a window of no bytes: a conditional request would
                // keep the registers of the rows alive across the whole iteration for the case that it is not made)
                {
                    const bool more = g + NW < total;
                    issue_loads(more ? g + NW : g, more);
                }
                run_stage(std::false_type(), cur, ctab, lds0, 0);
                MI_STREAM_PROBE(2);
            }
            MI_STREAM_PROBE(3);
            MI_STREAM_PROBE_ITER(g / NW);
        }
        MI_STREAM_PROBE_END();
    }

    __device__ unsigned long long g_stream_clock[4];        // {cycles, wall} at entry, {cycles, wall} at exit of workgroup 0

    template <int NW, bool QLDS>
    __global__ __launch_bounds__(64 * NW, (NW >= 4) ? 4 : 2)
    void biquad_stream_kernel(const stream_args a, size_t out_stride, size_t in_stride, int n /* multiple of 16 */,
                              const float *__restrict__ tab, float *state, const uint32_t *__restrict__ nsec, int max_sec, int cap)
    {
        // (the shader clock the launch ran at: workgroup 0 stamps the shader's cycle counter and the 100 MHz wall clock at its
        // entry and exit -- mi_dspu_last_stream_clock; a workgroup's life is the launch's for a channel's run of blocks)
        const bool stamp = blockIdx.x == 0 && threadIdx.x == 0;
        if (stamp)
        {
            g_stream_clock[0] = __builtin_readcyclecounter();
            g_stream_clock[1] = wall_clock64();
        }
        biquad_stream_body<NW, false, QLDS>(&a, nullptr, out_stride, in_stride, n, tab, state, nsec, max_sec, cap);
        if (stamp)
        {
            g_stream_clock[2] = __builtin_readcyclecounter();
            g_stream_clock[3] = wall_clock64();
        }
    }

    // the chain on a run of blocks
    template <int NW, bool QLDS>
    __global__ __launch_bounds__(64 * NW, 4)
    void biquad_stream_chain_kernel(const stream_chain_args c, size_t out_stride, size_t in_stride, int n /* multiple of 16 */, int cap)
    {
        biquad_stream_body<NW, true, QLDS>(nullptr, &c, out_stride, in_stride, n, nullptr, nullptr, nullptr, 0, cap);
    }

    // Dynamic LDS of a stream launch: a cell per (section, wave), and with the scan operands in LDS 256 bytes per section more.
    // The operands go to LDS as long as four workgroups of four waves still share a CU's 160 KiB with them (the tiles take
    // 36 KiB per workgroup): up to twelve sections.
    inline bool stream_qlds(int nw, int cap) { return size_t(nw) * 64 * geom<16>::PITCH * 4 + size_t(cap) * (nw + 16) * 16 <= 40960; }
    inline size_t stream_lds(int nw, int cap, bool qlds) { return size_t(cap) * (nw + (qlds ? 16 : 0)) * 16; }

    // The last samples % L samples of a call: one thread per channel walks them through the cascade with the same
    // recurrence (FilterBank.cpp:256-291 semantics for block sizes that are not a multiple of the chunk length).
    __global__ void biquad_tail_kernel(float *out, const float *in, size_t out_stride, size_t in_stride,
                                       size_t start, int count /* < 16 */, const float *__restrict__ tab, int tab_row,
                                       float *state, const uint32_t *__restrict__ nsec, int max_sec, uint32_t channels,
                                       const sumsq_args sq)
    {
        const uint32_t ch = blockIdx.x * blockDim.x + threadIdx.x;
        if (ch >= channels)
            return;
        const int ns = int(nsec[ch]);
        if (ns < 0)
            return;
        float xs[16];
        for (int k = 0; k < 16; ++k)
            xs[k] = (k < count) ? in[size_t(ch) * in_stride + start + k] : 0.0f;
        for (int s = 0; s < ns; ++s)
        {
            const float *q = tab + (size_t(ch) * max_sec + s) * tab_row;
            float *st = state + (size_t(ch) * max_sec + s) * 2;
            const float b0 = q[0], b1 = q[1], b2 = q[2], a1 = q[3], a2 = q[4];
            float d0 = st[0], d1 = st[1];
            for (int k = 0; k < count; ++k)
            {
                const float xx = xs[k];
                const float tq = fmaf(b1, xx, d1);
                const float u  = b2 * xx;
                const float y  = fmaf(b0, xx, d0);
                d0 = fmaf(a1, y, tq);
                d1 = fmaf(a2, y, u);
                xs[k] = y;
            }
            st[0] = d0;
            st[1] = d1;
        }
        if (sq.sums != nullptr)                             // the meters' epilogue: squares into the segments' sums
        {
            float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
            for (int k = 0; k < count; ++k)
            {
                const int i = int(start) + k, sk = int(i >= sq.e0) + int(i >= sq.e1) + int(i >= sq.e2);
                for (int j = 0; j < 4; ++j)
                    acc[j] = (sk == j) ? fmaf(xs[k], xs[k], acc[j]) : acc[j];
            }
            for (int j = 0; j < 4; ++j)
                sq.sums[size_t(ch) * 4 + j] += acc[j];
            return;
        }
        for (int k = 0; k < count; ++k)
            out[size_t(ch) * out_stride + start + k] = xs[k];
    }

    // The impulse response of a channel's cascade in the REFERENCE's arithmetic, operation for operation: the generic
    // transposed direct form II section of lsp-dsp-lib (y = b0 x + d0; p1 = b1 x + a1 y; p2 = b2 x + a2 y; d0 = d1 + p1;
    // d1 = p2 -- every product and every sum rounded on its own, no fused multiply-add), sample after sample.  Design-time
    // work: the Equalizer synthesises its FIR from this response (Equalizer.cpp:284-288), and the round-off of a float32
    // recursion through 32 sections is worth 1e-4 of the taps -- taken in another order the taps are another filter of the
    // same accuracy, taken in this order they are the reference's to the last bit.
    // One wave per channel, lane j = section j, a systolic line: at step t lane j works on sample t - j, which lane j - 1
    // finished one step earlier (wave_shr:1); what a section computes for a sample does not depend on when it does.  More
    // than 64 sections: passes of 64, the later ones reading what the pass before left in `out`.
    __global__ __launch_bounds__(64)
    void biquad_reference_ir_kernel(float *out, size_t stride, int n, const float *__restrict__ tab, int tab_row,
                                    const uint32_t *__restrict__ nsec, int max_sec)
    {
        const int ch = int(blockIdx.x), t = int(threadIdx.x);
        const int ns = int(nsec[ch] & 0x7fffffffu);
        float *o = out + size_t(ch) * stride;
        const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(o, 0, n * 4, BUFFER_DWORD3);
        if (ns == 0)                                         // FilterBank.cpp:261-265: an empty bank copies (the impulse)
        {
            for (int i = t; i < n; i += 64)
                o[i] = (i == 0) ? 1.0f : 0.0f;
            return;
        }
        for (int s0 = 0; s0 < ns; s0 += 64)
        {
            const int g = (ns - s0 < 64) ? ns - s0 : 64;    // sections of this pass
            const float *q = tab + (size_t(ch) * max_sec + s0 + ((t < g) ? t : 0)) * tab_row;
            const float b0 = q[0], b1 = q[1], b2 = q[2], a1 = q[3], a2 = q[4];
            float d0 = 0.0f, d1 = 0.0f, y = 0.0f, chunk = 0.0f;
            for (int s64 = 0; s64 < n + g - 1; s64 += 64)
            {
                // the next 64 input samples, one per lane (waited for once, in front of the 64 steps)
                chunk = (s0 == 0) ? ((s64 + t == 0) ? 1.0f : 0.0f) : ((s64 + t < n) ? o[s64 + t] : 0.0f);
                const int steps = (n + g - 1 - s64 < 64) ? n + g - 1 - s64 : 64;
                for (int k = 0; k < steps; ++k)
                {
                    const float x0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(chunk), k));
                    // lane 0 takes the input, lane j what lane j - 1 produced in the step before; lanes that have not
                    // started yet pass zeros through a zero state, lanes past their last sample produce what nobody stores
                    const float x = dpp_or<DPP_WAVE_SHR1, 0xf>(x0, y);
                    y = __fadd_rn(__fmul_rn(b0, x), d0);
                    const float p1 = __fadd_rn(__fmul_rn(b1, x), __fmul_rn(a1, y));
                    const float p2 = __fadd_rn(__fmul_rn(b2, x), __fmul_rn(a2, y));
                    d0 = __fadd_rn(d1, p1);
                    d1 = p2;
                    const int i = s64 + k - (g - 1);        // the last section of the pass emits sample i
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), dst, (t == g - 1 && i >= 0) ? i * 4 : -4, 0, 0);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");      // the next pass reads this pass's output
        }
    }

    // ---- FilterBank::process in the REFERENCE's arithmetic, operation for operation (the bank's opt-in exact mode, round 6) --------
    // The kernels above split time into chunks and re-join them with a scan: the same filter in another order of roundings
    // (only the chunks' start states differ from the serial recursion's, by round-off -- but a recursive filter near the unit
    // circle remembers that for thousands of samples: the distance from the serial recursion reaches 3e-4 of the peak at C2's 200 Hz low-passes, and
    // tests/conftest.py holds the fast kernels to a noise rule there instead of north_star's 1e-5).  This kernel runs the
    // recurrence of FilterBank.cpp:256-291 as lsp-dsp-lib's generic biquad_process_x1 writes it -- y = b0 x + d0;
    // p1 = b1 x + a1 y; p2 = b2 x + a2 y; d0 = d1 + p1; d1 = p2, every product and every sum rounded on its own -- sample after
    // sample: what the CPU's x8 form does with its eight SIMD lanes.  G lanes per channel (the power of two that holds the
    // bank's longest cascade), lane j = section j, a systolic line: at step s lane j works on sample s - j, which lane j - 1
    // finished one step earlier (wave_shr:1); what a section computes for a sample does not depend on when it does.  The filter
    // memory is the bank's own {d0, d1} per section, read at the start and left as the recursion leaves it: calls in this mode
    // and in the fast one can follow each other.  The result is the serial CPU recursion's BIT FOR BIT (tests/test_biquad_gpu.py); the price
    // is the recursion's latency chain, n steps of some eighteen instructions one behind the other per call (220 us per 4096
    // samples whatever the channel count up to a wave per SIMD's worth: 19 000 Msamples/s at C2 against the fast kernels'
    // 280 000 - 550 000; a 16-core host runs the vectorised x8 form at 17 700).
    // 64 input samples per lane sit in registers one chunk ahead (every lane of a channel loads the same addresses); more than
    // 64 sections: passes of 64, the later ones in place on `out`.
    template <int G>
    __global__ __launch_bounds__(64)
    void biquad_exact_kernel(float *out, const float *in, size_t out_stride, size_t in_stride, int n,
                             const float *__restrict__ tab, int tab_row, float *state, const uint32_t *__restrict__ nsec,
                             int max_sec, int channels, int cap /* the longest cascade of the bank */)
    {
        constexpr int CPW = 64 / G;                         // channels of a wave
        const int t = int(threadIdx.x), j = t % G;
        const int ch = int(blockIdx.x) * CPW + t / G;
        const uint32_t nv = (ch < channels) ? nsec[ch] : 0x80000000u;       // (sign bit: the row is switched off -- nothing read, nothing written)
        const bool on = (nv & 0x80000000u) == 0u;
        const int ns = int(nv & 0x7fffffffu);
        const int chc = (ch < channels) ? ch : 0;
        float *const o = out + size_t(chc) * out_stride;
        const float *const x_in = in + size_t(chc) * in_stride;
        if (on && ns == 0 && o != x_in)                     // FilterBank.cpp:261-265: an empty bank copies
            for (int i = j; i < n; i += G)
                o[i] = x_in[i];
        for (int s0 = 0; s0 < cap; s0 += 64)
        {
            const bool mine = on && s0 + j < ns;            // this lane's section exists
            const int cnt = on ? ((ns - s0 < G) ? ns - s0 : G) : 0;        // sections of the channel in this pass (<= 0: it is through)
            const bool emit = cnt >= 1 && j == cnt - 1;     // the pass's last section of the channel hands the samples out
            const float *q = tab + (size_t(chc) * max_sec + (mine ? s0 + j : 0)) * tab_row;
            const float b0 = mine ? q[0] : 0.0f, b1 = mine ? q[1] : 0.0f, b2 = mine ? q[2] : 0.0f, a1 = mine ? q[3] : 0.0f, a2 = mine ? q[4] : 0.0f;
            float *const st = state + (size_t(chc) * max_sec + (mine ? s0 + j : 0)) * 2;
            float d0 = mine ? st[0] : 0.0f, d1 = mine ? st[1] : 0.0f, y = 0.0f;
            const float *const src = (s0 == 0) ? x_in : o;  // later passes run in place (FilterBank.cpp:270)
            // (unconditional loads: a load under a condition per element compiles to a cascade of 64 x 64 register copies.  A chunk
            // that reaches past the call's end reads its last sample again -- a value no active lane takes)
            auto fetch = [&](int s64, float (&v)[64]) {
                if (s64 + 64 <= n)
                {
                    const float *p = src + s64;
                    #pragma unroll
                    for (int i = 0; i < 64; ++i)
                        v[i] = p[i];
                    return;
                }
                #pragma unroll
                for (int i = 0; i < 64; ++i)
                    v[i] = src[(s64 + i < n) ? s64 + i : n - 1];
            };
            const int steps = n + G - 1;
            float xs[64], xn[64];
            fetch(0, xs);
            // one chunk of 64 steps on `cur` while `nxt` takes the chunk behind it; two chunks per trip of the loop, the two sets
            // of registers changing places by name (copying one into the other was 64 of a chunk's 770 vector instructions)
            auto chunk = [&](const int s64, const float (&cur)[64], float (&nxt)[64]) __attribute__((always_inline))
            {
                fetch(s64 + 64, nxt);                        // the next chunk's samples, in flight over this chunk's steps (zeros behind the call's end)
                const bool edge = s64 < G - 1 || s64 + 64 > n;      // lanes start one step after another and stop one after another
                if (!edge)
                {
                    // (the chunk's 64 results stay in registers and leave as sixteen 16-byte stores behind the chunk: a store per
                    // step was five instructions of a step's twenty -- exec mask, branch, address, store, exec mask; the shift
                    // fills lane 0 with a zero it never takes instead of copying one in)
                    float ys[64];
                    #pragma unroll
                    for (int k = 0; k < 64; ++k)
                    {
                        const float up = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(y), DPP_WAVE_SHR1, 0xf, 0xf, true));
                        const float x = (j == 0) ? cur[k] : up;
                        y = __fadd_rn(__fmul_rn(b0, x), d0);
                        const float p1 = __fadd_rn(__fmul_rn(b1, x), __fmul_rn(a1, y));
                        const float p2 = __fadd_rn(__fmul_rn(b2, x), __fmul_rn(a2, y));
                        d0 = __fadd_rn(d1, p1);
                        d1 = p2;
                        ys[k] = y;
                    }
                    if (emit)
                    {
                        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));      // (rows and the lane's skew: 4-byte aligned)
                        float *const p = o + (s64 - j);
                        #pragma unroll
                        for (int k = 0; k < 64; k += 4)
                            *reinterpret_cast<f4u *>(p + k) = f4u{ys[k], ys[k + 1], ys[k + 2], ys[k + 3]};
                    }
                }
                else
                {
                    #pragma unroll
                    for (int k = 0; k < 64; ++k)
                    {
                        const int i = s64 + k - j;          // the sample this lane's section meets at this step
                        const bool act = mine && i >= 0 && i < n;
                        const float up = dpp_or<DPP_WAVE_SHR1, 0xf>(0.0f, y);
                        const float x = (j == 0) ? cur[k] : up;
                        const float yn = __fadd_rn(__fmul_rn(b0, x), d0);
                        const float p1 = __fadd_rn(__fmul_rn(b1, x), __fmul_rn(a1, yn));
                        const float p2 = __fadd_rn(__fmul_rn(b2, x), __fmul_rn(a2, yn));
                        y = yn;
                        d0 = act ? __fadd_rn(d1, p1) : d0;
                        d1 = act ? p2 : d1;
                        if (emit && act)
                            o[i] = yn;
                    }
                }
            };
            for (int s64 = 0; s64 < steps; s64 += 128)
            {
                chunk(s64, xs, xn);
                if (s64 + 64 < steps)
                    chunk(s64 + 64, xn, xs);
            }
            if (mine)
            {
                st[0] = d0;
                st[1] = d1;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");      // the next pass reads this pass's output
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

    template <int L>
    void fill_row(float *row, const float *q /* b0 b1 b2 a1 a2 */)
    {
        const double b0 = q[0], b1 = q[1], b2 = q[2], a1 = q[3], a2 = q[4];
        for (int i = 0; i < 8; ++i)
            row[i] = (i < 5) ? q[i] : 0.0f;
        // s' = A s + B x  with  A = [a1 1; a2 0],  B = [b1 + a1 b0, b2 + a2 b0]
        const mat2 A = { a1, 1.0, a2, 0.0 };
        double v0 = b1 + a1 * b0, v1 = b2 + a2 * b0;
        float *pq = row + TAB_PQ;
        for (int k = L - 1; k >= 0; --k)        // (p[k], q[k]) = A^(L-1-k) B
        {
            pq[2 * k]     = float(v0);
            pq[2 * k + 1] = float(v1);
            const double n0 = A.a * v0 + A.b * v1, n1 = A.c * v0 + A.d * v1;
            v0 = n0;
            v1 = n1;
        }
        const auto put = [](float *m, const mat2 &M) { m[0] = float(M.a); m[1] = float(M.c); m[2] = float(M.b); m[3] = float(M.d); };
        mat2 P = { 1.0, 0.0, 0.0, 1.0 };
        for (int k = 0; k < L; ++k)
            P = mul(P, A);
        mat2 M = P;                              // P, P^2, P^4, P^8, P^16, P^32
        for (int i = 0; i < 6; ++i)
        {
            put(row + 8 + 4 * i, M);
            M = mul(M, M);
        }
        const mat2 P2 = mul(P, P);
        mat2 Qi = P2;                            // (P^2)^(i+1)
        for (int i = 0; i < 16; ++i)
        {
            put(row + TAB_PQ + 2 * L + 4 * i, Qi);
            Qi = mul(Qi, P2);
        }
    }

    using big   = geom<16>;     // sub-blocks of 2048 samples: calls longer than 2048 samples
    using small = geom<8>;      // sub-blocks of 1024 samples: short calls
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
    std::vector<uint8_t>    row_off;        // channel switched off: process() leaves its state and its output alone
    bool                    nsec_dirty  = false;
    bool                    pending     = false;
    bool                    out_reread  = false;   // the owner's next launch reads process()'s output (LoudnessMeter): keep it in L2
    bool                    exact       = false;   // process() on biquad_exact_kernel: the reference's serial recurrence, bit for bit
    std::vector<float>      h_big, h_small; // host images of the device tables
    float                  *d_big       = nullptr;
    float                  *d_small     = nullptr;
    float                  *d_state     = nullptr;
    float                  *d_backup    = nullptr;
    uint32_t               *d_nsec      = nullptr;
};

namespace
{
    template <int L, int NW>
    hipError_t launch(mi_biquad_bank *b, float *out, const float *in, size_t out_stride,
                      size_t in_stride, int n, bool aligned, const float *tab, hipStream_t st, const sumsq_args *sq = nullptr,
                      const mi_meters::ilufs_epilogue *ep = nullptr, bool pair = false /* the stereo meter in one workgroup */)
    {
        const dim3 grid(b->channels), block(64 * NW);
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        mi::take_profile_events(&ev0, &ev1);
        if constexpr (L == 16 && NW == 2)
        {
            if (sq != nullptr && ep != nullptr && pair && ep->channels == 1)
            {
                if (aligned)
                    MI_LAUNCH((biquad_sumsq_ilufs_pair_kernel<L, NW, true, 1>), grid, block, 0, st, ev0, ev1, in, in_stride, n, tab,
                              b->d_state, b->d_nsec, int(b->max_sec), *sq, *ep);
                else
                    MI_LAUNCH((biquad_sumsq_ilufs_pair_kernel<L, NW, false, 1>), grid, block, 0, st, ev0, ev1, in, in_stride, n, tab,
                              b->d_state, b->d_nsec, int(b->max_sec), *sq, *ep);
                return hipGetLastError();
            }
            if (sq != nullptr && ep != nullptr && pair)
            {
                const dim3 pgrid(b->channels / 2), pblock(2 * 64 * NW);
                if (aligned)
                    MI_LAUNCH((biquad_sumsq_ilufs_pair_kernel<L, NW, true>), pgrid, pblock, 0, st, ev0, ev1, in, in_stride, n, tab,
                              b->d_state, b->d_nsec, int(b->max_sec), *sq, *ep);
                else
                    MI_LAUNCH((biquad_sumsq_ilufs_pair_kernel<L, NW, false>), pgrid, pblock, 0, st, ev0, ev1, in, in_stride, n, tab,
                              b->d_state, b->d_nsec, int(b->max_sec), *sq, *ep);
                return hipGetLastError();
            }
            if (sq != nullptr && ep != nullptr)
            {
                if (aligned)
                    MI_LAUNCH((biquad_sumsq_ilufs_kernel<L, NW, true>), grid, block, 0, st, ev0, ev1, in, in_stride, n, tab,
                              b->d_state, b->d_nsec, int(b->max_sec), *sq, *ep);
                else
                    MI_LAUNCH((biquad_sumsq_ilufs_kernel<L, NW, false>), grid, block, 0, st, ev0, ev1, in, in_stride, n, tab,
                              b->d_state, b->d_nsec, int(b->max_sec), *sq, *ep);
                return hipGetLastError();
            }
        }
        if (sq != nullptr)
        {
            if (aligned)
                MI_LAUNCH((biquad_sumsq_kernel<L, NW, true>), grid, block, 0, st, ev0, ev1, in, in_stride, n, tab,
                          b->d_state, b->d_nsec, int(b->max_sec), *sq);
            else
                MI_LAUNCH((biquad_sumsq_kernel<L, NW, false>), grid, block, 0, st, ev0, ev1, in, in_stride, n, tab,
                          b->d_state, b->d_nsec, int(b->max_sec), *sq);
        }
        else if (aligned)
            MI_LAUNCH((biquad_bank_kernel<L, NW, true>), grid, block, 0, st, ev0, ev1, out, in,
                                  out_stride, in_stride, n, tab, b->d_state, b->d_nsec, int(b->max_sec), int(b->out_reread));
        else
            MI_LAUNCH((biquad_bank_kernel<L, NW, false>), grid, block, 0, st, ev0, ev1, out, in,
                                  out_stride, in_stride, n, tab, b->d_state, b->d_nsec, int(b->max_sec), int(b->out_reread));
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
                fill_row<16>(&b->h_big[c * row_big + size_t(s) * big::TAB], q);
                fill_row<8>(&b->h_small[c * row_small + size_t(s) * small::TAB], q);
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
            b->nsec_dirty = true;
        }
        if (b->nsec_dirty)
        {
            // section count per channel; the sign bit marks a channel that is switched off
            std::vector<uint32_t> v(b->nsec);
            for (uint32_t c = 0; c < b->channels; ++c)
                if (b->row_off[c])
                    v[c] |= 0x80000000u;
            MI_HIP_CHECK(hipMemcpyAsync(b->d_nsec, v.data(), b->channels * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            MI_HIP_CHECK(hipStreamSynchronize(st));
            b->nsec_dirty = false;
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

namespace mi
{
    int biquad_chain_process_blocks(const biquad_chain_stage *stages, const int *slot, int count, int outs,
                                    float *const *out, const float *const *in, size_t blocks, size_t samples,
                                    size_t out_stride, size_t in_stride, hipStream_t st);

    // One launch for a chain of banks on the same samples (see chain_stage above).  Returns MI_OK when the fused launch
    // was issued, 1 when the call does not qualify (the caller then runs the banks one after the other as before):
    // blocks longer than one wave's sub-block of 2048 samples that are a multiple of the chunk, 16-byte aligned rows, at
    // most CHAIN_MAX stages, the sections of all stages together within the LDS state slots, every stage with at least one
    // section on every channel and no row switched off.
    int biquad_chain_process(const biquad_chain_stage *stages, int count, const float *in, size_t in_stride,
                             size_t samples, hipStream_t st, bool long_calls_as_streams)
    {
        if (count <= 0 || count > CHAIN_MAX || samples <= 2 * size_t(small::BLOCK) || (samples % 16) != 0 ||
            samples >= (size_t(1) << 28))
            return 1;
        for (int k = 0; k < count; ++k)                     // a bank in the exact mode runs its own launch (biquad_exact_kernel)
            if (stages[k].bank != nullptr && stages[k].bank->exact)
                return 1;
        // A long call (four sub-blocks of 2048 samples and more) is a stream of sub-blocks like a run of blocks: the stream
        // kernel walks it with four waves per channel, hand-over for hand-over what the super-block loop below does.
        if (long_calls_as_streams && samples >= 4 * size_t(big::BLOCK))
        {
            int slot[CHAIN_MAX], outs = 0;
            float *po[CHAIN_MAX];
            size_t stride = 0;
            bool same = true;
            for (int k = 0; k < count; ++k)
            {
                slot[k] = -1;
                if (stages[k].out == nullptr)
                    continue;
                same = same && (outs == 0 || stages[k].out_stride == stride);
                stride = stages[k].out_stride;
                po[outs] = stages[k].out;
                slot[k] = outs++;
            }
            if (same && outs > 0)
            {
                const int r = biquad_chain_process_blocks(stages, slot, count, outs, po, &in, 1, samples, stride, in_stride, st);
                if (r != 1)
                    return r;
            }
        }
        const uint32_t channels = stages[0].bank->channels;
        bool aligned = (reinterpret_cast<uintptr_t>(in) % 16 == 0) && (in_stride % 4 == 0) && in_stride >= samples;
        for (int k = 0; k < count; ++k)
        {
            const mi_biquad_bank *b = stages[k].bank;
            if (b == nullptr || b->channels != channels)
                return 1;
            if (stages[k].out != nullptr)
                aligned = aligned && (reinterpret_cast<uintptr_t>(stages[k].out) % 16 == 0) &&
                          (stages[k].out_stride % 4 == 0) && stages[k].out_stride >= samples;
        }
        if (!aligned)
            return 1;
        for (uint32_t c = 0; c < channels; ++c)
        {
            size_t total = 0;
            for (int k = 0; k < count; ++k)
            {
                if (stages[k].bank->row_off[c] || stages[k].bank->nsec[c] == 0)
                    return 1;
                total += stages[k].bank->nsec[c];
            }
            if (total > size_t(big::SG))
                return 1;
        }
        chain_args a;
        a.stages = count;
        for (int k = 0; k < count; ++k)
        {
            mi_biquad_bank *b = stages[k].bank;
            const int r = commit(b, st);
            if (r != MI_OK)
                return r;
            a.st[k].tab = b->d_big;
            a.st[k].state = b->d_state;
            a.st[k].nsec = b->d_nsec;
            a.st[k].out = stages[k].out;
            a.st[k].out_stride = stages[k].out_stride;
            a.st[k].max_sec = int(b->max_sec);
            a.st[k].branch = stages[k].branch ? 1 : 0;
        }
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        take_profile_events(&ev0, &ev1);
        MI_LAUNCH((biquad_chain_kernel<16, 2, true>), dim3(channels), dim3(128), 0, st, ev0, ev1, in, in_stride, int(samples), a);
        MI_HIP_CHECK(hipGetLastError());
        return MI_OK;
    }

    // The chain over a run of blocks (see mi_common.h).  A block joins the run under the rules of
    // mi_biquad_bank_process_blocks: nothing a block of the run writes is read or written by another one (except the very
    // same rows through the same wave: an output that comes round again a multiple of the waves' stride later), nothing one
    // of them reads is written by another; a block's output may be its own input.
    int biquad_chain_process_blocks(const biquad_chain_stage *stages, const int *slot, int count, int outs,
                                    float *const *out, const float *const *in, size_t blocks, size_t samples,
                                    size_t out_stride, size_t in_stride, hipStream_t st)
    {
        constexpr int NW = 4;                               // (two for the shortest runs; the rules below hold for both)
        if (count <= 0 || count > CHAIN_MAX || outs <= 0 || stages[count - 1].branch || samples <= 2 * size_t(small::BLOCK) || (samples % 16) != 0 ||
            samples >= (size_t(1) << 28) || (out_stride % 4) != 0 || (in_stride % 4) != 0 || out_stride < samples || in_stride < samples)
            return 1;
        const uint32_t channels = stages[0].bank->channels;
        for (int k = 0; k < count; ++k)
            if (stages[k].bank == nullptr || stages[k].bank->channels != channels || slot[k] >= outs || stages[k].bank->exact)
                return 1;
        int sec_cap = 1;
        for (uint32_t c = 0; c < channels; ++c)
        {
            size_t total = 0;
            for (int k = 0; k < count; ++k)
            {
                if (stages[k].bank->row_off[c] || stages[k].bank->nsec[c] == 0)
                    return 1;
                total += stages[k].bank->nsec[c];
            }
            if (total > size_t(STREAM_SG))
                return 1;
            sec_cap = std::max(sec_cap, int(total));
        }
        for (size_t i = 0; i < blocks; ++i)
        {
            if (in[i] == nullptr || (reinterpret_cast<uintptr_t>(in[i]) % 16) != 0)
                return 1;
            for (int s = 0; s < outs; ++s)
                if (out[i * outs + s] == nullptr || (reinterpret_cast<uintptr_t>(out[i * outs + s]) % 16) != 0)
                    return 1;
        }
        stream_chain_args a;
        a.stages = count;
        a.outs = outs;
        for (int k = 0; k < count; ++k)
        {
            mi_biquad_bank *b = stages[k].bank;
            const int r = commit(b, st);
            if (r != MI_OK)
                return r;
            a.st[k].tab = b->d_big;
            a.st[k].state = b->d_state;
            a.st[k].nsec = b->d_nsec;
            a.st[k].max_sec = int(b->max_sec);
            a.st[k].slot = slot[k];
            a.st[k].branch = stages[k].branch ? 1 : 0;
        }
        // (test knob: a launch per block.  Read per call like mi_biquad_bank_process / _process_blocks read it, so that a test
        // that flips it in-process switches the bank path and the chain path together: ADVICE r04)
        const bool loop = mi::test_path("blocks_loop");
        const size_t spb = (samples + big::BLOCK - 1) / big::BLOCK;
        const size_t out_bytes = (size_t(channels - 1) * out_stride + samples) * sizeof(float);
        const size_t in_bytes  = (size_t(channels - 1) * in_stride + samples) * sizeof(float);
        auto overlap = [](const void *p, size_t pn, const void *q, size_t qn) -> bool {
            const uintptr_t a0 = reinterpret_cast<uintptr_t>(p), b0 = reinterpret_cast<uintptr_t>(q);
            return a0 < b0 + qn && b0 < a0 + pn;
        };
        auto joins = [&](size_t first, size_t k) -> bool {
            if (loop)
                return false;
            for (int s = 0; s < outs; ++s)                  // a block's own outputs: apart from each other, and from its input unless equal
            {
                const float *o = out[k * outs + s];
                if (o != in[k] && overlap(o, out_bytes, in[k], in_bytes))
                    return false;
                for (int q = 0; q < s; ++q)
                    if (overlap(o, out_bytes, out[k * outs + q], out_bytes))
                        return false;
            }
            for (size_t i = first; i < k; ++i)
                for (int s = 0; s < outs; ++s)
                {
                    if (overlap(out[i * outs + s], out_bytes, in[k], in_bytes) || overlap(in[i], in_bytes, out[k * outs + s], out_bytes))
                        return false;
                    for (int q = 0; q < outs; ++q)
                        if (overlap(out[i * outs + s], out_bytes, out[k * outs + q], out_bytes) &&
                            (out[i * outs + s] != out[k * outs + q] || (((k - i) * spb) % NW) != 0))
                            return false;
                }
            return true;
        };
        const size_t cap = std::min<size_t>(size_t(STREAM_CHAIN_BLOCKS), size_t(STREAM_CHAIN_PTRS) / size_t(outs));
        size_t first = 0;
        while (first < blocks)
        {
            size_t n = 0;
            while (first + n < blocks && n < cap && joins(first, first + n))
                ++n;
            if (n >= 2 || (n == 1 && spb >= 4 && !loop))
            {
                a.blocks = int(n);
                for (size_t i = 0; i < n; ++i)
                {
                    a.in[i] = in[first + i];
                    for (int s = 0; s < outs; ++s)
                        a.out[i * outs + s] = out[(first + i) * outs + s];
                }
                hipEvent_t ev0 = nullptr, ev1 = nullptr;
                take_profile_events(&ev0, &ev1);
                const int nw = (n * spb >= 4) ? 4 : 2;
                const bool qlds = stream_qlds(nw, sec_cap);
                const size_t lds = stream_lds(nw, sec_cap, qlds);
                #define MI_STREAM(NWV, Q) MI_LAUNCH((biquad_stream_chain_kernel<NWV, Q>), dim3(channels), dim3(64 * NWV), lds, st, ev0, ev1, a, \
                                                    out_stride, in_stride, int(samples), sec_cap)
                if (nw == 4) { if (qlds) MI_STREAM(4, true); else MI_STREAM(4, false); }
                else         { if (qlds) MI_STREAM(2, true); else MI_STREAM(2, false); }
                #undef MI_STREAM
                MI_HIP_CHECK(hipGetLastError());
            }
            else
            {
                n = 1;                                      // (also a block whose own buffers overlap: joins() turned it down)
                biquad_chain_stage one[CHAIN_MAX];
                for (int k = 0; k < count; ++k)
                {
                    one[k] = stages[k];
                    one[k].out = (slot[k] >= 0) ? out[first * outs + slot[k]] : nullptr;
                    one[k].out_stride = out_stride;
                }
                const int r = biquad_chain_process(one, count, in[first], in_stride, samples, st, false);
                if (r != MI_OK)
                    return (r == 1) ? fail(MI_ESTATE, "biquad_chain_process_blocks: a block of a qualified run was turned down") : r;
            }
            first += n;
        }
        return MI_OK;
    }
} // namespace mi

static int bank_run(mi_biquad_bank_t *b, float *out, const float *in, size_t samples, size_t out_stride, size_t in_stride,
                    hipStream_t st, const sumsq_args *sq, const mi_meters::ilufs_epilogue *ep = nullptr, bool *rode = nullptr);

namespace mi
{
    // FilterBank::impulse_response (FilterBank.cpp:293-330) in the reference's own operation order (see
    // biquad_reference_ir_kernel): the filter memory is not touched (the response starts from a zeroed one by definition).
    int biquad_bank_reference_impulse_response(mi_biquad_bank *b, float *out, size_t samples, size_t out_stride, hipStream_t st)
    {
        MI_REQUIRE(b != nullptr && out != nullptr && out_stride >= samples && samples < (size_t(1) << 29), MI_EINVAL,
                   "biquad_bank_reference_impulse_response: bad argument");
        if (samples == 0)
            return MI_OK;
        const int r = commit(b, st);
        if (r != MI_OK)
            return r;
        hipLaunchKernelGGL(biquad_reference_ir_kernel, dim3(b->channels), dim3(64), 0, st, out, out_stride, int(samples),
                           b->d_small, int(small::TAB), b->d_nsec, int(b->max_sec));
        MI_HIP_CHECK(hipGetLastError());
        return MI_OK;
    }

    // The bank over `samples` samples without an output: sums[channel * 4 + s] += the sum of the squares of the filtered
    // samples [seg_end[s - 1], seg_end[s]) (seg_end[3] = samples).  A channel switched off adds nothing.
    // `ep` != NULL: the integrated meter's bookkeeping for this call; *rode tells whether it went with the launch (then
    // the caller has nothing left to do) or the call did not qualify (the caller launches its own kernel behind this one).
    void biquad_bank_output_reread(mi_biquad_bank *b, bool yes) { if (b != nullptr) b->out_reread = yes; }

    int biquad_bank_sumsq(mi_biquad_bank *b, const float *in, size_t in_stride, size_t samples, const uint32_t seg_end[3],
                          float *sums, hipStream_t st, const mi_meters::ilufs_epilogue *ep, bool *rode)
    {
        if (rode != nullptr)
            *rode = false;
        if (samples == 0)
            return MI_OK;
        MI_REQUIRE(b != nullptr && in != nullptr && sums != nullptr && in_stride >= samples && samples < (size_t(1) << 31),
                   MI_EINVAL, "biquad_bank_sumsq: bad argument");
        sumsq_args sq;
        sq.sums = sums;
        sq.e0 = int(std::min<size_t>(seg_end[0], samples));
        sq.e1 = int(std::min<size_t>(seg_end[1], samples));
        sq.e2 = int(std::min<size_t>(seg_end[2], samples));
        return bank_run(b, nullptr, in, samples, 0, in_stride, st, &sq, ep, rode);
    }
} // namespace mi

extern "C" {

static std::atomic<int> g_exact_default{0};     // mi_dspu_set_exact_iir_default: what banks made from now on start with

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
    // (MI_DSPU_EXACT_IIR=1: the same default for hosts that cannot be changed to call mi_dspu_set_exact_iir_default)
    const char *const exact_env = getenv("MI_DSPU_EXACT_IIR");
    b->exact = g_exact_default.load() != 0 || (exact_env != nullptr && atoi(exact_env) != 0);
    b->max_sec  = max_sections;
    const size_t cs = size_t(channels) * max_sections;
    try
    {
        b->nsec.assign(channels, 0);
        b->row_off.assign(channels, 0);
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

int mi_biquad_bank_set_row_enabled(mi_biquad_bank_t *b, uint32_t channel, int enabled)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_biquad_bank_set_row_enabled: NULL bank");
    MI_REQUIRE(channel < b->channels, MI_EINVAL, "mi_biquad_bank_set_row_enabled: channel %u out of range", channel);
    const uint8_t off = enabled ? 0 : 1;
    if (b->row_off[channel] == off)
        return MI_OK;
    b->row_off[channel] = off;
    b->nsec_dirty = true;
    b->pending = true;
    return MI_OK;
}

int mi_dspu_last_stream_clock(double *ghz, double *microseconds)
{
    MI_REQUIRE(ghz != nullptr, MI_EINVAL, "mi_dspu_last_stream_clock: NULL result pointer");
    unsigned long long h[4] = { 0, 0, 0, 0 };
    MI_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stream_clock), sizeof(h)));      // (synchronises with the device)
    MI_REQUIRE(h[3] > h[1] && h[2] > h[0], MI_ESTATE, "mi_dspu_last_stream_clock: no mi_biquad_bank_process_blocks launch has run yet");
    const double us = double(h[3] - h[1]) / 100.0;          // wall_clock64: 100 MHz
    *ghz = double(h[2] - h[0]) / us / 1e3;
    if (microseconds != nullptr)
        *microseconds = us;
    return MI_OK;
}

int mi_dspu_set_exact_iir_default(int on)
{
    g_exact_default.store(on ? 1 : 0);
    return MI_OK;
}

int mi_biquad_bank_set_exact(mi_biquad_bank_t *b, int on)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_biquad_bank_set_exact: NULL bank");
    b->exact = on != 0;
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

static int stream_launch(mi_biquad_bank_t *b, float *const *out, const float *const *in, size_t first, size_t count,
                         size_t samples, size_t out_stride, size_t in_stride, hipStream_t st);

// The call on biquad_exact_kernel (mi_biquad_bank_set_exact): G lanes per channel, the power of two that holds the longest cascade
static int exact_run(mi_biquad_bank_t *b, float *out, const float *in, size_t samples, size_t out_stride, size_t in_stride, hipStream_t st)
{
    int cap = 0;
    for (uint32_t c = 0; c < b->channels; ++c)
        cap = std::max(cap, int(b->nsec[c]));
    int g = 1;
    while (g < cap && g < 64)
        g *= 2;
    size_t done = 0;
    while (done < samples)
    {
        const size_t step = std::min<size_t>(samples - done, size_t(1) << 28);
        const dim3 grid((b->channels * unsigned(g) + 63) / 64);
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        mi::take_profile_events(&ev0, &ev1);
        #define MI_EXACT(G_) MI_LAUNCH((biquad_exact_kernel<G_>), grid, dim3(64), 0, st, ev0, ev1, out + done, in + done, out_stride, in_stride, \
                                       int(step), b->d_small, int(small::TAB), b->d_state, b->d_nsec, int(b->max_sec), int(b->channels), std::max(cap, 1))
        switch (g)
        {
            case 1:  { MI_EXACT(1);  break; }
            case 2:  { MI_EXACT(2);  break; }
            case 4:  { MI_EXACT(4);  break; }
            case 8:  { MI_EXACT(8);  break; }
            case 16: { MI_EXACT(16); break; }
            case 32: { MI_EXACT(32); break; }
            default: { MI_EXACT(64); break; }
        }
        #undef MI_EXACT
        MI_HIP_CHECK(hipGetLastError());
        done += step;
    }
    return MI_OK;
}

// One call of the bank over `samples` samples of every channel; sq != NULL: the meters' epilogue instead of the output
static int bank_run(mi_biquad_bank_t *b, float *out, const float *in, size_t samples, size_t out_stride, size_t in_stride,
                    hipStream_t st, const sumsq_args *sq, const mi_meters::ilufs_epilogue *ep, bool *rode)
{
    int r = commit(b, st);
    if (r != MI_OK)
        return r;
    if (b->exact && sq == nullptr)
        return exact_run(b, out, in, samples, out_stride, in_stride, st);

    const bool aligned = ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(in)) % 16 == 0) &&
                         (out_stride % 4 == 0) && (in_stride % 4 == 0);
    // One launch walks the whole call.  The variant sets the chunk length and how many waves share a channel:
    // a wave owns a sub-block of 64 x 2L samples, NW waves cover NW consecutive sub-blocks at once.
    //   <= 1024 samples: L = 8, one wave;  <= 2048: L = 8, two waves;  longer: L = 16, two waves.
    const bool use_small = samples <= 2 * size_t(small::BLOCK);
    const size_t chunk = use_small ? 8 : 16;
    const size_t tail  = samples % chunk;                   // < chunk samples, done by biquad_tail_kernel
    const size_t body  = samples - tail;
    // the meter's bookkeeping rides on the launch when the launch is the whole call: the long-call variant, no tail kernel
    // behind it (whose sums would come too late), one launch
    const bool no_ride = ep != nullptr && getenv("MI_ILUFS_TWO_LAUNCHES") != nullptr;       // (fall-back: the hand-over inside the launch rests on gfx950 behaviour)
    const bool ride = sq != nullptr && ep != nullptr && !use_small && tail == 0 && body < (size_t(1) << 28) && !no_ride;
    if (rode != nullptr)
        *rode = ride;
    // the stereo meter in one workgroup (biquad_sumsq_ilufs_pair_kernel): rows in pairs, all of them enabled and alike in
    // their number of sections (the workgroup's barriers are shared by its two rows)
    bool pair_ok = ride && (ep->channels == 1 || (ep->channels == 2 && (b->channels % 2) == 0)) && !mi::test_path("ilufs_rows_apart");
    for (uint32_t c = 0; pair_ok && c < b->channels; ++c)
        pair_ok = !b->row_off[c] && b->nsec[c] == b->nsec[0];
    size_t done = 0;
    while (done < body)
    {
        const size_t left = body - done;
        const size_t step = (left < (size_t(1) << 28)) ? left : (size_t(1) << 28);      // multiple of 16
        sumsq_args part, *pq = nullptr;
        if (sq != nullptr)                                  // segment ends counted from this launch's first sample
        {
            auto rel = [&](int e) -> int { return (size_t(e) <= done) ? 0 : (size_t(e) - done >= step) ? int(step) : int(size_t(e) - done); };
            part.sums = sq->sums; part.e0 = rel(sq->e0); part.e1 = rel(sq->e1); part.e2 = rel(sq->e2);
            pq = &part;
        }
        float *o = (out != nullptr) ? out + done : nullptr;
        hipError_t e;
        if (samples <= size_t(small::BLOCK))
            e = launch<8, 1>(b, o, in + done, out_stride, in_stride, int(step), aligned, b->d_small, st, pq);
        else if (use_small)
            e = launch<8, 2>(b, o, in + done, out_stride, in_stride, int(step), aligned, b->d_small, st, pq);
        else
            e = launch<16, 2>(b, o, in + done, out_stride, in_stride, int(step), aligned, b->d_big, st, pq, ride ? ep : nullptr,
                              ride && pair_ok);
        MI_HIP_CHECK(e);
        done += step;
    }
    if (tail > 0)
    {
        hipLaunchKernelGGL(biquad_tail_kernel, dim3((b->channels + 63) / 64), dim3(64), 0, st, out, in, out_stride,
                           in_stride, body, int(tail), b->d_small, int(small::TAB), b->d_state, b->d_nsec,
                           int(b->max_sec), b->channels, sq ? *sq : sumsq_args{ nullptr, 0, 0, 0 });
        MI_HIP_CHECK(hipGetLastError());
    }
    return MI_OK;
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
    // A long call (four sub-blocks of 2048 samples and more) is a stream of sub-blocks like a run of blocks: the stream kernel
    // walks it with four waves per channel (7.7 instead of 9.2 us per 4096 samples at 65536-sample calls) -- the same bits as
    // the super-block loop of biquad_bank_kernel, whose hand-overs it reproduces kind for kind.
    if (samples >= 4 * size_t(big::BLOCK) && (samples % 16) == 0 && samples < (size_t(1) << 28) && (out_stride % 4) == 0 &&
        (in_stride % 4) == 0 && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(in)) % 16) == 0 && !b->exact &&
        !mi::test_path("blocks_loop"))
    {
        bool fits = true;
        for (uint32_t c = 0; fits && c < b->channels; ++c)
            fits = b->nsec[c] <= uint32_t(STREAM_SG);
        if (fits)
        {
            const int r = commit(b, mi::as_stream(stream));
            if (r != MI_OK)
                return r;
            float *po[1] = { out };
            const float *pi[1] = { in };
            return stream_launch(b, po, pi, 0, 1, samples, out_stride, in_stride, mi::as_stream(stream));
        }
    }
    return bank_run(b, out, in, samples, out_stride, in_stride, mi::as_stream(stream), nullptr);
}

// Blocks [first, first + count) of a process_blocks call as ONE launch of biquad_stream_kernel
static int stream_launch(mi_biquad_bank_t *b, float *const *out, const float *const *in, size_t first, size_t count,
                         size_t samples, size_t out_stride, size_t in_stride, hipStream_t st)
{
    stream_args a;
    a.blocks = int(count);
    for (size_t k = 0; k < count; ++k)
    {
        a.out[k] = out[first + k];
        a.in[k]  = in[first + k];
    }
    const size_t subs = count * ((samples + big::BLOCK - 1) / big::BLOCK);
    const int nw = (subs >= 4) ? 4 : 2;
    const dim3 grid(b->channels);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    mi::take_profile_events(&ev0, &ev1);
    int cap = 1;                                            // sections the launch's cells (and scan operands) must hold
    for (uint32_t c = 0; c < b->channels; ++c)
        cap = std::max(cap, int(b->nsec[c]));
    const bool qlds = stream_qlds(nw, cap);
    const size_t lds = stream_lds(nw, cap, qlds);
    #define MI_STREAM(NWV, Q) MI_LAUNCH((biquad_stream_kernel<NWV, Q>), grid, dim3(64 * NWV), lds, st, ev0, ev1, a, out_stride, in_stride, \
                                        int(samples), b->d_big, b->d_state, b->d_nsec, int(b->max_sec), cap)
    if (nw == 4)      { if (qlds) MI_STREAM(4, true); else MI_STREAM(4, false); }
    else if (nw == 2) { if (qlds) MI_STREAM(2, true); else MI_STREAM(2, false); }
    else              { if (qlds) MI_STREAM(2, true); else MI_STREAM(2, false); }
    #undef MI_STREAM
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

int mi_biquad_bank_process_blocks(mi_biquad_bank_t *b, float *const *out, const float *const *in, size_t blocks,
                                  size_t samples, size_t out_stride, size_t in_stride, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_biquad_bank_process_blocks: NULL bank");
    if (samples == 0 || blocks == 0)
        return MI_OK;
    MI_REQUIRE(out != nullptr && in != nullptr, MI_EINVAL, "mi_biquad_bank_process_blocks: NULL pointer table");
    MI_REQUIRE(out_stride >= samples && in_stride >= samples, MI_EINVAL,
               "mi_biquad_bank_process_blocks: stride shorter than the block");
    for (size_t k = 0; k < blocks; ++k)
        MI_REQUIRE(out[k] != nullptr && in[k] != nullptr, MI_EINVAL, "mi_biquad_bank_process_blocks: NULL buffer of block %zu", k);
    hipStream_t st = mi::as_stream(stream);
    int r = commit(b, st);
    if (r != MI_OK)
        return r;

    // One launch for a run of blocks (biquad_stream_kernel) where the blocks are what the long-call kernel takes -- more
    // than 2048 samples, whole chunks of 16, 16-byte aligned rows -- and every channel's sections have a hand-over cell.
    bool streamable = samples > 2 * size_t(small::BLOCK) && (samples % 16) == 0 && samples < (size_t(1) << 28) && !b->exact &&
                      (out_stride % 4) == 0 && (in_stride % 4) == 0 && !mi::test_path("blocks_loop");
    for (uint32_t c = 0; streamable && c < b->channels; ++c)
        streamable = b->nsec[c] <= uint32_t(STREAM_SG);
    const size_t spb = (samples + big::BLOCK - 1) / big::BLOCK;
    const size_t out_bytes = (size_t(b->channels - 1) * out_stride + samples) * sizeof(float);
    const size_t in_bytes  = (size_t(b->channels - 1) * in_stride + samples) * sizeof(float);
    auto overlap = [](const void *p, size_t pn, const void *q, size_t qn) -> bool {
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(p), b0 = reinterpret_cast<uintptr_t>(q);
        return a0 < b0 + qn && b0 < a0 + pn;
    };
    // Inside a launch a block's loads are issued while blocks before it are still computing, and its stores are not
    // ordered against another wave's: a block joins the run only if it neither reads nor overwrites what a block of the
    // run writes (except the same rows through the same wave: an output buffer that comes round again a multiple of four
    // sub-blocks later), and does not write what one of them reads.  A block may be processed in place.
    auto joins = [&](size_t first, size_t k) -> bool {
        if (!streamable || ((reinterpret_cast<uintptr_t>(out[k]) | reinterpret_cast<uintptr_t>(in[k])) % 16) != 0)
            return false;
        for (size_t i = first; i < k; ++i)
        {
            if (overlap(out[i], out_bytes, in[k], in_bytes) || overlap(in[i], in_bytes, out[k], out_bytes))
                return false;
            if (overlap(out[i], out_bytes, out[k], out_bytes) && (out[i] != out[k] || (((k - i) * spb) % 4) != 0))
                return false;
        }
        return true;
    };
    size_t first = 0;
    while (first < blocks)
    {
        size_t count = 0;
        while (first + count < blocks && count < size_t(STREAM_MAX_BLOCKS) && joins(first, first + count))
            ++count;
        if (count >= 2)
            r = stream_launch(b, out, in, first, count, samples, out_stride, in_stride, st);
        else
        {
            count = 1;
            r = bank_run(b, out[first], in[first], samples, out_stride, in_stride, st, nullptr);
        }
        if (r != MI_OK)
            return r;
        first += count;
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
        geometry[0] = 16; geometry[1] = 64; geometry[2] = 16; geometry[3] = big::TAB;
        if (table != nullptr)
            fill_row<16>(table, q);
    }
    else
    {
        geometry[0] = 8; geometry[1] = 64; geometry[2] = 16; geometry[3] = small::TAB;
        if (table != nullptr)
            fill_row<8>(table, q);
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
