// Device side of the integrated loudness meter's bookkeeping (lsp::dspu::ILUFSMeter, reference:
// src/main/meters/ILUFSMeter.cpp:324-353 gated / infinite loudness, :355-470 process), shared by two launches:
//   * loudness.hip's ilufs_call_kernel: one workgroup of 256 threads per meter, behind the weighting filter's launch;
//   * biquad.hip's biquad_sumsq_kernel<.., true>: the weighting filter's own launch, where the workgroup that is the
//     LAST of a meter's rows to leave its sums of squares does the meter's bookkeeping (one launch per call instead of
//     two); its 128 threads then stand in for the 256.  Hand-over without fences: the sums are agent-scope atomics
//     (performed at the memory side, not in an XCD's L2), every adding thread waits for their completion before the
//     workgroup counts itself in, and the reader takes them with agent-scope loads (an agent-scope release fence per
//     workgroup -- an L2 write-back each -- took the step from 38 to 60 us).
// Every sum below is laid out over VTH = 256 "virtual" threads in four virtual waves, whatever the number of real
// threads: the two launches give the same floats bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mi_meters
{
    struct chan_cfg
    {
        float   weight;
        float   link;
        int     enabled;
        int     unbound;        // LoudnessMeter: no input bound -- the channel is left out of the block (:421-422) but stays
    };                          // enabled for refresh_rms() and clear(); always 0 for the integrated meter

    constexpr float GATING_ABS_THRESH = 1.17246530458e-07f;         // ILUFSMeter.cpp:39
    constexpr uint32_t MIN_GATING_BLOCKS = 64;                      // :55
    constexpr int VTH = 256;                                        // virtual threads per meter (the order of the sums)

    struct ilufs_state { uint32_t head, count; float loudness; uint32_t pad; };

    // The pieces of one process() call.  A piece is a run of samples inside one quarter of a gating block; the weighting
    // filter has left the sum of squares of every row and piece in seg[row][piece].  Piece by piece: the held loudness
    // value into the output row (ILUFSMeter.cpp:386-387), vBlock[row][part] += the piece's sum for every enabled row
    // (:372-384), then -- when the piece ends the quarter -- the gating arithmetic of a complete block and the reset of
    // the quarter that is filled next (:402-466).
    struct ilufs_piece { uint32_t offset, n, part; int gate, zero_part; };
    struct ilufs_pieces { uint32_t count; ilufs_piece p[4]; };

    // What the last-arriving workgroup of the weighting filter's launch needs for a meter's bookkeeping
    struct ilufs_epilogue
    {
        uint32_t       *arrived;        // [meters] rows of the meter that have left their sums (back to 0 when the last one has)
        float          *block;          // [rows][4] the quarters of the gating block in hand
        const chan_cfg *cfg;
        uint32_t        channels;
        float          *out;
        size_t          out_stride;
        ilufs_state    *st;
        float           gain;
        float          *hist;
        uint32_t        size, ms_int;
        float           avg;
        ilufs_pieces    pieces;
    };

#if defined(__HIPCC__)
    // the four virtual waves' partial sums -> one number, in the order ((w0 + w1) + w2) + w3
    template <int TT>
    __device__ __forceinline__ void virtual_wave_sums(float (&s)[VTH / TT], float *s_sum)
    {
        const uint32_t tid = threadIdx.x;
        #pragma unroll
        for (int r = 0; r < VTH / TT; ++r)
        {
            float v = s[r];
            #pragma unroll
            for (int w = 32; w > 0; w >>= 1)
                v += __shfl_xor(v, w);
            if ((tid & 63) == 0)
                s_sum[(tid >> 6) + r * (TT / 64)] = v;
        }
    }

    // What a meter's bookkeeping reads that the call in hand does NOT produce -- its state, the quarters of the block in
    // hand, the channel settings, and the history as the call's first complete block will see it -- is asked for EARLY:
    // by every workgroup of the filter's launch before it waits for its own sums to be performed (whichever turns out to
    // be the meter's last has the answers by then), by ilufs_call_kernel at its start.  What is left behind the count is
    // one round trip for the rows' sums; the bookkeeping itself then runs out of registers and LDS (it used to read its
    // own stores back three times: the quarters, the state, the history entry it had just appended -- a microsecond each on
    // the tail of the launch).
    template <int TT>
    struct ilufs_early
    {
        ilufs_state me;
        float4      blk;                // tid < channels: the row's quarters
        float       weight;
        int         counts;
        float       hv[VTH / TT];       // history entry j = tid + r TT of the first gate's window (the newest is not there yet)
        int         hist_ok;            // ... valid: finite integration, at most one entry per virtual thread
    };

    template <int TT>
    __device__ __forceinline__ ilufs_early<TT> ilufs_ask(uint32_t meter, const float *block, const chan_cfg *__restrict__ cfg,
                                                         uint32_t channels, const ilufs_state *st, const float *hist,
                                                         uint32_t size, uint32_t ms_int)
    {
        ilufs_early<TT> e;
        const uint32_t tid = threadIdx.x;
        e.me = st[meter];
        e.blk = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        e.weight = 0.0f;
        e.counts = 0;
        if (tid < channels)
        {
            e.blk = *reinterpret_cast<const float4 *>(block + (size_t(meter) * channels + tid) * 4);
            e.weight = cfg[tid].weight;
            e.counts = cfg[tid].enabled != 0;
        }
        // the window of the gate that follows: count' = min(count + 1, ms_int) entries that end with the one the gate appends
        const uint32_t count = (e.me.count + 1 < ms_int) ? e.me.count + 1 : ms_int;
        uint32_t head = e.me.head + 1;
        head = (head >= size) ? head - size : head;
        uint32_t tail = head + size - count;
        tail = (tail >= size) ? tail - size : tail;
        e.hist_ok = (ms_int > 0 && count <= uint32_t(VTH) && count <= size && e.me.head < size) ? 1 : 0;
        #pragma unroll
        for (int r = 0; r < VTH / TT; ++r)
        {
            const uint32_t j = tid + r * TT;
            uint32_t at = tail + j;
            at = (at >= size) ? at - size : at;
            e.hv[r] = (e.hist_ok && j + 1 < count) ? hist[size_t(meter) * size + at] : 0.0f;
        }
        return e;
    }

    // mean of the last `count` history entries above the ABSOLUTE gate (compute_gated_loudness, ILUFSMeter.cpp:324-341:
    // its `threshold` argument is not used by the reference -- both gating stages compare with GATING_ABS_THRESH, so the
    // relative stage returns what the absolute stage returned; one pass gives the reference's result for both)
    // early != nullptr: the window's entries are in registers (ilufs_ask), the newest of them is `newest`
    template <int TT>
    __device__ float gated_mean(const float *hist, uint32_t size, uint32_t head, uint32_t count, float *s_sum, uint32_t *s_cnt,
                                const float *early = nullptr, float newest = 0.0f)
    {
        static_assert(TT == 128 || TT == 256, "two or four real waves");
        constexpr int R = VTH / TT;
        const uint32_t tid = threadIdx.x;
        const uint32_t tail = (head + size - count) % size;
        float s[R];
        uint32_t c[R];
        #pragma unroll
        for (int r = 0; r < R; ++r)
        {
            s[r] = 0.0f;
            c[r] = 0;
            if (early != nullptr)                           // (count <= VTH: one entry per virtual thread)
            {
                const uint32_t j = tid + r * TT;
                const float l = (j + 1 == count) ? newest : early[r];
                if (j < count && l > GATING_ABS_THRESH)
                {
                    s[r] += l;
                    ++c[r];
                }
                continue;
            }
            for (uint32_t j = tid + r * TT; j < count; j += VTH)
            {
                const float l = hist[(tail + j) % size];
                if (l > GATING_ABS_THRESH)
                {
                    s[r] += l;
                    ++c[r];
                }
            }
        }
        // the waves' sums by shuffles, the four virtual ones through LDS
        virtual_wave_sums<TT>(s, s_sum);
        #pragma unroll
        for (int r = 0; r < R; ++r)
        {
            uint32_t v = c[r];
            #pragma unroll
            for (int w = 32; w > 0; w >>= 1)
                v += __shfl_xor(v, w);
            if ((tid & 63) == 0)
                s_cnt[(tid >> 6) + r * (TT / 64)] = v;
        }
        __syncthreads();
        if (tid == 0)
        {
            s_sum[0] = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
            s_cnt[0] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        }
        __syncthreads();
        const float r = (s_cnt[0] > 0) ? s_sum[0] / float(s_cnt[0]) : 0.0f;
        __syncthreads();
        return r;
    }

    // a gating block is complete (ILUFSMeter.cpp:402-458); the workgroup of the meter.  `me`: the meter's state, carried in
    // registers from gate to gate of a call (every thread holds the same); blk / weight: the thread's row (tid < channels) as
    // the pieces have left it; early: ilufs_ask's history window, for the first gate of a call.
    template <int TT>
    __device__ float ilufs_gate(uint32_t meter, ilufs_state *st, ilufs_state &me, float *hist, uint32_t size, uint32_t ms_int,
                                const float *block, const float4 blk, const float weight,
                                const chan_cfg *__restrict__ cfg, uint32_t channels, float avg,
                                float *s_sum, uint32_t *s_cnt, float &s_val, float *s_chan /* [2 TT] */, const float *early)
    {
        constexpr int R = VTH / TT;
        const uint32_t tid = threadIdx.x;
        float *h = hist + size_t(meter) * size;
        if (channels <= uint32_t(TT))                       // the rows' quarters are in their threads' registers
        {
            if (tid < channels)
            {
                s_chan[tid] = (blk.x + blk.y + blk.z + blk.w) * avg;
                s_chan[TT + tid] = weight;
            }
            __syncthreads();
        }
        if (tid == 0)
        {
            float loudness = 0.0f;                          // every channel's block enters, enabled or not (:407-414)
            if (channels <= uint32_t(TT))
                for (uint32_t c = 0; c < channels; ++c)
                    loudness += s_chan[TT + c] * s_chan[c];
            else
                for (uint32_t c = 0; c < channels; ++c)
                {
                    const float *blk4 = block + (size_t(meter) * channels + c) * 4;
                    loudness += cfg[c].weight * ((blk4[0] + blk4[1] + blk4[2] + blk4[3]) * avg);
                }
            s_val = loudness;
        }
        __syncthreads();
        float loudness = s_val;
        __syncthreads();
        if (ms_int > 0)                                     // finite integration period
        {
            me.count = (me.count + 1 < ms_int) ? me.count + 1 : ms_int;
            if (tid == 0)
                h[me.head] = loudness;
            me.head = (me.head + 1) % size;
            if (early == nullptr)
                __syncthreads();                            // the new entry is read back with the others
            loudness = gated_mean<TT>(h, size, me.head, me.count, s_sum, s_cnt, early, loudness);
        }
        else                                                // since the last clear(): running mean of the gated blocks
        {
            if (loudness > GATING_ABS_THRESH)
            {
                if (me.count >= 0x100)                      // floating-point overflow protection (:440-444)
                {
                    for (uint32_t j = tid; j < size; j += TT)
                        h[j] *= 0.5f;
                    me.count >>= 1;
                    __syncthreads();
                }
                ++me.count;
                if (tid == 0)
                    h[me.head] += loudness;
                me.head = (me.head + 1) % size;
                __syncthreads();
            }
            if (me.count > 0)                               // compute_infinite_loudness: sum of hist[j] / count
            {
                const float mult = 1.0f / float(me.count);
                float s[R];
                #pragma unroll
                for (int r = 0; r < R; ++r)
                {
                    s[r] = 0.0f;
                    for (uint32_t j = tid + r * TT; j < size; j += VTH)
                        s[r] += mult * h[j];
                }
                virtual_wave_sums<TT>(s, s_sum);
                __syncthreads();
                if (tid == 0)
                    s_sum[0] = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
                __syncthreads();
                loudness = s_sum[0];
            }
            else
                loudness = 0.0f;
        }
        me.loudness = sqrtf(loudness);
        if (tid == 0)
            st[meter] = me;
        return me.loudness;                                 // every thread has it: the next piece's output starts at once
    }

    // seg[] cell; FRESH: read at the memory side (agent scope), for the launch in which other workgroups, possibly on
    // another XCD with an L2 of its own, have only just added to it
    template <bool FRESH>
    __device__ __forceinline__ float seg_load(const float *p)
    {
        return FRESH ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
    }

    // a run of the output row = the value being held (ILUFSMeter.cpp:386-387): 16-byte stores over its aligned middle
    template <int TT>
    __device__ __forceinline__ void ilufs_fill(float *o, uint32_t n, float v)
    {
        const uint32_t tid = threadIdx.x;
        const uint32_t lead = uint32_t((4u - (uint32_t(reinterpret_cast<uintptr_t>(o) >> 2) & 3u)) & 3u);
        const uint32_t head = (lead < n) ? lead : n, quads = (n - head) >> 2;
        if (tid < head)
            o[tid] = v;
        float4 *o4 = reinterpret_cast<float4 *>(o + head);
        for (uint32_t i = tid; i < quads; i += TT)
            o4[i] = make_float4(v, v, v, v);
        for (uint32_t i = head + 4u * quads + tid; i < n; i += TT)
            o[i] = v;
    }

    // The pieces of one call for one meter, by a workgroup of TT threads (s_sum, s_cnt: four cells of LDS each)
    // LOCAL: `seg` is this workgroup's own LDS, [channels][4], left there by the rows' filters a barrier ago (nothing to take
    // from memory and nothing to clear behind)
    template <int TT, bool FRESH = false, bool LOCAL = false>
    __device__ void ilufs_call_body(uint32_t meter, float *block, float *seg, const ilufs_pieces &pieces,
                                    const chan_cfg *__restrict__ cfg, uint32_t channels, float *out, size_t out_stride,
                                    ilufs_state *st, float gain, float *hist, uint32_t size, uint32_t ms_int, float avg,
                                    float *s_sum, uint32_t *s_cnt, float &s_val, float *s_chan /* [2 TT] */,
                                    const ilufs_early<TT> &early, bool first_filled = false /* the first piece's output run is written */)
    {
        const uint32_t tid = threadIdx.x;
        // the value being held, the row's quarters and whether it counts came early (ilufs_ask); what this call has
        // produced -- the pieces' sums of squares of the thread's row -- is asked for now
        ilufs_state me = early.me;
        float held = me.loudness;
        const uint32_t myrow = meter * channels + tid;
        float4 myseg = make_float4(0.0f, 0.0f, 0.0f, 0.0f), myblk = early.blk;
        const bool counts = early.counts != 0;
        bool first_gate = true;
        if (tid < channels)
        {
            const float *ms = LOCAL ? seg + size_t(tid) * 4 : seg + size_t(myrow) * 4;
            myseg = LOCAL ? make_float4(ms[0], ms[1], ms[2], ms[3])
                  : FRESH ? make_float4(seg_load<true>(ms), seg_load<true>(ms + 1), seg_load<true>(ms + 2), seg_load<true>(ms + 3))
                          : *reinterpret_cast<const float4 *>(ms);
        }
        for (uint32_t k = 0; k < pieces.count; ++k)
        {
            const ilufs_piece pc = pieces.p[k];
            if (out != nullptr && pc.n > 0 && !(k == 0 && first_filled))
                ilufs_fill<TT>(out + size_t(meter) * out_stride + pc.offset, pc.n, held * gain);
            if (pc.n > 0)
            {
                if (tid < channels && counts)
                {
                    // (components picked by value: a picked POINTER into myblk parks it in scratch memory)
                    const float add = (k == 0) ? myseg.x : (k == 1) ? myseg.y : (k == 2) ? myseg.z : myseg.w;
                    const float now = ((pc.part == 0) ? myblk.x : (pc.part == 1) ? myblk.y : (pc.part == 2) ? myblk.z : myblk.w) + add;
                    myblk.x = (pc.part == 0) ? now : myblk.x;
                    myblk.y = (pc.part == 1) ? now : myblk.y;
                    myblk.z = (pc.part == 2) ? now : myblk.z;
                    myblk.w = (pc.part == 3) ? now : myblk.w;
                    block[size_t(myrow) * 4 + pc.part] = now;
                }
                for (uint32_t c = tid + TT; !LOCAL && c < channels; c += TT)       // meters of more channels than threads
                    if (cfg[c].enabled)
                    {
                        const uint32_t row = meter * channels + c;
                        block[row * 4 + pc.part] += seg_load<FRESH>(seg + row * 4 + k);
                    }
            }
            __syncthreads();
            if (pc.gate)
            {
                held = ilufs_gate<TT>(meter, st, me, hist, size, ms_int, block, myblk, early.weight, cfg, channels, avg, s_sum, s_cnt,
                                      s_val, s_chan, (first_gate && early.hist_ok) ? early.hv : nullptr);
                first_gate = false;
            }
            __syncthreads();
            if (pc.zero_part >= 0)
            {
                if (tid < channels)
                {
                    myblk.x = (pc.zero_part == 0) ? 0.0f : myblk.x;
                    myblk.y = (pc.zero_part == 1) ? 0.0f : myblk.y;
                    myblk.z = (pc.zero_part == 2) ? 0.0f : myblk.z;
                    myblk.w = (pc.zero_part == 3) ? 0.0f : myblk.w;
                }
                for (uint32_t c = tid; c < channels; c += TT)
                    block[(meter * channels + c) * 4 + uint32_t(pc.zero_part)] = 0.0f;
            }
            __syncthreads();
        }
        for (uint32_t i = tid; !LOCAL && i < channels * 4; i += TT)   // consumed: the next call's filter launch adds to zeros
            seg[size_t(meter) * channels * 4 + i] = 0.0f;
    }
#endif
} // namespace mi_meters
