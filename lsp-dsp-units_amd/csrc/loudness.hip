// Loudness meter bank: lsp::dspu::LoudnessMeter for many meters
// (reference: src/main/meters/LoudnessMeter.cpp:85-185 init, :328-379 update_settings, :381-407 refresh_rms,
//  :409-466 process_channels, :468-560 process).
//
// Per channel: weighting filter (the biquad cascade bank, designer on the host), squares into a power-of-two line,
// sliding sum over the last nPeriod squares.  The reference updates that sum sample by sample (ms += new - old) and
// re-sums the window exactly every max(4096, nPeriod/4) samples to stop the float32 drift; here a block is one
// prefix scan of (new - old) on top of the carried sum, and the exact re-summation runs on the same schedule.
// One workgroup per meter: it walks the meter's channels, mixes the mean squares with the designation weights,
// takes the square root and writes the mixed loudness and the (linked) per-channel values.
#include "mi_common.h"
#include "ilufs_device.h"
#include "host/filter_design.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace
{
    constexpr uint32_t BUFFER_SIZE = 0x400;                 // LoudnessMeter.cpp:32
    constexpr int      LT = 256;                            // threads per meter
    constexpr uint32_t MAX_BLOCK = 4096;                    // samples per launch: register-resident passes of the workgroup
    constexpr uint32_t SEG = 256;                           // cells per segment of a line whose sum is kept beside the line

    // bs::channel_weighting (src/main/misc/broadcast.cpp:32-55)
    float channel_weighting(int designation)
    {
        if (designation >= 6 && designation <= 11)
            return 1.41f;                                   // ~ +1.5 dB
        if (designation == MI_BS_CHANNEL_LFE1 || designation == MI_BS_CHANNEL_LFE2)
            return 0.0f;
        return 1.0f;
    }

    using mi_meters::chan_cfg;          // ilufs_device.h (shared with the integrated meter's fused launch)

    // inclusive sum scan over the 64 lanes with DPP: shifts inside the rows of 16 lanes, then the sums of rows 0 and 2 to
    // rows 1 and 3 (row_bcast:15) and of the lower half to the upper half (row_bcast:31)
    template <int CTRL, int ROW_MASK, bool ZERO_FILL>
    __device__ __forceinline__ float dpp_term(float v)
    {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, ZERO_FILL));
    }

    __device__ __forceinline__ float wave_scan(float v)
    {
        v += dpp_term<0x111, 0xf, true>(v);                 // row_shr:1
        v += dpp_term<0x112, 0xf, true>(v);                 // row_shr:2
        v += dpp_term<0x114, 0xf, true>(v);                 // row_shr:4
        v += dpp_term<0x118, 0xf, true>(v);                 // row_shr:8
        v += dpp_term<0x142, 0xa, false>(v);                // row_bcast:15 into rows 1 and 3
        v += dpp_term<0x143, 0xc, false>(v);                // row_bcast:31 into rows 2 and 3
        return v;
    }

    // sum of `cnt` cells of a ring line from cell `start` on, by the whole workgroup of T threads (every thread gets it)
    template <uint32_t T>
    __device__ float window_sum(const float *d, uint32_t size, uint32_t start, uint32_t cnt, float *part /* [T / 64] */)
    {
        const uint32_t tid = threadIdx.x, mask = size - 1;
        // 16-byte cells of the ring that cover [start, start + cnt); the cells at both ends are cut to the window
        const uint32_t lead = start & 3u, first = start - lead, cells = (cnt > 0) ? (lead + cnt + 3u) >> 2 : 0u;
        float s0 = 0.0f, s1 = 0.0f;
        for (uint32_t q = tid; q < cells; q += T)
        {
            const float4 v = *reinterpret_cast<const float4 *>(d + ((first + 4u * q) & mask));
            const uint32_t at = 4u * q;                     // position of the cell's first float, counted from `first`
            if (at >= lead && at + 4u <= lead + cnt)
            {
                s0 += v.x + v.z;
                s1 += v.y + v.w;
            }
            else
            {
                const float e[4] = { v.x, v.y, v.z, v.w };
                #pragma unroll
                for (uint32_t k = 0; k < 4; ++k)
                    if (at + k >= lead && at + k < lead + cnt)
                        s0 += e[k];
            }
        }
        float s = s0 + s1;
        #pragma unroll
        for (int w = 32; w > 0; w >>= 1)
            s += __shfl_xor(s, w);
        __syncthreads();                                    // part[] is free
        if ((tid & 63) == 0)
            part[tid >> 6] = s;
        __syncthreads();
        float r = 0.0f;
        #pragma unroll
        for (uint32_t w = 0; w < T / 64; ++w)
            r += part[w];
        return r;
    }

    // One block of `n` <= E * T samples of one meter: T threads, sample j = i * T + tid lives in register i of thread tid.
    // refresh_at < n: the reference re-sums the window exactly before that sample of the block (refresh_rms(),
    // LoudnessMeter.cpp:381-407 on its schedule :496-503); it is done here, inside the block, so that a block never has to
    // be cut at the refresh point.
    // The kernel is a chain of dependent phases per channel (one workgroup round: its duration is the chain's latency, not
    // the bytes moved -- with every global access of the block switched off it still takes 17 of its 20 us), so whatever can
    // be asked for early is: the channel settings travel by value with the launch (up to CFG_BY_VALUE channels), the
    // cells and segment sums of the exact re-summation are requested together with the block's samples.
    constexpr uint32_t CFG_BY_VALUE = 8;
    struct cfg_pack { chan_cfg c[CFG_BY_VALUE]; };

    template <uint32_t T, uint32_t E>
    __global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4)))     // <= 128 VGPRs: two workgroups of 512 on a CU
    void loudness_block_kernel(float *out, float *ch_out, size_t out_stride, const float *__restrict__ flt, size_t flt_stride,
                               float *data, uint32_t size, uint32_t head, uint32_t period, float avg, float *ms,
                               float *msbuf, size_t msbuf_stride, const chan_cfg *__restrict__ cfg_mem, const cfg_pack pack,
                               uint32_t channels, uint32_t n, float gain, float *loud, uint32_t refresh_at, float *segsum,
                               int use_seg)
    {
        constexpr uint32_t NWV = T / 64, MAXSEG = E * T / SEG + 1;
        auto cfg_of = [&](uint32_t c) -> chan_cfg { return (channels <= CFG_BY_VALUE) ? pack.c[c] : cfg_mem[c]; };
        __shared__ float segnew[MAXSEG];                    // sums of the line's segments this block writes to
        __shared__ float sq[E * T];                         // this channel's squares (new values)
        __shared__ __align__(16) float wtot[E][NWV];        // sums of the waves of every pass
        __shared__ float part[NWV];
        __shared__ float s_before;                          // prefix sum just before the refresh point
        const uint32_t meter = blockIdx.x, tid = threadIdx.x, mask = size - 1;
        const uint32_t lane = tid & 63, wave = tid >> 6;
        const uint32_t tail = (head + size - period) & mask;
        const bool refresh = refresh_at < n;
        float mix[E];                                       // weighted sum of the channels' mean squares
        #pragma unroll
        for (uint32_t i = 0; i < E; ++i)
            mix[i] = 0.0f;
        uint32_t mixed = 0;
        // enabled channels without an input: the line stands still, but refresh_rms() re-sums every enabled channel's window
        if (refresh)
            for (uint32_t c = 0; c < channels; ++c)
            {
                const chan_cfg cc = cfg_of(c);
                if (!cc.enabled || !cc.unbound)
                    continue;
                const uint32_t row = meter * channels + c;
                const float r = window_sum<T>(data + size_t(row) * size, size, (head + refresh_at + size - period) & mask, period, part);
                if (tid == 0)
                    ms[row] = r;
            }
        // the channels with an input, one after the other
        for (uint32_t c = 0; c < channels; ++c)
        {
            const chan_cfg cc = cfg_of(c);
            if (!cc.enabled || cc.unbound)
                continue;
            const uint32_t row = meter * channels + c;
            float *line = data + size_t(row) * size;
            const float *x = flt + size_t(row) * flt_stride;
            float d[E], old[E];
            #pragma unroll
            for (uint32_t i = 0; i < E; ++i)                // every load of the block is issued before anything waits
            {
                const uint32_t j = i * T + tid;
                d[i] = (j < n) ? x[j] : 0.0f;
                old[i] = (j < n && j < period) ? line[(tail + j) & mask] : 0.0f;
            }
            const float start = ms[row];
            // what the exact re-summation needs from memory, asked for now: whole segments from their sums, the cut
            // segments at both ends of the window cell by cell (positions are counted with one lap added so that the
            // window's start stays positive; the block's own squares and segments come from LDS further down)
            const uint32_t segs = size / SEG, q0 = head / SEG, nseg = (head + n - 1) / SEG - q0 + 1;
            float *gseg = segsum + size_t(row) * segs;
            const uint32_t H = head + size, we = H + refresh_at, ws = we - period;
            const uint32_t bl = ((ws + SEG - 1) / SEG) * SEG, br = (we / SEG) * SEG;
            float pre = 0.0f;
            if (refresh && use_seg)
            {
                if (bl > br)
                    for (uint32_t u = ws + tid; u < we && u < H; u += T)
                        pre += line[u & mask];
                else
                {
                    for (uint32_t u = ws + tid; u < bl && u < H; u += T)
                        pre += line[u & mask];
                    for (uint32_t u = br + tid; u < we && u < H; u += T)
                        pre += line[u & mask];
                    for (uint32_t q = bl / SEG + tid; q < br / SEG && q < H / SEG; q += T)
                        pre += gseg[q & (segs - 1)];
                }
            }
            const float gfirst = (head % SEG != 0) ? gseg[q0 & (segs - 1)] : 0.0f;     // the segment the block starts inside
            __syncthreads();                                // the previous channel is through with sq[] and wtot[]
            #pragma unroll
            for (uint32_t i = 0; i < E; ++i)                // dsp::sqr2 into the line (LoudnessMeter.cpp:428-436)
            {
                const uint32_t j = i * T + tid;
                d[i] *= d[i];
                if (j < n)
                {
                    sq[j] = d[i];
                    line[(head + j) & mask] = d[i];
                }
            }
            __syncthreads();
            // Sums of the segments of SEG cells the block writes to, kept beside the line: a segment is always filled from
            // its first cell on, block after block, so the block that holds the first cell restarts the sum and the
            // others add to it.  (The sums stand for the cells themselves only where the line was written without a
            // break; the host says when that holds -- use_seg.)
            for (uint32_t sgi = wave; sgi < nseg; sgi += NWV)
            {
                const uint32_t lo = ((q0 + sgi) * SEG > head) ? (q0 + sgi) * SEG - head : 0u;
                const uint32_t hi = ((q0 + sgi + 1) * SEG - head < n) ? (q0 + sgi + 1) * SEG - head : n;
                float v = 0.0f;
                for (uint32_t j = lo + lane; j < hi; j += 64)
                    v += sq[j];
                #pragma unroll
                for (int w = 32; w > 0; w >>= 1)
                    v += __shfl_xor(v, w);
                if (lane == 0)
                {
                    const uint32_t phys = (q0 + sgi) & (segs - 1);
                    const float total = ((q0 + sgi) * SEG >= head) ? v : gfirst + v;
                    segnew[sgi] = total;
                    gseg[phys] = total;
                }
            }
            // the exact window sum at the refresh point
            float exact = 0.0f;
            if (refresh && use_seg)
            {
                // the part of the window that this block wrote: its squares and its segments' sums, from LDS
                __syncthreads();                            // segnew[]
                float s = pre;
                if (bl > br)
                    for (uint32_t u = ((ws > H) ? ws : H) + tid; u < we; u += T)
                        s += sq[u - H];
                else
                {
                    for (uint32_t u = ((ws > H) ? ws : H) + tid; u < bl; u += T)
                        s += sq[u - H];
                    for (uint32_t u = ((br > H) ? br : H) + tid; u < we; u += T)
                        s += sq[u - H];
                    for (uint32_t q = ((bl / SEG > H / SEG) ? bl / SEG : H / SEG) + tid; q < br / SEG; q += T)
                        s += segnew[q - H / SEG];
                }
                #pragma unroll
                for (int w = 32; w > 0; w >>= 1)
                    s += __shfl_xor(s, w);
                __syncthreads();
                if (lane == 0)
                    part[wave] = s;
                __syncthreads();
                #pragma unroll
                for (uint32_t w = 0; w < NWV; ++w)
                    exact += part[w];
            }
            else if (refresh)
            {
                // cell by cell: the cells behind the block from the line, the block's own squares before the point from LDS
                const uint32_t from_line = (refresh_at < period) ? period - refresh_at : 0u;
                exact = window_sum<T>(line, size, (head + refresh_at + size - period) & mask, from_line, part);
                float s = 0.0f;
                for (uint32_t j = refresh_at - (period - from_line) + tid; j < refresh_at; j += T)
                    s += sq[j];
                #pragma unroll
                for (int w = 32; w > 0; w >>= 1)
                    s += __shfl_xor(s, w);
                __syncthreads();
                if (lane == 0)
                    part[wave] = s;
                __syncthreads();
                #pragma unroll
                for (uint32_t w = 0; w < NWV; ++w)
                    exact += part[w];
            }
            // ms_j = ms_(j-1) + (new_j - old_j): an inclusive scan of every pass inside the wave, the sums of the waves
            // through LDS, then the running sum carried from pass to pass
            #pragma unroll
            for (uint32_t i = 0; i < E; ++i)
            {
                const uint32_t j = i * T + tid;
                if (j < n)
                    d[i] -= (j >= period) ? sq[j - period] : old[i];
                d[i] = wave_scan(d[i]);
                if (lane == 63)
                    wtot[i][wave] = d[i];
            }
            __syncthreads();
            float carry = 0.0f;                             // prefix sums counted from the start of the block
            #pragma unroll
            for (uint32_t i = 0; i < E; ++i)
            {
                float before = carry, total = 0.0f;
                #pragma unroll
                for (uint32_t w = 0; w < NWV; w += 4)
                {
                    const float4 t = *reinterpret_cast<const float4 *>(&wtot[i][w]);
                    before += ((w + 0 < wave) ? t.x : 0.0f) + ((w + 1 < wave) ? t.y : 0.0f) +
                              ((w + 2 < wave) ? t.z : 0.0f) + ((w + 3 < wave) ? t.w : 0.0f);
                    total += (t.x + t.y) + (t.z + t.w);
                }
                d[i] += before;                             // P_j
                carry += total;
                if (refresh && refresh_at > 0 && i * T + tid + 1 == refresh_at)
                    s_before = d[i];
            }
            if (refresh)
                __syncthreads();
            const float rebase = refresh ? exact - ((refresh_at > 0) ? s_before : 0.0f) : 0.0f;
            float *mb = msbuf + size_t(row) * msbuf_stride;
            #pragma unroll
            for (uint32_t i = 0; i < E; ++i)
            {
                const uint32_t j = i * T + tid;
                const float base = (refresh && j >= refresh_at) ? rebase : start;
                const float m = avg * (base + d[i]);        // vMS[j] = fAvgCoeff * ms
                if (j < n && ch_out != nullptr)
                    mb[j] = m;
                mix[i] = (mixed > 0) ? fmaf(m, cc.weight, mix[i]) : m * cc.weight;     // fmadd_k3 / mul_k3
            }
            if (tid == 0)
                ms[row] = (refresh ? rebase : start) + carry;    // the running sum goes on with the next block
            ++mixed;
        }
        // ssqrt1: sqrt of the non-negative part; then the outputs
        #pragma unroll
        for (uint32_t i = 0; i < E; ++i)
        {
            const uint32_t j = i * T + tid;
            mix[i] = (mix[i] > 0.0f) ? sqrtf(mix[i]) : 0.0f;
            if (out != nullptr && j < n)
                out[size_t(meter) * out_stride + j] = mix[i] * gain;
            if (loud != nullptr && j + 1 == n)
                loud[meter] = mix[i];
        }
        if (ch_out == nullptr)
            return;
        for (uint32_t c = 0; c < channels; ++c)
        {
            const chan_cfg cc = cfg_of(c);
            if (!cc.enabled || cc.unbound)                  // (the reference hands an unbound channel's stale buffer on: nothing here)
                continue;
            const uint32_t row = meter * channels + c;
            const float *mb = msbuf + size_t(row) * msbuf_stride;
            float *o = ch_out + size_t(row) * out_stride;
            #pragma unroll
            for (uint32_t i = 0; i < E; ++i)
            {
                const uint32_t j = i * T + tid;
                if (j >= n)
                    continue;
                const float r = (mb[j] > 0.0f) ? sqrtf(mb[j]) : 0.0f;   // written by this same thread above
                float v;
                if (cc.link <= 0.0f)       v = r * gain;
                else if (cc.link >= 1.0f)  v = mix[i] * gain;
                else                       v = mix[i] * (cc.link * gain) + r * ((1.0f - cc.link) * gain);     // mix_copy2
                o[j] = v;
            }
        }
    }
    // The same block with FOUR consecutive samples per lane (16-byte accesses): sample j = 4 (i T + tid) + k.  The scalar
    // kernel above spends its time issuing instructions, not moving bytes (SQ counters: 1900 VALU instructions per wave,
    // most of them address arithmetic and per-pass scan bookkeeping; with every global access switched off it still takes
    // 17 of its 20 us) -- a quarter of the passes means a quarter of that.  Needs n, head and period to be multiples of 4
    // (then no 16-byte cell straddles the window's tail, the end of the line or the end of the block); anything else
    // takes the scalar kernel.
    template <uint32_t T, uint32_t E>
    __global__ __launch_bounds__(T)
    void loudness_block4_kernel(float *out, float *ch_out, size_t out_stride, const float *__restrict__ flt, size_t flt_stride,
                                float *data, uint32_t size, uint32_t head, uint32_t period, float avg, float *ms,
                                float *msbuf, size_t msbuf_stride, const chan_cfg *__restrict__ cfg_mem, const cfg_pack pack,
                                uint32_t channels, uint32_t n, float gain, float *loud, uint32_t refresh_at, float *segsum,
                                int use_seg)
    {
        constexpr uint32_t NWV = T / 64, MAXSEG = E * T * 4 / SEG + 1;
        auto cfg_of = [&](uint32_t c) -> chan_cfg { return (channels <= CFG_BY_VALUE) ? pack.c[c] : cfg_mem[c]; };
        __shared__ float segnew[MAXSEG];
        __shared__ __align__(16) float sq[E * T * 4];
        __shared__ __align__(16) float wtot[E][NWV];
        __shared__ float part[NWV];
        __shared__ float s_before;
        const uint32_t meter = blockIdx.x, tid = threadIdx.x, mask = size - 1;
        const uint32_t lane = tid & 63, wave = tid >> 6;
        const uint32_t tail = (head + size - period) & mask;
        const bool refresh = refresh_at < n;
        float4 mix[E];
        #pragma unroll
        for (uint32_t i = 0; i < E; ++i)
            mix[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        uint32_t mixed = 0;
        if (refresh)
            for (uint32_t c = 0; c < channels; ++c)
            {
                const chan_cfg cc = cfg_of(c);
                if (!cc.enabled || !cc.unbound)
                    continue;
                const uint32_t row = meter * channels + c;
                const float r = window_sum<T>(data + size_t(row) * size, size, (head + refresh_at + size - period) & mask, period, part);
                if (tid == 0)
                    ms[row] = r;
            }
        for (uint32_t c = 0; c < channels; ++c)
        {
            const chan_cfg cc = cfg_of(c);
            if (!cc.enabled || cc.unbound)
                continue;
            const uint32_t row = meter * channels + c;
            float *line = data + size_t(row) * size;
            const float *x = flt + size_t(row) * flt_stride;
            float4 q[E], old[E];
            #pragma unroll
            for (uint32_t i = 0; i < E; ++i)
            {
                const uint32_t j = (i * T + tid) * 4;
                q[i] = (j < n) ? *reinterpret_cast<const float4 *>(x + j) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                old[i] = (j < n && j < period) ? *reinterpret_cast<const float4 *>(line + ((tail + j) & mask))
                                               : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            const float start = ms[row];
            const uint32_t segs = size / SEG, q0 = head / SEG, nseg = (head + n - 1) / SEG - q0 + 1;
            float *gseg = segsum + size_t(row) * segs;
            const uint32_t H = head + size, we = H + refresh_at, ws = we - period;
            const uint32_t bl = ((ws + SEG - 1) / SEG) * SEG, br = (we / SEG) * SEG;
            float pre = 0.0f;
            if (refresh && use_seg)
            {
                if (bl > br)
                    for (uint32_t u = ws + tid; u < we && u < H; u += T)
                        pre += line[u & mask];
                else
                {
                    for (uint32_t u = ws + tid; u < bl && u < H; u += T)
                        pre += line[u & mask];
                    for (uint32_t u = br + tid; u < we && u < H; u += T)
                        pre += line[u & mask];
                    for (uint32_t s2 = bl / SEG + tid; s2 < br / SEG && s2 < H / SEG; s2 += T)
                        pre += gseg[s2 & (segs - 1)];
                }
            }
            const float gfirst = (head % SEG != 0) ? gseg[q0 & (segs - 1)] : 0.0f;
            __syncthreads();                                // the previous channel is through with sq[] and wtot[]
            #pragma unroll
            for (uint32_t i = 0; i < E; ++i)                // dsp::sqr2 into the line (LoudnessMeter.cpp:428-436)
            {
                const uint32_t j = (i * T + tid) * 4;
                q[i] = make_float4(q[i].x * q[i].x, q[i].y * q[i].y, q[i].z * q[i].z, q[i].w * q[i].w);
                if (j < n)
                {
                    *reinterpret_cast<float4 *>(sq + j) = q[i];
                    *reinterpret_cast<float4 *>(line + ((head + j) & mask)) = q[i];
                }
            }
            __syncthreads();
            // sums of the segments the block writes to (see the scalar kernel)
            static_assert(SEG == 256, "one 16-byte cell per lane covers a segment");
            for (uint32_t sgi = wave; sgi < nseg; sgi += NWV)
            {
                const uint32_t lo = ((q0 + sgi) * SEG > head) ? (q0 + sgi) * SEG - head : 0u;
                const uint32_t hi = ((q0 + sgi + 1) * SEG - head < n) ? (q0 + sgi + 1) * SEG - head : n;
                float v = 0.0f;
                const uint32_t j = lo + 4 * lane;           // SEG = 4 x 64: one 16-byte cell per lane
                if (j < hi)
                {
                    const float4 t = *reinterpret_cast<const float4 *>(sq + j);
                    v = (t.x + t.y) + (t.z + t.w);
                }
                #pragma unroll
                for (int w = 32; w > 0; w >>= 1)
                    v += __shfl_xor(v, w);
                if (lane == 0)
                {
                    const uint32_t phys = (q0 + sgi) & (segs - 1);
                    const float total = ((q0 + sgi) * SEG >= head) ? v : gfirst + v;
                    segnew[sgi] = total;
                    gseg[phys] = total;
                }
            }
            float exact = 0.0f;
            if (refresh && use_seg)
            {
                __syncthreads();                            // segnew[]
                float s = pre;
                if (bl > br)
                    for (uint32_t u = ((ws > H) ? ws : H) + tid; u < we; u += T)
                        s += sq[u - H];
                else
                {
                    for (uint32_t u = ((ws > H) ? ws : H) + tid; u < bl; u += T)
                        s += sq[u - H];
                    for (uint32_t u = ((br > H) ? br : H) + tid; u < we; u += T)
                        s += sq[u - H];
                    for (uint32_t s2 = ((bl / SEG > H / SEG) ? bl / SEG : H / SEG) + tid; s2 < br / SEG; s2 += T)
                        s += segnew[s2 - H / SEG];
                }
                #pragma unroll
                for (int w = 32; w > 0; w >>= 1)
                    s += __shfl_xor(s, w);
                __syncthreads();
                if (lane == 0)
                    part[wave] = s;
                __syncthreads();
                #pragma unroll
                for (uint32_t w = 0; w < NWV; ++w)
                    exact += part[w];
            }
            else if (refresh)
            {
                const uint32_t from_line = (refresh_at < period) ? period - refresh_at : 0u;
                exact = window_sum<T>(line, size, (head + refresh_at + size - period) & mask, from_line, part);
                float s = 0.0f;
                for (uint32_t j = refresh_at - (period - from_line) + tid; j < refresh_at; j += T)
                    s += sq[j];
                #pragma unroll
                for (int w = 32; w > 0; w >>= 1)
                    s += __shfl_xor(s, w);
                __syncthreads();
                if (lane == 0)
                    part[wave] = s;
                __syncthreads();
                #pragma unroll
                for (uint32_t w = 0; w < NWV; ++w)
                    exact += part[w];
            }
            // ms_j = ms_(j-1) + (new_j - old_j): the four samples of a lane in sequence, an inclusive scan of the lanes'
            // sums inside the wave, the sums of the waves through LDS, then the running sum carried from pass to pass
            #pragma unroll
            for (uint32_t i = 0; i < E; ++i)
            {
                const uint32_t j = (i * T + tid) * 4;
                float4 d = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (j < n)
                {
                    const float4 o = (j >= period) ? *reinterpret_cast<const float4 *>(sq + (j - period)) : old[i];
                    d = make_float4(q[i].x - o.x, q[i].y - o.y, q[i].z - o.z, q[i].w - o.w);
                }
                d.y += d.x; d.z += d.y; d.w += d.z;
                const float incl = wave_scan(d.w);
                const float excl = incl - d.w;
                q[i] = make_float4(d.x + excl, d.y + excl, d.z + excl, d.w + excl);
                if (lane == 63)
                    wtot[i][wave] = incl;
            }
            __syncthreads();
            float carry = 0.0f;
            #pragma unroll
            for (uint32_t i = 0; i < E; ++i)
            {
                const uint32_t j = (i * T + tid) * 4;
                float before = carry, total = 0.0f;
                #pragma unroll
                for (uint32_t w = 0; w < NWV; w += 4)
                {
                    const float4 t = *reinterpret_cast<const float4 *>(&wtot[i][w]);
                    before += ((w + 0 < wave) ? t.x : 0.0f) + ((w + 1 < wave) ? t.y : 0.0f) +
                              ((w + 2 < wave) ? t.z : 0.0f) + ((w + 3 < wave) ? t.w : 0.0f);
                    total += (t.x + t.y) + (t.z + t.w);
                }
                q[i] = make_float4(q[i].x + before, q[i].y + before, q[i].z + before, q[i].w + before);     // P_j
                carry += total;
                if (refresh && refresh_at > 0 && refresh_at - 1 >= j && refresh_at - 1 < j + 4)
                {
                    const uint32_t k = refresh_at - 1 - j;
                    s_before = (k == 0) ? q[i].x : (k == 1) ? q[i].y : (k == 2) ? q[i].z : q[i].w;
                }
            }
            if (refresh)
                __syncthreads();
            const float rebase = refresh ? exact - ((refresh_at > 0) ? s_before : 0.0f) : 0.0f;
            float *mb = msbuf + size_t(row) * msbuf_stride;
            #pragma unroll
            for (uint32_t i = 0; i < E; ++i)
            {
                const uint32_t j = (i * T + tid) * 4;
                auto at = [&](uint32_t k, float p) -> float { return avg * (((refresh && j + k >= refresh_at) ? rebase : start) + p); };
                const float4 m = make_float4(at(0, q[i].x), at(1, q[i].y), at(2, q[i].z), at(3, q[i].w));   // vMS[j] = fAvgCoeff * ms
                if (j < n && ch_out != nullptr)
                    *reinterpret_cast<float4 *>(mb + j) = m;
                if (mixed > 0)
                    mix[i] = make_float4(fmaf(m.x, cc.weight, mix[i].x), fmaf(m.y, cc.weight, mix[i].y),
                                         fmaf(m.z, cc.weight, mix[i].z), fmaf(m.w, cc.weight, mix[i].w));  // fmadd_k3
                else
                    mix[i] = make_float4(m.x * cc.weight, m.y * cc.weight, m.z * cc.weight, m.w * cc.weight);   // mul_k3
            }
            if (tid == 0)
                ms[row] = (refresh ? rebase : start) + carry;
            ++mixed;
        }
        // ssqrt1: sqrt of the non-negative part; then the outputs
        auto root = [](float v) -> float { return (v > 0.0f) ? sqrtf(v) : 0.0f; };
        #pragma unroll
        for (uint32_t i = 0; i < E; ++i)
        {
            const uint32_t j = (i * T + tid) * 4;
            mix[i] = make_float4(root(mix[i].x), root(mix[i].y), root(mix[i].z), root(mix[i].w));
            if (out != nullptr && j < n)
                *reinterpret_cast<float4 *>(out + size_t(meter) * out_stride + j) =
                    make_float4(mix[i].x * gain, mix[i].y * gain, mix[i].z * gain, mix[i].w * gain);
            if (loud != nullptr && j + 4 == n)
                loud[meter] = mix[i].w;
        }
        if (ch_out == nullptr)
            return;
        for (uint32_t c = 0; c < channels; ++c)
        {
            const chan_cfg cc = cfg_of(c);
            if (!cc.enabled || cc.unbound)
                continue;
            const uint32_t row = meter * channels + c;
            const float *mb = msbuf + size_t(row) * msbuf_stride;
            float *o = ch_out + size_t(row) * out_stride;
            #pragma unroll
            for (uint32_t i = 0; i < E; ++i)
            {
                const uint32_t j = (i * T + tid) * 4;
                if (j >= n)
                    continue;
                const float4 m = *reinterpret_cast<const float4 *>(mb + j);   // written by this same thread above
                auto one = [&](float mv, float mx) -> float
                {
                    const float r = root(mv);
                    if (cc.link <= 0.0f)       return r * gain;
                    else if (cc.link >= 1.0f)  return mx * gain;
                    return mx * (cc.link * gain) + r * ((1.0f - cc.link) * gain);     // mix_copy2
                };
                *reinterpret_cast<float4 *>(o + j) = make_float4(one(m.x, mix[i].x), one(m.y, mix[i].y), one(m.z, mix[i].z), one(m.w, mix[i].w));
            }
        }
    }
} // namespace

struct mi_loudness_bank
{
    uint32_t    meters = 0, channels = 0, rows = 0;
    uint32_t    sample_rate = 0, period = 0, ms_refresh = 0, data_size = 0, head = 0;
    float       period_ms = 400.0f, max_period_ms = 400.0f, avg = 1.0f;
    int         weighting = MI_BS_WEIGHT_K;
    bool        upd_filters = true, upd_time = true, cfg_dirty = true;
    std::vector<chan_cfg> cfg;
    std::vector<int>      designation;
    mi_biquad_bank_t *filters = nullptr;
    float      *d_data = nullptr, *d_ms = nullptr, *d_flt = nullptr, *d_msbuf = nullptr, *d_loud = nullptr;
    float      *d_segsum = nullptr;         // [rows][data_size / SEG] sums of the lines' segments
    uint64_t    raw_left = 0;               // samples for which the window is still re-summed cell by cell
    chan_cfg   *d_cfg = nullptr;
    size_t      cap = 0;
};

namespace
{
    uint32_t round_pow2(uint32_t v)
    {
        uint32_t p = 1;
        while (p < v)
            p <<= 1;
        return p;
    }

    int update_settings(mi_loudness_bank *b, hipStream_t st)           // LoudnessMeter.cpp:328-379
    {
        if (b->upd_time)
        {
            const uint32_t p = uint32_t((b->period_ms * 0.001f) * float(b->sample_rate));    // millis_to_samples, truncated
            b->period = (p > 1u) ? p : 1u;
            b->avg = 1.0f / float(b->period);
            b->ms_refresh = 0;
            b->upd_time = false;
        }
        if (b->upd_filters)
        {
            static const uint32_t types[6] = { MI_FLT_NONE, MI_FLT_A_WEIGHTED, MI_FLT_B_WEIGHTED, MI_FLT_C_WEIGHTED,
                                               MI_FLT_D_WEIGHTED, MI_FLT_K_WEIGHTED };
            mi_filter_params_t fp;
            fp.nType = types[b->weighting]; fp.nSlope = 0; fp.fFreq = 0.0f; fp.fFreq2 = 0.0f; fp.fGain = 1.0f; fp.fQuality = 0.0f;
            mi::design d;
            d.cascades.reserve(mi::CHAINS_MAX + 1);
            mi::design_filter(&d, &fp, b->sample_rate);
            for (uint32_t r = 0; r < b->rows; ++r)                      // sBank.end(true): state cleared
            {
                const int e = mi_biquad_bank_set_chains(b->filters, r, d.sections.data(), uint32_t(d.sections.size()), 1);
                if (e != MI_OK)
                    return e;
            }
            const int e = mi_biquad_bank_commit(b->filters, st);
            if (e != MI_OK)
                return e;
            b->upd_filters = false;
        }
        if (b->cfg_dirty)
        {
            MI_HIP_CHECK(hipMemcpyAsync(b->d_cfg, b->cfg.data(), b->channels * sizeof(chan_cfg), hipMemcpyHostToDevice, st));
            MI_HIP_CHECK(hipStreamSynchronize(st));
            b->cfg_dirty = false;
        }
        return MI_OK;
    }
} // namespace

extern "C" {

int mi_loudness_bank_create(mi_loudness_bank_t **bank, uint32_t meters, uint32_t channels, float max_period_ms)
{
    MI_REQUIRE(bank != nullptr, MI_EINVAL, "mi_loudness_bank_create: NULL result pointer");
    *bank = nullptr;
    MI_REQUIRE(meters > 0 && channels > 0, MI_EINVAL, "mi_loudness_bank_create: meters and channels must be > 0");
    MI_REQUIRE(mi_dspu_device_count() > 0, MI_ENODEV, "no HIP device available (there is no CPU fallback)");
    mi_loudness_bank *b = new (std::nothrow) mi_loudness_bank();
    MI_REQUIRE(b != nullptr, MI_ENOMEM, "mi_loudness_bank_create: out of host memory");
    b->meters = meters;
    b->channels = channels;
    b->rows = meters * channels;
    b->max_period_ms = max_period_ms;
    b->period_ms = (max_period_ms < 400.0f) ? max_period_ms : 400.0f;          // LoudnessMeter.cpp:166
    b->cfg.assign(channels, chan_cfg{ 0.0f, 1.0f, 1, 0 });
    b->designation.assign(channels, MI_BS_CHANNEL_NONE);
    if (channels == 1)
        b->designation[0] = MI_BS_CHANNEL_CENTER;
    else if (channels == 2)
    {
        b->designation[0] = MI_BS_CHANNEL_LEFT;
        b->designation[1] = MI_BS_CHANNEL_RIGHT;
    }
    for (uint32_t c = 0; c < channels; ++c)                                     // others keep fWeight = 0 (:131)
        if (b->designation[c] != MI_BS_CHANNEL_NONE)
            b->cfg[c].weight = channel_weighting(b->designation[c]);
    int r = mi_biquad_bank_create(&b->filters, b->rows, 4);                     // sBank.init(4)
    if (r == MI_OK)
        mi::biquad_bank_output_reread(b->filters, true);                        // d_flt is the meter kernel's input
    hipError_t e = hipSuccess;
    if (r == MI_OK)
    {
        e = hipMalloc(reinterpret_cast<void **>(&b->d_ms), b->rows * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_loud), meters * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_cfg), channels * sizeof(chan_cfg));
        if (e == hipSuccess) e = hipMemset(b->d_ms, 0, b->rows * sizeof(float));
        if (e == hipSuccess) e = hipMemset(b->d_loud, 0, meters * sizeof(float));
    }
    if (r != MI_OK || e != hipSuccess)
    {
        mi_loudness_bank_destroy(b);
        return (r != MI_OK) ? r : mi::fail(MI_EHIP, "mi_loudness_bank_create: %s", hipGetErrorString(e));
    }
    *bank = b;
    return MI_OK;
}

int mi_loudness_bank_destroy(mi_loudness_bank_t *b)
{
    if (b == nullptr)
        return MI_OK;
    mi_biquad_bank_destroy(b->filters);
    (void)hipFree(b->d_data); (void)hipFree(b->d_ms); (void)hipFree(b->d_flt); (void)hipFree(b->d_msbuf);
    (void)hipFree(b->d_loud); (void)hipFree(b->d_cfg); (void)hipFree(b->d_segsum);
    delete b;
    return MI_OK;
}

int mi_loudness_bank_clear(mi_loudness_bank_t *b, void *stream)                // LoudnessMeter.cpp:280-295
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_loudness_bank_clear: NULL bank");
    hipStream_t st = mi::as_stream(stream);
    MI_HIP_CHECK(hipMemsetAsync(b->d_loud, 0, b->meters * sizeof(float), st));
    const int r = mi_biquad_bank_reset(b->filters, UINT32_MAX, stream);
    if (r != MI_OK)
        return r;
    // enabled channels only: rows of channel c are strided, clear them one channel at a time
    for (uint32_t c = 0; c < b->channels && b->d_data != nullptr; ++c)
    {
        if (!b->cfg[c].enabled)
            continue;
        MI_HIP_CHECK(hipMemset2DAsync(b->d_data + size_t(c) * b->data_size, size_t(b->channels) * b->data_size * sizeof(float), 0,
                                      size_t(b->data_size) * sizeof(float), b->meters, st));
        MI_HIP_CHECK(hipMemset2DAsync(b->d_ms + c, b->channels * sizeof(float), 0, sizeof(float), b->meters, st));
        const size_t segs = b->data_size / SEG;
        MI_HIP_CHECK(hipMemset2DAsync(b->d_segsum + size_t(c) * segs, size_t(b->channels) * segs * sizeof(float), 0,
                                      segs * sizeof(float), b->meters, st));
    }
    return MI_OK;
}

int mi_loudness_bank_set_sample_rate(mi_loudness_bank_t *b, uint32_t sample_rate, void *stream)    // LoudnessMeter.cpp:297-321
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_loudness_bank_set_sample_rate: NULL bank");
    if (b->sample_rate == sample_rate)
        return MI_OK;
    const uint32_t len = round_pow2(uint32_t((b->max_period_ms * 0.001f) * float(sample_rate)) + BUFFER_SIZE);
    (void)hipFree(b->d_data);
    b->d_data = nullptr;
    MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_data), size_t(b->rows) * len * sizeof(float)));
    MI_HIP_CHECK(hipMemsetAsync(b->d_data, 0, size_t(b->rows) * len * sizeof(float), mi::as_stream(stream)));
    MI_HIP_CHECK(hipMemsetAsync(b->d_ms, 0, b->rows * sizeof(float), mi::as_stream(stream)));
    (void)hipFree(b->d_segsum);
    b->d_segsum = nullptr;
    MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_segsum), size_t(b->rows) * (len / SEG) * sizeof(float)));
    MI_HIP_CHECK(hipMemsetAsync(b->d_segsum, 0, size_t(b->rows) * (len / SEG) * sizeof(float), mi::as_stream(stream)));
    b->sample_rate = sample_rate;
    b->data_size = len;
    b->head = 0;
    b->raw_left = 0;
    b->upd_filters = b->upd_time = true;
    return mi_loudness_bank_clear(b, stream);
}

int mi_loudness_bank_set_period(mi_loudness_bank_t *b, float period_ms)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_loudness_bank_set_period: NULL bank");
    period_ms = (period_ms < 0.0f) ? 0.0f : (period_ms > b->max_period_ms) ? b->max_period_ms : period_ms;
    if (b->period_ms == period_ms)
        return MI_OK;
    b->period_ms = period_ms;
    b->upd_time = true;
    return MI_OK;
}

int mi_loudness_bank_set_weighting(mi_loudness_bank_t *b, int weighting)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_loudness_bank_set_weighting: NULL bank");
    MI_REQUIRE(weighting >= MI_BS_WEIGHT_NONE && weighting <= MI_BS_WEIGHT_K, MI_EINVAL, "mi_loudness_bank_set_weighting: bad weighting %d", weighting);
    if (weighting == b->weighting)
        return MI_OK;
    b->weighting = weighting;
    b->upd_filters = true;
    return MI_OK;
}

int mi_loudness_bank_set_designation(mi_loudness_bank_t *b, uint32_t channel, int designation)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_loudness_bank_set_designation: NULL bank");
    MI_REQUIRE(channel < b->channels, MI_EINVAL, "mi_loudness_bank_set_designation: channel %u out of range", channel);   // STATUS_OVERFLOW
    b->designation[channel] = designation;
    b->cfg[channel].weight = channel_weighting(designation);
    b->cfg_dirty = true;
    return MI_OK;
}

int mi_loudness_bank_set_link(mi_loudness_bank_t *b, uint32_t channel, float link)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_loudness_bank_set_link: NULL bank");
    MI_REQUIRE(channel < b->channels, MI_EINVAL, "mi_loudness_bank_set_link: channel %u out of range", channel);
    b->cfg[channel].link = (link < 0.0f) ? 0.0f : (link > 1.0f) ? 1.0f : link;
    b->cfg_dirty = true;
    return MI_OK;
}

int mi_loudness_bank_set_active(mi_loudness_bank_t *b, uint32_t channel, int active, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_loudness_bank_set_active: NULL bank");
    MI_REQUIRE(channel < b->channels, MI_EINVAL, "mi_loudness_bank_set_active: channel %u out of range", channel);
    if ((b->cfg[channel].enabled != 0) == (active != 0))
        return MI_OK;
    b->cfg[channel].enabled = active ? 1 : 0;
    b->cfg_dirty = true;
    for (uint32_t m = 0; m < b->meters; ++m)                // a disabled channel's filter is not run: its memory freezes (:420-422)
    {
        const int r = mi_biquad_bank_set_row_enabled(b->filters, m * b->channels + channel, active && !b->cfg[channel].unbound);
        if (r != MI_OK)
            return r;
    }
    if (active && b->d_data != nullptr)                     // re-enabled: the channel starts from silence (:249-253)
    {
        hipStream_t st = mi::as_stream(stream);
        MI_HIP_CHECK(hipMemset2DAsync(b->d_data + size_t(channel) * b->data_size, size_t(b->channels) * b->data_size * sizeof(float), 0,
                                      size_t(b->data_size) * sizeof(float), b->meters, st));
        MI_HIP_CHECK(hipMemset2DAsync(b->d_ms + channel, b->channels * sizeof(float), 0, sizeof(float), b->meters, st));
        const size_t segs = b->data_size / SEG;
        MI_HIP_CHECK(hipMemset2DAsync(b->d_segsum + size_t(channel) * segs, size_t(b->channels) * segs * sizeof(float), 0,
                                      segs * sizeof(float), b->meters, st));
    }
    return MI_OK;
}

int mi_loudness_bank_set_bound(mi_loudness_bank_t *b, uint32_t channel, int bound)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_loudness_bank_set_bound: NULL bank");
    MI_REQUIRE(channel < b->channels, MI_EINVAL, "mi_loudness_bank_set_bound: channel %u out of range", channel);
    const int unbound = bound ? 0 : 1;
    if (b->cfg[channel].unbound == unbound)
        return MI_OK;
    b->cfg[channel].unbound = unbound;
    b->cfg_dirty = true;
    if (!unbound)                                           // its line was not written for a while: the segment sums of the
        b->raw_left = uint64_t(b->data_size) + SEG;         // bank are trusted again once the line has been written all round
    const int run = (b->cfg[channel].enabled && !unbound) ? 1 : 0;
    for (uint32_t m = 0; m < b->meters; ++m)
    {
        const int r = mi_biquad_bank_set_row_enabled(b->filters, m * b->channels + channel, run);
        if (r != MI_OK)
            return r;
    }
    return MI_OK;
}

int mi_loudness_bank_needs_update(const mi_loudness_bank_t *b, int *pending)                  // LoudnessMeter.h:264
{
    MI_REQUIRE(b != nullptr && pending != nullptr, MI_EINVAL, "mi_loudness_bank_needs_update: bad argument");
    *pending = (b->upd_filters || b->upd_time) ? 1 : 0;
    return MI_OK;
}

int mi_loudness_bank_update_settings(mi_loudness_bank_t *b, void *stream)                     // LoudnessMeter.cpp:328-379
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_loudness_bank_update_settings: NULL bank");
    MI_REQUIRE(b->sample_rate != 0, MI_ESTATE, "mi_loudness_bank_update_settings: set_sample_rate() first");
    return update_settings(b, mi::as_stream(stream));
}

int mi_loudness_bank_latency(const mi_loudness_bank_t *b, uint32_t *samples)
{
    MI_REQUIRE(b != nullptr && samples != nullptr, MI_EINVAL, "mi_loudness_bank_latency: bad argument");
    *samples = uint32_t((b->period_ms * 0.001f) * float(b->sample_rate));
    return MI_OK;
}

// what the launches of a call take by value from the host: head of the lines, distance to the next exact re-summation
static uint64_t loudness_bank_positions(const void *bank)
{
    const mi_loudness_bank *b = static_cast<const mi_loudness_bank *>(bank);
    uint64_t h = mi::position_mix(b->head, b->ms_refresh);
    return mi::position_mix(h, (uint64_t(b->raw_left != 0) << 2) | (uint64_t(b->upd_time || b->upd_filters) << 1) | uint64_t(b->cfg_dirty));
}

static int loudness_process(mi_loudness_bank_t *b, float *out, float *ch_out, const float *in, size_t count,
                            size_t out_stride, size_t in_stride, float gain, bool remember, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_loudness_bank_process: NULL bank");
    if (count == 0)
        return MI_OK;
    {
        const int rc = mi::capture_touch(mi::as_stream(stream), b, "loudness meter", loudness_bank_positions);
        if (rc != MI_OK)
            return rc;
    }
    MI_REQUIRE(in != nullptr, MI_EINVAL, "mi_loudness_bank_process: NULL input");
    MI_REQUIRE(b->sample_rate != 0 && b->d_data != nullptr, MI_ESTATE, "mi_loudness_bank_process: set_sample_rate() first");
    hipStream_t st = mi::as_stream(stream);
    int r = update_settings(b, st);
    if (r != MI_OK)
        return r;
    const uint32_t room = b->data_size - b->period;         // cells that may be written before the window's tail is reached
    cfg_pack pack;
    for (uint32_t c = 0; c < CFG_BY_VALUE; ++c)
        pack.c[c] = (c < b->channels) ? b->cfg[c] : chan_cfg{ 0.0f, 0.0f, 0, 0 };
    const uint32_t interval = std::max<uint32_t>(BUFFER_SIZE << 2, b->period >> 2);     // between exact re-summations (:496-503)
    size_t offset = 0;
    while (offset < count)
    {
        size_t n = count - offset;
        n = std::min<size_t>(n, MAX_BLOCK);                 // <= interval: at most one re-summation falls into a block
        n = std::min<size_t>(n, (room > BUFFER_SIZE) ? room : BUFFER_SIZE);
        const uint32_t refresh_at = (b->ms_refresh < n) ? b->ms_refresh : UINT32_MAX;
        if (n > b->cap)
        {
            (void)hipFree(b->d_flt); (void)hipFree(b->d_msbuf);
            b->d_flt = b->d_msbuf = nullptr;
            b->cap = 0;
            const size_t cap = std::min<size_t>(std::max<size_t>(n, 4096), MAX_BLOCK);
            MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_flt), size_t(b->rows) * cap * sizeof(float)));
            MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_msbuf), size_t(b->rows) * cap * sizeof(float)));
            b->cap = cap;
        }
        // the weighting filter of every row, then the block kernel (one workgroup per meter)
        r = mi_biquad_bank_process(b->filters, b->d_flt, in + offset, n, b->cap, in_stride, stream);
        if (r != MI_OK)
            return r;
        // four samples per lane when nothing straddles a 16-byte cell; else one per lane, 512 threads x 8 passes for the
        // long blocks (<= 128 VGPRs: two workgroups on a CU)
        float *o_main = out ? out + offset : nullptr, *o_ch = ch_out ? ch_out + offset : nullptr;
        const bool vec4 = (n % 4 == 0) && (b->head % 4 == 0) && (b->period % 4 == 0) && (out_stride % 4 == 0) &&
                          ((reinterpret_cast<uintptr_t>(o_main) | reinterpret_cast<uintptr_t>(o_ch)) % 16 == 0) &&
                          !mi::test_path("loudness_scalar");          // (the one-sample kernel that unaligned calls take anyway)
        float *loud_dst = remember ? b->d_loud : static_cast<float *>(nullptr);
        #define MI_LARGS o_main, o_ch, out_stride, b->d_flt, b->cap, b->d_data, b->data_size, b->head, b->period, b->avg, b->d_ms, \
                         b->d_msbuf, b->cap, b->d_cfg, pack, b->channels, uint32_t(n), gain, loud_dst, refresh_at, \
                         b->d_segsum, (b->raw_left == 0) ? 1 : 0
        if (vec4)
        {
            if (n <= 1024)      hipLaunchKernelGGL((loudness_block4_kernel<256, 1>), dim3(b->meters), dim3(256), 0, st, MI_LARGS);
            else if (n <= 2048) hipLaunchKernelGGL((loudness_block4_kernel<512, 1>), dim3(b->meters), dim3(512), 0, st, MI_LARGS);
            else                hipLaunchKernelGGL((loudness_block4_kernel<512, MAX_BLOCK / 2048>), dim3(b->meters), dim3(512), 0, st, MI_LARGS);
        }
        else
        {
            auto kernel = (n <= 512) ? loudness_block_kernel<256, 2> : (n <= 1024) ? loudness_block_kernel<256, 4> :
                          (n <= 2048) ? loudness_block_kernel<512, 4> : loudness_block_kernel<512, MAX_BLOCK / 512>;
            hipLaunchKernelGGL(kernel, dim3(b->meters), dim3((n <= 1024) ? 256 : 512), 0, st, MI_LARGS);
        }
        #undef MI_LARGS
        MI_HIP_CHECK(hipGetLastError());
        b->raw_left = (b->raw_left > n) ? b->raw_left - n : 0;
        b->head = (b->head + uint32_t(n)) & (b->data_size - 1);
        b->ms_refresh = (refresh_at != UINT32_MAX) ? interval - (uint32_t(n) - refresh_at) : b->ms_refresh - uint32_t(n);
        offset += n;
    }
    return MI_OK;
}

// process(out, count): also remembers the last loudness value for loudness() (LoudnessMeter.cpp:485)
int mi_loudness_bank_process(mi_loudness_bank_t *b, float *out, float *ch_out, const float *in, size_t count,
                             size_t out_stride, size_t in_stride, void *stream)
{
    return loudness_process(b, out, ch_out, in, count, out_stride, in_stride, 1.0f, true, stream);
}

// process(out, count, gain): every output times gain; loudness() keeps the value of the last call WITHOUT gain -- the
// reference's second form does not touch fLoudness (LoudnessMeter.cpp:518-564)
int mi_loudness_bank_process_gain(mi_loudness_bank_t *b, float *out, float *ch_out, const float *in, size_t count,
                                  size_t out_stride, size_t in_stride, float gain, void *stream)
{
    return loudness_process(b, out, ch_out, in, count, out_stride, in_stride, gain, false, stream);
}

int mi_loudness_bank_loudness(mi_loudness_bank_t *b, float *loudness, void *stream)
{
    MI_REQUIRE(b != nullptr && loudness != nullptr, MI_EINVAL, "mi_loudness_bank_loudness: bad argument");
    hipStream_t st = mi::as_stream(stream);
    MI_HIP_CHECK(hipMemcpyAsync(loudness, b->d_loud, b->meters * sizeof(float), hipMemcpyDeviceToHost, st));
    MI_HIP_CHECK(hipStreamSynchronize(st));
    return MI_OK;
}

} // extern "C"

// =====================================================================================================================
// Integrated loudness: lsp::dspu::ILUFSMeter for many meters
// (reference: src/main/meters/ILUFSMeter.cpp:113-211 init, :291-322 set_sample_rate, :324-353 gated / infinite
//  loudness, :355-470 process, :472-513 update_settings, :515-560 clear).
// The weighting filter is the biquad bank; a block quarter's square sums are one reduction per row; at every quarter
// boundary one workgroup per meter does the reference's gating arithmetic over the meter's history of gating blocks.
namespace
{
    using mi_meters::ilufs_state;
    using mi_meters::ilufs_piece;
    using mi_meters::ilufs_pieces;
    using mi_meters::MIN_GATING_BLOCKS;

    // The pieces of one process() call behind the weighting filter's launch, one workgroup per meter
    // (ilufs_device.h; calls that qualify do the same inside the filter's launch, mi::biquad_bank_sumsq)
    __global__ __launch_bounds__(LT)
    void ilufs_call_kernel(float *block, float *seg, const ilufs_pieces pieces,
                           const chan_cfg *__restrict__ cfg, uint32_t channels, float *out, size_t out_stride,
                           ilufs_state *st, float gain, float *hist, uint32_t size, uint32_t ms_int, float avg)
    {
        __shared__ float s_sum[4];
        __shared__ uint32_t s_cnt[4];
        __shared__ float s_val;
        __shared__ float s_chan[2 * LT];
        static_assert(LT == mi_meters::VTH, "one real thread per virtual one");
        const mi_meters::ilufs_early<LT> early = mi_meters::ilufs_ask<LT>(blockIdx.x, block, cfg, channels, st, hist, size, ms_int);
        mi_meters::ilufs_call_body<LT>(blockIdx.x, block, seg, pieces, cfg, channels, out, out_stride, st, gain, hist, size,
                                       ms_int, avg, s_sum, s_cnt, s_val, s_chan, early);
    }
} // namespace

struct mi_ilufs_bank
{
    uint32_t    meters = 0, channels = 0, rows = 0;
    uint32_t    sample_rate = 0, block_size = 0, block_offset = 0, block_part = 0, ms_size = 0, ms_int = 0;
    float       block_period = 400.0f, int_time = 60.0f, max_int_time = 60.0f, avg = 1.0f;
    int         weighting = MI_BS_WEIGHT_K;
    bool        upd_filters = true, upd_time = true, cfg_dirty = true, blk_full = false;
    std::vector<chan_cfg> cfg;
    mi_biquad_bank_t *filters = nullptr;
    float      *d_block = nullptr, *d_hist = nullptr, *d_seg = nullptr;
    uint32_t   *d_arrived = nullptr;        // [meters] rows counted in by the weighting filter's launch (ilufs_device.h)
    ilufs_state *d_state = nullptr;
    chan_cfg   *d_cfg = nullptr;
};

namespace
{
    int ilufs_clear_blocks(mi_ilufs_bank *b, hipStream_t st)           // clear_block_buffers(), ILUFSMeter.cpp:515-526
    {
        MI_HIP_CHECK(hipMemsetAsync(b->d_block, 0, size_t(b->rows) * 4 * sizeof(float), st));
        MI_HIP_CHECK(hipMemsetAsync(b->d_seg, 0, size_t(b->rows) * 4 * sizeof(float), st));
        if (b->d_hist != nullptr)
            MI_HIP_CHECK(hipMemsetAsync(b->d_hist, 0, size_t(b->meters) * b->ms_size * sizeof(float), st));
        b->blk_full = false;
        return MI_OK;
    }

    int ilufs_update(mi_ilufs_bank *b, hipStream_t st)                 // update_settings(), ILUFSMeter.cpp:472-513
    {
        if (b->upd_time)
        {
            const float int_time = (b->int_time < b->max_int_time) ? b->int_time : b->max_int_time;
            const size_t blk = size_t((b->block_period * 0.25f * 0.001f) * float(b->sample_rate));
            if (int_time > 0)
            {
                const size_t total = size_t(int_time * float(b->sample_rate));
                const long v = (long(total) - long(blk) * 2 - 1) / long(blk ? blk : 1);
                b->ms_int = uint32_t((v > 1) ? v : 1);
            }
            else
                b->ms_int = 0;
            // nMSCount = min(nMSCount, nMSInt) for every meter: done on the host side of the state
            std::vector<ilufs_state> h(b->meters);
            MI_HIP_CHECK(hipMemcpyAsync(h.data(), b->d_state, h.size() * sizeof(ilufs_state), hipMemcpyDeviceToHost, st));
            MI_HIP_CHECK(hipStreamSynchronize(st));
            for (ilufs_state &s : h)
                s.count = (s.count < b->ms_int) ? s.count : b->ms_int;
            MI_HIP_CHECK(hipMemcpyAsync(b->d_state, h.data(), h.size() * sizeof(ilufs_state), hipMemcpyHostToDevice, st));
            MI_HIP_CHECK(hipStreamSynchronize(st));
            b->upd_time = false;
        }
        if (b->upd_filters)
        {
            static const uint32_t types[6] = { MI_FLT_NONE, MI_FLT_A_WEIGHTED, MI_FLT_B_WEIGHTED, MI_FLT_C_WEIGHTED,
                                               MI_FLT_D_WEIGHTED, MI_FLT_K_WEIGHTED };
            mi_filter_params_t fp;
            fp.nType = types[b->weighting]; fp.nSlope = 0; fp.fFreq = 0.0f; fp.fFreq2 = 0.0f; fp.fGain = 1.0f; fp.fQuality = 0.0f;
            mi::design d;
            d.cascades.reserve(mi::CHAINS_MAX + 1);
            mi::design_filter(&d, &fp, b->sample_rate);
            for (uint32_t r = 0; r < b->rows; ++r)
            {
                const int e = mi_biquad_bank_set_chains(b->filters, r, d.sections.data(), uint32_t(d.sections.size()), 1);
                if (e != MI_OK)
                    return e;
            }
            const int e = mi_biquad_bank_commit(b->filters, st);
            if (e != MI_OK)
                return e;
            b->upd_filters = false;
        }
        if (b->cfg_dirty)
        {
            MI_HIP_CHECK(hipMemcpyAsync(b->d_cfg, b->cfg.data(), b->channels * sizeof(chan_cfg), hipMemcpyHostToDevice, st));
            MI_HIP_CHECK(hipStreamSynchronize(st));
            b->cfg_dirty = false;
        }
        return MI_OK;
    }
} // namespace

extern "C" {

int mi_ilufs_bank_create(mi_ilufs_bank_t **bank, uint32_t meters, uint32_t channels, float max_int_time, float block_period_ms)
{
    MI_REQUIRE(bank != nullptr, MI_EINVAL, "mi_ilufs_bank_create: NULL result pointer");
    *bank = nullptr;
    MI_REQUIRE(meters > 0 && channels > 0 && block_period_ms > 0.0f, MI_EINVAL, "mi_ilufs_bank_create: bad argument");
    MI_REQUIRE(mi_dspu_device_count() > 0, MI_ENODEV, "no HIP device available (there is no CPU fallback)");
    mi_ilufs_bank *b = new (std::nothrow) mi_ilufs_bank();
    MI_REQUIRE(b != nullptr, MI_ENOMEM, "mi_ilufs_bank_create: out of host memory");
    b->meters = meters;
    b->channels = channels;
    b->rows = meters * channels;
    b->block_period = block_period_ms;
    b->int_time = b->max_int_time = max_int_time;
    b->cfg.assign(channels, chan_cfg{ 0.0f, 1.0f, 1, 0 });
    if (channels == 1)
        b->cfg[0].weight = channel_weighting(MI_BS_CHANNEL_CENTER);
    else if (channels == 2)
        b->cfg[0].weight = b->cfg[1].weight = channel_weighting(MI_BS_CHANNEL_LEFT);
    int r = mi_biquad_bank_create(&b->filters, b->rows, 4);
    hipError_t e = hipSuccess;
    if (r == MI_OK)
    {
        e = hipMalloc(reinterpret_cast<void **>(&b->d_block), size_t(b->rows) * 4 * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_seg), size_t(b->rows) * 4 * sizeof(float));
        if (e == hipSuccess) e = hipMemset(b->d_seg, 0, size_t(b->rows) * 4 * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_state), meters * sizeof(ilufs_state));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_arrived), meters * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemset(b->d_arrived, 0, meters * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_cfg), channels * sizeof(chan_cfg));
        if (e == hipSuccess) e = hipMemset(b->d_block, 0, size_t(b->rows) * 4 * sizeof(float));
        if (e == hipSuccess) e = hipMemset(b->d_state, 0, meters * sizeof(ilufs_state));
    }
    if (r != MI_OK || e != hipSuccess)
    {
        mi_ilufs_bank_destroy(b);
        return (r != MI_OK) ? r : mi::fail(MI_EHIP, "mi_ilufs_bank_create: %s", hipGetErrorString(e));
    }
    *bank = b;
    return MI_OK;
}

int mi_ilufs_bank_destroy(mi_ilufs_bank_t *b)
{
    if (b == nullptr)
        return MI_OK;
    mi_biquad_bank_destroy(b->filters);
    (void)hipFree(b->d_block); (void)hipFree(b->d_hist); (void)hipFree(b->d_seg); (void)hipFree(b->d_state); (void)hipFree(b->d_cfg); (void)hipFree(b->d_arrived);
    delete b;
    return MI_OK;
}

int mi_ilufs_bank_clear(mi_ilufs_bank_t *b, void *stream)                      // ILUFSMeter.cpp:547-560
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_ilufs_bank_clear: NULL bank");
    hipStream_t st = mi::as_stream(stream);
    int r = mi_biquad_bank_reset(b->filters, UINT32_MAX, stream);
    if (r == MI_OK)
        r = ilufs_clear_blocks(b, st);
    if (r != MI_OK)
        return r;
    MI_HIP_CHECK(hipMemsetAsync(b->d_state, 0, b->meters * sizeof(ilufs_state), st));
    // the rows counted in by a weighting-filter launch (biquad_sumsq_ilufs_kernel): a launch that ended on an error would
    // leave its meter's count short of a full round and no later launch could elect a last workgroup
    MI_HIP_CHECK(hipMemsetAsync(b->d_arrived, 0, b->meters * sizeof(uint32_t), st));
    b->block_offset = 0;
    b->block_part = 0;
    return MI_OK;
}

int mi_ilufs_bank_set_sample_rate(mi_ilufs_bank_t *b, uint32_t sample_rate, void *stream)      // ILUFSMeter.cpp:291-322
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_ilufs_bank_set_sample_rate: NULL bank");
    if (b->sample_rate == sample_rate)
        return MI_OK;
    const size_t blk = size_t((b->block_period * 0.25f * 0.001f) * float(sample_rate));          // 75 % overlap
    MI_REQUIRE(blk > 0, MI_EINVAL, "mi_ilufs_bank_set_sample_rate: block period too short for %u Hz", sample_rate);
    const size_t int_count = (size_t(b->max_int_time * float(sample_rate)) + blk - 1) / blk;
    size_t blocks = (int_count > MIN_GATING_BLOCKS) ? int_count : MIN_GATING_BLOCKS;
    blocks = (blocks + 3) & ~size_t(3);                     // align_size(.., DEFAULT_ALIGN = 16 bytes)
    (void)hipFree(b->d_hist);
    b->d_hist = nullptr;
    MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_hist), size_t(b->meters) * blocks * sizeof(float)));
    b->avg = 0.25f / float(blk);
    b->sample_rate = sample_rate;
    b->block_size = uint32_t(blk);
    b->ms_size = uint32_t(blocks);
    b->upd_filters = b->upd_time = true;
    return mi_ilufs_bank_clear(b, stream);
}

int mi_ilufs_bank_set_integration_period(mi_ilufs_bank_t *b, float period, void *stream)        // ILUFSMeter.cpp:264-289
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_ilufs_bank_set_integration_period: NULL bank");
    const float lo = b->block_period * 0.001f;
    period = (period < lo) ? lo : (period > b->max_int_time) ? b->max_int_time : period;
    if (b->int_time == period)
        return MI_OK;
    hipStream_t st = mi::as_stream(stream);
    if (b->int_time <= 0)
    {
        // nMSCount = 0 (ILUFSMeter.cpp:276); the history position and the loudness being held stay as they are
        std::vector<ilufs_state> h(b->meters);
        MI_HIP_CHECK(hipMemcpyAsync(h.data(), b->d_state, h.size() * sizeof(ilufs_state), hipMemcpyDeviceToHost, st));
        MI_HIP_CHECK(hipStreamSynchronize(st));
        for (ilufs_state &v : h)
            v.count = 0;
        MI_HIP_CHECK(hipMemcpyAsync(b->d_state, h.data(), h.size() * sizeof(ilufs_state), hipMemcpyHostToDevice, st));
        MI_HIP_CHECK(hipStreamSynchronize(st));
        const int r = ilufs_clear_blocks(b, st);
        if (r != MI_OK)
            return r;
    }
    else if (period <= 0.0f)
    {
        const int r = ilufs_clear_blocks(b, st);
        if (r != MI_OK)
            return r;
    }
    b->int_time = period;
    b->upd_time = true;
    return MI_OK;
}

int mi_ilufs_bank_set_weighting(mi_ilufs_bank_t *b, int weighting)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_ilufs_bank_set_weighting: NULL bank");
    MI_REQUIRE(weighting >= MI_BS_WEIGHT_NONE && weighting <= MI_BS_WEIGHT_K, MI_EINVAL, "mi_ilufs_bank_set_weighting: bad weighting %d", weighting);
    if (weighting == b->weighting)
        return MI_OK;
    b->weighting = weighting;
    b->upd_filters = true;
    return MI_OK;
}

int mi_ilufs_bank_set_designation(mi_ilufs_bank_t *b, uint32_t channel, int designation)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_ilufs_bank_set_designation: NULL bank");
    MI_REQUIRE(channel < b->channels, MI_EINVAL, "mi_ilufs_bank_set_designation: channel %u out of range", channel);
    b->cfg[channel].weight = channel_weighting(designation);
    b->cfg_dirty = true;
    return MI_OK;
}

int mi_ilufs_bank_set_active(mi_ilufs_bank_t *b, uint32_t channel, int active)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_ilufs_bank_set_active: NULL bank");
    MI_REQUIRE(channel < b->channels, MI_EINVAL, "mi_ilufs_bank_set_active: channel %u out of range", channel);
    b->cfg[channel].enabled = active ? 1 : 0;
    b->cfg_dirty = true;
    for (uint32_t m = 0; m < b->meters; ++m)                // the filter of a disabled channel is not run (ILUFSMeter.cpp:370)
    {
        const int r = mi_biquad_bank_set_row_enabled(b->filters, m * b->channels + channel, active);
        if (r != MI_OK)
            return r;
    }
    return MI_OK;
}

// what the launches of a call take by value from the host: the position inside the gating block
static uint64_t ilufs_bank_positions(const void *bank)
{
    const mi_ilufs_bank *b = static_cast<const mi_ilufs_bank *>(bank);
    uint64_t h = mi::position_mix(b->block_offset, b->block_part);
    return mi::position_mix(h, (uint64_t(b->upd_time || b->upd_filters) << 1) | uint64_t(b->cfg_dirty));
}

int mi_ilufs_bank_process(mi_ilufs_bank_t *b, float *out, const float *in, size_t count, size_t out_stride,
                          size_t in_stride, float gain, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_ilufs_bank_process: NULL bank");
    if (count == 0)
        return MI_OK;
    {
        const int rc = mi::capture_touch(mi::as_stream(stream), b, "integrated loudness meter", ilufs_bank_positions);
        if (rc != MI_OK)
            return rc;
    }
    MI_REQUIRE(in != nullptr, MI_EINVAL, "mi_ilufs_bank_process: NULL input");
    MI_REQUIRE(b->sample_rate != 0 && b->d_hist != nullptr, MI_ESTATE, "mi_ilufs_bank_process: set_sample_rate() first");
    hipStream_t st = mi::as_stream(stream);
    int r = ilufs_update(b, st);
    if (r != MI_OK)
        return r;
    // update_settings() ends with `nFlags = 0` (ILUFSMeter.cpp:519) and F_BLK_FULL is one of those flags: every process()
    // call starts with the flag cleared, and gating blocks are evaluated only from the point where the quarter counter
    // wraps inside the same call.  Reproduced so that a host that switches banks sees the same meter readings.
    b->blk_full = false;
    // Up to four pieces (runs inside one quarter of a gating block) per pair of launches: the weighting filter walks
    // them as one block and leaves the pieces' sums of squares, the meters' kernel does the bookkeeping piece by piece.
    size_t offset = 0;
    while (offset < count)
    {
        ilufs_pieces pcs;
        pcs.count = 0;
        uint32_t ends[3] = { 0, 0, 0 };
        size_t taken = 0;
        while (pcs.count < 4 && offset + taken < count && taken < (size_t(1) << 30))
        {
            const size_t n = std::min<size_t>(std::min<size_t>(count - offset - taken, b->block_size - b->block_offset),
                                              (size_t(1) << 30) - taken);
            ilufs_piece &pc = pcs.p[pcs.count];
            pc.offset = uint32_t(taken);
            pc.n = uint32_t(n);
            pc.part = b->block_part;
            pc.gate = 0;
            pc.zero_part = -1;
            b->block_offset += uint32_t(n);
            taken += n;
            if (b->block_offset >= b->block_size)           // a quarter of a gating block is complete
            {
                b->block_offset = 0;
                if (++b->block_part >= 4)
                {
                    b->block_part = 0;
                    b->blk_full = true;
                }
                pc.gate = b->blk_full ? 1 : 0;
                pc.zero_part = int(b->block_part);
            }
            if (pcs.count < 3)
                ends[pcs.count] = uint32_t(taken);
            ++pcs.count;
        }
        for (uint32_t k = pcs.count; k < 3; ++k)
            ends[k] = uint32_t(taken);
        mi_meters::ilufs_epilogue ep;
        ep.arrived = b->d_arrived; ep.block = b->d_block; ep.cfg = b->d_cfg; ep.channels = b->channels;
        ep.out = out ? out + offset : nullptr; ep.out_stride = out_stride; ep.st = b->d_state; ep.gain = gain;
        ep.hist = b->d_hist; ep.size = b->ms_size; ep.ms_int = b->ms_int; ep.avg = b->avg; ep.pieces = pcs;
        bool rode = false;
        r = mi::biquad_bank_sumsq(b->filters, in + offset, in_stride, taken, ends, b->d_seg, st, &ep, &rode);
        if (r != MI_OK)
            return r;
        if (!rode)                                          // short or ragged call: the bookkeeping in a launch of its own
        {
            hipLaunchKernelGGL(ilufs_call_kernel, dim3(b->meters), dim3(LT), 0, st, b->d_block, b->d_seg, pcs, b->d_cfg, b->channels,
                               ep.out, out_stride, b->d_state, gain, b->d_hist, b->ms_size, b->ms_int, b->avg);
            MI_HIP_CHECK(hipGetLastError());
        }
        offset += taken;
    }
    return MI_OK;
}

int mi_ilufs_bank_needs_update(const mi_ilufs_bank_t *b, int *pending)                        // ILUFSMeter.h:244
{
    MI_REQUIRE(b != nullptr && pending != nullptr, MI_EINVAL, "mi_ilufs_bank_needs_update: bad argument");
    *pending = (b->upd_filters || b->upd_time) ? 1 : 0;
    return MI_OK;
}

int mi_ilufs_bank_update_settings(mi_ilufs_bank_t *b, void *stream)                           // ILUFSMeter.cpp:472-513
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_ilufs_bank_update_settings: NULL bank");
    MI_REQUIRE(b->sample_rate != 0, MI_ESTATE, "mi_ilufs_bank_update_settings: set_sample_rate() first");
    return ilufs_update(b, mi::as_stream(stream));
}

int mi_ilufs_bank_loudness(mi_ilufs_bank_t *b, float *loudness, void *stream)
{
    MI_REQUIRE(b != nullptr && loudness != nullptr, MI_EINVAL, "mi_ilufs_bank_loudness: bad argument");
    hipStream_t st = mi::as_stream(stream);
    std::vector<ilufs_state> h(b->meters);
    MI_HIP_CHECK(hipMemcpyAsync(h.data(), b->d_state, h.size() * sizeof(ilufs_state), hipMemcpyDeviceToHost, st));
    MI_HIP_CHECK(hipStreamSynchronize(st));
    for (uint32_t m = 0; m < b->meters; ++m)
        loudness[m] = h[m].loudness;
    return MI_OK;
}

int mi_ilufs_bank_history(mi_ilufs_bank_t *b, float *hist, uint32_t *size, uint32_t *head, uint32_t *count, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_ilufs_bank_history: NULL bank");
    if (size)
        *size = b->ms_size;
    if (hist == nullptr && head == nullptr && count == nullptr)
        return MI_OK;
    hipStream_t st = mi::as_stream(stream);
    std::vector<ilufs_state> h(b->meters);
    MI_HIP_CHECK(hipMemcpyAsync(h.data(), b->d_state, h.size() * sizeof(ilufs_state), hipMemcpyDeviceToHost, st));
    if (hist != nullptr && b->d_hist != nullptr && b->ms_size > 0)
        MI_HIP_CHECK(hipMemcpyAsync(hist, b->d_hist, size_t(b->meters) * b->ms_size * sizeof(float), hipMemcpyDeviceToHost, st));
    MI_HIP_CHECK(hipStreamSynchronize(st));
    for (uint32_t m = 0; m < b->meters; ++m)
    {
        if (head)  head[m] = h[m].head;
        if (count) count[m] = h[m].count;
    }
    return MI_OK;
}

} // extern "C"

