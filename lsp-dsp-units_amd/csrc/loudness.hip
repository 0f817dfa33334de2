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
#include "host/filter_design.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace
{
    constexpr uint32_t BUFFER_SIZE = 0x400;                 // LoudnessMeter.cpp:32
    constexpr int      LT = 256;                            // threads per meter
    constexpr uint32_t MAX_BLOCK = 4096;                    // samples per launch: 16 register-resident passes of LT samples

    // bs::channel_weighting (src/main/misc/broadcast.cpp:32-55)
    float channel_weighting(int designation)
    {
        if (designation >= 6 && designation <= 11)
            return 1.41f;                                   // ~ +1.5 dB
        if (designation == MI_BS_CHANNEL_LFE1 || designation == MI_BS_CHANNEL_LFE2)
            return 0.0f;
        return 1.0f;
    }

    struct chan_cfg
    {
        float   weight;
        float   link;
        int     enabled;
        int     unbound;        // LoudnessMeter: no input bound -- the channel is left out of the block (:421-422) but stays
    };                          // enabled for refresh_rms() and clear(); always 0 for the integrated meter

    // exact window sums (refresh_rms): ms[row] = sum of the last `period` cells behind head
    __global__ __launch_bounds__(LT)
    void loudness_refresh_kernel(float *ms, const float *__restrict__ data, uint32_t size, uint32_t head, uint32_t period,
                                 const chan_cfg *__restrict__ cfg, uint32_t channels)
    {
        __shared__ float part[LT];
        const uint32_t row = blockIdx.x, tid = threadIdx.x;
        if (!cfg[row % channels].enabled)
            return;
        const float *d = data + size_t(row) * size;
        const uint32_t tail = (head + size - period) & (size - 1);
        // 16-byte cells of the ring that cover [tail, tail + period); the cells at both ends are cut to the window
        const uint32_t lead = tail & 3u, first = tail - lead, cells = (lead + period + 3u) >> 2;
        float s0 = 0.0f, s1 = 0.0f;
        for (uint32_t q = tid; q < cells; q += LT)
        {
            const float4 v = *reinterpret_cast<const float4 *>(d + ((first + 4u * q) & (size - 1)));
            const uint32_t at = 4u * q;                     // position of the cell's first float, counted from `first`
            const bool inner = at >= lead && at + 4u <= lead + period;
            if (inner)
            {
                s0 += v.x + v.z;
                s1 += v.y + v.w;
            }
            else
            {
                const float e[4] = { v.x, v.y, v.z, v.w };
                #pragma unroll
                for (uint32_t k = 0; k < 4; ++k)
                    if (at + k >= lead && at + k < lead + period)
                        s0 += e[k];
            }
        }
        const float s = s0 + s1;
        part[tid] = s;
        __syncthreads();
        for (int w = LT / 2; w > 0; w >>= 1)
        {
            if (int(tid) < w)
                part[tid] += part[tid + w];
            __syncthreads();
        }
        if (tid == 0)
            ms[row] = part[0];
    }

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

    // one block of `n` <= E * LT samples of one meter
    template <uint32_t E>                                   // passes: sample j = i * LT + tid lives in register i of thread tid
    __global__ __launch_bounds__(LT)
    void loudness_block_kernel(float *out, float *ch_out, size_t out_stride, const float *__restrict__ flt, size_t flt_stride,
                               float *data, uint32_t size, uint32_t head, uint32_t period, float avg, float *ms,
                               float *msbuf, size_t msbuf_stride, const chan_cfg *__restrict__ cfg, uint32_t channels,
                               uint32_t n, float gain, float *loud)
    {
        __shared__ float sq[E * LT];                        // this channel's squares (new values)
        __shared__ __align__(16) float wtot[E][LT / 64];    // sums of the waves of every pass
        const uint32_t meter = blockIdx.x, tid = threadIdx.x, mask = size - 1;
        const uint32_t lane = tid & 63, wave = tid >> 6;
        const uint32_t tail = (head + size - period) & mask;
        float mix[E];                                       // weighted sum of the channels' mean squares
        #pragma unroll
        for (uint32_t i = 0; i < E; ++i)
            mix[i] = 0.0f;
        uint32_t mixed = 0;
        for (uint32_t c = 0; c < channels; ++c)
        {
            const chan_cfg cc = cfg[c];
            if (!cc.enabled || cc.unbound)
                continue;
            const uint32_t row = meter * channels + c;
            float *line = data + size_t(row) * size;
            const float *x = flt + size_t(row) * flt_stride;
            float d[E], old[E];
            #pragma unroll
            for (uint32_t i = 0; i < E; ++i)                // every load of the block is issued before anything waits
            {
                const uint32_t j = i * LT + tid;
                d[i] = (j < n) ? x[j] : 0.0f;
                old[i] = (j < n && j < period) ? line[(tail + j) & mask] : 0.0f;
            }
            const float start = ms[row];
            __syncthreads();                                // the previous channel is through with sq[] and wtot[]
            #pragma unroll
            for (uint32_t i = 0; i < E; ++i)                // dsp::sqr2 into the line (LoudnessMeter.cpp:428-436)
            {
                const uint32_t j = i * LT + tid;
                d[i] *= d[i];
                if (j < n)
                {
                    sq[j] = d[i];
                    line[(head + j) & mask] = d[i];
                }
            }
            __syncthreads();
            // ms_j = ms_(j-1) + (new_j - old_j): an inclusive scan of every pass inside the wave, the sums of the waves
            // through LDS, then the running sum carried from pass to pass
            #pragma unroll
            for (uint32_t i = 0; i < E; ++i)
            {
                const uint32_t j = i * LT + tid;
                if (j < n)
                    d[i] -= (j >= period) ? sq[j - period] : old[i];
                d[i] = wave_scan(d[i]);
                if (lane == 63)
                    wtot[i][wave] = d[i];
            }
            __syncthreads();
            float carry = start;
            float *mb = msbuf + size_t(row) * msbuf_stride;
            #pragma unroll
            for (uint32_t i = 0; i < E; ++i)
            {
                const uint32_t j = i * LT + tid;
                const float4 t = *reinterpret_cast<const float4 *>(wtot[i]);
                static_assert(LT / 64 == 4, "one float4 of wave sums per pass");
                float before = carry;
                if (wave > 0) before += t.x;
                if (wave > 1) before += t.y;
                if (wave > 2) before += t.z;
                const float m = avg * (before + d[i]);      // vMS[j] = fAvgCoeff * ms
                if (j < n && ch_out != nullptr)
                    mb[j] = m;
                mix[i] = (mixed > 0) ? fmaf(m, cc.weight, mix[i]) : m * cc.weight;     // fmadd_k3 / mul_k3
                carry += ((t.x + t.y) + t.z) + t.w;
            }
            if (tid == 0)
                ms[row] = carry;                            // the running sum goes on with the next block
            ++mixed;
        }
        // ssqrt1: sqrt of the non-negative part; then the outputs
        #pragma unroll
        for (uint32_t i = 0; i < E; ++i)
        {
            const uint32_t j = i * LT + tid;
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
            const chan_cfg cc = cfg[c];
            if (!cc.enabled || cc.unbound)                  // (the reference hands an unbound channel's stale buffer on: nothing here)
                continue;
            const uint32_t row = meter * channels + c;
            const float *mb = msbuf + size_t(row) * msbuf_stride;
            float *o = ch_out + size_t(row) * out_stride;
            #pragma unroll
            for (uint32_t i = 0; i < E; ++i)
            {
                const uint32_t j = i * LT + tid;
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
    (void)hipFree(b->d_loud); (void)hipFree(b->d_cfg);
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
    b->sample_rate = sample_rate;
    b->data_size = len;
    b->head = 0;
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
    const int run = (b->cfg[channel].enabled && !unbound) ? 1 : 0;
    for (uint32_t m = 0; m < b->meters; ++m)
    {
        const int r = mi_biquad_bank_set_row_enabled(b->filters, m * b->channels + channel, run);
        if (r != MI_OK)
            return r;
    }
    return MI_OK;
}

int mi_loudness_bank_latency(const mi_loudness_bank_t *b, uint32_t *samples)
{
    MI_REQUIRE(b != nullptr && samples != nullptr, MI_EINVAL, "mi_loudness_bank_latency: bad argument");
    *samples = uint32_t((b->period_ms * 0.001f) * float(b->sample_rate));
    return MI_OK;
}

static int loudness_process(mi_loudness_bank_t *b, float *out, float *ch_out, const float *in, size_t count,
                            size_t out_stride, size_t in_stride, float gain, bool remember, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_loudness_bank_process: NULL bank");
    if (count == 0)
        return MI_OK;
    MI_REQUIRE(in != nullptr, MI_EINVAL, "mi_loudness_bank_process: NULL input");
    MI_REQUIRE(b->sample_rate != 0 && b->d_data != nullptr, MI_ESTATE, "mi_loudness_bank_process: set_sample_rate() first");
    hipStream_t st = mi::as_stream(stream);
    int r = update_settings(b, st);
    if (r != MI_OK)
        return r;
    const uint32_t room = b->data_size - b->period;         // cells that may be written before the window's tail is reached
    size_t offset = 0;
    while (offset < count)
    {
        if (b->ms_refresh == 0)                             // refresh_rms(), LoudnessMeter.cpp:381-407
        {
            hipLaunchKernelGGL(loudness_refresh_kernel, dim3(b->rows), dim3(LT), 0, st, b->d_ms, b->d_data, b->data_size, b->head,
                               b->period, b->d_cfg, b->channels);
            MI_HIP_CHECK(hipGetLastError());
            const uint32_t a = BUFFER_SIZE << 2, q = b->period >> 2;
            b->ms_refresh = (a > q) ? a : q;
        }
        size_t n = count - offset;
        n = std::min<size_t>(n, b->ms_refresh);
        n = std::min<size_t>(n, MAX_BLOCK);
        n = std::min<size_t>(n, (room > BUFFER_SIZE) ? room : BUFFER_SIZE);
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
        auto kernel = (n <= 2 * LT) ? loudness_block_kernel<2> : (n <= 4 * LT) ? loudness_block_kernel<4> :
                      (n <= 8 * LT) ? loudness_block_kernel<8> : loudness_block_kernel<MAX_BLOCK / LT>;
        hipLaunchKernelGGL(kernel, dim3(b->meters), dim3(LT), 0, st,
                           out ? out + offset : nullptr, ch_out ? ch_out + offset : nullptr, out_stride, b->d_flt, b->cap,
                           b->d_data, b->data_size, b->head, b->period, b->avg, b->d_ms, b->d_msbuf, b->cap, b->d_cfg,
                           b->channels, uint32_t(n), gain, remember ? b->d_loud : static_cast<float *>(nullptr));
        MI_HIP_CHECK(hipGetLastError());
        b->head = (b->head + uint32_t(n)) & (b->data_size - 1);
        b->ms_refresh -= uint32_t(n);
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
    constexpr float GATING_ABS_THRESH = 1.17246530458e-07f;         // ILUFSMeter.cpp:39
    constexpr uint32_t MIN_GATING_BLOCKS = 64;                      // :55

    struct ilufs_state { uint32_t head, count; float loudness; uint32_t pad; };

    // mean of the last `count` history entries above the ABSOLUTE gate (compute_gated_loudness, ILUFSMeter.cpp:324-341:
    // its `threshold` argument is not used by the reference -- both gating stages compare with GATING_ABS_THRESH, so the
    // relative stage returns what the absolute stage returned; one pass gives the reference's result for both)
    __device__ float gated_mean(const float *hist, uint32_t size, uint32_t head, uint32_t count, float *s_sum, uint32_t *s_cnt)
    {
        const uint32_t tid = threadIdx.x;
        const uint32_t tail = (head + size - count) % size;
        float s = 0.0f;
        uint32_t c = 0;
        for (uint32_t j = tid; j < count; j += LT)
        {
            const float l = hist[(tail + j) % size];
            if (l > GATING_ABS_THRESH)
            {
                s += l;
                ++c;
            }
        }
        s_sum[tid] = s;
        s_cnt[tid] = c;
        __syncthreads();
        for (int w = LT / 2; w > 0; w >>= 1)
        {
            if (int(tid) < w)
            {
                s_sum[tid] += s_sum[tid + w];
                s_cnt[tid] += s_cnt[tid + w];
            }
            __syncthreads();
        }
        const float r = (s_cnt[0] > 0) ? s_sum[0] / float(s_cnt[0]) : 0.0f;
        __syncthreads();
        return r;
    }

    // a gating block is complete (ILUFSMeter.cpp:402-458); the workgroup of the meter
    __device__ void ilufs_gate(uint32_t meter, ilufs_state *st, float *hist, uint32_t size, uint32_t ms_int, const float *block,
                               const chan_cfg *__restrict__ cfg, uint32_t channels, float avg,
                               float *s_sum, uint32_t *s_cnt, float &s_val)
    {
        const uint32_t tid = threadIdx.x;
        float *h = hist + size_t(meter) * size;
        ilufs_state me = st[meter];
        if (tid == 0)
        {
            float loudness = 0.0f;                          // every channel's block enters, enabled or not (:407-414)
            for (uint32_t c = 0; c < channels; ++c)
            {
                const float *blk = block + (size_t(meter) * channels + c) * 4;
                loudness += cfg[c].weight * ((blk[0] + blk[1] + blk[2] + blk[3]) * avg);
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
            __syncthreads();
            loudness = gated_mean(h, size, me.head, me.count, s_sum, s_cnt);
        }
        else                                                // since the last clear(): running mean of the gated blocks
        {
            if (loudness > GATING_ABS_THRESH)
            {
                if (me.count >= 0x100)                      // floating-point overflow protection (:440-444)
                {
                    for (uint32_t j = tid; j < size; j += LT)
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
                float s = 0.0f;
                for (uint32_t j = tid; j < size; j += LT)
                    s += mult * h[j];
                s_sum[tid] = s;
                __syncthreads();
                for (int w = LT / 2; w > 0; w >>= 1)
                {
                    if (int(tid) < w)
                        s_sum[tid] += s_sum[tid + w];
                    __syncthreads();
                }
                loudness = s_sum[0];
            }
            else
                loudness = 0.0f;
        }
        if (tid == 0)
        {
            me.loudness = sqrtf(loudness);
            st[meter] = me;
        }
    }

    // One piece of a block quarter, one workgroup per meter: the held loudness value into the output row
    // (ILUFSMeter.cpp:386-387), vBlock[row][part] += the sum of squares of every enabled row's filtered samples
    // (:372-384), then -- when the piece ends the quarter -- the gating arithmetic of a complete block and the reset of
    // the quarter that is filled next (:402-466).
    __global__ __launch_bounds__(LT)
    void ilufs_piece_kernel(float *block, uint32_t part, const float *__restrict__ flt, size_t flt_stride, uint32_t n,
                            const chan_cfg *__restrict__ cfg, uint32_t channels, float *out, size_t out_stride,
                            ilufs_state *st, float gain, int gate, int zero_part, float *hist, uint32_t size, uint32_t ms_int,
                            float avg)
    {
        __shared__ float s_sum[LT];
        __shared__ uint32_t s_cnt[LT];
        __shared__ float s_val;
        const uint32_t meter = blockIdx.x, tid = threadIdx.x;
        if (out != nullptr)
        {
            const float v = st[meter].loudness * gain;
            for (uint32_t i = tid; i < n; i += LT)
                out[size_t(meter) * out_stride + i] = v;
        }
        for (uint32_t c = 0; c < channels && n > 0; ++c)
        {
            if (!cfg[c].enabled)
                continue;
            const uint32_t row = meter * channels + c;
            const float *x = flt + size_t(row) * flt_stride;
            float s = 0.0f;
            for (uint32_t i = tid; i < n; i += LT)
                s = fmaf(x[i], x[i], s);
            s_sum[tid] = s;
            __syncthreads();
            for (int w = LT / 2; w > 0; w >>= 1)
            {
                if (int(tid) < w)
                    s_sum[tid] += s_sum[tid + w];
                __syncthreads();
            }
            if (tid == 0)
                block[row * 4 + part] += s_sum[0];
            __syncthreads();
        }
        if (gate)
            ilufs_gate(meter, st, hist, size, ms_int, block, cfg, channels, avg, s_sum, s_cnt, s_val);   // thread 0 reads its own sums
        if (zero_part >= 0)
        {
            __syncthreads();
            for (uint32_t c = tid; c < channels; c += LT)
                block[(meter * channels + c) * 4 + uint32_t(zero_part)] = 0.0f;
        }
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
    float      *d_block = nullptr, *d_hist = nullptr, *d_flt = nullptr;
    ilufs_state *d_state = nullptr;
    chan_cfg   *d_cfg = nullptr;
    size_t      cap = 0;
};

namespace
{
    int ilufs_clear_blocks(mi_ilufs_bank *b, hipStream_t st)           // clear_block_buffers(), ILUFSMeter.cpp:515-526
    {
        MI_HIP_CHECK(hipMemsetAsync(b->d_block, 0, size_t(b->rows) * 4 * sizeof(float), st));
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
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_state), meters * sizeof(ilufs_state));
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
    (void)hipFree(b->d_block); (void)hipFree(b->d_hist); (void)hipFree(b->d_flt); (void)hipFree(b->d_state); (void)hipFree(b->d_cfg);
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

int mi_ilufs_bank_process(mi_ilufs_bank_t *b, float *out, const float *in, size_t count, size_t out_stride,
                          size_t in_stride, float gain, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_ilufs_bank_process: NULL bank");
    if (count == 0)
        return MI_OK;
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
    size_t offset = 0;
    while (offset < count)
    {
        size_t n = std::min<size_t>(count - offset, b->block_size - b->block_offset);
        n = std::min<size_t>(n, 16384);
        if (n > 0)
        {
            if (n > b->cap)
            {
                (void)hipFree(b->d_flt);
                b->d_flt = nullptr;
                b->cap = 0;
                const size_t cap = std::min<size_t>(std::max<size_t>(n, 4096), 16384);
                MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_flt), size_t(b->rows) * cap * sizeof(float)));
                b->cap = cap;
            }
            r = mi_biquad_bank_process(b->filters, b->d_flt, in + offset, n, b->cap, in_stride, stream);
            if (r != MI_OK)
                return r;
            b->block_offset += uint32_t(n);
        }
        int gate = 0, zero_part = -1;
        const uint32_t part = b->block_part;
        if (b->block_offset >= b->block_size)               // a quarter of a gating block is complete
        {
            b->block_offset = 0;
            if (++b->block_part >= 4)
            {
                b->block_part = 0;
                b->blk_full = true;
            }
            gate = b->blk_full ? 1 : 0;
            zero_part = int(b->block_part);
        }
        if (n > 0 || zero_part >= 0)
        {
            hipLaunchKernelGGL(ilufs_piece_kernel, dim3(b->meters), dim3(LT), 0, st, b->d_block, part, b->d_flt, b->cap, uint32_t(n),
                               b->d_cfg, b->channels, out ? out + offset : nullptr, out_stride, b->d_state, gain, gate, zero_part,
                               b->d_hist, b->ms_size, b->ms_int, b->avg);
            MI_HIP_CHECK(hipGetLastError());
        }
        offset += n;
    }
    return MI_OK;
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

