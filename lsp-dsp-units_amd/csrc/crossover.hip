// IIR crossover bank: lsp::dspu::Crossover for `channels` channels on top of the biquad cascade bank
// (reference: src/main/util/Crossover.cpp:71-160 init, :342-449 reconfigure, :451-498 process, :500-590 freq_chart).
//
// A split point of the reference owns an Equalizer in IIR mode (the low-pass plus the all-pass filters of the split
// points above it, all sections in one FilterBank) and a Filter (the high-pass).  Here each of them is one
// mi_biquad_bank over all channels; the designer (host C++) produces the sections.  The dataflow of process() is the
// reference's: band k = LPF_k(src), src = HPF_k(src), last band = src.
#include "mi_common.h"
#include "host/filter_design.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace
{
    constexpr float SPEC_FREQ_MIN = 10.0f, SPEC_FREQ_MAX = 24000.0f;    // LSP_DSP_UNITS_SPEC_FREQ_MIN/MAX (const.h:30-31)

    enum xover_type { X_LPF, X_HPF, X_APF };

    // Crossover::select_filter / select_slope (Crossover.cpp:162-198)
    uint32_t select_filter(xover_type type, int mode, uint32_t slope)
    {
        const bool bt = (mode == MI_CROSS_MODE_BT);
        if (slope == 1)                                                 // CROSS_SLOPE_LR2
            switch (type)
            {
                case X_LPF: return bt ? MI_FLT_BT_RLC_LOPASS  : MI_FLT_MT_RLC_LOPASS;
                case X_HPF: return bt ? MI_FLT_BT_RLC_HIPASS  : MI_FLT_MT_RLC_HIPASS;
                default:    return bt ? MI_FLT_BT_RLC_ALLPASS : MI_FLT_MT_RLC_ALLPASS;
            }
        switch (type)
        {
            case X_LPF: return bt ? MI_FLT_BT_LRX_LOPASS  : MI_FLT_MT_LRX_LOPASS;
            case X_HPF: return bt ? MI_FLT_BT_LRX_HIPASS  : MI_FLT_MT_LRX_HIPASS;
            default:    return bt ? MI_FLT_BT_LRX_ALLPASS : MI_FLT_MT_LRX_ALLPASS;
        }
    }

    uint32_t select_slope(xover_type type, uint32_t slope)
    {
        if (slope == 1)
            return (type == X_APF) ? 1 : 2;
        return slope - 1;
    }

    __global__ __launch_bounds__(256)
    void scale_kernel(float *dst, size_t dst_stride, const float *src, size_t src_stride, float k, size_t count)
    {
        const uint32_t ch = blockIdx.y;
        for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < count; i += size_t(gridDim.x) * 256)
            dst[size_t(ch) * dst_stride + i] = src[size_t(ch) * src_stride + i] * k;
    }

    struct split_t
    {
        uint32_t    band_id = 0, slope = 0;
        float       freq = 0.0f;
        int         mode = MI_CROSS_MODE_BT;
        mi_biquad_bank_t *lpf = nullptr, *hpf = nullptr;
        uint32_t    lpf_cap = 0, hpf_cap = 0;
        std::vector<mi::design> lpf_designs;        // filter 0 = the low-pass, then the all-pass filters
        mi::design  hpf_design;
        uint32_t    hpf_type = 0, hpf_slope = 0;    // Filter::update clears the state when these change (Filter.cpp:157-158)
    };

    struct band_t
    {
        float   gain = 1.0f, start = 0.0f, end = 0.0f;
        bool    enabled = false;
        int     p_start = -1, p_end = -1;           // plan positions of the split points around the band
    };
} // namespace

struct mi_crossover_bank
{
    uint32_t    channels = 0, splits = 0, sample_rate = 48000;     // LSP_DSP_UNITS_DEFAULT_SAMPLE_RATE
    bool        dirty = true, clear = true;
    std::vector<split_t> split;
    std::vector<band_t>  band;
    std::vector<int>     plan;                      // indices into split[], ascending frequency
    float      *d_hpf = nullptr;                    // [channels][cap]
    size_t      cap = 0;
};

namespace
{
    int ensure_bank(mi_biquad_bank_t **bank, uint32_t *cap, uint32_t channels, size_t sections)
    {
        if (*bank != nullptr && sections <= *cap)
            return MI_OK;
        mi_biquad_bank_destroy(*bank);
        *bank = nullptr;
        *cap = uint32_t((sections + 7) & ~size_t(7));
        if (*cap == 0)
            *cap = 8;
        return mi_biquad_bank_create(bank, channels, *cap);
    }

    int set_all_channels(mi_biquad_bank_t *bank, uint32_t channels, const std::vector<mi_biquad_x1_t> &sec, bool clear)
    {
        for (uint32_t c = 0; c < channels; ++c)
        {
            const int r = mi_biquad_bank_set_chains(bank, c, sec.data(), uint32_t(sec.size()), clear ? 1 : 0);
            if (r != MI_OK)
                return r;
        }
        return MI_OK;
    }

    mi_filter_params_t make_params(uint32_t type, uint32_t slope, float freq, float gain)
    {
        mi_filter_params_t fp;
        fp.nType = type; fp.nSlope = slope; fp.fFreq = freq; fp.fFreq2 = freq; fp.fGain = gain; fp.fQuality = 0.0f;
        return fp;
    }

    int reconfigure(mi_crossover_bank *b, void *stream)             // Crossover.cpp:342-449
    {
        if (!b->dirty)
            return MI_OK;
        b->plan.clear();
        for (uint32_t i = 0; i < b->splits; ++i)
            if (b->split[i].slope != 0)
                b->plan.push_back(int(i));
        for (band_t &bd : b->band)
            bd.enabled = false;
        // the reference's exchange sort (Crossover.cpp:357-361)
        for (size_t si = 0; si + 1 < b->plan.size(); ++si)
            for (size_t sj = si + 1; sj < b->plan.size(); ++sj)
                if (b->split[b->plan[sj]].freq < b->split[b->plan[si]].freq)
                    std::swap(b->plan[si], b->plan[sj]);

        band_t *left = &b->band[0];
        left->start = SPEC_FREQ_MIN;
        left->enabled = true;
        left->p_start = -1;
        const size_t np = b->plan.size();
        for (size_t i = 0; i < np; ++i)
        {
            split_t &sp = b->split[b->plan[i]];
            band_t *right = &b->band[sp.band_id];
            left->end = sp.freq;
            left->p_end = int(i);
            right->start = sp.freq;
            right->p_start = int(i);
            right->enabled = true;

            // low-pass with the gain of the band on its left, then the all-pass filters of the split points above
            std::vector<mi_biquad_x1_t> sec;
            sp.lpf_designs.assign(1 + (np - 1 - i), mi::design());
            mi_filter_params_t fp = make_params(select_filter(X_LPF, sp.mode, sp.slope), select_slope(X_LPF, sp.slope), sp.freq, left->gain);
            sp.lpf_designs[0].cascades.reserve(mi::CHAINS_MAX + 1);
            mi::design_filter(&sp.lpf_designs[0], &fp, b->sample_rate);
            sec.insert(sec.end(), sp.lpf_designs[0].sections.begin(), sp.lpf_designs[0].sections.end());
            for (size_t j = i + 1; j < np; ++j)
            {
                const split_t &x = b->split[b->plan[j]];
                fp = make_params(select_filter(X_APF, x.mode, x.slope), select_slope(X_APF, x.slope), x.freq, 1.0f);
                mi::design &d = sp.lpf_designs[j - i];
                d.cascades.reserve(mi::CHAINS_MAX + 1);
                mi::design_filter(&d, &fp, b->sample_rate);
                sec.insert(sec.end(), d.sections.begin(), d.sections.end());
            }
            int r = ensure_bank(&sp.lpf, &sp.lpf_cap, b->channels, sec.size());
            if (r == MI_OK) r = set_all_channels(sp.lpf, b->channels, sec, b->clear);      // Equalizer: EF_CLEAR only on rate/mode changes
            if (r == MI_OK) r = mi_biquad_bank_commit(sp.lpf, stream);
            if (r != MI_OK)
                return r;

            // high-pass: unity gain except for the last split point, which carries the gain of the last band;
            // LR2 inverts the polarity (Crossover.cpp:412-417)
            float g = (i + 1 < np) ? 1.0f : right->gain;
            if (sp.slope == 1)
                g = -g;
            fp = make_params(select_filter(X_HPF, sp.mode, sp.slope), select_slope(X_HPF, sp.slope), sp.freq, g);
            const bool hclear = b->clear || fp.nType != sp.hpf_type || fp.nSlope != sp.hpf_slope;      // Filter.cpp:157-158
            sp.hpf_type = fp.nType;
            sp.hpf_slope = fp.nSlope;
            sp.hpf_design = mi::design();
            sp.hpf_design.cascades.reserve(mi::CHAINS_MAX + 1);
            mi::design_filter(&sp.hpf_design, &fp, b->sample_rate);
            r = ensure_bank(&sp.hpf, &sp.hpf_cap, b->channels, sp.hpf_design.sections.size());
            if (r == MI_OK) r = set_all_channels(sp.hpf, b->channels, sp.hpf_design.sections, hclear);
            if (r == MI_OK) r = mi_biquad_bank_commit(sp.hpf, stream);
            if (r != MI_OK)
                return r;
            left = right;
        }
        left->end = float(b->sample_rate) * 0.5f;
        left->p_end = -1;
        b->dirty = false;
        b->clear = false;
        return MI_OK;
    }
} // namespace

extern "C" {

int mi_crossover_bank_create(mi_crossover_bank_t **bank, uint32_t channels, uint32_t bands)
{
    MI_REQUIRE(bank != nullptr, MI_EINVAL, "mi_crossover_bank_create: NULL result pointer");
    *bank = nullptr;
    MI_REQUIRE(channels > 0 && bands >= 1, MI_EINVAL, "mi_crossover_bank_create: channels and bands must be > 0");  // Crossover.cpp:73-74
    MI_REQUIRE(mi_dspu_device_count() > 0, MI_ENODEV, "no HIP device available (there is no CPU fallback)");
    mi_crossover_bank *b = new (std::nothrow) mi_crossover_bank();
    MI_REQUIRE(b != nullptr, MI_ENOMEM, "mi_crossover_bank_create: out of host memory");
    b->channels = channels;
    b->splits = bands - 1;
    b->split.resize(b->splits);
    b->band.resize(bands);
    const float step = logf(SPEC_FREQ_MAX / SPEC_FREQ_MIN) / float(bands);                 // Crossover.cpp:113
    for (uint32_t i = 0; i < b->splits; ++i)
    {
        b->split[i].band_id = i + 1;
        b->split[i].slope = 0;
        b->split[i].freq = SPEC_FREQ_MIN * expf(float(i + 1) * step);
        b->split[i].mode = MI_CROSS_MODE_BT;
    }
    for (uint32_t i = 0; i < bands; ++i)
    {
        b->band[i].gain = 1.0f;
        b->band[i].start = (i == 0) ? SPEC_FREQ_MIN : b->split[i - 1].freq;
        b->band[i].end = (i < b->splits) ? b->split[i].freq : float(b->sample_rate >> 1);
    }
    *bank = b;
    return MI_OK;
}

int mi_crossover_bank_destroy(mi_crossover_bank_t *b)
{
    if (b == nullptr)
        return MI_OK;
    for (split_t &sp : b->split)
    {
        mi_biquad_bank_destroy(sp.lpf);
        mi_biquad_bank_destroy(sp.hpf);
    }
    (void)hipFree(b->d_hpf);
    delete b;
    return MI_OK;
}

int mi_crossover_bank_set_sample_rate(mi_crossover_bank_t *b, uint32_t sample_rate)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_crossover_bank_set_sample_rate: NULL bank");
    if (b->sample_rate == sample_rate)
        return MI_OK;
    b->sample_rate = sample_rate;
    b->band[b->splits].end = float(sample_rate >> 1);
    b->dirty = b->clear = true;                     // the filters' set_sample_rate() clears their state
    return MI_OK;
}

int mi_crossover_bank_set_slope(mi_crossover_bank_t *b, uint32_t split, uint32_t slope)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_crossover_bank_set_slope: NULL bank");
    if (split >= b->splits || slope == b->split[split].slope)      // silently ignored like the reference
        return MI_OK;
    MI_REQUIRE(slope <= 9, MI_EINVAL, "mi_crossover_bank_set_slope: slope %u outside CROSS_SLOPE_OFF..LR32", slope);
    b->split[split].slope = slope;
    b->dirty = true;
    return MI_OK;
}

int mi_crossover_bank_set_frequency(mi_crossover_bank_t *b, uint32_t split, float freq)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_crossover_bank_set_frequency: NULL bank");
    if (split >= b->splits || freq == b->split[split].freq)
        return MI_OK;
    b->split[split].freq = freq;
    b->dirty = true;
    return MI_OK;
}

int mi_crossover_bank_set_mode(mi_crossover_bank_t *b, uint32_t split, int mode)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_crossover_bank_set_mode: NULL bank");
    if (split >= b->splits || mode == b->split[split].mode)
        return MI_OK;
    b->split[split].mode = (mode == MI_CROSS_MODE_MT) ? MI_CROSS_MODE_MT : MI_CROSS_MODE_BT;
    b->dirty = true;
    return MI_OK;
}

int mi_crossover_bank_set_gain(mi_crossover_bank_t *b, uint32_t band, float gain)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_crossover_bank_set_gain: NULL bank");
    if (band > b->splits || gain == b->band[band].gain)
        return MI_OK;
    b->band[band].gain = gain;
    b->dirty = true;
    return MI_OK;
}

int mi_crossover_bank_needs_reconfiguration(const mi_crossover_bank_t *b, int *pending)     // Crossover.h:352
{
    MI_REQUIRE(b != nullptr && pending != nullptr, MI_EINVAL, "mi_crossover_bank_needs_reconfiguration: bad argument");
    *pending = b->dirty ? 1 : 0;
    return MI_OK;
}

int mi_crossover_bank_get_split(const mi_crossover_bank_t *b, uint32_t split, uint32_t *slope, float *freq, int *mode)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_crossover_bank_get_split: NULL bank");
    MI_REQUIRE(split < b->splits, MI_EINVAL, "mi_crossover_bank_get_split: split %u out of range", split);
    if (slope) *slope = b->split[split].slope;
    if (freq)  *freq = b->split[split].freq;
    if (mode)  *mode = b->split[split].mode;
    return MI_OK;
}

int mi_crossover_bank_get_band(mi_crossover_bank_t *b, uint32_t band, float *gain, float *start, float *end, int *active,
                               void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_crossover_bank_get_band: NULL bank");
    MI_REQUIRE(band <= b->splits, MI_EINVAL, "mi_crossover_bank_get_band: band %u out of range", band);
    const int r = reconfigure(b, stream);
    if (r != MI_OK)
        return r;
    if (gain)   *gain = b->band[band].gain;
    if (start)  *start = b->band[band].start;
    if (end)    *end = b->band[band].end;
    if (active) *active = (band == 0) ? 1 : (b->band[band].enabled ? 1 : 0);               // Crossover.cpp:311-319
    return MI_OK;
}

int mi_crossover_bank_process(mi_crossover_bank_t *b, float *const *band_out, const float *in, size_t samples,
                              size_t out_stride, size_t in_stride, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_crossover_bank_process: NULL bank");
    if (samples == 0)
        return MI_OK;
    MI_REQUIRE(band_out != nullptr && in != nullptr, MI_EINVAL, "mi_crossover_bank_process: NULL buffer");
    hipStream_t st = mi::as_stream(stream);
    int r = reconfigure(b, stream);
    if (r != MI_OK)
        return r;
    const size_t np = b->plan.size();
    if (np == 0)                                                    // Crossover.cpp:486-490
    {
        if (band_out[0] != nullptr)
        {
            const unsigned gx = unsigned(std::min<size_t>((samples + 255) / 256, 64));
            hipLaunchKernelGGL(scale_kernel, dim3(gx, b->channels), dim3(256), 0, st, band_out[0], out_stride, in, in_stride,
                               b->band[0].gain, samples);
            MI_HIP_CHECK(hipGetLastError());
        }
        return MI_OK;
    }
    if (samples > b->cap)
    {
        (void)hipFree(b->d_hpf);
        b->d_hpf = nullptr;
        b->cap = 0;
        MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_hpf), size_t(b->channels) * samples * sizeof(float)));
        b->cap = samples;
    }
    // The whole plan in one launch when the call qualifies (Crossover.cpp:451-498 as a chain on a block held in
    // registers): band k = LPF_k(src) is a branch, src = HPF_k(src) runs in place, the last high-pass is the last band.
    // The source is read once and every band written once: 4 + 4 * bands bytes per sample instead of 8 per filter.
    const bool unfused = mi::test_path("crossover_unfused");          // (one launch per filter: what calls that do not qualify take anyway)
    if (!unfused)
    {
        std::vector<mi::biquad_chain_stage> chain;
        uint32_t lband = 0;
        for (size_t i = 0; i < np; ++i)
        {
            split_t &sp = b->split[b->plan[i]];
            if (band_out[lband] != nullptr)
                chain.push_back({ sp.lpf, band_out[lband], out_stride, 1 });
            chain.push_back({ sp.hpf, (i + 1 == np) ? band_out[sp.band_id] : nullptr, out_stride, 0 });
            lband = sp.band_id;
        }
        r = mi::biquad_chain_process(chain.data(), int(chain.size()), in, in_stride, samples, st);
        if (r <= 0)
            return r;                                               // issued (or failed): nothing left to do
    }
    const float *src = in;
    size_t src_stride = in_stride;
    uint32_t left = 0;
    for (size_t i = 0; i < np; ++i)
    {
        split_t &sp = b->split[b->plan[i]];
        if (band_out[left] != nullptr)                              // no handler: the low-pass is skipped, its state rests
            if ((r = mi_biquad_bank_process(sp.lpf, band_out[left], src, samples, out_stride, src_stride, stream)) != MI_OK)
                return r;
        // the high-passed signal travels on; the last one IS the last band
        float *dst = b->d_hpf;
        size_t dst_stride = b->cap;
        if (i + 1 == np && band_out[sp.band_id] != nullptr)
        {
            dst = band_out[sp.band_id];
            dst_stride = out_stride;
        }
        if ((r = mi_biquad_bank_process(sp.hpf, dst, src, samples, dst_stride, src_stride, stream)) != MI_OK)
            return r;
        src = dst;
        src_stride = dst_stride;
        left = sp.band_id;
    }
    return MI_OK;
}

// `blocks` consecutive process() calls in one C call: block i reads in[i] and writes its bands to band_out[i * (splits + 1) +
// band].  Runs of blocks go out as ONE launch (biquad_stream_chain_kernel: the plan's chain on a stream of sub-blocks, the
// loads of a sub-block and the stores of the bands underneath the sections of its neighbours) -- the same bits as the calls
// one by one.  Crossover.cpp:451-498 per block.
int mi_crossover_bank_process_blocks(mi_crossover_bank_t *b, float *const *band_out, const float *const *in, size_t blocks,
                                     size_t samples, size_t out_stride, size_t in_stride, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_crossover_bank_process_blocks: NULL bank");
    if (samples == 0 || blocks == 0)
        return MI_OK;
    MI_REQUIRE(band_out != nullptr && in != nullptr, MI_EINVAL, "mi_crossover_bank_process_blocks: NULL pointer table");
    for (size_t i = 0; i < blocks; ++i)
        MI_REQUIRE(in[i] != nullptr, MI_EINVAL, "mi_crossover_bank_process_blocks: NULL input of block %zu", i);
    hipStream_t st = mi::as_stream(stream);
    int r = reconfigure(b, stream);
    if (r != MI_OK)
        return r;
    const size_t nb = size_t(b->splits) + 1, np = b->plan.size();
    auto one_by_one = [&]() -> int {
        for (size_t i = 0; i < blocks; ++i)
        {
            const int rc = mi_crossover_bank_process(b, band_out + i * nb, in[i], samples, out_stride, in_stride, stream);
            if (rc != MI_OK)
                return rc;
        }
        return MI_OK;
    };
    const bool unfused = mi::test_path("crossover_unfused");
    if (np == 0 || blocks < 2 || unfused)
        return one_by_one();
    // the same bands have a handler in every block (a band without one is skipped, its low-pass rests: Crossover.cpp:462-466)
    for (size_t i = 1; i < blocks; ++i)
        for (size_t k = 0; k < nb; ++k)
            if ((band_out[i * nb + k] == nullptr) != (band_out[k] == nullptr))
                return one_by_one();
    std::vector<mi::biquad_chain_stage> chain;
    std::vector<int> slot;
    std::vector<uint32_t> band_of_slot;
    auto slot_of = [&](uint32_t band) -> int {
        if (band_out[band] == nullptr)
            return -1;
        band_of_slot.push_back(band);
        return int(band_of_slot.size()) - 1;
    };
    uint32_t lband = 0;
    for (size_t i = 0; i < np; ++i)
    {
        split_t &sp = b->split[b->plan[i]];
        if (band_out[lband] != nullptr)
        {
            chain.push_back({ sp.lpf, nullptr, out_stride, 1 });
            slot.push_back(slot_of(lband));
        }
        chain.push_back({ sp.hpf, nullptr, out_stride, 0 });
        slot.push_back((i + 1 == np) ? slot_of(sp.band_id) : -1);
        lband = sp.band_id;
    }
    const int outs = int(band_of_slot.size());
    if (outs == 0)
        return one_by_one();
    std::vector<float *> po(blocks * size_t(outs));
    for (size_t i = 0; i < blocks; ++i)
        for (int s = 0; s < outs; ++s)
            po[i * outs + s] = band_out[i * nb + band_of_slot[s]];
    r = mi::biquad_chain_process_blocks(chain.data(), slot.data(), int(chain.size()), outs, po.data(), in, blocks, samples,
                                        out_stride, in_stride, st);
    return (r == 1) ? one_by_one() : r;
}

int mi_crossover_bank_freq_chart(mi_crossover_bank_t *b, uint32_t band, float *c, const float *f, size_t count, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_crossover_bank_freq_chart: NULL bank");
    MI_REQUIRE(band <= b->splits, MI_EINVAL, "mi_crossover_bank_freq_chart: band %u out of range", band);
    MI_REQUIRE(c != nullptr && (f != nullptr || count == 0), MI_EINVAL, "mi_crossover_bank_freq_chart: NULL buffer");
    const int r = reconfigure(b, stream);
    if (r != MI_OK)
        return r;
    const band_t &bd = b->band[band];
    if (!bd.enabled)                                                // Crossover.cpp:570-571
    {
        std::fill(c, c + 2 * count, 0.0f);
        return MI_OK;
    }
    if (b->plan.empty())
    {
        for (size_t i = 0; i < count; ++i) { c[2 * i] = 1.0f; c[2 * i + 1] = 0.0f; }
        return MI_OK;
    }
    std::vector<float> t(2 * count);
    auto chart_mul = [&](const mi::design &d, bool first)
    {
        if (d.mode == mi::FM_BYPASS)
        {
            if (first)
                for (size_t i = 0; i < count; ++i) { c[2 * i] = 1.0f; c[2 * i + 1] = 0.0f; }
            return;
        }
        if (first)
        {
            mi::freq_chart(d, c, f, count);
            return;
        }
        mi::freq_chart(d, t.data(), f, count);
        for (size_t i = 0; i < count; ++i)
        {
            const float re = c[2 * i] * t[2 * i] - c[2 * i + 1] * t[2 * i + 1];
            const float im = c[2 * i] * t[2 * i + 1] + c[2 * i + 1] * t[2 * i];
            c[2 * i] = re;
            c[2 * i + 1] = im;
        }
    };
    if (bd.p_end < 0)                                               // last band: the high-pass before it
        chart_mul(b->split[b->plan[bd.p_start]].hpf_design, true);
    else if (bd.p_start < 0)                                        // first band: the whole low-pass equalizer
    {
        const split_t &sp = b->split[b->plan[bd.p_end]];
        bool first = true;
        for (const mi::design &d : sp.lpf_designs)
        {
            chart_mul(d, first);
            first = false;
        }
    }
    else                                                            // in between: HPF x the low-pass filter alone (Crossover.cpp:531-533)
    {
        chart_mul(b->split[b->plan[bd.p_start]].hpf_design, true);
        chart_mul(b->split[b->plan[bd.p_end]].lpf_designs[0], false);
    }
    return MI_OK;
}

} // extern "C"
