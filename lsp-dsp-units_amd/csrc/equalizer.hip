// Equalizer bank for gfx950: `channels` x lsp::dspu::Equalizer
// (reference: src/main/filters/Equalizer.cpp:67-160 init, :243-358 reconfigure, :460-571 process).
//
// Every mode is built from the banks of this library; nothing here is a new streaming kernel:
//   EQM_IIR  all filters' sections in one biquad bank per channel (Equalizer.cpp:256-267,466-470)
//   EQM_FIR  linear-phase FIR synthesised from the IIR impulse response (Equalizer.cpp:281-289,328-349)
//   EQM_FFT  same FIR synthesised from the product of the filters' frequency charts (:290-324)
//            -> streamed as  convolver(FIR) followed by a delay of nFirSize samples: the reference buffers one
//            block of nFirSize samples before each fastconv_parse_apply (:477-511), which is exactly that delay;
//            total latency nFirSize + nFirSize/2 (:347)
//   EQM_SPM  magnitude mask inside a 50 %-overlap STFT with no analysis window and a squared-cosine synthesis
//            window (:350-357,523-562) -> the spectral bank with those windows; latency nFirSize (:355)
// Reconfiguration (designer, frequency charts) runs on the host per changed channel; the FIR synthesis
// (window, FFT, magnitude, zero-phase IFFT, rotation, window) runs on the device for all channels at once.
#include "mi_common.h"
#include "fft_device.h"
#include "host/filter_design.h"

#include <cmath>
#include <cstdint>
#include <vector>

namespace mi
{
    void make_window(float *dst, size_t n, int type);
}

namespace
{
    using namespace mi_fft;

    // mag[k] = | FFT( ir[n] * wnd[n] ) |, k < N            (Equalizer.cpp:283-288)
    template <int LOGN>
    __global__ __launch_bounds__(plan<LOGN>::T)
    void eq_ir_to_magnitude_kernel(float *mag, const float *__restrict__ ir, size_t ir_stride,
                                   const float *__restrict__ wnd_tail, const float2 *__restrict__ tw)
    {
        using PL = plan<LOGN>;
        constexpr int N = PL::N, T = PL::T;
        __shared__ float2 buf[plan<LOGN>::BUF], scr[plan<LOGN>::BUF];    // sequence + the padding of the intermediate layouts
        const int ch = blockIdx.x, tid = threadIdx.x;
        fft_tw<LOGN> ft;
        load_fft_tw<LOGN>(ft, tw, TWN / N, tid);
        finish_fft_tw<LOGN>(ft);
        for (int n = tid; n < N; n += T)
            buf[n] = make_float2(ir[size_t(ch) * ir_stride + n] * wnd_tail[n], 0.0f);
        __syncthreads();
        fft_lds<LOGN, false>(buf, scr, ft, tid);
        for (int k = tid; k < N; k += T)
            mag[size_t(ch) * N + k] = sqrtf(buf[k].x * buf[k].x + buf[k].y * buf[k].y);
    }

    // taps[i] = wnd[i] * Re(IFFT(mag))[(i + N/2) mod N]   (Equalizer.cpp:330-336)
    template <int LOGN>
    __global__ __launch_bounds__(plan<LOGN>::T)
    void eq_magnitude_to_fir_kernel(float *taps, const float *__restrict__ mag, const float *__restrict__ wnd,
                                    const float2 *__restrict__ tw)
    {
        using PL = plan<LOGN>;
        constexpr int N = PL::N, T = PL::T;
        __shared__ float2 buf[plan<LOGN>::BUF], scr[plan<LOGN>::BUF];    // sequence + the padding of the intermediate layouts
        const int ch = blockIdx.x, tid = threadIdx.x;
        fft_tw<LOGN> ft;
        load_fft_tw<LOGN>(ft, tw, TWN / N, tid);
        finish_fft_tw<LOGN>(ft);
        for (int k = tid; k < N; k += T)
            buf[k] = make_float2(mag[size_t(ch) * N + k], 0.0f);
        __syncthreads();
        fft_lds<LOGN, true>(buf, scr, ft, tid);
        const float scale = 1.0f / float(N);
        for (int i = tid; i < N; i += T)
            taps[size_t(ch) * N + i] = buf[(i + N / 2) & (N - 1)].x * scale * wnd[i];
    }

    #define MI_LOGN_SWITCH(ln, CALL)                    \
        switch (ln)                                     \
        {                                               \
            case 5:  { CALL(5);  break; }               \
            case 6:  { CALL(6);  break; }               \
            case 7:  { CALL(7);  break; }               \
            case 8:  { CALL(8);  break; }               \
            case 9:  { CALL(9);  break; }               \
            case 10: { CALL(10); break; }               \
            case 11: { CALL(11); break; }               \
            case 12: { CALL(12); break; }               \
            default: { CALL(13); break; }               \
        }

    enum { EF_REBUILD = 1, EF_CLEAR = 2 };
} // namespace

struct mi_equalizer_bank
{
    uint32_t    channels = 0, filters = 0, fir_rank = 0, fir_size = 0;
    uint32_t    sample_rate = 0, actual_sample_rate = 0, latency = 0;
    int         mode = MI_EQM_BYPASS;
    uint32_t    flags = EF_REBUILD | EF_CLEAR;
    std::vector<mi_filter_params_t> params;         // [channels][filters]
    std::vector<uint8_t>            dirty;          // per channel
    mi_biquad_bank_t    *biquads = nullptr;
    mi_convolver_bank_t *conv = nullptr;
    mi_delay_bank_t     *delay = nullptr;
    mi_spectral_bank_t  *spm = nullptr;
    float      *d_ir = nullptr, *d_mag = nullptr, *d_taps = nullptr, *d_wnd2n_tail = nullptr, *d_wndn = nullptr;
    const float2 *d_tw = nullptr;
    std::vector<float> h_mag;                       // [channels][N] host magnitudes (FFT/SPM modes)
    bool        streaming_ready = false;
    uint32_t    primed = 0;                         // samples pushed into the (cleared) delay line so far, up to N
    bool        smooth = false;                     // EF_SMOOTH: retunes cross-fade over one block (Equalizer.cpp:339-343)
};

namespace
{
    uint32_t eff_sample_rate(const mi_equalizer_bank *b)
    {
        return (b->actual_sample_rate != 0) ? b->actual_sample_rate : b->sample_rate;     // Equalizer.h:271
    }

    // sections of all filters of one channel, in filter order (shared bank, Equalizer.cpp:256-259)
    void channel_sections(const mi_equalizer_bank *b, uint32_t ch, std::vector<mi_biquad_x1_t> *out,
                          std::vector<mi::design> *designs)
    {
        out->clear();
        designs->resize(b->filters);
        for (uint32_t f = 0; f < b->filters; ++f)
        {
            mi::design &d = (*designs)[f];
            d.cascades.reserve(mi::CHAINS_MAX + 1);
            mi::design_filter(&d, &b->params[size_t(ch) * b->filters + f], b->sample_rate);
            out->insert(out->end(), d.sections.begin(), d.sections.end());
        }
    }

    // |H| of one channel on the linear grid 0 .. sr/2 mirrored to N points (Equalizer.cpp:290-324)
    void channel_magnitude(const mi_equalizer_bank *b, const std::vector<mi::design> &designs, float *mag)
    {
        const size_t N = b->fir_size, half = N >> 1, fs = half + 1;
        std::vector<float> f(fs), c(2 * fs);
        const float k = (0.5f * eff_sample_rate(b)) / float(half);      // dsp::lin_inter_set(…, 0, 0, half, 0.5 sr, 0, fs)
        for (size_t i = 0; i < fs; ++i)
            f[i] = float(i) * k;
        size_t active = 0;
        for (const mi::design &d : designs)
        {
            if (d.mode == mi::FM_BYPASS)                                // Filter::inactive()
                continue;
            mi::freq_chart(d, c.data(), f.data(), fs);
            for (size_t i = 0; i < fs; ++i)
            {
                const float m = sqrtf(c[2 * i] * c[2 * i] + c[2 * i + 1] * c[2 * i + 1]);
                mag[i] = (active == 0) ? m : mag[i] * m;
            }
            ++active;
        }
        if (active > 0)
            for (size_t j = 0; j + 1 < half; ++j)                        // reverse2(&vTemp[fs], &vTemp[1], half - 1)
                mag[fs + j] = mag[half - 1 - j];
        else
            for (size_t i = 0; i < N; ++i)
                mag[i] = 1.0f;
    }

    int reconfigure(mi_equalizer_bank *b, hipStream_t st)               // Equalizer.cpp:243-358
    {
        if (!(b->flags & (EF_REBUILD | EF_CLEAR)))
            return MI_OK;
        void *stream = st;
        const bool clear = (b->flags & EF_CLEAR) != 0;
        if (b->mode == MI_EQM_BYPASS)
        {
            // an object that reconfigures outside the FIR modes drops a cross-fade that was still waiting (:250)
            std::vector<uint8_t> touched(b->channels, 0);
            for (uint32_t ch = 0; ch < b->channels; ++ch)
                touched[ch] = (b->dirty[ch] || clear) ? 1 : 0;
            mi::convolver_cancel_crossfade(b->conv, touched.data());
            b->flags = 0;
            b->latency = 0;
            return MI_OK;
        }
        const size_t N = b->fir_size;
        std::vector<mi_biquad_x1_t> sections;
        std::vector<mi::design> designs;
        const bool need_mag = (b->mode == MI_EQM_FFT || b->mode == MI_EQM_SPM);
        bool any = false;
        std::vector<uint8_t> rebuilt(b->channels, 0);                   // the objects whose reconfigure() does something
        for (uint32_t ch = 0; ch < b->channels; ++ch)
        {
            if (!b->dirty[ch] && !clear)
                continue;
            any = true;
            rebuilt[ch] = 1;
            channel_sections(b, ch, &sections, &designs);
            int r = mi_biquad_bank_set_chains(b->biquads, ch, sections.data(), uint32_t(sections.size()), clear ? 1 : 0);
            if (r != MI_OK)
                return r;
            if (need_mag)
                channel_magnitude(b, designs, &b->h_mag[size_t(ch) * N]);
            b->dirty[ch] = 0;
        }
        int r = mi_biquad_bank_commit(b->biquads, stream);
        if (r != MI_OK)
            return r;
        if (b->mode == MI_EQM_IIR || b->mode == MI_EQM_SPM)
            mi::convolver_cancel_crossfade(b->conv, rebuilt.data());       // Equalizer.cpp:264,356
        if (b->mode == MI_EQM_IIR)
        {
            b->flags = 0;
            b->latency = 0;
            return MI_OK;
        }
        MI_REQUIRE(b->fir_rank > 0, MI_ESTATE, "equalizer: FIR/FFT/SPM modes need fir_rank > 0 at init");
        if (clear)
        {
            if ((r = mi_convolver_bank_reset(b->conv, stream)) != MI_OK) return r;
            if ((r = mi_delay_bank_clear(b->delay, stream)) != MI_OK) return r;
            b->primed = 0;
            if ((r = mi_spectral_bank_set_rank(b->spm, b->fir_rank)) != MI_OK) return r;
            if ((r = mi_spectral_bank_set_phase(b->spm, 0.0f)) != MI_OK) return r;     // forces the STFT buffers to clear
        }
        if (any || !b->streaming_ready)
        {
            if (b->mode == MI_EQM_FIR)
            {
                // (the reference's arithmetic operation for operation: the taps are then the reference's, see biquad.hip)
                if ((r = mi::biquad_bank_reference_impulse_response(b->biquads, b->d_ir, N, N, st)) != MI_OK) return r;
                #define MI_CALL(LN) hipLaunchKernelGGL((eq_ir_to_magnitude_kernel<LN>), dim3(b->channels), dim3(plan<LN>::T), 0, st, \
                                                       b->d_mag, b->d_ir, N, b->d_wnd2n_tail, b->d_tw)
                MI_LOGN_SWITCH(int(b->fir_rank), MI_CALL)
                #undef MI_CALL
                MI_HIP_CHECK(hipGetLastError());
            }
            else
            {
                MI_HIP_CHECK(hipMemcpyAsync(b->d_mag, b->h_mag.data(), b->h_mag.size() * sizeof(float), hipMemcpyHostToDevice, st));
                MI_HIP_CHECK(hipStreamSynchronize(st));
            }
            if (b->mode != MI_EQM_SPM)
            {
                #define MI_CALL(LN) hipLaunchKernelGGL((eq_magnitude_to_fir_kernel<LN>), dim3(b->channels), dim3(plan<LN>::T), 0, st, \
                                                       b->d_taps, b->d_mag, b->d_wndn, b->d_tw)
                MI_LOGN_SWITCH(int(b->fir_rank), MI_CALL)
                #undef MI_CALL
                MI_HIP_CHECK(hipGetLastError());
                // EF_SMOOTH: the new response waits for the next block boundary and is cross-faded in over that block
                // only the objects that reconfigure touch their responses: the others may have a cross-fade waiting
                r = b->smooth ? mi_convolver_bank_crossfade_irs_device(b->conv, b->d_taps, N, uint32_t(N), rebuilt.data(), stream)
                              : mi_convolver_bank_set_irs_device(b->conv, b->d_taps, N, uint32_t(N), rebuilt.data(), stream);
                if (r != MI_OK) return r;
            }
            else
            {
                // mask rows of N/2+1 gains per channel; the spectral bank wants them in host memory
                std::vector<float> mask(size_t(b->channels) * (N / 2 + 1));
                if (b->mode == MI_EQM_SPM)
                    for (uint32_t ch = 0; ch < b->channels; ++ch)
                        std::memcpy(&mask[size_t(ch) * (N / 2 + 1)], &b->h_mag[size_t(ch) * N], (N / 2 + 1) * sizeof(float));
                if ((r = mi_spectral_bank_bind_mask(b->spm, mask.data(), N / 2 + 1, stream)) != MI_OK) return r;
            }
            b->streaming_ready = true;
        }
        b->latency = (b->mode == MI_EQM_SPM) ? uint32_t(N) : uint32_t(N + (N >> 1));    // Equalizer.cpp:347,355
        b->flags = 0;
        return MI_OK;
    }
} // namespace

extern "C" {

int mi_equalizer_bank_create(mi_equalizer_bank_t **bank, uint32_t channels, uint32_t filters, uint32_t fir_rank)
{
    MI_REQUIRE(bank != nullptr, MI_EINVAL, "mi_equalizer_bank_create: NULL result pointer");
    *bank = nullptr;
    MI_REQUIRE(channels > 0 && filters > 0, MI_EINVAL, "mi_equalizer_bank_create: channels and filters must be > 0");
    MI_REQUIRE(fir_rank == 0 || (fir_rank >= 5 && fir_rank <= 13), MI_EINVAL,
               "mi_equalizer_bank_create: fir_rank %u outside the supported 0, 5..13", fir_rank);
    MI_REQUIRE(mi_dspu_device_count() > 0, MI_ENODEV, "no HIP device available (there is no CPU fallback)");
    mi_equalizer_bank *b = new (std::nothrow) mi_equalizer_bank();
    MI_REQUIRE(b != nullptr, MI_ENOMEM, "mi_equalizer_bank_create: out of host memory");
    b->channels = channels;
    b->filters = filters;
    b->fir_rank = fir_rank;
    b->fir_size = fir_rank ? (1u << fir_rank) : 0;
    mi_filter_params_t none = { MI_FLT_NONE, 1, 1000.0f, 1000.0f, 1.0f, 0.0f };        // Filter::init defaults (Filter.cpp:71-78)
    b->params.assign(size_t(channels) * filters, none);
    b->dirty.assign(channels, 1);
    int r = mi_biquad_bank_create(&b->biquads, channels, filters * mi::CHAINS_MAX);   // Equalizer.cpp:79
    if (r == MI_OK && fir_rank > 0)
    {
        const size_t N = b->fir_size;
        int twn = 0;
        r = mi::fft_twiddles(&b->d_tw, &twn);
        std::vector<float> zeros(size_t(channels) * N, 0.0f);
        zeros[0] = 0.0f;
        if (r == MI_OK) r = mi_convolver_bank_create(&b->conv, channels, zeros.data(), N, nullptr, uint32_t(N), fir_rank + 1, 0.0f, nullptr);
        // the line is longer than the delay so that a whole block fits the in-place push/pull piece (delay.hip)
        if (r == MI_OK) r = mi_delay_bank_create(&b->delay, channels, N + ((N > 8192) ? N : 8192));
        if (r == MI_OK) r = mi_delay_bank_set_delay(b->delay, UINT32_MAX, N);
        if (r == MI_OK) r = mi_spectral_bank_create(&b->spm, channels, fir_rank);
        if (r == MI_OK) r = mi_spectral_bank_set_windows(b->spm, -1, MI_WINDOW_SQR_COSINE);
        if (r == MI_OK)
        {
            hipError_t e = hipMalloc(reinterpret_cast<void **>(&b->d_ir), size_t(channels) * N * sizeof(float));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_mag), size_t(channels) * N * sizeof(float));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_taps), size_t(channels) * N * sizeof(float));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_wnd2n_tail), N * sizeof(float));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_wndn), N * sizeof(float));
            if (e == hipSuccess)
            {
                std::vector<float> w(2 * N);
                mi::make_window(w.data(), 2 * N, MI_WINDOW_BLACKMAN_NUTTALL);          // Equalizer.cpp:283-285: second half
                e = hipMemcpy(b->d_wnd2n_tail, w.data() + N, N * sizeof(float), hipMemcpyHostToDevice);
                mi::make_window(w.data(), N, MI_WINDOW_BLACKMAN_NUTTALL);              // Equalizer.cpp:335
                if (e == hipSuccess) e = hipMemcpy(b->d_wndn, w.data(), N * sizeof(float), hipMemcpyHostToDevice);
            }
            if (e != hipSuccess)
                r = mi::fail(e == hipErrorOutOfMemory ? MI_ENOMEM : MI_EHIP, "mi_equalizer_bank_create: %s", hipGetErrorString(e));
            try { b->h_mag.assign(size_t(channels) * N, 1.0f); }
            catch (...) { r = mi::fail(MI_ENOMEM, "mi_equalizer_bank_create: out of host memory"); }
        }
    }
    if (r != MI_OK)
    {
        mi_equalizer_bank_destroy(b);
        return r;
    }
    *bank = b;
    return MI_OK;
}

int mi_equalizer_bank_destroy(mi_equalizer_bank_t *b)
{
    if (b == nullptr)
        return MI_OK;
    mi_biquad_bank_destroy(b->biquads);
    mi_convolver_bank_destroy(b->conv);
    mi_delay_bank_destroy(b->delay);
    mi_spectral_bank_destroy(b->spm);
    (void)hipFree(b->d_ir); (void)hipFree(b->d_mag); (void)hipFree(b->d_taps);
    (void)hipFree(b->d_wnd2n_tail); (void)hipFree(b->d_wndn);
    delete b;
    return MI_OK;
}

int mi_equalizer_bank_set_params(mi_equalizer_bank_t *b, uint32_t channel, uint32_t filter, const mi_filter_params_t *params)
{
    MI_REQUIRE(b != nullptr && params != nullptr, MI_EINVAL, "mi_equalizer_bank_set_params: bad argument");
    MI_REQUIRE(filter < b->filters, MI_EINVAL, "mi_equalizer_bank_set_params: filter %u out of range", filter);   // Equalizer.cpp:212-213
    const uint32_t first = (channel == UINT32_MAX) ? 0 : channel;
    const uint32_t last = (channel == UINT32_MAX) ? b->channels : channel + 1;
    MI_REQUIRE(last <= b->channels, MI_EINVAL, "mi_equalizer_bank_set_params: channel %u out of range", channel);
    for (uint32_t c = first; c < last; ++c)
    {
        // stored as given; Filter::limit (Filter.cpp:161-167) is applied by the designer at rebuild time
        b->params[size_t(c) * b->filters + filter] = *params;
        b->dirty[c] = 1;
    }
    b->flags |= EF_REBUILD;
    return MI_OK;
}

int mi_equalizer_bank_get_params(const mi_equalizer_bank_t *b, uint32_t channel, uint32_t filter, mi_filter_params_t *params)
{
    MI_REQUIRE(b != nullptr && params != nullptr && channel < b->channels && filter < b->filters, MI_EINVAL,
               "mi_equalizer_bank_get_params: bad argument");
    *params = b->params[size_t(channel) * b->filters + filter];
    return MI_OK;
}

int mi_equalizer_bank_set_mode(mi_equalizer_bank_t *b, int mode)        // Equalizer.cpp:360-366
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_equalizer_bank_set_mode: NULL bank");
    MI_REQUIRE(mode >= MI_EQM_BYPASS && mode <= MI_EQM_SPM, MI_EINVAL, "mi_equalizer_bank_set_mode: unknown mode %d", mode);
    if (mode == b->mode)
        return MI_OK;
    b->mode = mode;
    b->flags |= EF_REBUILD | EF_CLEAR;
    // every channel's response has to be rebuilt for the new mode, also when a reset() takes EF_CLEAR away again before
    // the next reconfigure (Equalizer.cpp:575)
    std::fill(b->dirty.begin(), b->dirty.end(), uint8_t(1));
    b->streaming_ready = false;
    return MI_OK;
}

int mi_equalizer_bank_set_sample_rate(mi_equalizer_bank_t *b, uint32_t sample_rate)    // Equalizer.cpp:188-203
{
    MI_REQUIRE(b != nullptr && sample_rate > 0, MI_EINVAL, "mi_equalizer_bank_set_sample_rate: bad argument");
    if (b->sample_rate == sample_rate)
        return MI_OK;
    b->sample_rate = sample_rate;
    std::fill(b->dirty.begin(), b->dirty.end(), uint8_t(1));
    b->flags |= EF_REBUILD | EF_CLEAR;
    return MI_OK;
}

int mi_equalizer_bank_set_actual_sample_rate(mi_equalizer_bank_t *b, uint32_t sample_rate)   // Equalizer.cpp:368-375
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_equalizer_bank_set_actual_sample_rate: NULL bank");
    if (b->actual_sample_rate == sample_rate)
        return MI_OK;
    b->actual_sample_rate = sample_rate;
    if (b->mode == MI_EQM_IIR || b->mode == MI_EQM_SPM)
    {
        std::fill(b->dirty.begin(), b->dirty.end(), uint8_t(1));
        b->flags |= EF_REBUILD;
    }
    return MI_OK;
}

int mi_equalizer_bank_get_latency(mi_equalizer_bank_t *b, uint32_t *latency, void *stream)   // Equalizer.cpp:237-241
{
    MI_REQUIRE(b != nullptr && latency != nullptr, MI_EINVAL, "mi_equalizer_bank_get_latency: bad argument");
    const int r = reconfigure(b, mi::as_stream(stream));
    if (r != MI_OK)
        return r;
    *latency = b->latency;
    return MI_OK;
}

int mi_equalizer_bank_set_smooth(mi_equalizer_bank_t *b, int smooth)    // Equalizer.cpp:618-626
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_equalizer_bank_set_smooth: NULL bank");
    b->smooth = (smooth != 0);
    return MI_OK;
}

int mi_equalizer_bank_reset(mi_equalizer_bank_t *b, void *stream)       // Equalizer.cpp:573-597
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_equalizer_bank_reset: NULL bank");
    b->flags &= ~uint32_t(EF_CLEAR);
    int r = MI_OK;
    switch (b->mode)
    {
        case MI_EQM_IIR:
            r = mi_biquad_bank_reset(b->biquads, UINT32_MAX, stream);
            break;
        case MI_EQM_FIR: case MI_EQM_FFT:
            if ((r = mi_convolver_bank_reset(b->conv, stream)) == MI_OK)
                r = mi_delay_bank_clear(b->delay, stream);
            b->primed = 0;
            break;
        case MI_EQM_SPM:
            // vInBuffer, vOutBuffer AND nBufSize go back to zero (:590-592); SpectralProcessor::reset() keeps its frame
            // position, so the STFT settings are re-applied instead: that clears both buffers and the position
            r = mi_spectral_bank_set_phase(b->spm, 0.0f);
            break;
        default:
            break;
    }
    return r;
}

// what the launches of a call take by value from the host: the positions of the delay line, the convolver and the
// spectral processor behind the current mode, and how far the cleared line has been primed
static uint64_t equalizer_bank_positions(const void *bank)
{
    const mi_equalizer_bank *b = static_cast<const mi_equalizer_bank *>(bank);
    uint64_t h = mi::position_mix(uint64_t(b->mode), b->primed);
    h = mi::position_mix(h, b->flags);
    if (b->delay != nullptr) h = mi::position_mix(h, mi::delay_bank_positions(b->delay));
    if (b->conv != nullptr)  h = mi::position_mix(h, mi::convolver_bank_positions(b->conv));
    if (b->spm != nullptr)   h = mi::position_mix(h, mi::spectral_bank_positions(b->spm));
    return h;
}

int mi_equalizer_bank_process(mi_equalizer_bank_t *b, float *out, const float *in, size_t samples,
                              size_t out_stride, size_t in_stride, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_equalizer_bank_process: NULL bank");
    if (samples == 0)
        return MI_OK;
    MI_REQUIRE(out != nullptr && in != nullptr, MI_EINVAL, "mi_equalizer_bank_process: NULL buffer");
    hipStream_t st = mi::as_stream(stream);
    {
        const int rc = mi::capture_touch(st, b, "equalizer", equalizer_bank_positions);
        if (rc != MI_OK)
            return rc;
    }
    int r = reconfigure(b, st);
    if (r != MI_OK)
        return r;
    switch (b->mode)
    {
        case MI_EQM_IIR:
            return mi_biquad_bank_process(b->biquads, out, in, samples, out_stride, in_stride, stream);
        case MI_EQM_FIR: case MI_EQM_FFT:
            // The reference buffers a block of N samples and then convolves it as a whole with the response in force at
            // that moment (Equalizer.cpp:477-511): a delay of N in FRONT of the zero-latency convolver -- same samples
            // out, and a retune (or its cross-fade) lands on the same block as in the reference.
            // While the first block after a clear is still being buffered the reference convolves nothing (its first
            // convolution happens when that block is complete): the convolver's frames start with the first real block.
            {
                size_t done = 0;
                if (b->primed < b->fir_size)
                {
                    done = (samples < size_t(b->fir_size - b->primed)) ? samples : size_t(b->fir_size - b->primed);
                    if ((r = mi_delay_bank_process(b->delay, out, in, done, out_stride, in_stride, 0, MI_GAIN_NONE, 0.0f, nullptr, 0, stream)) != MI_OK)
                        return r;                               // emits the zeros the cleared line holds
                    b->primed += uint32_t(done);
                    if (done == samples)
                        return MI_OK;
                }
                const size_t rest = samples - done;
                mi::delay_view dl;
                // several whole blocks at a plain frame boundary: ONE launch walks them (conv_frames_kernel: the response's
                // image and the overlap-add tail stay in registers, the delay line is touched once at either end)
                if (rest >= 2 * size_t(b->fir_size) && rest % b->fir_size == 0 &&
                    mi::convolver_takes_delayed_frames(b->conv, b->fir_size) && mi::delay_bank_view(b->delay, &dl) == MI_OK &&
                    dl.delay == b->fir_size && (dl.size % 2 == 0) && (dl.head % 2 == 0) && dl.size >= 2 * b->fir_size)
                {
                    const size_t N = b->fir_size, blocks = rest / N;
                    // column slices of the caller's two buffers: distinct outputs, and no output overlaps an input unless the
                    // buffers themselves do (in place: the workgroup's kernel, block after block)
                    const size_t fo = (b->channels - 1) * out_stride + samples, fi = (b->channels - 1) * in_stride + samples;
                    const bool apart = (out + fo <= in) || (in + fi <= out);
                    float *po[mi::CONV_FRAMES_MAX];
                    const float *pi[mi::CONV_FRAMES_MAX];
                    for (size_t k0 = 0, cnt = 0; k0 < blocks; k0 += cnt)
                    {
                        cnt = mi::conv_frames_chunk(blocks - k0);
                        for (size_t k = 0; k < cnt; ++k)
                        {
                            po[k] = out + done + (k0 + k) * N;
                            pi[k] = in + done + (k0 + k) * N;
                        }
                        if (mi::delay_bank_view(b->delay, &dl) != MI_OK)
                            return MI_ESTATE;
                        if ((r = mi::convolver_process_delayed_frames(b->conv, po, pi, cnt, out_stride, in_stride, dl, st, apart)) != MI_OK)
                            return r;
                        mi::delay_bank_advance(b->delay, cnt * N);
                    }
                    return MI_OK;
                }
                // a whole block at a plain frame boundary: the frame kernel reads its frame out of the delay line and
                // pushes the new samples into it itself (one launch)
                if (mi::convolver_takes_delayed_frame(b->conv, rest) && mi::delay_bank_view(b->delay, &dl) == MI_OK &&
                    dl.delay == b->fir_size && (dl.size % 2 == 0) && (dl.head % 2 == 0) && (dl.delay % 2 == 0) &&
                    size_t(dl.size - dl.delay) >= rest)
                {
                    r = mi::convolver_process_delayed_frame(b->conv, out + done, in + done, out_stride, in_stride, dl, st);
                    if (r == MI_OK)
                        mi::delay_bank_advance(b->delay, rest);
                    return r;
                }
                if ((r = mi_delay_bank_process(b->delay, out + done, in + done, rest, out_stride, in_stride, 0, MI_GAIN_NONE, 0.0f, nullptr, 0, stream)) != MI_OK)
                    return r;
                return mi_convolver_bank_process(b->conv, out + done, out + done, rest, out_stride, out_stride, stream);
            }
        case MI_EQM_SPM:
            return mi_spectral_bank_process(b->spm, out, in, samples, out_stride, in_stride, stream);
        default:                                                        // EQM_BYPASS (Equalizer.cpp:564-569)
            if (out != in)
                MI_HIP_CHECK(hipMemcpy2DAsync(out, out_stride * sizeof(float), in, in_stride * sizeof(float),
                                              samples * sizeof(float), b->channels, hipMemcpyDeviceToDevice, st));
            return MI_OK;
    }
}

// `blocks` consecutive process() calls in one C call; runs of whole FIR blocks go as ONE launch (conv_frames_kernel)
int mi_equalizer_bank_process_blocks(mi_equalizer_bank_t *b, float *const *out, const float *const *in, size_t blocks, size_t samples,
                                     size_t out_stride, size_t in_stride, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_equalizer_bank_process_blocks: NULL bank");
    if (samples == 0 || blocks == 0)
        return MI_OK;
    MI_REQUIRE(out != nullptr && in != nullptr, MI_EINVAL, "mi_equalizer_bank_process_blocks: NULL pointer table");
    for (size_t k = 0; k < blocks; ++k)
        MI_REQUIRE(out[k] != nullptr && in[k] != nullptr, MI_EINVAL, "mi_equalizer_bank_process_blocks: NULL buffer of block %zu", k);
    hipStream_t st = mi::as_stream(stream);
    const size_t ob = (size_t(b->channels - 1) * out_stride + samples) * sizeof(float), ib = (size_t(b->channels - 1) * in_stride + samples) * sizeof(float);
    auto overlap = [](const void *p, size_t pn, const void *q, size_t qn) -> bool {
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(p), b0 = reinterpret_cast<uintptr_t>(q);
        return a0 < b0 + qn && b0 < a0 + pn;
    };
    if (b->mode == MI_EQM_IIR || b->mode == MI_EQM_SPM)     // the banks with runs of blocks of their own (SPM was shut out)
    {
        const int rc = mi::capture_touch(st, b, "equalizer", equalizer_bank_positions);
        if (rc != MI_OK)
            return rc;
        const int r = reconfigure(b, st);
        if (r != MI_OK)
            return r;
        if (b->mode == MI_EQM_IIR)                          // (reconfigure settles a pending change of mode)
            return mi_biquad_bank_process_blocks(b->biquads, out, in, blocks, samples, out_stride, in_stride, stream);
        if (b->mode == MI_EQM_SPM)                          // the spectral bank's own runs of blocks (spectral.hip)
            return mi_spectral_bank_process_blocks(b->spm, out, in, blocks, samples, out_stride, in_stride, stream);
    }
    size_t k = 0;
    while (k < blocks)
    {
        // a run of blocks for one launch: FIR / FFT mode in its steady state, blocks of exactly the FIR size, and no block that
        // partly overlaps another block's buffers
        size_t cnt = 0;
        if (b->mode == MI_EQM_FIR || b->mode == MI_EQM_FFT)
        {
            const int rc = mi::capture_touch(st, b, "equalizer", equalizer_bank_positions);
            if (rc != MI_OK)
                return rc;
            const int r = reconfigure(b, st);
            if (r != MI_OK)
                return r;
            mi::delay_view dl;
            if (samples == size_t(b->fir_size) && b->primed >= b->fir_size &&
                mi::convolver_takes_delayed_frames(b->conv, samples) && mi::delay_bank_view(b->delay, &dl) == MI_OK &&
                dl.delay == b->fir_size && (dl.size % 2 == 0) && (dl.head % 2 == 0) && dl.size >= 2 * b->fir_size)
            {
                const size_t most = mi::conv_frames_chunk(blocks - k);
                while (k + cnt < blocks && cnt < most)
                {
                    bool ok = true;
                    // (the same buffer again is fine -- a ring of buffers, a block in place: a thread of the kernel touches
                    // the same sample positions of every block, in the blocks' order; buffers that overlap otherwise are not)
                    const float *oj = out[k + cnt], *ij = in[k + cnt];
                    for (size_t i = k; ok && i < k + cnt; ++i)
                        ok = (out[i] == ij || !overlap(out[i], ob, ij, ib)) && (in[i] == oj || !overlap(in[i], ib, oj, ob)) &&
                             (out[i] == oj || !overlap(out[i], ob, oj, ob));
                    ok = ok && (oj == ij || !overlap(oj, ob, ij, ib));
                    if (!ok)
                        break;
                    ++cnt;
                }
                if (cnt >= 2)
                {
                    const int rr = mi::convolver_process_delayed_frames(b->conv, out + k, in + k, cnt, out_stride, in_stride, dl, st);
                    if (rr != MI_OK)
                        return rr;
                    mi::delay_bank_advance(b->delay, cnt * samples);
                    k += cnt;
                    continue;
                }
            }
        }
        const int r = mi_equalizer_bank_process(b, out[k], in[k], samples, out_stride, in_stride, stream);
        if (r != MI_OK)
            return r;
        ++k;
    }
    return MI_OK;
}

int mi_equalizer_bank_info(const mi_equalizer_bank_t *b, uint32_t *filters, uint32_t *fir_rank, int *mode, uint32_t *ir_size)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_equalizer_bank_info: NULL bank");
    if (filters)  *filters = b->filters;
    if (fir_rank) *fir_rank = b->fir_rank;
    if (mode)     *mode = b->mode;
    if (ir_size)  *ir_size = (b->mode >= MI_EQM_FIR) ? (b->fir_size << 1) : 0;         // Equalizer.cpp:599-616
    return MI_OK;
}

} // extern "C"
