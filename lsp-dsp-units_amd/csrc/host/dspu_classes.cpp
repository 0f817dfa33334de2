// lsp::dspu::* compatibility classes (one object == one channel, HOST sample pointers) implemented on the
// C-ABI banks with channels = 1.  Every process() stages the caller's block through device memory:
// correct and convenient, but it measures PCIe and launch latency -- anything performance critical should
// hold many channels in one bank and keep the samples in HBM (mi_*_bank_* in include/mi_dspu.h).
// Error behaviour follows the reference: init() returns false on failure, process() is void and, on an
// unusable object, leaves the output as the reference would (zeros or a copy).
#include <lsp-plug.in/dsp-units/filters/Filter.h>
#include <lsp-plug.in/dsp-units/filters/FilterBank.h>
#include <lsp-plug.in/dsp-units/filters/FilterArray.h>
#include <lsp-plug.in/dsp-units/filters/EqualizerArray.h>
#include <lsp-plug.in/dsp-units/util/ConvolverArray.h>
#include <lsp-plug.in/dsp-units/filters/Equalizer.h>
#include <lsp-plug.in/dsp-units/filters/DynamicFilters.h>
#include <lsp-plug.in/dsp-units/util/Convolver.h>
#include <lsp-plug.in/dsp-units/util/SpectralProcessor.h>
#include <lsp-plug.in/dsp-units/util/MultiSpectralProcessor.h>
#include <lsp-plug.in/dsp-units/util/Crossover.h>
#include <lsp-plug.in/dsp-units/meters/ILUFSMeter.h>
#include <lsp-plug.in/dsp-units/misc/envelope.h>
#include <lsp-plug.in/dsp-units/misc/fft_crossover.h>
#include <lsp-plug.in/dsp-units/util/FFTCrossover.h>
#include <lsp-plug.in/dsp-units/util/SpectralSplitter.h>
#include <lsp-plug.in/dsp-units/meters/LoudnessMeter.h>
#include <lsp-plug.in/dsp-units/util/Delay.h>
#include <lsp-plug.in/dsp-units/util/RingBuffer.h>
#include <lsp-plug.in/dsp-units/util/Analyzer.h>
#include <lsp-plug.in/dsp-units/misc/windows.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

#include "filter_design.h"

// lsp::dspu::filter_params_t is the reference's own type (so that the mangled names match); the C-ABI's record has the same
// fields in the same order (static_assert in filters/common.h)
static inline mi_filter_params_t *cfp(lsp::dspu::filter_params_t *p)             { return reinterpret_cast<mi_filter_params_t *>(p); }
static inline const mi_filter_params_t *cfp(const lsp::dspu::filter_params_t *p) { return reinterpret_cast<const mi_filter_params_t *>(p); }

namespace lsp
{
namespace dspu
{
// The class API has no error channel (process() is void, as in the reference), so the last status of a C-ABI call made
// on behalf of an object of this thread is kept and can be asked for: lsp::dspu::last_status() (mi_dspu_last_error()
// holds the text).  A failed call leaves the output as documented per class (a copy, or zeros) -- never silently "ok".
namespace
{
    thread_local int tl_last_status = MI_OK;
    inline int last_status(int r)
    {
        if (r != MI_OK)
            tl_last_status = r;
        return r;
    }
}
int last_status()           { return tl_last_status; }
void clear_last_status()    { tl_last_status = MI_OK; }

namespace
{
    // a pair of device rows that grows on demand
    struct staging
    {
        float  *d_in = nullptr, *d_out = nullptr;
        size_t  cap = 0;

        bool reserve(size_t n)
        {
            if (n <= cap)
                return true;
            release();
            if (mi_dspu_malloc(reinterpret_cast<void **>(&d_in), n * sizeof(float)) != MI_OK) return false;
            if (mi_dspu_malloc(reinterpret_cast<void **>(&d_out), n * sizeof(float)) != MI_OK) return false;
            cap = n;
            return true;
        }
        void release()
        {
            mi_dspu_free(d_in); mi_dspu_free(d_out);
            d_in = d_out = nullptr;
            cap = 0;
        }
        bool up(const float *src, size_t n)     { return mi_dspu_copy_h2d(d_in, src, n * sizeof(float), nullptr) == MI_OK; }
        bool down(float *dst, size_t n)
        {
            return mi_dspu_copy_d2h(dst, d_out, n * sizeof(float), nullptr) == MI_OK &&
                   mi_dspu_stream_synchronize(nullptr) == MI_OK;
        }
    };
} // namespace

// ---- windows ----------------------------------------------------------------------------------------------------
namespace windows
{
    void window(float *dst, size_t n, window_t type)
    {
        mi_window(dst, n, int(type));
    }

    #define MI_WND(fn, id) void fn(float *dst, size_t n) { mi_window(dst, n, int(id)); }
    MI_WND(hann, HANN) MI_WND(hamming, HAMMING) MI_WND(blackman, BLACKMAN) MI_WND(lanczos, LANCZOS)
    MI_WND(gaussian, GAUSSIAN) MI_WND(poisson, POISSON) MI_WND(parzen, PARZEN) MI_WND(tukey, TUKEY)
    MI_WND(welch, WELCH) MI_WND(nuttall, NUTTALL) MI_WND(blackman_nuttall, BLACKMAN_NUTTALL)
    MI_WND(blackman_harris, BLACKMAN_HARRIS) MI_WND(hann_poisson, HANN_POISSON)
    MI_WND(bartlett_hann, BARTLETT_HANN) MI_WND(bartlett_fejer, BARTLETT_FEJER) MI_WND(triangular, TRIANGULAR)
    MI_WND(rectangular, RECTANGULAR) MI_WND(flat_top, FLAT_TOP) MI_WND(cosine, COSINE)
    MI_WND(sqr_cosine, SQR_COSINE) MI_WND(cubic, CUBIC)
    #undef MI_WND

    void triangular_general(float *dst, size_t n, int dn)
    {
        const float q[] = { float(dn) };
        mi_window_general(dst, n, MI_WINDOW_TRIANGULAR, q, 1);
    }
    void hamming_general(float *dst, size_t n, float a, float b)
    {
        const float q[] = { a, b };
        mi_window_general(dst, n, MI_WINDOW_HAMMING, q, 2);
    }
    void blackman_general(float *dst, size_t n, float a)
    {
        mi_window_general(dst, n, MI_WINDOW_BLACKMAN, &a, 1);
    }
    void nuttall_general(float *dst, size_t n, float a0, float a1, float a2, float a3)
    {
        const float q[] = { a0, a1, a2, a3 };
        mi_window_general(dst, n, MI_WINDOW_NUTTALL, q, 4);
    }
    void nutall_general(float *dst, size_t n, float a0, float a1, float a2, float a3) { nuttall_general(dst, n, a0, a1, a2, a3); }
    void flat_top_general(float *dst, size_t n, float a0, float a1, float a2, float a3, float a4)
    {
        const float q[] = { a0, a1, a2, a3, a4 };
        mi_window_general(dst, n, MI_WINDOW_FLAT_TOP, q, 5);
    }
    void gaussian_general(float *dst, size_t n, float s)        { mi_window_general(dst, n, MI_WINDOW_GAUSSIAN, &s, 1); }
    void poisson_general(float *dst, size_t n, float t)         { mi_window_general(dst, n, MI_WINDOW_POISSON, &t, 1); }
    void bartlett_hann_general(float *dst, size_t n, float a0, float a1, float a2)
    {
        const float q[] = { a0, a1, a2 };
        mi_window_general(dst, n, MI_WINDOW_BARTLETT_HANN, q, 3);
    }
    void hann_poisson_general(float *dst, size_t n, float a)    { mi_window_general(dst, n, MI_WINDOW_HANN_POISSON, &a, 1); }
    void tukey_general(float *dst, size_t n, float a)           { mi_window_general(dst, n, MI_WINDOW_TUKEY, &a, 1); }
}

// ---- FilterBank -------------------------------------------------------------------------------------------------
// Reference members (filters/FilterBank.h:39-46): nItems / nMaxItems / nLastItems / vChains are live, vFilters carries
// the device bank's handle, vBackup the staging rows, vData the host allocation.
namespace
{
    struct bank_ext                                     // lives at the start of FilterBank::vData
    {
        staging st;
    };
    inline mi_biquad_bank_t *dev_bank(dsp::biquad_t *p) { return reinterpret_cast<mi_biquad_bank_t *>(p); }
}
static_assert(sizeof(FilterBank) == 56, "FilterBank keeps the reference's layout (SURVEY.md 0.4)");

FilterBank::FilterBank() { construct(); }
FilterBank::~FilterBank() { destroy(); }

void FilterBank::construct()
{
    vFilters    = nullptr;
    vChains     = nullptr;
    nItems      = 0;
    nMaxItems   = 0;
    nLastItems  = size_t(-1);                           // FilterBank.cpp:45: the first end() clears
    vBackup     = nullptr;
    vData       = nullptr;
}

bool FilterBank::init(size_t filters)
{
    destroy();
    const size_t cap = (filters > 0) ? filters : 1;
    const size_t head = (sizeof(bank_ext) + 63) & ~size_t(63);
    uint8_t *raw = static_cast<uint8_t *>(std::calloc(1, head + cap * sizeof(dsp::biquad_x1_t)));
    if (raw == nullptr)
        return false;
    mi_biquad_bank_t *bank = nullptr;
    if (mi_biquad_bank_create(&bank, 1, uint32_t(filters)) != MI_OK)
    {
        std::free(raw);
        return false;
    }
    new (raw) bank_ext();
    vData       = raw;
    vChains     = reinterpret_cast<dsp::biquad_x1_t *>(raw + head);
    vFilters    = reinterpret_cast<dsp::biquad_t *>(bank);
    nItems      = 0;
    nMaxItems   = cap;
    nLastItems  = size_t(-1);
    return true;
}

void FilterBank::destroy()
{
    if (vData != nullptr)
    {
        bank_ext *x = reinterpret_cast<bank_ext *>(vData);
        x->st.release();
        x->~bank_ext();
        std::free(vData);
    }
    if (vFilters != nullptr)
        mi_biquad_bank_destroy(dev_bank(vFilters));
    construct();
}

dsp::biquad_x1_t *FilterBank::add_chain()
{
    if (vChains == nullptr)
        return nullptr;
    if (nItems >= nMaxItems)                            // FilterBank.cpp:94-99: the last slot again
        return (nItems == 0) ? nullptr : &vChains[nItems - 1];
    return &vChains[nItems++];
}

dsp::biquad_x1_t *FilterBank::chain(size_t id)
{
    return (vChains != nullptr && id < nItems) ? &vChains[id] : nullptr;
}

void FilterBank::end(bool clear)
{
    if (vFilters == nullptr)
        return;
    static_assert(sizeof(dsp::biquad_x1_t) == sizeof(mi_biquad_x1_t), "section layout");
    // FilterBank.cpp:233-235: the delays are cleared on request or when the section count changed since begin()
    const bool wipe = clear || (nItems != nLastItems);
    last_status(mi_biquad_bank_set_chains(dev_bank(vFilters), 0, reinterpret_cast<const mi_biquad_x1_t *>(vChains),
                                          uint32_t(nItems), wipe ? 1 : 0));
    nLastItems = nItems;
}

void FilterBank::process(float *out, const float *in, size_t samples)
{
    if (samples == 0)
        return;
    bank_ext *x = reinterpret_cast<bank_ext *>(vData);
    if (x == nullptr || !x->st.reserve(samples) || !x->st.up(in, samples) ||
        last_status(mi_biquad_bank_process(dev_bank(vFilters), x->st.d_out, x->st.d_in, samples, samples, samples, nullptr)) != MI_OK ||
        !x->st.down(out, samples))
    {
        if (out != in)
            std::memmove(out, in, samples * sizeof(float));
    }
}

void FilterBank::impulse_response(float *out, size_t samples)
{
    if (samples == 0)
        return;
    bank_ext *x = reinterpret_cast<bank_ext *>(vData);
    if (x == nullptr || !x->st.reserve(samples) ||
        last_status(mi_biquad_bank_impulse_response(dev_bank(vFilters), x->st.d_out, samples, samples, nullptr)) != MI_OK ||
        !x->st.down(out, samples))
        std::memset(out, 0, samples * sizeof(float));
}

void FilterBank::reset()                { if (vFilters) last_status(mi_biquad_bank_reset(dev_bank(vFilters), 0, nullptr)); }

void FilterBank::dump(IStateDumper *v) const
{
    // FilterBank.cpp:332-424: the sections as end() packs them -- groups of 8, then 4, 2, 1 chains, every coefficient of a group
    // side by side -- then the chains themselves.  The packed copy lives on the device here; its groups are gathered from
    // vChains, which is what end() sent (the delay memory is not part of the reference's dump either).
    const size_t groups = (nItems >> 3) + ((nItems >> 2) & 1) + ((nItems >> 1) & 1) + (nItems & 1);
    const dsp::biquad_x1_t *c = vChains;
    auto gather = [&](size_t lanes, bool with_p)
    {
        float col[5][8];
        for (size_t i = 0; i < lanes; ++i)
        {
            col[0][i] = c[i].b0; col[1][i] = c[i].b1; col[2][i] = c[i].b2; col[3][i] = c[i].a1; col[4][i] = c[i].a2;
        }
        static const char *const names[5] = { "b0", "b1", "b2", "a1", "a2" };
        v->begin_object(c, sizeof(dsp::biquad_t));
        for (int k = 0; k < 5; ++k)
            v->writev(names[k], col[k], lanes);
        if (with_p)
        {
            const float p[2] = { 0.0f, 0.0f };
            v->writev("p", p, 2);
        }
        v->end_object();
        c += lanes;
    };
    auto single = [&](const dsp::biquad_x1_t *q)
    {
        v->begin_object(q, sizeof(dsp::biquad_x1_t));
        v->write("b0", q->b0); v->write("b1", q->b1); v->write("b2", q->b2);
        v->write("a1", q->a1); v->write("a2", q->a2);
        v->write("p0", q->p0); v->write("p1", q->p1); v->write("p2", q->p2);
        v->end_object();
    };
    v->begin_array("vFilters", vFilters, groups);
    if (c != nullptr)
    {
        for (size_t g = 0; g < (nItems >> 3); ++g)
            gather(8, false);
        if (nItems & 4)
            gather(4, false);
        if (nItems & 2)
            gather(2, true);
        if (nItems & 1)
            single(c);
    }
    v->end_array();
    v->begin_array("vChains", vChains, nItems);
    for (size_t i = 0; vChains != nullptr && i < nItems; ++i)
        single(&vChains[i]);
    v->end_array();
    v->write("nItems", nItems);
    v->write("nMaxItems", nMaxItems);
    v->write("nLastItems", nLastItems);
    v->write("vBackup", vBackup);
    v->write("vData", vData);
}

// ---- Filter ------------------------------------------------------------------------------------------------------
// Reference members (filters/Filter.h:57-65), all live.  vData holds the designer's result (cascades + sections);
// vItems / nItems view its cascades the way the reference's own array does.
namespace
{
    inline mi::design *design_of(uint8_t *p) { return reinterpret_cast<mi::design *>(p); }
}
static_assert(sizeof(Filter) == 88, "Filter keeps the reference's layout (SURVEY.md 0.4)");
static_assert(sizeof(dsp::f_cascade_t) == sizeof(mi::cascade), "cascade layout");

Filter::Filter() { construct(); }
Filter::~Filter() { destroy(); }

void Filter::construct()                                    // Filter.cpp:43-58
{
    pBank           = nullptr;
    sParams.nType   = FLT_NONE;
    sParams.nSlope  = 1;
    sParams.fFreq   = 0.0f;
    sParams.fFreq2  = 0.0f;
    sParams.fGain   = 0.0f;
    sParams.fQuality= 0.0f;
    nSampleRate     = 0;
    nMode           = FM_BYPASS;
    nItems          = 0;
    vItems          = nullptr;
    vData           = nullptr;
    nFlags          = 0;
    nLatency        = 0;
}

bool Filter::init(FilterBank *fb)
{
    destroy();
    mi::design *d = new (std::nothrow) mi::design();
    if (d == nullptr)
        return false;
    vData = reinterpret_cast<uint8_t *>(d);
    if (fb != nullptr)
        pBank = fb;
    else
    {
        pBank = new (std::nothrow) FilterBank();
        if (pBank == nullptr || !pBank->init(FILTER_CHAINS_MAX))
        {
            delete pBank;
            pBank = nullptr;
            delete d;
            vData = nullptr;
            return false;
        }
        nFlags |= FF_OWN_BANK;
    }
    filter_params_t fp = { FLT_NONE, 1, 1000.0f, 1000.0f, 1.0f, 0.0f };
    sParams = fp;
    update(48000, &fp);
    nFlags |= FF_REBUILD | FF_CLEAR;                        // Filter.cpp:111-112
    return true;
}

void Filter::destroy()
{
    if ((nFlags & FF_OWN_BANK) && pBank != nullptr)
    {
        pBank->destroy();
        delete pBank;
    }
    delete design_of(vData);
    construct();
}

void Filter::update(size_t sr, const filter_params_t *params)   // Filter.cpp:141-159
{
    const uint32_t type = sParams.nType, slope = sParams.nSlope;
    nSampleRate = sr;
    nMode       = FM_BYPASS;                                // Filter.cpp:150: inactive until the next rebuild()
    nLatency    = 0;
    sParams     = *params;
    mi::limit_params(cfp(&sParams), uint32_t(sr));
    nFlags     |= FF_REBUILD;
    if (type != sParams.nType || slope != sParams.nSlope)
        nFlags |= FF_CLEAR;
}

void Filter::limit(size_t, filter_params_t *fp)
{
    mi::limit_params(cfp(fp), uint32_t(nSampleRate));           // the reference ignores its sr argument too (Filter.cpp:161-163)
}

void Filter::set_sample_rate(size_t sr)     { filter_params_t p = sParams; update(sr, &p); }
void Filter::get_params(filter_params_t *params) { if (params) *params = sParams; }

void Filter::rebuild()
{
    mi::design *d = design_of(vData);
    if (d == nullptr || pBank == nullptr)
        return;
    if (nFlags & FF_OWN_BANK)
        pBank->begin();
    d->cascades.reserve(mi::CHAINS_MAX + 1);
    mi::design_filter(d, cfp(&sParams), uint32_t(nSampleRate));
    nMode   = filter_mode_t(d->mode);
    nItems  = d->cascades.size();
    vItems  = reinterpret_cast<dsp::f_cascade_t *>(d->cascades.data());
    for (const mi_biquad_x1_t &s : d->sections)
    {
        dsp::biquad_x1_t *c = pBank->add_chain();
        if (c == nullptr)
            break;
        std::memcpy(c, &s, sizeof(s));
    }
    if (nFlags & FF_OWN_BANK)
        pBank->end((nFlags & FF_CLEAR) != 0);
    nFlags &= ~size_t(FF_REBUILD | FF_CLEAR);
}

void Filter::process(float *out, const float *in, size_t samples)
{
    if (vData == nullptr)
    {
        if (out != in)
            std::memmove(out, in, samples * sizeof(float));
        return;
    }
    if (nFlags & (FF_REBUILD | FF_CLEAR))
        rebuild();
    if (nMode == FM_BYPASS)
    {
        if (out != in)
            std::memmove(out, in, samples * sizeof(float));
        return;
    }
    pBank->process(out, in, samples);
}

bool Filter::impulse_response(float *out, size_t length)
{
    if (vData == nullptr || !(nFlags & FF_OWN_BANK))
        return false;
    if (nFlags & (FF_REBUILD | FF_CLEAR))
        rebuild();
    pBank->impulse_response(out, length);
    return true;
}

void Filter::freq_chart(float *c, const float *f, size_t count)
{
    mi::design *d = design_of(vData);
    if (d == nullptr)
        return;
    if (nFlags & FF_REBUILD)
    {
        d->cascades.reserve(mi::CHAINS_MAX + 1);
        mi::design_filter(d, cfp(&sParams), uint32_t(nSampleRate));
    }
    mi::freq_chart(*d, c, f, count);
}

void Filter::freq_chart(float *re, float *im, const float *f, size_t count)
{
    std::vector<float> c(2 * count);
    freq_chart(c.data(), f, count);
    for (size_t i = 0; i < count; ++i)
    {
        re[i] = c[2 * i];
        im[i] = c[2 * i + 1];
    }
}

void Filter::dump(IStateDumper *v) const
{
    // Filter.cpp:2430-2466
    if (nFlags & FF_OWN_BANK)
        v->write_object("pBank", pBank);
    else
        v->write("pBank", pBank);
    v->begin_object("sParams", &sParams, sizeof(filter_params_t));
    v->write("nType", sParams.nType);
    v->write("fFreq", sParams.fFreq);
    v->write("fFreq2", sParams.fFreq2);
    v->write("fGain", sParams.fGain);
    v->write("nSlope", sParams.nSlope);
    v->write("fQuality", sParams.fQuality);
    v->end_object();
    v->write("nSampleRate", nSampleRate);
    v->write("nMode", int(nMode));
    v->write("nItems", nItems);
    v->begin_array("vItems", vItems, nItems);
    for (size_t i = 0; vItems != nullptr && i < nItems; ++i)
    {
        v->begin_object(&vItems[i], sizeof(dsp::f_cascade_t));
        v->writev("t", vItems[i].t, 4);
        v->writev("b", vItems[i].b, 4);
        v->end_object();
    }
    v->end_array();
    v->write("vData", vData);
    v->write("nFlags", nFlags);
    v->write("nLatency", nLatency);
}

// ---- Equalizer ---------------------------------------------------------------------------------------------------
// Reference members (filters/Equalizer.h:59-78).  The Filter objects of vFilters hold the parameters and the mode of
// each filter; the sections themselves run in the device bank behind pData.
struct Equalizer::impl_t
{
    mi_equalizer_bank_t *bank = nullptr;
    staging st;
};
static_assert(sizeof(Equalizer) == 160, "Equalizer keeps the reference's layout (SURVEY.md 0.4)");

Equalizer::Equalizer() { construct(); }
Equalizer::~Equalizer() { destroy(); }

void Equalizer::construct()                                 // Equalizer.cpp:43-65
{
    sBank.construct();
    vFilters = nullptr;
    nFilters = nSampleRate = nActualSampleRate = nFirSize = nFirRank = nLatency = nBufSize = 0;
    nMode = EQM_BYPASS;
    vInBuffer = vOutBuffer = vNewConv = vConv = vFft = vTemp = nullptr;
    pData = nullptr;
    nFlags = EF_REBUILD | EF_CLEAR;
}

// The device bank has reconfigured: every filter is rebuilt, i.e. its mode is that of its current design again
// (Equalizer.cpp:256-259: sBank.begin(); vFilters[i].rebuild(); sBank.end()).  The embedded bank has no storage, so
// rebuild() designs, records nMode / nItems and hands out no section.
void Equalizer::rebuilt()
{
    for (uint32_t i = 0; i < nFilters; ++i)
        vFilters[i].rebuild();
    nFlags &= ~size_t(EF_REBUILD | EF_CLEAR);
}

bool Equalizer::init(size_t filters, size_t fir_rank)
{
    if (impl() != nullptr && nFilters == filters && nFirRank == fir_rank)
    {
        reset();
        return true;
    }
    destroy();
    impl_t *p = new (std::nothrow) impl_t();
    if (p == nullptr)
        return false;
    if (last_status(mi_equalizer_bank_create(&p->bank, 1, uint32_t(filters), uint32_t(fir_rank))) != MI_OK)
    {
        delete p;
        return false;
    }
    pData = reinterpret_cast<uint8_t *>(p);
    vFilters = new (std::nothrow) Filter[filters];
    if (vFilters == nullptr)
    {
        destroy();
        return false;
    }
    nFilters = uint32_t(filters);
    nFirRank = uint32_t(fir_rank);
    nFirSize = (fir_rank > 0) ? (1u << fir_rank) : 0;
    for (size_t i = 0; i < filters; ++i)
        if (!vFilters[i].init(&sBank))
        {
            destroy();
            return false;
        }
    nFlags |= EF_REBUILD | EF_CLEAR;
    nLatency = 0;
    nBufSize = 0;
    return true;
}

void Equalizer::destroy()
{
    if (vFilters != nullptr)
    {
        for (uint32_t i = 0; i < nFilters; ++i)
            vFilters[i].destroy();
        delete [] vFilters;
    }
    if (impl_t *p = impl())
    {
        mi_equalizer_bank_destroy(p->bank);
        p->st.release();
        delete p;
    }
    sBank.destroy();
    const size_t smooth = nFlags & EF_SMOOTH;
    construct();
    nFlags |= smooth;
}

bool Equalizer::configuration_changed() const { return (nFlags & EF_REBUILD) != 0; }

bool Equalizer::set_params(size_t id, const filter_params_t *params)
{
    if (impl() == nullptr || id >= nFilters)
        return false;
    vFilters[id].update(nSampleRate, params);               // mode reads FM_BYPASS until the next reconfigure
    nFlags |= EF_REBUILD;
    return last_status(mi_equalizer_bank_set_params(impl()->bank, 0, uint32_t(id), cfp(params))) == MI_OK;
}

bool Equalizer::limit_params(size_t id, filter_params_t *fp)
{
    if (id >= nFilters)
        return false;
    vFilters[id].limit(nSampleRate, fp);
    return true;
}

bool Equalizer::get_params(size_t id, filter_params_t *params)
{
    if (id >= nFilters)
        return false;
    vFilters[id].get_params(params);
    return true;
}

void Equalizer::set_mode(equalizer_mode_t mode)
{
    if (impl() == nullptr || nMode == mode)                 // Equalizer.cpp:212-218
        return;
    nMode = mode;
    nFlags |= EF_REBUILD | EF_CLEAR;
    mi_equalizer_bank_set_mode(impl()->bank, int(mode));
}

void Equalizer::set_actual_sample_rate(size_t sr)
{
    if (nActualSampleRate == sr)                            // Equalizer.cpp:368-375
        return;
    nActualSampleRate = uint32_t(sr);
    if (nMode == EQM_IIR || nMode == EQM_SPM)
        nFlags |= EF_REBUILD;
    if (impl() != nullptr)
        mi_equalizer_bank_set_actual_sample_rate(impl()->bank, uint32_t(sr));
}

void Equalizer::set_sample_rate(size_t sr)
{
    if (nSampleRate == sr)                                  // Equalizer.cpp:190-191
        return;
    nSampleRate = uint32_t(sr);
    filter_params_t fp;
    for (uint32_t i = 0; i < nFilters; ++i)                 // every filter is update()d (:196-200)
    {
        vFilters[i].get_params(&fp);
        vFilters[i].update(nSampleRate, &fp);
    }
    nFlags |= EF_REBUILD | EF_CLEAR;
    if (impl() != nullptr)
        mi_equalizer_bank_set_sample_rate(impl()->bank, uint32_t(sr));
}

size_t Equalizer::get_latency()
{
    uint32_t lat = 0;
    if (impl() != nullptr)
    {
        last_status(mi_equalizer_bank_get_latency(impl()->bank, &lat, nullptr));
        rebuilt();
    }
    nLatency = lat;
    return lat;
}

bool Equalizer::freq_chart(size_t id, float *c, const float *f, size_t count)
{
    filter_params_t fp;
    if (!get_params(id, &fp))
        return false;
    return mi_filter_freq_chart(cfp(&fp), nSampleRate, c, f, count) == MI_OK;
}

bool Equalizer::freq_chart(size_t id, float *re, float *im, const float *f, size_t count)
{
    std::vector<float> c(2 * count);
    if (!freq_chart(id, c.data(), f, count))
        return false;
    for (size_t i = 0; i < count; ++i)
    {
        re[i] = c[2 * i];
        im[i] = c[2 * i + 1];
    }
    return true;
}

void Equalizer::freq_chart(float *c, const float *f, size_t count)
{
    for (size_t i = 0; i < count; ++i)
    {
        c[2 * i] = 1.0f;
        c[2 * i + 1] = 0.0f;
    }
    std::vector<float> t(2 * count);
    for (size_t id = 0; id < nFilters; ++id)
    {
        filter_params_t fp;
        if (!get_params(id, &fp) || fp.nType == FLT_NONE)
            continue;
        if (mi_filter_freq_chart(cfp(&fp), nSampleRate, t.data(), f, count) != MI_OK)
            continue;
        for (size_t i = 0; i < count; ++i)
        {
            const float re = c[2 * i] * t[2 * i] - c[2 * i + 1] * t[2 * i + 1];
            const float im = c[2 * i] * t[2 * i + 1] + c[2 * i + 1] * t[2 * i];
            c[2 * i] = re;
            c[2 * i + 1] = im;
        }
    }
}

void Equalizer::freq_chart(float *re, float *im, const float *f, size_t count)
{
    std::vector<float> c(2 * count);
    freq_chart(c.data(), f, count);
    for (size_t i = 0; i < count; ++i)
    {
        re[i] = c[2 * i];
        im[i] = c[2 * i + 1];
    }
}

void Equalizer::process(float *out, const float *in, size_t samples)
{
    if (samples == 0)
        return;
    impl_t *p = impl();
    if (p == nullptr || !p->st.reserve(samples) || !p->st.up(in, samples) ||
        last_status(mi_equalizer_bank_process(p->bank, p->st.d_out, p->st.d_in, samples, samples, samples, nullptr)) != MI_OK ||
        !p->st.down(out, samples))
    {
        if (out != in)
            std::memmove(out, in, samples * sizeof(float));
        return;
    }
    if (nFlags & (EF_REBUILD | EF_CLEAR))
    {
        uint32_t lat = 0;
        mi_equalizer_bank_get_latency(p->bank, &lat, nullptr);
        nLatency = lat;
        rebuilt();
    }
}

void Equalizer::reset()                     { if (impl()) mi_equalizer_bank_reset(impl()->bank, nullptr); }
bool Equalizer::smooth() const              { return (nFlags & EF_SMOOTH) != 0; }

void Equalizer::set_smooth(bool smooth)
{
    nFlags = smooth ? (nFlags | EF_SMOOTH) : (nFlags & ~size_t(EF_SMOOTH));
    if (impl() != nullptr)
        mi_equalizer_bank_set_smooth(impl()->bank, smooth ? 1 : 0);
}

size_t Equalizer::ir_size() const
{
    uint32_t n = 0;
    if (impl() != nullptr)
        mi_equalizer_bank_info(impl()->bank, nullptr, nullptr, nullptr, &n);
    return n;
}

void Equalizer::dump(IStateDumper *v) const
{
    // Equalizer.cpp:628-652
    v->write_object("sBank", &sBank);
    v->begin_array("vFilters", vFilters, nFilters);
    for (size_t i = 0; vFilters != nullptr && i < nFilters; ++i)
        v->write_object(&vFilters[i]);
    v->end_array();
    v->write("nFilters", nFilters);
    v->write("nSampleRate", nSampleRate);
    v->write("nFirSize", nFirSize);
    v->write("nFirRank", nFirRank);
    v->write("nLatency", nLatency);
    v->write("nBufSize", nBufSize);
    v->write("nMode", int(nMode));
    v->write("vInBuffer", vInBuffer);
    v->write("vOutBuffer", vOutBuffer);
    v->write("vConv", vConv);
    v->write("vNewConv", vNewConv);
    v->write("vFft", vFft);
    v->write("vTemp", vTemp);
    v->write("nFlags", nFlags);
    v->write("pData", pData);
}

// ---- DynamicFilters ----------------------------------------------------------------------------------------------
// Reference members (filters/DynamicFilters.h:43-75).  vFilters is the host array the inline members read and write;
// the raw parameters of set_params() are kept next to the device bank, which needs them untransformed.
struct DynamicFilters::impl_t
{
    mi_dynfilter_bank_t            *bank = nullptr;
    std::vector<filter_params_t>    raw;            // as handed to set_params()
    std::vector<uint8_t>            sent;           // the bank has this filter's parameters
    staging                         st;
    float                          *d_gain = nullptr;
    size_t                          gain_cap = 0;
};
static_assert(sizeof(DynamicFilters) == 64, "DynamicFilters keeps the reference's layout");

DynamicFilters::DynamicFilters() { construct(); }
DynamicFilters::~DynamicFilters() { destroy(); }

void DynamicFilters::construct()                            // DynamicFilters.cpp:56-66
{
    vFilters = nullptr;
    vCascades = nullptr;
    vMemory = nullptr;
    vBiquads.ptr = nullptr;
    nFilters = 0;
    nSampleRate = 0;
    pData = nullptr;
    bClearMem = false;
}

status_t DynamicFilters::init(size_t filters)
{
    destroy();
    impl_t *p = new (std::nothrow) impl_t();
    filter_t *fl = static_cast<filter_t *>(std::calloc(filters ? filters : 1, sizeof(filter_t)));
    if (p == nullptr || fl == nullptr ||
        last_status(mi_dynfilter_bank_create(&p->bank, 1, uint32_t(filters))) != MI_OK)
    {
        delete p;
        std::free(fl);
        return STATUS_NO_MEM;
    }
    for (size_t i = 0; i < filters; ++i)                    // DynamicFilters.cpp:95-108
    {
        fl[i].sParams.nType = FLT_NONE;
        fl[i].sParams.nSlope = 0;
        fl[i].bActive = false;
    }
    p->raw.assign(filters, fl[0].sParams);
    p->sent.assign(filters, 1);
    pData = p;
    vFilters = fl;
    nFilters = filters;
    return STATUS_OK;
}

void DynamicFilters::destroy()
{
    if (impl_t *p = impl())
    {
        mi_dynfilter_bank_destroy(p->bank);
        mi_dspu_free(p->d_gain);
        p->st.release();
        delete p;
    }
    std::free(vFilters);
    construct();
}

void DynamicFilters::set_sample_rate(size_t sr)
{
    nSampleRate = sr;
    if (impl() != nullptr)
        mi_dynfilter_bank_set_sample_rate(impl()->bank, uint32_t(sr));
}

bool DynamicFilters::set_params(size_t id, const filter_params_t *params)
{
    impl_t *p = impl();
    if (p == nullptr || id >= nFilters)
        return false;
    if (vFilters[id].sParams.nType != params->nType)        // DynamicFilters.cpp:132-133
        bClearMem = true;
    p->raw[id] = *params;
    if (last_status(mi_dynfilter_bank_set_params(p->bank, uint32_t(id), cfp(params))) == MI_OK)
        mi_dynfilter_bank_get_params(p->bank, uint32_t(id), cfp(&vFilters[id].sParams), nullptr);     // transformed (:170-178)
    else
        vFilters[id].sParams = *params;                     // a type without a dynamic form: process() copies
    return true;
}

bool DynamicFilters::get_params(size_t id, filter_params_t *params)
{
    if (id >= nFilters)
        return false;
    *params = vFilters[id].sParams;
    return true;
}

void DynamicFilters::process(size_t id, float *out, const float *in, const float *gain, size_t samples)
{
    if (samples == 0)
        return;
    impl_t *p = impl();
    const filter_t *f = (id < nFilters) ? &vFilters[id] : nullptr;
    bool done = false;
    if (p != nullptr && f != nullptr && f->bActive && f->sParams.nType != FLT_NONE && f->sParams.nSlope != 0 && nSampleRate != 0)
    {
        // the inline set_filter_active() only touches the host array: tell the bank now
        mi_dynfilter_bank_set_filter_active(p->bank, uint32_t(id), 1);
        if (samples > p->gain_cap)
        {
            mi_dspu_free(p->d_gain);
            p->d_gain = nullptr;
            p->gain_cap = 0;
            if (mi_dspu_malloc(reinterpret_cast<void **>(&p->d_gain), samples * sizeof(float)) == MI_OK)
                p->gain_cap = samples;
        }
        done = p->gain_cap >= samples && p->st.reserve(samples) && p->st.up(in, samples) &&
               mi_dspu_copy_h2d(p->d_gain, gain, samples * sizeof(float), nullptr) == MI_OK &&
               last_status(mi_dynfilter_bank_process(p->bank, uint32_t(id), p->st.d_out, p->st.d_in, p->d_gain, samples,
                                                     samples, samples, samples, nullptr)) == MI_OK &&
               p->st.down(out, samples);
        bClearMem = false;
    }
    if (!done && out != in)                                 // DynamicFilters.cpp:207-212
        std::memmove(out, in, samples * sizeof(float));
}

bool DynamicFilters::freq_chart(size_t id, float *dst, const float *f, float gain, size_t count)
{
    impl_t *p = impl();
    if (p == nullptr || id >= nFilters)
        return false;
    return mi_dynfilter_freq_chart(cfp(&p->raw[id]), uint32_t(nSampleRate ? nSampleRate : 48000), dst, f, gain, count) == MI_OK;
}

bool DynamicFilters::freq_chart(size_t id, float *re, float *im, const float *f, float gain, size_t count)
{
    std::vector<float> c(2 * count);
    if (!freq_chart(id, c.data(), f, gain, count))
        return false;
    for (size_t i = 0; i < count; ++i)
    {
        re[i] = c[2 * i];
        im[i] = c[2 * i + 1];
    }
    return true;
}

void DynamicFilters::dump(IStateDumper *v) const
{
    // DynamicFilters.cpp:1973-1998
    v->begin_array("vFilters", vFilters, nFilters);
    for (size_t i = 0; vFilters != nullptr && i < nFilters; ++i)
    {
        const filter_t *f = &vFilters[i];
        v->begin_object(f, sizeof(filter_t));
        v->write("nType", f->sParams.nType);
        v->write("fFreq", f->sParams.fFreq);
        v->write("fFreq2", f->sParams.fFreq2);
        v->write("fGain", f->sParams.fGain);
        v->write("nSlope", f->sParams.nSlope);
        v->write("fQuality", f->sParams.fQuality);
        v->write("bActive", f->bActive);
        v->end_object();
    }
    v->end_array();
    v->write("vCascades", vCascades);
    v->write("vBiquads", vBiquads.ptr);
    v->write("nFilters", nFilters);
    v->write("nSampleRate", nSampleRate);
    v->write("pData", pData);
    v->write("bClearMem", bClearMem);
}

// ---- Convolver ---------------------------------------------------------------------------------------------------
// Reference members (util/Convolver.h:38-56).  The counters describe the response as Convolver::init computes them
// (Convolver.cpp:87-142); the device-side state hangs off vData.
struct Convolver::impl_t
{
    mi_convolver_bank_t *bank = nullptr;
    staging st;
};
static_assert(sizeof(Convolver) == 144, "Convolver keeps the reference's layout (SURVEY.md 0.4)");

Convolver::Convolver() { construct(); }
Convolver::~Convolver() { destroy(); }

void Convolver::construct()
{
    vDataBuffer = vFrame = vConvBuffer = vTaskData = vConvData = vDirectData = nullptr;
    nDataBufferSize = nDirectSize = nFrameSize = nFrameOff = nConvSize = 0;
    nLevels = nBlocks = nBlocksDone = nRank = nBlkInit = 0;
    fBlkCoef = 0.0f;
    vData = nullptr;
}

void Convolver::destroy()
{
    if (impl_t *p = impl())
    {
        mi_convolver_bank_destroy(p->bank);
        p->st.release();
        delete p;
    }
    construct();
}

bool Convolver::init(const float *data, size_t count, size_t rank, float phase)
{
    destroy();
    if (count == 0)                                     // stays uninitialised, still "success"
        return true;
    impl_t *p = new (std::nothrow) impl_t();
    if (p == nullptr)
        return false;
    if (last_status(mi_convolver_bank_create(&p->bank, 1, data, count, nullptr, uint32_t(count), uint32_t(rank), phase, nullptr)) != MI_OK)
    {
        delete p;
        return false;
    }
    vData = reinterpret_cast<uint8_t *>(p);
    // the reference's bookkeeping for this response (Convolver.cpp:87,90-93,137-142,166-210): same numbers, no buffers
    const size_t r          = std::min<size_t>(std::max<size_t>(rank, CONVOLVER_RANK_MIN), CONVOLVER_RANK_MAX);
    const size_t frame      = size_t(1) << (r - 1);
    const size_t bins       = (count + frame - 1) >> (r - 1);
    const size_t head       = size_t(1) << (CONVOLVER_RANK_MIN - 1);
    nRank                   = r;
    nConvSize               = count;
    nDataBufferSize         = (bins + 1) * frame;
    nFrameSize              = frame;
    nFrameOff               = size_t(phase * float(frame)) % frame;
    nDirectSize             = std::min(count, head);
    size_t left             = count - nDirectSize;
    nLevels                 = 0;
    for (size_t brank = CONVOLVER_RANK_MIN; left > 0 && brank < r; ++brank)     // raising levels
    {
        left               -= std::min(left, size_t(1) << (brank - 1));
        ++nLevels;
    }
    nBlocks                 = (left + frame - 1) / frame;                       // constant-size blocks
    nBlocksDone             = nBlocks;
    const size_t steps      = frame >> (CONVOLVER_RANK_MIN - 1);
    if (steps <= 1)
    {
        nBlkInit            = nBlocks;
        fBlkCoef            = 0.0f;
    }
    else
    {
        nBlkInit            = 1;
        fBlkCoef            = (float(nBlocks) + 1e-3f) / (float(steps) - 1.0f);
    }
    return true;
}

void Convolver::process(float *dst, const float *src, size_t count)
{
    if (count == 0)
        return;
    impl_t *p = impl();
    if (p == nullptr || !p->st.reserve(count) || !p->st.up(src, count) ||
        last_status(mi_convolver_bank_process(p->bank, p->st.d_out, p->st.d_in, count, count, count, nullptr)) != MI_OK ||
        !p->st.down(dst, count))
    {
        std::memset(dst, 0, count * sizeof(float));
        return;
    }
    nFrameOff = (nFrameOff + count) & (nFrameSize - 1);     // Convolver.cpp:299-305
}

void Convolver::dump(IStateDumper *v) const
{
    // Convolver.cpp:315-337 (the first key is spelt "pDataBuffer" there)
    v->write("pDataBuffer", vDataBuffer);
    v->write("vFrame", vFrame);
    v->write("vConvBuffer", vConvBuffer);
    v->write("vTaskData", vTaskData);
    v->write("vConvData", vConvData);
    v->write("vDirectData", vDirectData);
    v->write("nDataBufferSize", nDataBufferSize);
    v->write("nDirectSize", nDirectSize);
    v->write("nFrameSize", nFrameSize);
    v->write("nFrameOff", nFrameOff);
    v->write("nConvSize", nConvSize);
    v->write("nLevels", nLevels);
    v->write("nBlocks", nBlocks);
    v->write("nBlocksDone", nBlocksDone);
    v->write("nRank", nRank);
    v->write("nBlkInit", nBlkInit);
    v->write("fBlkCoef", fBlkCoef);
    v->write("vData", vData);
}

// ---- SpectralProcessor ------------------------------------------------------------------------------------------
// Reference members (util/SpectralProcessor.h:47-62): rank, phase, update flag and the binding are the members
// themselves; the device-side state hangs off pData.
struct SpectralProcessor::impl_t
{
    mi_spectral_bank_t *bank = nullptr;
    SpectralProcessor  *owner = nullptr;
    std::vector<float> host_spec;
    staging st;

    // device-side hook: bring the spectrum to the host, run the user's function, send it back
    static void trampoline(void *object, void *, float *spectrum, size_t rank, size_t, void *stream)
    {
        impl_t *p = static_cast<impl_t *>(object);
        const size_t floats = size_t(2) << rank;
        p->host_spec.resize(floats);
        if (mi_dspu_copy_d2h(p->host_spec.data(), spectrum, floats * sizeof(float), stream) != MI_OK ||
            mi_dspu_stream_synchronize(stream) != MI_OK)
            return;
        p->owner->pFunc(p->owner->pObject, p->owner->pSubject, p->host_spec.data(), rank);
        mi_dspu_copy_h2d(spectrum, p->host_spec.data(), floats * sizeof(float), stream);
        mi_dspu_stream_synchronize(stream);
    }
};

static_assert(sizeof(SpectralProcessor) == 104, "SpectralProcessor keeps the reference's layout");

SpectralProcessor::SpectralProcessor() { construct(); }
SpectralProcessor::~SpectralProcessor() { destroy(); }

void SpectralProcessor::construct()                         // SpectralProcessor.cpp:33-50
{
    nRank = nMaxRank = 0;
    fPhase = 0.0f;
    pWnd = pOutBuf = pInBuf = pFftBuf = nullptr;
    nOffset = 0;
    pData = nullptr;
    bUpdate = true;
    pFunc = nullptr;
    pObject = pSubject = nullptr;
}

bool SpectralProcessor::init(size_t max_rank)
{
    destroy();
    impl_t *p = new (std::nothrow) impl_t();
    if (p == nullptr)
        return false;
    // the library holds frames up to 2^13; larger max_rank values are accepted as long as the rank in use fits
    const uint32_t cap = uint32_t(std::min<size_t>(std::max<size_t>(max_rank, 5), 18));
    if (mi_spectral_bank_create(&p->bank, 1, cap) != MI_OK)
    {
        delete p;
        return false;
    }
    p->owner = this;
    pData = reinterpret_cast<uint8_t *>(p);
    nMaxRank = nRank = max_rank;
    fPhase = 0.0f;
    nOffset = 0;
    bUpdate = true;
    return true;
}

void SpectralProcessor::destroy()
{
    if (impl_t *p = impl())
    {
        mi_spectral_bank_destroy(p->bank);
        p->st.release();
        delete p;
    }
    construct();
}

void SpectralProcessor::bind(spectral_processor_func_t func, void *object, void *subject)
{
    pFunc = func;
    pObject = object;
    pSubject = subject;
    if (impl() == nullptr)
        return;
    if (func != nullptr)
        mi_spectral_bank_bind(impl()->bank, &impl_t::trampoline, impl(), nullptr);
    else
        mi_spectral_bank_unbind(impl()->bank);
}

void SpectralProcessor::unbind()            { bind(nullptr, nullptr, nullptr); }
void SpectralProcessor::update_settings()   { bUpdate = false; }

void SpectralProcessor::set_phase(float phase)
{
    fPhase = std::min(std::max(phase, 0.0f), 1.0f);
    bUpdate = true;
    if (impl() != nullptr)
        mi_spectral_bank_set_phase(impl()->bank, fPhase);
}

void SpectralProcessor::set_rank(size_t rank)
{
    if (rank == nRank || rank > nMaxRank)
        return;
    nRank = rank;
    bUpdate = true;
    if (impl() != nullptr)
        mi_spectral_bank_set_rank(impl()->bank, uint32_t(rank));
}

void SpectralProcessor::process(float *dst, const float *src, size_t count)
{
    if (count == 0)
        return;
    impl_t *p = impl();
    if (p == nullptr || !p->st.reserve(count) || !p->st.up(src, count) ||
        last_status(mi_spectral_bank_process(p->bank, p->st.d_out, p->st.d_in, count, count, count, nullptr)) != MI_OK ||
        !p->st.down(dst, count))
    {
        std::memset(dst, 0, count * sizeof(float));
        return;
    }
    bUpdate = false;
    nOffset = (size_t(1) << (nRank - 1)) - remaining();
}

void SpectralProcessor::process(const float *src, size_t count)
{
    impl_t *p = impl();
    if (count == 0 || p == nullptr || !p->st.reserve(count) || !p->st.up(src, count))
        return;
    last_status(mi_spectral_bank_process(p->bank, nullptr, p->st.d_in, count, count, count, nullptr));
    mi_dspu_stream_synchronize(nullptr);
    bUpdate = false;
    nOffset = (size_t(1) << (nRank - 1)) - remaining();
}

void SpectralProcessor::reset()             { if (impl()) mi_spectral_bank_reset(impl()->bank, nullptr); nOffset = 0; }

size_t SpectralProcessor::remaining() const
{
    uint32_t r = 0;
    if (impl() != nullptr)
        mi_spectral_bank_get(impl()->bank, nullptr, nullptr, &r);
    return r;
}

void SpectralProcessor::dump(IStateDumper *v) const
{
    // SpectralProcessor.cpp:265-281
    v->write("nRank", nRank);
    v->write("nMaxRank", nMaxRank);
    v->write("fPhase", fPhase);
    v->write("pWnd", pWnd);
    v->write("pOutBuf", pOutBuf);
    v->write("pInBuf", pInBuf);
    v->write("pFftBuf", pFftBuf);
    v->write("nOffset", nOffset);
    v->write("pData", pData);
    v->write("bUpdate", bUpdate);
    v->write("pFunc", reinterpret_cast<const void *>(pFunc));
    v->write("pObject", pObject);
    v->write("pSubject", pSubject);
}

// ---- MultiSpectralProcessor -------------------------------------------------------------------------------------
struct MultiSpectralProcessor::impl_t
{
    mi_spectral_bank_t *bank = nullptr;
    MultiSpectralProcessor *owner = nullptr;
    std::vector<channel_t>     ch;          // storage behind vChannels
    std::vector<uint8_t>       has_in, has_out;
    std::vector<float>         host_io, host_spec;
    std::vector<float *>       spec_ptr;
    float  *d_in = nullptr, *d_out = nullptr;
    size_t  cap = 0;

    bool reserve(size_t channels, size_t count)
    {
        if (count <= cap)
            return true;
        mi_dspu_free(d_in); mi_dspu_free(d_out);
        d_in = d_out = nullptr;
        cap = 0;
        if (mi_dspu_malloc(reinterpret_cast<void **>(&d_in), channels * count * sizeof(float)) != MI_OK) return false;
        if (mi_dspu_malloc(reinterpret_cast<void **>(&d_out), channels * count * sizeof(float)) != MI_OK) return false;
        cap = count;
        return true;
    }

    // device-side hook: bring all spectra to the host, run the user's handler on the per-channel pointers, send them back
    static void trampoline(void *object, void *, float *spectrum, size_t rank, size_t channels, void *stream)
    {
        impl_t *p = static_cast<impl_t *>(object);
        MultiSpectralProcessor *o = p->owner;
        const size_t floats = size_t(2) << rank;
        p->host_spec.resize(floats * channels);
        if (o->pFunc == nullptr ||
            mi_dspu_copy_d2h(p->host_spec.data(), spectrum, floats * channels * sizeof(float), stream) != MI_OK ||
            mi_dspu_stream_synchronize(stream) != MI_OK)
            return;
        p->spec_ptr.resize(channels);
        for (size_t i = 0; i < channels; ++i)
            p->spec_ptr[i] = p->has_in[i] ? &p->host_spec[i * floats] : nullptr;    // MultiSpectralProcessor.cpp:338-350
        o->pFunc(o->pObject, o->pSubject, p->spec_ptr.data(), rank);
        mi_dspu_copy_h2d(spectrum, p->host_spec.data(), floats * channels * sizeof(float), stream);
        mi_dspu_stream_synchronize(stream);
    }
};

MultiSpectralProcessor::MultiSpectralProcessor() { construct(); }
MultiSpectralProcessor::~MultiSpectralProcessor() { destroy(); }

void MultiSpectralProcessor::construct()                    // MultiSpectralProcessor.cpp:33-58
{
    nChannels = nRank = nMaxRank = nOffset = 0;
    vChannels = nullptr;
    vFftBuf = nullptr;
    pWnd = nullptr;
    fPhase = 0.0f;
    bUpdate = true;
    pFunc = nullptr;
    pObject = pSubject = nullptr;
    pData = nullptr;
}

bool MultiSpectralProcessor::init(size_t channels, size_t max_rank)
{
    if (channels <= 0)                                      // MultiSpectralProcessor.cpp:62-63
        return false;
    impl_t *p = new (std::nothrow) impl_t();
    if (p == nullptr)
        return false;
    const uint32_t cap = uint32_t(std::min<size_t>(std::max<size_t>(max_rank, 5), 18));
    if (mi_spectral_bank_create(&p->bank, uint32_t(channels), cap) != MI_OK)
    {
        delete p;
        return false;
    }
    mi_spectral_bank_set_timing(p->bank, 1);                // transform as soon as the frame is complete (:324)
    destroy();
    p->owner = this;
    p->ch.assign(channels, channel_t{ nullptr, nullptr, nullptr, nullptr, nullptr });
    p->has_in.assign(channels, 0);
    p->has_out.assign(channels, 0);
    pData = reinterpret_cast<uint8_t *>(p);
    vChannels = p->ch.data();
    nChannels = uint32_t(channels);
    nMaxRank = nRank = uint32_t(max_rank);
    nOffset = 0;
    fPhase = 0.0f;
    bUpdate = true;
    pFunc = nullptr;
    pObject = pSubject = nullptr;
    return true;
}

void MultiSpectralProcessor::destroy()
{
    if (impl_t *p = impl())
    {
        mi_spectral_bank_destroy(p->bank);
        mi_dspu_free(p->d_in);
        mi_dspu_free(p->d_out);
        delete p;
    }
    pData = nullptr;
    vChannels = nullptr;
}

void MultiSpectralProcessor::bind_handler(multi_spectral_processor_func_t func, void *object, void *subject)
{
    impl_t *p = impl();
    if (p == nullptr)
        return;
    pFunc = func;
    pObject = object;
    pSubject = subject;
    if (func != nullptr)
        mi_spectral_bank_bind(p->bank, &impl_t::trampoline, p, nullptr);
    else
        mi_spectral_bank_unbind(p->bank);
}

void MultiSpectralProcessor::unbind_handler() { bind_handler(nullptr, nullptr, nullptr); }

status_t MultiSpectralProcessor::bind(size_t index, float *out, const float *in)
{
    if (impl() == nullptr)
        return STATUS_BAD_STATE;
    if (index >= nChannels)
        return STATUS_INVALID_VALUE;
    vChannels[index].pIn = in;
    vChannels[index].pOut = out;
    return STATUS_OK;
}

status_t MultiSpectralProcessor::bind_in(size_t index, const float *in)
{
    if (impl() == nullptr)
        return STATUS_BAD_STATE;
    if (index >= nChannels)
        return STATUS_INVALID_VALUE;
    vChannels[index].pIn = in;
    return STATUS_OK;
}

status_t MultiSpectralProcessor::bind_out(size_t index, float *out)
{
    if (impl() == nullptr)
        return STATUS_BAD_STATE;
    if (index >= nChannels)
        return STATUS_INVALID_VALUE;
    vChannels[index].pOut = out;
    return STATUS_OK;
}

status_t MultiSpectralProcessor::unbind(size_t index)     { return bind(index, nullptr, nullptr); }
status_t MultiSpectralProcessor::unbind_in(size_t index)  { return bind_in(index, nullptr); }
status_t MultiSpectralProcessor::unbind_out(size_t index) { return bind_out(index, nullptr); }

void MultiSpectralProcessor::unbind_all()
{
    for (uint32_t i = 0; impl() != nullptr && i < nChannels; ++i)
    {
        vChannels[i].pIn = nullptr;
        vChannels[i].pOut = nullptr;
    }
}

void MultiSpectralProcessor::update_settings()     { if (impl() != nullptr) bUpdate = false; }

void MultiSpectralProcessor::set_phase(float phase)
{
    impl_t *p = impl();
    if (p == nullptr)
        return;
    fPhase = std::min(std::max(phase, 0.0f), 1.0f);
    bUpdate = true;
    mi_spectral_bank_set_phase(p->bank, fPhase);
}

void MultiSpectralProcessor::set_rank(size_t rank)
{
    impl_t *p = impl();
    if (p == nullptr || rank == nRank || rank > nMaxRank)
        return;
    nRank = uint32_t(rank);
    bUpdate = true;
    mi_spectral_bank_set_rank(p->bank, uint32_t(rank));
}

void MultiSpectralProcessor::process(size_t count)
{
    impl_t *p = impl();
    if (p == nullptr || count == 0)
        return;
    const size_t C = nChannels;
    for (size_t i = 0; i < C; ++i)
    {
        p->has_in[i]  = (vChannels[i].pIn != nullptr) ? 1 : 0;
        p->has_out[i] = (vChannels[i].pOut != nullptr) ? 1 : 0;
    }
    bool ok = p->reserve(C, count) &&
              mi_spectral_bank_bind_channels(p->bank, p->has_in.data(), p->has_out.data(), nullptr) == MI_OK;
    if (ok)
    {
        // channels without an input take zeros (MultiSpectralProcessor.cpp:316-317)
        p->host_io.assign(C * count, 0.0f);
        for (size_t i = 0; i < C; ++i)
            if (vChannels[i].pIn != nullptr)
                std::memcpy(&p->host_io[i * count], vChannels[i].pIn, count * sizeof(float));
        ok = mi_dspu_copy_h2d(p->d_in, p->host_io.data(), C * count * sizeof(float), nullptr) == MI_OK &&
             mi_spectral_bank_process(p->bank, p->d_out, p->d_in, count, count, count, nullptr) == MI_OK &&
             mi_dspu_copy_d2h(p->host_io.data(), p->d_out, C * count * sizeof(float), nullptr) == MI_OK &&
             mi_dspu_stream_synchronize(nullptr) == MI_OK;
    }
    for (size_t i = 0; i < C; ++i)                           // the bound pointers move on (:310-320)
    {
        if (vChannels[i].pOut != nullptr)
        {
            if (ok)
                std::memcpy(vChannels[i].pOut, &p->host_io[i * count], count * sizeof(float));
            else
                std::memset(vChannels[i].pOut, 0, count * sizeof(float));
            vChannels[i].pOut += count;
        }
        if (vChannels[i].pIn != nullptr)
            vChannels[i].pIn += count;
    }
    uint32_t r = 0;
    mi_spectral_bank_get(p->bank, nullptr, nullptr, &r);
    nOffset = uint32_t((size_t(1) << (nRank - 1)) - r);
    bUpdate = false;
}

void MultiSpectralProcessor::reset()             { if (impl_t *p = impl()) mi_spectral_bank_reset(p->bank, nullptr); }

size_t MultiSpectralProcessor::remaining() const
{
    uint32_t r = 0;
    if (impl_t *p = impl())
        mi_spectral_bank_get(p->bank, nullptr, nullptr, &r);
    return r;
}

void MultiSpectralProcessor::dump(IStateDumper *v) const
{
    // MultiSpectralProcessor.cpp:411-443 (the channel records go out as bare runs of five keys, without objects, there too)
    v->write("nChannels", nChannels);
    v->write("nRank", nRank);
    v->write("nMaxRank", nMaxRank);
    v->write("nOffset", nOffset);
    v->begin_array("vChannels", vChannels, nChannels);
    for (size_t i = 0; vChannels != nullptr && i < nChannels; ++i)
    {
        const channel_t *c = &vChannels[i];
        v->write("pIn", c->pIn);
        v->write("pOut", c->pOut);
        v->write("pInBuf", c->pInBuf);
        v->write("pOutBuf", c->pOutBuf);
        v->write("pFftBuf", c->pFftBuf);
    }
    v->end_array();
    v->writev("vFftBuf", vFftBuf, (vFftBuf != nullptr) ? size_t(nChannels) : 0);
    v->write("pWnd", pWnd);
    v->write("fPhase", fPhase);
    v->write("bUpdate", bUpdate);
    v->write("pFunc", reinterpret_cast<const void *>(pFunc));
    v->write("pObject", pObject);
    v->write("pSubject", pSubject);
    v->write("pData", pData);
}

// ---- Crossover --------------------------------------------------------------------------------------------------
struct Crossover::impl_t
{
    mi_crossover_bank_t *bank = nullptr;
    size_t  bands = 0;
    struct handler_t { crossover_func_t func = nullptr; void *object = nullptr, *subject = nullptr; };
    std::vector<handler_t> handlers;
    std::vector<float *>   d_band;          // one device buffer of buf_size samples per band
    std::vector<float>     host;
    float  *d_in = nullptr;

    ~impl_t()
    {
        mi_crossover_bank_destroy(bank);
        for (float *p : d_band)
            mi_dspu_free(p);
        mi_dspu_free(d_in);
    }
};

Crossover::Crossover() { construct(); }
Crossover::~Crossover() { destroy(); }

void Crossover::construct()                                 // Crossover.cpp:41-56
{
    nReconfigure = R_ALL;
    nSplits = 0;
    nBufSize = 0;
    nSampleRate = 48000;                                    // LSP_DSP_UNITS_DEFAULT_SAMPLE_RATE
    nPlanSize = 0;
    vBands = nullptr;
    vSplit = nullptr;
    vPlan = nullptr;
    vLpfBuf = vHpfBuf = nullptr;
    pData = nullptr;
}

void Crossover::destroy()
{
    delete impl();
    construct();
}

void Crossover::sync_flags()
{
    int pending = 0;
    impl_t *p = impl();
    nReconfigure = (p != nullptr && mi_crossover_bank_needs_reconfiguration(p->bank, &pending) == MI_OK && pending != 0) ? uint32_t(R_ALL) : 0u;
}

bool Crossover::init(size_t bands, size_t buf_size)
{
    if (bands < 1)                                          // Crossover.cpp:73-74
        return false;
    impl_t *p = new (std::nothrow) impl_t();
    if (p == nullptr)
        return false;
    const size_t cap = (buf_size > 0) ? buf_size : 1;
    bool ok = mi_crossover_bank_create(&p->bank, 1, uint32_t(bands)) == MI_OK;
    p->bands = bands;
    p->handlers.resize(bands);
    p->d_band.assign(bands, nullptr);
    for (size_t i = 0; ok && i < bands; ++i)
        ok = mi_dspu_malloc(reinterpret_cast<void **>(&p->d_band[i]), cap * sizeof(float)) == MI_OK;
    ok = ok && mi_dspu_malloc(reinterpret_cast<void **>(&p->d_in), cap * sizeof(float)) == MI_OK;
    if (!ok)
    {
        delete p;
        return false;
    }
    destroy();
    pData = reinterpret_cast<uint8_t *>(p);
    nSplits = uint32_t(bands - 1);
    nBufSize = uint32_t(buf_size);
    vLpfBuf = p->d_in;
    mi_crossover_bank_set_sample_rate(p->bank, nSampleRate);
    sync_flags();
    return true;
}

void Crossover::set_slope(size_t sp, size_t slope)
{
    if (impl() == nullptr)
        return;
    mi_crossover_bank_set_slope(impl()->bank, uint32_t(sp), uint32_t(slope));
    sync_flags();
}

ssize_t Crossover::get_slope(size_t sp) const
{
    uint32_t v = 0;
    return (impl() && sp < nSplits && mi_crossover_bank_get_split(impl()->bank, uint32_t(sp), &v, nullptr, nullptr) == MI_OK) ? ssize_t(v) : -1;
}

void Crossover::set_frequency(size_t sp, float freq)
{
    if (impl() == nullptr)
        return;
    mi_crossover_bank_set_frequency(impl()->bank, uint32_t(sp), freq);
    sync_flags();
}

float Crossover::get_frequency(size_t sp) const
{
    float v = -1.0f;
    return (impl() && sp < nSplits && mi_crossover_bank_get_split(impl()->bank, uint32_t(sp), nullptr, &v, nullptr) == MI_OK) ? v : -1.0f;
}

void Crossover::set_mode(size_t sp, crossover_mode_t mode)
{
    if (impl() == nullptr)
        return;
    mi_crossover_bank_set_mode(impl()->bank, uint32_t(sp), int(mode));
    sync_flags();
}

ssize_t Crossover::get_mode(size_t sp) const
{
    int v = -1;
    return (impl() && sp < nSplits && mi_crossover_bank_get_split(impl()->bank, uint32_t(sp), nullptr, nullptr, &v) == MI_OK) ? ssize_t(v) : -1;
}

void Crossover::set_gain(size_t band, float gain)
{
    if (impl() == nullptr)
        return;
    mi_crossover_bank_set_gain(impl()->bank, uint32_t(band), gain);
    sync_flags();
}

namespace
{
    // one field of a band after reconfigure(); `fallback` for bad indices (Crossover.cpp:256-325)
    float crossover_band_field(mi_crossover_bank_t *bank, size_t bands, size_t band, int which, float fallback)
    {
        float g = 0.0f, s = 0.0f, e = 0.0f;
        int a = 0;
        if (bank == nullptr || band >= bands || mi_crossover_bank_get_band(bank, uint32_t(band), &g, &s, &e, &a, nullptr) != MI_OK)
            return fallback;
        return (which == 0) ? g : (which == 1) ? s : (which == 2) ? e : float(a);
    }
}

// the getters below reconfigure first, as the reference's do (Crossover.cpp:262-325)
float Crossover::get_gain(size_t band) const
{
    return impl() ? crossover_band_field(impl()->bank, impl()->bands, band, 0, -1.0f) : -1.0f;
}

float Crossover::get_band_start(size_t band)
{
    const float v = impl() ? crossover_band_field(impl()->bank, impl()->bands, band, 1, -1.0f) : -1.0f;
    sync_flags();
    return v;
}

float Crossover::get_band_end(size_t band)
{
    const float v = impl() ? crossover_band_field(impl()->bank, impl()->bands, band, 2, -1.0f) : -1.0f;
    sync_flags();
    return v;
}

bool Crossover::band_active(size_t band)
{
    const bool v = impl() && crossover_band_field(impl()->bank, impl()->bands, band, 3, 0.0f) != 0.0f;
    sync_flags();
    return v;
}

bool Crossover::set_handler(size_t band, crossover_func_t func, void *object, void *subject)
{
    impl_t *p = impl();
    if (p == nullptr || band >= p->bands)
        return false;
    p->handlers[band].func = func;
    p->handlers[band].object = object;
    p->handlers[band].subject = subject;
    return true;
}

bool Crossover::unset_handler(size_t band)     { return set_handler(band, nullptr, nullptr, nullptr); }

void Crossover::set_sample_rate(size_t sr)
{
    if (nSampleRate == sr)                                  // Crossover.cpp:327-345
        return;
    nSampleRate = uint32_t(sr);
    if (impl() != nullptr)
    {
        mi_crossover_bank_set_sample_rate(impl()->bank, uint32_t(sr));
        sync_flags();
    }
}

void Crossover::reconfigure()
{
    if (impl() != nullptr)
        mi_crossover_bank_get_band(impl()->bank, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
    sync_flags();
}

bool Crossover::freq_chart(size_t band, float *c, const float *f, size_t count)
{
    impl_t *p = impl();
    if (p == nullptr || band >= p->bands)
        return false;
    const bool ok = mi_crossover_bank_freq_chart(p->bank, uint32_t(band), c, f, count, nullptr) == MI_OK;
    sync_flags();
    return ok;
}

bool Crossover::freq_chart(size_t band, float *re, float *im, const float *f, size_t count)
{
    impl_t *p = impl();
    if (p == nullptr || band >= p->bands)
        return false;
    std::vector<float> c(2 * count);
    const bool ok = mi_crossover_bank_freq_chart(p->bank, uint32_t(band), c.data(), f, count, nullptr) == MI_OK;
    sync_flags();
    if (!ok)
        return false;
    for (size_t i = 0; i < count; ++i)
    {
        re[i] = c[2 * i];
        im[i] = c[2 * i + 1];
    }
    return true;
}

void Crossover::process(const float *in, size_t samples)   // Crossover.cpp:451-498: chunks of buf_size, handlers per chunk
{
    impl_t *p = impl();
    if (p == nullptr)
        return;
    std::vector<float *> outs(p->bands);
    for (size_t sample = 0; sample < samples; )
    {
        const size_t to_do = std::min<size_t>(samples - sample, nBufSize);
        if (to_do == 0)
            break;
        for (size_t b = 0; b < p->bands; ++b)
            outs[b] = (p->handlers[b].func != nullptr) ? p->d_band[b] : nullptr;
        if (mi_dspu_copy_h2d(p->d_in, in, to_do * sizeof(float), nullptr) != MI_OK ||
            mi_crossover_bank_process(p->bank, outs.data(), p->d_in, to_do, to_do, to_do, nullptr) != MI_OK)
            break;
        p->host.resize(to_do);
        // handlers fire from the lowest band upwards, the order the reference walks its plan in
        std::vector<std::pair<float, size_t> > order;
        for (size_t b = 0; b < p->bands; ++b)
            if (outs[b] != nullptr && band_active(b))
                order.push_back(std::make_pair((b == 0) ? -1.0f : get_band_start(b), b));
        std::sort(order.begin(), order.end());
        for (size_t k = 0; k < order.size(); ++k)
        {
            const size_t b = order[k].second;
            if (mi_dspu_copy_d2h(p->host.data(), outs[b], to_do * sizeof(float), nullptr) != MI_OK ||
                mi_dspu_stream_synchronize(nullptr) != MI_OK)
                return;
            p->handlers[b].func(p->handlers[b].object, p->handlers[b].subject, b, p->host.data(), sample, to_do);
        }
        in += to_do;
        sample += to_do;
    }
    sync_flags();
}

void Crossover::dump(IStateDumper *v) const
{
    // Crossover.cpp:588-641.  The band and split records live in the device bank: their fields come from its getters, the
    // per-split Filter objects (sLPF / sHPF) do not exist on this side and go out as null.  Key spellings as in the
    // reference ("pOpbject", "nSlopw").
    impl_t *p = impl();
    const size_t bands = (p != nullptr) ? p->bands : 0, splits = (bands > 0) ? bands - 1 : 0;
    v->write("nReconfigure", nReconfigure);
    v->write("nSplits", nSplits);
    v->write("nBufSize", nBufSize);
    v->write("nSampleRate", nSampleRate);
    v->write("nPlanSize", nPlanSize);
    v->begin_array("vBands", vBands, bands);
    for (size_t i = 0; i < bands; ++i)
    {
        float gain = 0.0f, start = 0.0f, end = 0.0f;
        int active = 0;
        mi_crossover_bank_get_band(p->bank, uint32_t(i), &gain, &start, &end, &active, nullptr);
        v->begin_object(&p->handlers[i], sizeof(impl_t::handler_t));
        v->write("fGain", gain);
        v->write("fStart", start);
        v->write("fEnd", end);
        v->write("bEnabled", active != 0);
        v->write("pStart", static_cast<const void *>(nullptr));
        v->write("pEnd", static_cast<const void *>(nullptr));
        v->write("pFunc", reinterpret_cast<const void *>(p->handlers[i].func));
        v->write("pOpbject", p->handlers[i].object);
        v->write("pSubject", p->handlers[i].subject);
        v->write("nId", i);
        v->end_object();
    }
    v->end_array();
    v->begin_array("vSplit", vSplit, splits);
    for (size_t i = 0; i < splits; ++i)
    {
        uint32_t slope = 0;
        float freq = 0.0f;
        int mode = 0;
        mi_crossover_bank_get_split(p->bank, uint32_t(i), &slope, &freq, &mode);
        v->begin_object(p->bank, 0);
        v->write("sLPF", static_cast<const void *>(nullptr));
        v->write("sHPF", static_cast<const void *>(nullptr));
        v->write("nBandId", i + 1);
        v->write("nSlopw", size_t(slope));
        v->write("fFreq", freq);
        v->write("nMode", mode);
        v->end_object();
    }
    v->end_array();
    v->writev("vPlan", reinterpret_cast<const void * const *>(vPlan), (vPlan != nullptr) ? size_t(nPlanSize) : 0);
    v->write("vLpfBuf", vLpfBuf);
    v->write("vHpfBuf", vHpfBuf);
    v->write("pData", pData);
}

// ---- envelope::* -----------------------------------------------------------------------------------------------
namespace envelope
{
    void noise_lin(float *dst, float first, float last, float center, size_t n, envelope_t type)         { mi_envelope_noise_lin(dst, first, last, center, n, int(type)); }
    void reverse_noise_lin(float *dst, float first, float last, float center, size_t n, envelope_t type) { mi_envelope_reverse_noise_lin(dst, first, last, center, n, int(type)); }
    void white_noise_lin(float *dst, float first, float last, float center, size_t n, envelope_t)        { mi_envelope_noise_lin(dst, first, last, center, n, MI_ENVELOPE_WHITE_NOISE); }
    void pink_noise_lin(float *dst, float first, float last, float center, size_t n, envelope_t)         { mi_envelope_noise_lin(dst, first, last, center, n, MI_ENVELOPE_PINK_NOISE); }
    void brown_noise_lin(float *dst, float first, float last, float center, size_t n, envelope_t)        { mi_envelope_noise_lin(dst, first, last, center, n, MI_ENVELOPE_BROWN_NOISE); }
    void blue_noise_lin(float *dst, float first, float last, float center, size_t n, envelope_t)         { mi_envelope_noise_lin(dst, first, last, center, n, MI_ENVELOPE_BLUE_NOISE); }
    void violet_noise_lin(float *dst, float first, float last, float center, size_t n, envelope_t)       { mi_envelope_noise_lin(dst, first, last, center, n, MI_ENVELOPE_VIOLET_NOISE); }
    void noise_log(float *dst, float first, float last, float center, size_t n, envelope_t type)         { mi_envelope_noise_log(dst, first, last, center, n, int(type), 0); }
    void reverse_noise_log(float *dst, float first, float last, float center, size_t n, envelope_t type) { mi_envelope_noise_log(dst, first, last, center, n, int(type), 1); }
    void white_noise_log(float *dst, float first, float last, float center, size_t n, envelope_t)        { mi_envelope_noise_log(dst, first, last, center, n, MI_ENVELOPE_WHITE_NOISE, 0); }
    void pink_noise_log(float *dst, float first, float last, float center, size_t n, envelope_t)         { mi_envelope_noise_log(dst, first, last, center, n, MI_ENVELOPE_PINK_NOISE, 0); }
    void brown_noise_log(float *dst, float first, float last, float center, size_t n, envelope_t)        { mi_envelope_noise_log(dst, first, last, center, n, MI_ENVELOPE_BROWN_NOISE, 0); }
    void blue_noise_log(float *dst, float first, float last, float center, size_t n, envelope_t)         { mi_envelope_noise_log(dst, first, last, center, n, MI_ENVELOPE_BLUE_NOISE, 0); }
    void violet_noise_log(float *dst, float first, float last, float center, size_t n, envelope_t)       { mi_envelope_noise_log(dst, first, last, center, n, MI_ENVELOPE_VIOLET_NOISE, 0); }
    void noise_list(float *dst, const float *freqs, float center, size_t n, envelope_t type)             { mi_envelope_noise_list(dst, freqs, center, n, int(type), 0); }
    void reverse_noise_list(float *dst, const float *freqs, float center, size_t n, envelope_t type)     { mi_envelope_noise_list(dst, freqs, center, n, int(type), 1); }
    void white_noise_list(float *dst, const float *freqs, float center, size_t n, envelope_t)            { mi_envelope_noise_list(dst, freqs, center, n, MI_ENVELOPE_WHITE_NOISE, 0); }
    void pink_noise_list(float *dst, const float *freqs, float center, size_t n, envelope_t)             { mi_envelope_noise_list(dst, freqs, center, n, MI_ENVELOPE_PINK_NOISE, 0); }
    void brown_noise_list(float *dst, const float *freqs, float center, size_t n, envelope_t)            { mi_envelope_noise_list(dst, freqs, center, n, MI_ENVELOPE_BROWN_NOISE, 0); }
    void blue_noise_list(float *dst, const float *freqs, float center, size_t n, envelope_t)             { mi_envelope_noise_list(dst, freqs, center, n, MI_ENVELOPE_BLUE_NOISE, 0); }
    void violet_noise_list(float *dst, const float *freqs, float center, size_t n, envelope_t)           { mi_envelope_noise_list(dst, freqs, center, n, MI_ENVELOPE_VIOLET_NOISE, 0); }
}

// ---- crossover::* / SpectralSplitter / FFTCrossover ------------------------------------------------------------
namespace crossover
{
    float hipass(float f, float f0, float slope) { return mi_crossover_hipass(f, f0, slope); }
    float lopass(float f, float f0, float slope) { return mi_crossover_lopass(f, f0, slope); }
    void hipass_set(float *gain, const float *f, float f0, float slope, size_t count)   { mi_crossover_hipass_set(gain, f, f0, slope, count); }
    void hipass_apply(float *gain, const float *f, float f0, float slope, size_t count) { mi_crossover_hipass_apply(gain, f, f0, slope, count); }
    void lopass_set(float *gain, const float *f, float f0, float slope, size_t count)   { mi_crossover_lopass_set(gain, f, f0, slope, count); }
    void lopass_apply(float *gain, const float *f, float f0, float slope, size_t count) { mi_crossover_lopass_apply(gain, f, f0, slope, count); }
    void hipass_fft_set(float *mag, float f0, float slope, float sample_rate, size_t rank)   { mi_crossover_hipass_fft_set(mag, f0, slope, sample_rate, rank); }
    void hipass_fft_apply(float *mag, float f0, float slope, float sample_rate, size_t rank) { mi_crossover_hipass_fft_apply(mag, f0, slope, sample_rate, rank); }
    void lopass_fft_set(float *mag, float f0, float slope, float sample_rate, size_t rank)   { mi_crossover_lopass_fft_set(mag, f0, slope, sample_rate, rank); }
    void lopass_fft_apply(float *mag, float f0, float slope, float sample_rate, size_t rank) { mi_crossover_lopass_fft_apply(mag, f0, slope, sample_rate, rank); }
}

namespace
{
    // one channel of a splitter bank driven with host buffers: the pieces between two transforms go up, every listening
    // handler's samples come down, and `emit` hands them on (handlers in index order, as SpectralSplitter.cpp:344-352)
    struct splitter_stream
    {
        mi_splitter_bank_t *bank = nullptr;
        size_t  handlers = 0, cap = 0;
        float  *d_in = nullptr;
        std::vector<float *> d_out;
        std::vector<float>   host;

        bool init(size_t max_rank, size_t n_handlers)
        {
            if (mi_splitter_bank_create(&bank, 1, uint32_t(max_rank), uint32_t(n_handlers)) != MI_OK)
                return false;
            handlers = n_handlers;
            cap = size_t(1) << (max_rank - 1);
            d_out.assign(n_handlers, nullptr);
            host.resize(cap);
            bool ok = mi_dspu_malloc(reinterpret_cast<void **>(&d_in), cap * sizeof(float)) == MI_OK;
            for (size_t i = 0; ok && i < n_handlers; ++i)
                ok = mi_dspu_malloc(reinterpret_cast<void **>(&d_out[i]), cap * sizeof(float)) == MI_OK;
            return ok;
        }
        void release()
        {
            mi_splitter_bank_destroy(bank);
            bank = nullptr;
            mi_dspu_free(d_in);
            d_in = nullptr;
            for (float *p : d_out)
                mi_dspu_free(p);
            d_out.clear();
        }
        template <class LISTENS, class EMIT>
        void process(const float *src, size_t count, LISTENS listens, EMIT emit)
        {
            std::vector<float *> outs(handlers);
            for (size_t offset = 0; offset < count; )
            {
                uint32_t remaining = 0;
                if (mi_splitter_bank_get(bank, nullptr, nullptr, nullptr, &remaining) != MI_OK || remaining == 0)
                    return;
                const size_t n = std::min<size_t>(remaining, count - offset);
                for (size_t i = 0; i < handlers; ++i)
                    outs[i] = listens(i) ? d_out[i] : nullptr;
                if (src != nullptr && mi_dspu_copy_h2d(d_in, src + offset, n * sizeof(float), nullptr) != MI_OK)
                    return;
                if (mi_splitter_bank_process(bank, outs.data(), (src != nullptr) ? d_in : nullptr, n, n, n, nullptr) != MI_OK)
                    return;
                for (size_t i = 0; i < handlers; ++i)
                {
                    if (outs[i] == nullptr)
                        continue;
                    if (mi_dspu_copy_d2h(host.data(), outs[i], n * sizeof(float), nullptr) != MI_OK ||
                        mi_dspu_stream_synchronize(nullptr) != MI_OK)
                        return;
                    emit(i, host.data(), offset, n);
                }
                offset += n;
            }
        }
    };
}

struct SpectralSplitter::impl_t
{
    splitter_stream st;
    std::vector<handler_t> h;               // storage behind vHandlers
    std::vector<float> spec_in, spec_out;
    struct hook_t { SpectralSplitter *owner; size_t id; };
    std::vector<hook_t> hooks;
    std::vector<uint8_t> gains;             // handler shaped by device-side gains (bind_gains)

    // mi_splitter_func_t: the device spectrum comes down, the user's function runs on host memory, its result goes up
    static void trampoline(void *object, void *, float *out, const float *in, size_t rank, size_t channels, void *stream)
    {
        hook_t *k = static_cast<hook_t *>(object);
        impl_t *p = k->owner->impl();
        const handler_t &h = p->h[k->id];
        const size_t bytes = channels * (size_t(2) << rank) * sizeof(float);
        if (h.pFunc == nullptr || mi_dspu_copy_d2h(p->spec_in.data(), in, bytes, stream) != MI_OK || mi_dspu_stream_synchronize(stream) != MI_OK)
            return;
        h.pFunc(h.pObject, h.pSubject, p->spec_out.data(), p->spec_in.data(), rank);
        if (mi_dspu_copy_h2d(out, p->spec_out.data(), bytes, stream) == MI_OK)
            mi_dspu_stream_synchronize(stream);
    }
};

SpectralSplitter::SpectralSplitter() { construct(); }
SpectralSplitter::~SpectralSplitter() { destroy(); }

void SpectralSplitter::construct()                          // SpectralSplitter.cpp:33-55
{
    nRank = nMaxRank = 0;
    nUserChunkRank = 0;
    nChunkRank = 0;
    fPhase = 0.0f;
    vWnd = vInBuf = vFftBuf = vFftTmp = nullptr;
    nFrameSize = nInOffset = 0;
    bUpdate = true;
    vHandlers = nullptr;
    nHandlers = nBindings = 0;
    pData = nullptr;
}

void SpectralSplitter::destroy()
{
    if (impl_t *p = impl())
    {
        p->st.release();
        delete p;
    }
    pData = nullptr;
    vHandlers = nullptr;
    nHandlers = nBindings = 0;
}

void SpectralSplitter::sync_ranks()
{
    uint32_t r = 0, c = 0;
    if (impl_t *p = impl())
        mi_splitter_bank_get(p->st.bank, &r, &c, nullptr, nullptr);
    nRank = r;
    nChunkRank = c;
}

status_t SpectralSplitter::init(size_t max_rank, size_t handlers)
{
    if (max_rank < 5)                                       // SpectralSplitter.cpp:64-65
        return STATUS_INVALID_VALUE;
    destroy();
    impl_t *p = new (std::nothrow) impl_t();
    if (p == nullptr)
        return STATUS_NO_MEM;
    if (handlers == 0 || !p->st.init(max_rank, handlers))
    {
        p->st.release();
        delete p;
        return STATUS_NO_MEM;
    }
    p->h.assign(handlers, handler_t{ nullptr, nullptr, nullptr, nullptr, nullptr });
    p->hooks.resize(handlers);
    p->gains.assign(handlers, 0);
    for (size_t i = 0; i < handlers; ++i)
        p->hooks[i] = impl_t::hook_t{ this, i };
    p->spec_in.resize(size_t(2) << max_rank);
    p->spec_out.resize(size_t(2) << max_rank);
    pData = reinterpret_cast<uint8_t *>(p);
    vHandlers = p->h.data();
    nHandlers = handlers;
    nBindings = 0;
    nMaxRank = max_rank;
    nUserChunkRank = 0;
    fPhase = 0.0f;
    nFrameSize = nInOffset = 0;
    bUpdate = true;
    sync_ranks();
    return STATUS_OK;
}

status_t SpectralSplitter::bind(size_t id, void *object, void *subject, spectral_splitter_func_t func, spectral_splitter_sink_t sink)
{
    impl_t *p = impl();
    if (p == nullptr || id >= nHandlers)
        return STATUS_OVERFLOW;
    if (func == nullptr && sink == nullptr)
        return STATUS_INVALID_VALUE;
    handler_t &h = vHandlers[id];
    if (h.pFunc == nullptr && h.pSink == nullptr)
        ++nBindings;
    h.pObject = object;
    h.pSubject = subject;
    h.pFunc = func;
    h.pSink = sink;
    p->gains[id] = 0;
    const int r = (func != nullptr) ?
        mi_splitter_bank_bind_callback(p->st.bank, uint32_t(id), impl_t::trampoline, &p->hooks[id], nullptr, nullptr) :
        mi_splitter_bank_bind_copy(p->st.bank, uint32_t(id), nullptr);
    return (r == MI_OK) ? STATUS_OK : STATUS_NO_MEM;
}

status_t SpectralSplitter::unbind(size_t id)
{
    impl_t *p = impl();
    if (p == nullptr || id >= nHandlers)
        return STATUS_OVERFLOW;
    handler_t &h = vHandlers[id];
    if (h.pFunc == nullptr && h.pSink == nullptr)
        return STATUS_NOT_BOUND;
    h = handler_t{ nullptr, nullptr, nullptr, nullptr, nullptr };
    --nBindings;
    p->gains[id] = 0;
    mi_splitter_bank_unbind(p->st.bank, uint32_t(id));
    return STATUS_OK;
}

status_t SpectralSplitter::bind_gains(size_t id, void *object, void *subject, const float *gains, spectral_splitter_sink_t sink)
{
    impl_t *p = impl();
    if (p == nullptr || id >= nHandlers)
        return STATUS_OVERFLOW;
    if (gains == nullptr || sink == nullptr)
        return STATUS_INVALID_VALUE;
    handler_t &h = vHandlers[id];
    if (h.pFunc == nullptr && h.pSink == nullptr)
        ++nBindings;
    h.pObject = object;
    h.pSubject = subject;
    h.pFunc = nullptr;
    h.pSink = sink;
    p->gains[id] = 1;
    return (mi_splitter_bank_bind_mask(p->st.bank, uint32_t(id), gains, 0, nullptr) == MI_OK) ? STATUS_OK : STATUS_NO_MEM;
}

void SpectralSplitter::set_gains(size_t id, const float *gains)
{
    impl_t *p = impl();
    if (p != nullptr && id < nHandlers && p->gains[id])
        mi_splitter_bank_bind_mask(p->st.bank, uint32_t(id), gains, 0, nullptr);
}

void SpectralSplitter::unbind_all()
{
    for (size_t i = 0; impl() != nullptr && i < nHandlers; ++i)
        unbind(i);
}

bool SpectralSplitter::bound(size_t id) const
{
    return impl() != nullptr && id < nHandlers && (vHandlers[id].pFunc != nullptr || vHandlers[id].pSink != nullptr);
}

void SpectralSplitter::update_settings()
{
    impl_t *p = impl();
    if (p == nullptr || !bUpdate)
        return;
    mi_splitter_bank_process(p->st.bank, nullptr, nullptr, 0, 0, 0, nullptr);          // applies the pending settings
    sync_ranks();
    bUpdate = false;
}

void SpectralSplitter::set_phase(float phase)
{
    impl_t *p = impl();
    if (p == nullptr)
        return;
    fPhase = (phase < 0.0f) ? 0.0f : (phase > 1.0f) ? 1.0f : phase;
    bUpdate = true;
    mi_splitter_bank_set_phase(p->st.bank, phase);
}

void SpectralSplitter::set_rank(size_t rank)
{
    impl_t *p = impl();
    if (p == nullptr || rank == nRank || rank > nMaxRank)
        return;
    if (mi_splitter_bank_set_rank(p->st.bank, uint32_t(rank)) == MI_OK)
    {
        nRank = rank;
        bUpdate = true;
    }
}

void SpectralSplitter::set_chunk_rank(ssize_t rank)
{
    impl_t *p = impl();
    if (p == nullptr || rank == nUserChunkRank)
        return;
    nUserChunkRank = rank;
    bUpdate = true;
    mi_splitter_bank_set_chunk_rank(p->st.bank, int32_t(rank));
}

size_t SpectralSplitter::latency() const
{
    uint32_t v = 0;
    if (impl_t *p = impl())
        mi_splitter_bank_get(p->st.bank, nullptr, nullptr, &v, nullptr);
    return v;
}

void SpectralSplitter::process(const float *src, size_t count)
{
    impl_t *p = impl();
    if (p == nullptr)
        return;
    update_settings();
    if (nBindings == 0)
        return;
    SpectralSplitter *self = this;
    p->st.process(src, count,
        [self](size_t i) { return self->vHandlers[i].pSink != nullptr; },
        [self](size_t i, const float *data, size_t first, size_t n)
        { const handler_t &h = self->vHandlers[i]; h.pSink(h.pObject, h.pSubject, data, first, n); });
}

void SpectralSplitter::clear()
{
    if (impl_t *p = impl())
        mi_splitter_bank_clear(p->st.bank, nullptr);
}

void SpectralSplitter::dump(IStateDumper *v) const
{
    // SpectralSplitter.cpp:388-424
    v->write("nRank", nRank);
    v->write("nMaxRank", nMaxRank);
    v->write("nUserChunkRank", nUserChunkRank);
    v->write("nChunkRank", nChunkRank);
    v->write("fPhase", fPhase);
    v->write("vWnd", vWnd);
    v->write("vInBuf", vInBuf);
    v->write("vFftBuf", vFftBuf);
    v->write("vFftTmp", vFftTmp);
    v->write("nFrameSize", nFrameSize);
    v->write("nInOffset", nInOffset);
    v->begin_array("vHandlers", vHandlers, nHandlers);
    for (size_t i = 0; vHandlers != nullptr && i < nHandlers; ++i)
    {
        const handler_t *h = &vHandlers[i];
        v->begin_object(h, sizeof(handler_t));
        v->write("pObject", h->pObject);
        v->write("pSubject", h->pSubject);
        v->write("pFunc", reinterpret_cast<const void *>(h->pFunc));
        v->write("pSink", reinterpret_cast<const void *>(h->pSink));
        v->write("vOutBuf", h->vOutBuf);
        v->end_object();
    }
    v->end_array();
    v->write("nHandlers", nHandlers);
    v->write("nBindings", nBindings);
    v->write("pData", pData);
}

// The reference object: an embedded SpectralSplitter whose handlers are the bands, the band records and their gains in
// one block behind pData (FFTCrossover.cpp:40-121).  Here the bands' gains act on the device (SpectralSplitter::bind_gains).
FFTCrossover::FFTCrossover() { construct(); }
FFTCrossover::~FFTCrossover() { destroy(); }

void FFTCrossover::construct()
{
    sSplitter.construct();
    vBands = nullptr;
    nSampleRate = 0;
    pData = nullptr;
}

void FFTCrossover::destroy()
{
    sSplitter.destroy();
    free(pData);
    pData = nullptr;
    vBands = nullptr;
    nSampleRate = 0;
}

status_t FFTCrossover::init(size_t max_rank, size_t bands)
{
    const status_t res = sSplitter.init(max_rank, bands);
    if (res != STATUS_OK)
        return res;
    free(pData);
    pData = nullptr;
    vBands = nullptr;

    const size_t bins = size_t(1) << max_rank;
    const size_t szof_bands = (sizeof(band_t) * bands + 63) & ~size_t(63);
    uint8_t *ptr = static_cast<uint8_t *>(calloc(1, szof_bands + bands * bins * sizeof(float)));
    if (ptr == nullptr)
    {
        sSplitter.destroy();
        return STATUS_NO_MEM;
    }
    pData = ptr;
    vBands = reinterpret_cast<band_t *>(ptr);
    ptr += szof_bands;
    for (size_t i = 0; i < bands; ++i, ptr += bins * sizeof(float))
    {
        band_t *b = &vBands[i];
        b->fHpfFreq = 100.0f;
        b->fLpfFreq = 1000.0f;
        b->fHpfSlope = -24.0f;
        b->fLpfSlope = -24.0f;
        b->fGain = 1.0f;
        b->fFlatten = 1.0f;
        b->bLpf = b->bHpf = b->bEnabled = false;
        b->bUpdate = true;
        b->pObject = b->pSubject = nullptr;
        b->pFunc = nullptr;
        b->vFFT = reinterpret_cast<float *>(ptr);
    }
    return STATUS_OK;
}

void FFTCrossover::spectral_sink(void *object, void *subject, const float *samples, size_t first, size_t count)     // :142-153
{
    band_t *b = static_cast<band_t *>(subject);
    if (b->pFunc == nullptr)
        return;
    FFTCrossover *self = static_cast<FFTCrossover *>(object);
    b->pFunc(b->pObject, b->pSubject, b - self->vBands, samples, first, count);
}

// FFTCrossover::update_band (FFTCrossover.cpp:459-486); the gains go to the band's handler when it is bound
void FFTCrossover::update_band(band_t *b)
{
    if (!b->bUpdate)
        return;
    const size_t rk = sSplitter.rank(), bins = size_t(1) << rk;
    if (b->bHpf || b->bLpf)
    {
        if (b->bHpf)
        {
            mi_crossover_hipass_fft_set(b->vFFT, b->fHpfFreq, b->fHpfSlope, float(nSampleRate), rk);
            if (b->bLpf)
                mi_crossover_lopass_fft_apply(b->vFFT, b->fLpfFreq, b->fLpfSlope, float(nSampleRate), rk);
        }
        else
            mi_crossover_lopass_fft_set(b->vFFT, b->fLpfFreq, b->fLpfSlope, float(nSampleRate), rk);
        for (size_t i = 0; i < bins; ++i)                    // limit1(0, fFlatten), mul_k2(fGain)
        {
            const float g = b->vFFT[i];
            b->vFFT[i] = ((g < 0.0f) ? 0.0f : (g > b->fFlatten) ? b->fFlatten : g) * b->fGain;
        }
    }
    else
        std::fill(b->vFFT, b->vFFT + bins, b->fFlatten * b->fGain);
    sSplitter.set_gains(b - vBands, b->vFFT);                // no-op while the band is not bound
    b->bUpdate = false;
}

void FFTCrossover::sync_binding(size_t band, band_t *b)     // :355-364
{
    const bool bound = sSplitter.bound(band);
    if (b->bEnabled && b->pFunc != nullptr)
    {
        if (!bound)
        {
            b->bUpdate = true;                               // the handler needs its gains: the values the reference's
            update_band(b);                                  // spectral_func would find or rebuild at the next transform
            sSplitter.bind_gains(band, this, b, b->vFFT, spectral_sink);
        }
    }
    else if (bound)
        sSplitter.unbind(band);
}

void FFTCrossover::mark_bands_for_update()
{
    for (size_t i = 0, n = sSplitter.handlers(); i < n; ++i)
        vBands[i].bUpdate = true;
}

#define MI_BAND_OR(ret)                                         \
    if (band >= sSplitter.handlers())                           \
        return ret;                                             \
    band_t &x = vBands[band];

// the update-flag rules are the reference's, including the asymmetric ones (FFTCrossover.cpp:155-345)
void FFTCrossover::set_slope(size_t band, float lpf, float hpf)
{
    MI_BAND_OR()
    if (!x.bUpdate)
        x.bUpdate = (x.bLpf && x.fLpfSlope != lpf) || (x.bHpf && x.fHpfSlope != hpf);
    x.fLpfSlope = lpf;
    x.fHpfSlope = hpf;
}

void FFTCrossover::set_lpf_slope(size_t band, float slope)
{
    MI_BAND_OR()
    if (!x.bUpdate)
        x.bUpdate = x.bLpf && x.fLpfSlope != slope;
    x.fLpfSlope = slope;
}

void FFTCrossover::set_hpf_slope(size_t band, float slope)
{
    MI_BAND_OR()
    if (!x.bUpdate)
        x.bUpdate = x.bHpf && x.fHpfSlope != slope;
    x.fHpfSlope = slope;
}

void FFTCrossover::set_frequency(size_t band, float lpf, float hpf)
{
    MI_BAND_OR()
    if (!x.bUpdate)
        x.bUpdate = (x.bLpf && x.fLpfFreq != lpf) || (x.bHpf && x.fHpfFreq != hpf);
    x.fLpfFreq = lpf;
    x.fHpfFreq = hpf;
}

void FFTCrossover::set_lpf_frequency(size_t band, float freq)
{
    MI_BAND_OR()
    if (!x.bUpdate)
        x.bUpdate = x.bLpf && x.fLpfFreq != freq;
    x.fLpfFreq = freq;
}

void FFTCrossover::set_hpf_frequency(size_t band, float freq)
{
    MI_BAND_OR()
    if (!x.bUpdate)
        x.bUpdate = x.bHpf && x.fHpfFreq != freq;
    x.fHpfFreq = freq;
}

void FFTCrossover::enable_filters(size_t band, bool lpf, bool hpf)
{
    MI_BAND_OR()
    if (!x.bUpdate)
        x.bUpdate = (x.bLpf != lpf) || (x.bHpf != hpf);
    x.bLpf = lpf;
    x.bHpf = hpf;
}

void FFTCrossover::enable_lpf(size_t band, bool enable)
{
    MI_BAND_OR()
    if (!x.bUpdate)
        x.bUpdate = x.bLpf != enable;
    x.bLpf = enable;
}

void FFTCrossover::enable_hpf(size_t band, bool enable)
{
    MI_BAND_OR()
    if (!x.bUpdate)
        x.bUpdate = x.bHpf != enable;
    x.bHpf = enable;
}

void FFTCrossover::set_lpf(size_t band, float freq, float slope, bool enabled)
{
    MI_BAND_OR()
    if (!x.bUpdate)
        x.bUpdate = enabled && (x.fLpfFreq != freq || x.fLpfSlope != slope || x.bLpf != enabled);
    x.fLpfFreq = freq;
    x.fLpfSlope = slope;
    x.bLpf = enabled;
}

void FFTCrossover::set_hpf(size_t band, float freq, float slope, bool enabled)
{
    MI_BAND_OR()
    if (!x.bUpdate)
        x.bUpdate = enabled && (x.fHpfFreq != freq || x.fHpfSlope != slope || x.bHpf != enabled);
    x.fHpfFreq = freq;
    x.fHpfSlope = slope;
    x.bHpf = enabled;
}

void FFTCrossover::set_gain(size_t band, float gain)
{
    MI_BAND_OR()
    if (x.fGain == gain)
        return;
    x.bUpdate = true;
    x.fGain = gain;
}

void FFTCrossover::set_flatten(size_t band, float amount)
{
    MI_BAND_OR()
    if (x.fFlatten == amount)
        return;
    x.bUpdate = true;
    x.fFlatten = amount;
}

#undef MI_BAND_OR
#define MI_BAND_GET(field, fallback) ((band < sSplitter.handlers()) ? vBands[band].field : (fallback))
float FFTCrossover::lpf_slope(size_t band) const     { return MI_BAND_GET(fLpfSlope, -1.0f); }
float FFTCrossover::hpf_slope(size_t band) const     { return MI_BAND_GET(fHpfSlope, -1.0f); }
float FFTCrossover::lpf_frequency(size_t band) const { return MI_BAND_GET(fLpfFreq, -1.0f); }
float FFTCrossover::hpf_frequency(size_t band) const { return MI_BAND_GET(fHpfFreq, -1.0f); }
bool  FFTCrossover::lpf_enabled(size_t band) const   { return MI_BAND_GET(bLpf, false); }
bool  FFTCrossover::hpf_enabled(size_t band) const   { return MI_BAND_GET(bHpf, false); }
float FFTCrossover::gain(size_t band) const          { return MI_BAND_GET(fGain, -1.0f); }
float FFTCrossover::flatten(size_t band) const       { return MI_BAND_GET(fFlatten, -1.0f); }
bool  FFTCrossover::band_enabled(size_t band) const  { return MI_BAND_GET(bEnabled, false); }
#undef MI_BAND_GET

void FFTCrossover::enable_band(size_t band, bool enable)
{
    if (band >= sSplitter.handlers() || vBands[band].bEnabled == enable)
        return;
    vBands[band].bEnabled = enable;
    sync_binding(band, &vBands[band]);
}

bool FFTCrossover::set_handler(size_t band, crossover_func_t func, void *object, void *subject)
{
    if (band >= sSplitter.handlers())
        return false;
    band_t *b = &vBands[band];
    b->pFunc = func;
    b->pObject = object;
    b->pSubject = subject;
    sync_binding(band, b);
    return true;
}

bool FFTCrossover::unset_handler(size_t band) { return set_handler(band, nullptr, nullptr, nullptr); }

void FFTCrossover::set_sample_rate(size_t sr)
{
    if (nSampleRate == sr)
        return;
    nSampleRate = sr;
    mark_bands_for_update();
}

void FFTCrossover::set_rank(size_t rank)
{
    rank = std::min(rank, sSplitter.max_rank());
    if (sSplitter.rank() == rank)
        return;
    sSplitter.set_rank(rank);
    mark_bands_for_update();
}

void FFTCrossover::set_phase(float phase) { sSplitter.set_phase(phase); }

bool FFTCrossover::needs_update() const
{
    for (size_t i = 0, n = sSplitter.handlers(); i < n; ++i)
        if (vBands[i].bEnabled && vBands[i].bUpdate)
            return true;
    return false;
}

void FFTCrossover::update_settings()
{
    sSplitter.update_settings();
    for (size_t i = 0, n = sSplitter.handlers(); i < n; ++i)
        if (vBands[i].bEnabled)
            update_band(&vBands[i]);
}

void FFTCrossover::process(const float *in, size_t samples)
{
    // the reference refreshes a band's gains inside its spectral function, i.e. before the next transform uses them
    for (size_t i = 0, n = sSplitter.handlers(); i < n; ++i)
        if (sSplitter.bound(i))
            update_band(&vBands[i]);
    sSplitter.process(in, samples);
}

bool FFTCrossover::freq_chart(size_t band, float *m, const float *f, size_t count)      // :497-521
{
    if (band >= sSplitter.handlers())
        return false;
    const band_t &x = vBands[band];
    if (x.bHpf || x.bLpf)
    {
        if (x.bHpf)
        {
            mi_crossover_hipass_set(m, f, x.fHpfFreq, x.fHpfSlope, count);
            if (x.bLpf)
                mi_crossover_lopass_apply(m, f, x.fLpfFreq, x.fLpfSlope, count);
        }
        else
            mi_crossover_lopass_set(m, f, x.fLpfFreq, x.fLpfSlope, count);
        for (size_t i = 0; i < count; ++i)
            m[i] = ((m[i] < 0.0f) ? 0.0f : (m[i] > x.fFlatten) ? x.fFlatten : m[i]) * x.fGain;
    }
    else
        std::fill(m, m + count, x.fFlatten * x.fGain);
    return true;
}

void FFTCrossover::dump(IStateDumper *v) const
{
    // FFTCrossover.cpp:524-559
    v->write_object("sSplitter", &sSplitter);
    const size_t n = sSplitter.handlers();
    v->begin_array("vBands", vBands, n);
    for (size_t i = 0; vBands != nullptr && i < n; ++i)
    {
        const band_t *b = &vBands[i];
        v->begin_object(b, sizeof(band_t));
        v->write("fHpfFreq", b->fHpfFreq);
        v->write("fLpfFreq", b->fLpfFreq);
        v->write("fHpfSlope", b->fHpfSlope);
        v->write("fLpfSlope", b->fLpfSlope);
        v->write("fGain", b->fGain);
        v->write("fFlatten", b->fFlatten);
        v->write("bLpf", b->bLpf);
        v->write("bHpf", b->bHpf);
        v->write("bEnabled", b->bEnabled);
        v->write("bUpdate", b->bUpdate);
        v->write("pObject", b->pObject);
        v->write("pSubject", b->pSubject);
        v->write("pFunc", reinterpret_cast<const void *>(b->pFunc));
        v->write("vFFT", b->vFFT);
        v->end_object();
    }
    v->end_array();
    v->write("nSampleRate", nSampleRate);
    v->write("pData", pData);
}

// ---- bs::channel_weighting / LoudnessMeter ---------------------------------------------------------------------
namespace bs
{
    float channel_weighting(channel_t designation)           // src/main/misc/broadcast.cpp:32-55
    {
        if (designation >= CHANNEL_FRONT_LEFT && designation <= CHANNEL_RIGHT_SURROUND)
            return 1.41f;
        if (designation == CHANNEL_LFE1 || designation == CHANNEL_LFE2)
            return 0.0f;
        return 1.0f;
    }
}

struct LoudnessMeter::impl_t
{
    mi_loudness_bank_t *bank = nullptr;
    struct chan_t { const float *in = nullptr; float *out = nullptr; size_t offset = 0; float link = 1.0f;
                    bs::channel_t designation = bs::CHANNEL_NONE; bool active = true; };
    std::vector<chan_t> ch;
    std::vector<float>  host;
    float  *d_in = nullptr, *d_out = nullptr, *d_ch = nullptr;
    size_t  cap = 0;

    bool reserve(size_t channels, size_t n)
    {
        if (n <= cap)
            return true;
        mi_dspu_free(d_in); mi_dspu_free(d_out); mi_dspu_free(d_ch);
        d_in = d_out = d_ch = nullptr;
        cap = 0;
        if (mi_dspu_malloc(reinterpret_cast<void **>(&d_in), channels * n * sizeof(float)) != MI_OK) return false;
        if (mi_dspu_malloc(reinterpret_cast<void **>(&d_ch), channels * n * sizeof(float)) != MI_OK) return false;
        if (mi_dspu_malloc(reinterpret_cast<void **>(&d_out), n * sizeof(float)) != MI_OK) return false;
        cap = n;
        return true;
    }
};

LoudnessMeter::LoudnessMeter() { construct(); }
LoudnessMeter::~LoudnessMeter() { destroy(); }

void LoudnessMeter::construct()                             // LoudnessMeter.cpp:37-63
{
    vChannels = nullptr;
    vBuffer = nullptr;
    fPeriod = 0.0f;
    fMaxPeriod = 0.0f;
    fAvgCoeff = 1.0f;
    fLoudness = 0.0f;
    nSampleRate = 0;
    nPeriod = 0;
    nMSRefresh = 0;
    nChannels = 0;
    nFlags = F_UPD_ALL;
    nDataHead = 0;
    nDataSize = 0;
    enWeight = bs::WEIGHT_K;
    pData = nullptr;
    pVarData = nullptr;
}

void LoudnessMeter::destroy()
{
    if (impl_t *p = impl())
    {
        mi_loudness_bank_destroy(p->bank);
        mi_dspu_free(p->d_in); mi_dspu_free(p->d_out); mi_dspu_free(p->d_ch);
        delete p;
    }
    pData = nullptr;
}

status_t LoudnessMeter::init(size_t channels, float max_period)
{
    destroy();
    impl_t *p = new (std::nothrow) impl_t();
    if (p == nullptr)
        return STATUS_NO_MEM;
    if (channels == 0 || mi_loudness_bank_create(&p->bank, 1, uint32_t(channels), max_period) != MI_OK)
    {
        delete p;
        return STATUS_NO_MEM;
    }
    p->ch.resize(channels);
    if (channels == 1)
        p->ch[0].designation = bs::CHANNEL_CENTER;
    else if (channels == 2)
    {
        p->ch[0].designation = bs::CHANNEL_LEFT;
        p->ch[1].designation = bs::CHANNEL_RIGHT;
    }
    pData = reinterpret_cast<uint8_t *>(p);
    nChannels = channels;                                   // LoudnessMeter.cpp:160-177
    fMaxPeriod = max_period;
    fPeriod = std::min(max_period, bs::LUFS_MEASURE_PERIOD_MS);
    fAvgCoeff = 1.0f;
    fLoudness = 0.0f;
    nSampleRate = 0;
    nPeriod = 0;
    nMSRefresh = 0;
    enWeight = bs::WEIGHT_K;
    nFlags = F_UPD_ALL;
    nDataHead = 0;
    nDataSize = 0;
    return STATUS_OK;
}

status_t LoudnessMeter::bind(size_t id, float *out, const float *in, size_t pos)
{
    impl_t *p = impl();
    if (p == nullptr || id >= nChannels)
        return STATUS_OVERFLOW;
    p->ch[id].in = in;
    p->ch[id].out = out;
    p->ch[id].offset = pos;
    return STATUS_OK;
}

status_t LoudnessMeter::set_designation(size_t id, bs::channel_t designation)
{
    impl_t *p = impl();
    if (p == nullptr || id >= nChannels)
        return STATUS_OVERFLOW;
    p->ch[id].designation = designation;
    mi_loudness_bank_set_designation(p->bank, uint32_t(id), int(designation));
    return STATUS_OK;
}

bs::channel_t LoudnessMeter::designation(size_t id) const
{
    impl_t *p = impl();
    return (p && id < nChannels) ? p->ch[id].designation : bs::CHANNEL_NONE;
}

status_t LoudnessMeter::set_link(size_t id, float link)
{
    impl_t *p = impl();
    if (p == nullptr || id >= nChannels)
        return STATUS_OVERFLOW;
    p->ch[id].link = std::min(std::max(link, 0.0f), 1.0f);
    mi_loudness_bank_set_link(p->bank, uint32_t(id), link);
    return STATUS_OK;
}

float LoudnessMeter::link(size_t id) const
{
    impl_t *p = impl();
    return (p && id < nChannels) ? p->ch[id].link : 0.0f;
}

status_t LoudnessMeter::set_active(size_t id, bool active)
{
    impl_t *p = impl();
    if (p == nullptr || id >= nChannels)
        return STATUS_OVERFLOW;
    p->ch[id].active = active;
    mi_loudness_bank_set_active(p->bank, uint32_t(id), active ? 1 : 0, nullptr);
    return STATUS_OK;
}

bool LoudnessMeter::active(size_t id) const
{
    impl_t *p = impl();
    return (p && id < nChannels) ? p->ch[id].active : false;
}

void LoudnessMeter::set_weighting(bs::weighting_t weighting)     // LoudnessMeter.cpp:199-206
{
    impl_t *p = impl();
    if (p == nullptr || enWeight == weighting)
        return;
    enWeight = weighting;
    nFlags |= F_UPD_FILTERS;
    mi_loudness_bank_set_weighting(p->bank, int(weighting));
}

void LoudnessMeter::set_period(float period)                     // :208-216
{
    impl_t *p = impl();
    if (p == nullptr)
        return;
    period = std::min(std::max(period, 0.0f), fMaxPeriod);
    if (fPeriod == period)
        return;
    fPeriod = period;
    nFlags |= F_UPD_TIME;
    mi_loudness_bank_set_period(p->bank, period);
}

void LoudnessMeter::update_settings()
{
    impl_t *p = impl();
    if (p == nullptr || nFlags == 0)
        return;
    mi_loudness_bank_update_settings(p->bank, nullptr);
    nFlags = 0;
}

status_t LoudnessMeter::set_sample_rate(size_t sample_rate)
{
    impl_t *p = impl();
    if (p == nullptr)
        return STATUS_BAD_STATE;
    if (nSampleRate == sample_rate)
        return STATUS_OK;
    if (mi_loudness_bank_set_sample_rate(p->bank, uint32_t(sample_rate), nullptr) != MI_OK)
        return STATUS_NO_MEM;
    nSampleRate = sample_rate;
    nFlags |= F_UPD_ALL;
    return STATUS_OK;
}

size_t LoudnessMeter::latency() const
{
    uint32_t v = 0;
    if (impl_t *p = impl())
        mi_loudness_bank_latency(p->bank, &v);
    return v;
}

void LoudnessMeter::process(float *out, size_t count)             { run(out, count, 1.0f, false); }
void LoudnessMeter::process(float *out, size_t count, float gain) { run(out, count, gain, true); }

// with_gain: the reference's second form, which does not record fLoudness (LoudnessMeter.cpp:518-564)
void LoudnessMeter::run(float *out, size_t count, float gain, bool with_gain)
{
    impl_t *p = impl();
    const size_t K = nChannels;
    if (p == nullptr || count == 0 || !p->reserve(K, count))
        return;
    p->host.assign(K * count, 0.0f);
    bool want_ch = false;
    for (size_t c = 0; c < K; ++c)
    {
        // a channel without an input is left out of the block (LoudnessMeter.cpp:421-422)
        mi_loudness_bank_set_bound(p->bank, uint32_t(c), p->ch[c].in != nullptr);
        if (p->ch[c].in != nullptr)
            std::memcpy(&p->host[c * count], p->ch[c].in, count * sizeof(float));      // vIn is read from its start every call (:424)
        want_ch = want_ch || (p->ch[c].out != nullptr);
    }
    bool ok = mi_dspu_copy_h2d(p->d_in, p->host.data(), K * count * sizeof(float), nullptr) == MI_OK &&
              (with_gain ? mi_loudness_bank_process_gain(p->bank, p->d_out, want_ch ? p->d_ch : nullptr, p->d_in, count, count, count, gain, nullptr)
                         : mi_loudness_bank_process(p->bank, p->d_out, want_ch ? p->d_ch : nullptr, p->d_in, count, count, count, nullptr)) == MI_OK;
    if (ok && out != nullptr)
        ok = mi_dspu_copy_d2h(out, p->d_out, count * sizeof(float), nullptr) == MI_OK;
    if (ok && want_ch)
        ok = mi_dspu_copy_d2h(p->host.data(), p->d_ch, K * count * sizeof(float), nullptr) == MI_OK;
    ok = ok && mi_dspu_stream_synchronize(nullptr) == MI_OK;
    for (size_t c = 0; c < K; ++c)
    {
        if (ok && p->ch[c].out != nullptr && p->ch[c].active && p->ch[c].in != nullptr)
            std::memcpy(p->ch[c].out + p->ch[c].offset, &p->host[c * count], count * sizeof(float));
        if (p->ch[c].active)
            p->ch[c].offset += count;                       // LoudnessMeter.cpp:499
    }
    if (ok && !with_gain)
        mi_loudness_bank_loudness(p->bank, &fLoudness, nullptr);
    nFlags = 0;                                             // process() starts with update_settings() (:471-472)
}

void LoudnessMeter::clear()
{
    impl_t *p = impl();
    if (p == nullptr)
        return;
    fLoudness = 0.0f;
    mi_loudness_bank_clear(p->bank, nullptr);
}

void LoudnessMeter::dump(IStateDumper *v) const
{
    // LoudnessMeter.cpp:566-615.  The channel records (weighting filter, square line, running sum) live in the device bank:
    // the bindings and settings this object keeps go out under the reference's keys, the per-channel FilterBank / Filter
    // objects and lines, which do not exist on this side, as null.
    impl_t *p = impl();
    const size_t n = (p != nullptr) ? p->ch.size() : 0;
    v->begin_array("vChannels", vChannels, n);
    for (size_t i = 0; i < n; ++i)
    {
        const impl_t::chan_t &c = p->ch[i];
        v->begin_object(&c, sizeof(c));
        v->write("sBank", static_cast<const void *>(nullptr));
        v->write("sFilter", static_cast<const void *>(nullptr));
        v->write("vIn", c.in);
        v->write("vOut", c.out);
        v->write("vData", static_cast<const void *>(nullptr));
        v->write("vMS", static_cast<const void *>(nullptr));
        v->write("fMS", 0.0f);
        v->write("fWeight", bs::channel_weighting(c.designation));
        v->write("fLink", c.link);
        v->write("enDesignation", int(c.designation));
        v->write("nFlags", size_t(c.active ? 1 : 0));         // C_ENABLED
        v->write("nOffset", c.offset);
        v->end_object();
    }
    v->end_array();
    v->write("vBuffer", vBuffer);
    v->write("fPeriod", fPeriod);
    v->write("fMaxPeriod", fMaxPeriod);
    v->write("fAvgCoeff", fAvgCoeff);
    v->write("fLoudness", fLoudness);
    v->write("nSampleRate", nSampleRate);
    v->write("nPeriod", nPeriod);
    v->write("nMSRefresh", nMSRefresh);
    v->write("nChannels", nChannels);
    v->write("nFlags", nFlags);
    v->write("nDataHead", nDataHead);
    v->write("nDataSize", nDataSize);
    v->write("enWeight", int(enWeight));
    v->write("pData", pData);
    v->write("pVarData", pVarData);
}

// ---- ILUFSMeter ------------------------------------------------------------------------------------------------
struct ILUFSMeter::impl_t
{
    mi_ilufs_bank_t *bank = nullptr;
    struct chan_t { const float *in = nullptr; bs::channel_t designation = bs::CHANNEL_NONE; bool active = true, bound = true; };
    std::vector<chan_t> ch;
    std::vector<float>  host;
    float  *d_in = nullptr, *d_out = nullptr;
    size_t  cap = 0;

    bool reserve(size_t channels, size_t n)
    {
        if (n <= cap)
            return true;
        mi_dspu_free(d_in); mi_dspu_free(d_out);
        d_in = d_out = nullptr;
        cap = 0;
        if (mi_dspu_malloc(reinterpret_cast<void **>(&d_in), channels * n * sizeof(float)) != MI_OK) return false;
        if (mi_dspu_malloc(reinterpret_cast<void **>(&d_out), n * sizeof(float)) != MI_OK) return false;
        cap = n;
        return true;
    }
};

ILUFSMeter::ILUFSMeter() { construct(); }
ILUFSMeter::~ILUFSMeter() { destroy(); }

void ILUFSMeter::construct()                                // ILUFSMeter.cpp:57-85
{
    vChannels = nullptr;
    vBuffer = vLoudness = nullptr;
    fBlockPeriod = 0.0f;
    fIntTime = 0.0f;
    fMaxIntTime = 0.0f;
    fAvgCoeff = 1.0f;
    fLoudness = 0.0f;
    nBlockSize = nBlockOffset = nBlockPart = 0;
    nMSSize = nMSHead = nMSInt = nMSCount = 0;
    nSampleRate = 0;
    nChannels = 0;
    nFlags = F_UPD_ALL;
    enWeight = bs::WEIGHT_K;
    pData = nullptr;
    pVarData = nullptr;
}

void ILUFSMeter::destroy()
{
    if (impl_t *p = impl())
    {
        mi_ilufs_bank_destroy(p->bank);
        mi_dspu_free(p->d_in); mi_dspu_free(p->d_out);
        delete p;
    }
    pData = nullptr;
}

status_t ILUFSMeter::init(size_t channels, float max_int_time, float block_period)
{
    destroy();
    impl_t *p = new (std::nothrow) impl_t();
    if (p == nullptr)
        return STATUS_NO_MEM;
    if (channels == 0 || mi_ilufs_bank_create(&p->bank, 1, uint32_t(channels), max_int_time, block_period) != MI_OK)
    {
        delete p;
        return STATUS_NO_MEM;
    }
    p->ch.resize(channels);
    if (channels == 1)
        p->ch[0].designation = bs::CHANNEL_CENTER;
    else if (channels == 2)
    {
        p->ch[0].designation = bs::CHANNEL_LEFT;
        p->ch[1].designation = bs::CHANNEL_RIGHT;
    }
    pData = reinterpret_cast<uint8_t *>(p);
    nChannels = uint32_t(channels);                         // ILUFSMeter.cpp:183-205
    fBlockPeriod = block_period;
    fIntTime = fMaxIntTime = max_int_time;
    fAvgCoeff = 1.0f;
    fLoudness = 0.0f;
    nBlockSize = nBlockOffset = nBlockPart = 0;
    nMSSize = nMSHead = nMSInt = nMSCount = 0;
    nSampleRate = 0;
    enWeight = bs::WEIGHT_K;
    nFlags = F_UPD_ALL;
    return STATUS_OK;
}

status_t ILUFSMeter::bind(size_t id, const float *in)
{
    impl_t *p = impl();
    if (p == nullptr || id >= nChannels)
        return STATUS_OVERFLOW;
    p->ch[id].in = in;
    return STATUS_OK;
}

status_t ILUFSMeter::set_designation(size_t id, bs::channel_t designation)
{
    impl_t *p = impl();
    if (p == nullptr || id >= nChannels)
        return STATUS_OVERFLOW;
    p->ch[id].designation = designation;
    mi_ilufs_bank_set_designation(p->bank, uint32_t(id), int(designation));
    return STATUS_OK;
}

bs::channel_t ILUFSMeter::designation(size_t id) const
{
    impl_t *p = impl();
    return (p && id < nChannels) ? p->ch[id].designation : bs::CHANNEL_NONE;
}

status_t ILUFSMeter::set_active(size_t id, bool active)
{
    impl_t *p = impl();
    if (p == nullptr || id >= nChannels)
        return STATUS_OVERFLOW;
    p->ch[id].active = active;
    return STATUS_OK;
}

bool ILUFSMeter::active(size_t id) const
{
    impl_t *p = impl();
    return (p && id < nChannels) ? p->ch[id].active : false;
}

void ILUFSMeter::set_weighting(bs::weighting_t weighting)       // ILUFSMeter.cpp:255-262
{
    impl_t *p = impl();
    if (p == nullptr || enWeight == weighting)
        return;
    enWeight = weighting;
    nFlags |= F_UPD_FILTERS;
    mi_ilufs_bank_set_weighting(p->bank, int(weighting));
}

void ILUFSMeter::update_settings()
{
    impl_t *p = impl();
    if (p == nullptr || nFlags == 0)
        return;
    mi_ilufs_bank_update_settings(p->bank, nullptr);
    nFlags = 0;
}

void ILUFSMeter::set_integration_period(float period)           // :264-289
{
    impl_t *p = impl();
    if (p == nullptr)
        return;
    const float lo = fBlockPeriod * 0.001f;
    period = (period < lo) ? lo : (period > fMaxIntTime) ? fMaxIntTime : period;       // lsp_limit, :266
    if (fIntTime == period)
        return;
    fIntTime = period;
    nFlags |= F_UPD_TIME;
    mi_ilufs_bank_set_integration_period(p->bank, period, nullptr);
}

status_t ILUFSMeter::set_sample_rate(size_t sample_rate)
{
    impl_t *p = impl();
    if (p == nullptr)
        return STATUS_BAD_STATE;
    if (nSampleRate == sample_rate)
        return STATUS_OK;
    if (mi_ilufs_bank_set_sample_rate(p->bank, uint32_t(sample_rate), nullptr) != MI_OK)
        return STATUS_NO_MEM;
    fLoudness = 0.0f;
    nSampleRate = uint32_t(sample_rate);
    nFlags |= F_UPD_ALL;
    return STATUS_OK;
}

void ILUFSMeter::process(float *out, size_t count, float gain)
{
    impl_t *p = impl();
    const size_t K = nChannels;
    if (p == nullptr || count == 0 || !p->reserve(K, count))
        return;
    p->host.assign(K * count, 0.0f);
    for (size_t c = 0; c < K; ++c)
    {
        // a channel takes part when it is bound AND enabled (ILUFSMeter.cpp:370)
        const bool on = p->ch[c].active && p->ch[c].in != nullptr;
        if (on != p->ch[c].bound)
        {
            mi_ilufs_bank_set_active(p->bank, uint32_t(c), on ? 1 : 0);
            p->ch[c].bound = on;
        }
        if (on)
            std::memcpy(&p->host[c * count], p->ch[c].in, count * sizeof(float));
    }
    bool ok = mi_dspu_copy_h2d(p->d_in, p->host.data(), K * count * sizeof(float), nullptr) == MI_OK &&
              mi_ilufs_bank_process(p->bank, (out != nullptr) ? p->d_out : nullptr, p->d_in, count, count, count, gain, nullptr) == MI_OK;
    if (ok && out != nullptr)
        ok = mi_dspu_copy_d2h(out, p->d_out, count * sizeof(float), nullptr) == MI_OK;
    ok = ok && mi_dspu_stream_synchronize(nullptr) == MI_OK;
    if (ok)
        mi_ilufs_bank_loudness(p->bank, &fLoudness, nullptr);
    nFlags = 0;                                             // process() starts with update_settings() (:357-358)
}

void ILUFSMeter::clear()
{
    impl_t *p = impl();
    if (p == nullptr)
        return;
    fLoudness = 0.0f;
    mi_ilufs_bank_clear(p->bank, nullptr);
}

void ILUFSMeter::dump(IStateDumper *v) const
{
    // ILUFSMeter.cpp:552-602 (channel records: see LoudnessMeter::dump)
    impl_t *p = impl();
    const size_t n = (p != nullptr) ? p->ch.size() : 0;
    v->begin_array("vChannels", vChannels, n);
    for (size_t i = 0; i < n; ++i)
    {
        const impl_t::chan_t &c = p->ch[i];
        const float block[4] = { 0.0f, 0.0f, 0.0f, 0.0f };     // (the quarter sums of the open block stay on the device)
        v->begin_object(&c, sizeof(c));
        v->write("sBank", static_cast<const void *>(nullptr));
        v->write("sFilter", static_cast<const void *>(nullptr));
        v->write("vIn", c.in);
        v->writev("vBlock", block, 4);
        v->write("fWeight", bs::channel_weighting(c.designation));
        v->write("enDesignation", int(c.designation));
        v->write("nFlags", size_t(c.active ? 1 : 0));         // C_ENABLED
        v->end_object();
    }
    v->end_array();
    v->write("vBuffer", vBuffer);
    v->write("vLoudness", vLoudness);
    v->write("fBlockPeriod", fBlockPeriod);
    v->write("fIntTime", fIntTime);
    v->write("fMaxIntTime", fMaxIntTime);
    v->write("fAvgCoeff", fAvgCoeff);
    v->write("fLoudness", fLoudness);
    v->write("nBlockSize", nBlockSize);
    v->write("nBlockOffset", nBlockOffset);
    v->write("nBlockPart", nBlockPart);
    v->write("nMSSize", nMSSize);
    v->write("nMSHead", nMSHead);
    v->write("nMSInt", nMSInt);
    v->write("nMSCount", nMSCount);
    v->write("nSampleRate", nSampleRate);
    v->write("nChannels", nChannels);
    v->write("nFlags", nFlags);
    v->write("enWeight", int(enWeight));
    v->write("pData", pData);
    v->write("pVarData", pVarData);
}

// ---- Delay -----------------------------------------------------------------------------------------------------
struct Delay::impl_t
{
    mi_delay_bank_t *bank = nullptr;
    staging st;
    float  *d_gain = nullptr;
    size_t  gain_cap = 0;

    bool gain_up(const float *g, size_t n)
    {
        if (n > gain_cap)
        {
            mi_dspu_free(d_gain);
            d_gain = nullptr;
            gain_cap = 0;
            if (mi_dspu_malloc(reinterpret_cast<void **>(&d_gain), n * sizeof(float)) != MI_OK)
                return false;
            gain_cap = n;
        }
        return mi_dspu_copy_h2d(d_gain, g, n * sizeof(float), nullptr) == MI_OK;
    }

    // common body of process / process_add with the three gain forms
    void run(float *dst, const float *src, size_t count, int add, int gmode, float gain, const float *gvec)
    {
        if (count == 0 || !st.reserve(count) || !st.up(src, count))
            return;
        if (add && mi_dspu_copy_h2d(st.d_out, dst, count * sizeof(float), nullptr) != MI_OK)
            return;
        if (gmode == MI_GAIN_VECTOR && !gain_up(gvec, count))
            return;
        if (mi_delay_bank_process(bank, st.d_out, st.d_in, count, count, count, add, gmode, gain, d_gain, count, nullptr) == MI_OK)
            st.down(dst, count);
    }

    void ramp(float *dst, const float *src, size_t delay, size_t count, int gmode, float gain, const float *gvec)
    {
        if (count == 0 || !st.reserve(count) || !st.up(src, count))
            return;
        if (gmode == MI_GAIN_VECTOR && !gain_up(gvec, count))
            return;
        const uint32_t nd = uint32_t(delay);
        if (mi_delay_bank_process_ramping(bank, st.d_out, st.d_in, &nd, count, count, count, gmode, gain, d_gain, count, nullptr) == MI_OK)
            st.down(dst, count);
    }
};

static_assert(sizeof(Delay) == 24, "Delay keeps the reference's layout");

Delay::Delay() { construct(); }
Delay::~Delay() { destroy(); }

void Delay::construct()                                     // Delay.cpp:42-49
{
    pBuffer = nullptr;
    nHead = nTail = nDelay = nSize = 0;
}

void Delay::destroy()
{
    if (impl_t *p = impl())
    {
        mi_delay_bank_destroy(p->bank);
        mi_dspu_free(p->d_gain);
        p->st.release();
        delete p;
    }
    construct();
}

// nHead / nTail / nDelay / nSize as the device line has them now (the reference's own absolute positions)
void Delay::sync_positions()
{
    if (impl() != nullptr)
        mi_delay_bank_get(impl()->bank, 0, &nDelay, &nSize, &nHead, &nTail);
}

bool Delay::init(size_t max_size)
{
    destroy();
    impl_t *p = new (std::nothrow) impl_t();
    if (p == nullptr)
        return false;
    if (mi_delay_bank_create(&p->bank, 1, max_size) != MI_OK)
    {
        delete p;
        return false;
    }
    pBuffer = reinterpret_cast<float *>(p);
    sync_positions();
    return true;
}

void Delay::append(const float *src, size_t count)
{
    impl_t *p = impl();
    if (p && count && p->st.reserve(count) && p->st.up(src, count))
    {
        last_status(mi_delay_bank_append(p->bank, p->st.d_in, count, count, nullptr));
        mi_dspu_stream_synchronize(nullptr);
        sync_positions();
    }
}

void Delay::process(float *dst, const float *src, size_t count)                     { if (impl()) { impl()->run(dst, src, count, 0, MI_GAIN_NONE, 0.0f, nullptr); sync_positions(); } }
void Delay::process(float *dst, const float *src, float gain, size_t count)         { if (impl()) { impl()->run(dst, src, count, 0, MI_GAIN_SCALAR, gain, nullptr); sync_positions(); } }
void Delay::process(float *dst, const float *src, const float *gain, size_t count)  { if (impl()) { impl()->run(dst, src, count, 0, MI_GAIN_VECTOR, 0.0f, gain); sync_positions(); } }
void Delay::process_add(float *dst, const float *src, size_t count)                 { if (impl()) { impl()->run(dst, src, count, 1, MI_GAIN_NONE, 0.0f, nullptr); sync_positions(); } }
void Delay::process_add(float *dst, const float *src, float gain, size_t count)     { if (impl()) { impl()->run(dst, src, count, 1, MI_GAIN_SCALAR, gain, nullptr); sync_positions(); } }
void Delay::process_add(float *dst, const float *src, const float *gain, size_t count) { if (impl()) { impl()->run(dst, src, count, 1, MI_GAIN_VECTOR, 0.0f, gain); sync_positions(); } }
void Delay::process_ramping(float *dst, const float *src, size_t delay, size_t count) { if (impl()) { impl()->ramp(dst, src, delay, count, MI_GAIN_NONE, 0.0f, nullptr); sync_positions(); } }
void Delay::process_ramping(float *dst, const float *src, float gain, size_t delay, size_t count) { if (impl()) { impl()->ramp(dst, src, delay, count, MI_GAIN_SCALAR, gain, nullptr); sync_positions(); } }
void Delay::process_ramping(float *dst, const float *src, const float *gain, size_t delay, size_t count) { if (impl()) { impl()->ramp(dst, src, delay, count, MI_GAIN_VECTOR, 0.0f, gain); sync_positions(); } }

float Delay::process(float src)
{
    float out = 0.0f;
    process(&out, &src, 1);
    return out;
}

float Delay::process(float src, float gain)
{
    float out = 0.0f;
    process(&out, &src, gain, 1);
    return out;
}

void Delay::set_delay(size_t delay)         { if (impl()) { mi_delay_bank_set_delay(impl()->bank, 0, delay); sync_positions(); } }
void Delay::clear()                         { if (impl()) { mi_delay_bank_clear(impl()->bank, nullptr); sync_positions(); } }

void Delay::dump(IStateDumper *v) const
{
    // Delay.cpp:581-588
    v->write("pBuffer", pBuffer);
    v->write("nHead", nHead);
    v->write("nTail", nTail);
    v->write("nDelay", nDelay);
    v->write("nSize", nSize);
}

// ---- RingBuffer --------------------------------------------------------------------------------------------------
// Reference members (util/RingBuffer.h:38-40): pData IS the raw storage (pinned host memory the device addresses
// through the same pointer), so there is no room for a handle in the object: the device bank that owns a storage block
// is looked up by that block's address.
struct RingBuffer::impl_t
{
    mi_ring_bank_t *bank = nullptr;
    size_t  capacity = 0;
    staging st;
};

namespace
{
    std::mutex                                   g_ring_lock;
    std::unordered_map<const float *, void *>    g_ring_owner;      // storage address -> RingBuffer::impl_t
}
static_assert(sizeof(RingBuffer) == 16, "RingBuffer keeps the reference's layout");

RingBuffer::impl_t *RingBuffer::impl() const
{
    if (pData == nullptr)
        return nullptr;
    std::lock_guard<std::mutex> g(g_ring_lock);
    auto it = g_ring_owner.find(pData);
    return (it != g_ring_owner.end()) ? static_cast<impl_t *>(it->second) : nullptr;
}

void RingBuffer::sync_head()
{
    uint32_t h = 0;
    if (impl_t *p = impl())
        mi_ring_bank_info(p->bank, 0, nullptr, &h, nullptr);
    nHead = h;
}

RingBuffer::RingBuffer() { construct(); }
RingBuffer::~RingBuffer() { destroy(); }

void RingBuffer::construct()                                // RingBuffer.cpp:41-46
{
    pData = nullptr;
    nCapacity = 0;
    nHead = 0;
}

bool RingBuffer::init(size_t size, float fill)
{
    if (impl_t *q = impl())
        if (q->capacity == size)
        {
            mi_ring_bank_fill(q->bank, fill, nullptr);      // note: the reference keeps nHead here; fill() resets it
            mi_dspu_stream_synchronize(nullptr);
            sync_head();
            return true;
        }
    destroy();
    impl_t *p = new (std::nothrow) impl_t();
    if (p == nullptr)
        return false;
    float *view = nullptr;
    if (last_status(mi_ring_bank_create_shared(&p->bank, 1, size, fill, &view)) != MI_OK)
    {
        delete p;
        return false;
    }
    p->capacity = size;
    {
        std::lock_guard<std::mutex> g(g_ring_lock);
        g_ring_owner[view] = p;
    }
    pData = view;
    nCapacity = uint32_t(size);
    nHead = 0;
    return true;
}

void RingBuffer::destroy()
{
    if (impl_t *p = impl())
    {
        {
            std::lock_guard<std::mutex> g(g_ring_lock);
            g_ring_owner.erase(pData);
        }
        mi_ring_bank_destroy(p->bank);
        p->st.release();
        delete p;
    }
    construct();
}

size_t RingBuffer::append(const float *data, size_t count)
{
    size_t n = 0;
    impl_t *p = impl();
    if (p && count && p->st.reserve(count) && p->st.up(data, count))
    {
        last_status(mi_ring_bank_append(p->bank, p->st.d_in, count, count, &n, nullptr));
        mi_dspu_stream_synchronize(nullptr);
        sync_head();
    }
    return n;
}

void RingBuffer::append(float data)         { append(&data, 1); }
void RingBuffer::fill(float value)
{
    if (impl_t *p = impl())
    {
        mi_ring_bank_fill(p->bank, value, nullptr);
        mi_dspu_stream_synchronize(nullptr);                // the host may look at data() right away
        sync_head();
    }
}

void RingBuffer::clear()                    { fill(0.0f); }

size_t RingBuffer::get(float *dst, size_t offset, size_t count) const
{
    size_t n = 0;
    impl_t *p = impl();
    if (p == nullptr || count == 0 || !p->st.reserve(count))
        return 0;
    if (last_status(mi_ring_bank_get(p->bank, p->st.d_out, offset, count, count, &n, nullptr)) != MI_OK || !p->st.down(dst, count))
        return 0;
    return n;
}

float RingBuffer::get(size_t offset) const
{
    float v = 0.0f;
    get(&v, offset, 1);
    return v;
}

float RingBuffer::lerp_get(float offset) const            // RingBuffer.cpp:131-138
{
    const ssize_t off = ssize_t(offset);
    const float s1 = get(size_t(off)), s2 = get(size_t(off + 1));
    return s1 + (s2 - s1) * (offset - float(off));          // lerp(s1, s2, k) of units.h
}

// Raw positions: the sample `offset` behind the newest sits at tail_position(offset), so position p is the sample
// (head - 1 - p) mod size behind the newest.
float RingBuffer::read(size_t position) const
{
    const size_t cap = size();
    if (position >= cap)
        return 0.0f;
    return get((head_position() + cap - 1 - position) % cap);
}

size_t RingBuffer::read(float *dst, size_t position, size_t count) const
{
    const size_t cap = size();
    if (position >= cap)
        return 0;
    // the whole storage, oldest sample first == raw order rotated by head
    std::vector<float> chrono(cap), raw(cap);
    if (get(chrono.data(), cap - 1, cap) == 0 && cap > 0)
        return 0;
    const size_t head = head_position();
    for (size_t j = 0; j < cap; ++j)
        raw[(head + j) % cap] = chrono[j];
    // the reference's loop as it is, including its never-advancing dst (RingBuffer.cpp:185-205)
    size_t to_copy = std::min(cap - position, count);
    std::memcpy(dst, &raw[position], to_copy * sizeof(float));
    position += to_copy;
    while (position < count)
    {
        to_copy = std::min(cap, count);
        std::memcpy(dst, raw.data(), to_copy * sizeof(float));
        position += to_copy;
    }
    return count;
}

size_t RingBuffer::tail_position(size_t offset) const
{
    uint32_t t = 0;
    if (impl_t *p = impl())
        mi_ring_bank_info(p->bank, offset, nullptr, nullptr, &t);
    return t;
}

void RingBuffer::dump(IStateDumper *v) const
{
    // RingBuffer.cpp:211-216
    v->write("pData", pData);
    v->write("nCapacity", nCapacity);
    v->write("nHead", nHead);
}

// ---- Analyzer ----------------------------------------------------------------------------------------------------
struct Analyzer::impl_t
{
    mi_analyzer_bank_t *bank = nullptr;
    std::vector<channel_t> ch;              // storage behind vChannels
    bool    pushed_active = true;           // bActive as the bank last saw it (set_activity() is inline: it only writes the member)
    float  *d_in = nullptr;
    size_t  in_cap = 0;
    float  *d_out = nullptr;
    uint32_t *d_idx = nullptr;
    size_t  q_cap = 0;
};

Analyzer::Analyzer() { construct(); }
Analyzer::~Analyzer() { destroy(); }

void Analyzer::construct()                                  // Analyzer.cpp:33-66
{
    nChannels = nMaxRank = nRank = nSampleRate = nMaxSampleRate = nBufSize = nCounter = nPeriod = nStep = nHead = 0;
    nReconfigure = 0;
    nEnvelope = envelope::PINK_NOISE;
    nWindow = windows::HANN;
    nMaxUserDelay = 0;
    fReactivity = 0.0f;
    fTau = 1.0f;
    fRate = 1.0f;
    fMinRate = 1.0f;
    fShift = 1.0f;
    bActive = true;
    vChannels = nullptr;
    vData = nullptr;
    vSigRe = vFftReIm = vWindow = vEnvelope = nullptr;
}

void Analyzer::destroy()
{
    if (impl_t *p = impl())
    {
        mi_analyzer_bank_destroy(p->bank);
        mi_dspu_free(p->d_in); mi_dspu_free(p->d_out); mi_dspu_free(p->d_idx);
        delete p;
    }
    vData = nullptr;
    vChannels = nullptr;
}

bool Analyzer::init(size_t channels, size_t max_rank, size_t max_sr, float min_rate, size_t max_delay)
{
    destroy();
    impl_t *p = new (std::nothrow) impl_t();
    if (p == nullptr)
        return false;
    if (mi_analyzer_bank_create(&p->bank, uint32_t(channels), uint32_t(max_rank), uint32_t(max_sr), min_rate, uint32_t(max_delay)) != MI_OK)
    {
        delete p;
        return false;
    }
    p->ch.assign(channels, channel_t{ nullptr, nullptr, nullptr, 0, 0, false, true });
    vData = p;
    vChannels = p->ch.data();
    nChannels = uint32_t(channels);                         // Analyzer.cpp:113-131
    nMaxRank = nRank = uint32_t(max_rank);
    nSampleRate = 0;
    nMaxSampleRate = uint32_t(max_sr);
    nMaxUserDelay = uint32_t(max_delay);
    fMinRate = float(uint32_t(min_rate));                   // fMinRate = uint32_t(min_rate), :120
    const size_t fft = size_t(1) << max_rank;
    nBufSize = uint32_t((fft + size_t(float(max_sr * 2) / min_rate) + max_delay + 0x40 + 0x3f) & ~size_t(0x3f));
    nCounter = nPeriod = nStep = nHead = 0;
    nEnvelope = envelope::PINK_NOISE;
    nWindow = windows::HANN;
    fReactivity = 0.0f;
    fTau = 1.0f;
    fRate = 1.0f;
    fShift = 1.0f;
    bActive = true;
    nReconfigure = R_ALL;
    return true;
}

// The setters keep the members the inline getters report and the flags of needs_reconfiguration(), with the reference's
// own no-change tests (Analyzer.cpp:154-250); the bank applies them at the next process() / reconfigure().
void Analyzer::set_sample_rate(size_t sr)
{
    impl_t *p = impl();
    if (p == nullptr)
        return;
    sr = std::min(sr, size_t(nMaxSampleRate));
    if (nSampleRate == sr)
        return;
    nSampleRate = uint32_t(sr);
    nReconfigure |= R_ALL;
    mi_analyzer_bank_configure(p->bank, MI_ANALYZER_SAMPLE_RATE, double(sr));
}

void Analyzer::set_rate(float rate)
{
    impl_t *p = impl();
    if (p == nullptr)
        return;
    rate = std::max(fMinRate, rate);
    if (fRate == rate)
        return;
    fRate = rate;
    nReconfigure |= R_COUNTERS;
    mi_analyzer_bank_configure(p->bank, MI_ANALYZER_RATE, rate);
}

void Analyzer::set_window(size_t window)
{
    impl_t *p = impl();
    if (p == nullptr || nWindow == window)
        return;
    nWindow = uint32_t(window);
    nReconfigure |= R_WINDOW;
    mi_analyzer_bank_configure(p->bank, MI_ANALYZER_WINDOW, double(window));
}

void Analyzer::set_envelope(size_t env)
{
    impl_t *p = impl();
    if (p == nullptr || nEnvelope == env)
        return;
    nEnvelope = uint32_t(env);
    nReconfigure |= R_ENVELOPE;
    mi_analyzer_bank_configure(p->bank, MI_ANALYZER_ENVELOPE, double(env));
}

void Analyzer::set_shift(float shift)
{
    impl_t *p = impl();
    if (p == nullptr || fShift == shift)
        return;
    fShift = shift;
    nReconfigure |= R_ENVELOPE;
    mi_analyzer_bank_configure(p->bank, MI_ANALYZER_SHIFT, shift);
}

void Analyzer::set_reactivity(float r)
{
    impl_t *p = impl();
    if (p == nullptr || fReactivity == r)
        return;
    fReactivity = r;
    nReconfigure |= R_TAU;
    mi_analyzer_bank_configure(p->bank, MI_ANALYZER_REACTIVITY, r);
}

// set_activity() and reset() are inline in the reference header: a caller compiled against it only writes bActive /
// nReconfigure.  Whatever changed that way reaches the bank here, before it is used.
void Analyzer::sync_inline_state()
{
    impl_t *p = impl();
    if (p == nullptr)
        return;
    if (p->pushed_active != bActive)
    {
        mi_analyzer_bank_configure(p->bank, MI_ANALYZER_ACTIVE, bActive ? 1.0 : 0.0);
        p->pushed_active = bActive;
    }
    if (nReconfigure & R_ANALYSIS)
        mi_analyzer_bank_reset(p->bank);                    // (R_ALL from the setters includes it: the bank has it set already)
}

void Analyzer::reconfigure()
{
    impl_t *p = impl();
    if (p == nullptr || !nReconfigure)
        return;
    sync_inline_state();
    mi_analyzer_bank_process(p->bank, nullptr, 0, 0, nullptr);      // applies the pending settings, consumes no samples
    uint32_t period = 0, step = 0;
    mi_analyzer_bank_info(p->bank, nullptr, nullptr, &period, &step);
    nPeriod = period;
    nStep = step;
    for (uint32_t i = 0; i < nChannels; ++i)
        vChannels[i].nDelay = i * nStep;                    // Analyzer.cpp:289-293
    nReconfigure = 0;
}

bool Analyzer::set_rank(size_t rank)
{
    impl_t *p = impl();
    if (p == nullptr || rank < 2 || rank > nMaxRank)
        return false;
    if (nRank == rank)
        return true;
    if (mi_analyzer_bank_configure(p->bank, MI_ANALYZER_RANK, double(rank)) != MI_OK)
        return false;
    nRank = uint32_t(rank);
    nReconfigure |= R_ALL;
    return true;
}

bool Analyzer::freeze_channel(size_t ch, bool freeze)
{
    impl_t *p = impl();
    if (p == nullptr || ch >= nChannels || mi_analyzer_bank_channel(p->bank, uint32_t(ch), MI_ANALYZER_CH_FREEZE, freeze) != MI_OK)
        return false;
    vChannels[ch].bFreeze = freeze;
    return true;
}

bool Analyzer::enable_channel(size_t ch, bool enable)
{
    impl_t *p = impl();
    if (p == nullptr || ch >= nChannels || vChannels[ch].bActive == enable)      // no change answers false (:232-235)
        return false;
    if (mi_analyzer_bank_channel(p->bank, uint32_t(ch), MI_ANALYZER_CH_ENABLE, enable) != MI_OK)
        return false;
    vChannels[ch].bActive = enable;
    nReconfigure |= R_COUNTERS;
    return true;
}

bool Analyzer::set_channel_delay(size_t ch, size_t d)
{
    impl_t *p = impl();
    if (p == nullptr || ch >= nChannels || d > nMaxUserDelay)
        return false;
    if (mi_analyzer_bank_channel(p->bank, uint32_t(ch), MI_ANALYZER_CH_DELAY, uint32_t(d)) != MI_OK)
        return false;
    vChannels[ch].nUserDelay = uint32_t(d);
    return true;
}

bool Analyzer::read_frequencies(float *frq, float start, float stop, size_t count, size_t flags)     // Analyzer.cpp:411-441
{
    if (impl() == nullptr || count == 0)
        return false;
    if (count == 1)
    {
        *frq = start;
        return true;
    }
    if (flags == FRQA_SCALE_LOGARITHMIC)
    {
        const float norm = logf(stop / start) / (--count);
        for (size_t i = 0; i < count; ++i)
            frq[i] = start * expf(i * norm);
    }
    else if (flags == FRQA_SCALE_LINEAR)
    {
        const float norm = (stop - start) / (--count);
        for (size_t i = 0; i < count; ++i)
            frq[i] = start + i * norm;
    }
    else
        return false;
    frq[count] = stop;
    return true;
}

void Analyzer::process(const float * const *in, size_t samples)
{
    impl_t *pImpl = impl();
    if (pImpl == nullptr || samples == 0)
        return;
    sync_inline_state();
    const size_t need = size_t(nChannels) * samples;
    if (need > pImpl->in_cap)
    {
        mi_dspu_free(pImpl->d_in);
        pImpl->d_in = nullptr;
        pImpl->in_cap = 0;
        if (mi_dspu_malloc(reinterpret_cast<void **>(&pImpl->d_in), need * sizeof(float)) != MI_OK)
            return;
        pImpl->in_cap = need;
    }
    for (size_t c = 0; c < nChannels; ++c)
    {
        if (in != nullptr && in[c] != nullptr)
            mi_dspu_copy_h2d(pImpl->d_in + c * samples, in[c], samples * sizeof(float), nullptr);
        else
            mi_dspu_memset(pImpl->d_in + c * samples, 0, samples * sizeof(float), nullptr);
    }
    mi_analyzer_bank_process(pImpl->bank, pImpl->d_in, samples, samples, nullptr);
    mi_dspu_stream_synchronize(nullptr);
    if (nReconfigure)                                       // process() reconfigures first (Analyzer.cpp:303)
    {
        uint32_t period = 0, step = 0;
        mi_analyzer_bank_info(pImpl->bank, nullptr, nullptr, &period, &step);
        nPeriod = period;
        nStep = step;
        for (uint32_t i = 0; i < nChannels; ++i)
            vChannels[i].nDelay = i * nStep;
        nReconfigure = 0;
    }
}

bool Analyzer::get_spectrum(size_t channel, float *out, const uint32_t *idx, size_t count)
{
    impl_t *pImpl = impl();
    if (pImpl == nullptr || channel >= nChannels || count == 0)
        return false;
    if (count > pImpl->q_cap)
    {
        mi_dspu_free(pImpl->d_out); mi_dspu_free(pImpl->d_idx);
        pImpl->d_out = nullptr; pImpl->d_idx = nullptr; pImpl->q_cap = 0;
        if (mi_dspu_malloc(reinterpret_cast<void **>(&pImpl->d_out), size_t(nChannels) * count * sizeof(float)) != MI_OK ||
            mi_dspu_malloc(reinterpret_cast<void **>(&pImpl->d_idx), count * sizeof(uint32_t)) != MI_OK)
            return false;
        pImpl->q_cap = count;
    }
    if (mi_dspu_copy_h2d(pImpl->d_idx, idx, count * sizeof(uint32_t), nullptr) != MI_OK ||
        mi_analyzer_bank_get_spectrum(pImpl->bank, pImpl->d_out, count, pImpl->d_idx, uint32_t(count), nullptr) != MI_OK ||
        mi_dspu_copy_d2h(out, pImpl->d_out + channel * count, count * sizeof(float), nullptr) != MI_OK)
        return false;
    return mi_dspu_stream_synchronize(nullptr) == MI_OK;
}

float Analyzer::get_level(size_t channel, const uint32_t idx)
{
    float v = 0.0f;
    return get_spectrum(channel, &v, &idx, 1) ? v : 0.0f;
}

void Analyzer::get_frequencies(float *frq, uint32_t *idx, float start, float stop, size_t count, bool linear)
{
    if (impl() == nullptr)
        return;
    const size_t fft_size = size_t(1) << nRank, fft_width = fft_size >> 1;
    const float scale = float(fft_size) / float(nSampleRate);
    const float norm = linear ? (stop - start) / (count - 1) : logf(stop / start) / (count - 1);
    for (size_t i = 0; i < count; ++i)
    {
        const float f = linear ? start + i * norm : start * expf(i * norm);
        frq[i] = f;
        const float pos = scale * f;
        idx[i] = uint32_t((pos < float(fft_width)) ? pos : float(fft_width));
    }
}

void Analyzer::dump(IStateDumper *v) const
{
    // Analyzer.cpp:496-544
    v->write("nChannels", nChannels);
    v->write("nMaxRank", nMaxRank);
    v->write("nRank", nRank);
    v->write("nSampleRate", nSampleRate);
    v->write("nMaxSampleRate", nMaxSampleRate);
    v->write("nBufSize", nBufSize);
    v->write("nCounter", nCounter);
    v->write("nPeriod", nPeriod);
    v->write("nStep", nStep);
    v->write("nHead", nHead);
    v->write("nReconfigure", nReconfigure);
    v->write("nEnvelope", nEnvelope);
    v->write("nWindow", nWindow);
    v->write("nMaxUserDelay", nMaxUserDelay);
    v->write("fReactivity", fReactivity);
    v->write("fTau", fTau);
    v->write("fRate", fRate);
    v->write("fMinRate", fMinRate);
    v->write("fShift", fShift);
    v->write("bActive", bActive);
    v->begin_array("vChannels", vChannels, nChannels);
    for (size_t i = 0; vChannels != nullptr && i < nChannels; ++i)
    {
        const channel_t *c = &vChannels[i];
        v->begin_object(c, sizeof(channel_t));
        v->write("vBuffer", c->vBuffer);
        v->write("vAmp", c->vAmp);
        v->write("vData", c->vData);
        v->write("nDelay", c->nDelay);
        v->write("nUserDelay", c->nUserDelay);
        v->write("bFreeze", c->bFreeze);
        v->write("bActive", c->bActive);
        v->end_object();
    }
    v->end_array();
    v->write("vData", vData);
    v->write("vSigRe", vSigRe);
    v->write("vFftReIm", vFftReIm);
    v->write("vWindow", vWindow);
    v->write("vEnvelope", vEnvelope);
}

// ---- FilterArray (extension: the batched mode under the class API) ------------------------------------------------------
// One device bank with one channel per Filter object; the objects' designs are kept on the host and sent to the bank
// lazily, as Filter::process() rebuilds lazily (Filter.cpp:698-720).
namespace
{
    struct filter_array
    {
        mi_biquad_bank_t               *bank = nullptr;
        size_t                          max_chains = 0;
        std::vector<filter_params_t>    params;
        std::vector<uint32_t>           rate;
        std::vector<uint8_t>            rebuild, wipe;
        bool                            pending = false;
        staging                         st;
    };
    inline filter_array *fa_of(void *p) { return static_cast<filter_array *>(p); }
}

FilterArray::FilterArray() { construct(); }
FilterArray::~FilterArray() { destroy(); }
void FilterArray::construct() { pImpl = nullptr; }

bool FilterArray::init(size_t filters, size_t max_chains)
{
    destroy();
    if (filters == 0 || max_chains == 0 || max_chains > size_t(mi::CHAINS_MAX))
        return false;
    filter_array *a = new (std::nothrow) filter_array();
    if (a == nullptr)
        return false;
    if (last_status(mi_biquad_bank_create(&a->bank, uint32_t(filters), uint32_t(max_chains))) != MI_OK)
    {
        delete a;
        return false;
    }
    a->max_chains = max_chains;
    filter_params_t none;
    none.nType = FLT_NONE; none.nSlope = 1; none.fFreq = none.fFreq2 = none.fGain = none.fQuality = 0.0f;
    a->params.assign(filters, none);
    a->rate.assign(filters, 0);
    a->rebuild.assign(filters, 0);
    a->wipe.assign(filters, 0);
    pImpl = a;
    return true;
}

void FilterArray::destroy()
{
    filter_array *a = fa_of(pImpl);
    if (a != nullptr)
    {
        a->st.release();
        mi_biquad_bank_destroy(a->bank);
        delete a;
    }
    pImpl = nullptr;
}

size_t FilterArray::size() const { return (pImpl != nullptr) ? fa_of(pImpl)->params.size() : 0; }

bool FilterArray::update(size_t id, size_t sr, const filter_params_t *params)
{
    filter_array *a = fa_of(pImpl);
    if (a == nullptr || id >= a->params.size() || params == nullptr || sr == 0)
        return false;
    // the design is checked now (so that an over-long one can be refused) and sent with the next process()
    uint32_t n = 0;
    if (last_status(mi_filter_design(cfp(params), uint32_t(sr), nullptr, 0, &n, nullptr, 0, nullptr, nullptr)) != MI_OK || n > a->max_chains)
        return false;
    a->params[id]  = *params;
    mi_filter_limit(cfp(&a->params[id]), uint32_t(sr));         // Filter::update stores the limited parameters (Filter.cpp:147-150)
    a->rate[id]    = uint32_t(sr);
    a->rebuild[id] = 1;
    a->pending     = true;
    return true;
}

bool FilterArray::get_params(size_t id, filter_params_t *params) const
{
    const filter_array *a = fa_of(pImpl);
    if (a == nullptr || id >= a->params.size() || params == nullptr)
        return false;
    *params = a->params[id];
    return true;
}

void FilterArray::clear(size_t id)
{
    filter_array *a = fa_of(pImpl);
    if (a == nullptr)
        return;
    for (size_t i = 0; i < a->params.size(); ++i)
        if (id == size_t(-1) || id == i)
            a->wipe[i] = 1;
    a->pending = true;
}

namespace
{
    bool fa_commit(filter_array *a, void *stream)
    {
        if (!a->pending)
            return true;
        std::vector<mi_biquad_x1_t> sec(a->max_chains);
        for (size_t i = 0; i < a->params.size(); ++i)
        {
            if (a->rebuild[i])
            {
                uint32_t n = 0;
                int mode = MI_FM_BYPASS;
                if (last_status(mi_filter_design(cfp(&a->params[i]), a->rate[i], sec.data(), uint32_t(sec.size()), &n, nullptr, 0,
                                                 nullptr, &mode)) != MI_OK)
                    return false;
                if (mode == MI_FM_BYPASS)
                    n = 0;                                      // Filter::process copies (Filter.cpp:712-717): a bank row without sections
                // FilterBank::end(clear): the memory goes when asked for or when the section count changed (FilterBank.cpp:233-235)
                if (last_status(mi_biquad_bank_set_chains(a->bank, uint32_t(i), sec.data(), n, a->wipe[i] ? 1 : 0)) != MI_OK)
                    return false;
                a->rebuild[i] = a->wipe[i] = 0;
            }
            else if (a->wipe[i])
            {
                if (last_status(mi_biquad_bank_reset(a->bank, uint32_t(i), stream)) != MI_OK)
                    return false;
                a->wipe[i] = 0;
            }
        }
        a->pending = false;
        return last_status(mi_biquad_bank_commit(a->bank, stream)) == MI_OK;
    }
}

bool FilterArray::process(float *dev_out, const float *dev_in, size_t samples, size_t stride, void *stream)
{
    filter_array *a = fa_of(pImpl);
    if (a == nullptr || dev_out == nullptr || dev_in == nullptr || stride < samples)
        return false;
    if (samples == 0)
        return true;
    return fa_commit(a, stream) &&
           last_status(mi_biquad_bank_process(a->bank, dev_out, dev_in, samples, stride, stride, stream)) == MI_OK;
}

bool FilterArray::process_blocks(float *const *dev_out, const float *const *dev_in, size_t blocks, size_t samples, size_t stride, void *stream)
{
    filter_array *a = fa_of(pImpl);
    if (a == nullptr || dev_out == nullptr || dev_in == nullptr || stride < samples)
        return false;
    if (samples == 0 || blocks == 0)
        return true;
    return fa_commit(a, stream) &&
           last_status(mi_biquad_bank_process_blocks(a->bank, dev_out, dev_in, blocks, samples, stride, stride, stream)) == MI_OK;
}

bool FilterArray::process_host(float *out, const float *in, size_t samples, size_t stride)
{
    filter_array *a = fa_of(pImpl);
    if (a == nullptr || out == nullptr || in == nullptr || stride < samples)
        return false;
    if (samples == 0)
        return true;
    const size_t n = a->params.size() * stride;
    return a->st.reserve(n) && a->st.up(in, n) && process(a->st.d_out, a->st.d_in, samples, stride, nullptr) && a->st.down(out, n);
}

// ---- EqualizerArray / ConvolverArray (extensions: the batched mode under the class API, like FilterArray) -----------------
// Thin: the device banks (mi_equalizer_bank_*, mi_convolver_bank_*) already hold one object per channel and do the lazy
// reconfiguration of the reference inside process(); the arrays only give them the classes' vocabulary.
namespace
{
    struct equalizer_array { mi_equalizer_bank_t *bank = nullptr; size_t count = 0; staging st; };
    struct convolver_array { mi_convolver_bank_t *bank = nullptr; size_t count = 0; staging st; };
    inline equalizer_array *ea_of(void *p) { return static_cast<equalizer_array *>(p); }
    inline convolver_array *ca_of(void *p) { return static_cast<convolver_array *>(p); }
}

EqualizerArray::EqualizerArray() { construct(); }
EqualizerArray::~EqualizerArray() { destroy(); }
void EqualizerArray::construct() { pImpl = nullptr; }

bool EqualizerArray::init(size_t equalizers, size_t filters, size_t fir_rank)
{
    destroy();
    if (equalizers == 0)
        return false;
    equalizer_array *a = new (std::nothrow) equalizer_array();
    if (a == nullptr)
        return false;
    if (last_status(mi_equalizer_bank_create(&a->bank, uint32_t(equalizers), uint32_t(filters), uint32_t(fir_rank))) != MI_OK)
    {
        delete a;
        return false;
    }
    a->count = equalizers;
    pImpl = a;
    return true;
}

void EqualizerArray::destroy()
{
    equalizer_array *a = ea_of(pImpl);
    if (a != nullptr)
    {
        a->st.release();
        mi_equalizer_bank_destroy(a->bank);
        delete a;
    }
    pImpl = nullptr;
}

size_t EqualizerArray::size() const { return (pImpl != nullptr) ? ea_of(pImpl)->count : 0; }

bool EqualizerArray::set_params(size_t eq, size_t id, const filter_params_t *params)
{
    equalizer_array *a = ea_of(pImpl);
    if (a == nullptr || params == nullptr || (eq != size_t(-1) && eq >= a->count))
        return false;
    return last_status(mi_equalizer_bank_set_params(a->bank, (eq == size_t(-1)) ? UINT32_MAX : uint32_t(eq), uint32_t(id), cfp(params))) == MI_OK;
}

bool EqualizerArray::get_params(size_t eq, size_t id, filter_params_t *params) const
{
    const equalizer_array *a = ea_of(pImpl);
    if (a == nullptr || params == nullptr || eq >= a->count)
        return false;
    return last_status(mi_equalizer_bank_get_params(a->bank, uint32_t(eq), uint32_t(id), cfp(params))) == MI_OK;
}

void EqualizerArray::set_mode(equalizer_mode_t mode)    { if (pImpl != nullptr) last_status(mi_equalizer_bank_set_mode(ea_of(pImpl)->bank, int(mode))); }
void EqualizerArray::set_sample_rate(size_t sr)         { if (pImpl != nullptr) last_status(mi_equalizer_bank_set_sample_rate(ea_of(pImpl)->bank, uint32_t(sr))); }
void EqualizerArray::set_smooth(bool smooth)            { if (pImpl != nullptr) last_status(mi_equalizer_bank_set_smooth(ea_of(pImpl)->bank, smooth ? 1 : 0)); }
void EqualizerArray::reset(void *stream)                { if (pImpl != nullptr) last_status(mi_equalizer_bank_reset(ea_of(pImpl)->bank, stream)); }

size_t EqualizerArray::get_latency(void *stream)
{
    uint32_t lat = 0;
    if (pImpl == nullptr || last_status(mi_equalizer_bank_get_latency(ea_of(pImpl)->bank, &lat, stream)) != MI_OK)
        return 0;
    return lat;
}

bool EqualizerArray::process(float *dev_out, const float *dev_in, size_t samples, size_t stride, void *stream)
{
    equalizer_array *a = ea_of(pImpl);
    if (a == nullptr || dev_out == nullptr || dev_in == nullptr || stride < samples)
        return false;
    return samples == 0 || last_status(mi_equalizer_bank_process(a->bank, dev_out, dev_in, samples, stride, stride, stream)) == MI_OK;
}

bool EqualizerArray::process_blocks(float *const *dev_out, const float *const *dev_in, size_t blocks, size_t samples, size_t stride, void *stream)
{
    equalizer_array *a = ea_of(pImpl);
    if (a == nullptr || dev_out == nullptr || dev_in == nullptr || stride < samples)
        return false;
    return samples == 0 || blocks == 0 ||
           last_status(mi_equalizer_bank_process_blocks(a->bank, dev_out, dev_in, blocks, samples, stride, stride, stream)) == MI_OK;
}

bool EqualizerArray::process_host(float *out, const float *in, size_t samples, size_t stride)
{
    equalizer_array *a = ea_of(pImpl);
    if (a == nullptr || out == nullptr || in == nullptr || stride < samples)
        return false;
    if (samples == 0)
        return true;
    const size_t n = a->count * stride;
    return a->st.reserve(n) && a->st.up(in, n) && process(a->st.d_out, a->st.d_in, samples, stride, nullptr) && a->st.down(out, n);
}

ConvolverArray::ConvolverArray() { construct(); }
ConvolverArray::~ConvolverArray() { destroy(); }
void ConvolverArray::construct() { pImpl = nullptr; }

bool ConvolverArray::init(size_t convolvers, const float *irs, size_t ir_stride, size_t count, size_t rank, float phase, const size_t *counts)
{
    destroy();
    if (convolvers == 0 || irs == nullptr || count == 0 || ir_stride < count)          // Convolver::init: no data, no object (Convolver.cpp:79-84)
        return false;
    convolver_array *a = new (std::nothrow) convolver_array();
    if (a == nullptr)
        return false;
    std::vector<uint32_t> cn;
    if (counts != nullptr)
    {
        cn.resize(convolvers);
        for (size_t c = 0; c < convolvers; ++c)
            cn[c] = uint32_t((counts[c] < count) ? counts[c] : count);
    }
    if (last_status(mi_convolver_bank_create(&a->bank, uint32_t(convolvers), irs, ir_stride, counts ? cn.data() : nullptr, uint32_t(count),
                                             uint32_t(rank), phase, nullptr)) != MI_OK)
    {
        delete a;
        return false;
    }
    a->count = convolvers;
    pImpl = a;
    return true;
}

void ConvolverArray::destroy()
{
    convolver_array *a = ca_of(pImpl);
    if (a != nullptr)
    {
        a->st.release();
        mi_convolver_bank_destroy(a->bank);
        delete a;
    }
    pImpl = nullptr;
}

size_t ConvolverArray::size() const { return (pImpl != nullptr) ? ca_of(pImpl)->count : 0; }

size_t ConvolverArray::rank() const
{
    uint32_t r = 0;
    return (pImpl != nullptr && mi_convolver_bank_info(ca_of(pImpl)->bank, &r, nullptr, nullptr, nullptr) == MI_OK) ? r : 0;
}

size_t ConvolverArray::data_size() const
{
    uint32_t n = 0;
    return (pImpl != nullptr && mi_convolver_bank_info(ca_of(pImpl)->bank, nullptr, nullptr, nullptr, &n) == MI_OK) ? n : 0;
}

void ConvolverArray::reset(void *stream) { if (pImpl != nullptr) last_status(mi_convolver_bank_reset(ca_of(pImpl)->bank, stream)); }

bool ConvolverArray::process(float *dev_out, const float *dev_in, size_t samples, size_t stride, void *stream)
{
    convolver_array *a = ca_of(pImpl);
    if (a == nullptr || dev_out == nullptr || dev_in == nullptr || stride < samples)
        return false;
    return samples == 0 || last_status(mi_convolver_bank_process(a->bank, dev_out, dev_in, samples, stride, stride, stream)) == MI_OK;
}

bool ConvolverArray::process_blocks(float *const *dev_out, const float *const *dev_in, size_t blocks, size_t samples, size_t stride, void *stream)
{
    convolver_array *a = ca_of(pImpl);
    if (a == nullptr || dev_out == nullptr || dev_in == nullptr || stride < samples)
        return false;
    return samples == 0 || blocks == 0 ||
           last_status(mi_convolver_bank_process_blocks(a->bank, dev_out, dev_in, blocks, samples, stride, stride, stream)) == MI_OK;
}

bool ConvolverArray::process_host(float *out, const float *in, size_t samples, size_t stride)
{
    convolver_array *a = ca_of(pImpl);
    if (a == nullptr || out == nullptr || in == nullptr || stride < samples)
        return false;
    if (samples == 0)
        return true;
    const size_t n = a->count * stride;
    return a->st.reserve(n) && a->st.up(in, n) && process(a->st.d_out, a->st.d_in, samples, stride, nullptr) && a->st.down(out, n);
}

} // namespace dspu
} // namespace lsp
