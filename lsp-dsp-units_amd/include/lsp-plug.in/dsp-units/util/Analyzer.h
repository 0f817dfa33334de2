// lsp::dspu::Analyzer on the GPU library (host pointers per channel; device-resident: mi_analyzer_bank_*).
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_UTIL_ANALYZER_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_UTIL_ANALYZER_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/iface/IStateDumper.h>
#include <lsp-plug.in/dsp-units/misc/windows.h>
#include <lsp-plug.in/dsp/dsp.h>

namespace lsp
{
    namespace dspu
    {
        enum freq_analyzer_flags_t
        {
            FRQA_SCALE_LOGARITHMIC  = 0x0000,
            FRQA_SCALE_LINEAR       = 0x0001,
            FRQA_SCALE_MASK         = 0x000f,
            FRQA_FUNC_NEAREST       = 0x0000,
            FRQA_FUNC_MAX           = 0x0010,
            FRQA_FUNC_MIN           = 0x0020,
            FRQA_FUNC_AVG           = 0x0030,
            FRQA_FUNC_MASK          = 0x00f0,
            FRQA_INT_NONE           = 0x0000,
            FRQA_INT_LINEAR         = 0x0100,
            FRQA_INT_CUBIC          = 0x0200,
            FRQA_INT_MASK           = 0x0300
        };

        class LSP_DSP_UNITS_PUBLIC Analyzer
        {
            private:
                struct impl_t;
                impl_t     *pImpl;

            public:
                explicit Analyzer();
                Analyzer(const Analyzer &) = delete;
                Analyzer & operator = (const Analyzer &) = delete;
                ~Analyzer();

                void        construct();
                void        destroy();

            public:
                bool        init(size_t channels, size_t max_rank, size_t max_sr, float min_rate, size_t max_delay = 0);
                void        set_sample_rate(size_t sr);
                void        set_rate(float rate);
                void        set_window(size_t window);
                void        set_envelope(size_t envelope);
                void        set_shift(float shift);
                void        set_reactivity(float reactivity);
                bool        set_rank(size_t rank);
                void        set_activity(bool active);
                bool        freeze_channel(size_t channel, bool freeze);
                bool        enable_channel(size_t channel, bool enable);
                bool        set_channel_delay(size_t channel, size_t delay);
                size_t      get_rank() const;
                size_t      get_channels() const;
                size_t      get_window() const;
                size_t      get_envelope() const;
                float       get_shift() const;
                size_t      get_sample_rate() const;
                size_t      get_max_sample_rate() const;
                float       get_rate() const;
                float       get_min_rate() const;
                float       get_reactivity() const;
                bool        activity() const;
                bool        channel_active(size_t channel) const;
                size_t      channel_delay(size_t channel) const;
                void        reset();
                bool        read_frequencies(float *frq, float start, float stop, size_t count, size_t flags = FRQA_SCALE_LOGARITHMIC);
                void        reconfigure();
                bool        needs_reconfiguration() const;
                void        process(const float * const *in, size_t samples);
                bool        get_spectrum(size_t channel, float *out, const uint32_t *idx, size_t count);
                float       get_level(size_t channel, const uint32_t idx);
                void        get_frequencies(float *frq, uint32_t *idx, float start, float stop, size_t count, bool linear = false);
                void        dump(IStateDumper *v) const;
        };
    }
}

#endif
