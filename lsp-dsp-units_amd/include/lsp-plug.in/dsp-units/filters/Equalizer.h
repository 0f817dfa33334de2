// lsp::dspu::Equalizer on the GPU library (one channel, host pointers; many channels: mi_equalizer_bank_*).
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_FILTERS_EQUALIZER_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_FILTERS_EQUALIZER_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/iface/IStateDumper.h>
#include <lsp-plug.in/dsp-units/filters/common.h>

namespace lsp
{
    namespace dspu
    {
        enum equalizer_mode_t
        {
            EQM_BYPASS  = MI_EQM_BYPASS,
            EQM_IIR     = MI_EQM_IIR,
            EQM_FIR     = MI_EQM_FIR,
            EQM_FFT     = MI_EQM_FFT,
            EQM_SPM     = MI_EQM_SPM
        };

        class LSP_DSP_UNITS_PUBLIC Equalizer
        {
            private:
                struct impl_t;
                impl_t     *pImpl;

            public:
                explicit Equalizer();
                Equalizer(const Equalizer &) = delete;
                Equalizer & operator = (const Equalizer &) = delete;
                ~Equalizer();

                void                construct();
                bool                init(size_t filters, size_t fir_rank);
                void                destroy();

            public:
                bool                configuration_changed() const;
                bool                set_params(size_t id, const filter_params_t *params);
                bool                limit_params(size_t id, filter_params_t *fp);
                bool                get_params(size_t id, filter_params_t *params);
                void                set_mode(equalizer_mode_t mode);
                void                set_actual_sample_rate(size_t sample_rate);
                void                set_sample_rate(size_t sr);
                equalizer_mode_t    get_mode() const;
                equalizer_mode_t    mode() const;
                size_t              get_latency();
                size_t              max_latency() const;
                bool                freq_chart(size_t id, float *c, const float *f, size_t count);
                void                freq_chart(float *c, const float *f, size_t count);
                void                process(float *out, const float *in, size_t samples);
                void                reset();
                size_t              fir_rank() const;
                size_t              ir_size() const;
                size_t              fir_ir_size() const;
                size_t              actual_sample_rate() const;
                bool                filter_active(size_t id) const;
                bool                filter_inactive(size_t id) const;
                bool                smooth() const;
                void                set_smooth(bool smooth);
                void                dump(IStateDumper *v) const;
        };
    }
}

#endif
