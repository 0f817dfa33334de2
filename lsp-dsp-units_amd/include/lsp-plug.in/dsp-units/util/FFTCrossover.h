// lsp::dspu::FFTCrossover on the GPU library (one channel, band handlers called with HOST data as in the reference; all
// bands are shaped from one forward transform per hop inside one kernel, mi_splitter_bank_*).
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_UTIL_FFTCROSSOVER_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_UTIL_FFTCROSSOVER_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/iface/IStateDumper.h>
#include <lsp-plug.in/dsp-units/util/Crossover.h>
#include <lsp-plug.in/dsp/dsp.h>

namespace lsp
{
    namespace dspu
    {
        class LSP_DSP_UNITS_PUBLIC FFTCrossover
        {
            private:
                struct impl_t;
                impl_t     *pImpl;

            public:
                explicit FFTCrossover();
                FFTCrossover(const FFTCrossover &) = delete;
                FFTCrossover & operator = (const FFTCrossover &) = delete;
                ~FFTCrossover();

                void            construct();
                void            destroy();
                status_t        init(size_t max_rank, size_t bands);

            public:
                size_t          bands() const;
                void            set_slope(size_t band, float lpf, float hpf);
                void            set_lpf_slope(size_t band, float slope);
                void            set_hpf_slope(size_t band, float slope);
                float           lpf_slope(size_t band) const;
                float           hpf_slope(size_t band) const;
                void            set_frequency(size_t band, float lpf, float hpf);
                void            set_lpf_frequency(size_t band, float freq);
                void            set_hpf_frequency(size_t band, float freq);
                float           lpf_frequency(size_t band) const;
                float           hpf_frequency(size_t band) const;
                void            enable_filters(size_t band, bool lpf, bool hpf);
                void            enable_hpf(size_t band, bool enable = true);
                void            enable_lpf(size_t band, bool enable = true);
                bool            lpf_enabled(size_t band) const;
                bool            hpf_enabled(size_t band) const;
                inline void     disable_filters(size_t band)    { enable_filters(band, false, false); }
                inline void     disable_lpf(size_t band)        { enable_lpf(band, false); }
                inline void     disable_hpf(size_t band)        { enable_hpf(band, false); }
                void            set_lpf(size_t band, float freq, float slope, bool enabled = true);
                void            set_hpf(size_t band, float freq, float slope, bool enabled = true);
                void            set_gain(size_t band, float gain);
                float           gain(size_t band) const;
                void            set_flatten(size_t band, float amount);
                float           flatten(size_t band) const;
                void            enable_band(size_t band, bool enable = true);
                inline void     disable_band(size_t band)       { enable_band(band, false); }
                bool            band_enabled(size_t band) const;
                bool            set_handler(size_t band, crossover_func_t func, void *object, void *subject);
                bool            unset_handler(size_t band);
                void            set_sample_rate(size_t sr);
                size_t          sample_rate() const;
                void            set_rank(size_t rank);
                void            set_phase(float phase);
                float           phase() const;
                size_t          rank() const;
                size_t          latency() const;
                bool            freq_chart(size_t band, float *m, const float *f, size_t count);
                bool            needs_update() const;
                void            update_settings();
                void            process(const float *in, size_t samples);
                void            clear();
                void            dump(IStateDumper *v) const;
        };
    }
}

#endif
