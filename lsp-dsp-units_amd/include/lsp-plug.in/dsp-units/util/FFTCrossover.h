// lsp::dspu::FFTCrossover on the GPU library (one channel, band handlers called with HOST data as in the reference; all
// bands are shaped from one forward transform per hop inside one kernel, mi_splitter_bank_*).
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_UTIL_FFTCROSSOVER_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_UTIL_FFTCROSSOVER_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/iface/IStateDumper.h>
#include <lsp-plug.in/dsp-units/util/Crossover.h>
#include <lsp-plug.in/dsp-units/util/SpectralSplitter.h>
#include <lsp-plug.in/dsp/dsp.h>

namespace lsp
{
    namespace dspu
    {
        class LSP_DSP_UNITS_PUBLIC FFTCrossover
        {
            // Binary layout: data members, order and inline members of the reference class
            // (include/lsp-plug.in/dsp-units/util/FFTCrossover.h:46-73,130,352-417 of lsp-dsp-units 1.0.36).
            protected:
                typedef struct band_t
                {
                    float               fHpfFreq;
                    float               fLpfFreq;
                    float               fHpfSlope;
                    float               fLpfSlope;
                    float               fGain;
                    float               fFlatten;
                    bool                bLpf;
                    bool                bHpf;
                    bool                bEnabled;
                    bool                bUpdate;

                    void               *pObject;
                    void               *pSubject;
                    crossover_func_t    pFunc;
                    float              *vFFT;           // the band's gains, host copy (the device holds what the transforms use)
                } split_t;

            protected:
                dspu::SpectralSplitter  sSplitter;
                band_t                 *vBands;
                size_t                  nSampleRate;
                uint8_t                *pData;

            protected:
                static void spectral_sink(void *object, void *subject, const float *samples, size_t first, size_t count);
                void            update_band(band_t *b);
                void            sync_binding(size_t band, band_t *b);
                void            mark_bands_for_update();

            public:
                explicit FFTCrossover();
                FFTCrossover(const FFTCrossover &) = delete;
                FFTCrossover & operator = (const FFTCrossover &) = delete;
                ~FFTCrossover();

                void            construct();
                void            destroy();
                status_t        init(size_t max_rank, size_t bands);

            public:
                inline size_t   bands() const                   { return sSplitter.handlers(); }
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
                inline size_t   sample_rate() const             { return nSampleRate; }
                void            set_rank(size_t rank);
                void            set_phase(float phase);
                inline float    phase() const                   { return sSplitter.phase(); }
                inline size_t   rank() const                    { return sSplitter.rank(); }
                inline size_t   latency() const                 { return sSplitter.latency(); }
                bool            freq_chart(size_t band, float *m, const float *f, size_t count);
                bool            needs_update() const;
                void            update_settings();
                void            process(const float *in, size_t samples);
                inline void     clear()                         { sSplitter.clear(); }
                void            dump(IStateDumper *v) const;
        };
    }
}

#endif
