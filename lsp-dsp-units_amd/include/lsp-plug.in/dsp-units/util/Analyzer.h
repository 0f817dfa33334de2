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

        // Binary layout: data members, order and inline members of the reference class
        // (include/lsp-plug.in/dsp-units/util/Analyzer.h:41-112,144-354 of lsp-dsp-units 1.0.36).  The members the inline
        // accessors read and write are live (the object follows what a caller sets through them at its next
        // reconfigure() / process()); vData owns the channel records and the GPU bank.
        class LSP_DSP_UNITS_PUBLIC Analyzer
        {
            protected:
                enum reconfigure_t
                {
                    R_ENVELOPE  = 1 << 0,
                    R_WINDOW    = 1 << 1,
                    R_ANALYSIS  = 1 << 2,
                    R_TAU       = 1 << 3,
                    R_COUNTERS  = 1 << 4,
                    R_ALL       = R_ENVELOPE | R_WINDOW | R_ANALYSIS | R_TAU | R_COUNTERS
                };

                typedef struct channel_t
                {
                    float      *vBuffer;            // (device side: the bank's ring)
                    float      *vAmp;
                    float      *vData;
                    uint32_t    nDelay;             // i * nStep after a reconfigure
                    uint32_t    nUserDelay;
                    bool        bFreeze;
                    bool        bActive;
                } channel_t;

            protected:
                uint32_t    nChannels;
                uint32_t    nMaxRank;
                uint32_t    nRank;
                uint32_t    nSampleRate;
                uint32_t    nMaxSampleRate;
                uint32_t    nBufSize;
                uint32_t    nCounter;
                uint32_t    nPeriod;
                uint32_t    nStep;
                uint32_t    nHead;
                uint32_t    nReconfigure;
                uint32_t    nEnvelope;
                uint32_t    nWindow;
                uint32_t    nMaxUserDelay;

                float       fReactivity;
                float       fTau;
                float       fRate;
                float       fMinRate;
                float       fShift;

                bool        bActive;

                channel_t  *vChannels;
                void       *vData;              // allocation behind vChannels and the GPU state
                float      *vSigRe;
                float      *vFftReIm;
                float      *vWindow;
                float      *vEnvelope;

            protected:
                struct impl_t;
                impl_t     *impl() const            { return static_cast<impl_t *>(vData); }
                void        sync_inline_state();    // what inline set_activity() / reset() changed, into the bank

            public:
                explicit Analyzer();
                Analyzer(const Analyzer &) = delete;
                Analyzer(Analyzer &&) = delete;
                Analyzer & operator = (const Analyzer &) = delete;
                Analyzer & operator = (Analyzer &&) = delete;
                ~Analyzer();

                void        construct();                    // valid on raw (e.g. zeroed) memory
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
                bool        freeze_channel(size_t channel, bool freeze);
                bool        enable_channel(size_t channel, bool enable);
                bool        set_channel_delay(size_t channel, size_t delay);

                inline size_t get_channels() const              { return nChannels; }
                inline size_t get_window() const                { return nWindow; }
                inline size_t get_envelope() const              { return nEnvelope; }
                inline float  get_shift() const                 { return fShift; }
                inline size_t get_sample_rate() const           { return nSampleRate; }
                inline size_t get_max_sample_rate() const       { return nMaxSampleRate; }
                inline float  get_rate() const                  { return fRate; }
                inline float  get_min_rate() const              { return fMinRate; }
                inline float  get_reactivity() const            { return fReactivity; }
                inline size_t get_rank() const                  { return nRank; }
                inline void   set_activity(bool active)         { bActive = active; }
                inline bool   activity() const                  { return bActive; }
                inline bool   channel_active(size_t channel) const { return (channel < nChannels) ? vChannels[channel].bActive : false; }
                inline size_t channel_delay(size_t channel) const  { return (channel < nChannels) ? vChannels[channel].nUserDelay : 0; }
                inline void   reset()                           { nReconfigure |= R_ANALYSIS; }
                inline bool   needs_reconfiguration() const     { return nReconfigure; }

                bool        read_frequencies(float *frq, float start, float stop, size_t count, size_t flags = FRQA_SCALE_LOGARITHMIC);
                void        reconfigure();
                void        process(const float * const *in, size_t samples);
                bool        get_spectrum(size_t channel, float *out, const uint32_t *idx, size_t count);
                float       get_level(size_t channel, const uint32_t idx);
                void        get_frequencies(float *frq, uint32_t *idx, float start, float stop, size_t count, bool linear = false);
                void        dump(IStateDumper *v) const;
        };
    }
}

#endif
