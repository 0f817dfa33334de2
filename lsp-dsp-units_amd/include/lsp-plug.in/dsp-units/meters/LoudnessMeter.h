// lsp::dspu::LoudnessMeter on the GPU library (one meter, bound host pointers; the device-resident form for many
// meters is mi_loudness_bank_*).
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_METERS_LOUDNESSMETER_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_METERS_LOUDNESSMETER_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/iface/IStateDumper.h>
#include <lsp-plug.in/dsp-units/misc/broadcast.h>
#include <lsp-plug.in/dsp/dsp.h>

namespace lsp
{
    namespace dspu
    {
        class LSP_DSP_UNITS_PUBLIC LoudnessMeter
        {
            // Binary layout: data members, order and inline members of the reference class
            // (include/lsp-plug.in/dsp-units/meters/LoudnessMeter.h:53-106,156-275 of lsp-dsp-units 1.0.36); pData owns the
            // channel records and the GPU bank, the members the inline accessors read are kept current.
            protected:
                enum flags_t
                {
                    F_UPD_FILTERS   = 1 << 0,
                    F_UPD_TIME      = 1 << 1,
                    F_UPD_ALL       = F_UPD_FILTERS | F_UPD_TIME
                };
                struct channel_t;                       // (the reference's record holds a FilterBank and a Filter by value)

            protected:
                channel_t              *vChannels;
                float                  *vBuffer;
                float                   fPeriod;
                float                   fMaxPeriod;
                float                   fAvgCoeff;
                float                   fLoudness;
                size_t                  nSampleRate;
                size_t                  nPeriod;
                size_t                  nMSRefresh;
                size_t                  nChannels;
                size_t                  nFlags;
                size_t                  nDataHead;
                size_t                  nDataSize;
                bs::weighting_t         enWeight;
                uint8_t                *pData;
                uint8_t                *pVarData;

            protected:
                struct impl_t;
                impl_t         *impl() const            { return reinterpret_cast<impl_t *>(pData); }
                void            run(float *out, size_t count, float gain, bool with_gain);

            public:
                explicit LoudnessMeter();
                LoudnessMeter(const LoudnessMeter &) = delete;
                LoudnessMeter(LoudnessMeter &&) = delete;
                LoudnessMeter & operator = (const LoudnessMeter &) = delete;
                LoudnessMeter & operator = (LoudnessMeter &&) = delete;
                ~LoudnessMeter();

                void            construct();            // valid on raw (e.g. zeroed) memory
                void            destroy();
                status_t        init(size_t channels, float max_period = bs::LUFS_MEASURE_PERIOD_MS);

            public:
                status_t        bind(size_t id, float *out, const float *in, size_t pos = 0);
                inline status_t unbind(size_t id)               { return bind(id, NULL, 0); }
                status_t        set_designation(size_t id, bs::channel_t designation);
                bs::channel_t   designation(size_t id) const;
                status_t        set_link(size_t id, float link);
                float           link(size_t id) const;
                status_t        set_active(size_t id, bool active = true);
                bool            active(size_t id) const;
                void            set_weighting(bs::weighting_t weighting);
                inline bs::weighting_t weighting() const        { return enWeight; }
                void            set_period(float period);
                inline float    period() const                  { return fPeriod; }
                inline bool     needs_update() const            { return nFlags != 0; }
                void            update_settings();
                status_t        set_sample_rate(size_t sample_rate);
                inline size_t   sample_rate() const             { return nSampleRate; }
                size_t          latency() const;
                void            process(float *out, size_t count);
                void            process(float *out, size_t count, float gain);
                inline float    loudness() const                { return fLoudness; }
                void            clear();
                void            dump(IStateDumper *v) const;
        };
    }
}

#endif
