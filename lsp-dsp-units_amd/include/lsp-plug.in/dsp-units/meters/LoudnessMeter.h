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
            private:
                struct impl_t;
                impl_t     *pImpl;
                void            run(float *out, size_t count, float gain, bool with_gain);

            public:
                explicit LoudnessMeter();
                LoudnessMeter(const LoudnessMeter &) = delete;
                LoudnessMeter & operator = (const LoudnessMeter &) = delete;
                ~LoudnessMeter();

                void            construct();
                void            destroy();
                status_t        init(size_t channels, float max_period = bs::LUFS_MEASURE_PERIOD_MS);

            public:
                status_t        bind(size_t id, float *out, const float *in, size_t pos = 0);
                status_t        unbind(size_t id);
                status_t        set_designation(size_t id, bs::channel_t designation);
                bs::channel_t   designation(size_t id) const;
                status_t        set_link(size_t id, float link);
                float           link(size_t id) const;
                status_t        set_active(size_t id, bool active = true);
                bool            active(size_t id) const;
                void            set_weighting(bs::weighting_t weighting);
                bs::weighting_t weighting() const;
                void            set_period(float period);
                float           period() const;
                bool            needs_update() const;           // settings changed since the last process() / update_settings()
                void            update_settings();
                status_t        set_sample_rate(size_t sample_rate);
                size_t          sample_rate() const;
                size_t          latency() const;
                void            process(float *out, size_t count);
                void            process(float *out, size_t count, float gain);
                float           loudness() const;
                void            clear();
                void            dump(IStateDumper *v) const;
        };
    }
}

#endif
