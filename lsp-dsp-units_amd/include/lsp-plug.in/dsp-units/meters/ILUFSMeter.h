// lsp::dspu::ILUFSMeter on the GPU library (one meter, bound host pointers; the device-resident form for many meters
// is mi_ilufs_bank_*).  A channel without a bound input is skipped, like a disabled one.
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_METERS_ILUFSMETER_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_METERS_ILUFSMETER_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/iface/IStateDumper.h>
#include <lsp-plug.in/dsp-units/misc/broadcast.h>
#include <lsp-plug.in/dsp/dsp.h>

namespace lsp
{
    namespace dspu
    {
        class LSP_DSP_UNITS_PUBLIC ILUFSMeter
        {
            private:
                struct impl_t;
                impl_t     *pImpl;

            public:
                explicit ILUFSMeter();
                ILUFSMeter(const ILUFSMeter &) = delete;
                ILUFSMeter & operator = (const ILUFSMeter &) = delete;
                ~ILUFSMeter();

                void            construct();
                void            destroy();
                status_t        init(size_t channels, float max_int_time = 60, float block_period = bs::LUFS_MEASURE_PERIOD_MS);

            public:
                status_t        bind(size_t id, const float *in);
                inline status_t unbind(size_t id)               { return bind(id, nullptr); }
                status_t        set_designation(size_t id, bs::channel_t designation);
                bs::channel_t   designation(size_t id) const;
                status_t        set_active(size_t id, bool active = true);
                bool            active(size_t id) const;
                void            set_weighting(bs::weighting_t weighting);
                bs::weighting_t weighting() const;
                bool            needs_update() const;           // settings changed since the last process() / update_settings()
                void            update_settings();
                void            set_integration_period(float period);
                float           integration_period() const;
                status_t        set_sample_rate(size_t sample_rate);
                size_t          sample_rate() const;
                void            process(float *out, size_t count, float gain = bs::DBFS_TO_LUFS_SHIFT_GAIN);
                float           loudness() const;
                void            clear();
                void            dump(IStateDumper *v) const;
        };
    }
}

#endif
