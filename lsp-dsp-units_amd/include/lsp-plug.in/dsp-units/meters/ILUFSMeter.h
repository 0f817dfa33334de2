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
            // Binary layout: data members, order and inline members of the reference class
            // (include/lsp-plug.in/dsp-units/meters/ILUFSMeter.h:42-108,158-244 of lsp-dsp-units 1.0.36); pData owns the channel
            // records and the GPU bank, the members the inline accessors read are kept current.
            protected:
                enum flags_t
                {
                    F_UPD_FILTERS   = 1 << 0,
                    F_UPD_TIME      = 1 << 1,
                    F_BLK_FULL      = 1 << 2,
                    F_UPD_ALL       = F_UPD_FILTERS | F_UPD_TIME
                };
                struct channel_t;                       // (the reference's record holds a FilterBank and a Filter by value)

            protected:
                channel_t              *vChannels;
                float                  *vBuffer;
                float                  *vLoudness;
                float                   fBlockPeriod;
                float                   fIntTime;
                float                   fMaxIntTime;
                float                   fAvgCoeff;
                float                   fLoudness;
                uint32_t                nBlockSize;
                uint32_t                nBlockOffset;
                uint32_t                nBlockPart;
                uint32_t                nMSSize;
                uint32_t                nMSHead;
                uint32_t                nMSInt;
                uint32_t                nMSCount;
                uint32_t                nSampleRate;
                uint32_t                nChannels;
                uint32_t                nFlags;
                bs::weighting_t         enWeight;
                uint8_t                *pData;
                uint8_t                *pVarData;

            protected:
                struct impl_t;
                impl_t         *impl() const            { return reinterpret_cast<impl_t *>(pData); }

            public:
                explicit ILUFSMeter();
                ILUFSMeter(const ILUFSMeter &) = delete;
                ILUFSMeter(ILUFSMeter &&) = delete;
                ILUFSMeter & operator = (const ILUFSMeter &) = delete;
                ILUFSMeter & operator = (ILUFSMeter &&) = delete;
                ~ILUFSMeter();

                void            construct();            // valid on raw (e.g. zeroed) memory
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
                inline bs::weighting_t weighting() const        { return enWeight; }
                inline bool     needs_update() const            { return nFlags != 0; }
                void            update_settings();
                void            set_integration_period(float period);
                inline float    integration_period() const      { return fIntTime; }
                status_t        set_sample_rate(size_t sample_rate);
                inline size_t   sample_rate() const             { return nSampleRate; }
                void            process(float *out, size_t count, float gain = bs::DBFS_TO_LUFS_SHIFT_GAIN);
                inline float    loudness() const                { return fLoudness; }
                void            clear();
                void            dump(IStateDumper *v) const;
        };
    }
}

#endif
