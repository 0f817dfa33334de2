// lsp::dspu::Equalizer on the GPU library (one channel, host pointers; many channels: mi_equalizer_bank_*).
//
// Binary layout: the reference's data members in the reference's order (filters/Equalizer.h:59-78 of lsp-dsp-units
// 1.0.36; 160 bytes, LP64) and its inline members.  sBank is a real (embedded) FilterBank and vFilters a real array of
// Filter objects bound to it: they carry the parameters and, after every reconfigure, the mode of each filter, so the
// inline filter_active() / filter_inactive() answer as the reference does (FM_BYPASS between update() and the next
// rebuild, Filter.cpp:150).  The scalar members (rates, FIR size and rank, latency, mode, flags) are live.  The six FIR /
// FFT work buffers of the CPU path are device memory here and stay NULL; pData owns the object's device-side state.
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_FILTERS_EQUALIZER_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_FILTERS_EQUALIZER_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/iface/IStateDumper.h>
#include <lsp-plug.in/dsp-units/filters/common.h>
#include <lsp-plug.in/dsp-units/filters/Filter.h>
#include <lsp-plug.in/dsp-units/filters/FilterBank.h>

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
            protected:
                enum eq_flags_t
                {
                    EF_REBUILD  = 1 << 0,
                    EF_CLEAR    = 1 << 1,
                    EF_XFADE    = 1 << 2,
                    EF_SMOOTH   = 1 << 3
                };

            protected:
                FilterBank          sBank;              // embedded as in the reference (the sections run in the device bank)
                Filter             *vFilters;           // nFilters objects: parameters and per-filter mode
                uint32_t            nFilters;
                uint32_t            nSampleRate;
                uint32_t            nActualSampleRate;
                uint32_t            nFirSize;
                uint32_t            nFirRank;
                uint32_t            nLatency;           // of the last reconfigure
                uint32_t            nBufSize;
                equalizer_mode_t    nMode;
                float              *vInBuffer;
                float              *vOutBuffer;
                float              *vNewConv;
                float              *vConv;
                float              *vFft;
                float              *vTemp;
                size_t              nFlags;
                uint8_t            *pData;              // here: the object's device-side state (opaque)

            private:
                struct impl_t;
                inline impl_t      *impl() const        { return reinterpret_cast<impl_t *>(pData); }
                void                rebuilt();

            public:
                explicit Equalizer();
                Equalizer(const Equalizer &) = delete;
                Equalizer(Equalizer &&) = delete;
                ~Equalizer();

                Equalizer & operator = (const Equalizer &) = delete;
                Equalizer & operator = (Equalizer &&) = delete;

                void                construct();            // valid on raw (e.g. zeroed) memory
                bool                init(size_t filters, size_t fir_rank);
                void                destroy();

            public:
                bool                configuration_changed() const;
                bool                set_params(size_t id, const filter_params_t *params);
                bool                limit_params(size_t id, filter_params_t *fp);
                bool                get_params(size_t id, filter_params_t *params);

                inline bool         filter_active(size_t id) const { return (id < nFilters) ? vFilters[id].active() : false; }

                inline bool         filter_inactive(size_t id) const { return (id < nFilters) ? vFilters[id].inactive() : false; }

                void                set_mode(equalizer_mode_t mode);
                void                set_actual_sample_rate(size_t sample_rate);
                void                set_sample_rate(size_t sr);

                inline equalizer_mode_t get_mode() const { return nMode; }

                size_t              get_latency();

                inline size_t       max_latency() const { return nFirSize + (nFirSize >> 1); }

                bool                freq_chart(size_t id, float *re, float *im, const float *f, size_t count);
                bool                freq_chart(size_t id, float *c, const float *f, size_t count);
                void                freq_chart(float *re, float *im, const float *f, size_t count);
                void                freq_chart(float *c, const float *f, size_t count);
                void                process(float *out, const float *in, size_t samples);
                void                reset();

                inline size_t       fir_rank() const        { return nFirRank;      }

                inline size_t       fir_ir_size() const     { return nFirSize << 1; }

                inline equalizer_mode_t     mode() const    { return nMode;         }

                inline size_t       actual_sample_rate() const  { return (nActualSampleRate != 0) ? nActualSampleRate : nSampleRate;    }

                size_t              ir_size() const;
                bool                smooth() const;
                void                set_smooth(bool smooth);
                void                dump(IStateDumper *v) const;
        };
    }
}

#endif
