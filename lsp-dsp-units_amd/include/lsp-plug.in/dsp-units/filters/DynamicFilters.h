// lsp::dspu::DynamicFilters on the GPU library (one channel, host pointers; many channels: mi_dynfilter_bank_*).
//
// Binary layout: the reference's data members in the reference's order and its inline members
// (include/lsp-plug.in/dsp-units/filters/DynamicFilters.h:43-75,147-164 of lsp-dsp-units 1.0.36; 64 bytes, LP64).
// vFilters is the live host array the inline filter_active() / filter_inactive() / set_filter_active() work on (the
// parameters in it are the transformed ones set_params() leaves, DynamicFilters.cpp:170-178); nFilters, nSampleRate and
// bClearMem are live.  The cascade / biquad work buffers and the filter memory of the CPU path are device memory here
// (vCascades, vMemory, vBiquads stay NULL); pData owns the object's device-side state.
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_FILTERS_DYNAMICFILTERS_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_FILTERS_DYNAMICFILTERS_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/iface/IStateDumper.h>
#include <lsp-plug.in/dsp-units/filters/common.h>
#include <lsp-plug.in/dsp/dsp.h>

namespace lsp
{
    namespace dspu
    {
        class LSP_DSP_UNITS_PUBLIC DynamicFilters
        {
            protected:
                typedef struct filter_t
                {
                    filter_params_t     sParams;
                    bool                bActive;
                } filter_t;

                union biquad_bank_t
                {
                    void               *ptr;
                    dsp::biquad_x1_t   *x1;
                    dsp::biquad_x2_t   *x2;
                    dsp::biquad_x4_t   *x4;
                    dsp::biquad_x8_t   *x8;
                };

            protected:
                filter_t           *vFilters;
                dsp::f_cascade_t   *vCascades;
                float              *vMemory;
                biquad_bank_t       vBiquads;
                size_t              nFilters;
                size_t              nSampleRate;
                void               *pData;              // here: the object's device-side state (opaque)
                bool                bClearMem;

            private:
                struct impl_t;
                inline impl_t      *impl() const        { return static_cast<impl_t *>(pData); }

            public:
                explicit DynamicFilters();
                DynamicFilters(const DynamicFilters &) = delete;
                DynamicFilters(DynamicFilters &&) = delete;
                ~DynamicFilters();

                DynamicFilters & operator = (const DynamicFilters &) = delete;
                DynamicFilters & operator = (DynamicFilters &&) = delete;

                void                construct();            // valid on raw (e.g. zeroed) memory
                status_t            init(size_t filters);
                void                destroy();

            public:
                void                set_sample_rate(size_t sr);

                inline bool         filter_active(size_t id) const { return (id < nFilters) ? vFilters[id].bActive : false; };

                inline bool         filter_inactive(size_t id) const { return (id < nFilters) ? !vFilters[id].bActive : true; };

                inline bool         set_filter_active(size_t id, bool active)
                {
                    if (id >= nFilters)
                        return false;
                    vFilters[id].bActive        = true;     // (sic: the reference activates whatever is asked)
                    return true;
                }

                bool                set_params(size_t id, const filter_params_t *params);
                bool                get_params(size_t id, filter_params_t *params);
                void                process(size_t id, float *out, const float *in, const float *gain, size_t samples);
                bool                freq_chart(size_t id, float *re, float *im, const float *f, float gain, size_t count);
                bool                freq_chart(size_t id, float *dst, const float *f, float gain, size_t count);
                void                dump(IStateDumper *v) const;
        };
    }
}

#endif
