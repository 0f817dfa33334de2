// lsp::dspu::Filter on the GPU library: filter_params_t -> biquad sections (host designer) -> FilterBank.
//
// Binary layout: data members, order and inline members of the reference class
// (include/lsp-plug.in/dsp-units/filters/Filter.h:41-65,172-180 of lsp-dsp-units 1.0.36; 88 bytes, LP64).  All of
// them are live: pBank, sParams, nSampleRate, nMode, nItems / vItems (the analog cascades of the last rebuild),
// nFlags (FF_OWN_BANK | FF_REBUILD | FF_CLEAR) and nLatency mean what they mean there, so the reference's inline
// clear() / latency() / inactive() / active() work on these objects unchanged.  vData owns the cascade storage.
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_FILTERS_FILTER_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_FILTERS_FILTER_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/iface/IStateDumper.h>
#include <lsp-plug.in/dsp-units/filters/common.h>
#include <lsp-plug.in/dsp-units/filters/FilterBank.h>

namespace lsp
{
    namespace dspu
    {
        class LSP_DSP_UNITS_PUBLIC Filter
        {
            protected:
                enum filter_mode_t
                {
                    FM_BYPASS,          // no sections: process() copies
                    FM_BILINEAR,        // bilinear transform of the analog cascades
                    FM_MATCHED,         // matched Z transform
                    FM_APO              // digital design (APO / weighting filters)
                };

                enum filter_flags_t
                {
                    FF_OWN_BANK     = 1 << 0,
                    FF_REBUILD      = 1 << 1,
                    FF_CLEAR        = 1 << 2
                };

            protected:
                FilterBank         *pBank;          // the bank the sections go to
                filter_params_t     sParams;        // limited parameters
                size_t              nSampleRate;
                filter_mode_t       nMode;          // mode of the last rebuild()
                size_t              nItems;         // analog cascades of the last rebuild()
                dsp::f_cascade_t   *vItems;
                uint8_t            *vData;          // allocation behind vItems and the designer's workspace
                size_t              nFlags;
                size_t              nLatency;

            public:
                explicit Filter();
                Filter(const Filter &) = delete;
                Filter(Filter &&) = delete;
                ~Filter();

                Filter & operator = (const Filter &) = delete;
                Filter & operator = (Filter &&) = delete;

                void                construct();                    // valid on raw (e.g. zeroed) memory
                bool                init(FilterBank *fb);           // NULL: the filter owns a private bank
                void                destroy();

            public:
                void                update(size_t sr, const filter_params_t *params);
                void                limit(size_t sr, filter_params_t *fp);
                void                set_sample_rate(size_t sr);
                void                get_params(filter_params_t *params);
                void                process(float *out, const float *in, size_t samples);
                bool                impulse_response(float *out, size_t length);
                void                freq_chart(float *re, float *im, const float *f, size_t count);
                void                freq_chart(float *c, const float *f, size_t count);

                inline void         clear()             { nFlags     |= FF_CLEAR;       }

                void                rebuild();

                inline size_t       latency() const     { return nLatency;  }

                inline bool         inactive() const    { return nMode == FM_BYPASS; }

                inline bool         active() const      { return nMode != FM_BYPASS; }

                void                dump(IStateDumper *v) const;
        };
    }
}

#endif
