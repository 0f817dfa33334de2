// lsp::dspu::Filter on the GPU library: filter_params_t -> biquad sections (host designer) -> FilterBank.
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
            private:
                struct impl_t;
                impl_t     *pImpl;

            public:
                explicit Filter();
                Filter(const Filter &) = delete;
                Filter & operator = (const Filter &) = delete;
                ~Filter();

                void                construct();
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
                void                clear();
                void                rebuild();
                size_t              latency() const;
                bool                inactive() const;
                bool                active() const;
                void                dump(IStateDumper *v) const;
        };
    }
}

#endif
