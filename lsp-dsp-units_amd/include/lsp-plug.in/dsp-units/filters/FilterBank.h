// lsp::dspu::FilterBank on the GPU library: a list of biquad sections that run in series.
// Host-pointer, one-channel compatibility class; many channels at once go through mi_biquad_bank_* directly.
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_FILTERS_FILTERBANK_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_FILTERS_FILTERBANK_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/iface/IStateDumper.h>
#include <lsp-plug.in/dsp/dsp.h>

namespace lsp
{
    namespace dspu
    {
        class LSP_DSP_UNITS_PUBLIC FilterBank
        {
            private:
                struct impl_t;
                impl_t     *pImpl;

            public:
                explicit FilterBank();
                FilterBank(const FilterBank &) = delete;
                FilterBank & operator = (const FilterBank &) = delete;
                ~FilterBank();

                void                construct();
                bool                init(size_t filters);           // capacity in biquad sections
                void                destroy();

            public:
                void                begin();                        // forget the current chains
                size_t              max_chains() const;
                dsp::biquad_x1_t   *add_chain();                    // next slot (the last one again when full)
                dsp::biquad_x1_t   *chain(size_t id);
                void                end(bool clear = false);        // publish chains; clears delays if asked / count changed
                void                process(float *out, const float *in, size_t samples);
                void                impulse_response(float *out, size_t samples);
                size_t              size() const;
                void                reset();
                void                dump(IStateDumper *v) const;
        };
    }
}

#endif
