// lsp::dspu::FilterBank on the GPU library: a list of biquad sections that run in series.
// Host-pointer, one-channel compatibility class; many channels at once go through mi_biquad_bank_* directly.
//
// Binary layout: the data members, their order and the inline members are those of the reference class
// (include/lsp-plug.in/dsp-units/filters/FilterBank.h:39-46,78-88,127 of lsp-dsp-units 1.0.36), so an object
// built against the reference headers has the same size (56 bytes, LP64) and the same meaning in every field that
// inline code touches: nItems / nMaxItems / nLastItems and the host array vChains are live.  vFilters (the packed
// x8/x4/x2/x1 banks of the CPU path) has no counterpart -- the sections live in a device table -- and carries the
// handle of the device bank instead; vData owns the host allocation as in the reference.
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
            protected:
                dsp::biquad_t      *vFilters;   // here: opaque handle of the device bank (never dereferenced as biquad_t)
                dsp::biquad_x1_t   *vChains;    // the sections handed out by add_chain(), host memory
                size_t              nItems;     // sections added since begin()
                size_t              nMaxItems;  // capacity in sections
                size_t              nLastItems; // section count before the last begin()
                float              *vBackup;    // here: staging rows on the device (see FilterBank::process)
                uint8_t            *vData;      // the host allocation behind vChains

            public:
                explicit FilterBank();
                FilterBank(const FilterBank &) = delete;
                FilterBank(FilterBank &&) = delete;
                ~FilterBank();

                FilterBank & operator = (const FilterBank &) = delete;
                FilterBank & operator = (FilterBank &&) = delete;

                void                construct();                    // valid on raw (e.g. zeroed) memory
                bool                init(size_t filters);           // capacity in biquad sections
                void                destroy();

            public:
                inline void         begin()                         // forget the current chains
                {
                    nLastItems      = nItems;
                    nItems          = 0;
                }

                inline size_t       max_chains() const  { return nMaxItems; }

                dsp::biquad_x1_t   *add_chain();                    // next slot (the last one again when full)
                dsp::biquad_x1_t   *chain(size_t id);
                void                end(bool clear = false);        // publish chains; clears delays if asked / count changed
                void                process(float *out, const float *in, size_t samples);
                void                impulse_response(float *out, size_t samples);

                inline size_t       size() const { return nItems; }

                void                reset();
                void                dump(IStateDumper *v) const;
        };
    }
}

#endif
