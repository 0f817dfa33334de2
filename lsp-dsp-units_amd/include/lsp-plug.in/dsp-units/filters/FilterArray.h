// lsp::dspu::FilterArray -- an EXTENSION of this library, not a class of the reference: the batched mode under the
// class API that SURVEY.md section 7 (hard part 2) asks for.
//
// The reference's granularity is one object = one channel = one call (filters/Filter.h, filters/FilterBank.h:114), which
// on a GPU means one launch and two PCIe copies per object and block.  A caller that keeps `dspu::Filter vF[1024]` for
// 1024 channels binds them to ONE FilterArray instead: object `id` of the array behaves like the Filter it replaces
// (update() = Filter::update + rebuild with the same designer, the same lazy rebuild inside process(), the same "memory
// is cleared when the number of sections changes", Filter.cpp:141-167,208-403, FilterBank.cpp:233-235), and process()
// runs ALL of them over one block in a single launch on rows that stay in device memory.
//
//     dspu::FilterArray fa;
//     fa.init(1024);                                   // 1024 filters (channels)
//     for (size_t c = 0; c < 1024; ++c) fa.update(c, 48000, &params[c]);
//     fa.process(dev_out, dev_in, 4096, 4096);         // rows [filter][stride] in DEVICE memory (mi_dspu_malloc / hipMalloc)
//     fa.process_host(out, in, 4096, 4096);            // the same on host rows: one upload, one launch, one download
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_FILTERS_FILTERARRAY_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_FILTERS_FILTERARRAY_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/filters/common.h>

namespace lsp
{
    namespace dspu
    {
        class LSP_DSP_UNITS_PUBLIC FilterArray
        {
            protected:
                void               *pImpl;

            public:
                explicit FilterArray();
                FilterArray(const FilterArray &) = delete;
                ~FilterArray();
                FilterArray & operator = (const FilterArray &) = delete;

                void                construct();
                /** `filters` objects, each with room for `max_chains` biquad sections (a Filter has FILTER_CHAINS_MAX = 128;
                 *  most types need <= 16: the device tables take 832 bytes per filter and section) */
                bool                init(size_t filters, size_t max_chains = 16);
                void                destroy();
                inline bool         valid() const { return pImpl != NULL; }
                size_t              size() const;

                /** Filter::update(sr, params) of object `id`; false for a bad id or a design that needs more sections
                 *  than max_chains (the object then keeps its previous design) */
                bool                update(size_t id, size_t sr, const filter_params_t *params);
                bool                get_params(size_t id, filter_params_t *params) const;
                /** Filter::clear() of object `id` (all objects: id = size_t(-1)): the filter memory is zeroed before the next block */
                void                clear(size_t id = size_t(-1));

                /** Filter::process(out, in, samples) of EVERY object: row `id` of the DEVICE arrays [filters][stride] float32;
                 *  out may be in.  One launch on `stream` (a hipStream_t, NULL = default stream), nothing is synchronised. */
                bool                process(float *dev_out, const float *dev_in, size_t samples, size_t stride, void *stream = NULL);
                /** `blocks` consecutive process() calls: dev_out[k] / dev_in[k] are the DEVICE arrays of block k (the pointer
                 *  tables themselves in host memory).  Blocks of more than 2048 samples in whole chunks of 16 ride ONE launch; the
                 *  samples and the filter memory are those of the calls one by one, bit for bit. */
                bool                process_blocks(float *const *dev_out, const float *const *dev_in, size_t blocks, size_t samples,
                                                   size_t stride, void *stream = NULL);
                /** the same on HOST rows: one upload, one launch, one download (synchronises the default stream) */
                bool                process_host(float *out, const float *in, size_t samples, size_t stride);
        };
    }
}

#endif
