// lsp::dspu::EqualizerArray -- an EXTENSION of this library (like FilterArray), not a class of the reference: N Equalizer
// objects of the same geometry (filters, fir_rank, mode, sample rate) behind ONE device bank.  Object `id` behaves like the
// dspu::Equalizer it replaces (filters/Equalizer.h:84-288: set_params of one of its filters, lazy reconfigure inside process(),
// the same latency), and process() runs ALL of them over one block in a single launch on rows that stay in device memory --
// instead of one launch and two PCIe copies per object and block.
//
//     dspu::EqualizerArray ea;
//     ea.init(256, 32, 12);                            // 256 equalizers of 32 filters, FIR of 2^12 taps
//     ea.set_mode(dspu::EQM_FIR); ea.set_sample_rate(48000);
//     for (c ...) for (i ...) ea.set_params(c, i, &params[c][i]);
//     ea.process(dev_out, dev_in, 4096, 4096);         // rows [equalizer][stride] in DEVICE memory
//     ea.process_blocks(dev_outs, dev_ins, K, 4096, 4096);   // K consecutive blocks: one launch for the run in FIR / FFT mode
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_FILTERS_EQUALIZERARRAY_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_FILTERS_EQUALIZERARRAY_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/filters/common.h>
#include <lsp-plug.in/dsp-units/filters/Equalizer.h>

namespace lsp
{
    namespace dspu
    {
        class LSP_DSP_UNITS_PUBLIC EqualizerArray
        {
            protected:
                void               *pImpl;

            public:
                explicit EqualizerArray();
                EqualizerArray(const EqualizerArray &) = delete;
                ~EqualizerArray();
                EqualizerArray & operator = (const EqualizerArray &) = delete;

                void                construct();
                /** Equalizer::init(filters, fir_rank) of `equalizers` objects (Equalizer.cpp:67-160) */
                bool                init(size_t equalizers, size_t filters, size_t fir_rank);
                void                destroy();
                inline bool         valid() const { return pImpl != NULL; }
                size_t              size() const;

                /** Equalizer::set_params(id, params) of object `eq` (all objects: eq = size_t(-1)), Equalizer.cpp:210-218 */
                bool                set_params(size_t eq, size_t id, const filter_params_t *params);
                bool                get_params(size_t eq, size_t id, filter_params_t *params) const;
                /** set_mode / set_sample_rate / set_smooth / reset of every object (Equalizer.cpp:360-375,188-203,618-626,573-597) */
                void                set_mode(equalizer_mode_t mode);
                void                set_sample_rate(size_t sr);
                void                set_smooth(bool smooth);
                void                reset(void *stream = NULL);
                /** Equalizer::get_latency() (reconfigures first), Equalizer.cpp:237-241: the same for every object */
                size_t              get_latency(void *stream = NULL);

                /** Equalizer::process(out, in, samples) of EVERY object: row `eq` of the DEVICE arrays [equalizers][stride];
                 *  out may be in.  Launches on `stream` (a hipStream_t, NULL = default stream), nothing is synchronised. */
                bool                process(float *dev_out, const float *dev_in, size_t samples, size_t stride, void *stream = NULL);
                /** `blocks` consecutive process() calls: dev_out[k] / dev_in[k] are the DEVICE arrays of block k (the pointer
                 *  tables themselves in host memory) */
                bool                process_blocks(float *const *dev_out, const float *const *dev_in, size_t blocks, size_t samples,
                                                   size_t stride, void *stream = NULL);
                /** process() on HOST rows: one upload, the launch, one download (synchronises the default stream) */
                bool                process_host(float *out, const float *in, size_t samples, size_t stride);
        };
    }
}

#endif
