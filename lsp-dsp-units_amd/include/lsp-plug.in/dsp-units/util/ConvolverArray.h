// lsp::dspu::ConvolverArray -- an EXTENSION of this library (like FilterArray), not a class of the reference: N Convolver
// objects of the same rank behind ONE device bank, each with an impulse response of its own.  Object `id` behaves like the
// dspu::Convolver it replaces (util/Convolver.h:60-113: init copies the response, process() is a zero-latency linear
// convolution), and process() runs ALL of them over one block in a single launch on rows that stay in device memory.
//
//     dspu::ConvolverArray ca;
//     ca.init(256, irs, 65536, 65536, 13, 0.0f);       // 256 convolvers, row c of irs[256][65536] is object c's response
//     ca.process(dev_out, dev_in, 4096, 4096);         // rows [convolver][stride] in DEVICE memory
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_UTIL_CONVOLVERARRAY_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_UTIL_CONVOLVERARRAY_H_

#include <lsp-plug.in/dsp-units/version.h>

namespace lsp
{
    namespace dspu
    {
        class LSP_DSP_UNITS_PUBLIC ConvolverArray
        {
            protected:
                void               *pImpl;

            public:
                explicit ConvolverArray();
                ConvolverArray(const ConvolverArray &) = delete;
                ~ConvolverArray();
                ConvolverArray & operator = (const ConvolverArray &) = delete;

                void                construct();
                /** Convolver::init(data, count, rank, phase) of `convolvers` objects (Convolver.cpp:77-215): object c takes
                 *  `count` taps (or counts[c], if given) from HOST row c of irs[convolvers][ir_stride] */
                bool                init(size_t convolvers, const float *irs, size_t ir_stride, size_t count, size_t rank,
                                         float phase = 0.0f, const size_t *counts = NULL);
                void                destroy();
                inline bool         valid() const { return pImpl != NULL; }
                size_t              size() const;
                /** Convolver::rank() / data_size() (util/Convolver.h:100-106): the longest response of the array */
                size_t              rank() const;
                size_t              data_size() const;
                /** forget all input history (the state right after init) */
                void                reset(void *stream = NULL);

                /** Convolver::process(dst, src, count) of EVERY object: row c of the DEVICE arrays [convolvers][stride];
                 *  out may be in.  Launches on `stream` (a hipStream_t, NULL = default stream), nothing is synchronised. */
                bool                process(float *dev_out, const float *dev_in, size_t samples, size_t stride, void *stream = NULL);
                /** `blocks` consecutive process() calls: dev_out[k] / dev_in[k] are the DEVICE arrays of block k (the pointer
                 *  tables themselves in host memory).  Whole frames go in batches of up to 16 whose tails come out of one pass
                 *  over the partitions' images; the samples are those of the calls one by one, bit for bit. */
                bool                process_blocks(float *const *dev_out, const float *const *dev_in, size_t blocks, size_t samples,
                                                   size_t stride, void *stream = NULL);
                /** the same on HOST rows: one upload, the launch, one download (synchronises the default stream) */
                bool                process_host(float *out, const float *in, size_t samples, size_t stride);
        };
    }
}

#endif
