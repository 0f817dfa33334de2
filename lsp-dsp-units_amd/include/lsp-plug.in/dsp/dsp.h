// Minimal stand-in for the lsp-dsp-lib types that appear in the lsp::dspu class API
// (FilterBank::add_chain() hands out dsp::biquad_x1_t, Filter keeps dsp::f_cascade_t).
// Only the two plain structs are provided; the arithmetic of lsp-dsp-lib is replaced by the GPU library.
#ifndef MI_LSP_PLUG_IN_DSP_DSP_H_
#define MI_LSP_PLUG_IN_DSP_DSP_H_

#include <cstddef>
#include <cstdint>
#include <sys/types.h>

namespace lsp
{
    // status codes used by the few hot-path APIs that return one
    typedef int status_t;
    // (lsp-common-lib's status.h is not part of the reference tree: callers compare these symbolically)
    enum { STATUS_OK = 0, STATUS_NO_MEM = 5, STATUS_BAD_STATE = 12, STATUS_OVERFLOW = 18, STATUS_INVALID_VALUE = 27,
           STATUS_NOT_BOUND = 50 };

    namespace dsp
    {
        // one digital section: y = b0 x + b1 x[-1] + b2 x[-2] + a1 y[-1] + a2 y[-2]; p* pad to 32 bytes
        struct biquad_x1_t
        {
            float b0, b1, b2;
            float a1, a2;
            float p0, p1, p2;
        };

        // numerator t[] / denominator b[] of an analog second-order cascade
        struct f_cascade_t
        {
            float t[4];
            float b[4];
        };

        // lsp-dsp-lib's per-thread context / init are no-ops here: there is no SIMD dispatch to select
        struct context_t { uint32_t top; uint32_t data[15]; };
        inline void init() {}
        inline void start(context_t *) {}
        inline void finish(context_t *) {}
    }
}

#endif
