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

        // lsp-dsp-lib's packed banks, kept for layout only (FilterBank holds a pointer of this type; field use as in
        // src/main/filters/FilterBank.cpp:112-230 of the reference): 16 delay values, then the coefficients of one
        // x1 / x2 / x4 / x8 group, 256 bytes, 64-byte aligned.  Nothing in this library computes with it.
        struct biquad_x2_t { float b0[2], b1[2], b2[2], a1[2], a2[2], p[2]; };
        struct biquad_x4_t { float b0[4], b1[4], b2[4], a1[4], a2[4]; };
        struct biquad_x8_t { float b0[8], b1[8], b2[8], a1[8], a2[8]; };
        struct alignas(64) biquad_t
        {
            float d[16];
            union
            {
                biquad_x1_t x1;
                biquad_x2_t x2;
                biquad_x4_t x4;
                biquad_x8_t x8;
            };
            float __pad[8];
        };
        static_assert(sizeof(biquad_x1_t) == 32 && sizeof(biquad_x8_t) == 160 && sizeof(biquad_t) == 256, "lsp-dsp-lib layouts");

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
