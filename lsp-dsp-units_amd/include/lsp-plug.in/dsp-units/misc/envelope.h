// lsp::dspu::envelope on the GPU library's host side: the colour list and the linear-grid generators the Analyzer path
// uses (mi_envelope_* of mi_dspu.h).  The logarithmic-grid and frequency-list generators of the reference are not part of
// the streaming path and are not provided.
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_MISC_ENVELOPE_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_MISC_ENVELOPE_H_

#include <lsp-plug.in/dsp-units/version.h>

#include <cstddef>

namespace lsp
{
    namespace dspu
    {
        namespace envelope
        {
            enum envelope_t
            {
                VIOLET_NOISE, BLUE_NOISE, WHITE_NOISE, PINK_NOISE, BROWN_NOISE, MINUS_4_5_DB, PLUS_4_5_DB,
                TOTAL, FIRST = VIOLET_NOISE, LAST = TOTAL - 1
            };

            // dst[i] = ((first + i (last - first) / (n - 1)) / center)^k, k = slope of the colour / of its opposite
            LSP_DSP_UNITS_PUBLIC void noise_lin(float *dst, float first, float last, float center, size_t n, envelope_t type);
            LSP_DSP_UNITS_PUBLIC void reverse_noise_lin(float *dst, float first, float last, float center, size_t n, envelope_t type);
            // fixed colours (the `type` argument is ignored, as in the reference)
            LSP_DSP_UNITS_PUBLIC void white_noise_lin(float *dst, float first, float last, float center, size_t n, envelope_t type);
            LSP_DSP_UNITS_PUBLIC void pink_noise_lin(float *dst, float first, float last, float center, size_t n, envelope_t type);
            LSP_DSP_UNITS_PUBLIC void brown_noise_lin(float *dst, float first, float last, float center, size_t n, envelope_t type);
            LSP_DSP_UNITS_PUBLIC void blue_noise_lin(float *dst, float first, float last, float center, size_t n, envelope_t type);
            LSP_DSP_UNITS_PUBLIC void violet_noise_lin(float *dst, float first, float last, float center, size_t n, envelope_t type);
        }
    }
}

#endif
