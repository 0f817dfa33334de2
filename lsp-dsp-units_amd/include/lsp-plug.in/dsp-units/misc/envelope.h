// lsp::dspu::envelope on the GPU library's host side (mi_envelope_* of mi_dspu.h): the colour list and the generators
// on a linear grid (what the Analyzer path uses), a logarithmic grid and an explicit list of frequencies
// (include/lsp-plug.in/dsp-units/misc/envelope.h:57-180 of the reference).
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
            // dst[i] = ((first exp(i ln(last / first) / (n - 1))) / center)^k
            LSP_DSP_UNITS_PUBLIC void noise_log(float *dst, float first, float last, float center, size_t n, envelope_t type);
            LSP_DSP_UNITS_PUBLIC void reverse_noise_log(float *dst, float first, float last, float center, size_t n, envelope_t type);
            LSP_DSP_UNITS_PUBLIC void white_noise_log(float *dst, float first, float last, float center, size_t n, envelope_t type);
            LSP_DSP_UNITS_PUBLIC void pink_noise_log(float *dst, float first, float last, float center, size_t n, envelope_t type);
            LSP_DSP_UNITS_PUBLIC void brown_noise_log(float *dst, float first, float last, float center, size_t n, envelope_t type);
            LSP_DSP_UNITS_PUBLIC void blue_noise_log(float *dst, float first, float last, float center, size_t n, envelope_t type);
            LSP_DSP_UNITS_PUBLIC void violet_noise_log(float *dst, float first, float last, float center, size_t n, envelope_t type);
            // dst[i] = (freqs[i] / center)^k
            LSP_DSP_UNITS_PUBLIC void noise_list(float *dst, const float *freqs, float center, size_t n, envelope_t type);
            LSP_DSP_UNITS_PUBLIC void reverse_noise_list(float *dst, const float *freqs, float center, size_t n, envelope_t type);
            LSP_DSP_UNITS_PUBLIC void white_noise_list(float *dst, const float *freqs, float center, size_t n, envelope_t type);
            LSP_DSP_UNITS_PUBLIC void pink_noise_list(float *dst, const float *freqs, float center, size_t n, envelope_t type);
            LSP_DSP_UNITS_PUBLIC void brown_noise_list(float *dst, const float *freqs, float center, size_t n, envelope_t type);
            LSP_DSP_UNITS_PUBLIC void blue_noise_list(float *dst, const float *freqs, float center, size_t n, envelope_t type);
            LSP_DSP_UNITS_PUBLIC void violet_noise_list(float *dst, const float *freqs, float center, size_t n, envelope_t type);
        }
    }
}

#endif
