// lsp::dspu::crossover::* on the GPU library's host side (mi_crossover_* of mi_dspu.h).
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_MISC_FFT_CROSSOVER_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_MISC_FFT_CROSSOVER_H_

#include <lsp-plug.in/dsp-units/version.h>

#include <cstddef>

namespace lsp
{
    namespace dspu
    {
        namespace crossover
        {
            // magnitude of the crossover's high-pass / low-pass with cut-off f0 and `slope` dB/oct (negative) at f
            LSP_DSP_UNITS_PUBLIC float hipass(float f, float f0, float slope);
            LSP_DSP_UNITS_PUBLIC float lopass(float f, float f0, float slope);
            // gain[i] = / *= magnitude at f[i]
            LSP_DSP_UNITS_PUBLIC void hipass_set(float *gain, const float *f, float f0, float slope, size_t count);
            LSP_DSP_UNITS_PUBLIC void hipass_apply(float *gain, const float *f, float f0, float slope, size_t count);
            LSP_DSP_UNITS_PUBLIC void lopass_set(float *gain, const float *f, float f0, float slope, size_t count);
            LSP_DSP_UNITS_PUBLIC void lopass_apply(float *gain, const float *f, float f0, float slope, size_t count);
            // the same on the 2^rank bins of an FFT at `sample_rate`, in FFT order
            LSP_DSP_UNITS_PUBLIC void hipass_fft_set(float *mag, float f0, float slope, float sample_rate, size_t rank);
            LSP_DSP_UNITS_PUBLIC void hipass_fft_apply(float *mag, float f0, float slope, float sample_rate, size_t rank);
            LSP_DSP_UNITS_PUBLIC void lopass_fft_set(float *mag, float f0, float slope, float sample_rate, size_t rank);
            LSP_DSP_UNITS_PUBLIC void lopass_fft_apply(float *mag, float f0, float slope, float sample_rate, size_t rank);
        }
    }
}

#endif
