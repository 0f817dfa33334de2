// lsp::dspu::windows: window generators (host memory), enumerators identical to mi_window.
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_MISC_WINDOWS_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_MISC_WINDOWS_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <mi_dspu.h>
#include <cstddef>

namespace lsp
{
    namespace dspu
    {
        namespace windows
        {
            enum window_t
            {
                HANN = MI_WINDOW_HANN, HAMMING, BLACKMAN, LANCZOS, GAUSSIAN, POISSON, PARZEN, TUKEY, WELCH, NUTTALL,
                BLACKMAN_NUTTALL, BLACKMAN_HARRIS, HANN_POISSON, BARTLETT_HANN, BARTLETT_FEJER, TRIANGULAR,
                RECTANGULAR, FLAT_TOP, COSINE, SQR_COSINE, CUBIC,
                TOTAL, FIRST = HANN, LAST = TOTAL - 1
            };

            LSP_DSP_UNITS_PUBLIC void window(float *dst, size_t n, window_t type);

            // the named windows (misc/windows.h:64-160 of the reference): exported, like there
            #define MI_WND(fn) LSP_DSP_UNITS_PUBLIC void fn(float *dst, size_t n);
            MI_WND(hann) MI_WND(hamming) MI_WND(blackman) MI_WND(lanczos) MI_WND(gaussian) MI_WND(poisson) MI_WND(parzen)
            MI_WND(tukey) MI_WND(welch) MI_WND(nuttall) MI_WND(blackman_nuttall) MI_WND(blackman_harris) MI_WND(hann_poisson)
            MI_WND(bartlett_hann) MI_WND(bartlett_fejer) MI_WND(triangular) MI_WND(rectangular) MI_WND(flat_top) MI_WND(cosine)
            MI_WND(sqr_cosine) MI_WND(cubic)
            #undef MI_WND

            // the parameterised families (misc/windows.h:71,86,95,101,113,128,134,143,146,155 of the reference)
            LSP_DSP_UNITS_PUBLIC void triangular_general(float *dst, size_t n, int dn);
            LSP_DSP_UNITS_PUBLIC void hamming_general(float *dst, size_t n, float a, float b);
            LSP_DSP_UNITS_PUBLIC void blackman_general(float *dst, size_t n, float a);
            LSP_DSP_UNITS_PUBLIC void nuttall_general(float *dst, size_t n, float a0, float a1, float a2, float a3);
            // (the reference's header spells this one nutall_general, its source nuttall_general: both are provided)
            LSP_DSP_UNITS_PUBLIC void nutall_general(float *dst, size_t n, float a0, float a1, float a2, float a3);
            LSP_DSP_UNITS_PUBLIC void flat_top_general(float *dst, size_t n, float a0, float a1, float a2, float a3, float a4);
            LSP_DSP_UNITS_PUBLIC void gaussian_general(float *dst, size_t n, float s);
            LSP_DSP_UNITS_PUBLIC void poisson_general(float *dst, size_t n, float t);
            LSP_DSP_UNITS_PUBLIC void bartlett_hann_general(float *dst, size_t n, float a0, float a1, float a2);
            LSP_DSP_UNITS_PUBLIC void hann_poisson_general(float *dst, size_t n, float a);
            LSP_DSP_UNITS_PUBLIC void tukey_general(float *dst, size_t n, float a);
        }
    }
}

#endif
