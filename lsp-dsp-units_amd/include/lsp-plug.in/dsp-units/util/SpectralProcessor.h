// lsp::dspu::SpectralProcessor on the GPU library (one channel, host pointers; the callback sees the spectrum
// in HOST memory exactly as in the reference -- the device-side callback lives in mi_spectral_bank_bind()).
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_UTIL_SPECTRALPROCESSOR_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_UTIL_SPECTRALPROCESSOR_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/iface/IStateDumper.h>
#include <lsp-plug.in/dsp/dsp.h>

namespace lsp
{
    namespace dspu
    {
        // spectrum: 2^rank packed complex bins (re, im interleaved), modified in place
        typedef void (* spectral_processor_func_t)(void *object, void *subject, float *spectrum, size_t rank);

        class LSP_DSP_UNITS_PUBLIC SpectralProcessor
        {
            private:
                struct impl_t;
                impl_t     *pImpl;

            public:
                explicit SpectralProcessor();
                SpectralProcessor(const SpectralProcessor &) = delete;
                SpectralProcessor & operator = (const SpectralProcessor &) = delete;
                ~SpectralProcessor();

                void            construct();
                bool            init(size_t max_rank);
                void            destroy();

            public:
                void            bind(spectral_processor_func_t func, void *object, void *subject);
                void            unbind();
                bool            needs_update() const;
                void            update_settings();
                size_t          get_rank() const;
                float           phase() const;
                void            set_phase(float phase);
                void            set_rank(size_t rank);
                size_t          latency() const;
                void            process(float *dst, const float *src, size_t count);
                void            process(const float *src, size_t count);
                void            reset();
                size_t          remaining() const;
                void            dump(IStateDumper *v) const;
        };
    }
}

#endif
