// lsp::dspu::MultiSpectralProcessor on the GPU library: `channels` channels with bound host pointers that advance with
// process(count); the handler sees one HOST spectrum pointer per channel (NULL for channels without an input), exactly
// as in the reference -- the device-side form is mi_spectral_bank_bind() + mi_spectral_bank_bind_channels().
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_UTIL_MULTISPECTRALPROCESSOR_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_UTIL_MULTISPECTRALPROCESSOR_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/iface/IStateDumper.h>
#include <lsp-plug.in/dsp/dsp.h>

namespace lsp
{
    namespace dspu
    {
        // spectrum[i]: 2^rank packed complex bins of channel i (re, im interleaved), modified in place; NULL: no input bound
        typedef void (* multi_spectral_processor_func_t)(void *object, void *subject, float * const * spectrum, size_t rank);

        class LSP_DSP_UNITS_PUBLIC MultiSpectralProcessor
        {
            private:
                struct impl_t;
                impl_t     *pImpl;

            public:
                explicit MultiSpectralProcessor();
                MultiSpectralProcessor(const MultiSpectralProcessor &) = delete;
                MultiSpectralProcessor & operator = (const MultiSpectralProcessor &) = delete;
                ~MultiSpectralProcessor();

                void            construct();
                bool            init(size_t channels, size_t max_rank);
                void            destroy();

            public:
                void            bind_handler(multi_spectral_processor_func_t func, void *object, void *subject);
                void            unbind_handler();
                status_t        bind(size_t index, float *out, const float *in);
                status_t        bind_in(size_t index, const float *in);
                status_t        bind_out(size_t index, float *out);
                status_t        unbind(size_t index);
                status_t        unbind_in(size_t index);
                status_t        unbind_out(size_t index);
                void            unbind_all();
                bool            needs_update() const;
                void            update_settings();
                size_t          get_rank() const;
                float           phase() const;
                void            set_phase(float phase);
                void            set_rank(size_t rank);
                size_t          latency() const;
                size_t          frame_size() const;
                void            process(size_t count);
                void            reset();
                size_t          remaining() const;
                void            dump(IStateDumper *v) const;
        };
    }
}

#endif
