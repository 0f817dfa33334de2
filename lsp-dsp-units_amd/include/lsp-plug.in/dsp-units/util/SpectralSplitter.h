// lsp::dspu::SpectralSplitter on the GPU library (one channel; spectral functions and sinks are called with HOST data
// exactly as in the reference; the device-resident form for many channels is mi_splitter_bank_*).
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_UTIL_SPECTRALSPLITTER_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_UTIL_SPECTRALSPLITTER_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/iface/IStateDumper.h>
#include <lsp-plug.in/dsp/dsp.h>

#include <sys/types.h>

namespace lsp
{
    namespace dspu
    {
        // out <- in: 2^rank packed complex bins
        typedef void (* spectral_splitter_func_t)(void *object, void *subject, float *out, const float *in, size_t rank);
        // samples: `count` finished samples; first: their offset inside the current process() call
        typedef void (* spectral_splitter_sink_t)(void *object, void *subject, const float *samples, size_t first, size_t count);

        class LSP_DSP_UNITS_PUBLIC SpectralSplitter
        {
            private:
                struct impl_t;
                impl_t     *pImpl;

            public:
                explicit SpectralSplitter();
                SpectralSplitter(const SpectralSplitter &) = delete;
                SpectralSplitter & operator = (const SpectralSplitter &) = delete;
                ~SpectralSplitter();

                void            construct();
                status_t        init(size_t max_rank, size_t handlers);
                void            destroy();

            public:
                status_t        bind(size_t id, void *object, void *subject, spectral_splitter_func_t func, spectral_splitter_sink_t sink);
                status_t        unbind(size_t id);
                void            unbind_all();
                bool            bound(size_t id) const;
                size_t          handlers() const;
                size_t          bindings() const;
                bool            needs_update() const;
                void            update_settings();
                size_t          rank() const;
                size_t          max_rank() const;
                ssize_t         chunk_rank() const;
                float           phase() const;
                void            set_phase(float phase);
                void            set_rank(size_t rank);
                void            set_chunk_rank(ssize_t rank);
                size_t          latency() const;
                void            process(const float *src, size_t count);
                void            clear();
                void            dump(IStateDumper *v) const;
        };
    }
}

#endif
