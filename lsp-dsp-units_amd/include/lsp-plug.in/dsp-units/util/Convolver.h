// lsp::dspu::Convolver on the GPU library (one channel, host pointers; many channels: mi_convolver_bank_*).
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_UTIL_CONVOLVER_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_UTIL_CONVOLVER_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/iface/IStateDumper.h>
#include <lsp-plug.in/dsp/dsp.h>

#define CONVOLVER_RANK_MIN          8
#define CONVOLVER_RANK_MAX          16

namespace lsp
{
    namespace dspu
    {
        class LSP_DSP_UNITS_PUBLIC Convolver
        {
            private:
                struct impl_t;
                impl_t     *pImpl;

            public:
                explicit Convolver();
                Convolver(const Convolver &) = delete;
                Convolver & operator = (const Convolver &) = delete;
                ~Convolver();

                void    construct();
                void    destroy();

            public:
                bool    init(const float *data, size_t count, size_t rank, float phase);
                void    process(float *dst, const float *src, size_t count);
                size_t  data_size() const;
                size_t  rank() const;
                void    dump(IStateDumper *v) const;
        };
    }
}

#endif
