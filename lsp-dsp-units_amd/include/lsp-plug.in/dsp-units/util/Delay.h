// lsp::dspu::Delay on the GPU library (one channel, host pointers; many channels: mi_delay_bank_*).
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_UTIL_DELAY_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_UTIL_DELAY_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/iface/IStateDumper.h>
#include <lsp-plug.in/dsp/dsp.h>

namespace lsp
{
    namespace dspu
    {
        class LSP_DSP_UNITS_PUBLIC Delay
        {
            private:
                struct impl_t;
                impl_t     *pImpl;

            public:
                explicit Delay();
                Delay(const Delay &) = delete;
                Delay & operator = (const Delay &) = delete;
                ~Delay();

                void    construct();
                void    destroy();

            public:
                bool    init(size_t max_size);
                void    append(const float *src, size_t count);
                void    process(float *dst, const float *src, size_t count);
                void    process(float *dst, const float *src, float gain, size_t count);
                void    process(float *dst, const float *src, const float *gain, size_t count);
                void    process_add(float *dst, const float *src, size_t count);
                void    process_add(float *dst, const float *src, float gain, size_t count);
                void    process_add(float *dst, const float *src, const float *gain, size_t count);
                void    process_ramping(float *dst, const float *src, size_t delay, size_t count);
                void    process_ramping(float *dst, const float *src, float gain, size_t delay, size_t count);
                void    process_ramping(float *dst, const float *src, const float *gain, size_t delay, size_t count);
                float   process(float src);
                float   process(float src, float gain);
                void    set_delay(size_t delay);
                size_t  get_delay() const;
                size_t  delay() const;
                void    clear();
                void    dump(IStateDumper *v) const;
        };
    }
}

#endif
