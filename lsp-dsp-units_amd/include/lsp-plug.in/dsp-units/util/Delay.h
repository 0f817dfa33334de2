// lsp::dspu::Delay on the GPU library (one channel, host pointers; many channels: mi_delay_bank_*).
//
// Binary layout: the reference's five data members in the reference's order (util/Delay.h:38-42 of lsp-dsp-units
// 1.0.36; 24 bytes, LP64) and its inline get_delay() / delay().  nHead, nTail, nDelay and nSize are the line's real
// positions after every call (the device bank keeps the reference's absolute index arithmetic, Delay.cpp:101,434);
// the line itself is in device memory, so pBuffer carries the object's device-side state, not host samples.
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
            protected:
                float      *pBuffer;            // here: the object's device-side state (opaque, never host samples)
                uint32_t    nHead;
                uint32_t    nTail;
                uint32_t    nDelay;
                uint32_t    nSize;

            private:
                struct impl_t;
                inline impl_t  *impl() const    { return reinterpret_cast<impl_t *>(pBuffer); }
                void            sync_positions();

            public:
                explicit Delay();
                Delay(const Delay &) = delete;
                Delay(Delay &&) = delete;
                ~Delay();

                Delay & operator = (const Delay &) = delete;
                Delay & operator = (Delay &&) = delete;

                void    construct();                    // valid on raw (e.g. zeroed) memory
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
                inline size_t get_delay() const { return nDelay; }
                inline size_t delay() const     { return nDelay; }
                void    clear();
                void    dump(IStateDumper *v) const;
        };
    }
}

#endif
