// lsp::dspu::RingBuffer on the GPU library (one channel, host pointers; many channels: mi_ring_bank_*).
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_UTIL_RINGBUFFER_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_UTIL_RINGBUFFER_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/iface/IStateDumper.h>
#include <lsp-plug.in/dsp/dsp.h>

namespace lsp
{
    namespace dspu
    {
        class LSP_DSP_UNITS_PUBLIC RingBuffer
        {
            private:
                struct impl_t;
                impl_t     *pImpl;

            public:
                explicit RingBuffer();
                RingBuffer(const RingBuffer &) = delete;
                RingBuffer & operator = (const RingBuffer &) = delete;
                ~RingBuffer();

                void            construct();
                bool            init(size_t size, float fill = 0.0f);
                void            destroy();

            public:
                size_t          append(const float *data, size_t count);
                void            append(float data);
                void            clear();
                void            fill(float value);
                float           get(size_t offset) const;
                size_t          get(float *dst, size_t offset, size_t count) const;
                // raw positions inside the buffer (the storage itself lives on the device: there is no data())
                size_t          read(float *dst, size_t position, size_t count) const;
                float           read(size_t position) const;
                float           lerp_get(float offset) const;
                size_t          size() const;
                size_t          head_position() const;
                size_t          tail_position(size_t offset) const;
                void            dump(IStateDumper *v) const;
        };
    }
}

#endif
