// lsp::dspu::RingBuffer on the GPU library (one channel, host pointers; many channels: mi_ring_bank_*).
//
// Binary layout: the reference's three data members in the reference's order (util/RingBuffer.h:38-40 of
// lsp-dsp-units 1.0.36; 16 bytes, LP64) and its inline size() / data() / head_position().  pData is the raw storage
// itself: pinned host memory that the device addresses through the same pointer, written by the bank's kernels and
// readable (and writable) by the host between calls, as data() promises; nCapacity and nHead are the ring's real
// capacity and head after every call.  The device bank's handle sits in a small header in front of that storage.
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
            protected:
                float      *pData;
                uint32_t    nCapacity;
                uint32_t    nHead;

            private:
                struct impl_t;
                impl_t         *impl() const;
                void            sync_head();

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
                inline float   *data()                  { return pData; }
                // raw positions inside the buffer
                size_t          read(float *dst, size_t position, size_t count) const;
                float           read(size_t position) const;
                float           lerp_get(float offset) const;
                inline size_t   size() const            { return nCapacity; }
                inline size_t   head_position() const   { return nHead; }
                size_t          tail_position(size_t offset) const;
                void            dump(IStateDumper *v) const;
        };
    }
}

#endif
