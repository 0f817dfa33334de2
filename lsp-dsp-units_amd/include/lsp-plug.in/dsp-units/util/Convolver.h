// lsp::dspu::Convolver on the GPU library (one channel, host pointers; many channels: mi_convolver_bank_*).
//
// Binary layout: the reference's data members in the reference's order (util/Convolver.h:38-56 of lsp-dsp-units
// 1.0.36; 144 bytes, LP64) and its inline data_size() / rank().  nConvSize, nRank, nFrameSize, nDirectSize,
// nLevels and nBlocks describe the response the way the reference's init() computes them (Convolver.cpp:87-142);
// the six buffer pointers of the CPU path have no host storage here (the partitions, the frequency-domain delay
// line and the frame live in device memory behind vData) and stay NULL.
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
                float          *vDataBuffer;
                float          *vFrame;
                float          *vConvBuffer;
                float          *vTaskData;
                float          *vConvData;
                float          *vDirectData;

                size_t          nDataBufferSize;
                size_t          nDirectSize;            // taps convolved directly (128 or fewer)
                size_t          nFrameSize;             // input frame, 2^(rank-1)
                size_t          nFrameOff;
                size_t          nConvSize;              // taps of the response
                size_t          nLevels;                // doubling levels above the direct part
                size_t          nBlocks;                // equal-size tail blocks
                size_t          nBlocksDone;
                size_t          nRank;                  // rank in force
                size_t          nBlkInit;
                float           fBlkCoef;

                uint8_t        *vData;                  // here: the object's device-side state (opaque)

            private:
                struct impl_t;
                inline impl_t  *impl() const            { return reinterpret_cast<impl_t *>(vData); }

            public:
                explicit Convolver();
                Convolver(const Convolver &) = delete;
                Convolver(Convolver &&) = delete;
                ~Convolver();

                Convolver & operator = (const Convolver &) = delete;
                Convolver & operator = (Convolver &&) = delete;

                void    construct();                    // valid on raw (e.g. zeroed) memory
                void    destroy();

            public:
                bool    init(const float *data, size_t count, size_t rank, float phase);
                void    process(float *dst, const float *src, size_t count);

                inline size_t data_size() const             { return nConvSize;     }

                inline size_t rank() const                  { return nRank;         }

                void    dump(IStateDumper *v) const;
        };
    }
}

#endif
