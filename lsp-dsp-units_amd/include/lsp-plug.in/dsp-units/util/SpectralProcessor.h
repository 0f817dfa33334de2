// lsp::dspu::SpectralProcessor on the GPU library (one channel, host pointers; the callback sees the spectrum
// in HOST memory exactly as in the reference -- the device-side callback lives in mi_spectral_bank_bind()).
//
// Binary layout: the reference's data members in the reference's order (util/SpectralProcessor.h:47-62 of
// lsp-dsp-units 1.0.36; 104 bytes, LP64) and its inline needs_update() / get_rank() / phase() / latency().
// nRank, nMaxRank, fPhase, bUpdate and the binding (pFunc, pObject, pSubject) are live; nOffset follows
// remaining(); the window and the three frame buffers live in device memory behind pData and stay NULL.
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
            protected:
                size_t                      nRank;
                size_t                      nMaxRank;
                float                       fPhase;
                float                      *pWnd;
                float                      *pOutBuf;
                float                      *pInBuf;
                float                      *pFftBuf;
                size_t                      nOffset;
                uint8_t                    *pData;      // here: the object's device-side state (opaque)
                bool                        bUpdate;

                spectral_processor_func_t   pFunc;
                void                       *pObject;
                void                       *pSubject;

            private:
                struct impl_t;
                inline impl_t  *impl() const                { return reinterpret_cast<impl_t *>(pData); }

            public:
                explicit SpectralProcessor();
                SpectralProcessor(const SpectralProcessor &) = delete;
                SpectralProcessor(SpectralProcessor &&) = delete;
                ~SpectralProcessor();

                SpectralProcessor & operator = (const SpectralProcessor &) = delete;
                SpectralProcessor & operator = (SpectralProcessor &&) = delete;

                void            construct();                // valid on raw (e.g. zeroed) memory
                bool            init(size_t max_rank);
                void            destroy();

            public:
                void            bind(spectral_processor_func_t func, void *object, void *subject);
                void            unbind();

                inline bool     needs_update() const        { return bUpdate;           }

                void            update_settings();

                inline size_t   get_rank() const            { return nRank;             }

                inline float    phase() const               { return fPhase;            }

                void            set_phase(float phase);
                void            set_rank(size_t rank);

                inline size_t   latency() const             { return 1 << nRank;        }

                void            process(float *dst, const float *src, size_t count);
                void            process(const float *src, size_t count);
                void            reset();
                size_t          remaining() const;
                void            dump(IStateDumper *v) const;
        };
    }
}

#endif
