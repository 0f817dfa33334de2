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
            // Binary layout: data members, order and inline members of the reference class
            // (include/lsp-plug.in/dsp-units/util/MultiSpectralProcessor.h:50-80,178-218 of lsp-dsp-units 1.0.36).  vChannels
            // holds the bound pointers exactly as there (advanced by process()); pData owns them and the GPU bank.
            protected:
                typedef struct channel_t
                {
                    const float            *pIn;        // bound input, NULL: none
                    float                  *pOut;       // bound output, NULL: none
                    float                  *pInBuf;     // (device side: the bank's buffers)
                    float                  *pOutBuf;
                    float                  *pFftBuf;
                } channel_t;

            protected:
                uint32_t                    nChannels;
                uint32_t                    nRank;
                uint32_t                    nMaxRank;
                uint32_t                    nOffset;
                channel_t                  *vChannels;
                float                     **vFftBuf;
                float                      *pWnd;
                float                       fPhase;
                bool                        bUpdate;

                multi_spectral_processor_func_t pFunc;
                void                       *pObject;
                void                       *pSubject;

                uint8_t                    *pData;      // the channel records and the GPU state

            protected:
                struct impl_t;
                impl_t                     *impl() const    { return reinterpret_cast<impl_t *>(pData); }

            public:
                explicit MultiSpectralProcessor();
                MultiSpectralProcessor(const MultiSpectralProcessor &) = delete;
                MultiSpectralProcessor(MultiSpectralProcessor &&) = delete;
                MultiSpectralProcessor & operator = (const MultiSpectralProcessor &) = delete;
                MultiSpectralProcessor & operator = (MultiSpectralProcessor &&) = delete;
                ~MultiSpectralProcessor();

                void            construct();                // valid on raw (e.g. zeroed) memory
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
                inline bool     needs_update() const        { return bUpdate;           }
                void            update_settings();
                inline size_t   get_rank() const            { return nRank;             }
                inline float    phase() const               { return fPhase;            }
                void            set_phase(float phase);
                void            set_rank(size_t rank);
                inline size_t   latency() const             { return 1 << nRank;        }
                inline size_t   frame_size() const          { return 1 << (nRank - 1);  }
                void            process(size_t count);
                void            reset();
                size_t          remaining() const;
                void            dump(IStateDumper *v) const;
        };
    }
}

#endif
