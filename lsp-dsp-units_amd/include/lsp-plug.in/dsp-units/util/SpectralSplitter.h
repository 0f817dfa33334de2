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
            // Binary layout: data members, order and inline members of the reference class
            // (include/lsp-plug.in/dsp-units/util/SpectralSplitter.h:58-88,165-206 of lsp-dsp-units 1.0.36); vHandlers holds the
            // bound functions exactly as there, pData owns them and the GPU bank.
            protected:
                typedef struct handler_t
                {
                    void                       *pObject;
                    void                       *pSubject;
                    spectral_splitter_func_t    pFunc;
                    spectral_splitter_sink_t    pSink;
                    float                      *vOutBuf;    // (device side: the bank's lines)
                } handler_t;

            protected:
                size_t                      nRank;
                size_t                      nMaxRank;
                ssize_t                     nUserChunkRank;
                size_t                      nChunkRank;
                float                       fPhase;
                float                      *vWnd;
                float                      *vInBuf;
                float                      *vFftBuf;
                float                      *vFftTmp;
                size_t                      nFrameSize;
                size_t                      nInOffset;
                bool                        bUpdate;
                handler_t                  *vHandlers;
                size_t                      nHandlers;
                size_t                      nBindings;
                uint8_t                    *pData;

            protected:
                struct impl_t;
                friend class FFTCrossover;
                impl_t         *impl() const            { return reinterpret_cast<impl_t *>(pData); }
                void            sync_ranks();           // nRank / nChunkRank as the bank has them
                // A handler whose spectral function is "multiply by 2^rank real gains" (FFTCrossover::spectral_func,
                // FFTCrossover.cpp:124-140) runs on the device: the gains go up once per change, no spectrum comes down.
                status_t        bind_gains(size_t id, void *object, void *subject, const float *gains, spectral_splitter_sink_t sink);
                void            set_gains(size_t id, const float *gains);

            public:
                explicit SpectralSplitter();
                SpectralSplitter(const SpectralSplitter &) = delete;
                SpectralSplitter(SpectralSplitter &&) = delete;
                SpectralSplitter & operator = (const SpectralSplitter &) = delete;
                SpectralSplitter & operator = (SpectralSplitter &&) = delete;
                ~SpectralSplitter();

                void            construct();            // valid on raw (e.g. zeroed) memory
                status_t        init(size_t max_rank, size_t handlers);
                void            destroy();

            public:
                status_t        bind(size_t id, void *object, void *subject, spectral_splitter_func_t func, spectral_splitter_sink_t sink);
                status_t        unbind(size_t id);
                void            unbind_all();
                bool            bound(size_t id) const;
                inline size_t   handlers() const            { return nHandlers;         }
                inline size_t   bindings() const            { return nBindings;         }
                inline bool     needs_update() const        { return bUpdate;           }
                void            update_settings();
                inline size_t   rank() const                { return nRank;             }
                inline size_t   max_rank() const            { return nMaxRank;          }
                inline ssize_t  chunk_rank() const          { return nChunkRank;        }
                inline float    phase() const               { return fPhase;            }
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
