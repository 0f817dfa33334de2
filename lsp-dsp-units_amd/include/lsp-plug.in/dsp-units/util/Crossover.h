// lsp::dspu::Crossover on the GPU library (one channel, host pointers, per-band handlers called with HOST data exactly
// as in the reference; the device-resident form for many channels is mi_crossover_bank_*).
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_UTIL_CROSSOVER_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_UTIL_CROSSOVER_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp-units/iface/IStateDumper.h>
#include <lsp-plug.in/dsp/dsp.h>

#include <sys/types.h>

namespace lsp
{
    namespace dspu
    {
        // data: `count` samples of band `band`, `first`: their offset inside the current process() call
        typedef void (* crossover_func_t)(void *object, void *subject, size_t band, const float *data, size_t first, size_t count);

        enum crossover_mode_t
        {
            CROSS_MODE_BT,      // bilinear transform
            CROSS_MODE_MT       // matched transform
        };

        enum crossover_slope_t
        {
            CROSS_SLOPE_OFF     = 0,
            CROSS_SLOPE_LR2     = 1,
            CROSS_SLOPE_LR4     = 2,
            CROSS_SLOPE_LR8     = 3,
            CROSS_SLOPE_LR12    = 4,
            CROSS_SLOPE_LR16    = 5,
            CROSS_SLOPE_LR20    = 6,
            CROSS_SLOPE_LR24    = 7,
            CROSS_SLOPE_LR28    = 8,
            CROSS_SLOPE_LR32    = 9
        };

        class LSP_DSP_UNITS_PUBLIC Crossover
        {
            // Binary layout: data members, order and inline members of the reference class
            // (include/lsp-plug.in/dsp-units/util/Crossover.h:149-163,201-216,323,352 of lsp-dsp-units 1.0.36).  The band and split
            // records of the reference hold Equalizer / Filter objects; here the filters live in the GPU bank behind pData
            // and the three list pointers stay NULL.
            protected:
                enum reconfigure_t
                {
                    R_GAIN          = 1 << 0,
                    R_SPLIT         = 1 << 1,
                    R_ALL           = R_GAIN | R_SPLIT
                };
                struct split_t;
                struct band_t;

            protected:
                uint32_t        nReconfigure;
                uint32_t        nSplits;
                uint32_t        nBufSize;
                uint32_t        nSampleRate;
                uint32_t        nPlanSize;

                band_t         *vBands;
                split_t        *vSplit;
                split_t       **vPlan;

                float          *vLpfBuf;        // (device: staging of the caller's block)
                float          *vHpfBuf;
                uint8_t        *pData;

            protected:
                struct impl_t;
                impl_t         *impl() const    { return reinterpret_cast<impl_t *>(pData); }
                void            sync_flags();   // nReconfigure as the bank sees it

            public:
                explicit Crossover();
                Crossover(const Crossover &) = delete;
                Crossover & operator = (const Crossover &) = delete;
                ~Crossover();

                void            construct();
                void            destroy();
                bool            init(size_t bands, size_t buf_size);

            public:
                inline size_t   num_bands() const                       { return nSplits+1;     }
                inline size_t   num_splits() const                      { return nSplits;       }
                inline size_t   max_buffer_size() const                 { return nBufSize;      }
                void            set_slope(size_t sp, size_t slope);
                ssize_t         get_slope(size_t sp) const;
                void            set_frequency(size_t sp, float freq);
                float           get_frequency(size_t sp) const;
                void            set_mode(size_t sp, crossover_mode_t mode);
                ssize_t         get_mode(size_t sp) const;
                void            set_gain(size_t band, float gain);
                float           get_gain(size_t band) const;
                float           get_band_start(size_t band);
                float           get_band_end(size_t band);
                bool            band_active(size_t band);
                bool            set_handler(size_t band, crossover_func_t func, void *object, void *subject);
                bool            unset_handler(size_t band);
                void            set_sample_rate(size_t sr);
                inline size_t   get_sample_rate()                       { return nSampleRate;   }
                bool            freq_chart(size_t band, float *re, float *im, const float *f, size_t count);
                bool            freq_chart(size_t band, float *c, const float *f, size_t count);
                void            reconfigure();
                inline bool     needs_reconfiguration() const           { return nReconfigure != 0; }
                void            process(const float *in, size_t samples);
                void            dump(IStateDumper *v) const;
        };
    }
}

#endif
