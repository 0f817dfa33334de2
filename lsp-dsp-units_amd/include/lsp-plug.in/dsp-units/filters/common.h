// Filter vocabulary of lsp::dspu: parameter record and filter types (values match the C-ABI's mi_filter_type).
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_FILTERS_COMMON_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_FILTERS_COMMON_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <lsp-plug.in/dsp/dsp.h>
#include <mi_dspu.h>

#define FILTER_BUFFER_MAX           0x1000
#define FILTER_RANK_MIN             8
#define FILTER_RANK_MAX             12
#define FILTER_CONVOLUTION_MAX      (1 << FILTER_RANK_MAX)
#define FILTER_CHAINS_MAX           0x80U

namespace lsp
{
    namespace dspu
    {
        // same enumerators, same values as mi_filter_type: FLT_x == MI_FLT_x
        enum filter_type_t
        {
            #define MI_FLT(name) FLT_##name = MI_FLT_##name
            MI_FLT(NONE), MI_FLT(BT_AMPLIFIER), MI_FLT(MT_AMPLIFIER),
            MI_FLT(BT_RLC_LOPASS), MI_FLT(MT_RLC_LOPASS), MI_FLT(BT_RLC_HIPASS), MI_FLT(MT_RLC_HIPASS),
            MI_FLT(BT_RLC_LOSHELF), MI_FLT(MT_RLC_LOSHELF), MI_FLT(BT_RLC_HISHELF), MI_FLT(MT_RLC_HISHELF),
            MI_FLT(BT_RLC_BELL), MI_FLT(MT_RLC_BELL), MI_FLT(BT_RLC_RESONANCE), MI_FLT(MT_RLC_RESONANCE),
            MI_FLT(BT_RLC_NOTCH), MI_FLT(MT_RLC_NOTCH), MI_FLT(BT_RLC_ALLPASS), MI_FLT(MT_RLC_ALLPASS),
            MI_FLT(BT_RLC_ALLPASS2), MI_FLT(MT_RLC_ALLPASS2), MI_FLT(BT_RLC_LADDERPASS), MI_FLT(MT_RLC_LADDERPASS),
            MI_FLT(BT_RLC_LADDERREJ), MI_FLT(MT_RLC_LADDERREJ), MI_FLT(BT_RLC_BANDPASS), MI_FLT(MT_RLC_BANDPASS),
            MI_FLT(BT_RLC_ENVELOPE), MI_FLT(MT_RLC_ENVELOPE),
            MI_FLT(BT_BWC_LOPASS), MI_FLT(MT_BWC_LOPASS), MI_FLT(BT_BWC_HIPASS), MI_FLT(MT_BWC_HIPASS),
            MI_FLT(BT_BWC_LOSHELF), MI_FLT(MT_BWC_LOSHELF), MI_FLT(BT_BWC_HISHELF), MI_FLT(MT_BWC_HISHELF),
            MI_FLT(BT_BWC_BELL), MI_FLT(MT_BWC_BELL), MI_FLT(BT_BWC_LADDERPASS), MI_FLT(MT_BWC_LADDERPASS),
            MI_FLT(BT_BWC_LADDERREJ), MI_FLT(MT_BWC_LADDERREJ), MI_FLT(BT_BWC_BANDPASS), MI_FLT(MT_BWC_BANDPASS),
            MI_FLT(BT_BWC_ALLPASS), MI_FLT(MT_BWC_ALLPASS),
            MI_FLT(BT_LRX_LOPASS), MI_FLT(MT_LRX_LOPASS), MI_FLT(BT_LRX_HIPASS), MI_FLT(MT_LRX_HIPASS),
            MI_FLT(BT_LRX_LOSHELF), MI_FLT(MT_LRX_LOSHELF), MI_FLT(BT_LRX_HISHELF), MI_FLT(MT_LRX_HISHELF),
            MI_FLT(BT_LRX_BELL), MI_FLT(MT_LRX_BELL), MI_FLT(BT_LRX_LADDERPASS), MI_FLT(MT_LRX_LADDERPASS),
            MI_FLT(BT_LRX_LADDERREJ), MI_FLT(MT_LRX_LADDERREJ), MI_FLT(BT_LRX_BANDPASS), MI_FLT(MT_LRX_BANDPASS),
            MI_FLT(BT_LRX_ALLPASS), MI_FLT(MT_LRX_ALLPASS),
            MI_FLT(DR_APO_LOPASS), MI_FLT(DR_APO_HIPASS), MI_FLT(DR_APO_BANDPASS), MI_FLT(DR_APO_NOTCH),
            MI_FLT(DR_APO_ALLPASS), MI_FLT(DR_APO_ALLPASS2), MI_FLT(DR_APO_PEAKING), MI_FLT(DR_APO_LOSHELF),
            MI_FLT(DR_APO_HISHELF), MI_FLT(DR_APO_LADDERPASS), MI_FLT(DR_APO_LADDERREJ),
            MI_FLT(A_WEIGHTED), MI_FLT(B_WEIGHTED), MI_FLT(C_WEIGHTED), MI_FLT(D_WEIGHTED), MI_FLT(K_WEIGHTED)
            #undef MI_FLT
        };

        // The reference's own record (filters/common.h:137-145), a type of this namespace so that the member functions that
        // take it come out of the library under the reference's mangled names; field for field the record of the C-ABI.
        typedef struct filter_params_t
        {
            uint32_t    nType;      // Filter class
            uint32_t    nSlope;     // Filter slope
            float       fFreq;      // Frequency
            float       fFreq2;     // Second frequency (for bandpass/allpass2 filter)
            float       fGain;      // Gain
            float       fQuality;   // Quality factor
        } filter_params_t;
        static_assert(sizeof(filter_params_t) == sizeof(mi_filter_params_t) && sizeof(filter_params_t) == 24,
                      "filter_params_t and mi_filter_params_t are the same record");
    }
}

#endif
