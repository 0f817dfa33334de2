// Host-side filter designer: filter_params_t -> analog prototype cascades -> digital biquad sections.
// Product code (C++), the counterpart of lsp::dspu::Filter::rebuild() and friends
// (reference: src/main/filters/Filter.cpp:208-403, 722-2416).  Runs on the CPU by design: it is
// O(#sections) per parameter change, its output feeds the device tables.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "mi_dspu.h"

namespace mi
{
    enum filter_mode
    {
        FM_BYPASS   = 0,
        FM_BILINEAR = 1,
        FM_MATCHED  = 2,
        FM_APO      = 3
    };

    // Numerator t[] / denominator b[] of one second-order analog (or, for FM_APO, digital) cascade.
    struct cascade
    {
        float t[4];
        float b[4];
    };

    struct design
    {
        int                         mode = FM_BYPASS;
        mi_filter_params_t          params;         // after limit()
        uint32_t                    sample_rate = 0;
        std::vector<cascade>        cascades;       // what freq_chart() evaluates
        std::vector<mi_biquad_x1_t> sections;       // what add_chain() receives
    };

    constexpr uint32_t CHAINS_MAX = 0x80;          // FILTER_CHAINS_MAX of filters/common.h

    // Filter::limit (Filter.cpp:161-167)
    void limit_params(mi_filter_params_t *fp, uint32_t sample_rate);
    // Filter::rebuild (Filter.cpp:208-403) without the bank bookkeeping
    void design_filter(design *out, const mi_filter_params_t *params, uint32_t sample_rate);
    // Filter::freq_chart, packed-complex form (Filter.cpp:602-696): c = {re0, im0, re1, im1, ...}
    void freq_chart(const design &d, float *c, const float *f, size_t count);
} // namespace mi
