// C-ABI entry points of the host-side filter designer (include/mi_dspu.h).
#include "../mi_common.h"
#include "filter_design.h"

extern "C" {

int mi_filter_design(const mi_filter_params_t *params, uint32_t sample_rate,
                     mi_biquad_x1_t *chains, uint32_t max_chains, uint32_t *n_chains,
                     mi_filter_cascade_t *cascades, uint32_t max_cascades, uint32_t *n_cascades, int *mode)
{
    MI_REQUIRE(params != nullptr, MI_EINVAL, "mi_filter_design: NULL parameters");
    MI_REQUIRE(sample_rate > 0, MI_EINVAL, "mi_filter_design: sample rate must be > 0");
    mi::design d;
    d.cascades.reserve(mi::CHAINS_MAX + 1);
    mi::design_filter(&d, params, sample_rate);
    if (n_chains != nullptr)
        *n_chains = uint32_t(d.sections.size());
    if (n_cascades != nullptr)
        *n_cascades = uint32_t(d.cascades.size());
    if (mode != nullptr)
        *mode = d.mode;
    if (chains != nullptr)
        for (size_t i = 0; i < d.sections.size() && i < max_chains; ++i)
            chains[i] = d.sections[i];
    if (cascades != nullptr)
        for (size_t i = 0; i < d.cascades.size() && i < max_cascades; ++i)
        {
            for (int k = 0; k < 4; ++k)
            {
                cascades[i].t[k] = d.cascades[i].t[k];
                cascades[i].b[k] = d.cascades[i].b[k];
            }
        }
    return MI_OK;
}

int mi_filter_limit(mi_filter_params_t *params, uint32_t sample_rate)
{
    MI_REQUIRE(params != nullptr, MI_EINVAL, "mi_filter_limit: NULL parameters");
    mi::limit_params(params, sample_rate);
    return MI_OK;
}

int mi_filter_freq_chart(const mi_filter_params_t *params, uint32_t sample_rate, float *c, const float *f, size_t count)
{
    MI_REQUIRE(params != nullptr && (count == 0 || (c != nullptr && f != nullptr)), MI_EINVAL,
               "mi_filter_freq_chart: bad argument");
    MI_REQUIRE(sample_rate > 0, MI_EINVAL, "mi_filter_freq_chart: sample rate must be > 0");
    mi::design d;
    d.cascades.reserve(mi::CHAINS_MAX + 1);
    mi::design_filter(&d, params, sample_rate);
    mi::freq_chart(d, c, f, count);
    return MI_OK;
}

} // extern "C"
