// Magnitude curves of the FFT crossover: lsp::dspu::crossover::* (reference: src/main/misc/fft_crossover.cpp:33-400,
// include/lsp-plug.in/dsp-units/misc/fft_crossover.h:47-154).  Host code, float arithmetic in the reference's order.
//
// Both filters meet at -6 dB at f0.  For slopes of -3 dB/oct and steeper the pass side is 1 - (f/f0)^(+-k)/2 and the stop
// side (f0/f)^(+-k)/2 with k = slope * 0.05 ln10 / ln2, which makes a high-pass and a low-pass at the same f0 add up to 1.
// Flatter slopes use a one-octave -6 dB/oct transition.  The *_apply forms multiply instead of assigning; where the
// assigned value would be exactly 1 they leave the gain alone, so "multiply by the curve" restates them.
#include "mi_dspu.h"

#include <cmath>
#include <cstddef>

namespace
{
    constexpr float XOVER_LEVEL     = 0.5f;
    constexpr float SLOPE_SCALE     = float((0.05f * M_LN10) / M_LN2);
    constexpr float SLOPE_SCALE_M6  = float((-0.3f * M_LN10) / M_LN2);

    struct hipass_curve
    {
        static constexpr float DC = 0.0f;
        static float flat(float f, float f0)
        {
            if (f <= f0)
                return XOVER_LEVEL;
            if (f >= f0 * 2.0f)
                return 1.0f;
            return expf(SLOPE_SCALE_M6 * logf(f0 / f)) * XOVER_LEVEL;
        }
        static float steep(float f, float f0, float k)
        {
            return (f >= f0) ? 1.0f - expf(k * logf(f / f0)) * XOVER_LEVEL : expf(k * logf(f0 / f)) * XOVER_LEVEL;
        }
    };

    struct lopass_curve
    {
        static constexpr float DC = 1.0f;
        static float flat(float f, float f0)
        {
            if (f >= f0)
                return XOVER_LEVEL;
            if (f <= f0 * 0.5f)
                return 1.0f;
            return expf(SLOPE_SCALE_M6 * logf(f / f0)) * XOVER_LEVEL;
        }
        static float steep(float f, float f0, float k)
        {
            return (f >= f0) ? expf(k * logf(f / f0)) * XOVER_LEVEL : 1.0f - expf(k * logf(f0 / f)) * XOVER_LEVEL;
        }
    };

    template <class C>
    inline float curve(float f, float f0, float slope)
    {
        return (slope > -3.0f) ? C::flat(f, f0) : C::steep(f, f0, slope * SLOPE_SCALE);
    }

    template <class C, bool APPLY>
    void on_list(float *gain, const float *vf, float f0, float slope, size_t count)
    {
        for (size_t i = 0; i < count; ++i)
        {
            const float v = curve<C>(vf[i], f0, slope);
            gain[i] = APPLY ? gain[i] * v : v;
        }
    }

    template <class C, bool APPLY>
    void on_bins(float *gain, float f0, float slope, float sample_rate, size_t rank)
    {
        const size_t n = size_t(1) << rank, half = n >> 1;
        const float kf = sample_rate / n;
        // bin 0: the high-pass pins it to 0 in both forms, the low-pass assigns 1 / leaves it alone
        if (!APPLY || C::DC == 0.0f)
            gain[0] = C::DC;
        for (size_t i = 1; i < n; ++i)
        {
            const float f = ((i <= half) ? i : n - i) * kf;
            const float v = curve<C>(f, f0, slope);
            gain[i] = APPLY ? gain[i] * v : v;
        }
    }
} // namespace

extern "C" {

float mi_crossover_hipass(float f, float f0, float slope) { return curve<hipass_curve>(f, f0, slope); }
float mi_crossover_lopass(float f, float f0, float slope) { return curve<lopass_curve>(f, f0, slope); }

void mi_crossover_hipass_set(float *gain, const float *f, float f0, float slope, size_t count)   { on_list<hipass_curve, false>(gain, f, f0, slope, count); }
void mi_crossover_hipass_apply(float *gain, const float *f, float f0, float slope, size_t count) { on_list<hipass_curve, true>(gain, f, f0, slope, count); }
void mi_crossover_lopass_set(float *gain, const float *f, float f0, float slope, size_t count)   { on_list<lopass_curve, false>(gain, f, f0, slope, count); }
void mi_crossover_lopass_apply(float *gain, const float *f, float f0, float slope, size_t count) { on_list<lopass_curve, true>(gain, f, f0, slope, count); }

void mi_crossover_hipass_fft_set(float *mag, float f0, float slope, float sample_rate, size_t rank)   { on_bins<hipass_curve, false>(mag, f0, slope, sample_rate, rank); }
void mi_crossover_hipass_fft_apply(float *mag, float f0, float slope, float sample_rate, size_t rank) { on_bins<hipass_curve, true>(mag, f0, slope, sample_rate, rank); }
void mi_crossover_lopass_fft_set(float *mag, float f0, float slope, float sample_rate, size_t rank)   { on_bins<lopass_curve, false>(mag, f0, slope, sample_rate, rank); }
void mi_crossover_lopass_fft_apply(float *mag, float f0, float slope, float sample_rate, size_t rank) { on_bins<lopass_curve, true>(mag, f0, slope, sample_rate, rank); }

} // extern "C"
