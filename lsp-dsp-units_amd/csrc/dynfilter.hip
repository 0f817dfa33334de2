// Dynamic filter bank for gfx950: the GPU side of lsp::dspu::DynamicFilters (reference:
// src/main/filters/DynamicFilters.cpp:127-181 set_params, :204-318 process, :320-1738 the cascade builders,
// :1774-1970 freq_chart, include/lsp-plug.in/dsp-units/filters/DynamicFilters.h).
//
// A dynamic filter is a filter whose GAIN changes from sample to sample (dynamic equalisers, de-essers): for every sample
// the reference rebuilds the analog cascades from the gain value, transforms them to digital sections
// (dsp::bilinear_transform_x* / matched_transform_x*) and applies them to that sample (dsp::dyn_biquad_process_x*).
// Those three primitives belong to the un-vendored lsp-dsp-lib; they are restated from their published algorithm:
// the transforms are Filter::bilinear_transform / matched_transform (Filter.cpp:2225-2267, 2291-2416) evaluated per
// sample in float, the processing is the transposed direct form II recurrence with one coefficient set per sample.
// The reference lays its cascades out diagonally for lsp-dsp-lib's section-pipelined x8/x4/x2 kernels (:320-369 and
// the `dst[(nc+1)*j]` walks); which numbers cascade J of sample n holds does not depend on that layout, and
// dyn_cascade() below returns exactly those numbers.
//
// Kernel: one wave per channel walks the call in blocks of 64 lanes x 16 samples.  Per cascade every lane builds the 16
// coefficient sets of its chunk from the gains (gain-independent parts are hoisted), composes the chunk's affine state
// map s -> M s + v (time-varying, so the maps are per lane, not per section as in biquad.hip), an inclusive scan of the
// maps over the wave gives every chunk its start state, and the exact recurrence then runs over the chunk with the
// per-sample coefficients.  Samples stay in registers from cascade to cascade.
#include "mi_common.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <new>
#include <type_traits>
#include <vector>

namespace
{
    constexpr float kPi = 3.14159265358979323846f, kPi2 = 1.57079632679489661923f, kTwoPi = 6.28318530717958647692f;
    constexpr int CHAINS_MAX = 0x80;            // FILTER_CHAINS_MAX (filters/common.h:28)

    // bilinear prototypes sit on odd enumerators, their matched-Z twins right after them (filters/common.h:38-135)
    __host__ __device__ inline uint32_t base_type(uint32_t t) { return (t & 1) ? t : t - 1; }

    // Number of cascades of a filter: the sum of build_filter_bank's pack sizes (DynamicFilters.cpp:625-1738)
    __host__ __device__ inline uint32_t cascade_count(uint32_t type, uint32_t slope)
    {
        uint32_t n;
        switch (base_type(type))
        {
            case MI_FLT_BT_AMPLIFIER: case MI_FLT_BT_RLC_NOTCH:                         n = 1; break;
            case MI_FLT_BT_RLC_LOPASS: case MI_FLT_BT_RLC_HIPASS:
            case MI_FLT_BT_BWC_LOPASS: case MI_FLT_BT_BWC_HIPASS:                       n = (slope >> 1) + (slope & 1); break;
            case MI_FLT_BT_RLC_LOSHELF: case MI_FLT_BT_RLC_HISHELF: case MI_FLT_BT_RLC_BANDPASS:
            case MI_FLT_BT_RLC_BELL: case MI_FLT_BT_RLC_RESONANCE:
            case MI_FLT_BT_BWC_LOSHELF: case MI_FLT_BT_BWC_HISHELF:                     n = slope; break;
            case MI_FLT_BT_RLC_LADDERPASS: case MI_FLT_BT_RLC_LADDERREJ:
            case MI_FLT_BT_BWC_LADDERPASS: case MI_FLT_BT_BWC_LADDERREJ:
            case MI_FLT_BT_BWC_BELL: case MI_FLT_BT_BWC_BANDPASS:
            case MI_FLT_BT_LRX_LOPASS: case MI_FLT_BT_LRX_HIPASS:
            case MI_FLT_BT_LRX_LOSHELF: case MI_FLT_BT_LRX_HISHELF:                     n = 2 * slope; break;
            case MI_FLT_BT_LRX_LADDERPASS: case MI_FLT_BT_LRX_LADDERREJ:
            case MI_FLT_BT_LRX_BELL: case MI_FLT_BT_LRX_BANDPASS:                       n = 4 * slope; break;
            default:                                                                    n = 0; break;   // incl. RLC_ENVELOPE, see header
        }
        return (n < uint32_t(CHAINS_MAX)) ? n : uint32_t(CHAINS_MAX);
    }

    struct dyn_params            // what a builder needs besides the gain (set_params result, DynamicFilters.cpp:127-181)
    {
        uint32_t base, slope;
        float    xf;             // fFreq2 AFTER the transformation of set_params (:170-178)
        float    Q;
    };

    __host__ __device__ inline void set3(float *d, float a, float b, float c) { d[0] = a; d[1] = b; d[2] = c; }
    __host__ __device__ inline void scale3(float *d, float g) { d[0] *= g; d[1] *= g; d[2] *= g; }
    // a / b.  Host: the division.  Device: reciprocal, one Newton step and one correction of the quotient -- six
    // instructions instead of the ten (and two mode switches) of the compiler's IEEE sequence, whose scaling steps serve
    // operands next to the ends of the exponent range; gains, their roots and the prototype's polynomials are nowhere near.
    // The same quotient bit for bit on 2^24 random pairs from [1e-4, 1e4]^2, half of them reciprocals
    // (tests/experiments/dyn_div_probe.hip, profiles/r03_experiments/dynfilter_per_type.txt); zero, infinite and NaN operands
    // get the IEEE result (tests/test_dynfilter_gpu.py::test_gain_samples_of_zero).
    __host__ __device__ __attribute__((always_inline)) inline float dv(float a, float b)
    {
#ifdef __HIP_DEVICE_COMPILE__
        float r = __builtin_amdgcn_rcpf(b);
        r = fmaf(fmaf(-b, r, 1.0f), r, r);
        const float q = a * r;
        const float v = fmaf(fmaf(-b, q, a), r, q);
        // Operands at the ends of the range -- a gain sample of exactly 0 makes logf(g) = -inf, 1 / g = inf -- turn the
        // correction steps into inf - inf.  The reference computes expf(-inf) = 0 there and goes on with a finite filter
        // (DynamicFilters.cpp:964-980): the IEEE result for zero, infinite and NaN operands is put in place by the
        // instruction the compiler's own division ends with (v_div_fixup_f32: the quotient as it is for ordinary operands)
        // -- one instruction instead of a class test and a branch around a second division per call (81 of them in the
        // bell's kernel).
        return __builtin_amdgcn_div_fixupf(v, b, a);
#else
        return a / b;
#endif
    }

    __host__ __device__ inline float iroot(float x, float n) { return expf(dv(logf(x), n)); }      // dsp::irootf
    // a polynomial on one side and its mirror image on the other (the shelving builders); the VALUES are selected, not
    // the destination: a selected pointer (also one the optimiser makes out of two mirrored branches) keeps t[] and b[]
    // in scratch memory on the device
    __host__ __device__ inline void set_mirrored(bool t_first, float *t, float *b, float a0, float a1, float a2)
    {
        const float lo = t_first ? a0 : a2, hi = t_first ? a2 : a0;
        set3(t, lo, a1, hi);
        set3(b, hi, a1, lo);
    }

    // Analog cascade J (global index) of a filter at gain g: numerator t[3], denominator b[3].
    // Returned BY VALUE from local arrays: with caller-owned arrays the optimiser sinks the branches' stores through
    // phi'd pointers before the arrays are split into registers, and t[] / b[] then live in scratch memory on the device.
    struct cascade { float t[3], b[3]; };

    __host__ __device__ __attribute__((always_inline)) inline cascade dyn_cascade(const dyn_params &p, uint32_t J, float g)
    {
        cascade cs_;
        float *const t = cs_.t, *const b = cs_.b;
        const float Q = p.Q, xf = p.xf;
        const uint32_t slope = p.slope;
        switch (p.base)
        {
            case MI_FLT_BT_AMPLIFIER:                                                   // :633-657
                set3(t, g, 0.0f, 0.0f); set3(b, 1.0f, 0.0f, 0.0f);
                break;

            case MI_FLT_BT_RLC_LOPASS: case MI_FLT_BT_RLC_HIPASS:                       // :660-732
            {
                const bool lo = p.base == MI_FLT_BT_RLC_LOPASS;
                if (J == 0 && (slope & 1))
                {
                    set3(b, 1.0f, 1.0f, 0.0f);
                    set3(t, lo ? g : 0.0f, lo ? 0.0f : g, 0.0f);
                    break;
                }
                set3(b, 1.0f, dv(2.0f, 1.0f + Q), 1.0f);
                set3(t, lo ? 1.0f : 0.0f, 0.0f, lo ? 0.0f : 1.0f);
                if (J == 0)
                    scale3(t, g);                                                       // "Patch volume"
                break;
            }
            case MI_FLT_BT_RLC_LOSHELF: case MI_FLT_BT_RLC_HISHELF:                     // :734-785
            {
                const float gs = sqrtf(g), fg = expf(dv(logf(gs), float(slope * 2))), k = dv(2.0f, 1.0f + Q);
                set_mirrored(p.base == MI_FLT_BT_RLC_LOSHELF, t, b, fg, k, dv(1.0f, fg));
                if (J == 0)
                    scale3(t, gs);
                break;
            }
            case MI_FLT_BT_RLC_LADDERPASS: case MI_FLT_BT_RLC_LADDERREJ:                // :790-876
            {
                const bool rej = p.base == MI_FLT_BT_RLC_LADDERREJ;
                const float s2 = float(slope * 2), sq = sqrtf(g), isq = sqrtf(dv(1.0f, g));
                float gain;
                if (J & 1)                                                              // second shelf, always a hi-shelf
                {
                    gain = rej ? sq : isq;
                    const float fg = expf(dv(logf(gain), s2)), k = dv(2.0f * xf, 1.0f + Q);
                    set3(b, fg, k, dv(xf * xf, fg));
                    set3(t, dv(1.0f, fg), k, fg * xf * xf);
                }
                else
                {
                    const float gain1 = rej ? isq : sq, gain2 = rej ? sq : isq;
                    const float fg = expf(dv(logf(rej ? gain2 : gain1), s2)), k = dv(2.0f, 1.0f + Q);
                    gain = rej ? gain2 : gain1;
                    set_mirrored(rej, t, b, fg, k, dv(1.0f, fg));
                }
                if ((J >> 1) == 0)
                    scale3(t, gain);
                break;
            }
            case MI_FLT_BT_RLC_BANDPASS:                                                // :878-911
            {
                const float f2 = dv(1.0f, xf), k = dv(1.0f + f2, 1.0f + Q);
                set3(t, 0.0f, (J == 0) ? expf(float(slope) * logf(k)) * g : 1.0f, 0.0f);
                set3(b, f2, k, 1.0f);
                break;
            }
            case MI_FLT_BT_RLC_BELL: case MI_FLT_BT_RLC_RESONANCE:                      // :913-991
            {
                const float fg = expf(dv(logf(g), float(slope)));
#ifdef __HIP_DEVICE_COMPILE__
                // sin(atan(x)) = x / sqrt(1 + x^2), cos(atan(x)) = 1 / sqrt(1 + x^2): the same numbers to the last bits of
                // float32 without the two transcendental calls per sample (the host keeps the reference's expression)
                const float rs = dv(1.0f, sqrtf(fmaf(fg, fg, 1.0f)));
                const float tsin = fg * rs, tcos = rs;
#else
                const float tsin = sinf(atanf(fg)), tcos = sqrtf(1.0f - tsin * tsin);
#endif
                const float k = (p.base == MI_FLT_BT_RLC_BELL) ? dv(2.0f * (dv(1.0f, fg) + fg), 1.0f + dv(2.0f * Q, float(slope)))
                                                               : dv(2.0f, 1.0f + Q);
                set3(t, 1.0f, k * tsin, 1.0f);
                set3(b, 1.0f, k * tcos, 1.0f);
                break;
            }
            case MI_FLT_BT_RLC_NOTCH:                                                   // :993-1020
                set3(t, g, 0.0f, g);
                set3(b, 1.0f, dv(2.0f, 1.0f + Q), 1.0f);
                break;

            case MI_FLT_BT_BWC_LOPASS: case MI_FLT_BT_BWC_HIPASS:                       // :1090-1170
            case MI_FLT_BT_LRX_LOPASS: case MI_FLT_BT_LRX_HIPASS:                       // :1509-1563
            {
                const bool lrx = p.base == MI_FLT_BT_LRX_LOPASS || p.base == MI_FLT_BT_LRX_HIPASS;
                const bool hi = p.base == MI_FLT_BT_BWC_HIPASS || p.base == MI_FLT_BT_LRX_HIPASS;
                if (!lrx && J == 0 && (slope & 1))
                {
                    set3(b, 1.0f, 1.0f, 0.0f);
                    set3(t, hi ? 0.0f : g, hi ? g : 0.0f, 0.0f);
                    break;
                }
                const float k = dv(1.0f, 1.0f + Q);
                const float theta = lrx ? dv(float((J & ~1u) + 1) * kPi2, float(slope * 2))
                                        : dv(float(2 * (J - (slope & 1)) + 1) * kPi2, float(slope));
                const float tsin = sinf(theta), tcos = sqrtf(1.0f - tsin * tsin);
                const float kf1 = dv(1.0f, tsin * tsin + k * k * tcos * tcos);
                const float lead = (J == 0) ? g : 1.0f;
                if (hi)
                {
                    set3(t, 0.0f, 0.0f, lead);
                    set3(b, kf1, 2.0f * k * tcos * kf1, 1.0f);
                }
                else
                {
                    set3(t, lead, 0.0f, 0.0f);
                    set3(b, 1.0f, 2.0f * k * tcos * kf1, kf1);
                }
                break;
            }
            case MI_FLT_BT_BWC_HISHELF: case MI_FLT_BT_BWC_LOSHELF:                     // :1172-1233
            {
                const float theta = dv(float(2 * J + 1) * kPi2, float(2 * slope));
                const float tsin = sinf(theta), tcos = sqrtf(1.0f - tsin * tsin);
                const float gain = sqrtf(g), fg = expf(dv(logf(gain), 2.0f * float(slope)));
                const float k = dv(1.0f, 1.0f + Q * (1.0f - expf(2.0f - gain - dv(1.0f, gain))));
                const float kf = tsin * tsin + k * k * tcos * tcos;
                set_mirrored(p.base == MI_FLT_BT_BWC_HISHELF, t, b, dv(kf, fg), 2.0f * k * tcos, fg);
                if (J == 0)
                    scale3(t, gain);
                break;
            }
            case MI_FLT_BT_BWC_LADDERPASS: case MI_FLT_BT_BWC_LADDERREJ:                // :1235-1347
            {
                const bool passing = p.base == MI_FLT_BT_BWC_LADDERPASS;
                const float rc = dv(1.0f, float(slope * 2));
                const float theta = (float((J & ~1u) + 1) * kPi2) * rc;
                const float tcos = cosf(theta), tcos2 = tcos * tcos, tsin2 = 1.0f - tcos2;
                if (J & 1)                                                              // second shelf, always a hi-shelf
                {
                    const float xf2 = xf * xf, xtcos = 2.0f * tcos * xf;
                    const float gain = passing ? sqrtf(g) : sqrtf(dv(1.0f, g));
                    const float fg = expf(logf(gain) * rc);
                    const float k = dv(1.0f, 1.0f + Q * (1.0f - expf(2.0f - gain - dv(1.0f, gain))));
                    const float kf = tsin2 + k * k * tcos2;
                    set3(b, dv(kf, fg), k * xtcos, fg * xf2);
                    set3(t, fg, b[1], b[0] * xf2);
                    if (!(J & ~1u))
                        scale3(t, dv(1.0f, gain));
                }
                else
                {
                    const float xtcos = 2.0f * tcos, gain = sqrtf(g);
                    const float k = dv(1.0f, 1.0f + Q * (1.0f - expf(2.0f - gain - dv(1.0f, gain))));
                    const float fg = expf(logf(gain) * rc), kf = tsin2 + k * k * tcos2;
                    set_mirrored(passing, t, b, dv(kf, fg), k * xtcos, fg);
                    if (!(J & ~1u))
                        scale3(t, gain);
                }
                break;
            }
            case MI_FLT_BT_BWC_BELL: case MI_FLT_BT_LRX_BELL:                           // :1349-1442, :1573-1666
            {
                const bool lrx = p.base == MI_FLT_BT_LRX_BELL;
                const float sl = float(slope * (lrx ? 4 : 2));
                const float theta = dv(float(lrx ? ((J & ~3u) + 2) : ((J & ~1u) + 1)) * kPi2, sl);
                const float tsin = sinf(theta), tcos = sqrtf(1.0f - tsin * tsin);
                const float k = dv(1.0f, 1.0f + Q), kf = tsin * tsin + k * k * tcos * tcos;
                const float fg = expf(dv(logf(g), sl)), c2 = 2.0f * k * tcos;
                if (J & 1)
                {
                    if (g >= 1.0f) { set3(t, 1.0f, dv(c2, fg), dv(kf, fg * fg));   set3(b, 1.0f, c2, kf); }
                    else           { set3(t, 1.0f, c2, kf);                        set3(b, 1.0f, c2 * fg, kf * fg * fg); }
                }
                else
                {
                    if (g >= 1.0f) { set3(t, 1.0f, dv(c2 * fg, kf), dv(1.0f * fg * fg, kf));    set3(b, 1.0f, dv(c2, kf), dv(1.0f, kf)); }
                    else           { set3(t, 1.0f, dv(c2, kf), dv(1.0f, kf));      set3(b, 1.0f, dv(c2, fg * kf), dv(1.0f, fg * fg * kf)); }
                }
                break;
            }
            case MI_FLT_BT_BWC_BANDPASS: case MI_FLT_BT_LRX_BANDPASS:                   // :1444-1505, :1668-1730
            {
                const bool lrx = p.base == MI_FLT_BT_LRX_BANDPASS;
                const float sl = float(slope * (lrx ? 4 : 2));
                const float theta = dv(float(lrx ? ((J & ~3u) + 2) : ((J & ~1u) + 1)) * kPi2, sl);
                const float tsin = sinf(theta), tcos = sqrtf(1.0f - tsin * tsin);
                const float k = dv(1.0f, 1.0f + Q), kf1 = dv(1.0f, tsin * tsin + k * k * tcos * tcos);
                if (J & 1)                                                              // hi-pass cascade
                {
                    set3(t, 1.0f, 0.0f, 0.0f);
                    set3(b, 1.0f, 2.0f * k * tcos * xf * kf1, xf * xf * kf1);
                }
                else
                {
                    set3(t, 0.0f, 0.0f, (J == 0) ? g : 1.0f);
                    set3(b, kf1, 2.0f * k * tcos * kf1, 1.0f);
                }
                break;
            }
            case MI_FLT_BT_LRX_HISHELF: case MI_FLT_BT_LRX_LOSHELF:                     // build_lrx_shelf_filter_bank, :509-623
            {
                const float b3 = sqrtf(g), gain = sqrtf(b3), fg = iroot(sqrtf(gain), float(slope));
                const float k = dv(1.0f, 1.0f + Q * (1.0f - expf(2.0f - gain - dv(1.0f, gain))));
                const float theta = dv(float((J & ~1u) + 1) * kPi2, float(2 * slope));
                const float tcos = cosf(theta), tcos2 = tcos * tcos, tsin2 = 1.0f - tcos2;
                const float kf = tsin2 + k * k * tcos2;
                set_mirrored(p.base == MI_FLT_BT_LRX_HISHELF, t, b, kf * dv(1.0f, fg), k * (2.0f * tcos), fg);
                if (J == 0)
                    scale3(t, b3);
                break;
            }
            case MI_FLT_BT_LRX_LADDERPASS: case MI_FLT_BT_LRX_LADDERREJ:                // :320-507
            {
                const bool passing = p.base == MI_FLT_BT_LRX_LADDERPASS;
                const float sl = float(slope * 4);
                const float gain = sqrtf(g), igain = dv(1.0f, gain), fg = iroot(gain, sl), ifg = dv(1.0f, fg);
                const float k = dv(1.0f, 1.0f + Q * (1.0f - expf(2.0f - gain - igain)));
                const float xf2 = xf * xf;
                const float theta = dv(float((J & ~3u) + 2) * kPi2, sl);
                const float tcos = cosf(theta), tcos2 = tcos * tcos, tsin2 = 1.0f - tcos2;
                const float xtcos = 2.0f * tcos, xtcos_xf = 2.0f * tcos * xf;
                const float kf = tsin2 + k * k * tcos2;
                float gn = gain;
                if (passing)
                {
                    if (J & 1)
                    {
                        gn = igain;
                        const float b0 = kf * ifg, b1 = k * xtcos_xf;
                        set3(t, fg, b1, b0 * xf2);
                        set3(b, b0, b1, fg * xf2);
                    }
                    else
                    {
                        const float t0 = kf * ifg, t1 = k * xtcos;
                        set3(t, t0, t1, fg);
                        set3(b, fg, t1, t0);
                    }
                }
                else if (J & 1)
                {
                    const float b0 = kf * fg, b1 = k * xtcos_xf;
                    set3(t, ifg, b1, b0 * xf2);
                    set3(b, b0, b1, ifg * xf2);
                }
                else
                {
                    const float b0 = kf * ifg, b1 = k * xtcos;
                    set3(t, fg, b1, b0);
                    set3(b, b0, b1, fg);
                }
                if (!(J & ~1u))
                    scale3(t, gn);
                break;
            }
            default:
                set3(t, 1.0f, 0.0f, 0.0f); set3(b, 1.0f, 0.0f, 0.0f);
                break;
        }
        return cs_;
    }

    struct section5 { float b0, b1, b2, a1, a2; };

    // dsp::bilinear_transform_x1 of one cascade: the formulas of Filter::bilinear_transform (Filter.cpp:2225-2262)
    __host__ __device__ inline section5 bilinear(const float *t, const float *b, float kf)
    {
        const float kf2 = kf * kf;
        const float T0 = t[0], T1 = t[1] * kf, T2 = t[2] * kf2;
        const float B0 = b[0], B1 = b[1] * kf, B2 = b[2] * kf2;
        const float N = dv(1.0f, B0 + B1 + B2);
        return section5{ (T0 + T1 + T2) * N, 2.0f * (T0 - T2) * N, (T0 - T1 + T2) * N,
                         2.0f * (B2 - B0) * N, (B1 - B2 - B0) * N };                     // denominator signs negated
    }

    // dsp::matched_transform_x1 of one cascade: Filter::matched_transform (Filter.cpp:2291-2416), td = 2 pi / sample rate;
    // the amplitude is matched at w = 0.1 f td (digital) against the prototype at 0.1 (normalised)
    __host__ __device__ inline void matched_poly(const float *p, float f, float td, float *Qo)
    {
        Qo[0] = Qo[1] = Qo[2] = 0.0f;
        if (p[2] == 0.0f)
        {
            if (p[1] == 0.0f)
                Qo[0] = p[0];
            else
            {
                const float k = p[1] / f, R = -p[0] / k;
                Qo[0] = k;
                Qo[1] = -k * expf(R * td);
            }
            return;
        }
        const float k = p[2], qa = 1.0f / (f * f), qb = p[1] / (f * p[2]), qc = p[0] / p[2];
        float D = qb * qb - 4.0f * qa * qc;
        if (D >= 0)
        {
            D = sqrtf(D);
            const float R0 = (-qb - D) / (2.0f * qa), R1 = (-qb + D) / (2.0f * qa);
            Qo[0] = k;
            Qo[1] = -k * (expf(R0 * td) + expf(R1 * td));
            Qo[2] = k * expf((R0 + R1) * td);
        }
        else
        {
            D = sqrtf(-D);
            const float R = -qb / (2.0f * qa), K = D / (2.0f * qa);
            Qo[0] = k;
            Qo[1] = -2.0f * k * expf(R * td) * cosf(K * td);
            Qo[2] = k * expf(2.0f * R * td);
        }
    }

    struct matched_side { float P[3]; double A, I; };      // one polynomial: its digital image, |digital| and |analog| at the match point

    __host__ __device__ inline matched_side matched_one(const float *p, float f, float td, double w)
    {
        matched_side r;
        matched_poly(p, f, td, r.P);
        double re = r.P[0] * cos(2.0 * w) + r.P[1] * cos(w) + r.P[2];
        double im = r.P[0] * sin(2.0 * w) + r.P[1] * sin(w);
        r.A = sqrt(re * re + im * im);
        re = p[0] - p[2] * 0.01;
        im = p[1] * 0.1;
        r.I = sqrt(re * re + im * im);
        return r;
    }

    // numerator and denominator go through the same function one after the other (not a loop over a selected pointer:
    // that keeps t[] and b[] in scratch memory on the device)
    __host__ __device__ inline section5 matched(const float *t, const float *b, float f, float td)
    {
        const double w = 0.1 * double(f) * double(td);
        const matched_side n = matched_one(t, f, td, w), d = matched_one(b, f, td, w);
        const double AN = (d.A * n.I) / (n.A * d.I), N = 1.0 / d.P[0];
        return section5{ float(n.P[0] * N * AN), float(n.P[1] * N * AN), float(n.P[2] * N * AN),
                         float(-d.P[1] * N), float(-d.P[2] * N) };
    }

    struct dyn_filter            // one filter of the bank, kernel view
    {
        dyn_params  p;
        uint32_t    nc;          // cascades
        int         bilinear;    // nType & 1
        float       kf;          // DynamicFilters.cpp:223-228: 0.95 | 1 / tan(pi f / sr) | 2 pi / sr
        float       f0;          // fFreq (matched transform)
    };

    __host__ __device__ __attribute__((always_inline)) inline section5 dyn_section(const dyn_filter &f, uint32_t J, float g)
    {
        const cascade c = dyn_cascade(f.p, J, g);
        return f.bilinear ? bilinear(c.t, c.b, f.kf) : matched(c.t, c.b, f.f0, f.kf);
    }

    // the bilinear section of a filter whose base type is known when the kernel is compiled: the builders' switch folds
    // to the one family, whose gain-independent terms then leave the per-sample code
    // (BASE: the base type, + MATCHED_BIT for its matched-Z twin)
    constexpr uint32_t MATCHED_BIT = 0x100;
    template <uint32_t BASE>
    __device__ __forceinline__ section5 dyn_section_of(const dyn_filter &f, uint32_t J, float g)
    {
        dyn_params p = f.p;
        p.base = BASE & (MATCHED_BIT - 1);
        const cascade c = dyn_cascade(p, J, g);
        if constexpr ((BASE & MATCHED_BIT) != 0)
            return matched(c.t, c.b, f.f0, f.kf);
        else
            return bilinear(c.t, c.b, f.kf);
    }

    // ---- kernel -----------------------------------------------------------------------------------------------
    // One workgroup of NW waves per channel; a lane owns LC consecutive samples, the workgroup a super-block of
    // NW x 64 x LC samples (longer calls walk super-block after super-block).  Per section: every lane builds the
    // coefficients of its samples from the gain curve, folds its chunk into the affine map of the section state
    // s -> M s + v, an inclusive shuffle scan composes the maps inside the wave, the waves' total maps meet in LDS, and
    // the exact per-sample recurrence runs from each chunk's true start state.  What dominates is the coefficient
    // arithmetic (transcendental functions per sample and section): hence as many waves as the block has work for
    // (four per SIMD at 1024 channels x 4096 samples instead of one), and the sections of one sample are built once
    // for the filter types whose cascades do not depend on the cascade index.
    constexpr int LC = 8;                        // samples per lane and super-block
#ifndef MI_DYN_BUILDERS_IN_FLIGHT
#define MI_DYN_BUILDERS_IN_FLIGHT 2
#endif
#ifndef MI_DYN_SAMPLES_PER_LANE
#define MI_DYN_SAMPLES_PER_LANE 8               // per-type kernels: 16 spills at 256 VGPRs (33.6 us against 31.2), 12 leaves a ragged second super-block
#endif
#ifndef MI_DYN_SPEC_FROM
#define MI_DYN_SPEC_FROM (64 * 4 * MI_DYN_SAMPLES_PER_LANE / 2) // calls longer than half a super-block of the per-type kernels
#endif
#ifndef MI_DYN_WAVES_PER_SIMD
#define MI_DYN_WAVES_PER_SIMD 2                 // register budget of the per-type kernels: 256 VGPRs
#endif

    struct aff { float m00, m01, m10, m11, v0, v1; };      // s -> M s + v  (a lane's own chunk)
    // The chunk maps are composed across lanes and waves in double: a product of several hundred 2 x 2 matrices with
    // eigenvalues next to the unit circle collects more float32 round-off than the recurrence it stands for (LRX low-pass:
    // 3e-5 of the peak against the recurrence's own 6e-6, also when only the 64 maps of a wave are composed in float32),
    // and the scan is a small part of the section's arithmetic.
    struct affd { double m00, m01, m10, m11, v0, v1; };

    // the map of `first` followed by `then`
    __device__ __forceinline__ affd then_(const affd &first, const affd &then)
    {
        affd r;
        r.m00 = then.m00 * first.m00 + then.m01 * first.m10;
        r.m01 = then.m00 * first.m01 + then.m01 * first.m11;
        r.m10 = then.m10 * first.m00 + then.m11 * first.m10;
        r.m11 = then.m10 * first.m01 + then.m11 * first.m11;
        r.v0  = then.m00 * first.v0 + then.m01 * first.v1 + then.v0;
        r.v1  = then.m10 * first.v0 + then.m11 * first.v1 + then.v1;
        return r;
    }

    // filter types whose analog cascade is the same for every cascade index J (dyn_cascade above)
    __host__ __device__ inline bool uniform_cascades(uint32_t base)
    {
        return base == MI_FLT_BT_RLC_BELL || base == MI_FLT_BT_RLC_RESONANCE || base == MI_FLT_BT_RLC_NOTCH ||
               base == MI_FLT_BT_AMPLIFIER;
    }

    //
    // BASE = 0: any filter (the type is a kernel argument; the sections of a lane's samples are built in ONE rolled
    // loop and parked in LDS).  BASE != 0: a filter of that base type (+ MATCHED_BIT: its matched-Z twin) -- one kernel per type, the sections of
    // the lane's eight samples are built side by side and stay in registers; without the 48 KiB of parked sections all
    // 1024 workgroups of the bench shape are resident at once (four waves per SIMD) instead of 768 and then 256.
    template <int NW, uint32_t BASE>
    __global__ __launch_bounds__(64 * NW, BASE != 0 ? MI_DYN_WAVES_PER_SIMD : 1)
    void dynfilter_kernel(float *out, const float *in, const float *gain, size_t out_stride, size_t in_stride,
                          size_t gain_stride, uint32_t samples, dyn_filter f, float *state /* [channels][CHAINS_MAX][2] */,
                          int aligned)
    {
        constexpr int LCK = (BASE != 0) ? MI_DYN_SAMPLES_PER_LANE : LC;     // the lane's chunk (LC_OF<BASE> on the host side)
        constexpr uint32_t SUPER = uint32_t(NW) * 64u * uint32_t(LCK);
        const uint32_t ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const float *x_in = in + size_t(ch) * in_stride;
        const float *g_in = gain + size_t(ch) * gain_stride;
        float *y_out = out + size_t(ch) * out_stride;
        float2 *gmem = reinterpret_cast<float2 *>(state + size_t(ch) * CHAINS_MAX * 2);
        __shared__ float2 mem[CHAINS_MAX];                  // the cascades' carried state, in LDS for the launch
        __shared__ affd wmap[NW];                           // each wave's map of the section in hand
        // The sections of the lane's samples, [sample][coefficient][thread]: they are built in ONE rolled loop (a single
        // inlined copy of the builders' switch, whose gain-independent parts -- the sines and cosines of angles that depend
        // on the cascade index and the slope only -- the compiler hoists out of the loop) and read back where needed.
        constexpr bool SPEC = BASE != 0;
        __shared__ float qs[SPEC ? 1 : LCK][5][SPEC ? 1 : 64 * NW];
        __shared__ float gs[SPEC ? 1 : LCK][SPEC ? 1 : 64 * NW];
        section5 q[SPEC ? LCK : 1];
        for (uint32_t J = tid; J < f.nc; J += 64 * NW)
            mem[J] = gmem[J];
        __syncthreads();
        const bool uniform = uniform_cascades(SPEC ? (BASE & (MATCHED_BIT - 1)) : f.p.base);

        for (uint32_t pos = 0; pos < samples; pos += SUPER)
        {
            const uint32_t left = samples - pos;
            const uint32_t valid = (left >= SUPER) ? SUPER : left;                      // samples of this super-block
            const uint32_t c0 = tid * LCK;                                               // the lane's chunk in it
            float x[LCK], g[LCK];
            if (aligned && c0 + LCK <= valid)
            {
                #pragma unroll
                for (int k = 0; k < LCK; k += 4)
                {
                    const float4 xv = *reinterpret_cast<const float4 *>(x_in + pos + c0 + k);
                    const float4 gv = *reinterpret_cast<const float4 *>(g_in + pos + c0 + k);
                    x[k] = xv.x; x[k + 1] = xv.y; x[k + 2] = xv.z; x[k + 3] = xv.w;
                    g[k] = gv.x; g[k + 1] = gv.y; g[k + 2] = gv.z; g[k + 3] = gv.w;
                }
            }
            else
            {
                #pragma unroll
                for (int k = 0; k < LCK; ++k)
                {
                    const bool ok = c0 + k < valid;
                    x[k] = ok ? x_in[pos + c0 + k] : 0.0f;
                    g[k] = ok ? g_in[pos + c0 + k] : 1.0f;
                }
            }
            const int nk = (c0 >= valid) ? 0 : ((valid - c0 >= uint32_t(LCK)) ? LCK : int(valid - c0));   // lane's samples
            const uint32_t last_tid = (valid - 1) / LCK;                                 // holds the super-block's last sample

            if constexpr (!SPEC)
            {
                #pragma unroll
                for (int k = 0; k < LCK; ++k)
                    gs[k][tid] = g[k];
            }
            // the sections one after the other; FULL: every lane holds LCK samples of the super-block (no per-sample guards)
            auto run_sections = [&](auto full_tag) __attribute__((always_inline))
            {
                constexpr bool FULL = decltype(full_tag)::value;
                for (uint32_t J = 0; J < f.nc; ++J)
                {
                    if (J == 0 || !uniform)
                    {
                        if constexpr (SPEC)
                        {
                            #pragma unroll
                            for (int k = 0; k < LCK; ++k)
                            {
                                q[k] = dyn_section_of<BASE>(f, J, g[k]);
                                if ((k + 1) % (((BASE & MATCHED_BIT) != 0) ? 1 : MI_DYN_BUILDERS_IN_FLIGHT) == 0)       // so many builders interleaved (matched-Z: one, its double-precision part is register-hungry)
                                    __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                        else
                        {
                            #pragma unroll 1
                            for (int k = 0; k < LCK; ++k)
                            {
                                const section5 c = dyn_section(f, J, gs[k][tid]);
                                qs[k][0][tid] = c.b0; qs[k][1][tid] = c.b1; qs[k][2][tid] = c.b2; qs[k][3][tid] = c.a1; qs[k][4][tid] = c.a2;
                            }
                        }
                    }
                    auto q_of = [&](int k) -> section5 {
                        if constexpr (SPEC)
                            return q[k];
                        else
                            return section5{ qs[k][0][tid], qs[k][1][tid], qs[k][2][tid], qs[k][3][tid], qs[k][4][tid] };
                    };
                    // the chunk's state map: d0' = a1 d0 + d1 + (b1 + a1 b0) x,  d1' = a2 d0 + (b2 + a2 b0) x
                    aff m = { 1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f };
                    #pragma unroll
                    for (int k = 0; k < LCK; ++k)
                    {
                        if (FULL || k < nk)
                        {
                            const section5 c = q_of(k);
                            const float a1 = c.a1, a2 = c.a2;
                            const float u0 = (c.b1 + a1 * c.b0) * x[k], u1 = (c.b2 + a2 * c.b0) * x[k];
                            aff r;
                            r.m00 = a1 * m.m00 + m.m10;  r.m01 = a1 * m.m01 + m.m11;
                            r.m10 = a2 * m.m00;          r.m11 = a2 * m.m01;
                            r.v0  = a1 * m.v0 + m.v1 + u0;
                            r.v1  = a2 * m.v0 + u1;
                            m = r;
                        }
                    }
                    // inclusive scan over the lanes: afterwards md maps the wave's start state to the state after this chunk
                    affd md = { double(m.m00), double(m.m01), double(m.m10), double(m.m11), double(m.v0), double(m.v1) };
                    #pragma unroll
                    for (int d = 1; d < 64; d <<= 1)
                    {
                        affd o;
                        o.m00 = __shfl_up(md.m00, d); o.m01 = __shfl_up(md.m01, d); o.m10 = __shfl_up(md.m10, d);
                        o.m11 = __shfl_up(md.m11, d); o.v0 = __shfl_up(md.v0, d);   o.v1 = __shfl_up(md.v1, d);
                        if (int(lane) >= d)
                            md = then_(o, md);
                    }
                    // the state entering this wave: the carried state through the maps of the waves before it
                    const float2 cs = mem[J];
                    double w0 = cs.x, w1 = cs.y;
                    if (NW > 1)
                    {
                        if (lane == 63)
                            wmap[wave] = md;
                        __syncthreads();
                        for (uint32_t v = 0; v < wave; ++v)
                        {
                            const affd p = wmap[v];
                            const double n0 = p.m00 * w0 + p.m01 * w1 + p.v0, n1 = p.m10 * w0 + p.m11 * w1 + p.v1;
                            w0 = n0;
                            w1 = n1;
                        }
                    }
                    // start state of the lane's chunk: the map of everything before it in the wave, applied to that state
                    affd e;
                    e.m00 = __shfl_up(md.m00, 1); e.m01 = __shfl_up(md.m01, 1); e.m10 = __shfl_up(md.m10, 1);
                    e.m11 = __shfl_up(md.m11, 1); e.v0 = __shfl_up(md.v0, 1);   e.v1 = __shfl_up(md.v1, 1);
                    float d0 = float((lane == 0) ? w0 : e.m00 * w0 + e.m01 * w1 + e.v0);
                    float d1 = float((lane == 0) ? w1 : e.m10 * w0 + e.m11 * w1 + e.v1);
                    // the exact recurrence with the sample's own coefficients (dsp::dyn_biquad_process_x1)
                    #pragma unroll
                    for (int k = 0; k < LCK; ++k)
                    {
                        if (FULL || k < nk)
                        {
                            const section5 c = q_of(k);
                            const float xx = x[k];                  // same operation order as biquad.hip's sections
                            const float tq = fmaf(c.b1, xx, d1);
                            const float u  = c.b2 * xx;
                            const float y  = fmaf(c.b0, xx, d0);
                            d0 = fmaf(c.a1, y, tq);
                            d1 = fmaf(c.a2, y, u);
                            x[k] = y;
                        }
                    }
                    __syncthreads();                                // every wave has read mem[J] and wmap[]
                    if (tid == last_tid)                            // carried to the next super-block / call
                        mem[J] = make_float2(d0, d1);
                }
            };
            if constexpr (SPEC)                                 // (one copy of the any-type kernel's builders is enough)
            {
                if (valid == SUPER)
                    run_sections(std::true_type{});
                else
                    run_sections(std::false_type{});
            }
            else
                run_sections(std::false_type{});
            if (aligned && c0 + LCK <= valid)
            {
                #pragma unroll
                for (int k = 0; k < LCK; k += 4)
                    *reinterpret_cast<float4 *>(y_out + pos + c0 + k) = make_float4(x[k], x[k + 1], x[k + 2], x[k + 3]);
            }
            else
            {
                #pragma unroll
                for (int k = 0; k < LCK; ++k)
                    if (c0 + k < valid)
                        y_out[pos + c0 + k] = x[k];
            }
            __syncthreads();                                    // mem[] of the last section before the next super-block reads it
        }
        for (uint32_t J = tid; J < f.nc; J += 64 * NW)
            gmem[J] = mem[J];
    }

    __global__ __launch_bounds__(256)
    void dyn_copy_kernel(float *out, const float *in, size_t out_stride, size_t in_stride, size_t count)
    {
        const uint32_t ch = blockIdx.y;
        for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < count; i += size_t(gridDim.x) * 256)
            out[size_t(ch) * out_stride + i] = in[size_t(ch) * in_stride + i];
    }
} // namespace

struct mi_dynfilter_bank
{
    uint32_t channels = 0, filters = 0, sample_rate = 0;
    struct filter_t
    {
        mi_filter_params_t  params;      // as set_params leaves them: fFreq2 transformed (DynamicFilters.cpp:170-178)
        bool                active = false;
    };
    std::vector<filter_t>   filter;
    bool                    clear_mem = false;
    float                  *d_state = nullptr;      // [filters][channels][CHAINS_MAX][2]
};

namespace
{
    // DynamicFilters::set_params' treatment of the parameters (DynamicFilters.cpp:135-178): band filters get f2 >= f, then
    // fFreq2 becomes the ratio the builders use (pre-warped for the bilinear types)
    void transform_params(mi_filter_params_t *fp, uint32_t sample_rate)
    {
        switch (base_type(fp->nType))
        {
            case MI_FLT_BT_RLC_LADDERPASS: case MI_FLT_BT_RLC_LADDERREJ: case MI_FLT_BT_RLC_BANDPASS:
            case MI_FLT_BT_BWC_LADDERPASS: case MI_FLT_BT_BWC_LADDERREJ: case MI_FLT_BT_BWC_BANDPASS:
            case MI_FLT_BT_LRX_LADDERPASS: case MI_FLT_BT_LRX_LADDERREJ: case MI_FLT_BT_LRX_BANDPASS:
                if (fp->nType != MI_FLT_NONE && fp->fFreq2 < fp->fFreq)
                {
                    const float f = fp->fFreq;
                    fp->fFreq = fp->fFreq2;
                    fp->fFreq2 = f;
                }
                break;
            default:
                break;
        }
        if (fp->nType & 1)
        {
            const float nf = kPi / float(sample_rate);
            fp->fFreq2 = tanf(fp->fFreq * nf) / tanf(fp->fFreq2 * nf);
        }
        else
            fp->fFreq2 = fp->fFreq / fp->fFreq2;
    }

    bool make_filter_from(const mi_filter_params_t &p, uint32_t sample_rate, dyn_filter *f)
    {
        f->p.base = base_type(p.nType);
        f->p.slope = p.nSlope;
        f->p.xf = p.fFreq2;
        f->p.Q = p.fQuality;
        f->nc = cascade_count(p.nType, p.nSlope);
        f->bilinear = int(p.nType & 1);
        f->f0 = p.fFreq;
        // DynamicFilters.cpp:223-228
        f->kf = (p.nType <= MI_FLT_MT_AMPLIFIER) ? 0.95f
              : (p.nType & 1) ? 1.0f / tanf(p.fFreq * kPi / float(sample_rate))
              : kTwoPi / float(sample_rate);
        return f->nc > 0;
    }

    bool make_filter(const mi_dynfilter_bank *b, uint32_t id, dyn_filter *f)
    {
        return make_filter_from(b->filter[id].params, b->sample_rate, f);
    }

    bool bypassed(const mi_dynfilter_bank *b, uint32_t id)     // DynamicFilters.cpp:207-212
    {
        if (id >= b->filters)
            return true;
        const mi_dynfilter_bank::filter_t &f = b->filter[id];
        return !f.active || f.params.nType == MI_FLT_NONE || f.params.nSlope == 0 || b->sample_rate == 0;
    }
} // namespace

extern "C" {

int mi_dynfilter_bank_create(mi_dynfilter_bank_t **bank, uint32_t channels, uint32_t filters)
{
    MI_REQUIRE(bank != nullptr, MI_EINVAL, "mi_dynfilter_bank_create: NULL result pointer");
    *bank = nullptr;
    MI_REQUIRE(channels > 0 && filters > 0, MI_EINVAL, "mi_dynfilter_bank_create: channels and filters must be > 0");
    MI_REQUIRE(mi_dspu_device_count() > 0, MI_ENODEV, "no HIP device available (there is no CPU fallback)");
    mi_dynfilter_bank *b = new (std::nothrow) mi_dynfilter_bank();
    MI_REQUIRE(b != nullptr, MI_ENOMEM, "mi_dynfilter_bank_create: out of host memory");
    b->channels = channels;
    b->filters = filters;
    b->filter.resize(filters);
    for (mi_dynfilter_bank::filter_t &f : b->filter)           // DynamicFilters.cpp:95-108
    {
        f.params.nType = MI_FLT_NONE; f.params.nSlope = 0;
        f.params.fFreq = f.params.fFreq2 = f.params.fGain = f.params.fQuality = 0.0f;
        f.active = false;
    }
    const size_t bytes = size_t(filters) * channels * CHAINS_MAX * 2 * sizeof(float);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&b->d_state), bytes);
    if (e == hipSuccess) e = hipMemset(b->d_state, 0, bytes);
    if (e != hipSuccess)
    {
        mi_dynfilter_bank_destroy(b);
        return mi::fail(e == hipErrorOutOfMemory ? MI_ENOMEM : MI_EHIP, "mi_dynfilter_bank_create: %s", hipGetErrorString(e));
    }
    *bank = b;
    return MI_OK;
}

int mi_dynfilter_bank_destroy(mi_dynfilter_bank_t *b)
{
    if (b == nullptr)
        return MI_OK;
    (void)hipFree(b->d_state);
    delete b;
    return MI_OK;
}

int mi_dynfilter_bank_set_sample_rate(mi_dynfilter_bank_t *b, uint32_t sample_rate)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_dynfilter_bank_set_sample_rate: NULL bank");
    b->sample_rate = sample_rate;
    return MI_OK;
}

int mi_dynfilter_bank_set_params(mi_dynfilter_bank_t *b, uint32_t id, const mi_filter_params_t *params)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_dynfilter_bank_set_params: NULL bank");
    MI_REQUIRE(params != nullptr, MI_EINVAL, "mi_dynfilter_bank_set_params: NULL params");
    MI_REQUIRE(id < b->filters, MI_EINVAL, "mi_dynfilter_bank_set_params: filter %u out of range", id);     // the class returns false
    MI_REQUIRE(params->nType == MI_FLT_NONE || cascade_count(params->nType, params->nSlope ? params->nSlope : 1) > 0, MI_EINVAL,
               "mi_dynfilter_bank_set_params: filter type %u has no dynamic form here (FLT_*_RLC_ENVELOPE and the "
               "static-only types of DynamicFilters.cpp:625-1738's default branch)", params->nType);
    mi_filter_params_t *fp = &b->filter[id].params;
    if (fp->nType != params->nType)                             // DynamicFilters.cpp:132-133
        b->clear_mem = true;
    *fp = *params;
    transform_params(fp, b->sample_rate);
    return MI_OK;
}

int mi_dynfilter_bank_get_params(const mi_dynfilter_bank_t *b, uint32_t id, mi_filter_params_t *params, int *active)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_dynfilter_bank_get_params: NULL bank");
    MI_REQUIRE(id < b->filters, MI_EINVAL, "mi_dynfilter_bank_get_params: filter %u out of range", id);
    if (params) *params = b->filter[id].params;
    if (active) *active = b->filter[id].active ? 1 : 0;
    return MI_OK;
}

int mi_dynfilter_bank_set_filter_active(mi_dynfilter_bank_t *b, uint32_t id, int active)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_dynfilter_bank_set_filter_active: NULL bank");
    MI_REQUIRE(id < b->filters, MI_EINVAL, "mi_dynfilter_bank_set_filter_active: filter %u out of range", id);
    (void)active;
    b->filter[id].active = true;                                // the reference sets true whatever is asked (DynamicFilters.h:147-153)
    return MI_OK;
}

int mi_dynfilter_bank_process(mi_dynfilter_bank_t *b, uint32_t id, float *out, const float *in, const float *gain,
                              size_t samples, size_t out_stride, size_t in_stride, size_t gain_stride, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_dynfilter_bank_process: NULL bank");
    if (samples == 0)
        return MI_OK;
    MI_REQUIRE(out != nullptr && in != nullptr, MI_EINVAL, "mi_dynfilter_bank_process: NULL buffer");
    MI_REQUIRE(out_stride >= samples && in_stride >= samples, MI_EINVAL, "mi_dynfilter_bank_process: stride shorter than the block");
    MI_REQUIRE(samples < (size_t(1) << 31), MI_EINVAL, "mi_dynfilter_bank_process: call too long");
    hipStream_t st = mi::as_stream(stream);
    dyn_filter f;
    if (bypassed(b, id) || !make_filter(b, id, &f))             // DynamicFilters.cpp:207-212: copy
    {
        if (out != in)
        {
            const unsigned gx = unsigned(std::min<size_t>((samples + 255) / 256, 64));
            hipLaunchKernelGGL(dyn_copy_kernel, dim3(gx, b->channels), dim3(256), 0, st, out, in, out_stride, in_stride, samples);
            MI_HIP_CHECK(hipGetLastError());
        }
        return MI_OK;
    }
    MI_REQUIRE(gain != nullptr && gain_stride >= samples, MI_EINVAL, "mi_dynfilter_bank_process: bad gain buffer");
    if (b->clear_mem)                                           // :214-219: every filter's memory
    {
        // a captured memset would zero the filter memory again on EVERY replay of the graph
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (st != nullptr && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return mi::fail(MI_ESTATE, "mi_dynfilter_bank_process: a clear of the filter memory is pending (init / set_sample_rate): "
                                       "make one eager call before capturing");
        MI_HIP_CHECK(hipMemsetAsync(b->d_state, 0, size_t(b->filters) * b->channels * CHAINS_MAX * 2 * sizeof(float), st));
        b->clear_mem = false;
    }
    float *state = b->d_state + size_t(id) * b->channels * CHAINS_MAX * 2;
    const int aligned = ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(gain)) % 16 == 0 &&
                         out_stride % 4 == 0 && in_stride % 4 == 0 && gain_stride % 4 == 0) ? 1 : 0;
    // as many waves per channel as the call has chunks for (a wave covers 64 x LC = 512 samples), up to four (48 KiB of
    // LDS per workgroup: three workgroups per CU)
    const size_t chunks = (samples + 64 * LC - 1) / (64 * LC);
    #define MI_DYN_LAUNCH(NW, BASE) hipLaunchKernelGGL((dynfilter_kernel<NW, BASE>), dim3(b->channels), dim3(64 * NW), 0, st, out, in, \
                                                       gain, out_stride, in_stride, gain_stride, uint32_t(samples), f, state, aligned)
    bool issued = false;
    // long calls: the kernel of the filter's type.  (Matched-Z bell and resonance are the two of the 54 that the any-type
    // kernel serves faster: 92 against 100 us per 1024 x 4096 call, profiles/r03_experiments/dynfilter_per_type.txt)
    const bool slower = !f.bilinear && (f.p.base == MI_FLT_BT_RLC_BELL || f.p.base == MI_FLT_BT_RLC_RESONANCE);
    if (samples > size_t(MI_DYN_SPEC_FROM) && !slower)
    {
        issued = true;
        switch (f.p.base)
        {
            #define MI_DYN_CASE(B) case B: if (f.bilinear) MI_DYN_LAUNCH(4, B); else MI_DYN_LAUNCH(4, (B | MATCHED_BIT)); break;
            MI_DYN_CASE(MI_FLT_BT_AMPLIFIER)
            MI_DYN_CASE(MI_FLT_BT_RLC_LOPASS)     MI_DYN_CASE(MI_FLT_BT_RLC_HIPASS)
            MI_DYN_CASE(MI_FLT_BT_RLC_LOSHELF)    MI_DYN_CASE(MI_FLT_BT_RLC_HISHELF)
            MI_DYN_CASE(MI_FLT_BT_RLC_LADDERPASS) MI_DYN_CASE(MI_FLT_BT_RLC_LADDERREJ)
            MI_DYN_CASE(MI_FLT_BT_RLC_BANDPASS)
            MI_DYN_CASE(MI_FLT_BT_RLC_BELL)       MI_DYN_CASE(MI_FLT_BT_RLC_RESONANCE)
            MI_DYN_CASE(MI_FLT_BT_RLC_NOTCH)
            MI_DYN_CASE(MI_FLT_BT_BWC_LOPASS)     MI_DYN_CASE(MI_FLT_BT_BWC_HIPASS)
            MI_DYN_CASE(MI_FLT_BT_LRX_LOPASS)     MI_DYN_CASE(MI_FLT_BT_LRX_HIPASS)
            MI_DYN_CASE(MI_FLT_BT_BWC_LOSHELF)    MI_DYN_CASE(MI_FLT_BT_BWC_HISHELF)
            MI_DYN_CASE(MI_FLT_BT_BWC_LADDERPASS) MI_DYN_CASE(MI_FLT_BT_BWC_LADDERREJ)
            MI_DYN_CASE(MI_FLT_BT_BWC_BELL)       MI_DYN_CASE(MI_FLT_BT_LRX_BELL)
            MI_DYN_CASE(MI_FLT_BT_BWC_BANDPASS)   MI_DYN_CASE(MI_FLT_BT_LRX_BANDPASS)
            MI_DYN_CASE(MI_FLT_BT_LRX_LOSHELF)    MI_DYN_CASE(MI_FLT_BT_LRX_HISHELF)
            MI_DYN_CASE(MI_FLT_BT_LRX_LADDERPASS) MI_DYN_CASE(MI_FLT_BT_LRX_LADDERREJ)
            #undef MI_DYN_CASE
            default: issued = false; break;
        }
    }
    if (issued)                ;
    else if (chunks <= 1)      MI_DYN_LAUNCH(1, 0);
    else if (chunks <= 2)      MI_DYN_LAUNCH(2, 0);
    else                       MI_DYN_LAUNCH(4, 0);
    #undef MI_DYN_LAUNCH
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

int mi_dynfilter_sections(const mi_filter_params_t *params, uint32_t sample_rate, float gain, mi_biquad_x1_t *sections,
                          uint32_t max_sections, uint32_t *count)
{
    MI_REQUIRE(params != nullptr && count != nullptr && sample_rate > 0, MI_EINVAL, "mi_dynfilter_sections: bad argument");
    *count = 0;
    mi_filter_params_t fp = *params;
    transform_params(&fp, sample_rate);
    dyn_filter f;
    if (fp.nType == MI_FLT_NONE || fp.nSlope == 0 || !make_filter_from(fp, sample_rate, &f))
        return MI_OK;
    *count = f.nc;
    for (uint32_t J = 0; J < f.nc && J < max_sections && sections != nullptr; ++J)
    {
        const section5 q = dyn_section(f, J, gain);
        sections[J] = mi_biquad_x1_t{ q.b0, q.b1, q.b2, q.a1, q.a2, 0.0f, 0.0f, 0.0f };
    }
    return MI_OK;
}

int mi_dynfilter_freq_chart(const mi_filter_params_t *params, uint32_t sample_rate, float *c, const float *f, float gain, size_t count)
{
    MI_REQUIRE(params != nullptr && sample_rate > 0, MI_EINVAL, "mi_dynfilter_freq_chart: bad argument");
    MI_REQUIRE(c != nullptr && (f != nullptr || count == 0), MI_EINVAL, "mi_dynfilter_freq_chart: NULL buffer");
    mi_filter_params_t p = *params;
    transform_params(&p, sample_rate);
    if (p.nType == MI_FLT_NONE || p.nType == MI_FLT_BT_AMPLIFIER || p.nType == MI_FLT_MT_AMPLIFIER)    // DynamicFilters.cpp:1882-1895
    {
        for (size_t i = 0; i < count; ++i)
        {
            c[2 * i] = (p.nType == MI_FLT_NONE) ? 1.0f : gain;
            c[2 * i + 1] = 0.0f;
        }
        return MI_OK;
    }
    dyn_filter df;
    make_filter_from(p, sample_rate, &df);
    const float nf = kPi / float(sample_rate), kf = 1.0f / tanf(p.fFreq * nf), lf = float(sample_rate) * 0.499f;
    for (size_t i = 0; i < count; ++i)
    {
        // the normalised frequency of the prototype (:1905-1912, :1938)
        const float w = (p.nType & 1) ? tanf((f[i] > lf ? lf : f[i]) * nf) * kf : f[i] * (1.0f / p.fFreq);
        const float w2 = w * w;
        float re = 1.0f, im = 0.0f;
        for (uint32_t J = 0; J < df.nc; ++J)                    // dsp::filter_transfer_calc_pc / apply_pc
        {
            const cascade cs = dyn_cascade(df.p, J, gain);
            const float *t = cs.t, *bb = cs.b;
            const float t_re = t[0] - t[2] * w2, t_im = t[1] * w, b_re = bb[0] - bb[2] * w2, b_im = bb[1] * w;
            const float n = 1.0f / (b_re * b_re + b_im * b_im);
            const float hr = (t_re * b_re + t_im * b_im) * n, hi = (t_im * b_re - t_re * b_im) * n;
            const float r2 = re * hr - im * hi, i2 = re * hi + im * hr;
            re = r2;
            im = i2;
        }
        c[2 * i] = re;
        c[2 * i + 1] = im;
    }
    return MI_OK;
}

} // extern "C"
