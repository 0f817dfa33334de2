// Host-side filter designer (see filter_design.h).  Every formula keeps the evaluation type the reference
// uses (float vs double, reference file:line quoted per block) so that the coefficients agree with the
// reference designer to the last bit on the same libm.
#include "filter_design.h"

#include <algorithm>
#include <cmath>

namespace mi
{
namespace
{
    constexpr float kPi      = float(M_PI);
    constexpr float kTwoPi   = float(M_PI * 2.0);
    constexpr float kHalfPi  = float(M_PI_2);
    constexpr float kMinApoQ = 0.1f;                // Filter.cpp:28

    enum : uint32_t
    {
        // base (bilinear) prototypes; the matched-Z twin of each is base + 1 (filters/common.h:38-135)
        T_NONE = MI_FLT_NONE, T_AMP = MI_FLT_BT_AMPLIFIER
    };

    struct builder
    {
        design              &d;
        const uint32_t       slope_req;     // sParams.nSlope (the LRX all-pass reads it directly, Filter.cpp:1436)

        explicit builder(design &dd) : d(dd), slope_req(dd.params.nSlope) {}

        // Filter::add_cascade (Filter.cpp:177-197): zeroed; past the limit the last slot is recycled
        cascade &next()
        {
            if (d.cascades.size() >= CHAINS_MAX)
                d.cascades.pop_back();
            d.cascades.push_back(cascade{ {0, 0, 0, 0}, {0, 0, 0, 0} });
            return d.cascades.back();
        }

        cascade &put(float t0, float t1, float t2, float b0, float b1, float b2)
        {
            cascade &c = next();
            c.t[0] = t0; c.t[1] = t1; c.t[2] = t2;
            c.b[0] = b0; c.b[1] = b1; c.b[2] = b2;
            return c;
        }

        static void scale_top(cascade &c, float g)
        {
            c.t[0] *= g; c.t[1] *= g; c.t[2] *= g;
        }

        // one digital section straight into the bank, mirrored as a plot cascade (Filter.cpp:1624-1646)
        mi_biquad_x1_t &section(float b0, float b1, float b2, float a1, float a2)
        {
            d.sections.push_back(mi_biquad_x1_t{ b0, b1, b2, a1, a2, 0.0f, 0.0f, 0.0f });
            return d.sections.back();
        }

        void mirror(const mi_biquad_x1_t &f)
        {
            put(f.b0, f.b1, f.b2, 1.0f, -f.a1, -f.a2);
        }
    };

    struct pole { float tsin, tcos, kf; };

    // shared by every Butterworth-Chebyshev pair: pole angle -> (sin, cos, sin^2 + k^2 cos^2)
    inline pole bwc_pole(float theta, float k, bool float_sqrt_arg)
    {
        pole p;
        p.tsin = sinf(theta);
        // Filter.cpp:1113 uses sqrtf(1.0 - s*s) (double subtraction), :1237/:1444 use 1.0f - s*s
        p.tcos = float_sqrt_arg ? sqrtf(1.0f - p.tsin * p.tsin) : sqrtf(1.0 - p.tsin * p.tsin);
        p.kf   = p.tsin * p.tsin + k * k * p.tcos * p.tcos;
        return p;
    }

    // ---- RLC prototypes (Filter.cpp:722-1082) ---------------------------------------------------------------
    void rlc(builder &b, uint32_t type, const mi_filter_params_t &fp)
    {
        const uint32_t n = fp.nSlope;
        const float q = fp.fQuality, g = fp.fGain;
        switch (type)
        {
            case MI_FLT_BT_AMPLIFIER:
                b.put(g, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f);
                break;

            case MI_FLT_BT_RLC_LOPASS:
            case MI_FLT_BT_RLC_HIPASS:
            {
                const bool lo = (type == MI_FLT_BT_RLC_LOPASS);
                const float k = 2.0 / (1.0 + q);
                const uint32_t odd = n & 1;
                if (odd)
                {
                    cascade &c = b.put(0, 0, 0, 1.0, 1.0, 0);
                    c.t[lo ? 0 : 1] = g;
                }
                for (uint32_t j = odd; j < n; j += 2)
                {
                    cascade &c = b.put(0, 0, 0, 1.0, k, 1.0);
                    c.t[lo ? 0 : 2] = (j == 0) ? g : 1.0;
                }
                break;
            }

            case MI_FLT_BT_RLC_LOSHELF:
            case MI_FLT_BT_RLC_HISHELF:
            {
                const size_t slope = n * 2;
                const float gain = sqrtf(g);
                const float fg = expf(logf(gain) / slope);
                for (uint32_t j = 0; j < n; ++j)
                {
                    const float m0 = fg, m1 = 2.0 / (1.0 + q), m2 = 1.0 / fg;
                    cascade &c = (type == MI_FLT_BT_RLC_LOSHELF) ? b.put(m0, m1, m2, m2, m1, m0)
                                                                 : b.put(m2, m1, m0, m0, m1, m2);
                    if (j == 0)
                        builder::scale_top(c, gain);
                }
                break;
            }

            case MI_FLT_BT_RLC_LADDERPASS:
            case MI_FLT_BT_RLC_LADDERREJ:
            {
                const bool rej = (type == MI_FLT_BT_RLC_LADDERREJ);
                const size_t slope = n * 2;
                const float gain1 = rej ? sqrtf(1.0 / g) : sqrtf(g);
                const float gain2 = rej ? sqrtf(g) : sqrtf(1.0 / g);
                const float fg1 = expf(logf(gain1) / slope);
                const float fg2 = expf(logf(gain2) / slope);
                const float kf = fp.fFreq2;
                for (uint32_t j = 0; j < n; ++j)
                {
                    // first shelf: low shelf when rejecting, high shelf when passing
                    const float fg = rej ? fg2 : fg1, gain = rej ? gain2 : gain1;
                    const float m0 = fg, m1 = 2.0 / (1.0 + q), m2 = 1.0 / fg;
                    cascade &c1 = rej ? b.put(m0, m1, m2, m2, m1, m0) : b.put(m2, m1, m0, m0, m1, m2);
                    if (j == 0)
                        builder::scale_top(c1, gain);
                    // second shelf: always a high shelf at the second frequency
                    const float h0 = fg2, h1 = 2.0 * kf / (1.0 + q), h2 = kf * kf / fg2;
                    const float l0 = 1.0 / fg2, l2 = fg2 * kf * kf;
                    cascade &c2 = b.put(l0, h1, l2, h0, h1, h2);
                    if (j == 0)
                        builder::scale_top(c2, gain2);
                }
                break;
            }

            case MI_FLT_BT_RLC_BANDPASS:
            {
                const float kf = fp.fFreq2, kf2 = kf * kf;
                const float k = 2.0f / (1.0f + q);
                const uint32_t odd = n & 1;
                if (odd)
                    b.put(0, g * g, 0, 1.0f, 1.0f + kf, kf);
                for (uint32_t j = odd; j < n; j += 2)
                {
                    b.put((j == 0) ? g : 1.0f, 0, 0, 1.0f, k, 1.0f);
                    b.put(0, 0, (j == 0) ? g : 1.0f, 1.0f, k * kf, kf2);
                }
                break;
            }

            case MI_FLT_BT_RLC_BELL:
            {
                const float fg = expf(logf(g) / n);
                const float angle = atanf(fg);
                const float k = 2.0 * (1.0 / fg + fg) / (1.0 + (2.0 * q) / n);
                const float kt = k * sinf(angle), kb = k * cosf(angle);
                for (uint32_t j = 0; j < n; ++j)
                    b.put(1.0, kt, 1.0, 1.0, kb, 1.0);
                break;
            }

            case MI_FLT_BT_RLC_RESONANCE:
            {
                const float angle = atanf(expf(logf(g) / n));
                const float k = 2.0 / (1.0 + q);
                const float kt = k * sinf(angle), kb = k * cosf(angle);
                for (uint32_t j = 0; j < n; ++j)
                    b.put(1.0, kt, 1.0, 1.0, kb, 1.0);
                break;
            }

            case MI_FLT_BT_RLC_NOTCH:
                b.put(g, 0, g, 1.0, 2.0 / (1.0 + q), 1.0);
                break;

            case MI_FLT_BT_RLC_ALLPASS:
            {
                const float k = 2.0f / (1.0f + q);
                cascade *last = nullptr;
                for (uint32_t j = 0; j < n; ++j)
                    last = &b.put(1.0f, -k, 1.0f, 1.0f, k, 1.0f);
                if (last != nullptr)
                    builder::scale_top(*last, g);
                break;
            }

            case MI_FLT_BT_RLC_ALLPASS2:
            {
                const float kf = fp.fFreq2;
                const float kfp1 = 1.0 + kf;
                cascade *last = nullptr;
                for (uint32_t j = 0; j < n; ++j)
                    last = &b.put(1.0, -kfp1, kf, 1.0, kfp1, kf);
                if (last != nullptr)
                    builder::scale_top(*last, g);
                break;
            }

            case MI_FLT_BT_RLC_ENVELOPE:
            {
                uint32_t slope = n, emitted = 0;
                if (slope & 1)
                {
                    float k = 1.0f;
                    for (int i = 0; i < 3; ++i)
                    {
                        cascade &c = b.put(1.0f, (1.0f + 0.25f) * k, 0.25f * k * k,
                                           1.0f, (0.5f + 0.125f) * k, 0.5f * 0.125f * k * k);
                        k *= 0.0625f;
                        if (!(emitted++))
                            builder::scale_top(c, g);
                    }
                }
                slope >>= 1;
                for (uint32_t j = 0; j < slope; ++j)
                {
                    const float lead = (emitted == 0) ? g : 1.0f;
                    b.put(lead, lead, 0, 1.0f, 0.0005f, 0);
                    ++emitted;
                }
                break;
            }

            default:
                break;
        }
    }

    // ---- Butterworth-Chebyshev prototypes (Filter.cpp:1084-1395) ------------------------------------------
    void bwc(builder &b, uint32_t type, const mi_filter_params_t &fp)
    {
        const uint32_t n = fp.nSlope;
        const float q = fp.fQuality, g = fp.fGain;
        switch (type)
        {
            case MI_FLT_BT_BWC_LOPASS:
            case MI_FLT_BT_BWC_HIPASS:
            {
                const bool lo = (type == MI_FLT_BT_BWC_LOPASS);
                const float k = 1.0f / (1.0f + q);
                const uint32_t odd = n & 1;
                if (odd)
                {
                    cascade &c = b.put(0, 0, 0, 1.0, 1.0, 0);
                    c.t[lo ? 0 : 1] = g;
                }
                for (uint32_t j = odd; j < n; j += 2)
                {
                    const float theta = ((j - odd + 1) * kHalfPi) / n;
                    const pole p = bwc_pole(theta, k, false);
                    const float lead = (j == 0) ? g : 1.0;
                    if (lo)
                        b.put(lead, 0, 0, 1.0, 2.0 * k * p.tcos / p.kf, 1.0 / p.kf);
                    else
                        b.put(0, 0, lead, 1.0 / p.kf, 2.0 * k * p.tcos / p.kf, 1.0);
                }
                break;
            }

            case MI_FLT_BT_BWC_ALLPASS:
            {
                const float k = 1.0f / (1.0f + q);
                const uint32_t odd = n & 1;
                if (odd)
                    b.put(-g, g, 0.0, 1.0, 1.0, 0.0);
                for (uint32_t j = odd; j < n; j += 2)
                {
                    const float theta = ((j - odd + 1) * kHalfPi) / n;
                    const pole p = bwc_pole(theta, k, false);
                    cascade &c = b.put(1.0, -2.0 * p.tcos, 1.0, 1.0 / p.kf, 2.0 * k * p.tcos / p.kf, 1.0);
                    if (j == 0)
                        builder::scale_top(c, g);
                }
                break;
            }

            case MI_FLT_BT_BWC_HISHELF:
            case MI_FLT_BT_BWC_LOSHELF:
            {
                const float gain = sqrtf(g);
                const float fg = expf(logf(gain) / (2.0 * n));
                const float k = 1.0f / (1.0 + q * (1.0 - expf(2.0 - gain - 1.0 / gain)));
                for (uint32_t j = 0; j < n; ++j)
                {
                    const float theta = ((2 * j + 1) * kHalfPi) / (2 * n);
                    const pole p = bwc_pole(theta, k, false);
                    const float m0 = p.kf / fg, m1 = 2.0 * k * p.tcos, m2 = fg;
                    cascade &c = (type == MI_FLT_BT_BWC_HISHELF) ? b.put(m0, m1, m2, m2, m1, m0)
                                                                 : b.put(m2, m1, m0, m0, m1, m2);
                    if (j == 0)
                        builder::scale_top(c, gain);
                }
                break;
            }

            case MI_FLT_BT_BWC_LADDERPASS:
            case MI_FLT_BT_BWC_LADDERREJ:
            {
                const bool pass = (type == MI_FLT_BT_BWC_LADDERPASS);
                const size_t slope = n * 2;
                const float gain1 = pass ? sqrtf(g) : sqrtf(1.0 / g);
                const float gain2 = pass ? sqrtf(1.0 / g) : sqrtf(g);
                const float fg1 = expf(logf(gain1) / (2.0 * n));
                const float fg2 = expf(logf(gain2) / (2.0 * n));
                const float k1 = 1.0f / (1.0f + q * (1.0f - expf(2.0f - gain1 - 1.0f / gain1)));
                const float k2 = 1.0f / (1.0f + q * (1.0f - expf(2.0f - gain2 - 1.0f / gain2)));
                const float xf = fp.fFreq2, xf2 = xf * xf;
                for (uint32_t j = 0; j < n; ++j)
                {
                    const float theta = ((2 * j + 1) * kHalfPi) / float(slope);
                    const float k = pass ? k1 : k2, fg = pass ? fg1 : fg2, gain = pass ? gain1 : gain2;
                    pole p = bwc_pole(theta, k, true);
                    {
                        const float m0 = p.kf / fg, m1 = 2.0f * k * p.tcos, m2 = fg;
                        cascade &c = pass ? b.put(m0, m1, m2, m2, m1, m0) : b.put(m2, m1, m0, m0, m1, m2);
                        if (j == 0)
                            builder::scale_top(c, gain);
                    }
                    {
                        const float kf = p.tsin * p.tsin + k1 * k1 * p.tcos * p.tcos;
                        const float h0 = kf / fg1, h1 = 2.0f * k1 * xf * p.tcos, h2 = fg1 * xf2;
                        cascade &c = b.put(fg1, h1, h0 * xf2, h0, h1, h2);
                        if (j == 0)
                            builder::scale_top(c, gain2);
                    }
                }
                break;
            }

            case MI_FLT_BT_BWC_BELL:
            {
                const float fg = expf(logf(g) / float(2 * n));
                const float k = 1.0f / (1.0 + q);
                for (uint32_t j = 0; j < n; ++j)
                {
                    const float theta = ((2 * j + 1) * kHalfPi) / (2 * n);
                    const pole p = bwc_pole(theta, k, false);
                    const float kf = p.kf, tcos = p.tcos;
                    if (g >= 1.0)
                    {
                        b.put(1.0f, 2.0f * k * tcos * fg / kf, 1.0f * fg * fg / kf,
                              1.0f, 2.0f * k * tcos / kf, 1.0f / kf);
                        b.put(1.0f, 2.0f * k * tcos / fg, kf / (fg * fg),
                              1.0f, 2.0f * k * tcos, kf);
                    }
                    else
                    {
                        b.put(1.0f, 2.0f * k * tcos / kf, 1.0f / kf,
                              1.0f, 2.0f * k * tcos / (fg * kf), 1.0f / (fg * fg * kf));
                        b.put(1.0f, 2.0f * k * tcos, kf,
                              1.0f, 2.0f * k * tcos * fg, kf * fg * fg);
                    }
                }
                break;
            }

            case MI_FLT_BT_BWC_BANDPASS:
            {
                const float f2 = fp.fFreq2;
                const float k = 1.0f / (1.0f + q);
                for (uint32_t j = 0; j < n; ++j)
                {
                    const float theta = ((2 * j + 1) * kHalfPi) / (2 * n);
                    const pole p = bwc_pole(theta, k, false);
                    b.put(0, 0, (j == 0) ? g : 1.0f, 1.0f / p.kf, 2.0f * k * p.tcos / p.kf, 1.0f);
                    b.put(1.0f, 0, 0, 1.0f, 2.0f * k * p.tcos * f2 / p.kf, f2 * f2 / p.kf);
                }
                break;
            }

            default:
                break;
        }
    }

    // ---- Linkwitz-Riley: the Butterworth chain emitted twice (Filter.cpp:1397-1487) ------------------------
    void lrx(builder &b, uint32_t type, const mi_filter_params_t &fp)
    {
        uint32_t twin = MI_FLT_NONE;
        switch (type)
        {
            case MI_FLT_BT_LRX_LOPASS:     twin = MI_FLT_BT_BWC_LOPASS; break;
            case MI_FLT_BT_LRX_HIPASS:     twin = MI_FLT_BT_BWC_HIPASS; break;
            case MI_FLT_BT_LRX_LOSHELF:    twin = MI_FLT_BT_BWC_LOSHELF; break;
            case MI_FLT_BT_LRX_HISHELF:    twin = MI_FLT_BT_BWC_HISHELF; break;
            case MI_FLT_BT_LRX_BELL:       twin = MI_FLT_BT_BWC_BELL; break;
            case MI_FLT_BT_LRX_BANDPASS:   twin = MI_FLT_BT_BWC_BANDPASS; break;
            case MI_FLT_BT_LRX_LADDERPASS: twin = MI_FLT_BT_BWC_LADDERPASS; break;
            case MI_FLT_BT_LRX_LADDERREJ:  twin = MI_FLT_BT_BWC_LADDERREJ; break;
            case MI_FLT_BT_LRX_ALLPASS:
            {
                const float k = 1.0f / (1.0f + fp.fQuality);
                const size_t i = b.slope_req * 2;
                for (size_t j = 0; j < i; j += 2)
                {
                    const float theta = ((j + 1) * kHalfPi) / i;
                    const pole p = bwc_pole(theta, k, true);
                    const float d0 = 1.0f / p.kf, d1 = 2.0f * k * p.tcos / p.kf;
                    float xeta = ((j + 0.5f) * kPi) / i;
                    cascade &c1 = b.put(1.0f, -2.0f * cosf(xeta), 1.0f, d0, d1, 1.0f);
                    if (j == 0)
                        builder::scale_top(c1, fp.fGain);
                    xeta = ((j + 1.5f) * kPi) / i;
                    b.put(1.0f, -2.0f * cosf(xeta), 1.0f, d0, d1, 1.0f);
                }
                return;
            }
            default:
                return;
        }
        mi_filter_params_t twice = fp;
        twice.nSlope = b.slope_req * 2;
        twice.fGain  = sqrtf(twice.fGain);
        bwc(b, twin, twice);
        bwc(b, twin, twice);
    }

    // ---- direct digital "APO"/RBJ sections (Filter.cpp:1489-1647) -------------------------------------------
    void apo(builder &b, uint32_t type, const mi_filter_params_t &fp)
    {
        const float omega = kTwoPi * fp.fFreq / float(b.d.sample_rate);
        const float cs = sinf(omega), cc = cosf(omega);
        const float Q = (fp.fQuality > kMinApoQ) ? fp.fQuality : kMinApoQ;
        const float alpha = 0.5f * cs / Q;
        float n0, n1, n2, d0, d1, d2;           // numerator / denominator before normalisation
        switch (type)
        {
            case MI_FLT_DR_APO_LOPASS:
                n0 = fp.fGain * 0.5f * (1.0f - cc); n1 = fp.fGain * (1.0f - cc); n2 = n0;
                d0 = 1.0f + alpha; d1 = -2.0f * cc; d2 = 1.0f - alpha;
                break;
            case MI_FLT_DR_APO_HIPASS:
                n0 = fp.fGain * 0.5f * (1.0f + cc); n1 = fp.fGain * (-1.0f - cc); n2 = n0;
                d0 = 1.0f + alpha; d1 = -2.0f * cc; d2 = 1.0f - alpha;
                break;
            case MI_FLT_DR_APO_BANDPASS:
                n0 = fp.fGain * alpha; n1 = 0.0f; n2 = fp.fGain * -alpha;
                d0 = 1.0f + alpha; d1 = -2.0f * cc; d2 = 1.0f - alpha;
                break;
            case MI_FLT_DR_APO_NOTCH:
                n0 = fp.fGain; n1 = fp.fGain * -2.0f * cc; n2 = n0;
                d0 = 1.0f + alpha; d1 = -2.0f * cc; d2 = 1.0f - alpha;
                break;
            case MI_FLT_DR_APO_ALLPASS:
                n0 = fp.fGain * (1.0f - alpha); n1 = fp.fGain * -2.0f * cc; n2 = fp.fGain * (1.0f + alpha);
                d0 = n2; d1 = n1; d2 = n0;
                break;
            case MI_FLT_DR_APO_PEAKING:
            {
                const float A = sqrtf(fp.fGain);
                n0 = 1.0f + alpha * A; n1 = -2.0f * cc; n2 = 1.0f - alpha * A;
                d0 = 1.0f + alpha / A; d1 = n1; d2 = 1.0f - alpha / A;
                break;
            }
            case MI_FLT_DR_APO_LOSHELF:
            {
                const float A = sqrtf(fp.fGain);
                const float beta = 2.0f * alpha * sqrtf(A);
                n0 = A * ((A + 1.0f) - (A - 1.0f) * cc + beta);
                n1 = 2.0f * A * ((A - 1.0f) - (A + 1.0f) * cc);
                n2 = A * ((A + 1.0f) - (A - 1.0f) * cc - beta);
                d0 = (A + 1.0f) + (A - 1.0f) * cc + beta;
                d1 = -2.0f * ((A - 1.0f) + (A + 1.0f) * cc);
                d2 = (A + 1.0f) + (A - 1.0f) * cc - beta;
                break;
            }
            case MI_FLT_DR_APO_HISHELF:
            {
                const float A = sqrtf(fp.fGain);
                const float beta = 2.0 * alpha * sqrtf(A);
                n0 = A * ((A + 1.0f) + (A - 1.0f) * cc + beta);
                n1 = -2.0f * A * ((A - 1.0f) + (A + 1.0f) * cc);
                n2 = A * ((A + 1.0f) + (A - 1.0f) * cc - beta);
                d0 = (A + 1.0f) - (A - 1.0f) * cc + beta;
                d1 = 2.0f * ((A - 1.0f) - (A + 1.0f) * cc);
                d2 = (A + 1.0f) - (A - 1.0f) * cc - beta;
                break;
            }
            default:
                return;
        }
        const float r = 1.0f / d0;
        const mi_biquad_x1_t f = b.section(n0 * r, n1 * r, n2 * r, -d1 * r, -d2 * r);
        b.mirror(f);
    }

    // Filter::normalize (Filter.cpp:1649-1676): scale the numerator so that |H| = gain at `frequency`
    void normalize(const design &d, mi_biquad_x1_t &f, float frequency, float gain)
    {
        const float sr = float(d.sample_rate);
        const float xf = kTwoPi * std::min(frequency, d.sample_rate * 0.5f) / sr;
        const float cw = cosf(xf), sw = sinf(xf);
        const float c2w = cw * cw - sw * sw, s2w = 2.0f * sw * cw;
        const float alpha = f.b0 + f.b1 * cw + f.b2 * c2w;
        const float beta  = f.b1 * sw + f.b2 * s2w;
        const float gamma = 1.0f - f.a1 * cw - f.a2 * c2w;
        const float delta = -f.a1 * sw - f.a2 * s2w;
        const float mag   = gamma * gamma + delta * delta;
        const float w_re  = alpha * gamma - beta * delta;
        const float w_im  = alpha * delta + beta * gamma;
        const float egain = (gain * mag) / sqrtf(w_re * w_re + w_im * w_im);
        f.b0 *= egain; f.b1 *= egain; f.b2 *= egain;
    }

    // ---- A/B/C/D/K weighting (Filter.cpp:1678-2190) ----------------------------------------------------------
    void weighted(builder &b, uint32_t type)
    {
        const float T = 1.0f / float(b.d.sample_rate);
        auto finish = [&](mi_biquad_x1_t &f, bool norm)
        {
            if (norm)
                normalize(b.d, f, 1000.0f, 1.0f);
            b.mirror(f);
        };
        // zeros 0,0 ; double pole at -p0 (high-pass corner of the A/B/C curves)
        auto corner_hp = [&](float p0)
        {
            const float ww = p0 * T, ws = sinf(ww), wc = cosf(ww);
            const float ka0 = 1.0f / (1.0f + ws);
            const float b0 = 0.5f * (1.0f + wc) * ka0;
            finish(b.section(b0, (-1.0f - wc) * ka0, b0, 2.0f * wc * ka0, (ws - 1.0f) * ka0), true);
        };
        // no zeros ; double pole at -p0 (low-pass corner)
        auto corner_lp = [&](float p0)
        {
            const float ww = p0 * T, ws = sinf(ww), wc = cosf(ww);
            const float ka0 = 1.0f / (1.0f + ws);
            const float b0 = 0.5f * (1.0f - wc) * ka0;
            finish(b.section(b0, (1.0f - wc) * ka0, b0, -2.0f * wc * ka0, (1.0f - ws) * ka0), true);
        };
        // two real poles -p0,-p1 with zeros 0,0 (A curve) or a single zero at 0 (D curve)
        auto two_poles = [&](float p0, float p1, bool a_curve)
        {
            const float ww0 = p0 * T, ww1 = p1 * T;
            const float ws0 = sinf(ww0), wc0 = cosf(ww0), ws1 = sinf(ww1), wc1 = cosf(ww1);
            const float kx0 = 1.0f / (1.0f + ws0 - wc0), kx1 = 1.0f / (1.0f + ws1 - wc1);
            const float ka0 = kx0 * kx1;
            const float ky0 = (1.0f - wc0 - ws0), ky1 = (1.0f - wc1 - ws1);
            const float a1 = -(ky0 * kx0 + ky1 * kx1), a2 = -ky0 * ky1 * ka0;
            if (a_curve)
            {
                const float b0 = ws0 * ws1 * ka0;
                finish(b.section(b0, -2.0f * b0, b0, a1, a2), true);
            }
            else
            {
                const float b0 = ws0 * (1.0f - wc1) * ka0;
                finish(b.section(b0, 0.0f, -b0, a1, a2), true);
            }
        };

        switch (type)
        {
            case MI_FLT_A_WEIGHTED:
                corner_hp(129.4f); two_poles(676.7f, 4636.0f, true); corner_lp(76655.0f);
                b.d.mode = FM_APO;
                break;
            case MI_FLT_B_WEIGHTED:
            {
                corner_hp(129.4f);
                const float ww = 995.9f * T, ws = sinf(ww), wc = cosf(ww);
                const float ka0 = 1.0f / (1.0f + ws - wc);
                const float b0 = ws * ka0;
                finish(b.section(b0, -b0, 0.0f, (ws + wc - 1.0f) * ka0, 0.0f), true);
                corner_lp(76655.0f);
                b.d.mode = FM_APO;
                break;
            }
            case MI_FLT_C_WEIGHTED:
                corner_hp(129.4f); corner_lp(76655.0f);
                b.d.mode = FM_APO;
                break;
            case MI_FLT_D_WEIGHTED:
            {
                two_poles(1776.3f, 7288.5f, false);
                constexpr float p0 = 6401.17f, p1 = 19706.85f, r0 = 1.02f, r1 = 1.092f;
                const float ww0 = p0 * T * 0.5f, ww1 = p1 * T * 0.5f;
                const float wt0 = 1.0f / tanf(ww0), wt1 = 1.0f / tanf(ww1);
                const float ka0 = 1.0f / (1.0f + wt1 * (wt1 + r1));
                finish(b.section((1.0f + wt0 * (wt0 + r0)) * ka0, 2.0f * (1.0f - wt0 * wt0) * ka0,
                                 (1.0f + wt0 * (wt0 - r0)) * ka0, -2.0f * (1.0f - wt1 * wt1) * ka0,
                                 -(1.0f + wt1 * (wt1 - r1)) * ka0), true);
                b.d.mode = FM_APO;
                break;
            }
            case MI_FLT_K_WEIGHTED:
            {
                // ITU-R BS.1770 pre-filter recomputed for the sample rate (constants of Filter.cpp:2117-2120,2152-2153)
                {
                    constexpr float Vh = 1.58486470113f, Vb = 1.25872093023f;
                    constexpr float f0 = 1681.974450955533f, Q = 0.7071752369554196f;
                    const float K = tanf(kPi * f0 * T), K2 = K * K, KQ = K / Q;
                    const float ka0 = 1.0f / (1.0f + KQ + K2);
                    finish(b.section((Vh + Vb * KQ + K2) * ka0, 2.0f * (K2 - Vh) * ka0, (Vh - Vb * KQ + K2) * ka0,
                                     -2.0f * (K2 - 1.0f) * ka0, -(1.0f - KQ + K2) * ka0), false);
                }
                {
                    constexpr float f0 = 38.13547087602444f, Q = 0.5003270373238773f;
                    const float K = tanf(kPi * f0 * T), K2 = K * K, KQ = K / Q;
                    const float ka0 = 1.0f / (1.0f + KQ + K2);
                    finish(b.section(1.0f, -2.0f, 1.0f, -2.0f * (K2 - 1.0f) * ka0, -(1.0f - KQ + K2) * ka0), false);
                }
                b.d.mode = FM_APO;
                break;
            }
            default:
                break;
        }
    }

    // ---- bilinear transform of every analog cascade (Filter.cpp:2225-2267) --------------------------------
    void bilinear(builder &b)
    {
        const double kf = 1.0f / tanf(b.d.params.fFreq * kPi / float(b.d.sample_rate));
        const double kf2 = kf * kf;
        size_t emitted = 0;
        for (const cascade &c : b.d.cascades)
        {
            const double T0 = c.t[0], T1 = c.t[1] * kf, T2 = c.t[2] * kf2;
            const double B0 = c.b[0], B1 = c.b[1] * kf, B2 = c.b[2] * kf2;
            const double N = 1.0 / (B0 + B1 + B2);
            if (++emitted > CHAINS_MAX)
                break;
            b.d.sections.push_back(mi_biquad_x1_t{
                float((T0 + T1 + T2) * N), float(2.0 * (T0 - T2) * N), float((T0 - T1 + T2) * N),
                float(2.0 * (B2 - B0) * N), float((B1 - B2 - B0) * N),      // denominator signs negated
                0.0f, 0.0f, 0.0f });
        }
    }

    // ---- matched Z transform (Filter.cpp:2291-2416) --------------------------------------------------------
    void matched(builder &b)
    {
        const float f = b.d.params.fFreq;
        const float TD = kTwoPi / b.d.sample_rate;
        size_t emitted = 0;
        for (const cascade &c : b.d.cascades)
        {
            float P[2][3], A[2], I[2];
            for (int side = 0; side < 2; ++side)
            {
                const float *p = side ? c.b : c.t;
                float *Q = P[side];
                Q[0] = Q[1] = Q[2] = 0.0f;
                if (p[2] == 0.0f)
                {
                    if (p[1] == 0.0f)
                        Q[0] = p[0];
                    else
                    {
                        const float k = p[1] / f;
                        const float R = -p[0] / k;
                        Q[0] = k;
                        Q[1] = -k * expf(R * TD);
                    }
                }
                else
                {
                    const float k = p[2];
                    const float qa = 1.0f / (f * f);
                    const float qb = p[1] / (f * p[2]);
                    const float qc = p[0] / p[2];
                    float D = qb * qb - 4.0f * qa * qc;
                    if (D >= 0)
                    {
                        D = sqrtf(D);
                        const float R0 = (-qb - D) / (2.0f * qa);
                        const float R1 = (-qb + D) / (2.0f * qa);
                        Q[0] = k;
                        Q[1] = -k * (expf(R0 * TD) + expf(R1 * TD));
                        Q[2] = k * expf((R0 + R1) * TD);
                    }
                    else
                    {
                        D = sqrtf(-D);
                        const float R = -qb / (2.0f * qa);
                        const float K = D / (2.0f * qa);
                        Q[0] = k;
                        Q[1] = -2.0f * k * expf(R * TD) * cosf(K * TD);
                        Q[2] = k * expf(2.0f * R * TD);
                    }
                }
                // amplitude of the discrete part at f/10 and of the analog part at 0.1 (normalised)
                double w  = kPi * 0.2f * b.d.params.fFreq / b.d.sample_rate;
                double re = Q[0] * cos(2.0 * w) + Q[1] * cos(w) + Q[2];
                double im = Q[0] * sin(2.0 * w) + Q[1] * sin(w);
                A[side]   = sqrt(re * re + im * im);
                w         = 0.1;
                re        = p[0] - p[2] * w * w;
                im        = p[1] * w;
                I[side]   = sqrt(re * re + im * im);
            }
            const double AN = (A[1] * I[0]) / (A[0] * I[1]);
            const double N  = 1.0 / P[1][0];
            if (++emitted > CHAINS_MAX)
                break;
            b.d.sections.push_back(mi_biquad_x1_t{
                float(P[0][0] * N * AN), float(P[0][1] * N * AN), float(P[0][2] * N * AN),
                float(-P[1][1] * N), float(-P[1][2] * N), 0.0f, 0.0f, 0.0f });
        }
    }

    inline bool in_range(uint32_t t, uint32_t lo, uint32_t hi) { return t >= lo && t <= hi; }
} // namespace

void limit_params(mi_filter_params_t *fp, uint32_t sample_rate)
{
    const float max_freq = 0.49f * sample_rate;
    fp->nSlope = std::min(std::max(fp->nSlope, 1U), CHAINS_MAX);
    fp->fFreq  = std::min(std::max(fp->fFreq, 0.0f), max_freq);
    fp->fFreq2 = std::min(std::max(fp->fFreq2, 0.0f), max_freq);
}

void design_filter(design *out, const mi_filter_params_t *params, uint32_t sample_rate)
{
    design &d = *out;
    d.mode = FM_BYPASS;
    d.params = *params;
    d.sample_rate = sample_rate;
    d.cascades.clear();
    d.sections.clear();
    limit_params(&d.params, sample_rate);

    builder b(d);
    mi_filter_params_t fp = d.params;
    const uint32_t t = fp.nType;
    const float nf = kPi / float(sample_rate);

    if (in_range(t, MI_FLT_BT_AMPLIFIER, MI_FLT_MT_LRX_ALLPASS))
    {
        // bilinear types sit on odd enumerators, their matched-Z twins right after them
        const bool is_matched = ((t - MI_FLT_BT_AMPLIFIER) & 1) != 0;
        const uint32_t base = is_matched ? t - 1 : t;
        fp.fFreq2 = is_matched ? fp.fFreq / fp.fFreq2                                   // Filter.cpp:261
                               : tanf(fp.fFreq * nf) / tanf(fp.fFreq2 * nf);            // Filter.cpp:201-205,238
        if (base <= MI_FLT_BT_RLC_ENVELOPE)
            rlc(b, base, fp);
        else if (base <= MI_FLT_BT_BWC_ALLPASS)
            bwc(b, base, fp);
        else
            lrx(b, base, fp);
        d.mode = is_matched ? FM_MATCHED : FM_BILINEAR;
    }
    else if (t == MI_FLT_DR_APO_ALLPASS2)
    {
        apo(b, MI_FLT_DR_APO_ALLPASS, fp);
        fp.fFreq = d.params.fFreq2;
        fp.fGain = 1.0f;
        apo(b, MI_FLT_DR_APO_ALLPASS, fp);
        d.mode = FM_APO;
    }
    else if (t == MI_FLT_DR_APO_LADDERPASS)
    {
        apo(b, MI_FLT_DR_APO_HISHELF, fp);
        fp.fFreq = d.params.fFreq2;
        fp.fGain = 1.0f / d.params.fGain;
        apo(b, MI_FLT_DR_APO_HISHELF, fp);
        d.mode = FM_APO;
    }
    else if (t == MI_FLT_DR_APO_LADDERREJ)
    {
        apo(b, MI_FLT_DR_APO_LOSHELF, fp);
        fp.fFreq = d.params.fFreq2;
        apo(b, MI_FLT_DR_APO_HISHELF, fp);
        d.mode = FM_APO;
    }
    else if (in_range(t, MI_FLT_DR_APO_LOPASS, MI_FLT_DR_APO_HISHELF))
    {
        apo(b, t, fp);
        d.mode = FM_APO;
    }
    else if (in_range(t, MI_FLT_A_WEIGHTED, MI_FLT_K_WEIGHTED))
        weighted(b, t);

    if (d.mode == FM_BILINEAR)
        bilinear(b);
    else if (d.mode == FM_MATCHED)
        matched(b);
}

// dsp::filter_transfer_calc_pc / apply_pc of lsp-dsp-lib (call sites Filter.cpp:625-627,648-650): the analog
// cascade evaluated at the normalised frequency w:  H = (t0 - t2 w^2 + j t1 w) / (b0 - b2 w^2 + j b1 w).
static inline void analog_response(float *re, float *im, const cascade &c, float w)
{
    const float w2 = w * w;
    const float t_re = c.t[0] - c.t[2] * w2, t_im = c.t[1] * w;
    const float b_re = c.b[0] - c.b[2] * w2, b_im = c.b[1] * w;
    const float n = 1.0f / (b_re * b_re + b_im * b_im);
    *re = (t_re * b_re + t_im * b_im) * n;
    *im = (t_im * b_re - t_re * b_im) * n;
}

void freq_chart(const design &d, float *c, const float *f, size_t count)
{
    const int mode = d.cascades.empty() ? FM_BYPASS : d.mode;
    const float sr = float(d.sample_rate);
    for (size_t i = 0; i < count; ++i)
    {
        float r_re = 1.0f, r_im = 0.0f;
        if (mode == FM_BILINEAR || mode == FM_MATCHED)
        {
            float w;
            if (mode == FM_BILINEAR)                    // pre-warped (Filter.cpp:609-621)
            {
                const float nf = kPi / sr;
                const float kf = 1.0f / tanf(d.params.fFreq * nf);
                const float lf = d.sample_rate * 0.499f;
                w = tanf((f[i] > lf ? lf : f[i]) * nf) * kf;
            }
            else
                w = f[i] * (1.0f / d.params.fFreq);     // Filter.cpp:641-648
            for (size_t k = 0; k < d.cascades.size(); ++k)
            {
                float h_re, h_im;
                analog_response(&h_re, &h_im, d.cascades[k], w);
                const float n_re = r_re * h_re - r_im * h_im, n_im = r_re * h_im + r_im * h_re;
                r_re = n_re; r_im = n_im;
            }
        }
        else if (mode == FM_APO)                        // digital cascade on the unit circle (Filter.cpp:451-498,662-688)
        {
            const float kf = kTwoPi / sr, lf = d.sample_rate * 0.5f;
            const float a = std::min(f[i], lf) * kf;
            const float cw = cosf(a), sw = sinf(a);
            const float c2w = cw * cw - sw * sw, s2w = 2.0 * sw * cw;
            for (const cascade &q : d.cascades)
            {
                const float alpha = q.t[0] + q.t[1] * cw + q.t[2] * c2w;
                const float beta  = q.t[1] * sw + q.t[2] * s2w;
                const float gamma = q.b[0] + q.b[1] * cw + q.b[2] * c2w;
                const float delta = q.b[1] * sw + q.b[2] * s2w;
                const float mag   = 1.0 / (gamma * gamma + delta * delta);
                const float w_re  = mag * (alpha * gamma - beta * delta);
                const float w_im  = mag * (alpha * delta + beta * gamma);
                const float n_re = r_re * w_re - r_im * w_im, n_im = r_re * w_im + r_im * w_re;
                r_re = n_re; r_im = n_im;
            }
        }
        c[2 * i]     = r_re;
        c[2 * i + 1] = r_im;
    }
}

} // namespace mi
