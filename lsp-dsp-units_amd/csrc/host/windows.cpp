// Host-side window and spectral-envelope generators (product code).
// Counterparts of lsp::dspu::windows::* (reference: src/main/misc/windows.cpp:62-401) and
// lsp::dspu::envelope::reverse_noise_lin (src/main/misc/envelope.cpp:40-123).  They run at reconfigure time
// only; the tables are uploaded to the device by the units that use them.
#include <cmath>
#include <cstddef>

#include "../mi_common.h"

namespace mi
{
namespace
{
    // a0 - a1 cos(f i) + a2 cos(2 f i) - a3 cos(3 f i) + a4 cos(4 f i), f = 2 pi / (n - 1)
    void cosine_sum(float *dst, size_t n, const float *a, int terms)
    {
        if (n == 0)
            return;
        const float f1 = 2.0f * M_PI / (n - 1);
        float f[5] = { 0.0f, f1, f1 * 2.0f, f1 * 3.0f, f1 * 4.0f };
        if (terms == 2)                     // Hann / Hamming (windows.cpp:139-160)
        {
            for (size_t i = 0; i < n; ++i)
                dst[i] = a[0] - a[1] * cosf(i * f[1]);
        }
        else if (terms == 3)                // Blackman (windows.cpp:162-180): note the double 0.5
        {
            for (size_t i = 0; i < n; ++i)
                dst[i] = a[0] - 0.5 * cosf(i * f[1]) + a[2] * cosf(i * f[2]);
        }
        else if (terms == 4)                // Nuttall family (windows.cpp:182-214)
        {
            for (size_t i = 0; i < n; ++i)
                dst[i] = a[0] - a[1] * cosf(i * f[1]) + a[2] * cosf(i * f[2]) - a[3] * cosf(i * f[3]);
        }
        else                                // flat top, normalised at the centre (windows.cpp:216-236)
        {
            const float norm = 1.0f / (a[0] - a[1] * cosf(n * 0.5 * f[1]) + a[2] * cosf(n * 0.5 * f[2])
                                       - a[3] * cosf(n * 0.5 * f[3]) + a[4] * cosf(n * 0.5 * f[4]));
            for (size_t i = 0; i < n; ++i)
                dst[i] = norm * (a[0] - a[1] * cosf(i * f[1]) + a[2] * cosf(i * f[2]) - a[3] * cosf(i * f[3])
                                 + a[4] * cosf(i * f[4]));
        }
    }

    // The triangular family (misc/windows.cpp:70-99: triangular_general and, with the zeros one step outside / on the ends,
    // Bartlett-Fejer and the plain triangle): a tent 1 - |i - middle| * slope over `feet`, the distance between its zeros.
    // Filled from both ends towards the middle: i - middle and (count - 1 - i) - middle are exact negatives of each other
    // (half-integers far below 2^24), so both flanks get the value the reference computes for them.
    void tent(float *w, size_t count, int ends /* > 0: zeros one step outside, < 0: on the first and last sample, 0: half a step outside */)
    {
        if (count == 0)
            return;
        const size_t feet = (ends > 0) ? count + 1 : (ends < 0) ? count - 1 : count;
        if (feet == 0)
        {
            w[0] = 0.0f;
            return;
        }
        const float slope = 2.0f / float(feet), middle = (count - 1) * 0.5;
        for (size_t lo = 0, hi = count - 1; lo <= hi; ++lo, --hi)
        {
            const float v = 1.0f - fabs((lo - middle) * slope);
            w[lo] = v;
            w[hi] = v;
            if (hi == 0)
                break;
        }
    }
} // namespace

// Parameters of the window families (the reference's *_general forms, misc/windows.h:71-155); the named windows
// are these with the reference's constants (windows.cpp:91-99,152-160,179-181,199-213,232-234,300-302,315-317,
// 332-334,351-353,398-400).  Returns how many parameters the family takes, 0 for the parameter-free windows.
int window_defaults(int type, size_t n, float *q)
{
    switch (type)
    {
        case MI_WINDOW_HANN:             q[0] = 0.5f;  q[1] = 0.5f;  return 2;
        case MI_WINDOW_HAMMING:          q[0] = 0.54f; q[1] = 0.46f; return 2;
        case MI_WINDOW_BLACKMAN:         q[0] = 0.16f; return 1;
        case MI_WINDOW_NUTTALL:          q[0] = 0.355768f;  q[1] = 0.487396f;  q[2] = 0.144232f;  q[3] = 0.012604f;  return 4;
        case MI_WINDOW_BLACKMAN_NUTTALL: q[0] = 0.3635819f; q[1] = 0.4891775f; q[2] = 0.1365995f; q[3] = 0.0106411f; return 4;
        case MI_WINDOW_BLACKMAN_HARRIS:  q[0] = 0.35875f;   q[1] = 0.48829f;   q[2] = 0.14128f;   q[3] = 0.01168f;   return 4;
        case MI_WINDOW_FLAT_TOP:         q[0] = 1.0f; q[1] = 1.93f; q[2] = 1.29f; q[3] = 0.388f; q[4] = 0.028f; return 5;
        case MI_WINDOW_TRIANGULAR:       q[0] = 0.0f;  return 1;
        case MI_WINDOW_BARTLETT_FEJER:   q[0] = -1.0f; return 1;
        case MI_WINDOW_GAUSSIAN:         q[0] = 0.4f;  return 1;
        case MI_WINDOW_POISSON:          q[0] = n * 0.5f; return 1;
        case MI_WINDOW_BARTLETT_HANN:    q[0] = 0.62f; q[1] = 0.48f; q[2] = 0.38f; return 3;
        case MI_WINDOW_HANN_POISSON:     q[0] = 2.0f;  return 1;
        case MI_WINDOW_TUKEY:            q[0] = 0.5f;  return 1;
        default:                         return 0;
    }
}

void make_window_params(float *dst, size_t n, int type, const float *q);

void make_window(float *dst, size_t n, int type)
{
    float q[5] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
    window_defaults(type, n, q);
    make_window_params(dst, n, type, q);
}

void make_window_params(float *dst, size_t n, int type, const float *q)
{
    switch (type)
    {
        case MI_WINDOW_HANN:
        case MI_WINDOW_HAMMING:         { const float a[] = { q[0], q[1] }; cosine_sum(dst, n, a, 2); break; }
        case MI_WINDOW_BLACKMAN:
        {
            const float alpha = q[0], a2 = alpha * 0.5f;
            const float a[] = { 0.5f - a2, 0.5f, a2 };
            cosine_sum(dst, n, a, 3);
            break;
        }
        case MI_WINDOW_NUTTALL:
        case MI_WINDOW_BLACKMAN_NUTTALL:
        case MI_WINDOW_BLACKMAN_HARRIS:  cosine_sum(dst, n, q, 4); break;
        case MI_WINDOW_FLAT_TOP:         cosine_sum(dst, n, q, 5); break;
        case MI_WINDOW_RECTANGULAR:
            for (size_t i = 0; i < n; ++i)
                dst[i] = 1.0f;
            break;
        case MI_WINDOW_TRIANGULAR:
        case MI_WINDOW_BARTLETT_FEJER:  tent(dst, n, int(q[0])); break;
        case MI_WINDOW_PARZEN:                              // windows.cpp:139-...: piecewise cubic
        {
            if (n == 0)
                break;
            const float n_2 = 0.5 * n, n_4 = 0.25 * n, n__2 = 1.0 / n_2;
            for (size_t i = 0; i < n; ++i)
            {
                const float x = fabs(i - n_2), k = x * n__2, p = 1.0f - k;
                dst[i] = (x <= n_4) ? 1.0f - 6.0f * k * k * p : 2.0f * p * p * p;
            }
            break;
        }
        case MI_WINDOW_WELCH:
        {
            if (n == 0)
                break;
            const float c = (n - 1) * 0.5f, mc = 1.0f / c;
            for (size_t i = 0; i < n; ++i)
            {
                const float t = (i - c) * mc;
                dst[i] = 1.0f - t * t;
            }
            break;
        }
        case MI_WINDOW_COSINE:                              // sin(pi i / n), windows.cpp:238-246
        {
            if (n == 0)
                break;
            const float f = M_PI / n;
            for (size_t i = 0; i < n; ++i)
                dst[i] = sinf(f * i);
            break;
        }
        case MI_WINDOW_SQR_COSINE:
        {
            if (n == 0)
                break;
            const float f = M_PI / n;
            for (size_t i = 0; i < n; ++i)
            {
                const float a = sinf(f * i);
                dst[i] = a * a;
            }
            break;
        }
        case MI_WINDOW_CUBIC:                               // windows.cpp:261-281 (its n == 1 case writes dst[1]; we write dst[0])
        {
            if (n <= 1)
            {
                if (n == 1)
                    dst[0] = 1.0f;
                break;
            }
            size_t middle = n >> 1;
            const float kx = 1.0f / middle;
            size_t i = 0;
            for (; i < middle; ++i)
            {
                const float x = i * kx;
                dst[i] = x * x * (3.0f - 2.0f * x);
            }
            middle = n - 1;
            for (; i < n; ++i)
                dst[i] = 1.0f - dst[middle - i];
            break;
        }
        case MI_WINDOW_GAUSSIAN:
        {
            const float s = q[0];
            if (n == 0 || s > 0.5)
                break;
            const float c = (n - 1) * 0.5f, sc = 1.0f / (c * s);
            for (size_t i = 0; i < n; ++i)
            {
                const float v = (i - c) * sc;
                dst[i] = expf(-0.5f * v * v);
            }
            break;
        }
        case MI_WINDOW_POISSON:
        {
            const float c = (n - 1) * 0.5f;
            const float t = -1.0f / q[0];
            for (size_t i = 0; i < n; ++i)
                dst[i] = expf(t * fabs(i - c));
            break;
        }
        case MI_WINDOW_BARTLETT_HANN:
        {
            if (n == 0)
                break;
            const float a0 = q[0], a1 = q[1], a2 = q[2];
            const float k1 = 1.0f / (n - 1), k2 = 2.0f * M_PI * k1;
            for (size_t i = 0; i < n; ++i)
                dst[i] = a0 - a1 * fabs(i * k1 - 0.5f) - a2 * cosf(i * k2);
            break;
        }
        case MI_WINDOW_HANN_POISSON:
        {
            if (n == 0)
                break;
            const float a = q[0];
            const float f = 2.0f * M_PI / (n - 1);
            const float k1 = (n - 1) * 0.5, k2 = -a / k1;
            for (size_t i = 0; i < n; ++i)
                dst[i] = (0.5 - 0.5 * cosf(i * f)) * expf(k2 * fabs(k1 - i));
            break;
        }
        case MI_WINDOW_LANCZOS:
        {
            if (n == 0)
                break;
            const float k = 2.0f * M_PI / (n - 1);
            for (size_t i = 0; i < n; ++i)
            {
                const float x = k * i - M_PI;
                dst[i] = (x == 0.0f) ? 1.0f : sinf(x) / x;
            }
            break;
        }
        case MI_WINDOW_TUKEY:
        {
            if (n == 0)
                break;
            const float a = q[0];
            if (a == 0.0f)                                  // windows.cpp:375-379
            {
                for (size_t i = 0; i < n; ++i)
                    dst[i] = 1.0f;
                break;
            }
            const size_t last = n - 1;
            const size_t b1 = 0.5 * a * last, b2 = last - b1;
            const float k = M_PI * 2.0f / (a * last);
            const float x = M_PI - 2.0f * M_PI / a;
            for (size_t i = 0; i < n; ++i)
            {
                if (i <= b1)      dst[i] = 0.5f + 0.5f * cosf(k * i - M_PI);
                else if (i > b2)  dst[i] = 0.5f + 0.5f * cosf(k * i + x);
                else              dst[i] = 1.0f;
            }
            break;
        }
        default:
            break;
    }
}

// envelope::noise_lin / reverse_noise_lin (envelope.cpp:40-123): (f / center)^k on a linear frequency grid, k the
// spectral slope of the colour (reverse: of the opposite colour)
namespace
{
    bool colour_exponent(int type, bool reverse, float *k)
    {
        constexpr float LOG10_2 = 0.30102999566398119521f;  // M_LOG10_2
        constexpr float PLUS_4_5 = 4.5f / (20.0f * LOG10_2), MINUS_4_5 = -4.5f / (20.0f * LOG10_2);
        float v;
        switch (type)
        {
            case MI_ENVELOPE_WHITE_NOISE:   v = 0.0f;  break;
            case MI_ENVELOPE_PINK_NOISE:    v = -0.5f; break;
            case MI_ENVELOPE_BROWN_NOISE:   v = -1.0f; break;
            case MI_ENVELOPE_BLUE_NOISE:    v = 0.5f;  break;
            case MI_ENVELOPE_VIOLET_NOISE:  v = 1.0f;  break;
            case MI_ENVELOPE_PLUS_4_5_DB:   v = PLUS_4_5;  break;
            case MI_ENVELOPE_MINUS_4_5_DB:  v = MINUS_4_5; break;
            default:
                return false;
        }
        *k = reverse ? -v : v;
        return true;
    }
}

void make_noise_lin(float *dst, float first, float last, float center, size_t n, int type, bool reverse)
{
    float k;
    if (!colour_exponent(type, reverse, &k))
        return;
    if (type == MI_ENVELOPE_WHITE_NOISE)                    // fill_one, whatever n is
    {
        for (size_t i = 0; i < n; ++i)
            dst[i] = 1.0f;
        return;
    }
    if (n <= 1)                                             // envelope.cpp:42-47
    {
        if (n > 0)
            dst[0] = 1.0f;
        return;
    }
    const float kf = 1.0f / center;
    first *= kf;
    last  *= kf;
    const float df = (last - first) / (n - 1);
    for (size_t i = 0; i < n; ++i)
        dst[i] = first + df * i;
    if (dst[0] <= 0.0f)
        dst[0] = dst[1];
    for (size_t i = 0; i < n; ++i)                          // dsp::powvc1(dst, k, n)
        dst[i] = powf(dst[i], k);
}

void make_reverse_noise_lin(float *dst, float first, float last, float center, size_t n, int type)
{
    make_noise_lin(dst, first, last, center, n, type, true);
}

// envelope::noise_log / reverse_noise_log (envelope.cpp:170-268): the same power law on a logarithmic frequency grid,
// f_i = first exp(i ln(last / first) / (n - 1)) in units of the centre frequency
void make_noise_log(float *dst, float first, float last, float center, size_t n, int type, bool reverse)
{
    float k;
    if (!colour_exponent(type, reverse, &k))
        return;
    if (type == MI_ENVELOPE_WHITE_NOISE)
    {
        for (size_t i = 0; i < n; ++i)
            dst[i] = 1.0f;
        return;
    }
    if (n <= 1)                                             // envelope.cpp:172-177
    {
        if (n > 0)
            dst[0] = 1.0f;
        return;
    }
    const float kf = 1.0f / center;
    first *= kf;
    last  *= kf;
    const float df = logf(last / first) / (n - 1);
    for (size_t i = 0; i < n; ++i)                          // dsp::exp1, mul_k2, powvc1
        dst[i] = powf(expf(df * i) * first, k);
}

// envelope::noise_list / reverse_noise_list (envelope.cpp:272-344): (freqs[i] / center)^k
void make_noise_list(float *dst, const float *freqs, float center, size_t n, int type, bool reverse)
{
    float k;
    if (!colour_exponent(type, reverse, &k))
        return;
    if (type == MI_ENVELOPE_WHITE_NOISE)
    {
        for (size_t i = 0; i < n; ++i)
            dst[i] = 1.0f;
        return;
    }
    const float kf = 1.0f / center;
    for (size_t i = 0; i < n; ++i)                          // dsp::mul_k3, powvc1
        dst[i] = powf(freqs[i] * kf, k);
}

} // namespace mi

extern "C" {

int mi_window(float *dst, size_t n, int type)
{
    MI_REQUIRE(n == 0 || dst != nullptr, MI_EINVAL, "mi_window: NULL destination");
    MI_REQUIRE(type >= 0 && type < MI_WINDOW_TOTAL, MI_EINVAL, "mi_window: unknown window %d", type);
    mi::make_window(dst, n, type);
    return MI_OK;
}

int mi_window_general(float *dst, size_t n, int type, const float *params, uint32_t count)
{
    MI_REQUIRE(n == 0 || dst != nullptr, MI_EINVAL, "mi_window_general: NULL destination");
    MI_REQUIRE(type >= 0 && type < MI_WINDOW_TOTAL, MI_EINVAL, "mi_window_general: unknown window %d", type);
    float q[5] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
    const int want = mi::window_defaults(type, n, q);
    MI_REQUIRE(want > 0, MI_EINVAL, "mi_window_general: window %d has no parameters", type);
    MI_REQUIRE(params != nullptr && int(count) == want, MI_EINVAL, "mi_window_general: window %d takes %d parameters", type, want);
    for (int i = 0; i < want; ++i)
        q[i] = params[i];
    mi::make_window_params(dst, n, type, q);
    return MI_OK;
}

int mi_envelope_noise_lin(float *dst, float first, float last, float center, size_t n, int type)
{
    MI_REQUIRE(n == 0 || dst != nullptr, MI_EINVAL, "mi_envelope_noise_lin: NULL destination");
    MI_REQUIRE(type >= 0 && type < MI_ENVELOPE_TOTAL, MI_EINVAL, "mi_envelope_noise_lin: unknown envelope %d", type);
    mi::make_noise_lin(dst, first, last, center, n, type, false);
    return MI_OK;
}

int mi_envelope_noise_log(float *dst, float first, float last, float center, size_t n, int type, int reverse)
{
    MI_REQUIRE(n == 0 || dst != nullptr, MI_EINVAL, "mi_envelope_noise_log: NULL destination");
    MI_REQUIRE(type >= 0 && type < MI_ENVELOPE_TOTAL, MI_EINVAL, "mi_envelope_noise_log: unknown envelope %d", type);
    mi::make_noise_log(dst, first, last, center, n, type, reverse != 0);
    return MI_OK;
}

int mi_envelope_noise_list(float *dst, const float *freqs, float center, size_t n, int type, int reverse)
{
    MI_REQUIRE(n == 0 || (dst != nullptr && freqs != nullptr), MI_EINVAL, "mi_envelope_noise_list: NULL buffer");
    MI_REQUIRE(type >= 0 && type < MI_ENVELOPE_TOTAL, MI_EINVAL, "mi_envelope_noise_list: unknown envelope %d", type);
    mi::make_noise_list(dst, freqs, center, n, type, reverse != 0);
    return MI_OK;
}

int mi_envelope_reverse_noise_lin(float *dst, float first, float last, float center, size_t n, int type)
{
    MI_REQUIRE(n == 0 || dst != nullptr, MI_EINVAL, "mi_envelope_reverse_noise_lin: NULL destination");
    MI_REQUIRE(type >= 0 && type < MI_ENVELOPE_TOTAL, MI_EINVAL, "mi_envelope_reverse_noise_lin: unknown envelope %d", type);
    mi::make_reverse_noise_lin(dst, first, last, center, n, type);
    return MI_OK;
}

} // extern "C"
