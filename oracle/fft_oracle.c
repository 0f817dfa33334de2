/*
 * ORACLE -- test infrastructure only (see oracle/__init__.py).
 *
 * CPU restatement of the lsp-dsp-lib 1.0.36 primitives the reference's FFT-based units call
 * (un-vendored dependency, modules.mk:29-33; call sites cited per function).  The reference tree
 * pins their semantics only through its unit tests:
 *   - packed_direct_fft / packed_reverse_fft: 2^rank interleaved re/im points, forward unnormalised
 *     with e^{-jwn}, inverse scaled by 1/N (identity + latency tests src/test/utest/util/spectral_proc.cpp:64-66,
 *     src/test/utest/filters/equalizer.cpp:76-81; Analyzer.cpp:270 divides by fft_size itself);
 *   - fastconv_parse / parse_apply / apply: linear convolution helpers whose image format is opaque
 *     (src/test/utest/util/convolver.cpp:113-123 compares against a plain double loop);
 *   - convolve: dst[i+j] += src[i]*conv[j] (convolver.cpp:32-40).
 * Arithmetic is float32 like the reference's; twiddles are rounded from double.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_RANK 20

static float *tw_cache[ORC_MAX_RANK + 1];      /* per rank: N/2 pairs (cos, -sin) */

static const float *twiddles(size_t rank)
{
    if (tw_cache[rank] != NULL)
        return tw_cache[rank];
    const size_t n = (size_t)1 << rank;
    float *t = (float *)malloc(sizeof(float) * (n > 1 ? n : 2));
    for (size_t k = 0; k < n / 2; ++k)
    {
        const double a = -2.0 * M_PI * (double)k / (double)n;
        t[2 * k]     = (float)cos(a);
        t[2 * k + 1] = (float)sin(a);
    }
#ifdef _OPENMP
#pragma omp critical(orc_tw)
#endif
    {
        if (tw_cache[rank] == NULL)
            tw_cache[rank] = t;
        else
            free(t);
    }
    return tw_cache[rank];
}

/* In-place radix-2 decimation-in-time transform; sign = -1 forward, +1 inverse (unscaled). */
static void fft_inplace(float *x, size_t rank, int inverse)
{
    const size_t n = (size_t)1 << rank;
    for (size_t i = 1, j = 0; i < n; ++i)          /* bit reversal permutation */
    {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1)
            j ^= bit;
        j ^= bit;
        if (i < j)
        {
            float tr = x[2 * i], ti = x[2 * i + 1];
            x[2 * i] = x[2 * j]; x[2 * i + 1] = x[2 * j + 1];
            x[2 * j] = tr;       x[2 * j + 1] = ti;
        }
    }
    const float *tw = twiddles(rank);
    for (size_t len = 2; len <= n; len <<= 1)
    {
        const size_t half = len >> 1, step = n / len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < half; ++k)
            {
                const float wr = tw[2 * k * step];
                const float wi = inverse ? -tw[2 * k * step + 1] : tw[2 * k * step + 1];
                float *a = &x[2 * (i + k)], *b = &x[2 * (i + k + half)];
                const float vr = b[0] * wr - b[1] * wi;
                const float vi = b[0] * wi + b[1] * wr;
                b[0] = a[0] - vr; b[1] = a[1] - vi;
                a[0] = a[0] + vr; a[1] = a[1] + vi;
            }
    }
}

/* dsp::packed_direct_fft(dst, src, rank): Equalizer.cpp:287,536; SpectralProcessor.cpp:166,220;
 * MultiSpectralProcessor.cpp:343; Analyzer.cpp:357.  dst may alias src. */
void orc_packed_direct_fft(float *dst, const float *src, size_t rank)
{
    const size_t n = (size_t)1 << rank;
    if (dst != src)
        memmove(dst, src, 2 * n * sizeof(float));
    fft_inplace(dst, rank, 0);
}

/* dsp::packed_reverse_fft(dst, src, rank): Equalizer.cpp:332,538; SpectralProcessor.cpp:168;
 * MultiSpectralProcessor.cpp:363.  Scaled by 1/N. */
void orc_packed_reverse_fft(float *dst, const float *src, size_t rank)
{
    const size_t n = (size_t)1 << rank;
    if (dst != src)
        memmove(dst, src, 2 * n * sizeof(float));
    fft_inplace(dst, rank, 1);
    const float k = 1.0f / (float)n;
    for (size_t i = 0; i < 2 * n; ++i)
        dst[i] *= k;
}

/* dsp::pcomplex_r2c / c2r / mod / mul2; ascending order so dst = src + N aliasing works. */
void orc_pcomplex_r2c(float *dst, const float *src, size_t n)
{
    /* SpectralProcessor.cpp:164-165 calls this with src = dst + n: walk upwards, read before write */
    for (size_t i = 0; i < n; ++i)
    {
        const float v = src[i];
        dst[2 * i] = v;
        dst[2 * i + 1] = 0.0f;
    }
}

void orc_pcomplex_c2r(float *dst, const float *src, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        dst[i] = src[2 * i];
}

void orc_pcomplex_mod(float *dst, const float *src, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        dst[i] = sqrtf(src[2 * i] * src[2 * i] + src[2 * i + 1] * src[2 * i + 1]);
}

void orc_pcomplex_mul2(float *dst, const float *src, size_t n)
{
    for (size_t i = 0; i < n; ++i)
    {
        const float ar = dst[2 * i], ai = dst[2 * i + 1], br = src[2 * i], bi = src[2 * i + 1];
        dst[2 * i]     = ar * br - ai * bi;
        dst[2 * i + 1] = ar * bi + ai * br;
    }
}

/* dsp::fastconv_parse(dst, src, rank): Convolver.cpp:159,174,191,270; Equalizer.cpp:342,345.
 * Image = interleaved spectrum (2^(rank+1) floats) of src[0..2^(rank-1)) zero-padded to 2^rank. */
void orc_fastconv_parse(float *dst, const float *src, size_t rank)
{
    const size_t n = (size_t)1 << rank, half = n >> 1;
    for (size_t i = 0; i < half; ++i)
    {
        const float v = src[i];
        dst[2 * i] = v;
        dst[2 * i + 1] = 0.0f;
    }
    memset(dst + 2 * half, 0, 2 * half * sizeof(float));
    fft_inplace(dst, rank, 0);
}

/* dsp::fastconv_apply(dst, tmp, c1, c2, rank): Convolver.cpp:282.  dst[0..2^rank) += IFFT(c1 * c2). */
void orc_fastconv_apply(float *dst, float *tmp, const float *c1, const float *c2, size_t rank)
{
    const size_t n = (size_t)1 << rank;
    for (size_t i = 0; i < n; ++i)
    {
        const float ar = c1[2 * i], ai = c1[2 * i + 1], br = c2[2 * i], bi = c2[2 * i + 1];
        tmp[2 * i]     = ar * br - ai * bi;
        tmp[2 * i + 1] = ar * bi + ai * br;
    }
    fft_inplace(tmp, rank, 1);
    const float k = 1.0f / (float)n;
    for (size_t i = 0; i < n; ++i)
        dst[i] += tmp[2 * i] * k;
}

/* dsp::fastconv_parse_apply(dst, tmp, c, src, rank): Convolver.cpp:256,293; Equalizer.cpp:484,493. */
void orc_fastconv_parse_apply(float *dst, float *tmp, const float *c, const float *src, size_t rank)
{
    const size_t n = (size_t)1 << rank;
    float *img = (float *)malloc(2 * n * sizeof(float));
    orc_fastconv_parse(img, src, rank);
    orc_fastconv_apply(dst, tmp, img, c, rank);
    free(img);
}

/* dsp::convolve(dst, src, conv, length, count): Convolver.cpp:295; pinned by utest/util/convolver.cpp:32-40. */
void orc_convolve(float *dst, const float *src, const float *conv, size_t length, size_t count)
{
    for (size_t i = 0; i < count; ++i)
    {
        const float k = src[i];
        for (size_t j = 0; j < length; ++j)
            dst[i + j] += k * conv[j];
    }
}

/* Exact linear convolution in double (ground truth for the parity yardstick). */
void orc_convolve_f64(double *dst, const float *src, const float *conv, size_t length, size_t count)
{
    for (size_t i = 0; i < count; ++i)
    {
        const double k = src[i];
        if (k == 0.0)
            continue;
        for (size_t j = 0; j < length; ++j)
            dst[i + j] += k * (double)conv[j];
    }
}
