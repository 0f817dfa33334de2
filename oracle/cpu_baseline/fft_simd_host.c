/*
 * CPU baseline of bench.py (measurement infrastructure, not product; never linked into liborc.so or the product):
 * the FFT-shaped dsp:: primitives of lsp-dsp-lib 1.0.36 (un-vendored: /root/reference/modules.mk:29-33) in a form the
 * host's vector units can run -- what the reference's own SIMD path (absent here) would be timed as.  Same entry points and
 * semantics as the scalar restatement oracle/fft_oracle.c (orc_packed_direct_fft, orc_packed_reverse_fft, orc_fastconv_parse /
 * _apply / _parse_apply, orc_convolve, orc_pcomplex_*), so that the oracle's Convolver (oracle/convolver_oracle.c) and the
 * block logic in fft_units_host.c link against either; tests/test_cpu_baseline.py holds the two against each other.
 *
 * A transform of N = 2^rank complex points is done as n1 x n2 (four-step): radix-4 Stockham passes ALONG the first index with
 * all columns side by side -- every butterfly is a loop over contiguous floats with scalar twiddles, which gcc -O3
 * -march=native turns into full-width vector code (AVX-512 on the GPU box's EPYC 9575F) --, the N twiddles, a transposition,
 * and the same along the other index; split real / imaginary planes inside.  fastconv images are kept split as well (the
 * format is opaque to the callers, Convolver.cpp:156-197): re[N] | im[N].
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define SIMD_MAX_RANK 20
#define SIMD_MIN_RANK 4        /* below: the plain loops */

typedef struct simd_plan
{
    size_t  rank, n, n1, n2;
    float  *w1r, *w1i;          /* e^{-2 pi i k / n1}, k < n1 */
    float  *w2r, *w2i;          /* e^{-2 pi i k / n2}, k < n2 */
    float  *tr, *ti;            /* [n1][n2]: e^{-2 pi i k1 j2 / N} */
} simd_plan;

static simd_plan *plans[SIMD_MAX_RANK + 1];

static void *amalloc(size_t bytes)
{
    void *p = NULL;
    if (posix_memalign(&p, 64, (bytes + 63) & ~(size_t)63) != 0)
        return NULL;
    return p;
}

static void roots(float *wr, float *wi, size_t m)
{
    for (size_t k = 0; k < m; ++k)
    {
        const double a = -2.0 * M_PI * (double)k / (double)m;
        wr[k] = (float)cos(a);
        wi[k] = (float)sin(a);
    }
}

static const simd_plan *plan_for(size_t rank)
{
    if (plans[rank] != NULL)
        return plans[rank];
    simd_plan *p = (simd_plan *)calloc(1, sizeof(simd_plan));
    p->rank = rank;
    p->n = (size_t)1 << rank;
    p->n1 = (size_t)1 << (rank / 2);
    p->n2 = p->n / p->n1;
    p->w1r = (float *)amalloc(p->n1 * sizeof(float)); p->w1i = (float *)amalloc(p->n1 * sizeof(float));
    p->w2r = (float *)amalloc(p->n2 * sizeof(float)); p->w2i = (float *)amalloc(p->n2 * sizeof(float));
    p->tr = (float *)amalloc(p->n * sizeof(float));   p->ti = (float *)amalloc(p->n * sizeof(float));
    roots(p->w1r, p->w1i, p->n1);
    roots(p->w2r, p->w2i, p->n2);
    for (size_t k1 = 0; k1 < p->n1; ++k1)
        for (size_t j2 = 0; j2 < p->n2; ++j2)
        {
            const double a = -2.0 * M_PI * (double)(k1 * j2) / (double)p->n;
            p->tr[k1 * p->n2 + j2] = (float)cos(a);
            p->ti[k1 * p->n2 + j2] = (float)sin(a);
        }
#ifdef _OPENMP
#pragma omp critical(simd_plan)
#endif
    {
        if (plans[rank] == NULL)
            plans[rank] = p;
    }
    return plans[rank];             /* (a plan made twice in a race leaks once: measurement infrastructure) */
}

/* per-thread scratch: four planes of N floats per rank */
static __thread float *scratch[SIMD_MAX_RANK + 1];
static float *scratch_for(size_t rank)
{
    if (scratch[rank] == NULL)
        scratch[rank] = (float *)amalloc(6 * ((size_t)1 << rank) * sizeof(float));
    return scratch[rank];
}

/* Stockham passes along the first index of [m][v] (split planes); the result lands in (*or, *oi), one of the two pairs. */
static void cols_fft(size_t m, size_t v, float *restrict xr, float *restrict xi, float *restrict yr, float *restrict yi,
                     const float *wr, const float *wi, int inverse, float **out_r, float **out_i)
{
    size_t n = m, s = 1;
    const float sg = inverse ? -1.0f : 1.0f;
    while (n >= 4)
    {
        const size_t n1 = n / 4, len = s * v, step = m / n;
        for (size_t p = 0; p < n1; ++p)
        {
            const float w1r = wr[p * step], w1i = sg * wi[p * step];
            const float w2r = wr[2 * p * step], w2i = sg * wi[2 * p * step];
            const float w3r = wr[3 * p * step], w3i = sg * wi[3 * p * step];
            const float *restrict ar = xr + s * p * v,            *restrict ai = xi + s * p * v;
            const float *restrict br = xr + s * (p + n1) * v,     *restrict bi = xi + s * (p + n1) * v;
            const float *restrict cr = xr + s * (p + 2 * n1) * v, *restrict ci = xi + s * (p + 2 * n1) * v;
            const float *restrict dr = xr + s * (p + 3 * n1) * v, *restrict di = xi + s * (p + 3 * n1) * v;
            float *restrict o0r = yr + s * (4 * p) * v,     *restrict o0i = yi + s * (4 * p) * v;
            float *restrict o1r = yr + s * (4 * p + 1) * v, *restrict o1i = yi + s * (4 * p + 1) * v;
            float *restrict o2r = yr + s * (4 * p + 2) * v, *restrict o2i = yi + s * (4 * p + 2) * v;
            float *restrict o3r = yr + s * (4 * p + 3) * v, *restrict o3i = yi + s * (4 * p + 3) * v;
#pragma omp simd
            for (size_t i = 0; i < len; ++i)
            {
                const float apcr = ar[i] + cr[i], apci = ai[i] + ci[i];
                const float amcr = ar[i] - cr[i], amci = ai[i] - ci[i];
                const float bpdr = br[i] + dr[i], bpdi = bi[i] + di[i];
                /* j (b - d), j = i for the forward transform's "- j" below, mirrored for the inverse */
                const float jr = -sg * (bi[i] - di[i]), ji = sg * (br[i] - dr[i]);
                const float t1r = amcr - jr, t1i = amci - ji;
                const float t2r = apcr - bpdr, t2i = apci - bpdi;
                const float t3r = amcr + jr, t3i = amci + ji;
                o0r[i] = apcr + bpdr;               o0i[i] = apci + bpdi;
                o1r[i] = t1r * w1r - t1i * w1i;     o1i[i] = t1r * w1i + t1i * w1r;
                o2r[i] = t2r * w2r - t2i * w2i;     o2i[i] = t2r * w2i + t2i * w2r;
                o3r[i] = t3r * w3r - t3i * w3i;     o3i[i] = t3r * w3i + t3i * w3r;
            }
        }
        float *t;
        t = xr; xr = yr; yr = t;
        t = xi; xi = yi; yi = t;
        n /= 4;
        s *= 4;
    }
    if (n == 2)
    {
        const size_t len = s * v;
        const float *restrict ar = xr, *restrict ai = xi, *restrict br = xr + len, *restrict bi = xi + len;
        float *restrict o0r = yr, *restrict o0i = yi, *restrict o1r = yr + len, *restrict o1i = yi + len;
#pragma omp simd
        for (size_t i = 0; i < len; ++i)
        {
            o0r[i] = ar[i] + br[i]; o0i[i] = ai[i] + bi[i];
            o1r[i] = ar[i] - br[i]; o1i[i] = ai[i] - bi[i];
        }
        float *t;
        t = xr; xr = yr; yr = t;
        t = xi; xi = yi; yi = t;
    }
    *out_r = xr;
    *out_i = xi;
}

static void transpose(float *restrict dst, const float *restrict src, size_t rows, size_t cols)   /* src[rows][cols] -> dst[cols][rows] */
{
    enum { TB = 16 };
    for (size_t r0 = 0; r0 < rows; r0 += TB)
        for (size_t c0 = 0; c0 < cols; c0 += TB)
            for (size_t r = r0; r < r0 + TB && r < rows; ++r)
                for (size_t c = c0; c < c0 + TB && c < cols; ++c)
                    dst[c * rows + r] = src[r * cols + c];
}

/* (ar, ai) hold the N points in planes 0, 1 of the rank's scratch; the transform comes back in (*or, *oi) (scratch planes). */
static void fft_split(const simd_plan *p, float *sc, int inverse, float **out_r, float **out_i)
{
    const size_t n = p->n, n1 = p->n1, n2 = p->n2;
    float *ar = sc, *ai = sc + n, *br = sc + 2 * n, *bi = sc + 3 * n;
    float *rr, *ri;
    cols_fft(n1, n2, ar, ai, br, bi, p->w1r, p->w1i, inverse, &rr, &ri);
    float *fr = (rr == ar) ? br : ar, *fi = (ri == ai) ? bi : ai;          /* the free pair */
    /* twiddles, then [n1][n2] -> [n2][n1] */
    {
        const float *restrict tr = p->tr, *restrict ti = p->ti;
        const float sg = inverse ? -1.0f : 1.0f;
        float *restrict xr = rr, *restrict xi = ri;
#pragma omp simd
        for (size_t i = 0; i < n; ++i)
        {
            const float vr = xr[i], vi = xi[i], wr = tr[i], wi = sg * ti[i];
            xr[i] = vr * wr - vi * wi;
            xi[i] = vr * wi + vi * wr;
        }
    }
    transpose(fr, rr, n1, n2);
    transpose(fi, ri, n1, n2);
    cols_fft(n2, n1, fr, fi, rr, ri, p->w2r, p->w2i, inverse, out_r, out_i);
}

/* plain O(N^2)-free fallback for tiny ranks: radix-2 on interleaved data (as oracle/fft_oracle.c does) */
static void tiny_fft(float *x, size_t rank, int inverse)
{
    const size_t n = (size_t)1 << rank;
    for (size_t i = 1, j = 0; i < n; ++i)
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
    for (size_t len = 2; len <= n; len <<= 1)
    {
        const size_t half = len >> 1;
        for (size_t k = 0; k < half; ++k)
        {
            const double a = (inverse ? 2.0 : -2.0) * M_PI * (double)k / (double)len;
            const float wr = (float)cos(a), wi = (float)sin(a);
            for (size_t i = 0; i < n; i += len)
            {
                float *u = &x[2 * (i + k)], *w = &x[2 * (i + k + half)];
                const float vr = w[0] * wr - w[1] * wi, vi = w[0] * wi + w[1] * wr;
                w[0] = u[0] - vr; w[1] = u[1] - vi;
                u[0] = u[0] + vr; u[1] = u[1] + vi;
            }
        }
    }
}

static void fft_interleaved(float *dst, const float *src, size_t rank, int inverse)
{
    const size_t n = (size_t)1 << rank;
    if (rank < SIMD_MIN_RANK)
    {
        if (dst != src)
            memmove(dst, src, 2 * n * sizeof(float));
        tiny_fft(dst, rank, inverse);
        if (inverse)
            for (size_t i = 0; i < 2 * n; ++i)
                dst[i] *= 1.0f / (float)n;
        return;
    }
    const simd_plan *p = plan_for(rank);
    float *sc = scratch_for(rank);
    float *restrict ar = sc, *restrict ai = sc + n;
    for (size_t i = 0; i < n; ++i)
    {
        ar[i] = src[2 * i];
        ai[i] = src[2 * i + 1];
    }
    float *rr, *ri;
    fft_split(p, sc, inverse, &rr, &ri);
    const float k = inverse ? 1.0f / (float)n : 1.0f;
    for (size_t i = 0; i < n; ++i)
    {
        dst[2 * i] = rr[i] * k;
        dst[2 * i + 1] = ri[i] * k;
    }
}

/* ---- the dsp:: entry points (same names and meaning as oracle/fft_oracle.c) ------------------------------------------ */

void orc_packed_direct_fft(float *dst, const float *src, size_t rank) { fft_interleaved(dst, src, rank, 0); }
void orc_packed_reverse_fft(float *dst, const float *src, size_t rank) { fft_interleaved(dst, src, rank, 1); }

void orc_pcomplex_r2c(float *dst, const float *src, size_t n)
{
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

/* image (2^(rank+1) floats): re[N] | im[N] of the transform of src[0 .. N/2) zero-padded to N = 2^rank */
void orc_fastconv_parse(float *dst, const float *src, size_t rank)
{
    const size_t n = (size_t)1 << rank, half = n >> 1;
    if (rank < SIMD_MIN_RANK)
    {
        float tmp[2 << SIMD_MIN_RANK];
        for (size_t i = 0; i < n; ++i)
        {
            tmp[2 * i] = (i < half) ? src[i] : 0.0f;
            tmp[2 * i + 1] = 0.0f;
        }
        tiny_fft(tmp, rank, 0);
        for (size_t i = 0; i < n; ++i)
        {
            dst[i] = tmp[2 * i];
            dst[n + i] = tmp[2 * i + 1];
        }
        return;
    }
    const simd_plan *p = plan_for(rank);
    float *sc = scratch_for(rank);
    memcpy(sc, src, half * sizeof(float));
    memset(sc + half, 0, half * sizeof(float));
    memset(sc + n, 0, n * sizeof(float));
    float *rr, *ri;
    fft_split(p, sc, 0, &rr, &ri);
    memcpy(dst, rr, n * sizeof(float));
    memcpy(dst + n, ri, n * sizeof(float));
}

/* dst[0 .. N) += Re IFFT(c1 * c2) */
void orc_fastconv_apply(float *dst, float *tmp, const float *c1, const float *c2, size_t rank)
{
    const size_t n = (size_t)1 << rank;
    (void)tmp;
    if (rank < SIMD_MIN_RANK)
    {
        float t[2 << SIMD_MIN_RANK];
        for (size_t i = 0; i < n; ++i)
        {
            t[2 * i]     = c1[i] * c2[i] - c1[n + i] * c2[n + i];
            t[2 * i + 1] = c1[i] * c2[n + i] + c1[n + i] * c2[i];
        }
        tiny_fft(t, rank, 1);
        for (size_t i = 0; i < n; ++i)
            dst[i] += t[2 * i] * (1.0f / (float)n);
        return;
    }
    const simd_plan *p = plan_for(rank);
    float *sc = scratch_for(rank);
    {
        float *restrict xr = sc, *restrict xi = sc + n;
        const float *restrict ar = c1, *restrict ai = c1 + n, *restrict br = c2, *restrict bi = c2 + n;
#pragma omp simd
        for (size_t i = 0; i < n; ++i)
        {
            xr[i] = ar[i] * br[i] - ai[i] * bi[i];
            xi[i] = ar[i] * bi[i] + ai[i] * br[i];
        }
    }
    float *rr, *ri;
    fft_split(p, sc, 1, &rr, &ri);
    const float k = 1.0f / (float)n;
    const float *restrict r = rr;
#pragma omp simd
    for (size_t i = 0; i < n; ++i)
        dst[i] += r[i] * k;
}

void orc_fastconv_parse_apply(float *dst, float *tmp, const float *c, const float *src, size_t rank)
{
    const size_t n = (size_t)1 << rank;
    float *sc = scratch_for(rank < SIMD_MIN_RANK ? SIMD_MIN_RANK : rank);
    float *img = sc + 4 * n;                               /* planes 4, 5 of the scratch: not touched by the transforms */
    orc_fastconv_parse(img, src, rank);
    orc_fastconv_apply(dst, tmp, img, c, rank);
}

void orc_convolve(float *dst, const float *src, const float *conv, size_t length, size_t count)
{
    for (size_t i = 0; i < count; ++i)
    {
        const float k = src[i];
        float *restrict d = dst + i;
#pragma omp simd
        for (size_t j = 0; j < length; ++j)
            d[j] += k * conv[j];
    }
}
