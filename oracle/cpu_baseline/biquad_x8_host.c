/*
 * ORACLE / CPU BASELINE -- test and measurement infrastructure only.  Nothing under lsp-dsp-units_amd/ may include,
 * link or call this file; bench.py's cpu_baseline leg and tests/test_cpu_baseline.py are its only users.
 *
 * What the host can do with the reference's structure for BASELINE config C2: one dsp::biquad_process_x8 pass per
 * channel and block (FilterBank::process, /root/reference/src/main/filters/FilterBank.cpp:256-291, the x8 branch at
 * :267-273).  lsp-dsp-lib 1.0.36 (un-vendored, modules.mk:29-33) implements x8 as a software pipeline: the eight
 * sections sit in the eight lanes of a SIMD register, lane j works on sample t-j at step t, and the lanes shift by
 * one after every step.  This file restates that structure with GCC vector extensions (one 8-float register per
 * channel, AVX2 / AVX-512VL or NEON chosen by -march=native), KCH channels interleaved per thread to cover the
 * FMA -> shuffle latency of the step, persistent threads over fixed channel ranges, and any number of consecutive
 * blocks inside ONE parallel region (channels are independent: no barrier between blocks).
 *
 * Per section the arithmetic is the reference's transposed direct form II with pre-negated denominators
 * (Filter.cpp:2261-2262):  y = b0 x + d0;  d0' = d1 + b1 x + a1 y;  d1' = b2 x + a2 y.
 * Built -O3 -march=native -ffp-contract=fast (the compiler may fuse a*b+c, as lsp-dsp-lib's FMA3 variants do), so
 * its output agrees with oracle/biquad_oracle.c to round-off, not bit for bit: it is a timing baseline, the parity
 * oracle stays biquad_oracle.c (checked against each other in tests/test_cpu_baseline.py).
 */
#define _POSIX_C_SOURCE 200112L
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef float v8f __attribute__((vector_size(32)));
typedef int   v8i __attribute__((vector_size(32)));

typedef struct { v8f b0, b1, b2, a1, a2, d0, d1; } x8_bank_t;

static inline v8f shift_in(v8f y, float x)
{
    /* lanes move up by one, lane 0 takes the next input sample */
    const v8i idx = { 8, 0, 1, 2, 3, 4, 5, 6 };
    const v8f xv = { x, x, x, x, x, x, x, x };
    return __builtin_shuffle(y, xv, idx);
}

/* One step of all eight sections of one channel; returns y of every lane. */
static inline v8f x8_step(x8_bank_t *f, v8f s)
{
    const v8f y = f->b0 * s + f->d0;
    const v8f p = f->b1 * s + f->a1 * y;
    f->d0 = f->d1 + p;
    f->d1 = f->b2 * s + f->a2 * y;
    return y;
}

/* The same with a per-lane mask (pipeline fill and drain: lanes whose sample index is outside the block keep their state). */
static inline v8f x8_step_masked(x8_bank_t *f, v8f s, v8i m)
{
    const v8f y  = f->b0 * s + f->d0;
    const v8f p  = f->b1 * s + f->a1 * y;
    const v8f n0 = f->d1 + p;
    const v8f n1 = f->b2 * s + f->a2 * y;
    f->d0 = (v8f)(((v8i)n0 & m) | ((v8i)f->d0 & ~m));
    f->d1 = (v8f)(((v8i)n1 & m) | ((v8i)f->d1 & ~m));
    return y;
}

#define KCH 4       /* channels interleaved per inner loop */

/* KCH channels x n samples through their 8-section banks (n >= 8). */
static void x8_process_group(float *const dst[KCH], const float *const src[KCH], size_t n, x8_bank_t f[KCH])
{
    v8f s[KCH];
    /* fill: steps 0..6, lane j active when j <= t */
    for (int c = 0; c < KCH; ++c)
        s[c] = (v8f){ src[c][0], 0, 0, 0, 0, 0, 0, 0 };
    for (size_t t = 0; t < 7; ++t)
    {
        v8i m;
        for (int j = 0; j < 8; ++j)
            m[j] = ((size_t)j <= t) ? -1 : 0;
        for (int c = 0; c < KCH; ++c)
        {
            const v8f y = x8_step_masked(&f[c], s[c], m);
            s[c] = shift_in(y, src[c][t + 1]);
        }
    }
    /* steady state: steps 7..n-1 produce outputs 0..n-8 */
    for (size_t t = 7; t + 1 < n; ++t)
    {
        for (int c = 0; c < KCH; ++c)
        {
            const v8f y = x8_step(&f[c], s[c]);
            dst[c][t - 7] = y[7];
            s[c] = shift_in(y, src[c][t + 1]);
        }
    }
    /* last full step (t = n-1), then drain: steps n..n+6, lane j active when t - j < n */
    for (int c = 0; c < KCH; ++c)
    {
        const v8f y = x8_step(&f[c], s[c]);
        dst[c][n - 8] = y[7];
        s[c] = shift_in(y, 0.0f);
    }
    for (size_t t = n; t < n + 7; ++t)
    {
        v8i m;
        for (int j = 0; j < 8; ++j)
            m[j] = (t - (size_t)j < n) ? -1 : 0;
        for (int c = 0; c < KCH; ++c)
        {
            const v8f y = x8_step_masked(&f[c], s[c], m);
            dst[c][t - 7] = y[7];
            s[c] = shift_in(y, 0.0f);
        }
    }
}

static void load_bank(x8_bank_t *f, const float *coef /* 8 x {b0,b1,b2,a1,a2} */, const float *state /* 8 x {d0,d1} */)
{
    for (int j = 0; j < 8; ++j)
    {
        f->b0[j] = coef[5 * j + 0]; f->b1[j] = coef[5 * j + 1]; f->b2[j] = coef[5 * j + 2];
        f->a1[j] = coef[5 * j + 3]; f->a2[j] = coef[5 * j + 4];
        f->d0[j] = state[2 * j];    f->d1[j] = state[2 * j + 1];
    }
}

static void store_state(const x8_bank_t *f, float *state)
{
    for (int j = 0; j < 8; ++j)
    {
        state[2 * j]     = f->d0[j];
        state[2 * j + 1] = f->d1[j];
    }
}

/*
 * `blocks` consecutive blocks of [channels][n] samples (block b at dst/src + b*block_stride, wrapping over `ring`
 * distinct blocks), 8 sections per channel, state [channels][8][2] carried from block to block.
 * threads <= 0: all of OpenMP's threads.  channels must be a multiple of KCH, n >= 8.
 * Returns the number of threads used.
 */
int cpu_biquad_x8_run(float *dst, const float *src, size_t channels, size_t n, size_t blocks, size_t ring,
                      const float *coef, float *state, int threads)
{
    int used = 1;
    if (channels % KCH != 0 || n < 8 || ring == 0)
        return -1;
#ifdef _OPENMP
    if (threads <= 0)
        threads = omp_get_max_threads();
    #pragma omp parallel num_threads(threads)
#endif
    {
#ifdef _OPENMP
        const size_t nt = (size_t)omp_get_num_threads(), me = (size_t)omp_get_thread_num();
        #pragma omp single
        used = (int)nt;
#else
        const size_t nt = 1, me = 0;
#endif
        const size_t groups = channels / KCH;
        const size_t g0 = groups * me / nt, g1 = groups * (me + 1) / nt;
        /* a thread keeps its channels for the whole run; blocks are walked in stream order */
        x8_bank_t *fs = NULL;
        if (g1 > g0 && posix_memalign((void **)&fs, 64, (g1 - g0) * KCH * sizeof(x8_bank_t)) != 0)
            fs = NULL;
        if (fs != NULL)
        {
            for (size_t g = g0; g < g1; ++g)
                for (int c = 0; c < KCH; ++c)
                    load_bank(&fs[(g - g0) * KCH + c], coef + (g * KCH + c) * 40, state + (g * KCH + c) * 16);
            for (size_t b = 0; b < blocks; ++b)
            {
                const size_t off = (b % ring) * channels * n;
                for (size_t g = g0; g < g1; ++g)
                {
                    float *d[KCH];
                    const float *s[KCH];
                    for (int c = 0; c < KCH; ++c)
                    {
                        d[c] = dst + off + (g * KCH + c) * n;
                        s[c] = src + off + (g * KCH + c) * n;
                    }
                    x8_process_group(d, s, n, &fs[(g - g0) * KCH]);
                }
            }
            for (size_t g = g0; g < g1; ++g)
                for (int c = 0; c < KCH; ++c)
                    store_state(&fs[(g - g0) * KCH + c], state + (g * KCH + c) * 16);
            free(fs);
        }
    }
    return used;
}
