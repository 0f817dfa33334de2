/*
 * CPU baseline of bench.py for the FFT-shaped configurations (measurement infrastructure, not product):
 * the reference's block logic on vectorised dsp:: primitives (fft_simd_host.c: four-step radix-4 transforms on split planes,
 * full-width vector code under -O3 -march=native; -DCPU_BASELINE_SCALAR: the oracle's scalar ones, ../fft_oracle.c), OpenMP
 * over the channels -- one object per channel as the reference has it.
 *
 *   cpu_equalizer_fir_run : Equalizer::process in EQM_FIR mode, /root/reference/src/main/filters/Equalizer.cpp:460-571
 *                           (per N = 2^fir_rank samples: shift vOutBuffer, one fastconv_parse_apply of rank + 1)
 *   cpu_analyzer_run      : Analyzer::process, /root/reference/src/main/util/Analyzer.cpp:299-409 (ring ingest, per period
 *                           one windowed 2^rank-point packed_direct_fft, pcomplex_mod, mix2 smoothing)
 *   cpu_convolver_bank_*  : Convolver::process, /root/reference/src/main/util/Convolver.cpp:217-313, through the oracle's
 *                           restatement of the non-uniform partitioned algorithm (../convolver_oracle.c), one object per channel
 * lsp-dsp-lib's hand-written kernels are not available offline (modules.mk:29-33); this is what stands in for them.
 */
#ifdef CPU_BASELINE_SCALAR
#include "../fft_oracle.c"
static void warm(size_t rank) { (void)twiddles(rank); }
#else
#include "fft_simd_host.c"
static void warm(size_t rank) { (void)plan_for(rank); }
#endif
#include "../convolver_oracle.c"

#include <omp.h>

/* conv: [channels][4N] images made by cpu_fastconv_image; in/out: [blocks % ring][channels][N]; state: [channels][4N] floats
 * (vInBuffer 2N | vOutBuffer 2N as in Equalizer.cpp:99-121), zeroed by the caller.  Returns the threads used. */
int cpu_equalizer_fir_run(float *out, const float *in, size_t channels, size_t fir_rank, size_t blocks, size_t ring,
                          const float *conv, float *state, int threads)
{
    const size_t n = (size_t)1 << fir_rank;
    warm(fir_rank + 1);
    int used = 1;
#pragma omp parallel num_threads(threads)
    {
#pragma omp single
        used = omp_get_num_threads();
        float *tmp = (float *)malloc(4 * n * sizeof(float));
#pragma omp for schedule(static)
        for (size_t c = 0; c < channels; ++c)
        {
            float *inb = state + c * 4 * n, *outb = inb + 2 * n;
            for (size_t b = 0; b < blocks; ++b)
            {
                const float *x = in + ((b % ring) * channels + c) * n;
                float *y = out + ((b % ring) * channels + c) * n;
                /* Equalizer.cpp:476-484: the buffer is full -> shift the output, convolve the collected block */
                memcpy(outb, outb + n, n * sizeof(float));
                memset(outb + n, 0, n * sizeof(float));
                orc_fastconv_parse_apply(outb, tmp, conv + c * 4 * n, inb, fir_rank + 1);
                /* :505-520: take the new block in, hand the finished one out */
                memcpy(inb, x, n * sizeof(float));
                memcpy(y, outb, n * sizeof(float));
            }
        }
        free(tmp);
    }
    return used;
}

void cpu_fastconv_image(float *dst, const float *fir, size_t fir_rank)
{
    orc_fastconv_parse(dst, fir, fir_rank + 1);
}

/* in: [frames % ring][channels][hop]; ring buffers: [channels][bufsize]; amp: [channels][N/2 + 1]; window: [N] */
int cpu_analyzer_run(float *amp, const float *in, size_t channels, size_t rank, size_t hop, size_t frames, size_t ring,
                     const float *window, float tau, float *buffers, size_t bufsize, int threads)
{
    const size_t n = (size_t)1 << rank, bins = n / 2 + 1;
    warm(rank);
    int used = 1;
#pragma omp parallel num_threads(threads)
    {
#pragma omp single
        used = omp_get_num_threads();
        float *sig = (float *)malloc(2 * n * sizeof(float)), *spec = (float *)malloc(2 * n * sizeof(float));
        float *mod = (float *)malloc(n * sizeof(float));
#pragma omp for schedule(static)
        for (size_t c = 0; c < channels; ++c)
        {
            float *buf = buffers + c * bufsize, *a = amp + c * bins;
            size_t head = 0;
            for (size_t f = 0; f < frames; ++f)
            {
                const float *x = in + ((f % ring) * channels + c) * hop;
                for (size_t i = 0; i < hop; ++i)                  /* Analyzer.cpp:386-400: into the ring */
                    buf[(head + i) % bufsize] = x[i];
                head = (head + hop) % bufsize;
                const size_t first = (head + bufsize - n) % bufsize;
                for (size_t i = 0; i < n; ++i)                    /* :349-352: window, real -> complex */
                {
                    sig[2 * i] = buf[(first + i) % bufsize] * window[i];
                    sig[2 * i + 1] = 0.0f;
                }
                orc_packed_direct_fft(spec, sig, rank);           /* :353 */
                orc_pcomplex_mod(mod, spec, bins);                /* :356 */
                for (size_t i = 0; i < bins; ++i)                 /* :361 mix2 */
                    a[i] = a[i] * (1.0f - tau) + mod[i] * tau;
            }
        }
        free(sig); free(spec); free(mod);
    }
    return used;
}

/* ---- Convolver: one oracle object per channel, whole frames, OpenMP over the channels -------------------------------- */
typedef struct { size_t channels; orc_convolver_t **obj; } cpu_convolver_bank;

void *cpu_convolver_bank_create(const float *irs, size_t taps, size_t channels, size_t rank, int threads)
{
    cpu_convolver_bank *b = (cpu_convolver_bank *)calloc(1, sizeof(cpu_convolver_bank));
    b->channels = channels;
    b->obj = (orc_convolver_t **)calloc(channels, sizeof(orc_convolver_t *));
    for (size_t r = 4; r <= rank; ++r)
        warm(r);
#pragma omp parallel for schedule(static) num_threads(threads)
    for (size_t c = 0; c < channels; ++c)
        b->obj[c] = orc_convolver_create(irs + c * taps, taps, rank, 0.0f);
    return b;
}

/* x, y: [frames % ring][channels][frame].  Returns the threads used. */
int cpu_convolver_bank_run(void *bank, float *y, const float *x, size_t frame, size_t frames, size_t ring, int threads)
{
    cpu_convolver_bank *b = (cpu_convolver_bank *)bank;
    int used = 1;
#pragma omp parallel num_threads(threads)
    {
#pragma omp single
        used = omp_get_num_threads();
#pragma omp for schedule(static)
        for (size_t c = 0; c < b->channels; ++c)
            for (size_t f = 0; f < frames; ++f)
                orc_convolver_process(b->obj[c], y + ((f % ring) * b->channels + c) * frame,
                                      x + ((f % ring) * b->channels + c) * frame, frame);
    }
    return used;
}

void cpu_convolver_bank_destroy(void *bank)
{
    cpu_convolver_bank *b = (cpu_convolver_bank *)bank;
    if (b == NULL)
        return;
    for (size_t c = 0; c < b->channels; ++c)
        orc_convolver_destroy(b->obj[c]);
    free(b->obj);
    free(b);
}
