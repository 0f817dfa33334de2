/*
 * ORACLE -- test infrastructure only (see oracle/__init__.py).
 *
 * CPU restatement of lsp::dspu::Convolver, the zero-latency non-uniform partitioned convolver
 * (/root/reference/src/main/util/Convolver.cpp:77-215 init, :217-313 process; limits in
 * include/lsp-plug.in/dsp-units/util/Convolver.h:28-29).
 *
 * Structure of the impulse response as the reference cuts it (Convolver.cpp:144-197):
 *   head    : taps [0,128)            direct form for partial steps, rank-8 image for whole steps
 *   level i : taps [128*2^i, 128*2^(i+1)), rank 8+i image, i = 0 .. rank-9   ("raising" levels)
 *   blocks  : equal pieces of 2^(rank-1) taps, rank `rank` images            (the long tail)
 * Work schedule per 128-sample step inside a frame of F = 2^(rank-1) samples (Convolver.cpp:230-287):
 *   a level fires whenever its bit of the step counter flips; at the start of a frame the previous frame
 *   is transformed once and the tail blocks are applied a few per step (nBlkInit + fBlkCoef * step).
 *
 * Pinned by the reference's own tests src/test/utest/util/convolver.cpp (test_small :88-136, test_large
 * :184-223) through tests/test_oracle_convolver.py: identical inputs, identical chunking, same tolerances.
 */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

void orc_fastconv_parse(float *dst, const float *src, size_t rank);
void orc_fastconv_apply(float *dst, float *tmp, const float *c1, const float *c2, size_t rank);
void orc_fastconv_parse_apply(float *dst, float *tmp, const float *c, const float *src, size_t rank);
void orc_convolve(float *dst, const float *src, const float *conv, size_t length, size_t count);

enum { STEP_RANK = 8, STEP = 1 << (STEP_RANK - 1) /* 128 */, RANK_MAX = 16 };

typedef struct orc_convolver
{
    size_t  rank, frame, taps;
    size_t  levels, blocks, blocks_done, blk_init;
    float   blk_coef;
    size_t  off;                /* position inside the current frame                        */
    size_t  tail_len;           /* (bins + 1) * frame: accumulated future output            */
    float  *tail;               /* output accumulator, index 0 = first sample of the frame   */
    float  *hist;               /* previous frame followed by the current one (2 * frame)    */
    float  *tmp, *task;         /* scratch image, image of the previous frame                */
    float  *head;               /* first min(taps,128) taps, time domain                     */
    size_t  head_len;
    float  *image_head;         /* rank-8 image of the head                                  */
    float **image_level;        /* per level                                                 */
    float **image_block;        /* per tail block                                            */
} orc_convolver_t;

static size_t min_sz(size_t a, size_t b) { return a < b ? a : b; }

static float *image_of(const float *taps, size_t n, size_t rank)
{
    const size_t half = (size_t)1 << (rank - 1);
    float *padded = (float *)calloc(half, sizeof(float));
    float *img = (float *)malloc(sizeof(float) << (rank + 1));
    memcpy(padded, taps, n * sizeof(float));
    orc_fastconv_parse(img, padded, rank);
    free(padded);
    return img;
}

void orc_convolver_destroy(orc_convolver_t *c)
{
    if (c == NULL)
        return;
    for (size_t i = 0; i < c->levels; ++i) free(c->image_level[i]);
    for (size_t i = 0; i < c->blocks; ++i) free(c->image_block[i]);
    free(c->image_level); free(c->image_block); free(c->image_head);
    free(c->head); free(c->tail); free(c->hist); free(c->tmp); free(c->task);
    free(c);
}

orc_convolver_t *orc_convolver_create(const float *data, size_t count, size_t rank, float phase)
{
    if (count == 0)
        return NULL;                                    /* Convolver.cpp:80-84: stays uninitialised */
    if (rank < STEP_RANK) rank = STEP_RANK;             /* Convolver.cpp:87 */
    if (rank > RANK_MAX)  rank = RANK_MAX;

    orc_convolver_t *c = (orc_convolver_t *)calloc(1, sizeof(*c));
    c->rank     = rank;
    c->frame    = (size_t)1 << (rank - 1);
    c->taps     = count;
    const size_t bins = (count + c->frame - 1) >> (rank - 1);
    c->tail_len = (bins + 1) * c->frame;
    c->tail     = (float *)calloc(c->tail_len, sizeof(float));
    c->hist     = (float *)calloc(2 * c->frame, sizeof(float));
    c->tmp      = (float *)calloc((size_t)2 << rank, sizeof(float));
    c->task     = (float *)calloc((size_t)2 << rank, sizeof(float));
    c->off      = (size_t)(phase * c->frame) % c->frame;            /* Convolver.cpp:139 */

    c->head_len   = min_sz(count, STEP);
    c->head       = (float *)calloc(STEP, sizeof(float));
    memcpy(c->head, data, c->head_len * sizeof(float));
    c->image_head = image_of(data, c->head_len, STEP_RANK);
    data += c->head_len; count -= c->head_len;

    c->image_level = (float **)calloc(RANK_MAX, sizeof(float *));
    for (size_t r = STEP_RANK; count > 0 && r < rank; ++r)
    {
        const size_t n = min_sz(count, (size_t)1 << (r - 1));
        c->image_level[c->levels++] = image_of(data, n, r);
        data += n; count -= n;
    }

    const size_t max_blocks = bins + 1;
    c->image_block = (float **)calloc(max_blocks, sizeof(float *));
    while (count > 0)
    {
        const size_t n = min_sz(count, c->frame);
        c->image_block[c->blocks++] = image_of(data, n, rank);
        data += n; count -= n;
    }

    c->blocks_done = c->blocks;
    const long steps = (long)(c->frame >> (STEP_RANK - 1));
    if (steps <= 1) { c->blk_init = c->blocks; c->blk_coef = 0.0f; }
    else            { c->blk_init = 1; c->blk_coef = ((float)c->blocks + 1e-3f) / ((float)steps - 1.0f); }
    return c;
}

void orc_convolver_process(orc_convolver_t *c, float *dst, const float *src, size_t count)
{
    if (c == NULL)                                      /* Convolver.cpp:219-223 */
    {
        memset(dst, 0, count * sizeof(float));
        return;
    }
    float *cur = c->hist + c->frame;                    /* current frame; previous one sits before it */
    while (count > 0)
    {
        const size_t in_step = c->off & (STEP - 1);
        if (in_step == 0)
        {
            const size_t step = c->off >> (STEP_RANK - 1);
            size_t flips = (step - 1) ^ step;           /* bits that changed since the previous step */
            for (size_t i = 0; i < c->levels; ++i, flips >>= 1)
                if (flips & 1)
                {
                    const size_t r = STEP_RANK + i, span = (size_t)1 << (r - 1);
                    orc_fastconv_parse_apply(&c->tail[c->off], c->tmp, c->image_level[i], cur + c->off - span, r);
                }
            if (c->blocks > 0)
            {
                if (flips & 1)                          /* frame start: transform the previous frame once */
                {
                    orc_fastconv_parse(c->task, c->hist, c->rank);
                    c->blocks_done = 0;
                }
                size_t target = (size_t)(c->blk_init + c->blk_coef * step);
                if (target > c->blocks) target = c->blocks;
                for (; c->blocks_done < target; ++c->blocks_done)
                    orc_fastconv_apply(&c->tail[c->blocks_done * c->frame], c->tmp,
                                       c->image_block[c->blocks_done], c->task, c->rank);
            }
        }
        const size_t n = min_sz(count, STEP - in_step);
        memcpy(cur + c->off, src, n * sizeof(float));
        if (n == STEP)
            orc_fastconv_parse_apply(&c->tail[c->off], c->tmp, c->image_head, src, STEP_RANK);
        else
            orc_convolve(&c->tail[c->off], src, c->head, c->head_len, n);
        memcpy(dst, &c->tail[c->off], n * sizeof(float));

        c->off += n; src += n; dst += n; count -= n;
        if (c->off >= c->frame)
        {
            c->off -= c->frame;
            memmove(c->hist, cur, c->frame * sizeof(float));
            memmove(c->tail, c->tail + c->frame, (c->tail_len - c->frame) * sizeof(float));
            memset(c->tail + c->tail_len - c->frame, 0, c->frame * sizeof(float));
        }
    }
}

size_t orc_convolver_data_size(const orc_convolver_t *c) { return c ? c->taps : 0; }
size_t orc_convolver_rank(const orc_convolver_t *c) { return c ? c->rank : 0; }
