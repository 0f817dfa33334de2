/*
 * mi_dspu.h -- C-ABI of the MI355X (gfx950) implementation of the lsp-dsp-units
 * block-streaming hot path.
 *
 * This is the drop-in boundary: plain C, plain pointers and sizes, no C++ or
 * torch types.  Every entry point names the reference interface it replaces
 * (paths relative to the reference tree, lsp-plugins/lsp-dsp-units 1.0.36).
 *
 * Conventions
 *   - every function returns MI_OK (0) or a negative MI_E* code;
 *     mi_dspu_last_error() returns a thread-local human-readable message;
 *   - sample buffers are DEVICE pointers to float32, laid out
 *     [channel][stride] (one reference object == one channel == one row);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); no
 *     entry point synchronises the stream unless its name says so;
 *   - objects are opaque handles created/destroyed by the library; they are
 *     not thread-safe (same contract as the reference objects).
 *   - there is NO CPU fallback: without a usable HIP device every compute
 *     entry point fails with MI_ENODEV.
 */
#ifndef MI_DSPU_H_
#define MI_DSPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_DSPU_ABI_VERSION     1

enum
{
    MI_OK            =  0,
    MI_EINVAL        = -1,   /* bad argument (reference: silently clamped or `false`)   */
    MI_ENOMEM        = -2,   /* allocation failure (reference: init() returns false)     */
    MI_ENODEV        = -3,   /* no HIP device / kernel image not loadable                 */
    MI_EHIP          = -4,   /* a HIP runtime call failed; see mi_dspu_last_error()       */
    MI_ESTATE        = -5    /* object not initialised (reference: STATUS_BAD_STATE)      */
};

/* ---- library / device plumbing ------------------------------------------------------- */

int         mi_dspu_abi_version(void);
const char *mi_dspu_last_error(void);
/* Number of visible HIP devices (0 when none; never fails). */
int         mi_dspu_device_count(void);
/* Select the device used by the calling thread (hipSetDevice). */
int         mi_dspu_set_device(int device);

int         mi_dspu_malloc(void **dev_ptr, size_t bytes);
int         mi_dspu_free(void *dev_ptr);
int         mi_dspu_memset(void *dev_ptr, int value, size_t bytes, void *stream);
int         mi_dspu_copy_h2d(void *dev_dst, const void *host_src, size_t bytes, void *stream);
int         mi_dspu_copy_d2h(void *host_dst, const void *dev_src, size_t bytes, void *stream);
int         mi_dspu_copy_d2d(void *dev_dst, const void *dev_src, size_t bytes, void *stream);
int         mi_dspu_stream_create(void **stream);
int         mi_dspu_stream_destroy(void *stream);
int         mi_dspu_stream_synchronize(void *stream);

/* Timing helper used by bench.py: runs `fn`-less event pairs on `stream`. */
int         mi_dspu_event_create(void **event);
int         mi_dspu_event_destroy(void *event);
int         mi_dspu_event_record(void *event, void *stream);
int         mi_dspu_event_synchronize(void *event);
int         mi_dspu_event_elapsed_ms(float *ms, void *start, void *stop);
/*
 * Environment switches -- the library reads exactly these five, at every call:
 *   MI_DSPU_COMPAT_BITS=1     runs of 4096-point blocks (Equalizer FIR/FFT, SpectralProcessor with a mask, SpectralSplitter at
 *                             rank 12) stay on the workgroup kernels, whose results are the block-by-block calls' BIT FOR BIT,
 *                             instead of the wave-resident transform kernels (1.3 - 1.7 x faster, within 1e-6 of the calls)
 *   MI_DSPU_EXACT_IIR=1       biquad banks START in the exact mode (mi_biquad_bank_set_exact below): the reference's serial recurrence,
 *                             its output bit for bit, for hosts that cannot be changed to call mi_dspu_set_exact_iir_default
 *   MI_CONV_TWO_LAUNCH=1      Convolver: frame launch + tail launch instead of the one-launch step   } fall-backs behind two
 *   MI_ILUFS_TWO_LAUNCHES=1   ILUFSMeter: bookkeeping in a launch of its own                         } in-launch hand-overs that
 *                             rest on gfx950's dispatch order / memory-side atomics (a wait that does not end raises a fault)
 *   MI_DSPU_TEST_PATH=a,b,..  TEST HOOK, not a tuning knob: sends calls down another LIVE path of the library -- one that other
 *                             inputs take anyway -- so that differential tests can hold two paths against each other:
 *                             blocks_loop, conv_batch_finish, conv_frame_per_launch, crossover_unfused, ilufs_rows_apart,
 *                             loudness_scalar, splitter_hop_launches, analyzer_strobe_per_launch
 */
/*
 * Arm a pair of events for the calling thread: the next hot-path kernel this
 * thread launches (the dominant kernel of the next process() call) records them
 * at its own begin and end (hipExtLaunchKernelGGL), so their elapsed time is the
 * kernel's launch duration as a profiler reports it.  One-shot.
 */
int         mi_dspu_profile_next_launch(void *start_event, void *stop_event);
/*
 * The hot-path kernel the calling thread launched last (its name as the library spells it, template arguments included;
 * "" before the first one).  For tests and profiles that must know WHICH launch a call took -- a run of blocks riding its
 * one-launch kernel or the block-by-block fallback -- not only that the result is right.  No reference counterpart.
 */
const char *mi_dspu_last_launch(void);
/*
 * The first 16 hex digits of the SHA-256 of a source file of csrc/ (e.g. "biquad.hip") as it was when THIS library was built,
 * or NULL for a name the build did not see.  Counter summaries under profiles/ carry the hashes of the sources they were
 * measured on; bench.py quotes a committed counter only while the library it loaded was built from the same source.
 */
const char *mi_dspu_source_sha(const char *file);
/*
 * hipGraph helpers for launch-bound inner loops: a run of steady-state *_process() calls on `stream` is captured once
 * and replayed (hipStreamBeginCapture / hipStreamEndCapture + hipGraphInstantiate / hipGraphLaunch).
 * begin: `stream` must be a created (non-NULL) stream.  end: returns an executable graph handle.
 * Banks without ring positions (biquad, crossover, dynamic filters) replay any run.  The others (delay, ring buffer,
 * convolver, equalizer, spectral processor, analyzer, splitter, loudness meters) hand ring positions kept on the host to
 * their kernels by value: end_capture() checks that the captured calls bring every such bank back to the positions it
 * started from and returns MI_ESTATE otherwise (capture a whole number of position periods: one lap of a delay line,
 * an even number of analyzer strobes, ...).  A refused capture is not lost work and leaves no inconsistent state: the
 * captured calls are executed ONCE inside end_capture() (synchronising the stream), so device rings and host positions
 * agree as after eager calls; no graph is returned.  Dynamic filters refuse capture (MI_ESTATE) while a clear of their
 * filter memory is pending (right after init / set_sample_rate): make one eager call first.  On a stream captured with
 * hipStreamBeginCapture directly those banks' process() returns MI_ESTATE and changes nothing.
 * launch: a bank that has re-made device buffers since the capture (the convolver's ring of frames is grown once, by
 * the bank's first mi_convolver_bank_process_blocks batch) makes the graph stale: MI_ESTATE, nothing is launched --
 * capture again.
 */
int         mi_dspu_graph_begin_capture(void *stream);
int         mi_dspu_graph_end_capture(void *stream, void **graph_exec);
int         mi_dspu_graph_launch(void *graph_exec, void *stream);
int         mi_dspu_graph_destroy(void *graph_exec);

/* ---- biquad cascade bank -------------------------------------------------------------- */
/*
 * One digital section, same field order/size as dsp::biquad_x1_t used by
 * FilterBank::add_chain() (include/lsp-plug.in/dsp-units/filters/FilterBank.h:86,
 * fields as written in src/main/filters/Filter.cpp:2254-2265):
 *      y[n] = b0 x[n] + b1 x[n-1] + b2 x[n-2] + a1 y[n-1] + a2 y[n-2]
 * (denominator signs pre-negated).  p0..p2 are padding.
 */
typedef struct mi_biquad_x1
{
    float b0, b1, b2;
    float a1, a2;
    float p0, p1, p2;
} mi_biquad_x1_t;

/*
 * mi_biquad_bank: `channels` independent lsp::dspu::FilterBank objects
 * (filters/FilterBank.h:34-139), each holding up to `max_sections` biquads that
 * run strictly in series (src/main/filters/FilterBank.cpp:256-291).
 */
typedef struct mi_biquad_bank mi_biquad_bank_t;

/* FilterBank::init(filters), FilterBank.cpp:62-92, for every channel. */
int mi_biquad_bank_create(mi_biquad_bank_t **bank, uint32_t channels, uint32_t max_sections);
/* FilterBank::destroy(), FilterBank.cpp:51-60. */
int mi_biquad_bank_destroy(mi_biquad_bank_t *bank);
/*
 * FilterBank::begin() + add_chain() x count + end(clear) for one channel
 * (FilterBank.h:78-82, FilterBank.cpp:94-99,106-236).  As in the reference the
 * delay memory of the channel is cleared when `clear` is non-zero or when the
 * number of sections changed (FilterBank.cpp:233-235).  count > max_sections
 * is clamped the way add_chain() clamps (the last slot keeps the last chain).
 * Host-side only; the device tables are refreshed by the next process()/commit().
 */
int mi_biquad_bank_set_chains(mi_biquad_bank_t *bank, uint32_t channel,
                              const mi_biquad_x1_t *chains, uint32_t count, int clear);
/* Same for all channels at once: chains is [channels][count]. */
int mi_biquad_bank_set_all_chains(mi_biquad_bank_t *bank, const mi_biquad_x1_t *chains,
                                  uint32_t count, int clear);
/* FilterBank::size() of one channel. */
int mi_biquad_bank_size(const mi_biquad_bank_t *bank, uint32_t channel, uint32_t *count);
/* A channel that is switched off is skipped by process(): its delay memory stays as it is and its output row is not
 * written -- what a caller gets by not calling FilterBank::process() for that object (the meters do this for disabled
 * channels, LoudnessMeter.cpp:420-422, ILUFSMeter.cpp:370). */
int mi_biquad_bank_set_row_enabled(mi_biquad_bank_t *bank, uint32_t channel, int enabled);
/*
 * The bank's arithmetic.  0 (default): the time-parallel kernels -- the reference's recurrence inside chunks of 16 samples,
 * the chunks joined by a scan: the same filter in another order of roundings, within the recursion's own float32 noise of the
 * reference's output (DESIGN.md section 4 states the bound), 550 000 Msamples/s at 1024 channels x 8 sections.
 * 1: FilterBank::process's serial recurrence (/root/reference/src/main/filters/FilterBank.cpp:256-291; lsp-dsp-lib's
 * biquad_process_x1 form, every product and sum rounded on its own) sample after sample, a section per lane: the reference's
 * output and filter memory BIT FOR BIT, at the price of the recursion's latency chain (32 000 Msamples/s at that size).
 * The filter memory is the same in both modes: calls may alternate.  process(), process_blocks() and impulse_response()
 * follow the mode, and chains of banks (Crossover, the Equalizer's IIR mode) then run bank by bank instead of through their
 * fused launch; the meters' fused sums (LoudnessMeter / ILUFSMeter) keep the fast kernels.
 * mi_dspu_set_exact_iir_default (or MI_DSPU_EXACT_IIR=1 in the environment): the mode banks created from now on start in (process-wide; how the class layer's
 * dspu::Filter / FilterBank / Equalizer objects are put into the exact mode: set it before constructing them).
 */
int mi_biquad_bank_set_exact(mi_biquad_bank_t *bank, int on);
int mi_dspu_set_exact_iir_default(int on);
/*
 * Measurement aid: the shader clock (GHz) the last run-of-blocks launch of the biquad bank (mi_biquad_bank_process_blocks)
 * held, from the cycle counter and the 100 MHz wall clock its first workgroup stamps at entry and exit, and that workgroup's
 * life in microseconds.  The part lowers its clock under dense packed arithmetic: an issue-rate roof priced at the nominal
 * 2.4 GHz is not the one the launch met (bench.py prices valu_issue_frac at this clock).  Synchronises with the device.
 */
int mi_dspu_last_stream_clock(double *ghz, double *microseconds);
/* Push pending coefficient tables / state clears to the device (async on stream). */
int mi_biquad_bank_commit(mi_biquad_bank_t *bank, void *stream);
/* FilterBank::reset(), FilterBank.cpp:238-254; channel = UINT32_MAX for all. */
int mi_biquad_bank_reset(mi_biquad_bank_t *bank, uint32_t channel, void *stream);
/*
 * FilterBank::process(out, in, samples) for every channel, FilterBank.cpp:256-291.
 * out/in: device float32 [channels][*_stride]; out may be the same buffer as in
 * (exact alias only, as in the reference).  A channel with 0 sections copies.
 */
int mi_biquad_bank_process(mi_biquad_bank_t *bank, float *out, const float *in,
                           size_t samples, size_t out_stride, size_t in_stride, void *stream);
/*
 * `blocks` consecutive FilterBank::process() calls in one C call: block k reads in[k] and writes out[k] (HOST arrays of
 * `blocks` DEVICE pointers, each [channels][*_stride]); the filter memory is carried from block to block exactly as by
 * `blocks` separate calls -- it IS `blocks` launches, issued back to back without returning to the caller in between
 * (a host language whose call overhead is of the order of the 12 us a 1024 x 4096 block takes keeps the GPU busy this way).
 */
int mi_biquad_bank_process_blocks(mi_biquad_bank_t *bank, float *const *out, const float *const *in, size_t blocks,
                                  size_t samples, size_t out_stride, size_t in_stride, void *stream);
/*
 * FilterBank::impulse_response(out, samples), FilterBank.cpp:293-330: delay
 * memory is saved, zeroed, a unit impulse is run and the memory restored.
 */
int mi_biquad_bank_impulse_response(mi_biquad_bank_t *bank, float *out, size_t samples,
                                    size_t out_stride, void *stream);
/* Delay memory access (biquad_t::d[], FilterBank.cpp:248-252): host arrays of
 * [channels][max_sections][2] floats {d0,d1}.  Synchronise the stream. */
int mi_biquad_bank_get_state(mi_biquad_bank_t *bank, float *host_state, void *stream);
int mi_biquad_bank_set_state(mi_biquad_bank_t *bank, const float *host_state, void *stream);

/* ---- filter designer (host side) --------------------------------------------------------- */
/*
 * filter_params_t, field for field (include/lsp-plug.in/dsp-units/filters/common.h:137-145).
 */
typedef struct mi_filter_params
{
    uint32_t    nType;      /* enum mi_filter_type */
    uint32_t    nSlope;
    float       fFreq;
    float       fFreq2;
    float       fGain;
    float       fQuality;
} mi_filter_params_t;

/* filter_type_t with the same enumerator values (filters/common.h:38-135). */
enum mi_filter_type
{
    MI_FLT_NONE = 0,
    MI_FLT_BT_AMPLIFIER = 1,
    MI_FLT_MT_AMPLIFIER = 2,
    MI_FLT_BT_RLC_LOPASS = 3,
    MI_FLT_MT_RLC_LOPASS = 4,
    MI_FLT_BT_RLC_HIPASS = 5,
    MI_FLT_MT_RLC_HIPASS = 6,
    MI_FLT_BT_RLC_LOSHELF = 7,
    MI_FLT_MT_RLC_LOSHELF = 8,
    MI_FLT_BT_RLC_HISHELF = 9,
    MI_FLT_MT_RLC_HISHELF = 10,
    MI_FLT_BT_RLC_BELL = 11,
    MI_FLT_MT_RLC_BELL = 12,
    MI_FLT_BT_RLC_RESONANCE = 13,
    MI_FLT_MT_RLC_RESONANCE = 14,
    MI_FLT_BT_RLC_NOTCH = 15,
    MI_FLT_MT_RLC_NOTCH = 16,
    MI_FLT_BT_RLC_ALLPASS = 17,
    MI_FLT_MT_RLC_ALLPASS = 18,
    MI_FLT_BT_RLC_ALLPASS2 = 19,
    MI_FLT_MT_RLC_ALLPASS2 = 20,
    MI_FLT_BT_RLC_LADDERPASS = 21,
    MI_FLT_MT_RLC_LADDERPASS = 22,
    MI_FLT_BT_RLC_LADDERREJ = 23,
    MI_FLT_MT_RLC_LADDERREJ = 24,
    MI_FLT_BT_RLC_BANDPASS = 25,
    MI_FLT_MT_RLC_BANDPASS = 26,
    MI_FLT_BT_RLC_ENVELOPE = 27,
    MI_FLT_MT_RLC_ENVELOPE = 28,
    MI_FLT_BT_BWC_LOPASS = 29,
    MI_FLT_MT_BWC_LOPASS = 30,
    MI_FLT_BT_BWC_HIPASS = 31,
    MI_FLT_MT_BWC_HIPASS = 32,
    MI_FLT_BT_BWC_LOSHELF = 33,
    MI_FLT_MT_BWC_LOSHELF = 34,
    MI_FLT_BT_BWC_HISHELF = 35,
    MI_FLT_MT_BWC_HISHELF = 36,
    MI_FLT_BT_BWC_BELL = 37,
    MI_FLT_MT_BWC_BELL = 38,
    MI_FLT_BT_BWC_LADDERPASS = 39,
    MI_FLT_MT_BWC_LADDERPASS = 40,
    MI_FLT_BT_BWC_LADDERREJ = 41,
    MI_FLT_MT_BWC_LADDERREJ = 42,
    MI_FLT_BT_BWC_BANDPASS = 43,
    MI_FLT_MT_BWC_BANDPASS = 44,
    MI_FLT_BT_BWC_ALLPASS = 45,
    MI_FLT_MT_BWC_ALLPASS = 46,
    MI_FLT_BT_LRX_LOPASS = 47,
    MI_FLT_MT_LRX_LOPASS = 48,
    MI_FLT_BT_LRX_HIPASS = 49,
    MI_FLT_MT_LRX_HIPASS = 50,
    MI_FLT_BT_LRX_LOSHELF = 51,
    MI_FLT_MT_LRX_LOSHELF = 52,
    MI_FLT_BT_LRX_HISHELF = 53,
    MI_FLT_MT_LRX_HISHELF = 54,
    MI_FLT_BT_LRX_BELL = 55,
    MI_FLT_MT_LRX_BELL = 56,
    MI_FLT_BT_LRX_LADDERPASS = 57,
    MI_FLT_MT_LRX_LADDERPASS = 58,
    MI_FLT_BT_LRX_LADDERREJ = 59,
    MI_FLT_MT_LRX_LADDERREJ = 60,
    MI_FLT_BT_LRX_BANDPASS = 61,
    MI_FLT_MT_LRX_BANDPASS = 62,
    MI_FLT_BT_LRX_ALLPASS = 63,
    MI_FLT_MT_LRX_ALLPASS = 64,
    MI_FLT_DR_APO_LOPASS = 65,
    MI_FLT_DR_APO_HIPASS = 66,
    MI_FLT_DR_APO_BANDPASS = 67,
    MI_FLT_DR_APO_NOTCH = 68,
    MI_FLT_DR_APO_ALLPASS = 69,
    MI_FLT_DR_APO_ALLPASS2 = 70,
    MI_FLT_DR_APO_PEAKING = 71,
    MI_FLT_DR_APO_LOSHELF = 72,
    MI_FLT_DR_APO_HISHELF = 73,
    MI_FLT_DR_APO_LADDERPASS = 74,
    MI_FLT_DR_APO_LADDERREJ = 75,
    MI_FLT_A_WEIGHTED = 76,
    MI_FLT_B_WEIGHTED = 77,
    MI_FLT_C_WEIGHTED = 78,
    MI_FLT_D_WEIGHTED = 79,
    MI_FLT_K_WEIGHTED = 80
};

/* dsp::f_cascade_t: numerator t[] and denominator b[] of one analog (or plot) cascade. */
typedef struct mi_filter_cascade
{
    float t[4];
    float b[4];
} mi_filter_cascade_t;

/* Filter::filter_mode_t (filters/Filter.h:41-47). */
enum { MI_FM_BYPASS = 0, MI_FM_BILINEAR = 1, MI_FM_MATCHED = 2, MI_FM_APO = 3 };

/*
 * Filter::update() + Filter::rebuild() without a bank (src/main/filters/Filter.cpp:141-167,208-403):
 * limits the parameters, builds the prototype cascades and transforms them into digital sections
 * exactly as they would be handed to FilterBank::add_chain().  Host-only, no GPU needed.
 * chains/cascades may be NULL to query the counts; at most max_* entries are written.
 */
int mi_filter_design(const mi_filter_params_t *params, uint32_t sample_rate,
                     mi_biquad_x1_t *chains, uint32_t max_chains, uint32_t *n_chains,
                     mi_filter_cascade_t *cascades, uint32_t max_cascades, uint32_t *n_cascades, int *mode);
/* Filter::limit(), Filter.cpp:161-167. */
int mi_filter_limit(mi_filter_params_t *params, uint32_t sample_rate);
/*
 * Filter::freq_chart(c, f, count), packed-complex form (Filter.cpp:602-696): c[2i], c[2i+1] = H(f[i]).
 */
int mi_filter_freq_chart(const mi_filter_params_t *params, uint32_t sample_rate, float *c, const float *f, size_t count);

/* ---- partitioned FFT convolver bank ------------------------------------------------------ */
/*
 * mi_convolver_bank: `channels` independent lsp::dspu::Convolver objects
 * (include/lsp-plug.in/dsp-units/util/Convolver.h:35-114): exact linear
 * convolution of each channel with its own impulse response, zero added
 * latency, any call size.
 */
typedef struct mi_convolver_bank mi_convolver_bank_t;

/*
 * Convolver::init(data, count, rank, phase) for every channel
 * (src/main/util/Convolver.cpp:77-215).  irs: HOST float32 [channels][ir_stride];
 * channel c uses counts[c] taps (counts == NULL: `count` for all).  rank is
 * clamped to [8,16] (Convolver.h:28-29, Convolver.cpp:87) and sets the frame
 * 2^(rank-1) at which whole-frame calls take the FFT path; frames above 4096
 * are processed as 4096-sample partitions.  phase only staggers the
 * reference's CPU load (Convolver.cpp:139) and has no effect on the output; it
 * is accepted and ignored.  All counts == 0 leaves the bank uninitialised:
 * process() then writes zeros (Convolver.cpp:80-84,219-223).  Synchronises `stream`.
 */
int mi_convolver_bank_create(mi_convolver_bank_t **bank, uint32_t channels, const float *irs, size_t ir_stride,
                             const uint32_t *counts, uint32_t count, uint32_t rank, float phase, void *stream);
/* Replace the impulse response of the named channels (`channels`: HOST flags, NULL = every channel) by `count` taps read
 * from DEVICE memory [channels][ir_stride] (count within the capacity the bank was created with); input history is kept.
 * Used by the equalizer bank when it retunes (the reference re-parses its FIR the same way, Equalizer.cpp:342-345).
 * Single-partition banks (taps <= frame) take the new response at the next frame boundary: the frame being received keeps
 * the one it began with.  Synchronises `stream`. */
int mi_convolver_bank_set_irs_device(mi_convolver_bank_t *bank, const float *irs, size_t ir_stride, uint32_t count,
                                     const uint8_t *channels, void *stream);
/*
 * Single-partition banks: the new responses of the named channels wait for the next frame boundary and are cross-faded in
 * over that frame -- weight of the new response 0 up to B/2, a linear ramp over the next B output positions of the frame's
 * 2B-long result, 1 after that (what Equalizer::process does for a "smooth" retune, Equalizer.cpp:486-501).  Every
 * channel behaves like one reference Equalizer object with its vConv, vNewConv and EF_XFADE: a later call for a channel
 * that is still waiting replaces its target; a channel that is not named takes no part (the reference's cross-fade is not a
 * no-op for an unchanged response -- it also scales the overlap tail of the previous blocks, Equalizer.cpp:496,499);
 * mi_convolver_bank_set_irs_device() for a waiting channel changes the response the fade starts from, not its target
 * (after the fade the target, i.e. the older request, is in force -- Equalizer.cpp:491).
 */
int mi_convolver_bank_crossfade_irs_device(mi_convolver_bank_t *bank, const float *irs, size_t ir_stride, uint32_t count,
                                           const uint8_t *channels, void *stream);
/* Convolver::destroy(), Convolver.cpp:71-75.  MI_EHIP (the bank is freed all the same) if a hand-over had timed out, see
 * mi_convolver_bank_faults. */
int mi_convolver_bank_destroy(mi_convolver_bank_t *bank);
/* Diagnostic (synchronises the stream): how often the two roles of the one-launch frame step gave up waiting for each other
 * (about a second each; 0 on a healthy device -- DESIGN.md 3.2).  A fault is not silent: it raises a host-visible flag, and
 * from then on mi_convolver_bank_process (without synchronising), this call and mi_convolver_bank_destroy return MI_EHIP
 * until mi_convolver_bank_reset -- the frame in which it happened and everything after it is invalid.  The one-launch step
 * is used on gfx950 only; MI_CONV_TWO_LAUNCH=1 in the environment selects the two-launch form (no in-launch waiting). */
int mi_convolver_bank_faults(mi_convolver_bank_t *bank, uint32_t *count, void *stream);
/* Forget all input history (state right after init); clears a recorded fault. */
int mi_convolver_bank_reset(mi_convolver_bank_t *bank, void *stream);
/* Convolver::rank() / data_size() (Convolver.h:100-106) plus the partition geometry in use. */
int mi_convolver_bank_info(const mi_convolver_bank_t *bank, uint32_t *rank, uint32_t *frame,
                           uint32_t *partitions, uint32_t *data_size);
/*
 * Convolver::process(dst, src, count) for every channel, Convolver.cpp:217-313.
 * out/in: device float32 [channels][*_stride]; out may be the same buffer as in.
 */
int mi_convolver_bank_process(mi_convolver_bank_t *bank, float *out, const float *in, size_t samples,
                              size_t out_stride, size_t in_stride, void *stream);
/*
 * `blocks` consecutive mi_convolver_bank_process calls in one C call (Convolver.cpp:217-313 per block): block k reads in[k]
 * and writes out[k] (HOST tables of DEVICE pointers).  Whole frames of a partitioned bank go in batches of up to 16 frames
 * whose tails come out of ONE pass over the partitions' images; samples and state are those of the calls one by one.
 */
int mi_convolver_bank_process_blocks(mi_convolver_bank_t *bank, float *const *out, const float *const *in, size_t blocks,
                                     size_t samples, size_t out_stride, size_t in_stride, void *stream);

/* ---- windows and spectral envelopes (host side) ---------------------------------------- */
/* windows::window_t with the same enumerator values (include/lsp-plug.in/dsp-units/misc/windows.h:34-62). */
enum mi_window
{
    MI_WINDOW_HANN, MI_WINDOW_HAMMING, MI_WINDOW_BLACKMAN, MI_WINDOW_LANCZOS, MI_WINDOW_GAUSSIAN,
    MI_WINDOW_POISSON, MI_WINDOW_PARZEN, MI_WINDOW_TUKEY, MI_WINDOW_WELCH, MI_WINDOW_NUTTALL,
    MI_WINDOW_BLACKMAN_NUTTALL, MI_WINDOW_BLACKMAN_HARRIS, MI_WINDOW_HANN_POISSON, MI_WINDOW_BARTLETT_HANN,
    MI_WINDOW_BARTLETT_FEJER, MI_WINDOW_TRIANGULAR, MI_WINDOW_RECTANGULAR, MI_WINDOW_FLAT_TOP,
    MI_WINDOW_COSINE, MI_WINDOW_SQR_COSINE, MI_WINDOW_CUBIC, MI_WINDOW_TOTAL
};
/* envelope::envelope_t (include/lsp-plug.in/dsp-units/misc/envelope.h:36-50). */
enum mi_envelope
{
    MI_ENVELOPE_VIOLET_NOISE, MI_ENVELOPE_BLUE_NOISE, MI_ENVELOPE_WHITE_NOISE, MI_ENVELOPE_PINK_NOISE,
    MI_ENVELOPE_BROWN_NOISE, MI_ENVELOPE_MINUS_4_5_DB, MI_ENVELOPE_PLUS_4_5_DB, MI_ENVELOPE_TOTAL
};
/* windows::window(dst, n, type), src/main/misc/windows.cpp:62-98 (host memory). */
int mi_window(float *dst, size_t n, int type);
/*
 * The parameterised families, windows::*_general of misc/windows.h:71,86,95,101,113,128,134,143,146,155:
 *   TRIANGULAR {dn}   HAMMING {a, b}   BLACKMAN {a}   NUTTALL {a0..a3}   FLAT_TOP {a0..a4}   GAUSSIAN {s}
 *   POISSON {t}   BARTLETT_HANN {a0, a1, a2}   HANN_POISSON {a}   TUKEY {a}
 * (HANN, BLACKMAN_NUTTALL, BLACKMAN_HARRIS and BARTLETT_FEJER are members of those families and take the same lists).
 */
int mi_window_general(float *dst, size_t n, int type, const float *params, uint32_t count);
/* envelope::noise_lin / reverse_noise_lin(dst, first, last, center, n, type), src/main/misc/envelope.cpp:63-123 (host
 * memory): the colour's spectral envelope (f / center)^k on n linearly spaced frequencies, or the opposite colour's. */
int mi_envelope_noise_lin(float *dst, float first, float last, float center, size_t n, int type);
int mi_envelope_reverse_noise_lin(float *dst, float first, float last, float center, size_t n, int type);
/* envelope::noise_log / reverse_noise_log (envelope.cpp:170-268: logarithmic frequency grid) and noise_list /
 * reverse_noise_list (:272-344: an explicit list of frequencies); reverse != 0 selects the compensating colour. */
int mi_envelope_noise_log(float *dst, float first, float last, float center, size_t n, int type, int reverse);
int mi_envelope_noise_list(float *dst, const float *freqs, float center, size_t n, int type, int reverse);

/* ---- spectral processor bank -------------------------------------------------------------- */
/*
 * mi_spectral_bank: `channels` lsp::dspu::SpectralProcessor objects sharing rank and phase, equivalently one
 * lsp::dspu::MultiSpectralProcessor with `channels` channels (util/SpectralProcessor.h:45-175,
 * util/MultiSpectralProcessor.h:47-247): 50 %-overlap sine-window STFT -> operation -> ISTFT overlap-add,
 * latency 2^rank samples.
 */
typedef struct mi_spectral_bank mi_spectral_bank_t;
enum { MI_SPECTRAL_OP_NONE = 0, MI_SPECTRAL_OP_MASK = 1, MI_SPECTRAL_OP_CALLBACK = 2 };
/*
 * spectral_processor_func_t / multi_spectral_processor_func_t on the device
 * (util/SpectralProcessor.h:39, util/MultiSpectralProcessor.h:41): `spectrum` is a DEVICE pointer to
 * [channels][2 * 2^rank] floats (packed complex, all 2^rank bins); the function runs on the host and may enqueue
 * work on `stream` (including collectives); the inverse transform is ordered after it on the same stream.
 */
typedef void (*mi_spectral_func_t)(void *object, void *subject, float *spectrum, size_t rank, size_t channels, void *stream);

/* SpectralProcessor::init(max_rank), SpectralProcessor.cpp:59-75 (ranks 5..18: frames up to 2^14 samples are transformed
 * in LDS in one launch per hop, longer ones through global memory in several). */
int mi_spectral_bank_create(mi_spectral_bank_t **bank, uint32_t channels, uint32_t max_rank);
int mi_spectral_bank_destroy(mi_spectral_bank_t *bank);
/* set_rank / set_phase, SpectralProcessor.cpp:127-145 (a rank above max_rank is ignored, phase is clamped to 0..1). */
int mi_spectral_bank_set_rank(mi_spectral_bank_t *bank, uint32_t rank);
int mi_spectral_bank_set_phase(mi_spectral_bank_t *bank, float phase);
/* Analysis / synthesis windows (enum mi_window, or -1 for none).  The reference processor always uses the sine
 * window on both sides (SpectralProcessor.cpp:118); the Equalizer's SPM mode uses none / squared cosine
 * (Equalizer.cpp:352-353,535-540) and is built on this. */
int mi_spectral_bank_set_windows(mi_spectral_bank_t *bank, int in_window, int out_window);
/* get_rank(), latency(), remaining() (SpectralProcessor.h:128,142, SpectralProcessor.cpp:251-255). */
int mi_spectral_bank_get(const mi_spectral_bank_t *bank, uint32_t *rank, uint32_t *latency, uint32_t *remaining);
/* bind(func, object, subject) / unbind(), SpectralProcessor.cpp:91-105. */
int mi_spectral_bank_bind(mi_spectral_bank_t *bank, mi_spectral_func_t func, void *object, void *subject);
int mi_spectral_bank_unbind(mi_spectral_bank_t *bank);
/*
 * Built-in callback: spectrum[k] *= mask[k], k = 0..2^(rank-1) (real gains, applied to both halves of the
 * spectrum), fused into the transform kernel.  mask is HOST memory, one row shared by all channels
 * (mask_stride == 0) or [channels][mask_stride].
 */
int mi_spectral_bank_bind_mask(mi_spectral_bank_t *bank, const float *mask, size_t mask_stride, void *stream);
/* MultiSpectralProcessor::bind_in/bind_out (MultiSpectralProcessor.cpp:160-227): which channels have an input /
 * an output bound (HOST byte arrays, NULL = unchanged; default all bound). Callback mode only. */
int mi_spectral_bank_bind_channels(mi_spectral_bank_t *bank, const uint8_t *has_in, const uint8_t *has_out, void *stream);
/* reset(), SpectralProcessor.cpp:257-266. */
int mi_spectral_bank_reset(mi_spectral_bank_t *bank, void *stream);
/* process(dst, src, count), SpectralProcessor.cpp:147-199.  With out == NULL: process(src, count), :201-249 -- analysis
 * only: the function is called on every frame, nothing is transformed back and the output buffer is shifted and its
 * tail zeroed instead of overlap-added (MultiSpectralProcessor timing: out == NULL just skips the copy-out).
 * Throughput note: a call that hands over whole frames from rows whose pointers are 8-byte aligned and whose strides are
 * even is one launch per hop for the operations NONE and MASK at ranks 8..13 (two around the function for CALLBACK with
 * MultiSpectralProcessor's timing); anything else adds two strided copies per piece. */
int mi_spectral_bank_process(mi_spectral_bank_t *bank, float *out, const float *in, size_t count,
                             size_t out_stride, size_t in_stride, void *stream);
/*
 * `blocks` consecutive mi_spectral_bank_process calls in one C call (SpectralProcessor.cpp:143-199 per block): block k reads
 * in[k] and writes out[k] (HOST tables of DEVICE pointers).  Runs of blocks of whole frames whose buffers lie apart go out as
 * one launch; the samples and the state are those of the calls one by one.
 */
int mi_spectral_bank_process_blocks(mi_spectral_bank_t *bank, float *const *out, const float *const *in, size_t blocks,
                                    size_t count, size_t out_stride, size_t in_stride, void *stream);
/* When a complete frame is transformed: eager == 0 (default) when the next sample arrives, as SpectralProcessor does
 * (SpectralProcessor.cpp:159: remaining() can read 0, the function runs at the start of the following call); eager != 0
 * as soon as the frame is complete, as MultiSpectralProcessor does (MultiSpectralProcessor.cpp:324). */
int mi_spectral_bank_set_timing(mi_spectral_bank_t *bank, int eager);

/* ---- spectrum analyzer bank ---------------------------------------------------------------- */
/*
 * mi_analyzer_bank: one lsp::dspu::Analyzer with `channels` channels (util/Analyzer.h:53-361): ring-buffer
 * ingest, windowed FFT magnitude per channel once per refresh period, exponential smoothing, envelope.
 */
typedef struct mi_analyzer_bank mi_analyzer_bank_t;
enum { MI_ANALYZER_SAMPLE_RATE, MI_ANALYZER_RATE, MI_ANALYZER_WINDOW, MI_ANALYZER_ENVELOPE, MI_ANALYZER_SHIFT,
       MI_ANALYZER_REACTIVITY, MI_ANALYZER_RANK, MI_ANALYZER_ACTIVE };
enum { MI_ANALYZER_CH_FREEZE, MI_ANALYZER_CH_ENABLE, MI_ANALYZER_CH_DELAY };

/* Analyzer::init(channels, max_rank, max_sr, min_rate, max_delay), Analyzer.cpp:83-152 (ranks 5..18; above 14 through global memory). */
int mi_analyzer_bank_create(mi_analyzer_bank_t **bank, uint32_t channels, uint32_t max_rank, uint32_t max_sample_rate,
                            float min_rate, uint32_t max_delay);
int mi_analyzer_bank_destroy(mi_analyzer_bank_t *bank);
/* set_sample_rate/set_rate/set_window/set_envelope/set_shift/set_reactivity/set_rank/set_activity, Analyzer.cpp:154-211. */
/* Analyzer::reset(): the smoothed and the published spectra are cleared at the next reconfigure (Analyzer.cpp:274-281) */
int mi_analyzer_bank_reset(mi_analyzer_bank_t *bank);
int mi_analyzer_bank_configure(mi_analyzer_bank_t *bank, int what, double value);
/* freeze_channel/enable_channel/set_channel_delay, Analyzer.cpp:213-249. */
int mi_analyzer_bank_channel(mi_analyzer_bank_t *bank, uint32_t channel, int what, uint32_t value);
/*
 * Analyzer::process(in, samples), Analyzer.cpp:299-409.  in: device [channels][in_stride] or NULL (zeros).
 * Unlike the reference (which divides by zero, Analyzer.cpp:258-260,315) a refresh rate above
 * sample_rate / channels is rejected with MI_EINVAL.
 */
int mi_analyzer_bank_process(mi_analyzer_bank_t *bank, const float *in, size_t samples, size_t in_stride, void *stream);
/* Analyzer::get_spectrum for every channel at once, Analyzer.cpp:443-456: out[c][i] = vData[c][idx[i]] * envelope[idx[i]]
 * (out and idx are DEVICE pointers). */
int mi_analyzer_bank_get_spectrum(mi_analyzer_bank_t *bank, float *out, size_t out_stride, const uint32_t *idx,
                                  uint32_t count, void *stream);
/* Per-bin sum over this bank's channels of the smoothed magnitudes (2^(rank-1)+1 floats, DEVICE): the local half
 * of the cross-channel per-bin reduction; the caller all-reduces it across GPUs. */
int mi_analyzer_bank_reduce_bins(mi_analyzer_bank_t *bank, float *out, int with_envelope, void *stream);
/*
 * mi_analyzer_bank_process followed by mi_analyzer_bank_reduce_bins in one call: the two launches.  (The reduction as a second
 * role of the analysis launch -- round 3, 20.3 against 17.2 us per step -- was removed in round 6.)
 */
int mi_analyzer_bank_process_reduce(mi_analyzer_bank_t *bank, const float *in, size_t samples, size_t in_stride,
                                    float *out, int with_envelope, void *stream);
/*
 * `frames` consecutive mi_analyzer_bank_process_reduce calls in one C call: frame k takes in[k] (HOST array of DEVICE
 * pointers, each [channels][in_stride]) and leaves its per-bin sums in out + k * out_stride.  Where every call is exactly
 * one strobe (samples == period) the spectra of up to 16 frames are kept and their reductions run as ONE launch; the
 * sums and the state left behind are those of `frames` separate calls, bit for bit.
 */
int mi_analyzer_bank_process_reduce_frames(mi_analyzer_bank_t *bank, const float *const *in, size_t frames, size_t samples,
                                           size_t in_stride, float *out, size_t out_stride, int with_envelope, void *stream);
int mi_analyzer_bank_info(const mi_analyzer_bank_t *bank, uint32_t *rank, uint32_t *bins, uint32_t *period, uint32_t *step);

/*
 * Multi-GPU: the one exchange step of the path.  Channels are sharded over the GPUs of a node, one process per GPU;
 * the per-bin sum over ALL channels (the MultiSpectralProcessor-style callback stage, util/MultiSpectralProcessor.h:41,
 * BASELINE config 5) is each rank's mi_analyzer_bank_reduce_bins followed by one RCCL all-reduce over xGMI, issued
 * from the library's C++ host side on the caller's stream.  A communicator is created like an ncclComm_t: one rank
 * makes the 128-byte id and hands it to the others by whatever means the host has (a file, MPI, torch.distributed).
 * mi_dspu_comm_adopt wraps an ncclComm_t the host already owns (not destroyed with the wrapper).
 * reduce_bins adds the channels in a shard-composable order (blocks of 16 channels, then a binary tree aligned to
 * powers of two), so the halves of a channel set reduce to partial sums whose sum is the reduction of the whole set.
 */
#define MI_DSPU_COMM_ID_BYTES 128
typedef struct mi_dspu_comm mi_dspu_comm_t;
int mi_dspu_comm_unique_id(void *id128);
int mi_dspu_comm_create(mi_dspu_comm_t **comm, const void *id128, int nranks, int rank);
int mi_dspu_comm_adopt(mi_dspu_comm_t **comm, void *nccl_comm);
int mi_dspu_comm_destroy(mi_dspu_comm_t *comm);
int mi_dspu_comm_info(const mi_dspu_comm_t *comm, int *nranks, int *rank);
/* bins: DEVICE [frames][2^(rank-1)+1], summed in place over the ranks of `comm` (float32, ncclSum). */
int mi_analyzer_bank_allreduce_bins(mi_analyzer_bank_t *bank, float *bins, size_t frames, mi_dspu_comm_t *comm, void *stream);
/*
 * The same exchange step beside the caller's stream, for callers that double-buffer the sums: begin() issues the collective on
 * the communicator's own (least urgent: it takes the CUs the next batch's launches leave) side stream behind everything `stream` has enqueued so far -- `partial` summed over the
 * ranks into `total` (may be the same buffer) -- and returns; `stream` goes on with the next batch of frames, whose reduction
 * writes ANOTHER buffer.  mi_dspu_comm_wait() makes `stream` wait, on the device, for the slot's collective: call it in front of
 * whatever reads `total` and in front of the work that overwrites the slot's buffers again.  Slots: 0 .. MI_DSPU_COMM_SLOTS - 1.
 * The result is the serial call's bit for bit (the same ncclAllReduce on the same data).  Not capturable into a hipGraph.
 * Reference for the semantic: the cross-channel callback of MultiSpectralProcessor
 * (/root/reference/include/lsp-plug.in/dsp-units/util/MultiSpectralProcessor.h:41).
 */
#define MI_DSPU_COMM_SLOTS 4
int mi_analyzer_bank_allreduce_bins_begin(mi_analyzer_bank_t *bank, const float *partial, float *total, size_t frames,
                                          mi_dspu_comm_t *comm, int slot, void *stream);
int mi_dspu_comm_wait(mi_dspu_comm_t *comm, int slot, void *stream);

/* ---- dynamic filter bank --------------------------------------------------- */
/*
 * mi_dynfilter_bank: `channels` x lsp::dspu::DynamicFilters(filters) that share their filter settings
 * (filters/DynamicFilters.h:41-189, src/main/filters/DynamicFilters.cpp): filters whose gain follows a per-sample
 * gain vector.  set_params / set_filter_active / process / freq_chart have the reference's meaning (:127-181, .h:147-153,
 * :204-318, :1774-1970); process() takes DEVICE rows [channels][stride] for the samples AND the gains (every channel its
 * own gain curve) and is a copy for an inactive / FLT_NONE / slope-0 filter (:207-212).  A change of a filter's type clears
 * the memory of every filter at the next process() (:132-133,214-219).  FLT_*_RLC_ENVELOPE has no dynamic form here (the
 * reference's builder reads unset fields, :1022-1085) and is refused with MI_EINVAL, as are the static-only types.
 */
typedef struct mi_dynfilter_bank mi_dynfilter_bank_t;
int mi_dynfilter_bank_create(mi_dynfilter_bank_t **bank, uint32_t channels, uint32_t filters);
int mi_dynfilter_bank_destroy(mi_dynfilter_bank_t *bank);
int mi_dynfilter_bank_set_sample_rate(mi_dynfilter_bank_t *bank, uint32_t sample_rate);
int mi_dynfilter_bank_set_params(mi_dynfilter_bank_t *bank, uint32_t id, const mi_filter_params_t *params);
/* the parameters as set_params left them (fFreq2 transformed, DynamicFilters.cpp:170-178) */
int mi_dynfilter_bank_get_params(const mi_dynfilter_bank_t *bank, uint32_t id, mi_filter_params_t *params, int *active);
int mi_dynfilter_bank_set_filter_active(mi_dynfilter_bank_t *bank, uint32_t id, int active);
int mi_dynfilter_bank_process(mi_dynfilter_bank_t *bank, uint32_t id, float *out, const float *in, const float *gain,
                              size_t samples, size_t out_stride, size_t in_stride, size_t gain_stride, void *stream);
/*
 * Host-only (no device needed, like mi_filter_design): `params` are what set_params() receives.
 * sections: the digital sections the filter has at a fixed gain -- what every sample of a constant gain vector is
 * filtered with.  freq_chart: DynamicFilters::freq_chart(id, c, f, gain, count), packed complex.
 */
int mi_dynfilter_sections(const mi_filter_params_t *params, uint32_t sample_rate, float gain, mi_biquad_x1_t *sections,
                          uint32_t max_sections, uint32_t *count);
int mi_dynfilter_freq_chart(const mi_filter_params_t *params, uint32_t sample_rate, float *c, const float *f, float gain, size_t count);

/* ---- equalizer bank --------------------------------------------------------------------------- */
/*
 * mi_equalizer_bank: `channels` x lsp::dspu::Equalizer(filters, fir_rank)
 * (include/lsp-plug.in/dsp-units/filters/Equalizer.h:47-289).  equalizer_mode_t values as in Equalizer.h:35-42.
 */
typedef struct mi_equalizer_bank mi_equalizer_bank_t;
enum { MI_EQM_BYPASS = 0, MI_EQM_IIR = 1, MI_EQM_FIR = 2, MI_EQM_FFT = 3, MI_EQM_SPM = 4 };

/* Equalizer::init(filters, fir_rank), src/main/filters/Equalizer.cpp:67-160 (fir_rank 0 or 5..13). */
int mi_equalizer_bank_create(mi_equalizer_bank_t **bank, uint32_t channels, uint32_t filters, uint32_t fir_rank);
int mi_equalizer_bank_destroy(mi_equalizer_bank_t *bank);
/* set_params(id, params) of one channel (or UINT32_MAX: all), Equalizer.cpp:210-218; MI_EINVAL for a bad id
 * where the reference returns false. */
int mi_equalizer_bank_set_params(mi_equalizer_bank_t *bank, uint32_t channel, uint32_t filter, const mi_filter_params_t *params);
int mi_equalizer_bank_get_params(const mi_equalizer_bank_t *bank, uint32_t channel, uint32_t filter, mi_filter_params_t *params);
/* set_mode / set_sample_rate / set_actual_sample_rate, Equalizer.cpp:360-375,188-203. */
int mi_equalizer_bank_set_mode(mi_equalizer_bank_t *bank, int mode);
int mi_equalizer_bank_set_sample_rate(mi_equalizer_bank_t *bank, uint32_t sample_rate);
int mi_equalizer_bank_set_actual_sample_rate(mi_equalizer_bank_t *bank, uint32_t sample_rate);
/* get_latency() (reconfigures first), Equalizer.cpp:237-241. */
int mi_equalizer_bank_get_latency(mi_equalizer_bank_t *bank, uint32_t *latency, void *stream);
/* reset(), Equalizer.cpp:573-597. */
int mi_equalizer_bank_reset(mi_equalizer_bank_t *bank, void *stream);
/* smooth() / set_smooth(), Equalizer.cpp:618-626: in FIR and FFT modes a retune is cross-faded in over the block
 * that completes next (EF_XFADE, Equalizer.cpp:339-343,486-501) instead of switching at its start. */
int mi_equalizer_bank_set_smooth(mi_equalizer_bank_t *bank, int smooth);
/* process(out, in, samples), Equalizer.cpp:460-571. */
int mi_equalizer_bank_process(mi_equalizer_bank_t *bank, float *out, const float *in, size_t samples,
                              size_t out_stride, size_t in_stride, void *stream);
/*
 * `blocks` consecutive Equalizer::process() calls in one C call: block k reads in[k] and writes out[k] (HOST arrays of
 * DEVICE pointers, each [channels][*_stride]).  In the FIR and FFT modes runs of blocks of exactly 2^fir_rank samples go
 * as ONE launch (the response's image and the overlap-add tail stay on chip from block to block); results and carried
 * state are those of `blocks` separate calls, bit for bit (Equalizer.cpp:460-571 called block after block).
 */
int mi_equalizer_bank_process_blocks(mi_equalizer_bank_t *bank, float *const *out, const float *const *in, size_t blocks,
                                     size_t samples, size_t out_stride, size_t in_stride, void *stream);
/* filter count, fir_rank(), mode(), ir_size() (Equalizer.cpp:599-616). */
int mi_equalizer_bank_info(const mi_equalizer_bank_t *bank, uint32_t *filters, uint32_t *fir_rank, int *mode, uint32_t *ir_size);

/* ---- IIR crossover bank (first caller of the filter path) ----------------------------- */
/*
 * mi_crossover_bank: one lsp::dspu::Crossover per channel, all channels sharing the split settings
 * (util/Crossover.h:97-330, src/main/util/Crossover.cpp): `bands` bands, bands-1 split points; an active split point
 * low-passes into the band on its left (+ all-pass filters of the split points above it, so that the bands sum to an
 * all-pass) and high-passes into the chain on its right.  Linkwitz-Riley slopes: 1 = LR2 (12 dB/oct) ... 9 = LR32,
 * 0 = split point off (enum crossover_slope_t).  Band outputs go to caller-owned device buffers instead of the
 * reference's per-band callback.
 */
typedef struct mi_crossover_bank mi_crossover_bank_t;
enum { MI_CROSS_MODE_BT = 0, MI_CROSS_MODE_MT = 1 };                /* crossover_mode_t */

/* Crossover::init(bands, buf_size), Crossover.cpp:71-160 (default split frequencies log-spaced over 10 Hz..24 kHz). */
int mi_crossover_bank_create(mi_crossover_bank_t **bank, uint32_t channels, uint32_t bands);
int mi_crossover_bank_destroy(mi_crossover_bank_t *bank);
/* set_sample_rate / set_slope / set_frequency / set_mode / set_gain, Crossover.cpp:185-254,327-341 */
int mi_crossover_bank_set_sample_rate(mi_crossover_bank_t *bank, uint32_t sample_rate);
int mi_crossover_bank_set_slope(mi_crossover_bank_t *bank, uint32_t split, uint32_t slope);
int mi_crossover_bank_set_frequency(mi_crossover_bank_t *bank, uint32_t split, float freq);
int mi_crossover_bank_set_mode(mi_crossover_bank_t *bank, uint32_t split, int mode);
int mi_crossover_bank_set_gain(mi_crossover_bank_t *bank, uint32_t band, float gain);
/* get_slope / get_frequency / get_mode (any pointer may be NULL) */
int mi_crossover_bank_get_split(const mi_crossover_bank_t *bank, uint32_t split, uint32_t *slope, float *freq, int *mode);
/* Crossover::needs_reconfiguration(), util/Crossover.h:352 */
int mi_crossover_bank_needs_reconfiguration(const mi_crossover_bank_t *bank, int *pending);
/* get_gain / get_band_start / get_band_end / band_active after reconfigure(), Crossover.cpp:256-325 */
int mi_crossover_bank_get_band(mi_crossover_bank_t *bank, uint32_t band, float *gain, float *start, float *end, int *active,
                               void *stream);
/*
 * Crossover::process(in, samples), Crossover.cpp:451-498.  band_out: HOST array of `bands` DEVICE pointers
 * [channels][out_stride]; NULL = no handler bound to that band (its low-pass is skipped like in the reference).
 * Inactive bands are not written.  in: [channels][in_stride].
 */
int mi_crossover_bank_process(mi_crossover_bank_t *bank, float *const *band_out, const float *in, size_t samples,
                              size_t out_stride, size_t in_stride, void *stream);
/*
 * `blocks` consecutive process() calls in one C call (Crossover.cpp:451-498 per block): block i reads in[i] and writes band k
 * to band_out[i * bands + k] (HOST tables of DEVICE pointers; a band is NULL in every block or in none).  Runs of blocks
 * whose buffers do not overlap each other go out as ONE launch; the results are bit for bit those of the calls one by one.
 */
int mi_crossover_bank_process_blocks(mi_crossover_bank_t *bank, float *const *band_out, const float *const *in, size_t blocks,
                                     size_t samples, size_t out_stride, size_t in_stride, void *stream);
/* freq_chart(band, c, f, count), packed complex (re, im interleaved), HOST memory, Crossover.cpp:500-590 */
int mi_crossover_bank_freq_chart(mi_crossover_bank_t *bank, uint32_t band, float *c, const float *f, size_t count, void *stream);

/* ---- loudness meter bank (SURVEY 8f, K-weighted cascade fused with a mean-square reduction) ------------ */
/*
 * mi_loudness_bank: `meters` x lsp::dspu::LoudnessMeter(channels) sharing one configuration
 * (meters/LoudnessMeter.h:47-330, src/main/meters/LoudnessMeter.cpp): per channel a weighting filter (K by default,
 * ITU-R BS.1770), a sliding mean square over `period` ms, the channels mixed with their BS.2051 designation weights,
 * square root.  Rows of the sample buffers: row = meter * channels + channel.
 */
typedef struct mi_loudness_bank mi_loudness_bank_t;
enum { MI_BS_WEIGHT_NONE = 0, MI_BS_WEIGHT_A, MI_BS_WEIGHT_B, MI_BS_WEIGHT_C, MI_BS_WEIGHT_D, MI_BS_WEIGHT_K };   /* bs::weighting_t */
/* bs::channel_t (misc/broadcast.h:58-94), numeric values as in the reference: 0 NONE, 1 CENTER, 4 LEFT, 5 RIGHT,
 * 6..11 the +1.5 dB group (FRONT_LEFT .. RIGHT_SURROUND), 32/33 LFE1/LFE2 (excluded from the measurement). */
enum { MI_BS_CHANNEL_NONE = 0, MI_BS_CHANNEL_CENTER = 1, MI_BS_CHANNEL_LEFT = 4, MI_BS_CHANNEL_RIGHT = 5,
       MI_BS_CHANNEL_FRONT_LEFT = 6, MI_BS_CHANNEL_RIGHT_SURROUND = 11, MI_BS_CHANNEL_LFE1 = 32, MI_BS_CHANNEL_LFE2 = 33 };

/* LoudnessMeter::init(channels, max_period), LoudnessMeter.cpp:85-185 (mono: CENTER, stereo: LEFT/RIGHT by default). */
int mi_loudness_bank_create(mi_loudness_bank_t **bank, uint32_t meters, uint32_t channels, float max_period_ms);
int mi_loudness_bank_destroy(mi_loudness_bank_t *bank);
/* set_sample_rate (reallocates and clears the mean-square lines), set_period, set_weighting, LoudnessMeter.cpp:203-309 */
int mi_loudness_bank_set_sample_rate(mi_loudness_bank_t *bank, uint32_t sample_rate, void *stream);
int mi_loudness_bank_set_period(mi_loudness_bank_t *bank, float period_ms);
int mi_loudness_bank_set_weighting(mi_loudness_bank_t *bank, int weighting);
/* per channel index, for every meter: set_designation / set_link / set_active, LoudnessMeter.cpp:213-268 */
int mi_loudness_bank_set_designation(mi_loudness_bank_t *bank, uint32_t channel, int designation);
int mi_loudness_bank_set_link(mi_loudness_bank_t *bank, uint32_t channel, float link);
int mi_loudness_bank_set_active(mi_loudness_bank_t *bank, uint32_t channel, int active, void *stream);
/* clear(), LoudnessMeter.cpp:280-295; latency() in samples, :323-326 */
int mi_loudness_bank_clear(mi_loudness_bank_t *bank, void *stream);
/* bind(id, out, in) with / without an input: an enabled channel without an input is left out of the blocks (its filter does
 * not run, its squares line is not written, it is not mixed: LoudnessMeter.cpp:421-422) but keeps taking part in
 * refresh_rms() and clear(); binding it again continues where it stopped (unlike set_active(1), which clears). */
int mi_loudness_bank_set_bound(mi_loudness_bank_t *bank, uint32_t channel, int bound);
/* LoudnessMeter::needs_update() / update_settings(), meters/LoudnessMeter.h:264-269 */
int mi_loudness_bank_needs_update(const mi_loudness_bank_t *bank, int *pending);
int mi_loudness_bank_update_settings(mi_loudness_bank_t *bank, void *stream);
int mi_loudness_bank_latency(const mi_loudness_bank_t *bank, uint32_t *samples);
/*
 * process(out, count) / process(out, count, gain), LoudnessMeter.cpp:462-564.  in: [meters*channels][in_stride];
 * out: [meters][out_stride] or NULL; ch_out: [meters*channels][out_stride] or NULL (the reference's per-channel vOut with
 * linking: link 0 = the channel's own RMS, 1 = the mixed loudness).  The second form multiplies every output by gain and,
 * like the reference's, leaves loudness() alone (only the first form records fLoudness, LoudnessMeter.cpp:485).
 */
int mi_loudness_bank_process(mi_loudness_bank_t *bank, float *out, float *ch_out, const float *in, size_t count,
                             size_t out_stride, size_t in_stride, void *stream);
int mi_loudness_bank_process_gain(mi_loudness_bank_t *bank, float *out, float *ch_out, const float *in, size_t count,
                                  size_t out_stride, size_t in_stride, float gain, void *stream);
/* loudness(): the last value of the mixed loudness recorded by process() (HOST array of `meters` floats; synchronises) */
int mi_loudness_bank_loudness(mi_loudness_bank_t *bank, float *loudness, void *stream);

/*
 * mi_ilufs_bank: `meters` x lsp::dspu::ILUFSMeter(channels) sharing one configuration
 * (meters/ILUFSMeter.h:41-260, src/main/meters/ILUFSMeter.cpp): integrated loudness per ITU-R BS.1770-4 -- weighting
 * filter, gating blocks of `block_period` ms overlapping by 75 %, absolute (-70 LKFS) and relative (-10 LU) gating over
 * the integration period (0 = since the last clear()).  The output is the integrated loudness as a gain, held between
 * block boundaries.  Rows of the input: row = meter * channels + channel.
 */
typedef struct mi_ilufs_bank mi_ilufs_bank_t;
/* ILUFSMeter::init(channels, max_int_time [s], block_period [ms]), ILUFSMeter.cpp:113-211 */
int mi_ilufs_bank_create(mi_ilufs_bank_t **bank, uint32_t meters, uint32_t channels, float max_int_time, float block_period_ms);
int mi_ilufs_bank_destroy(mi_ilufs_bank_t *bank);
/* set_sample_rate / set_integration_period [s] / set_weighting, ILUFSMeter.cpp:258-322 */
int mi_ilufs_bank_set_sample_rate(mi_ilufs_bank_t *bank, uint32_t sample_rate, void *stream);
int mi_ilufs_bank_set_integration_period(mi_ilufs_bank_t *bank, float period, void *stream);
int mi_ilufs_bank_set_weighting(mi_ilufs_bank_t *bank, int weighting);
/* per channel index, for every meter: set_designation / set_active, ILUFSMeter.cpp:223-256 */
int mi_ilufs_bank_set_designation(mi_ilufs_bank_t *bank, uint32_t channel, int designation);
int mi_ilufs_bank_set_active(mi_ilufs_bank_t *bank, uint32_t channel, int active);
int mi_ilufs_bank_clear(mi_ilufs_bank_t *bank, void *stream);                      /* ILUFSMeter.cpp:547-560 */
/* process(out, count, gain), ILUFSMeter.cpp:355-470.  in: [meters*channels][in_stride]; out: [meters][out_stride] or NULL. */
int mi_ilufs_bank_process(mi_ilufs_bank_t *bank, float *out, const float *in, size_t count, size_t out_stride,
                          size_t in_stride, float gain, void *stream);
/* loudness() of every meter (HOST array of `meters` floats; synchronises) */
/* ILUFSMeter::needs_update() / update_settings(), meters/ILUFSMeter.h:244-249 */
int mi_ilufs_bank_needs_update(const mi_ilufs_bank_t *bank, int *pending);
int mi_ilufs_bank_update_settings(mi_ilufs_bank_t *bank, void *stream);
int mi_ilufs_bank_loudness(mi_ilufs_bank_t *bank, float *loudness, void *stream);
/* The gating-block history as the meters hold it (vLoudness / nMSHead / nMSCount of ILUFSMeter.h): hist is HOST memory
 * [meters][*size] (NULL: only the size is wanted), head / count HOST [meters] or NULL. */
int mi_ilufs_bank_history(mi_ilufs_bank_t *bank, float *hist, uint32_t *size, uint32_t *head, uint32_t *count, void *stream);

/*
 * mi_splitter_bank: lsp::dspu::SpectralSplitter for `channels` channels sharing the settings
 * (util/SpectralSplitter.h:62-250, src/main/util/SpectralSplitter.cpp:62-361) -- the engine of lsp::dspu::FFTCrossover.
 * The last 2^rank samples of every channel are transformed once per 2^(chunk_rank-1) samples; each handler shapes that
 * spectrum, returns to the time domain and overlap-adds the last 2^chunk_rank samples (sin^2 window) into its output.
 * A handler is bound as COPY (no spectral function), MASK (2^rank real gains in FFT order: FFTCrossover::spectral_func,
 * FFTCrossover.cpp:124-140) or CALLBACK; its sink is the device buffer passed to process().
 */
typedef struct mi_splitter_bank mi_splitter_bank_t;
/* spectral_splitter_func_t on the device (util/SpectralSplitter.h:41-46): in/out are DEVICE pointers to
 * [channels][2 * 2^rank] floats (packed complex); the function runs on the host and may enqueue work on `stream`. */
typedef void (*mi_splitter_func_t)(void *object, void *subject, float *out, const float *in, size_t rank, size_t channels, void *stream);
/* init(max_rank, handlers), SpectralSplitter.cpp:62-127 (max_rank 5..18; frames above 2^14 samples go through global memory,
 * one plain launch per step of the hop instead of the fused kernel) */
int mi_splitter_bank_create(mi_splitter_bank_t **bank, uint32_t channels, uint32_t max_rank, uint32_t handlers);
int mi_splitter_bank_destroy(mi_splitter_bank_t *bank);
/* set_rank / set_chunk_rank / set_phase, SpectralSplitter.cpp:260-282 */
int mi_splitter_bank_set_rank(mi_splitter_bank_t *bank, uint32_t rank);
int mi_splitter_bank_set_chunk_rank(mi_splitter_bank_t *bank, int32_t rank);
int mi_splitter_bank_set_phase(mi_splitter_bank_t *bank, float phase);
/* rank(), chunk_rank(), latency() (:284-293) and the samples left until the next transform */
int mi_splitter_bank_get(const mi_splitter_bank_t *bank, uint32_t *rank, uint32_t *chunk_rank, uint32_t *latency, uint32_t *remaining);
/* bind(id, object, subject, func, sink) in its three forms, unbind(id) (:143-180).  mask: HOST memory, one row of 2^rank
 * gains for all channels (mask_stride == 0) or [channels][mask_stride]; binding a mask to a handler that already has
 * one only replaces the gains. */
int mi_splitter_bank_bind_copy(mi_splitter_bank_t *bank, uint32_t handler, void *stream);
int mi_splitter_bank_bind_mask(mi_splitter_bank_t *bank, uint32_t handler, const float *mask, size_t mask_stride, void *stream);
int mi_splitter_bank_bind_callback(mi_splitter_bank_t *bank, uint32_t handler, mi_splitter_func_t func, void *object, void *subject, void *stream);
int mi_splitter_bank_unbind(mi_splitter_bank_t *bank, uint32_t handler);
int mi_splitter_bank_clear(mi_splitter_bank_t *bank, void *stream);                /* SpectralSplitter.cpp:246-258 */
/* process(src, count), :295-361.  in: [channels][in_stride] or NULL (silence); outs: HOST array of `handlers` device
 * pointers, each [channels][out_stride] or NULL (no sink).  An output buffer must not overlap the input rows (MI_EINVAL): the
 * reference hands bands to sink functions, never back into the block being read. */
int mi_splitter_bank_process(mi_splitter_bank_t *bank, float *const *outs, const float *in, size_t count, size_t out_stride,
                             size_t in_stride, void *stream);
/*
 * `blocks` consecutive mi_splitter_bank_process calls in one C call (SpectralSplitter.cpp:295-361 per block): block k reads
 * in[k] and hands band i to outs[k * handlers + i] (HOST tables of DEVICE pointers; a band is NULL in every block of a run
 * or in none).  Runs of blocks of whole frames go out as one launch where the several-hops kernel applies; the samples and
 * the state are those of the calls one by one.
 */
int mi_splitter_bank_process_blocks(mi_splitter_bank_t *bank, float *const *outs, const float *const *in, size_t blocks,
                                    size_t count, size_t out_stride, size_t in_stride, void *stream);

/*
 * crossover::* of misc/fft_crossover.h:47-154 (src/main/misc/fft_crossover.cpp): magnitude of the FFT crossover's high-
 * and low-pass at single frequencies, on frequency lists and on FFT-ordered bins.  Host functions, host memory.
 */
float mi_crossover_hipass(float f, float f0, float slope);
float mi_crossover_lopass(float f, float f0, float slope);
void  mi_crossover_hipass_set(float *gain, const float *f, float f0, float slope, size_t count);
void  mi_crossover_hipass_apply(float *gain, const float *f, float f0, float slope, size_t count);
void  mi_crossover_lopass_set(float *gain, const float *f, float f0, float slope, size_t count);
void  mi_crossover_lopass_apply(float *gain, const float *f, float f0, float slope, size_t count);
void  mi_crossover_hipass_fft_set(float *mag, float f0, float slope, float sample_rate, size_t rank);
void  mi_crossover_hipass_fft_apply(float *mag, float f0, float slope, float sample_rate, size_t rank);
void  mi_crossover_lopass_fft_set(float *mag, float f0, float slope, float sample_rate, size_t rank);
void  mi_crossover_lopass_fft_apply(float *mag, float f0, float slope, float sample_rate, size_t rank);

/* ---- delay line and ring buffer banks ----------------------------------------------------- */
/*
 * mi_delay_bank: `channels` x lsp::dspu::Delay (include/lsp-plug.in/dsp-units/util/Delay.h:35-209).  The delay is
 * per channel.  Calls on the whole bank move every line by the same amount; the *_rows calls move the listed lines
 * only, each line then keeps a write position of its own (every reference object has its nHead, Delay.h:35).
 * Index arithmetic is the reference's (uint32 head/tail/size, src/main/util/Delay.cpp:76-102,434,569-574):
 * outputs are bit-exact.
 */
typedef struct mi_delay_bank mi_delay_bank_t;
enum { MI_GAIN_NONE = 0, MI_GAIN_SCALAR = 1, MI_GAIN_VECTOR = 2 };

/* Delay::init(max_size), Delay.cpp:51-66: line length = align(max_size + 0x200, 0x200). */
int mi_delay_bank_create(mi_delay_bank_t **bank, uint32_t channels, size_t max_size);
int mi_delay_bank_destroy(mi_delay_bank_t *bank);
/* Delay::set_delay(delay) (delay %= size), Delay.cpp:569-574; channel = UINT32_MAX for all. */
int mi_delay_bank_set_delay(mi_delay_bank_t *bank, uint32_t channel, size_t delay);
/* get_delay() and the raw nSize / nHead / nTail of one channel. */
int mi_delay_bank_get(const mi_delay_bank_t *bank, uint32_t channel, uint32_t *delay, uint32_t *size, uint32_t *head, uint32_t *tail);
/* Delay::clear(), Delay.cpp:576-582. */
int mi_delay_bank_clear(mi_delay_bank_t *bank, void *stream);
/* Delay::append(src, count), Delay.cpp:76-102. */
int mi_delay_bank_append(mi_delay_bank_t *bank, const float *in, size_t count, size_t in_stride, void *stream);
/*
 * Delay::process / process_add and their gain variants (Delay.cpp:104-397): out (+)= gain * delayed(in).
 * add != 0 selects process_add; gain_mode MI_GAIN_NONE / MI_GAIN_SCALAR (gain) / MI_GAIN_VECTOR
 * (gain_vec: device [channels][gain_stride]).  out may be the same buffer as in.
 */
int mi_delay_bank_process(mi_delay_bank_t *bank, float *out, const float *in, size_t count, size_t out_stride,
                          size_t in_stride, int add, int gain_mode, float gain, const float *gain_vec,
                          size_t gain_stride, void *stream);
/*
 * Delay::process_ramping(dst, src, [gain,] delay, count), Delay.cpp:399-546: the delay of channel c slides
 * linearly to new_delays[c] (HOST array) over the block; tail index (old_tail + ssize_t(delta * offset)) % size.
 */
int mi_delay_bank_process_ramping(mi_delay_bank_t *bank, float *out, const float *in, const uint32_t *new_delays,
                                  size_t count, size_t out_stride, size_t in_stride, int gain_mode, float gain,
                                  const float *gain_vec, size_t gain_stride, void *stream);
/*
 * The same calls for a SUBSET of the bank's lines -- Delay objects with different call histories in one bank:
 * rows (HOST array of n_rows distinct channel indices) names the line behind every row of the call's buffers
 * (row r of in / out / gain_vec belongs to channel rows[r]); only those lines are written and move on.  One launch
 * however the lines' positions differ (they are read from a per-line table on the device).  The first *_rows call
 * switches the bank's kernels to that table; whole-bank calls keep working and move every line.
 */
int mi_delay_bank_append_rows(mi_delay_bank_t *bank, const uint32_t *rows, uint32_t n_rows, const float *in, size_t count,
                              size_t in_stride, void *stream);
int mi_delay_bank_process_rows(mi_delay_bank_t *bank, const uint32_t *rows, uint32_t n_rows, float *out, const float *in,
                               size_t count, size_t out_stride, size_t in_stride, int add, int gain_mode, float gain,
                               const float *gain_vec, size_t gain_stride, void *stream);
/* Delay::process_ramping for the listed lines only: new_delays[r] is the delay line rows[r] slides to (HOST array of n_rows). */
int mi_delay_bank_process_ramping_rows(mi_delay_bank_t *bank, const uint32_t *rows, uint32_t n_rows, float *out, const float *in,
                                       const uint32_t *new_delays, size_t count, size_t out_stride, size_t in_stride, int gain_mode,
                                       float gain, const float *gain_vec, size_t gain_stride, void *stream);

/* mi_ring_bank: `channels` x lsp::dspu::RingBuffer (util/RingBuffer.h:35-179, src/main/util/RingBuffer.cpp:48-209). */
typedef struct mi_ring_bank mi_ring_bank_t;
/* RingBuffer::init(size, fill), RingBuffer.cpp:48-63. */
int mi_ring_bank_create(mi_ring_bank_t **bank, uint32_t channels, size_t size, float fill);
/* The same with the storage in pinned host memory mapped into the device: *host_view ([channels][size]) is the raw
 * storage as RingBuffer::data() exposes it (util/RingBuffer.h:130); valid after the stream of a call has been synchronised. */
int mi_ring_bank_create_shared(mi_ring_bank_t **bank, uint32_t channels, size_t size, float fill, float **host_view);
int mi_ring_bank_destroy(mi_ring_bank_t *bank);
/* clear() / fill(value), RingBuffer.cpp:108-120 (head returns to 0). */
int mi_ring_bank_fill(mi_ring_bank_t *bank, float value, void *stream);
/* append(data, count) -> *appended = min(count, capacity), RingBuffer.cpp:76-106. */
int mi_ring_bank_append(mi_ring_bank_t *bank, const float *in, size_t count, size_t in_stride, size_t *appended, void *stream);
/* get(dst, offset, count) -> *read, zero-filled where nothing is stored, RingBuffer.cpp:147-183.
 * The single-sample get(offset) (RingBuffer.cpp:122-129) is get(dst, offset, 1). */
int mi_ring_bank_get(mi_ring_bank_t *bank, float *out, size_t offset, size_t count, size_t out_stride, size_t *read, void *stream);
/* size(), head position and tail_position(offset), RingBuffer.cpp:140-145. */
int mi_ring_bank_info(const mi_ring_bank_t *bank, size_t offset, uint32_t *capacity, uint32_t *head, uint32_t *tail_position);

/*
 * Host-only introspection of the per-section device table (no GPU needed): the
 * chunk-parallel form of the TDF-II section used by the kernel.
 * variant 0 = 16-sample chunks (calls longer than 2048 samples), 1 = 8-sample
 * chunks; a lane owns two adjacent chunks, a wave 64 such pairs.
 * geometry[4] receives {chunk L, lanes, per-lane matrices, floats_per_row};
 * table (may be NULL) receives one row, 2x2 matrices column-major (m00 m10 m01 m11):
 *   {b0 b1 b2 a1 a2 0 0 0 | P P^2 P^4 P^8 P^16 P^32 | (p[k],q[k]) k<L | (P^2)^(i+1) i<16},  P = A^L.
 */
int mi_biquad_section_tables(const mi_biquad_x1_t *chain, int variant, float *table, uint32_t *geometry);

#ifdef __cplusplus
}
#endif

#endif /* MI_DSPU_H_ */
