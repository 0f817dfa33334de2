// The reference's own unit tests for the hot path, replayed against the lsp::dspu::* classes of this library
// (same class names, same calls, same tolerances):
//   src/test/utest/util/convolver.cpp      test_small (:88-136), test_large (:184-223)
//   src/test/utest/filters/equalizer.cpp   test_latency FIR/FFT/SPM (:35-92)
//   src/test/utest/util/spectral_proc.cpp  test_simple (:37-67)
//   src/test/utest/util/ringbuffer.cpp     (:30-192)
//   README.md:145-230                      the Filter demo (C1)
// Plain C++11, no test framework: exit code 0 == all passed.  Needs a GPU (the classes have no CPU fallback).
#include <lsp-plug.in/dsp-units/filters/Filter.h>
#include <lsp-plug.in/dsp-units/filters/FilterArray.h>
#include <lsp-plug.in/dsp-units/filters/EqualizerArray.h>
#include <lsp-plug.in/dsp-units/util/ConvolverArray.h>
#include <lsp-plug.in/dsp-units/filters/Equalizer.h>
#include <lsp-plug.in/dsp-units/filters/DynamicFilters.h>
#include <lsp-plug.in/dsp-units/util/Convolver.h>
#include <lsp-plug.in/dsp-units/util/SpectralProcessor.h>
#include <lsp-plug.in/dsp-units/util/MultiSpectralProcessor.h>
#include <lsp-plug.in/dsp-units/util/Analyzer.h>
#include <lsp-plug.in/dsp-units/util/Crossover.h>
#include <lsp-plug.in/dsp-units/meters/ILUFSMeter.h>
#include <lsp-plug.in/dsp-units/meters/LoudnessMeter.h>
#include <lsp-plug.in/dsp-units/misc/envelope.h>
#include <lsp-plug.in/dsp-units/misc/fft_crossover.h>
#include <lsp-plug.in/dsp-units/util/FFTCrossover.h>
#include <lsp-plug.in/dsp-units/util/SpectralSplitter.h>
#include <lsp-plug.in/dsp-units/util/RingBuffer.h>
#include <lsp-plug.in/dsp-units/util/Delay.h>
#include <lsp-plug.in/dsp-units/units.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

using namespace lsp;

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++failures; printf("  FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

static void naive_convolve(std::vector<float> &dst, const std::vector<float> &src, const std::vector<float> &conv, size_t count)
{
    for (size_t i = 0; i < count; ++i)
        for (size_t j = 0; j < conv.size(); ++j)
            dst[i + j] += src[i] * conv[j];
}

static void chunked(dspu::Convolver &c, std::vector<float> &dst, const std::vector<float> &src, size_t step)
{
    for (size_t i = 0; i < src.size();)
    {
        const size_t todo = (src.size() - i > step) ? step : src.size() - i;
        c.process(&dst[i], &src[i], todo);
        i += todo;
    }
}

// FloatBuffer::equals_relative of lsp-test-fw 1.0.33 (modules.mk:47-51; not under /root/reference), as the reference's utest
// uses it (src/test/utest/util/convolver.cpp:123): a zero on either side is compared absolutely, everything else by the
// ratio of the two values -- no floor under small values.
static bool equals_relative(float a, float b, float tolerance)
{
    if (a == 0.0f)
        return fabsf(b) < tolerance;
    if (b == 0.0f)
        return fabsf(a) < tolerance;
    const float ratio = (fabsf(a) > fabsf(b)) ? a / b : b / a;
    return fabsf(1.0f - ratio) < tolerance;
}

static void convolver_small()
{
    printf("convolver.test_small\n");
    dspu::Convolver c;
    std::vector<float> conv(0x1f), src(0x2000 + 0x1f, 0.0f);
    for (size_t i = 0; i < conv.size(); ++i) conv[i] = i + 1;
    for (size_t i = 0, j = 0; i < 0x2000; i += 5, ++j)
        src[i] = ((j % 3) == 0) ? 1.0f : ((j % 3) == 1) ? 0.1f : 0.01f;
    std::vector<float> d1(src.size() + conv.size(), 0.0f), d3(src.size(), 0.0f);
    CHECK(c.init(conv.data(), conv.size(), 9, 0), "init");
    naive_convolve(d1, src, conv, 0x2000);
    chunked(c, d3, src, 31);
    for (size_t i = 0; i < src.size(); ++i)
    {
        if (!equals_relative(d3[i], d1[i], 1e-4f)) { CHECK(false, "sample %zu: %.6f vs %.6f", i, d1[i], d3[i]); break; }
    }
    c.destroy();
}

static void convolver_large()
{
    printf("convolver.test_large\n");
    dspu::Convolver c;
    srand(1);
    std::vector<float> conv(0x2000), src(0x20 + 0x2000, 0.0f);
    for (float &v : conv) v = float(rand()) / float(RAND_MAX);
    for (size_t i = 0; i < 0x20; ++i) src[i] = float(rand()) / float(RAND_MAX);
    std::vector<float> d1(src.size() + conv.size(), 0.0f), d3(src.size(), 0.0f);
    CHECK(c.init(conv.data(), conv.size(), 10, 0), "init");
    CHECK(c.data_size() == 0x2000 && c.rank() == 10, "data_size/rank");
    naive_convolve(d1, src, conv, 0x20);
    chunked(c, d3, src, 31);
    for (size_t i = 0; i < src.size(); ++i)
        if (fabsf(d3[i] - d1[i]) > 1e-4f) { CHECK(false, "sample %zu: %.6f vs %.6f", i, d1[i], d3[i]); break; }   // equals_absolute 1e-4
    c.destroy();
}

static void equalizer_latency(const char *label, dspu::equalizer_mode_t mode)
{
    printf("equalizer.test_latency %s\n", label);
    const size_t RANK = 13, BUF = size_t(1) << (RANK + 2);
    dspu::Equalizer eq;
    dspu::filter_params_t fp;
    CHECK(eq.init(1, RANK), "init");
    eq.set_mode(mode);
    eq.set_sample_rate(48000);
    fp.nType = dspu::FLT_BT_LRX_HIPASS; fp.fFreq = 100.0f; fp.fFreq2 = 100.0f; fp.fGain = 1.0f; fp.nSlope = 2; fp.fQuality = 0.0f;
    eq.set_params(0, &fp);
    std::vector<float> src(BUF, 0.0f), dst(BUF, 0.0f);
    src[0] = 1.0f;
    eq.process(dst.data(), src.data(), BUF);
    const size_t latency = eq.get_latency();
    size_t index = 0;
    for (size_t i = 1; i < BUF; ++i)
        if (fabsf(dst[i]) > fabsf(dst[index])) index = i;
    printf("  latency = %zu, maximum = %zu\n", latency, index);
    CHECK(latency == index, "latency %zu != peak %zu", latency, index);
    eq.destroy();
}

static void spectral_proc_simple()
{
    printf("spectral_proc.test_simple\n");
    const size_t SAMPLES = 8192;
    std::vector<float> in(SAMPLES), out(SAMPLES, 0.0f);
    const float w = 2 * M_PI * 440.0f / 48000.0f;
    for (size_t i = 0; i < SAMPLES; ++i) in[i] = sinf(w * i);
    dspu::SpectralProcessor sp;
    CHECK(sp.init(14), "init");
    sp.set_phase(0.0f);
    sp.set_rank(8);
    sp.process(out.data(), in.data(), SAMPLES);
    const size_t latency = sp.latency();
    CHECK(latency == 256, "latency %zu", latency);
    for (size_t i = 0; i < SAMPLES - latency; ++i)
        if (fabsf(in[i] - out[latency + i]) > 1e-5f) { CHECK(false, "sample %zu: %.7f vs %.7f", i, in[i], out[latency + i]); break; }
}

static void halve(void *, void *, float *spectrum, size_t rank)
{
    for (size_t i = 0; i < (size_t(2) << rank); ++i) spectrum[i] *= 0.5f;
}

static void spectral_proc_callback()
{
    printf("spectral_proc host callback (x0.5)\n");
    const size_t SAMPLES = 4096;
    std::vector<float> in(SAMPLES), out(SAMPLES, 0.0f);
    srand(2);
    for (float &v : in) v = float(rand()) / float(RAND_MAX) - 0.5f;
    dspu::SpectralProcessor sp;
    CHECK(sp.init(10), "init");
    sp.bind(halve, NULL, NULL);
    sp.process(out.data(), in.data(), SAMPLES);
    for (size_t i = 0; i + 1024 < SAMPLES; ++i)
        if (fabsf(0.5f * in[i] - out[1024 + i]) > 1e-5f) { CHECK(false, "sample %zu", i); break; }
}

// MultiSpectralProcessor (no reference utest; semantics of MultiSpectralProcessor.cpp:160-393): three channels --
// #0 in+out, #1 input only (spectrum visible to the handler, no output), #2 nothing bound (NULL spectrum pointer);
// the handler writes 0.25 * spectrum[1] into spectrum[0], so out[0] = 0.25 * in[1] delayed by the latency;
// bound pointers advance with process(count).
static int multi_calls = 0, multi_null = 0;
static void multi_mix(void *, void *, float * const *spectrum, size_t rank)
{
    ++multi_calls;
    if (spectrum[2] == NULL) ++multi_null;
    for (size_t i = 0; i < (size_t(2) << rank); ++i) spectrum[0][i] = 0.25f * spectrum[1][i];
}

static void multi_spectral_proc()
{
    printf("multi_spectral_proc (3 channels, cross-channel handler)\n");
    const size_t SAMPLES = 4096, RANK = 9, LAT = 512;
    std::vector<float> in0(SAMPLES), in1(SAMPLES), out0(SAMPLES, -1.0f);
    srand(3);
    for (float &v : in0) v = float(rand()) / float(RAND_MAX) - 0.5f;
    for (float &v : in1) v = float(rand()) / float(RAND_MAX) - 0.5f;
    dspu::MultiSpectralProcessor mp;
    CHECK(!mp.init(0, 10), "init(0 channels) must fail");
    CHECK(mp.bind(0, NULL, NULL) == STATUS_BAD_STATE, "bind before init");
    CHECK(mp.init(3, 10), "init");
    mp.set_rank(RANK);
    CHECK(mp.latency() == LAT && mp.frame_size() == LAT / 2, "latency/frame_size");
    CHECK(mp.bind(3, NULL, NULL) == STATUS_INVALID_VALUE, "bind out of range");
    CHECK(mp.bind(0, out0.data(), in0.data()) == STATUS_OK, "bind 0");
    CHECK(mp.bind_in(1, in1.data()) == STATUS_OK, "bind_in 1");
    mp.bind_handler(multi_mix, NULL, NULL);
    for (size_t done = 0; done < SAMPLES; done += 1000)                 // uneven calls: pointers must advance
        mp.process((SAMPLES - done < 1000) ? SAMPLES - done : 1000);
    CHECK(multi_calls == int(SAMPLES / (LAT / 2)), "handler calls %d", multi_calls);
    CHECK(multi_null == multi_calls, "unbound channel must be NULL in the handler");
    for (size_t i = 0; i + LAT < SAMPLES; ++i)
        if (fabsf(0.25f * in1[i] - out0[LAT + i]) > 1e-5f) { CHECK(false, "sample %zu: %.7f vs %.7f", i, 0.25f * in1[i], out0[LAT + i]); break; }
    mp.destroy();
}

// Crossover (no reference utest): three LR4 bands with per-band handlers that add their data into one buffer -- an LR4
// crossover sums to an all-pass, so the sum keeps the input's energy; handler offsets follow the buf_size chunks.
struct xover_sum_t { std::vector<float> sum; size_t calls; size_t last_first; };
static void xover_collect(void *object, void *, size_t band, const float *data, size_t first, size_t count)
{
    xover_sum_t *s = static_cast<xover_sum_t *>(object);
    for (size_t i = 0; i < count; ++i) s->sum[s->last_first + first + i] += data[i];
    ++s->calls;
    (void)band;
}

static void crossover_bands_sum_to_allpass()
{
    printf("crossover (3 bands LR4, handlers)\n");
    const size_t SAMPLES = 16384, BUF = 1000;
    std::vector<float> in(2 * SAMPLES, 0.0f);
    srand(4);
    for (size_t i = 0; i < SAMPLES; ++i) in[i] = float(rand()) / float(RAND_MAX) - 0.5f;
    dspu::Crossover x;
    CHECK(!x.init(0, BUF), "init(0 bands) must fail");
    CHECK(x.init(3, BUF), "init");
    CHECK(x.num_bands() == 3 && x.num_splits() == 2 && x.max_buffer_size() == BUF, "geometry");
    x.set_sample_rate(48000);
    x.set_slope(0, dspu::CROSS_SLOPE_LR4); x.set_frequency(0, 300.0f);
    x.set_slope(1, dspu::CROSS_SLOPE_LR4); x.set_frequency(1, 3000.0f);
    CHECK(x.get_slope(1) == dspu::CROSS_SLOPE_LR4 && x.get_frequency(0) == 300.0f && x.get_mode(0) == dspu::CROSS_MODE_BT, "getters");
    CHECK(x.get_slope(2) == -1 && x.get_gain(3) == -1.0f, "getters out of range");
    CHECK(x.band_active(0) && x.band_active(1) && x.band_active(2), "bands active");
    CHECK(x.get_band_end(0) == 300.0f && x.get_band_start(2) == 3000.0f && x.get_band_end(2) == 24000.0f, "band ranges");
    xover_sum_t acc; acc.sum.assign(2 * SAMPLES, 0.0f); acc.calls = 0; acc.last_first = 0;
    for (size_t b = 0; b < 3; ++b) CHECK(x.set_handler(b, xover_collect, &acc, NULL), "set_handler");
    CHECK(!x.set_handler(3, xover_collect, &acc, NULL), "set_handler out of range");
    x.process(in.data(), 2 * SAMPLES);
    CHECK(acc.calls == 3 * ((2 * SAMPLES + BUF - 1) / BUF), "handler calls %zu", acc.calls);
    double ein = 0.0, eout = 0.0;
    for (size_t i = 0; i < 2 * SAMPLES; ++i) { ein += double(in[i]) * in[i]; eout += double(acc.sum[i]) * acc.sum[i]; }
    CHECK(fabs(eout / ein - 1.0) < 2e-3, "energy ratio %.5f", eout / ein);
    float c[4]; const float f[2] = { 30.0f, 10000.0f };
    CHECK(x.freq_chart(0, c, f, 2) && fabsf(hypotf(c[0], c[1]) - 1.0f) < 1e-2f && hypotf(c[2], c[3]) < 1e-3f, "freq_chart band 0");
    x.destroy();
}

// LoudnessMeter (no reference utest): ITU-R BS.1770-4 anchor -- a 0 dBFS 997 Hz sine on one front channel reads
// -3.01 LKFS -- plus the bookkeeping of bound outputs (offset advances, link 0 = the channel's own RMS).
static void loudness_meter_bs1770()
{
    printf("loudness_meter (BS.1770 sine anchor)\n");
    const size_t SR = 48000, N = 48000;
    std::vector<float> l(N), r(N, 0.0f), out(N), lout(2 * N, -1.0f);
    for (size_t i = 0; i < N; ++i) l[i] = sinf(2.0f * float(M_PI) * 997.0f * float(i) / float(SR));
    dspu::LoudnessMeter m;
    CHECK(m.init(2) == STATUS_OK, "init");
    CHECK(m.designation(0) == dspu::bs::CHANNEL_LEFT && m.designation(1) == dspu::bs::CHANNEL_RIGHT, "default designations");
    CHECK(m.set_sample_rate(SR) == STATUS_OK, "set_sample_rate");
    CHECK(m.latency() == 19200, "latency %zu", m.latency());
    CHECK(m.bind(2, NULL, NULL) == STATUS_OVERFLOW, "bind out of range");
    CHECK(m.bind(0, lout.data(), l.data(), N / 2) == STATUS_OK && m.bind(1, NULL, r.data()) == STATUS_OK, "bind");
    CHECK(m.set_link(0, 0.0f) == STATUS_OK, "set_link");
    m.process(out.data(), N / 2);
    CHECK(m.bind(0, lout.data(), l.data() + N / 2, N) == STATUS_OK, "rebind");          // second half of the signal
    m.process(out.data() + N / 2, N / 2);
    const float lkfs = -0.691f + 20.0f * log10f(out[N - 1]);
    printf("  loudness %.3f LKFS\n", lkfs);
    CHECK(fabsf(lkfs + 3.01f) < 0.02f, "LKFS %.3f", lkfs);
    CHECK(fabsf(m.loudness() - out[N - 1]) < 1e-7f, "loudness()");
    CHECK(lout[N / 2 - 1] == -1.0f && lout[N / 2] >= 0.0f && lout[3 * N / 2 - 1] > 0.5f && lout[3 * N / 2] == -1.0f, "bound output window");
    CHECK(fabsf(lout[3 * N / 2 - 1] - out[N - 1]) < 1e-6f, "link 0 of the only sounding channel equals the mix");
    m.destroy();
}

// ILUFSMeter, the flow of the reference's manual test (src/test/mtest/meters/ilufs.cpp:44-78: stereo file, integration
// period = file length, blocks of 0x400) on a synthetic signal with a known answer: a 0 dBFS 997 Hz sine on the left
// channel integrates to -3.01 LUFS.
static void ilufs_meter_mtest_flow()
{
    printf("ilufs_meter (mtest flow, BS.1770 sine anchor)\n");
    const size_t SR = 48000, N = 2 * 48000, BUF = 0x400;
    std::vector<float> l(N), r(N, 0.0f), out(N, -1.0f);
    for (size_t i = 0; i < N; ++i) l[i] = sinf(2.0f * float(M_PI) * 997.0f * float(i) / float(SR));
    dspu::ILUFSMeter lm;
    const float integration_period = float(N) / float(SR);
    CHECK(lm.init(2, integration_period, dspu::bs::LUFS_MEASURE_PERIOD_MS) == STATUS_OK, "init");
    CHECK(lm.set_sample_rate(SR) == STATUS_OK, "set_sample_rate");
    lm.set_integration_period(integration_period);
    lm.set_weighting(dspu::bs::WEIGHT_K);
    lm.set_active(0, true);
    lm.set_active(1, true);
    CHECK(lm.set_designation(0, dspu::bs::CHANNEL_LEFT) == STATUS_OK && lm.set_designation(1, dspu::bs::CHANNEL_RIGHT) == STATUS_OK, "designation");
    CHECK(lm.set_designation(2, dspu::bs::CHANNEL_LEFT) == STATUS_OVERFLOW, "designation out of range");
    for (size_t offset = 0; offset < N; )
    {
        const size_t to_process = std::min(N - offset, BUF);
        lm.bind(0, &l[offset]);
        lm.bind(1, &r[offset]);
        lm.process(&out[offset], to_process);
        offset += to_process;
    }
    const float lufs = 20.0f * log10f(lm.loudness() * dspu::bs::DBFS_TO_LUFS_SHIFT_GAIN);
    printf("  integrated loudness %.3f LUFS\n", lufs);
    CHECK(fabsf(lufs + 3.01f) < 0.02f, "LUFS %.3f", lufs);
    CHECK(out[0] == 0.0f && out[19199] == 0.0f && out[N - 1] > 0.6f, "held output");
    lm.clear();
    CHECK(lm.loudness() == 0.0f, "clear");
    lm.destroy();
}

// SpectralSplitter, the flow of the reference's manual test (src/test/mtest/util/spectral_splitter.cpp:86-150: four
// brick-wall bands on rank 12 with chunk rank 10, process(src) then process(NULL, 4096)) on seeded noise, with the
// property the manual test lets one hear: the bands add up to the input delayed by latency().
struct split_band_t { size_t imin, imax, offset; std::vector<float> s; };

static void split_func(void *, void *subject, float *out, const float *in, size_t rank)
{
    split_band_t *band = static_cast<split_band_t *>(subject);
    const size_t len = size_t(1) << rank, freq = len >> 1;
    for (size_t i = 0; i < len; ++i)
    {
        const size_t idx = (i < freq) ? i : len - i;
        const bool keep = (idx >= band->imin) && (idx < band->imax);
        out[2 * i]     = keep ? in[2 * i] : 0.0f;
        out[2 * i + 1] = keep ? in[2 * i + 1] : 0.0f;
    }
}

static void split_sink(void *, void *subject, const float *samples, size_t, size_t count)
{
    split_band_t *band = static_cast<split_band_t *>(subject);
    memcpy(&band->s[band->offset], samples, count * sizeof(float));
    band->offset += count;
}

static void spectral_splitter_mtest_flow()
{
    printf("spectral_splitter (mtest flow: brick-wall bands add up to the delayed input)\n");
    const size_t rank = 12, xlength = size_t(1) << rank, N = 20000, SR = 48000;
    std::vector<float> src(N);
    uint32_t seed = 12345;
    for (size_t i = 0; i < N; ++i) { seed = seed * 1664525u + 1013904223u; src[i] = float(int32_t(seed)) * (0.5f / 2147483648.0f); }
    const float flist[] = { 0.0f, 100.0f, 1000.0f, 10000.0f, SR * 0.5f + 100.0f };
    split_band_t bands[4];
    for (size_t i = 0; i < 4; ++i)
    {
        bands[i].imin = size_t((flist[i] * xlength) / SR);
        bands[i].imax = size_t((flist[i + 1] * xlength) / SR);
        bands[i].offset = 0;
        bands[i].s.assign(N + xlength, 0.0f);
    }
    dspu::SpectralSplitter split;
    CHECK(split.init(4, 6) == STATUS_INVALID_VALUE, "init with rank 4");
    CHECK(split.init(rank, 6) == STATUS_OK, "init");
    split.set_rank(rank);
    split.set_chunk_rank(rank - 2);
    split.set_phase(0);
    CHECK(split.latency() == 1024 && split.needs_update(), "latency %zu", split.latency());
    CHECK(split.bind(6, NULL, NULL, split_func, split_sink) == STATUS_OVERFLOW, "bind out of range");
    CHECK(split.bind(0, NULL, NULL, NULL, NULL) == STATUS_INVALID_VALUE, "bind nothing");
    CHECK(split.unbind(0) == STATUS_NOT_BOUND, "unbind unbound");
    for (size_t i = 0; i < 4; ++i)
        CHECK(split.bind(i, NULL, &bands[i], split_func, split_sink) == STATUS_OK, "bind");
    CHECK(split.bindings() == 4 && split.bound(3) && !split.bound(4), "bindings");
    split.process(src.data(), N);
    split.process(NULL, xlength);
    CHECK(bands[0].offset == N + xlength, "sink sample count %zu", bands[0].offset);
    const size_t lat = split.latency();
    float err = 0.0f, peak[4] = { 0, 0, 0, 0 };
    for (size_t i = 2 * lat; i < N; ++i)
    {
        float sum = 0.0f;
        for (size_t b = 0; b < 4; ++b) { sum += bands[b].s[i]; peak[b] = std::max(peak[b], fabsf(bands[b].s[i])); }
        err = std::max(err, fabsf(sum - src[i - lat]));
    }
    printf("  latency %zu, max |sum of bands - delayed input| = %g\n", lat, err);
    CHECK(err < 5e-5f, "bands do not add up: %g", err);
    CHECK(peak[0] > 1e-3f && peak[1] > 1e-2f && peak[2] > 1e-2f && peak[3] > 1e-2f, "a band is silent");
    split.destroy();
}

// FFTCrossover, the band plan of the reference's manual test (src/test/mtest/util/fft_crossover.cpp:74-110).  Every band
// is flattened to -3 dB there; with flatten = 1 (and the same frequencies and slopes) neighbouring bands are exact
// complements at their crossover point, so the five outputs add up to the delayed input within the steep-slope leakage.
struct xover_band_t { size_t offset; std::vector<float> s; };

static void xover_func(void *, void *subject, size_t, const float *data, size_t, size_t count)
{
    xover_band_t *b = static_cast<xover_band_t *>(subject);
    memcpy(&b->s[b->offset], data, count * sizeof(float));
    b->offset += count;
}

static void fft_crossover_mtest_flow()
{
    printf("fft_crossover (mtest flow: five bands add up to the delayed input)\n");
    const size_t rank = 12, xlength = size_t(1) << rank, N = 20000, SR = 48000;
    std::vector<float> src(N);
    uint32_t seed = 777;
    for (size_t i = 0; i < N; ++i) { seed = seed * 1664525u + 1013904223u; src[i] = float(int32_t(seed)) * (0.5f / 2147483648.0f); }
    xover_band_t bands[5];
    for (size_t i = 0; i < 5; ++i) { bands[i].offset = 0; bands[i].s.assign(N + xlength, 0.0f); }
    dspu::FFTCrossover xo;
    CHECK(xo.init(rank, 5) == STATUS_OK, "init");
    xo.set_sample_rate(SR);
    const float split[] = { 90.0f, 425.0f, 1750.0f, 7300.0f };
    for (size_t i = 0; i < 5; ++i)
    {
        if (i > 0) xo.set_hpf(i, split[i - 1], -32.0f, true);
        if (i < 4) xo.set_lpf(i, split[i], -32.0f, true);
        xo.enable_band(i, true);
        CHECK(xo.set_handler(i, xover_func, NULL, &bands[i]), "set_handler");
    }
    CHECK(!xo.set_handler(5, xover_func, NULL, NULL), "set_handler out of range");
    CHECK(xo.latency() == xlength && xo.bands() == 5 && xo.hpf_enabled(1) && !xo.hpf_enabled(0), "settings");
    xo.process(src.data(), N);
    xo.process(NULL, xlength);
    CHECK(bands[4].offset == N + xlength, "handler sample count %zu", bands[4].offset);
    const size_t lat = xo.latency();
    float err = 0.0f;
    for (size_t i = 2 * lat; i < N; ++i)
    {
        float sum = 0.0f;
        for (size_t b = 0; b < 5; ++b) sum += bands[b].s[i];
        err = std::max(err, fabsf(sum - src[i - lat]));
    }
    printf("  max |sum of bands - delayed input| = %g\n", err);
    CHECK(err < 5e-4f, "bands do not add up: %g", err);
    float f[3] = { 425.0f, 1000.0f, 40.0f }, m[3];
    CHECK(xo.freq_chart(2, m, f, 3) && m[0] == 0.5f * dspu::crossover::lopass(425.0f, 1750.0f, -32.0f) && m[1] > 0.9f && m[2] < 1e-3f,
          "freq_chart %g %g %g", m[0], m[1], m[2]);
    // a disabled band stops receiving data; the reference's flag rule: set_lpf(.., false) alone requests no update
    xo.enable_band(0, false);
    xo.set_lpf(1, 425.0f, -32.0f, false);
    CHECK(!xo.needs_update() && !xo.lpf_enabled(1), "update flag rule");
    const size_t before0 = bands[0].offset, before1 = bands[1].offset;
    for (size_t i = 0; i < 5; ++i) bands[i].s.resize(bands[i].s.size() + 512);
    xo.process(src.data(), 512);
    CHECK(bands[0].offset == before0 && bands[1].offset == before1 + 512, "disabled band");
    xo.destroy();
}

static void ringbuffer()
{
    printf("ringbuffer\n");
    dspu::RingBuffer rb;
    float dst[16];
    CHECK(rb.init(8), "init");
    CHECK(rb.size() == 8, "size");
    rb.append(1.0f); rb.append(2.0f); rb.append(3.0f); rb.append(4.0f);
    const float e1[9] = { 0, 0, 0, 0, 0, 1, 2, 3, 4 };
    for (int o = 8; o >= 0; --o) CHECK(rb.get(size_t(o)) == e1[8 - o], "get(%d)", o);
    static const float buf1[2] = { 5.0f, 6.0f };
    CHECK(rb.append(buf1, 2) == 2, "append 2");
    CHECK(rb.get(dst, 9, 10) == 8, "get(dst, 9, 10)");
    const float e2[10] = { 0, 0, 0, 0, 1, 2, 3, 4, 5, 6 };
    CHECK(memcmp(dst, e2, sizeof(e2)) == 0, "block 1");
    static const float buf2[4] = { 7.0f, 8.0f, 9.0f, 10.0f };
    CHECK(rb.append(buf2, 4) == 4, "append 4");
    CHECK(rb.get(dst, 7, 10) == 8, "get(dst, 7, 10)");
    const float e3[10] = { 3, 4, 5, 6, 7, 8, 9, 10, 0, 0 };
    CHECK(memcmp(dst, e3, sizeof(e3)) == 0, "block 2");
    static const float buf3[12] = { -1, -2, -3, -4, -5, -6, -7, -8, -9, -10, -11, -12 };
    CHECK(rb.append(buf3, 12) == 8, "append 12");
    CHECK(rb.get(dst, 16, 8) == 0, "get(dst, 16, 8)");
    CHECK(rb.get(dst, 12, 16) == 8, "get(dst, 12, 16)");
    const float e4[16] = { 0, 0, 0, 0, 0, -5, -6, -7, -8, -9, -10, -11, -12, 0, 0, 0 };
    CHECK(memcmp(dst, e4, sizeof(e4)) == 0, "block 3");
    CHECK(rb.get(&dst[0], 8, 2) == 1 && rb.get(&dst[2], 6, 2) == 2 && rb.get(&dst[8], 0, 2) == 1, "short reads");
    CHECK(dst[0] == 0.0f && dst[1] == -5.0f && dst[2] == -6.0f && dst[3] == -7.0f && dst[8] == -12.0f && dst[9] == 0.0f, "short read values");
    // raw positions: read(tail_position(o)) is get(o); lerp_get interpolates between neighbours (RingBuffer.cpp:122-145)
    for (size_t o = 0; o < 8; ++o)
        CHECK(rb.read(rb.tail_position(o)) == rb.get(o), "read(tail_position(%zu))", o);
    CHECK(rb.read(8) == 0.0f && rb.read(dst, 8, 4) == 0, "read past the end");
    CHECK(rb.lerp_get(1.5f) == 0.5f * (rb.get(1) + rb.get(2)) && rb.lerp_get(3.0f) == rb.get(3), "lerp_get");
    const size_t p0 = rb.tail_position(5);                  // three samples up to the end of the storage or fewer
    const size_t n = std::min<size_t>(3, 8 - p0);
    CHECK(rb.read(dst, p0, n) == n, "block read");
    for (size_t i = 0; i < n; ++i)
        CHECK(dst[i] == rb.read(p0 + i), "block read value %zu", i);
}

// Getters and state flags that the reference keeps next to the hot path (Analyzer.h:144-354, Equalizer.h:143-261)
static void accessors()
{
    printf("accessors (Analyzer, Equalizer)\n");
    dspu::Analyzer a;
    CHECK(a.init(2, 10, 96000, 5.5f, 100), "analyzer init");
    CHECK(a.get_channels() == 2 && a.get_rank() == 10 && a.get_max_sample_rate() == 96000 && a.get_min_rate() == 5.0f, "init values");
    CHECK(a.get_window() == size_t(dspu::windows::HANN) && a.get_envelope() == size_t(dspu::envelope::PINK_NOISE) && a.get_shift() == 1.0f && a.get_reactivity() == 0.0f && a.activity(), "defaults");
    CHECK(a.needs_reconfiguration(), "dirty after init");
    a.set_sample_rate(192000);
    CHECK(a.get_sample_rate() == 96000, "sample rate clamp %zu", a.get_sample_rate());
    a.set_rate(1.0f);
    CHECK(a.get_rate() == 5.0f, "rate clamp %g", a.get_rate());
    CHECK(!a.set_rank(11) && !a.set_rank(1) && a.set_rank(9) && a.get_rank() == 9, "set_rank range");
    a.set_window(dspu::windows::BLACKMAN); a.set_envelope(0); a.set_shift(2.0f); a.set_reactivity(0.2f);
    CHECK(a.get_window() == size_t(dspu::windows::BLACKMAN) && a.get_envelope() == 0 && a.get_shift() == 2.0f && a.get_reactivity() == 0.2f, "setters");
    a.reconfigure();
    CHECK(!a.needs_reconfiguration(), "clean after reconfigure");
    a.set_shift(2.0f);
    CHECK(!a.needs_reconfiguration(), "same value requests nothing");
    CHECK(!a.enable_channel(1, true) && a.enable_channel(1, false) && !a.channel_active(1) && a.channel_active(0) && a.needs_reconfiguration(), "enable_channel");
    CHECK(!a.set_channel_delay(0, 101) && a.set_channel_delay(0, 100) && a.channel_delay(0) == 100 && a.channel_delay(5) == 0, "channel delay");
    float f[5];
    CHECK(a.read_frequencies(f, 10.0f, 160.0f, 5) && f[0] == 10.0f && f[4] == 160.0f && fabsf(f[2] - 40.0f) < 1e-3f, "log frequencies %g", f[2]);
    CHECK(a.read_frequencies(f, 10.0f, 50.0f, 5, dspu::FRQA_SCALE_LINEAR) && f[1] == 20.0f && f[4] == 50.0f, "linear frequencies");
    CHECK(!a.read_frequencies(f, 10.0f, 50.0f, 5, 7) && !a.read_frequencies(f, 10.0f, 50.0f, 0), "bad frequency requests");
    a.destroy();

    float env[5];
    dspu::envelope::reverse_noise_lin(env, 0.0f, 400.0f, 100.0f, 5, dspu::envelope::PINK_NOISE);      // blue: sqrt(f / centre)
    CHECK(env[0] == env[1] && env[1] == 1.0f && fabsf(env[4] - 2.0f) < 1e-6f, "reverse pink envelope %g %g", env[1], env[4]);
    dspu::envelope::brown_noise_lin(env, 100.0f, 500.0f, 100.0f, 5, dspu::envelope::WHITE_NOISE);
    CHECK(env[0] == 1.0f && fabsf(env[4] - 0.2f) < 1e-6f, "brown envelope %g", env[4]);
    dspu::envelope::noise_log(env, 100.0f, 1600.0f, 100.0f, 5, dspu::envelope::VIOLET_NOISE);         // octaves: 1 2 4 8 16
    CHECK(fabsf(env[0] - 1.0f) < 1e-5f && fabsf(env[2] - 4.0f) < 1e-4f && fabsf(env[4] - 16.0f) < 1e-3f, "violet on a log grid %g %g", env[2], env[4]);
    const float fl[3] = { 50.0f, 100.0f, 400.0f };
    dspu::envelope::reverse_noise_list(env, fl, 100.0f, 3, dspu::envelope::BROWN_NOISE);               // violet: f / centre
    CHECK(fabsf(env[0] - 0.5f) < 1e-6f && env[1] == 1.0f && fabsf(env[2] - 4.0f) < 1e-6f, "reverse brown on a list %g", env[2]);
    dspu::envelope::pink_noise_list(env, fl, 100.0f, 3, dspu::envelope::WHITE_NOISE);
    CHECK(fabsf(env[2] - 0.5f) < 1e-6f, "pink on a list %g", env[2]);

    {   // needs_update() / update_settings() of the meters, needs_reconfiguration() of the crossover
        dspu::LoudnessMeter lm;
        CHECK(lm.init(1, 400.0f) == STATUS_OK && lm.set_sample_rate(48000) == STATUS_OK, "loudness meter init");
        CHECK(lm.needs_update(), "settings pending after set_sample_rate");
        lm.update_settings();
        CHECK(!lm.needs_update(), "clean after update_settings");
        lm.set_period(200.0f);
        CHECK(lm.needs_update(), "a new period is pending");
        lm.destroy();
        dspu::ILUFSMeter im;
        CHECK(im.init(1, 5.0f, 400.0f) == STATUS_OK && im.set_sample_rate(48000) == STATUS_OK, "ilufs meter init");
        CHECK(im.needs_update(), "settings pending after set_sample_rate");
        im.update_settings();
        CHECK(!im.needs_update(), "clean after update_settings");
        im.destroy();
        dspu::Crossover xo;
        CHECK(xo.init(3, 256), "crossover init");
        xo.set_sample_rate(48000);
        CHECK(xo.needs_reconfiguration(), "dirty after init");
        xo.reconfigure();
        CHECK(!xo.needs_reconfiguration(), "clean after reconfigure");
        xo.set_frequency(0, 1234.0f);
        CHECK(xo.needs_reconfiguration(), "a moved split point is pending");
        xo.destroy();
    }

    dspu::Equalizer eq;
    CHECK(eq.init(2, 8), "equalizer init");
    eq.set_sample_rate(48000);
    dspu::filter_params_t fp;
    fp.nType = dspu::FLT_BT_RLC_BELL; fp.fFreq = 1000.0f; fp.fFreq2 = 1000.0f; fp.fGain = 2.0f; fp.nSlope = 1; fp.fQuality = 1.0f;
    eq.set_params(0, &fp);
    eq.set_mode(dspu::EQM_IIR);
    // Filter::update() parks the filter in FM_BYPASS until the equalizer rebuilds (Filter.cpp:150)
    CHECK(eq.filter_inactive(0) && !eq.filter_active(0) && eq.configuration_changed(), "stale filter reads inactive");
    (void)eq.get_latency();
    CHECK(eq.filter_active(0) && !eq.filter_inactive(0) && eq.filter_inactive(1) && !eq.configuration_changed(), "after reconfigure");
    CHECK(!eq.filter_active(2) && !eq.filter_inactive(2), "bad id");
    CHECK(eq.fir_ir_size() == 512 && eq.actual_sample_rate() == 48000, "sizes");
    eq.set_actual_sample_rate(44100);
    CHECK(eq.actual_sample_rate() == 44100, "actual sample rate");
    eq.destroy();
}

static void readme_filter()
{
    printf("README filter demo (C1)\n");
    dspu::Filter f;
    dspu::filter_params_t fp;
    fp.nType = dspu::FLT_BT_BWC_HISHELF; fp.fFreq = 1000.0f; fp.fFreq2 = 1000.0f;
    fp.fGain = dspu::db_to_gain(6.0f); fp.nSlope = 2; fp.fQuality = 0.0f;
    CHECK(f.init(NULL), "init");
    f.update(48000, &fp);
    std::vector<float> c(48000, 0.0f);
    c[0] = 1.0f;
    f.clear();
    f.process(c.data(), c.data(), c.size());                        // in place, as the README does
    const float head[8] = { 1.93714225f, -0.114121534f, -0.109540939f, -0.10430833f, -0.0985007137f,
                            -0.0922033042f, -0.0855074227f, -0.0785082579f };
    for (int i = 0; i < 8; ++i) CHECK(fabsf(c[i] - head[i]) <= 1e-6f, "impulse[%d] = %.8f", i, c[i]);
    // delay line used as the README-style latency compensation (utest/dynamics/limiter.cpp:58-77 pattern)
    dspu::Delay d;
    CHECK(d.init(1000), "delay init");
    d.set_delay(480);
    std::vector<float> y(2000);
    d.process(y.data(), c.data(), 2000);
    CHECK(y[479] == 0.0f && y[480] == c[0] && y[1999] == c[1519], "delay");
}

// ---- FilterArray: N Filter objects behind one bank (this library's extension) ------------------------------------------
// Every object of the array must behave like the dspu::Filter it stands for: same designer, same lazy rebuild, the filter
// memory cleared when the section count changes and kept otherwise, bypass for FLT_NONE -- checked block by block against
// N separate Filter objects, over retunes in the middle of the stream, on host rows and on device rows.
static void filter_array_equals_n_filters()
{
    printf("FilterArray against separate Filter objects\n");
    const size_t N = 9, n = 3000, blocks = 4;
    const int types[N] = { dspu::FLT_BT_RLC_BELL, dspu::FLT_BT_BWC_HISHELF, dspu::FLT_BT_LRX_LOPASS, dspu::FLT_MT_RLC_BELL,
                           dspu::FLT_NONE, dspu::FLT_BT_RLC_LOPASS, dspu::FLT_BT_BWC_HIPASS, dspu::FLT_BT_RLC_NOTCH, dspu::FLT_BT_LRX_HISHELF };
    dspu::FilterArray fa;
    CHECK(fa.init(N, 16), "FilterArray::init");
    CHECK(fa.size() == N, "size");
    std::vector<dspu::Filter> fl(N);
    std::vector<dspu::filter_params_t> fp(N);
    for (size_t i = 0; i < N; ++i)
    {
        fp[i].nType = types[i]; fp[i].nSlope = 1 + (i % 3); fp[i].fFreq = 300.0f * float(i + 1); fp[i].fFreq2 = fp[i].fFreq * 2.0f;
        fp[i].fGain = (i & 1) ? 2.0f : 0.5f; fp[i].fQuality = 0.3f * float(i % 4);
        CHECK(fl[i].init(NULL), "Filter::init");
        fl[i].update(48000, &fp[i]);
        CHECK(fa.update(i, 48000, &fp[i]), "FilterArray::update %zu", i);
    }
    dspu::filter_params_t big = fp[2];
    big.nSlope = 20;                                                 // LRX slope 20 = 40 sections: more than the array's 16
    CHECK(!fa.update(2, 48000, &big), "an over-long design must be refused");
    std::vector<float> x(N * n), ya(N * n), yf(N * n);
    float *din = NULL, *dout = NULL;
    CHECK(mi_dspu_malloc(reinterpret_cast<void **>(&din), x.size() * sizeof(float)) == MI_OK &&
          mi_dspu_malloc(reinterpret_cast<void **>(&dout), x.size() * sizeof(float)) == MI_OK, "device rows");
    unsigned seed = 12345;
    for (size_t b = 0; b < blocks; ++b)
    {
        for (size_t i = 0; i < x.size(); ++i) { seed = seed * 1664525u + 1013904223u; x[i] = (float(seed >> 8) / 8388608.0f - 1.0f) * 0.25f; }
        if (b == 2)                                                  // retune in mid-stream: object 0 keeps its section count (memory kept),
        {                                                            // object 1 changes it (memory cleared), object 4 comes alive
            fp[0].fGain = 1.5f;                 fl[0].update(48000, &fp[0]); CHECK(fa.update(0, 48000, &fp[0]), "retune 0");
            fp[1].nSlope = 3;                   fl[1].update(48000, &fp[1]); CHECK(fa.update(1, 48000, &fp[1]), "retune 1");
            fp[4].nType = dspu::FLT_BT_RLC_BELL; fl[4].update(48000, &fp[4]); CHECK(fa.update(4, 48000, &fp[4]), "retune 4");
        }
        for (size_t i = 0; i < N; ++i)
            fl[i].process(&yf[i * n], &x[i * n], n);
        if (b & 1)                                                   // device rows
        {
            CHECK(mi_dspu_copy_h2d(din, x.data(), x.size() * sizeof(float), NULL) == MI_OK, "h2d");
            CHECK(fa.process(dout, din, n, n), "FilterArray::process");
            CHECK(mi_dspu_copy_d2h(ya.data(), dout, ya.size() * sizeof(float), NULL) == MI_OK && mi_dspu_stream_synchronize(NULL) == MI_OK, "d2h");
        }
        else
            CHECK(fa.process_host(ya.data(), x.data(), n, n), "FilterArray::process_host");
        // the same kernels run either way (one row of a nine-row bank against nine one-row banks): bit for bit
        size_t bad = 0;
        for (size_t i = 0; i < x.size(); ++i)
            bad += (ya[i] != yf[i]);
        CHECK(bad == 0, "block %zu: %zu samples differ from the separate Filter objects", b, bad);
    }
    dspu::filter_params_t back;
    CHECK(fa.get_params(0, &back) && back.fGain == 1.5f, "get_params");
    mi_dspu_free(din); mi_dspu_free(dout);
    // a run of blocks in one call (ONE launch for blocks of more than 2048 samples): the samples of the calls one by one
    {
        const size_t nb = 4096, K = 5;
        std::vector<float> xb(K * N * nb), y1(K * N * nb), y2(K * N * nb);
        for (size_t i = 0; i < xb.size(); ++i) { seed = seed * 1664525u + 1013904223u; xb[i] = (float(seed >> 8) / 8388608.0f - 1.0f) * 0.25f; }
        float *bi = NULL, *bo = NULL;
        CHECK(mi_dspu_malloc(reinterpret_cast<void **>(&bi), xb.size() * sizeof(float)) == MI_OK &&
              mi_dspu_malloc(reinterpret_cast<void **>(&bo), xb.size() * sizeof(float)) == MI_OK, "device rows");
        CHECK(mi_dspu_copy_h2d(bi, xb.data(), xb.size() * sizeof(float), NULL) == MI_OK, "h2d");
        dspu::FilterArray fb;
        CHECK(fb.init(N, 16), "FilterArray::init");
        for (size_t i = 0; i < N; ++i)
            CHECK(fb.update(i, 48000, &fp[i]), "update");
        float *po[K]; const float *pi[K];
        for (size_t k = 0; k < K; ++k) { po[k] = bo + k * N * nb; pi[k] = bi + k * N * nb; }
        CHECK(fb.process_blocks(po, pi, K, nb, nb), "FilterArray::process_blocks");
        CHECK(mi_dspu_copy_d2h(y1.data(), bo, y1.size() * sizeof(float), NULL) == MI_OK && mi_dspu_stream_synchronize(NULL) == MI_OK, "d2h");
        dspu::FilterArray fc;
        CHECK(fc.init(N, 16), "FilterArray::init");
        for (size_t i = 0; i < N; ++i)
            CHECK(fc.update(i, 48000, &fp[i]), "update");
        for (size_t k = 0; k < K; ++k)
            CHECK(fc.process(bo + k * N * nb, bi + k * N * nb, nb, nb), "process");
        CHECK(mi_dspu_copy_d2h(y2.data(), bo, y2.size() * sizeof(float), NULL) == MI_OK && mi_dspu_stream_synchronize(NULL) == MI_OK, "d2h");
        size_t bad = 0;
        for (size_t i = 0; i < y1.size(); ++i)
            bad += (y1[i] != y2[i]);
        CHECK(bad == 0, "process_blocks: %zu samples differ from block-by-block calls", bad);
        mi_dspu_free(bi); mi_dspu_free(bo);
    }
}

// ---- EqualizerArray / ConvolverArray: N objects behind one bank (this library's extensions) ---------------------------------
// Object c of an array must behave like the dspu::Equalizer / dspu::Convolver it stands for: checked block by block against N
// separate objects (the same kernels on one row of an N-row bank and on N one-row banks: bit for bit), on device rows, on host
// rows, and -- the equalizer -- over a run of blocks in one call.
static void equalizer_array_equals_n_equalizers()
{
    printf("EqualizerArray against separate Equalizer objects\n");
    const size_t N = 5, NF = 4, RANK = 9, n = 512, blocks = 6;
    dspu::EqualizerArray ea;
    CHECK(ea.init(N, NF, RANK) && ea.size() == N, "EqualizerArray::init");
    ea.set_mode(dspu::EQM_FIR);
    ea.set_sample_rate(48000);
    std::vector<dspu::Equalizer> eq(N);
    for (size_t c = 0; c < N; ++c)
    {
        CHECK(eq[c].init(NF, RANK), "Equalizer::init");
        eq[c].set_mode(dspu::EQM_FIR);
        eq[c].set_sample_rate(48000);
        for (size_t i = 0; i < NF; ++i)
        {
            dspu::filter_params_t fp;
            fp.nType = dspu::FLT_BT_RLC_BELL; fp.nSlope = 1; fp.fFreq = fp.fFreq2 = 200.0f * float(1 + i) * float(1 + c);
            fp.fGain = (i + c) & 1 ? 2.0f : 0.5f; fp.fQuality = 1.0f;
            eq[c].set_params(i, &fp);
            CHECK(ea.set_params(c, i, &fp), "EqualizerArray::set_params");
        }
    }
    CHECK(ea.get_latency() == eq[0].get_latency(), "latency %zu != %zu", ea.get_latency(), size_t(eq[0].get_latency()));
    std::vector<float> x(blocks * N * n), ya(blocks * N * n), ye(blocks * N * n);
    unsigned seed = 777;
    for (size_t i = 0; i < x.size(); ++i) { seed = seed * 1664525u + 1013904223u; x[i] = (float(seed >> 8) / 8388608.0f - 1.0f) * 0.25f; }
    for (size_t b = 0; b < blocks; ++b)
        for (size_t c = 0; c < N; ++c)
            eq[c].process(&ye[(b * N + c) * n], &x[(b * N + c) * n], n);
    float *din = NULL, *dout = NULL;
    CHECK(mi_dspu_malloc(reinterpret_cast<void **>(&din), x.size() * sizeof(float)) == MI_OK &&
          mi_dspu_malloc(reinterpret_cast<void **>(&dout), x.size() * sizeof(float)) == MI_OK, "device rows");
    CHECK(mi_dspu_copy_h2d(din, x.data(), x.size() * sizeof(float), NULL) == MI_OK, "h2d");
    // block 0 on host rows, block 1 on device rows, blocks 2.. as one run
    CHECK(ea.process_host(&ya[0], &x[0], n, n), "process_host");
    CHECK(ea.process(dout + N * n, din + N * n, n, n), "process");
    float *po[blocks]; const float *pi[blocks];
    for (size_t b = 2; b < blocks; ++b) { po[b - 2] = dout + b * N * n; pi[b - 2] = din + b * N * n; }
    CHECK(ea.process_blocks(po, pi, blocks - 2, n, n), "process_blocks");
    CHECK(mi_dspu_copy_d2h(&ya[N * n], dout + N * n, (blocks - 1) * N * n * sizeof(float), NULL) == MI_OK && mi_dspu_stream_synchronize(NULL) == MI_OK, "d2h");
    size_t bad = 0;
    for (size_t i = 0; i < x.size(); ++i)
        bad += (ya[i] != ye[i]);
    CHECK(bad == 0, "%zu samples differ from the separate Equalizer objects", bad);
    mi_dspu_free(din); mi_dspu_free(dout);
}

static void convolver_array_equals_n_convolvers()
{
    printf("ConvolverArray against separate Convolver objects\n");
    const size_t N = 4, TAPS = 3000, RANK = 10, n = 512, blocks = 5;
    std::vector<float> irs(N * TAPS), x(blocks * N * n), ya(blocks * N * n), yc(blocks * N * n);
    unsigned seed = 4242;
    for (size_t i = 0; i < irs.size(); ++i) { seed = seed * 1664525u + 1013904223u; irs[i] = (float(seed >> 8) / 8388608.0f - 1.0f) * 0.1f; }
    for (size_t i = 0; i < x.size(); ++i) { seed = seed * 1664525u + 1013904223u; x[i] = (float(seed >> 8) / 8388608.0f - 1.0f) * 0.25f; }
    const size_t counts[N] = { TAPS, 700, TAPS - 1, 1 };
    dspu::ConvolverArray ca;
    CHECK(ca.init(N, irs.data(), TAPS, TAPS, RANK, 0.0f, counts) && ca.size() == N, "ConvolverArray::init");
    std::vector<dspu::Convolver> cv(N);
    for (size_t c = 0; c < N; ++c)
        CHECK(cv[c].init(&irs[c * TAPS], counts[c], RANK, 0.0f), "Convolver::init");
    CHECK(ca.rank() == cv[0].rank() && ca.data_size() == TAPS, "rank / data_size");
    for (size_t b = 0; b < blocks; ++b)
        for (size_t c = 0; c < N; ++c)
            cv[c].process(&yc[(b * N + c) * n], &x[(b * N + c) * n], n);
    float *din = NULL, *dout = NULL;
    CHECK(mi_dspu_malloc(reinterpret_cast<void **>(&din), N * n * sizeof(float)) == MI_OK &&
          mi_dspu_malloc(reinterpret_cast<void **>(&dout), N * n * sizeof(float)) == MI_OK, "device rows");
    for (size_t b = 0; b < blocks; ++b)
    {
        if (b & 1)
        {
            CHECK(mi_dspu_copy_h2d(din, &x[b * N * n], N * n * sizeof(float), NULL) == MI_OK, "h2d");
            CHECK(ca.process(dout, din, n, n), "process");
            CHECK(mi_dspu_copy_d2h(&ya[b * N * n], dout, N * n * sizeof(float), NULL) == MI_OK && mi_dspu_stream_synchronize(NULL) == MI_OK, "d2h");
        }
        else
            CHECK(ca.process_host(&ya[b * N * n], &x[b * N * n], n, n), "process_host");
    }
    // (a one-row bank and a row of a four-row bank take the same kernels except where the launch geometry depends on the
    // channel count: compared to the last few bits)
    double worst = 0.0, peak = 0.0;
    for (size_t i = 0; i < x.size(); ++i) { worst = std::max(worst, double(std::fabs(ya[i] - yc[i]))); peak = std::max(peak, double(std::fabs(yc[i]))); }
    CHECK(worst <= 1e-6 * peak, "ConvolverArray differs from the separate objects by %.3g of the peak", worst / peak);
    mi_dspu_free(din); mi_dspu_free(dout);
    // a run of whole frames in one call (batches of frames): the samples of the calls one by one, bit for bit
    {
        const size_t K = 7;
        std::vector<float> xb(K * N * n), y1(K * N * n), y2(K * N * n);
        for (size_t i = 0; i < xb.size(); ++i) { seed = seed * 1664525u + 1013904223u; xb[i] = (float(seed >> 8) / 8388608.0f - 1.0f) * 0.25f; }
        float *bi = NULL, *bo = NULL;
        CHECK(mi_dspu_malloc(reinterpret_cast<void **>(&bi), xb.size() * sizeof(float)) == MI_OK &&
              mi_dspu_malloc(reinterpret_cast<void **>(&bo), xb.size() * sizeof(float)) == MI_OK, "device rows");
        CHECK(mi_dspu_copy_h2d(bi, xb.data(), xb.size() * sizeof(float), NULL) == MI_OK, "h2d");
        dspu::ConvolverArray cb, cc;
        CHECK(cb.init(N, irs.data(), TAPS, TAPS, RANK, 0.0f, counts) && cc.init(N, irs.data(), TAPS, TAPS, RANK, 0.0f, counts), "ConvolverArray::init");
        float *po[K]; const float *pi[K];
        for (size_t k = 0; k < K; ++k) { po[k] = bo + k * N * n; pi[k] = bi + k * N * n; }
        CHECK(cb.process_blocks(po, pi, K, n, n), "ConvolverArray::process_blocks");
        CHECK(mi_dspu_copy_d2h(y1.data(), bo, y1.size() * sizeof(float), NULL) == MI_OK && mi_dspu_stream_synchronize(NULL) == MI_OK, "d2h");
        for (size_t k = 0; k < K; ++k)
            CHECK(cc.process(bo + k * N * n, bi + k * N * n, n, n), "process");
        CHECK(mi_dspu_copy_d2h(y2.data(), bo, y2.size() * sizeof(float), NULL) == MI_OK && mi_dspu_stream_synchronize(NULL) == MI_OK, "d2h");
        size_t bad = 0;
        for (size_t i = 0; i < y1.size(); ++i)
            bad += (y1[i] != y2[i]);
        CHECK(bad == 0, "process_blocks: %zu samples differ from frame-by-frame calls", bad);
        mi_dspu_free(bi); mi_dspu_free(bo);
    }
}

// ---- binary layout of the drop-in classes ---------------------------------------------------------------------------
// Object sizes and member offsets of the reference headers (lsp-dsp-units 1.0.36, LP64; sizes of SURVEY.md 0.4 plus the
// member lists of filters/FilterBank.h:39-46, filters/Filter.h:57-65, filters/Equalizer.h:59-78, util/Convolver.h:38-56,
// util/Delay.h:38-42, util/RingBuffer.h:38-40, util/SpectralProcessor.h:47-62): a caller compiled against the
// reference's headers embeds these objects by value and runs the reference's inline members on them.
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Winvalid-offsetof"
namespace layout
{
    struct FilterBankP : dspu::FilterBank { static void check(); };
    struct FilterP : dspu::Filter { static void check(); };
    struct EqualizerP : dspu::Equalizer { static void check(); };
    struct DelayP : dspu::Delay { static void check(); };
    struct RingBufferP : dspu::RingBuffer { static void check(); };
    struct SpectralProcessorP : dspu::SpectralProcessor { static void check(); };

    static_assert(sizeof(dspu::FilterBank) == 56 && sizeof(dspu::Filter) == 88 && sizeof(dspu::Equalizer) == 160 &&
                  sizeof(dspu::Convolver) == 144, "object sizes of SURVEY.md 0.4");
    static_assert(sizeof(dspu::Delay) == 24 && sizeof(dspu::RingBuffer) == 16 && sizeof(dspu::SpectralProcessor) == 104,
                  "object sizes of the reference headers");
    static_assert(sizeof(dspu::filter_params_t) == 24 && sizeof(dsp::biquad_x1_t) == 32 && sizeof(dsp::f_cascade_t) == 32 &&
                  sizeof(dsp::biquad_t) == 256, "plain structs of the API");

    void FilterBankP::check()
    {
        static_assert(offsetof(FilterBankP, vFilters) == 0 && offsetof(FilterBankP, vChains) == 8 && offsetof(FilterBankP, nItems) == 16 &&
                      offsetof(FilterBankP, nMaxItems) == 24 && offsetof(FilterBankP, nLastItems) == 32 &&
                      offsetof(FilterBankP, vBackup) == 40 && offsetof(FilterBankP, vData) == 48, "FilterBank members");
    }
    void FilterP::check()
    {
        static_assert(offsetof(FilterP, pBank) == 0 && offsetof(FilterP, sParams) == 8 && offsetof(FilterP, nSampleRate) == 32 &&
                      offsetof(FilterP, nMode) == 40 && offsetof(FilterP, nItems) == 48 && offsetof(FilterP, vItems) == 56 &&
                      offsetof(FilterP, vData) == 64 && offsetof(FilterP, nFlags) == 72 && offsetof(FilterP, nLatency) == 80, "Filter members");
    }
    void EqualizerP::check()
    {
        static_assert(offsetof(EqualizerP, sBank) == 0 && offsetof(EqualizerP, vFilters) == 56 && offsetof(EqualizerP, nFilters) == 64 &&
                      offsetof(EqualizerP, nSampleRate) == 68 && offsetof(EqualizerP, nActualSampleRate) == 72 &&
                      offsetof(EqualizerP, nFirSize) == 76 && offsetof(EqualizerP, nFirRank) == 80 && offsetof(EqualizerP, nLatency) == 84 &&
                      offsetof(EqualizerP, nBufSize) == 88 && offsetof(EqualizerP, nMode) == 92 && offsetof(EqualizerP, vInBuffer) == 96 &&
                      offsetof(EqualizerP, vTemp) == 136 && offsetof(EqualizerP, nFlags) == 144 && offsetof(EqualizerP, pData) == 152,
                      "Equalizer members");
    }
    void DelayP::check()
    {
        static_assert(offsetof(DelayP, pBuffer) == 0 && offsetof(DelayP, nHead) == 8 && offsetof(DelayP, nTail) == 12 &&
                      offsetof(DelayP, nDelay) == 16 && offsetof(DelayP, nSize) == 20, "Delay members");
    }
    void RingBufferP::check()
    {
        static_assert(offsetof(RingBufferP, pData) == 0 && offsetof(RingBufferP, nCapacity) == 8 && offsetof(RingBufferP, nHead) == 12,
                      "RingBuffer members");
    }
    void SpectralProcessorP::check()
    {
        static_assert(offsetof(SpectralProcessorP, nRank) == 0 && offsetof(SpectralProcessorP, nMaxRank) == 8 &&
                      offsetof(SpectralProcessorP, fPhase) == 16 && offsetof(SpectralProcessorP, pWnd) == 24 &&
                      offsetof(SpectralProcessorP, nOffset) == 56 && offsetof(SpectralProcessorP, pData) == 64 &&
                      offsetof(SpectralProcessorP, bUpdate) == 72 && offsetof(SpectralProcessorP, pFunc) == 80 &&
                      offsetof(SpectralProcessorP, pSubject) == 96, "SpectralProcessor members");
    }
}
#pragma GCC diagnostic pop

template <class T> static T *raw_object()                   // zeroed raw memory, as a host that embeds the object would hold it
{
    void *p = calloc(1, sizeof(T));
    return static_cast<T *>(p);
}

// construct() on calloc()ed memory (no constructor has run), the object used through the reference's inline members,
// destroy(), memory released with free(): what `construct()` is there for (filters/Filter.h:97-100).
static void raw_memory_objects()
{
    printf("construct() on raw memory + inline members\n");
    dspu::clear_last_status();

    dspu::FilterBank *fb = raw_object<dspu::FilterBank>();
    fb->construct();
    CHECK(fb->size() == 0 && fb->max_chains() == 0, "constructed bank is empty");
    CHECK(fb->init(4), "bank init");
    fb->begin();
    dsp::biquad_x1_t *c = fb->add_chain();
    CHECK(c != NULL && fb->size() == 1 && fb->max_chains() == 4, "add_chain / size / max_chains");
    c->b0 = 0.5f; c->b1 = 0.0f; c->b2 = 0.0f; c->a1 = 0.0f; c->a2 = 0.0f; c->p0 = c->p1 = c->p2 = 0.0f;
    fb->end(true);
    float x[8] = { 1, 2, 3, 4, 5, 6, 7, 8 }, y[8];
    fb->process(y, x, 8);
    CHECK(y[0] == 0.5f && y[7] == 4.0f, "one-section bank scales by b0");
    fb->begin();
    CHECK(fb->size() == 0, "begin() forgets the chains");
    fb->destroy();
    free(fb);

    {   // MultiSpectralProcessor on raw memory: inline getters read the members
        dspu::MultiSpectralProcessor *mp = raw_object<dspu::MultiSpectralProcessor>();
        mp->construct();
        CHECK(mp->get_rank() == 0 && mp->needs_update() && mp->phase() == 0.0f, "constructed multi-processor");
        CHECK(mp->init(2, 9), "multi-processor init");
        CHECK(mp->get_rank() == 9 && mp->latency() == 512 && mp->frame_size() == 256, "rank / latency / frame_size");
        mp->set_rank(8);
        CHECK(mp->get_rank() == 8 && mp->latency() == 256 && mp->needs_update(), "set_rank");
        std::vector<float> in0(2048), out0(2048, -1.0f);
        for (size_t i = 0; i < in0.size(); ++i)
            in0[i] = sinf(0.05f * float(i));
        CHECK(mp->bind(0, out0.data(), in0.data()) == STATUS_OK && mp->bind(5, NULL, NULL) == STATUS_INVALID_VALUE, "bind");
        mp->process(2048);
        CHECK(!mp->needs_update(), "process() applies the settings");
        float worst = 0.0f;
        for (size_t i = 256; i < 2048; ++i)
            worst = fmaxf(worst, fabsf(out0[i] - in0[i - 256]));
        CHECK(worst <= 1e-5f, "unbound handler: the input delayed by the latency (%g)", worst);
        mp->destroy();
        free(mp);
    }

    {   // SpectralSplitter on raw memory: the inline getters read the members the out-of-line calls keep current
        dspu::SpectralSplitter *ss = raw_object<dspu::SpectralSplitter>();
        ss->construct();
        CHECK(ss->handlers() == 0 && ss->bindings() == 0 && ss->needs_update() && ss->rank() == 0 && ss->phase() == 0.0f, "constructed splitter");
        CHECK(ss->init(4, 2) == STATUS_INVALID_VALUE && ss->init(10, 2) == STATUS_OK, "splitter init");
        CHECK(ss->handlers() == 2 && ss->max_rank() == 10 && ss->rank() == 10 && ss->bindings() == 0, "init members");
        ss->set_rank(8); ss->set_chunk_rank(6); ss->set_phase(0.25f);
        CHECK(ss->rank() == 8 && ss->phase() == 0.25f && ss->needs_update(), "setters mark the settings dirty");
        ss->update_settings();
        CHECK(!ss->needs_update() && ss->chunk_rank() == 6 && ss->latency() == 64, "update_settings(): chunk rank %d latency %d",
              int(ss->chunk_rank()), int(ss->latency()));
        struct sink_t { std::vector<float> out; } sink;
        sink.out.assign(1024, -1.0f);
        auto fn = [](void *, void *, float *out, const float *in, size_t rank) { memcpy(out, in, sizeof(float) * (size_t(2) << rank)); };
        auto snk = [](void *obj, void *, const float *samples, size_t first, size_t count)
        { memcpy(static_cast<sink_t *>(obj)->out.data() + first, samples, count * sizeof(float)); };
        CHECK(ss->bind(1, &sink, NULL, fn, snk) == STATUS_OK && ss->bindings() == 1 && ss->bound(1) && !ss->bound(0), "bind");
        CHECK(ss->bind(2, &sink, NULL, fn, snk) == STATUS_OVERFLOW && ss->bind(0, NULL, NULL, NULL, NULL) == STATUS_INVALID_VALUE, "bind errors");
        std::vector<float> x(1024);
        for (size_t i = 0; i < x.size(); ++i)
            x[i] = sinf(0.03f * float(i));
        ss->process(x.data(), x.size());
        float worst = 0.0f;
        for (size_t i = 64; i < 1024; ++i)
            worst = fmaxf(worst, fabsf(sink.out[i] - x[i - 64]));
        CHECK(worst <= 1e-5f, "identity handler: the input delayed by the latency (%g)", worst);
        CHECK(ss->unbind(1) == STATUS_OK && ss->unbind(1) == STATUS_NOT_BOUND && ss->bindings() == 0, "unbind");
        ss->destroy();
        CHECK(ss->handlers() == 0, "destroyed splitter");
        free(ss);
    }

    {   // Crossover on raw memory: counts, sample rate and the reconfiguration flag are members the inline getters read
        dspu::Crossover *xo = raw_object<dspu::Crossover>();
        xo->construct();
        CHECK(xo->num_bands() == 1 && xo->num_splits() == 0 && xo->max_buffer_size() == 0 && xo->get_sample_rate() == 48000 &&
              xo->needs_reconfiguration(), "constructed crossover");
        CHECK(!xo->init(0, 256) && xo->init(3, 256), "crossover init");
        CHECK(xo->num_bands() == 3 && xo->num_splits() == 2 && xo->max_buffer_size() == 256 && xo->needs_reconfiguration(), "init members");
        xo->set_sample_rate(44100);
        xo->set_slope(0, dspu::CROSS_SLOPE_LR4); xo->set_frequency(0, 500.0f);
        xo->set_slope(1, dspu::CROSS_SLOPE_LR4); xo->set_frequency(1, 4000.0f);
        CHECK(xo->get_sample_rate() == 44100 && xo->get_slope(1) == dspu::CROSS_SLOPE_LR4 && xo->get_frequency(0) == 500.0f &&
              xo->get_slope(2) == -1, "split records");
        xo->reconfigure();
        CHECK(!xo->needs_reconfiguration() && xo->band_active(1) && xo->get_band_start(1) == 500.0f && xo->get_band_end(1) == 4000.0f, "reconfigure()");
        xo->set_gain(1, 0.5f);
        CHECK(xo->needs_reconfiguration() && xo->get_gain(1) == 0.5f, "set_gain marks the crossover dirty");
        xo->set_gain(1, 1.0f);
        struct got_t { std::vector<float> y[3]; } got;
        for (auto &v : got.y) v.assign(2000, 0.0f);
        auto take = [](void *obj, void *, size_t band, const float *data, size_t first, size_t count)
        { memcpy(static_cast<got_t *>(obj)->y[band].data() + first, data, count * sizeof(float)); };
        for (size_t b = 0; b < 3; ++b)
            CHECK(xo->set_handler(b, take, &got, NULL), "set_handler");
        std::vector<float> x(2000, 0.0f);
        x[0] = 1.0f;
        xo->process(x.data(), x.size());
        CHECK(!xo->needs_reconfiguration(), "process() reconfigures");
        // Linkwitz-Riley bands add up to an all-pass: the energy of the sum of the bands' impulse responses is 1
        double energy = 0.0;
        for (size_t i = 0; i < 2000; ++i)
        {
            const double v = double(got.y[0][i]) + got.y[1][i] + got.y[2][i];
            energy += v * v;
        }
        CHECK(fabs(energy - 1.0) < 1e-3, "bands add up to an all-pass (energy %g)", energy);
        xo->destroy();
        CHECK(xo->num_bands() == 1 && xo->needs_reconfiguration(), "destroyed crossover");
        free(xo);
    }

    {   // FFTCrossover on raw memory: the embedded splitter answers the inline members
        dspu::FFTCrossover *fx = raw_object<dspu::FFTCrossover>();
        fx->construct();
        CHECK(fx->bands() == 0 && fx->sample_rate() == 0 && fx->rank() == 0 && !fx->needs_update(), "constructed FFT crossover");
        CHECK(fx->init(10, 2) == STATUS_OK && fx->bands() == 2 && fx->rank() == 10, "FFT crossover init");
        fx->set_sample_rate(48000); fx->set_rank(9); fx->set_phase(0.5f);
        CHECK(fx->sample_rate() == 48000 && fx->rank() == 9 && fx->phase() == 0.5f, "inline getters follow the setters");
        struct got_t { std::vector<float> y[2]; } got;
        got.y[0].assign(4096, 0.0f); got.y[1].assign(4096, 0.0f);
        auto take = [](void *obj, void *, size_t band, const float *data, size_t first, size_t count)
        { memcpy(static_cast<got_t *>(obj)->y[band].data() + first, data, count * sizeof(float)); };
        fx->set_lpf(0, 1000.0f, -48.0f, true);
        fx->set_hpf(1, 1000.0f, -48.0f, true);
        CHECK(fx->set_handler(0, take, &got, NULL) && fx->set_handler(1, take, &got, NULL) && !fx->set_handler(2, take, &got, NULL), "handlers");
        fx->enable_band(0); fx->enable_band(1);
        CHECK(fx->band_enabled(0) && fx->lpf_enabled(0) && !fx->hpf_enabled(0) && fx->lpf_frequency(0) == 1000.0f && fx->gain(5) == -1.0f, "band records");
        fx->update_settings();
        CHECK(!fx->needs_update() && fx->latency() == 512, "update_settings(): latency %d", int(fx->latency()));
        std::vector<float> x(4096);
        uint32_t seed = 12345;
        for (float &v : x)
        {
            seed = seed * 1664525u + 1013904223u;
            v = float(int32_t(seed >> 8) - (1 << 23)) / float(1 << 23);
        }
        fx->process(x.data(), x.size());
        float worst = 0.0f;
        for (size_t i = 512; i < 4096; ++i)
            worst = fmaxf(worst, fabsf(got.y[0][i] + got.y[1][i] - x[i - 512]));
        CHECK(worst <= 1e-4f, "complementary bands add up to the delayed input (%g)", worst);
        fx->disable_band(1);
        CHECK(!fx->band_enabled(1), "disable_band");
        // frames above 2^14 samples (the reference's init() has no upper limit): rank 15, the same complementary pair
        fx->enable_band(1);
        fx->set_rank(15);
        CHECK(fx->rank() == 10, "set_rank above max_rank is clamped to it (FFTCrossover.cpp:433)");
        fx->destroy();
        CHECK(fx->init(15, 2) == STATUS_OK && fx->rank() == 15, "FFT crossover init at rank 15");
        fx->set_sample_rate(48000);
        fx->set_lpf(0, 1000.0f, -48.0f, true);
        fx->set_hpf(1, 1000.0f, -48.0f, true);
        const size_t big_n = 3 * 32768 + 1000;
        got.y[0].assign(big_n, 0.0f); got.y[1].assign(big_n, 0.0f);
        CHECK(fx->set_handler(0, take, &got, NULL) && fx->set_handler(1, take, &got, NULL), "handlers at rank 15");
        fx->enable_band(0); fx->enable_band(1);
        std::vector<float> xb(big_n);
        for (float &v : xb)
        {
            seed = seed * 1664525u + 1013904223u;
            v = float(int32_t(seed >> 8) - (1 << 23)) / float(1 << 23);
        }
        fx->process(xb.data(), xb.size());
        CHECK(fx->latency() == 32768, "latency at rank 15: %d", int(fx->latency()));
        worst = 0.0f;
        for (size_t i = 32768; i < big_n; ++i)
            worst = fmaxf(worst, fabsf(got.y[0][i] + got.y[1][i] - xb[i - 32768]));
        CHECK(worst <= 1e-4f, "rank 15: complementary bands add up to the delayed input (%g)", worst);
        fx->clear();
        fx->destroy();
        CHECK(fx->bands() == 0, "destroyed FFT crossover");
        free(fx);
    }

    {   // Analyzer: inline setters only write members; the object follows them at its next process()
        dspu::Analyzer *an = raw_object<dspu::Analyzer>();
        an->construct();
        CHECK(an->get_channels() == 0 && an->get_rank() == 0 && !an->needs_reconfiguration() && an->activity(), "constructed analyzer");
        CHECK(an->init(2, 9, 48000, 10.0f, 64), "analyzer init");
        CHECK(an->get_channels() == 2 && an->get_rank() == 9 && an->needs_reconfiguration(), "init members");
        an->set_sample_rate(48000); an->set_rate(93.75f); an->set_reactivity(0.0001f);
        std::vector<float> a0(512), a1(512, 0.0f);
        for (int i = 0; i < 512; ++i)
            a0[i] = sinf(2.0f * float(M_PI) * 32.0f * i / 512.0f);
        const float *ins[2] = { a0.data(), a1.data() };
        for (int k = 0; k < 4; ++k)
            an->process(ins, 512);
        CHECK(!an->needs_reconfiguration(), "process() reconfigures");
        const uint32_t bin = 32;
        const float lit = an->get_level(0, bin);
        CHECK(lit > 0.0f, "a sine lights its bin: %g", lit);
        an->set_activity(false);                            // inline: bActive = false
        for (int k = 0; k < 3; ++k)
            an->process(ins, 512);
        CHECK(an->get_level(0, bin) == 0.0f, "an inactive analyzer publishes zeros: %g", an->get_level(0, bin));
        an->set_activity(true);
        for (int k = 0; k < 3; ++k)
            an->process(ins, 512);
        CHECK(an->get_level(0, bin) > 0.0f, "active again");
        an->reset();                                        // inline: nReconfigure |= R_ANALYSIS
        CHECK(an->needs_reconfiguration(), "reset() asks for a reconfigure");
        an->reconfigure();
        CHECK(an->get_level(0, bin) == 0.0f, "reset() clears the spectra: %g", an->get_level(0, bin));
        CHECK(an->channel_active(1) && an->enable_channel(1, false) && !an->channel_active(1) && an->channel_delay(1) == 0, "channel records");
        an->destroy();
        free(an);
    }

    dspu::Filter *f = raw_object<dspu::Filter>();
    f->construct();
    CHECK(f->inactive() && f->latency() == 0, "constructed filter is inactive");
    CHECK(f->init(NULL), "filter init");
    dspu::filter_params_t fp;
    fp.nType = dspu::FLT_BT_RLC_BELL; fp.fFreq = 1000.0f; fp.fFreq2 = 1000.0f; fp.fGain = 2.0f; fp.nSlope = 1; fp.fQuality = 1.0f;
    f->update(48000, &fp);
    CHECK(f->inactive(), "update() parks the filter in FM_BYPASS (Filter.cpp:150)");
    f->rebuild();
    CHECK(f->active() && !f->inactive(), "active after rebuild");
    f->clear();                                             // inline: nFlags |= FF_CLEAR, picked up by the next process()
    std::vector<float> ir(64, 0.0f);
    ir[0] = 1.0f;
    f->process(ir.data(), ir.data(), ir.size());
    CHECK(ir[0] > 1.0f && ir[0] < 2.0f, "bell +6 dB impulse head %g", ir[0]);
    f->destroy();
    free(f);

    dspu::Delay *d = raw_object<dspu::Delay>();
    d->construct();
    CHECK(d->get_delay() == 0, "constructed delay");
    CHECK(d->init(100), "delay init");
    d->set_delay(7);
    CHECK(d->get_delay() == 7 && d->delay() == 7, "inline get_delay()");
    float z[16];
    d->process(z, x, 8);
    d->process(&z[8], x, 8);
    CHECK(z[6] == 0.0f && z[7] == 1.0f && z[14] == 8.0f && z[15] == 1.0f, "delayed by 7");
    d->destroy();
    free(d);

    dspu::RingBuffer *rb = raw_object<dspu::RingBuffer>();
    rb->construct();
    CHECK(rb->size() == 0 && rb->data() == NULL, "constructed ring");
    CHECK(rb->init(8, 0.25f), "ring init");
    CHECK(rb->size() == 8 && rb->data() != NULL && rb->data()[0] == 0.25f && rb->data()[7] == 0.25f, "data() shows the fill value");
    rb->append(x, 3);
    CHECK(rb->head_position() == 3 && rb->data()[0] == 1.0f && rb->data()[2] == 3.0f && rb->data()[3] == 0.25f, "data() shows appended samples");
    rb->data()[2] = 9.0f;                                   // a host write into the raw storage is what get() returns
    CHECK(rb->get(0) == 9.0f && rb->get(1) == 2.0f, "host write through data()");
    rb->destroy();
    free(rb);

    dspu::Convolver *cv = raw_object<dspu::Convolver>();
    cv->construct();
    CHECK(cv->data_size() == 0 && cv->rank() == 0, "constructed convolver");
    std::vector<float> taps(300, 0.0f);
    taps[0] = 1.0f; taps[299] = 0.5f;
    CHECK(cv->init(taps.data(), taps.size(), 9, 0.0f), "convolver init");
    CHECK(cv->data_size() == 300 && cv->rank() == 9, "inline data_size() / rank()");
    std::vector<float> in(600, 0.0f), out(600, 0.0f);
    in[0] = 1.0f;
    cv->process(out.data(), in.data(), 600);
    CHECK(fabsf(out[0] - 1.0f) < 1e-5f && fabsf(out[299] - 0.5f) < 1e-5f && fabsf(out[300]) < 1e-5f, "convolver output");
    cv->destroy();
    free(cv);

    dspu::SpectralProcessor *sp = raw_object<dspu::SpectralProcessor>();
    sp->construct();
    CHECK(sp->needs_update() && sp->get_rank() == 0, "constructed spectral processor");
    CHECK(sp->init(8), "spectral init");
    CHECK(sp->get_rank() == 8 && sp->latency() == 256 && sp->phase() == 0.0f, "inline rank / latency / phase");
    sp->set_rank(7);
    sp->set_phase(0.5f);
    CHECK(sp->get_rank() == 7 && sp->latency() == 128 && sp->phase() == 0.5f && sp->needs_update(), "setters reach the inline getters");
    sp->update_settings();
    CHECK(!sp->needs_update(), "update_settings()");
    sp->destroy();
    free(sp);

    dspu::Equalizer *eq = raw_object<dspu::Equalizer>();
    eq->construct();
    CHECK(eq->get_mode() == dspu::EQM_BYPASS && eq->fir_rank() == 0 && !eq->filter_active(0), "constructed equalizer");
    CHECK(eq->init(3, 9), "equalizer init");
    eq->set_sample_rate(44100);
    CHECK(eq->fir_rank() == 9 && eq->fir_ir_size() == 1024 && eq->max_latency() == 768 && eq->actual_sample_rate() == 44100, "inline sizes");
    eq->set_params(1, &fp);
    eq->set_mode(dspu::EQM_FIR);
    CHECK(eq->mode() == dspu::EQM_FIR && eq->filter_inactive(1), "mode / stale filter");
    CHECK(eq->get_latency() == 768 && eq->filter_active(1) && eq->filter_inactive(0), "latency 1.5 N and filter modes after reconfigure");
    eq->destroy();
    free(eq);

    // DynamicFilters: a bell whose gain follows a per-sample vector; gain 1 is transparent, a constant gain is the static bell
    static_assert(sizeof(dspu::DynamicFilters) == 64, "DynamicFilters object size of the reference header");
    static_assert(sizeof(dspu::LoudnessMeter) == 112 && sizeof(dspu::ILUFSMeter) == 104, "meter object sizes of the reference headers");
    static_assert(sizeof(dspu::SpectralSplitter) == 128, "SpectralSplitter object size of the reference header");
    static_assert(sizeof(dspu::Crossover) == 72, "Crossover object size of the reference header (5 x u32, 6 pointers)");
    static_assert(sizeof(dspu::FFTCrossover) == 152, "FFTCrossover object size of the reference header (splitter by value + 3 words)");
    static_assert(sizeof(dspu::MultiSpectralProcessor) == 80, "MultiSpectralProcessor object size of the reference header");
    static_assert(sizeof(dspu::Analyzer) == 128, "Analyzer object size of the reference header (14 x u32, 5 x f32, bool, 6 pointers)");
    dspu::DynamicFilters *df = raw_object<dspu::DynamicFilters>();
    df->construct();
    CHECK(df->filter_inactive(0) && !df->filter_active(0), "constructed dynamic filters");
    CHECK(df->init(2) == STATUS_OK, "dynamic filters init");
    df->set_sample_rate(48000);
    dspu::filter_params_t dp;
    dp.nType = dspu::FLT_BT_RLC_BELL; dp.fFreq = 1000.0f; dp.fFreq2 = 1000.0f; dp.fGain = 1.0f; dp.nSlope = 2; dp.fQuality = 0.5f;
    CHECK(df->set_params(1, &dp) && !df->set_params(2, &dp), "set_params range");
    CHECK(df->filter_inactive(1) && df->set_filter_active(1, false) && df->filter_active(1), "inline activity (activates whatever is asked)");
    std::vector<float> sx(3000), sg(3000, 1.0f), sy(3000);
    for (size_t i = 0; i < sx.size(); ++i) sx[i] = sinf(0.13f * float(i)) + 0.3f * sinf(1.7f * float(i));
    df->process(1, sy.data(), sx.data(), sg.data(), sx.size());
    float worst = 0.0f;
    for (size_t i = 0; i < sx.size(); ++i) worst = std::max(worst, fabsf(sy[i] - sx[i]));
    CHECK(worst < 2e-5f, "gain 1: the bell is transparent (%g)", worst);
    df->process(0, sy.data(), sx.data(), sg.data(), sx.size());
    CHECK(memcmp(sy.data(), sx.data(), sx.size() * sizeof(float)) == 0, "an inactive filter copies");
    float fr[3] = { 100.0f, 1000.0f, 10000.0f }, cre[3], cim[3];
    CHECK(df->freq_chart(1, cre, cim, fr, 2.0f, 3) && fabsf(hypotf(cre[1], cim[1]) - 2.0f) < 0.02f && fabsf(hypotf(cre[0], cim[0]) - 1.0f) < 0.05f,
          "freq_chart: +6 dB at the centre (%g)", hypotf(cre[1], cim[1]));
    df->destroy();
    free(df);

    CHECK(dspu::last_status() == MI_OK, "a device call failed on the way: %d (%s)", dspu::last_status(), mi_dspu_last_error());
}

// dump(): every unit writes the reference's keys (tests/golden/dump_keys.json is checked statically by
// tests/test_cpp_classes.py); here the objects are live and the visitor checks what a dump is at run time -- balanced
// objects and arrays, the names in order, the FilterBank's packed groups equal to the chains they were gathered from.
namespace
{
    struct recorder: public dspu::IStateDumper
    {
        std::vector<std::string> names;
        std::vector<std::pair<std::string, std::vector<float> > > vectors;
        int depth = 0, worst = 0, objects = 0, arrays = 0;
        void enter(const char *n)                   { if (n) names.push_back(n); ++depth; }
        void leave()                                { --depth; worst = std::min(worst, depth); }
        void begin_object(const char *n, const void *, size_t) override  { ++objects; enter(n); }
        void begin_object(const void *, size_t) override                 { ++objects; enter(nullptr); }
        void end_object() override                                        { leave(); }
        void begin_array(const char *n, const void *, size_t) override   { ++arrays; enter(n); }
        void begin_array(const void *, size_t) override                  { ++arrays; enter(nullptr); }
        void end_array() override                                         { leave(); }
        void write(const char *n, const void *) override        { names.push_back(n); }
        void write(const char *n, const char *) override        { names.push_back(n); }
        #define REC(T) void write(const char *n, T) override { names.push_back(n); }
        MI_DUMPER_TYPES(REC)
        #undef REC
        void writev(const char *n, const void * const *, size_t) override { names.push_back(n); }
        void writev(const char *n, const float *p, size_t c) override     { names.push_back(n); vectors.push_back(std::make_pair(std::string(n), std::vector<float>(p, p + c))); }
        bool has(const char *n) const               { return std::find(names.begin(), names.end(), n) != names.end(); }
    };

    template <class T>
    void dumped(const char *what, const T &unit, const char *first, const char *last, size_t at_least)
    {
        recorder r;
        unit.dump(&r);
        CHECK(r.depth == 0 && r.worst == 0, "%s: objects / arrays not balanced (%d, %d)", what, r.depth, r.worst);
        CHECK(r.names.size() >= at_least, "%s: %zu names, expected at least %zu", what, r.names.size(), at_least);
        CHECK(!r.names.empty() && r.names.front() == first && r.names.back() == last, "%s: runs from %s to %s",
              what, r.names.empty() ? "-" : r.names.front().c_str(), r.names.empty() ? "-" : r.names.back().c_str());
    }
}

static void state_dumps()
{
    printf("state dumps\n");
    {   // FilterBank.cpp:332-424: 11 chains = one group of 8, one of 2, one single
        dspu::FilterBank fb;
        CHECK(fb.init(16), "bank init");
        fb.begin();
        for (int i = 0; i < 11; ++i)
        {
            dsp::biquad_x1_t *c = fb.add_chain();
            c->b0 = 1.0f + i; c->b1 = 0.1f * i; c->b2 = 0.01f * i; c->a1 = -0.001f * i; c->a2 = 0.0001f * i;
            c->p0 = c->p1 = c->p2 = 0.0f;
        }
        fb.end(true);
        recorder r;
        fb.dump(&r);
        CHECK(r.depth == 0 && r.arrays == 2 && r.objects == 3 + 11, "bank: %d arrays, %d objects", r.arrays, r.objects);
        CHECK(r.vectors.size() == 5 + 6 && r.vectors[0].first == "b0" && r.vectors[0].second.size() == 8 && r.vectors[5].second.size() == 2 &&
              r.vectors[10].first == "p", "bank: packed groups");
        if (r.vectors.size() == 11)
        {
            for (int i = 0; i < 8; ++i)
                CHECK(r.vectors[0].second[i] == 1.0f + i && r.vectors[3].second[i] == -0.001f * i, "group of 8, chain %d", i);
            CHECK(r.vectors[5].second[0] == 9.0f && r.vectors[5].second[1] == 10.0f, "group of 2");
        }
        CHECK(r.names.front() == "vFilters" && r.names.back() == "vData" && r.has("vChains") && r.has("nLastItems"), "bank: names");
        fb.destroy();
    }
    {
        dspu::Filter f;
        CHECK(f.init(NULL), "filter init");
        dspu::filter_params_t fp;
        fp.nType = dspu::FLT_BT_RLC_BELL; fp.fFreq = 1000.0f; fp.fFreq2 = 1000.0f; fp.fGain = 2.0f; fp.nSlope = 2; fp.fQuality = 0.5f;
        f.update(48000, &fp);
        float y[8] = { 1, 0, 0, 0, 0, 0, 0, 0 };
        f.process(y, y, 8);
        recorder r;
        f.dump(&r);
        CHECK(r.depth == 0 && r.names.front() == "pBank" && r.names.back() == "nLatency" && r.has("vItems") && r.has("t") && r.has("vChains"),
              "filter: an own bank goes out as an object, the cascades with their polynomials");
        f.destroy();
    }
    {
        dspu::Equalizer eq;
        CHECK(eq.init(2, 9), "equalizer init");
        eq.set_sample_rate(48000);
        dumped("equalizer", eq, "sBank", "pData", 17 + 2 * 10);
        eq.destroy();
        dspu::DynamicFilters df;
        CHECK(df.init(2) == STATUS_OK, "dynamic filters init");
        dumped("dynamic filters", df, "vFilters", "bClearMem", 6 + 2 * 7);
        df.destroy();
        dspu::Convolver cv;
        const float taps[4] = { 1.0f, 0.5f, 0.25f, 0.125f };
        CHECK(cv.init(taps, 4, 9, 0.0f), "convolver init");
        dumped("convolver", cv, "pDataBuffer", "vData", 18);
        cv.destroy();
        dspu::SpectralProcessor sp;
        CHECK(sp.init(10), "spectral processor init");
        dumped("spectral processor", sp, "nRank", "pSubject", 13);
        sp.destroy();
        dspu::MultiSpectralProcessor mp;
        CHECK(mp.init(2, 10), "multi spectral processor init");
        dumped("multi spectral processor", mp, "nChannels", "pData", 13 + 2 * 5);
        mp.destroy();
        dspu::Crossover xo;
        CHECK(xo.init(3, 256), "crossover init");
        xo.set_sample_rate(48000);
        xo.reconfigure();
        dumped("crossover", xo, "nReconfigure", "pData", 11 + 3 * 10 + 2 * 6);
        xo.destroy();
        dspu::SpectralSplitter ss;
        CHECK(ss.init(10, 2) == STATUS_OK, "spectral splitter init");
        dumped("spectral splitter", ss, "nRank", "pData", 15 + 2 * 5);
        ss.destroy();
        dspu::FFTCrossover fx;
        CHECK(fx.init(10, 2) == STATUS_OK, "fft crossover init");
        dumped("fft crossover", fx, "sSplitter", "pData", 4 + 2 * 14 + 15);
        fx.destroy();
        dspu::LoudnessMeter lm;
        CHECK(lm.init(2, 400.0f) == STATUS_OK, "loudness meter init");
        dumped("loudness meter", lm, "vChannels", "pVarData", 16 + 2 * 12);
        lm.destroy();
        dspu::ILUFSMeter im;
        CHECK(im.init(2, 5.0f, 400.0f) == STATUS_OK, "ilufs meter init");
        dumped("ilufs meter", im, "vChannels", "pVarData", 21 + 2 * 7);
        im.destroy();
        dspu::Delay dl;
        CHECK(dl.init(100), "delay init");
        dumped("delay", dl, "pBuffer", "nSize", 5);
        dl.destroy();
        dspu::RingBuffer rb;
        CHECK(rb.init(16), "ring buffer init");
        dumped("ring buffer", rb, "pData", "nHead", 3);
        rb.destroy();
        dspu::Analyzer an;
        CHECK(an.init(2, 10, 48000, 10.0f), "analyzer init");
        dumped("analyzer", an, "nChannels", "vEnvelope", 26 + 2 * 7);
        an.destroy();
    }
}

int main(int argc, char **argv)
{
    if (argc > 1 && strcmp(argv[1], "--list") == 0)
    {
        puts("convolver.test_small convolver.test_large equalizer.FIR equalizer.FFT equalizer.SPM spectral_proc multi_spectral_proc crossover loudness_meter ilufs_meter spectral_splitter fft_crossover ringbuffer accessors readme_filter raw_memory_objects filter_array equalizer_array convolver_array state_dumps");
        return 0;
    }
    if (mi_dspu_device_count() <= 0)
    {
        puts("no HIP device: the lsp::dspu classes of this library have no CPU fallback");
        return 2;
    }
    convolver_small();
    convolver_large();
    equalizer_latency("FIR", dspu::EQM_FIR);
    equalizer_latency("FFT", dspu::EQM_FFT);
    equalizer_latency("SPM", dspu::EQM_SPM);
    spectral_proc_simple();
    spectral_proc_callback();
    multi_spectral_proc();
    crossover_bands_sum_to_allpass();
    loudness_meter_bs1770();
    ilufs_meter_mtest_flow();
    spectral_splitter_mtest_flow();
    fft_crossover_mtest_flow();
    ringbuffer();
    accessors();
    readme_filter();
    raw_memory_objects();
    filter_array_equals_n_filters();
    equalizer_array_equals_n_equalizers();
    convolver_array_equals_n_convolvers();
    state_dumps();
    CHECK(dspu::last_status() == MI_OK, "device status after the whole replay: %d (%s)", dspu::last_status(), mi_dspu_last_error());
    printf("%s (%d failure%s)\n", failures ? "FAILED" : "ALL PASSED", failures, failures == 1 ? "" : "s");
    return failures ? 1 : 0;
}
