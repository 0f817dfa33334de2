// The batched modes under the class API (lsp::dspu::EqualizerArray / ConvolverArray, this library's extensions) against the
// C-ABI banks they sit on: 256 objects, rows resident in device memory.
//   C4: 256 equalizers, 32 x FLT_BT_RLC_BELL, fir_rank 12, EQM_FIR, 4096-sample blocks (per block and as runs of 64 blocks)
//   C3: 256 convolvers, 65536 taps each, rank 13, 4096-sample frames
// Build: g++ -std=c++11 -O2 -I lsp-dsp-units_amd/include -I include tests/experiments/array_rate.cpp \
//        -o tests/experiments/array_rate -L lsp-dsp-units_amd -lmi_dspu -Wl,-rpath,$PWD/lsp-dsp-units_amd -Wl,-rpath,/opt/rocm/lib
#include <lsp-plug.in/dsp-units/filters/EqualizerArray.h>
#include <lsp-plug.in/dsp-units/util/ConvolverArray.h>
#include <mi_dspu.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <vector>

using namespace lsp;

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F> static double rate(F f, size_t warm, size_t steps)
{
    for (size_t i = 0; i < warm; ++i) f(i);
    mi_dspu_stream_synchronize(NULL);
    const double t0 = now();
    for (size_t i = 0; i < steps; ++i) f(i);
    mi_dspu_stream_synchronize(NULL);
    return (now() - t0) / steps;
}

int main()
{
    const size_t C = 256, n = 4096, ring = 8;
    if (mi_dspu_device_count() <= 0) { puts("no HIP device"); return 2; }
    float *din = NULL, *dout = NULL;
    mi_dspu_malloc(reinterpret_cast<void **>(&din), ring * C * n * sizeof(float));
    mi_dspu_malloc(reinterpret_cast<void **>(&dout), ring * C * n * sizeof(float));
    {
        std::vector<float> h(ring * C * n);
        unsigned seed = 99;
        for (float &v : h) { seed = seed * 1664525u + 1013904223u; v = (float(seed >> 8) / 8388608.0f - 1.0f) * 0.25f; }
        mi_dspu_copy_h2d(din, h.data(), h.size() * sizeof(float), NULL);
    }
    // ---- equalizers ------------------------------------------------------------------------------------------------
    const size_t NF = 32;
    mi_equalizer_bank_t *bank = NULL;
    mi_equalizer_bank_create(&bank, C, NF, 12);
    mi_equalizer_bank_set_mode(bank, MI_EQM_FIR);
    mi_equalizer_bank_set_sample_rate(bank, 48000);
    dspu::EqualizerArray ea;
    ea.init(C, NF, 12); ea.set_mode(dspu::EQM_FIR); ea.set_sample_rate(48000);
    for (size_t c = 0; c < C; ++c)
        for (size_t i = 0; i < NF; ++i)
        {
            dspu::filter_params_t fp;
            fp.nType = dspu::FLT_BT_RLC_BELL; fp.nSlope = 1; fp.fQuality = 2.0f;
            fp.fFreq = fp.fFreq2 = 20.0f * std::pow(1000.0f, float(i) / float(NF - 1));
            fp.fGain = std::pow(10.0f, (float((c * 7 + i * 13) % 25) - 12.0f) / 20.0f);
            mi_equalizer_bank_set_params(bank, uint32_t(c), uint32_t(i), reinterpret_cast<const mi_filter_params_t *>(&fp));
            ea.set_params(c, i, &fp);
        }
    uint32_t lat = 0;
    mi_equalizer_bank_get_latency(bank, &lat, NULL);
    (void)ea.get_latency();
    const double ms = double(C) * n / 1e6;
    const double tb = rate([&](size_t i) { mi_equalizer_bank_process(bank, dout + (i % ring) * C * n, din + (i % ring) * C * n, n, n, n, NULL); }, 20, 400);
    const double ta = rate([&](size_t i) { ea.process(dout + (i % ring) * C * n, din + (i % ring) * C * n, n, n); }, 20, 400);
    const size_t K = 64;
    float *po[K]; const float *pi[K];
    for (size_t k = 0; k < K; ++k) { po[k] = dout + (k % ring) * C * n; pi[k] = din + (k % ring) * C * n; }
    const double tbk = rate([&](size_t) { mi_equalizer_bank_process_blocks(bank, po, pi, K, n, n, n, NULL); }, 3, 20) / K;
    const double tak = rate([&](size_t) { ea.process_blocks(po, pi, K, n, n); }, 3, 20) / K;
    printf("C4: 256 equalizers (32 bells, FIR 2^12), 4096-sample blocks, resident rows\n");
    printf("  mi_equalizer_bank_process                  %8.2f us per block  %9.1f Msamples/s\n", tb * 1e6, ms / tb);
    printf("  dspu::EqualizerArray::process              %8.2f us per block  %9.1f Msamples/s   = %.3f of the bank\n", ta * 1e6, ms / ta, tb / ta);
    printf("  mi_equalizer_bank_process_blocks (64)      %8.2f us per block  %9.1f Msamples/s\n", tbk * 1e6, ms / tbk);
    printf("  dspu::EqualizerArray::process_blocks (64)  %8.2f us per block  %9.1f Msamples/s   = %.3f of the bank\n", tak * 1e6, ms / tak, tbk / tak);
    mi_equalizer_bank_destroy(bank);
    ea.destroy();
    // ---- convolvers ------------------------------------------------------------------------------------------------
    const size_t TAPS = 65536;
    std::vector<float> irs(C * TAPS);
    unsigned seed = 4;
    for (size_t c = 0; c < C; ++c)
        for (size_t k = 0; k < TAPS; ++k) { seed = seed * 1664525u + 1013904223u; irs[c * TAPS + k] = (float(seed >> 8) / 8388608.0f - 1.0f) * std::exp(-float(k) / 16384.0f); }
    mi_convolver_bank_t *cb = NULL;
    mi_convolver_bank_create(&cb, C, irs.data(), TAPS, NULL, TAPS, 13, 0.0f, NULL);
    dspu::ConvolverArray ca;
    ca.init(C, irs.data(), TAPS, TAPS, 13, 0.0f);
    const double tcb = rate([&](size_t i) { mi_convolver_bank_process(cb, dout + (i % ring) * C * n, din + (i % ring) * C * n, n, n, n, NULL); }, 20, 200);
    const double tca = rate([&](size_t i) { ca.process(dout + (i % ring) * C * n, din + (i % ring) * C * n, n, n); }, 20, 200);
    printf("C3: 256 convolvers (65536 taps, rank 13), 4096-sample frames, resident rows\n");
    printf("  mi_convolver_bank_process                  %8.2f us per frame  %9.1f Msamples/s\n", tcb * 1e6, ms / tcb);
    printf("  dspu::ConvolverArray::process              %8.2f us per frame  %9.1f Msamples/s   = %.3f of the bank\n", tca * 1e6, ms / tca, tcb / tca);
    mi_convolver_bank_destroy(cb);
    return 0;
}
