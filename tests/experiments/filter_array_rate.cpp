// The batched mode under the class API (lsp::dspu::FilterArray, this library's extension) against (a) the C-ABI bank it
// sits on and (b) what the reference's shape costs: 1024 separate dspu::Filter objects, one process() call each.
// C2: 1024 filters, FLT_BT_LRX_LOPASS slope 4 (8 sections), 4096-sample blocks, rows resident in device memory.
// Build: g++ -std=c++11 -O2 -I lsp-dsp-units_amd/include -I include tests/experiments/filter_array_rate.cpp \
//        -o tests/experiments/filter_array_rate -L lsp-dsp-units_amd -lmi_dspu -Wl,-rpath,$PWD/lsp-dsp-units_amd -Wl,-rpath,/opt/rocm/lib
#include <lsp-plug.in/dsp-units/filters/Filter.h>
#include <lsp-plug.in/dsp-units/filters/FilterArray.h>
#include <mi_dspu.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <vector>

using namespace lsp;

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t C = 1024, n = 4096, ring = 4, steps = 400;
    if (mi_dspu_device_count() <= 0) { puts("no HIP device"); return 2; }
    std::vector<dspu::filter_params_t> fp(C);
    for (size_t c = 0; c < C; ++c)
    {
        fp[c].nType = dspu::FLT_BT_LRX_LOPASS; fp[c].nSlope = 4; fp[c].fQuality = 0.75f; fp[c].fGain = 1.0f;
        fp[c].fFreq = fp[c].fFreq2 = 200.0f * std::pow(90.0f, float(c) / float(C - 1));        // 200 Hz .. 18 kHz
    }
    float *din = NULL, *dout = NULL;
    mi_dspu_malloc(reinterpret_cast<void **>(&din), ring * C * n * sizeof(float));
    mi_dspu_malloc(reinterpret_cast<void **>(&dout), ring * C * n * sizeof(float));
    mi_dspu_memset(din, 0, ring * C * n * sizeof(float), NULL);

    // (a) the C-ABI bank
    mi_biquad_bank_t *bank = NULL;
    mi_biquad_bank_create(&bank, C, 8);
    for (size_t c = 0; c < C; ++c)
    {
        mi_biquad_x1_t sec[8]; uint32_t k = 0;
        mi_filter_design(reinterpret_cast<const mi_filter_params_t *>(&fp[c]), 48000, sec, 8, &k, NULL, 0, NULL, NULL);
        mi_biquad_bank_set_chains(bank, uint32_t(c), sec, k, 1);
    }
    mi_biquad_bank_commit(bank, NULL);
    for (size_t i = 0; i < 20; ++i) mi_biquad_bank_process(bank, dout, din, n, n, n, NULL);
    mi_dspu_stream_synchronize(NULL);
    double t0 = now();
    for (size_t i = 0; i < steps; ++i)
        mi_biquad_bank_process(bank, dout + (i % ring) * C * n, din + (i % ring) * C * n, n, n, n, NULL);
    mi_dspu_stream_synchronize(NULL);
    const double t_bank = (now() - t0) / steps;

    // (b) FilterArray on the same rows
    dspu::FilterArray fa;
    if (!fa.init(C, 8)) { puts("FilterArray::init failed"); return 1; }
    for (size_t c = 0; c < C; ++c) fa.update(c, 48000, &fp[c]);
    for (size_t i = 0; i < 20; ++i) fa.process(dout, din, n, n);
    mi_dspu_stream_synchronize(NULL);
    t0 = now();
    for (size_t i = 0; i < steps; ++i)
        fa.process(dout + (i % ring) * C * n, din + (i % ring) * C * n, n, n);
    mi_dspu_stream_synchronize(NULL);
    const double t_array = (now() - t0) / steps;

    // (c) the reference's shape: one object per channel, host rows, one call each (a few blocks are enough)
    std::vector<dspu::Filter> fl(C);
    for (size_t c = 0; c < C; ++c) { fl[c].init(NULL); fl[c].update(48000, &fp[c]); }
    std::vector<float> hx(C * n, 0.1f), hy(C * n);
    for (size_t c = 0; c < C; ++c) fl[c].process(&hy[c * n], &hx[c * n], n);
    t0 = now();
    const size_t obj_steps = 3;
    for (size_t i = 0; i < obj_steps; ++i)
        for (size_t c = 0; c < C; ++c) fl[c].process(&hy[c * n], &hx[c * n], n);
    const double t_objects = (now() - t0) / obj_steps;
    // (d) FilterArray on host rows: one upload, one launch, one download
    fa.process_host(hy.data(), hx.data(), n, n);
    t0 = now();
    for (size_t i = 0; i < 20; ++i) fa.process_host(hy.data(), hx.data(), n, n);
    const double t_array_host = (now() - t0) / 20;

    const double ms = double(C * n) / 1e6;
    printf("1024 filters x 4096 samples, 8 sections each, per block:\n");
    printf("  mi_biquad_bank_process, resident rows      %9.1f us  %10.1f Msamples/s\n", t_bank * 1e6, ms / t_bank);
    printf("  dspu::FilterArray::process, resident rows  %9.1f us  %10.1f Msamples/s   = %.2f of the bank\n", t_array * 1e6, ms / t_array, t_bank / t_array);
    printf("  dspu::FilterArray::process_host (pageable) %9.1f us  %10.1f Msamples/s\n", t_array_host * 1e6, ms / t_array_host);
    printf("  1024 x dspu::Filter::process (host rows)   %9.1f us  %10.1f Msamples/s\n", t_objects * 1e6, ms / t_objects);
    mi_dspu_free(din); mi_dspu_free(dout);
    mi_biquad_bank_destroy(bank);
    return (t_bank / t_array >= 0.5) ? 0 : 1;
}
