// Phase timing of analyzer_frames_wave_kernel at C5 (lane 0 of every wave, 100 MHz wall clock), 16 strobes per launch, a wave's
// units u = 0 .. 3 (channels blockIdx.x + 256 u):  6u + 0 hops here, + 1 window formed and hops filed, + 2 the last block's rows
// filed, + 3 the transform's exchange done (the next unit's hops are asked for), + 4 transform done, + 5 rows stored;
// 30 entry, 31 tables filled, 29 exit.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMI_AN_PROBE -I include -I lsp-dsp-units_amd/csrc -I lsp-dsp-units_amd/include \
//        tests/experiments/analyzer_wave_probe.hip $(ls lsp-dsp-units_amd/build/*.o lsp-dsp-units_amd/build/host/*.o | grep -v spectral) -o tests/experiments/analyzer_wave_probe
#include "../../lsp-dsp-units_amd/csrc/spectral.hip"
#include <algorithm>
#include <cstdio>

int main(int argc, char **argv)
{
    const uint32_t C = 1024, rank = 12, period = 2048, F = 16;
    mi_analyzer_bank_t *bank = nullptr;
    if (mi_analyzer_bank_create(&bank, C, rank, 48000, 1.0f, 0) != MI_OK) { printf("create: %s\n", mi_dspu_last_error()); return 1; }
    mi_analyzer_bank_configure(bank, MI_ANALYZER_SAMPLE_RATE, 48000.0f);
    mi_analyzer_bank_configure(bank, MI_ANALYZER_RANK, float(rank));
    mi_analyzer_bank_configure(bank, MI_ANALYZER_RATE, 48000.0f / float(period));
    mi_analyzer_bank_configure(bank, MI_ANALYZER_REACTIVITY, 0.2f);
    float *in, *sums;
    (void)hipMalloc(&in, size_t(F) * C * period * 4);
    (void)hipMalloc(&sums, size_t(F) * 2049 * 4);
    {
        std::vector<float> h(size_t(F) * C * period);
        unsigned r = 7;
        for (float &v : h) { r = r * 1664525u + 1013904223u; v = (float(r >> 8) / 8388608.0f - 1.0f) * 0.5f; }
        (void)hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    }
    const float *ptr[16];
    for (uint32_t f = 0; f < F; ++f) ptr[f] = in + size_t(f) * C * period;
    for (int rep = 0; rep < 20; ++rep)
        if (mi_analyzer_bank_process_reduce_frames(bank, ptr, F, period, period, sums, 2049, 0, nullptr) != MI_OK) { printf("process: %s\n", mi_dspu_last_error()); return 1; }
    (void)hipDeviceSynchronize();
    const int W = 256 * 8;
    std::vector<unsigned long long> h(size_t(W) * 32);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_anw_probe), h.size() * 8);
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < W; ++w) t0 = std::min(t0, h[size_t(w) * 32 + 30]);
    auto stat = [&](const char *name, int a, int b) {      // b - a per wave (a < 0: since t0)
        std::vector<double> v;
        for (int w = 0; w < W; ++w)
            v.push_back(double(h[size_t(w) * 32 + b] - (a < 0 ? t0 : h[size_t(w) * 32 + a])) / 100.0);
        std::sort(v.begin(), v.end());
        printf("  %-46s %7.2f %7.2f %7.2f %7.2f us (min / median / p90 / max over %d waves)\n", name, v.front(), v[v.size() / 2], v[v.size() * 9 / 10], v.back(), W);
    };
    stat("entry", -1, 30);
    stat("tables filled (since entry)", 30, 31);
    for (int u = 0; u < 4; ++u)
    {
        char nm[64];
        snprintf(nm, sizeof nm, "unit %d: head reached (since launch)", u); stat(nm, -1, 6 * u);
        if (u == 0) stat("   wait from tables to hops", 31, 0); else { snprintf(nm, sizeof nm, "   wait for hops (unit %d end -> head)", u - 1); stat(nm, 6 * (u - 1) + 5, 6 * u); }
        stat("   window + filing", 6 * u, 6 * u + 1);
        stat("   last block's rows", 6 * u + 1, 6 * u + 2);
        stat("   transform, first half + exchange", 6 * u + 2, 6 * u + 3);
        stat("   requests + second half", 6 * u + 3, 6 * u + 4);
        stat("   magnitudes + rows", 6 * u + 4, 6 * u + 5);
    }
    stat("exit (since launch)", -1, 29);
    return 0;
}
