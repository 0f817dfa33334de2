// Phase timing of analyzer_kernel<11> at C5 (thread 0 of every workgroup, 100 MHz wall clock):
//   0 entry, 1 operands requested, 2 twiddles ready, 3 windowed frame in LDS, 4 forward transform done, 5 spectrum stored, 6 exit.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMI_AN_PROBE -I include -I lsp-dsp-units_amd/csrc -I lsp-dsp-units_amd/include \
//        tests/experiments/analyzer_probe.hip lsp-dsp-units_amd/csrc/runtime.hip lsp-dsp-units_amd/csrc/host/windows.cpp lsp-dsp-units_amd/csrc/comm.hip ... (see below)
#include "../../lsp-dsp-units_amd/csrc/spectral.hip"
#include <algorithm>
#include <cstdio>

int main()
{
    const uint32_t C = 1024, rank = 12, period = 2048;
    mi_analyzer_bank_t *bank = nullptr;
    if (mi_analyzer_bank_create(&bank, C, rank, 48000, 10.0f, 0) != MI_OK) { printf("create: %s\n", mi_dspu_last_error()); return 1; }
    mi_analyzer_bank_configure(bank, MI_ANALYZER_SAMPLE_RATE, 48000.0f);
    mi_analyzer_bank_configure(bank, MI_ANALYZER_RANK, float(rank));
    mi_analyzer_bank_configure(bank, MI_ANALYZER_RATE, 48000.0f / float(period));
    mi_analyzer_bank_configure(bank, MI_ANALYZER_REACTIVITY, 0.2f);
    float *in;
    (void)hipMalloc(&in, size_t(C) * period * 4);
    (void)hipMemset(in, 0, size_t(C) * period * 4);
    for (int rep = 0; rep < 20; ++rep)
        if (mi_analyzer_bank_process(bank, in, period, period, nullptr) != MI_OK) { printf("process: %s\n", mi_dspu_last_error()); return 1; }
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(4096 * 8);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_an_probe), h.size() * 8);
    unsigned long long t0 = ~0ull;
    for (uint32_t b = 0; b < C; ++b) t0 = std::min(t0, h[b * 8]);
    for (int s = 0; s < 7; ++s)
    {
        std::vector<double> v;
        for (uint32_t b = 0; b < C; ++b) v.push_back((h[b * 8 + s] - t0) / 100.0);
        std::sort(v.begin(), v.end());
        printf("  slot %d: %7.2f %7.2f %7.2f us (min / median / max over %u workgroups)\n", s, v.front(), v[v.size() / 2], v.back(), C);
    }
    return 0;
}
