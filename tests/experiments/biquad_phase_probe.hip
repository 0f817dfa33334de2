// Phase timing of the biquad bank kernel from inside the wave (lane 0 of every block):
//   slot 0 kernel entry, then per sub-block sb: 1+4sb input tile ready, 2+4sb sections done, 3+4sb stores issued; 15 exit.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMI_BIQUAD_PROBE -I include -I lsp-dsp-units_amd/csrc \
//        tests/experiments/biquad_phase_probe.hip lsp-dsp-units_amd/csrc/runtime.hip -o tests/experiments/biquad_phase_probe
#include "../../lsp-dsp-units_amd/csrc/biquad.hip"
#include <algorithm>
#include <cstdio>

int main(int argc, char **argv)
{
    const uint32_t C = 1024, WAVES = (getenv("MI_BIQUAD_WAVES") && atoi(getenv("MI_BIQUAD_WAVES")) == 1) ? C : 2 * C, NS = (argc > 1) ? atoi(argv[1]) : 8;
    const size_t n = (argc > 2) ? atoi(argv[2]) : 4096;
    mi_biquad_bank_t *bank = nullptr;
    if (mi_biquad_bank_create(&bank, C, NS ? NS : 1) != MI_OK) { printf("create: %s\n", mi_dspu_last_error()); return 1; }
    std::vector<mi_biquad_x1_t> ch(size_t(C) * (NS ? NS : 1));
    for (auto &q : ch) { q.b0 = 0.2f; q.b1 = 0.4f; q.b2 = 0.2f; q.a1 = 0.5f; q.a2 = -0.3f; q.p0 = q.p1 = q.p2 = 0.0f; }
    mi_biquad_bank_set_all_chains(bank, ch.data(), NS, 1);
    float *in, *out;
    const int ring = 16;
    hipMalloc(&in, ring * C * n * sizeof(float)); hipMalloc(&out, ring * C * n * sizeof(float));
    hipMemset(in, 0, ring * C * n * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 240; ++rep)
    {
        if (rep == 40) hipEventRecord(e0, nullptr);
        mi_biquad_bank_process(bank, out + size_t(rep % ring) * C * n, in + size_t(rep % ring) * C * n, n, n, n, nullptr);
    }
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("ablate %d: %.2f us per call (back-to-back launches)\n", MI_ABLATE, ms * 1000.0f / 200.0f);
    if (MI_ABLATE) return 0;
    std::vector<unsigned long long> h(4096 * 16 * 2);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_probe), h.size() * sizeof(h[0]));
    unsigned long long t0 = ~0ull;
    for (uint32_t b = 0; b < WAVES; ++b) t0 = std::min(t0, h[(b * 16) * 2]);
    const int slots[] = {0, 1, 2, 3, 5, 6, 7, 15};
    printf("sections %u, n %zu: wall-clock us since the first wave started (min / median / max over %u waves)\n", NS, n, WAVES);
    for (int s : slots)
    {
        std::vector<double> v;
        for (uint32_t b = 0; b < WAVES; ++b) v.push_back((h[(b * 16 + s) * 2] - t0) / 100.0);
        std::sort(v.begin(), v.end());
        printf("  slot %2d: %7.2f %7.2f %7.2f\n", s, v.front(), v[v.size() / 2], v.back());
    }
    // per-wave shader-cycle durations of the phases
    const char *names[] = {"load0->tile", "sections(0)", "store(0)", "tile(1)", "sections(1)", "store(1)"};
    const int pairs[][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 5}, {5, 6}, {6, 7}};
    for (int p = 0; p < 6; ++p)
    {
        std::vector<double> v;
        for (uint32_t b = 0; b < WAVES; ++b)
            v.push_back(double(h[(b * 16 + pairs[p][1]) * 2 + 1] - h[(b * 16 + pairs[p][0]) * 2 + 1]));
        std::sort(v.begin(), v.end());
        printf("  %-12s cycles: %8.0f %8.0f %8.0f\n", names[p], v.front(), v[v.size() / 2], v.back());
    }
    const char *sn[] = {"tables+pass 1", "combine+row scan", "totals+publish", "barrier", "carry+chain", "select+starts", "pass 2"};
    for (int p = 0; p < 7; ++p)
    {
        if (WAVES == C && (p == 2 || p == 3)) continue;
        std::vector<double> v;
        const int from = (WAVES == C && p == 4) ? 2 : p;
        for (uint32_t b = 0; b < WAVES; ++b)
            v.push_back(double(h[(b * 16 + 9 + p) * 2 + 1] - h[(b * 16 + 8 + from) * 2 + 1]));
        std::sort(v.begin(), v.end());
        printf("  section 1: %-18s cycles: %8.0f %8.0f %8.0f\n", sn[p], v.front(), v[v.size() / 2], v.back());
    }
    return 0;
}
