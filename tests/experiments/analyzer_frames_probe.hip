// Phase timing of analyzer_frames_kernel<11> at C5 (thread 0 of every workgroup, 100 MHz wall clock), 8 strobes per launch:
//   0 entry, 1 first frame and operands in registers, 7 end of strobe 2; strobe 3: 2 windowed frame formed, 3 transform + split
//   done, 4 magnitude, mix and row stored, 5 block loaded and in the ring (end of the strobe); 6 exit.
// Build: like analyzer_probe.hip (-DMI_AN_PROBE).
#include "../../lsp-dsp-units_amd/csrc/spectral.hip"
#include <algorithm>
#include <cstdio>

int main(int argc, char **argv)
{
    const uint32_t C = 1024, rank = 12, period = 2048, F = (argc > 1) ? atoi(argv[1]) : 8;
    mi_analyzer_bank_t *bank = nullptr;
    if (mi_analyzer_bank_create(&bank, C, rank, 48000, 10.0f, 0) != MI_OK) { printf("create: %s\n", mi_dspu_last_error()); return 1; }
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
    std::vector<unsigned long long> h(4096 * 8);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_an_probe), h.size() * 8);
    unsigned long long t0 = ~0ull;
    for (uint32_t b = 0; b < C; ++b) t0 = std::min(t0, h[b * 8]);
    const int order[] = {0, 1, 7, 2, 3, 4, 5, 6};
    for (int s : order)
    {
        std::vector<double> v;
        for (uint32_t b = 0; b < C; ++b) v.push_back((h[b * 8 + s] - t0) / 100.0);
        std::sort(v.begin(), v.end());
        printf("  slot %d: %7.2f %7.2f %7.2f us (min / median / max over %u workgroups)\n", s, v.front(), v[v.size() / 2], v.back(), C);
    }
    printf("workgroups inside the kernel at t (entered - left):");
    for (double t = 2.0; t < 170.0; t += 8.0)
    {
        int n = 0;
        for (uint32_t b = 0; b < C; ++b)
            n += ((h[b * 8] - t0) / 100.0 <= t) - ((h[b * 8 + 6] - t0) / 100.0 <= t);
        printf(" %.0f us: %d |", t, n);
    }
    printf("\n");
    {
        std::vector<double> v;
        for (uint32_t b = 0; b < C; ++b) v.push_back(double(h[b * 8 + 6] - h[b * 8]) / 100.0);
        std::sort(v.begin(), v.end());
        printf("  a workgroup's life: %.1f / %.1f / %.1f us (min / median / max), %u strobes\n", v.front(), v[v.size() / 2], v.back(), F);
    }
    printf("  median life by workgroup index mod 8 (the XCD a workgroup goes to):");
    for (uint32_t x = 0; x < 8; ++x)
    {
        std::vector<double> v;
        for (uint32_t b = x; b < C; b += 8) v.push_back(double(h[b * 8 + 6] - h[b * 8]) / 100.0);
        std::sort(v.begin(), v.end());
        printf(" %.1f (%.1f - %.1f)", v[v.size() / 2], v.front(), v.back());
    }
    printf("\n  median life by index / 128 (dispatch order):");
    for (uint32_t x = 0; x < C / 128; ++x)
    {
        std::vector<double> v;
        for (uint32_t b = x * 128; b < (x + 1) * 128; ++b) v.push_back(double(h[b * 8 + 6] - h[b * 8]) / 100.0);
        std::sort(v.begin(), v.end());
        printf(" %.1f", v[v.size() / 2]);
    }
    printf("\n");
    // durations inside strobe 3, per workgroup
    const char *names[] = {"strobe 2 end -> frame formed", "transform + split", "magnitude, mix, row", "block load + ingest + barrier"};
    const int pairs[][2] = {{7, 2}, {2, 3}, {3, 4}, {4, 5}};
    for (int p = 0; p < 4; ++p)
    {
        std::vector<double> v;
        for (uint32_t b = 0; b < C; ++b) v.push_back(double(h[b * 8 + pairs[p][1]] - h[b * 8 + pairs[p][0]]) / 100.0);
        std::sort(v.begin(), v.end());
        printf("  %-32s %6.2f %6.2f %6.2f us\n", names[p], v.front(), v[v.size() / 2], v.back());
    }
    return 0;
}
