// Phase timeline of biquad_roles_kernel from inside the waves (lane 0 of every wave; slots per tile i of the call):
//   role A (wave 0 of a workgroup): 1+4i tile i in LDS, 2+4i its sections done, 3+4i handed over (behind barrier i)
//   role B (wave 1):                1+4i tile i received,  2+4i its sections done, 3+4i stores issued;   0 entry, 15 exit
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMI_BIQUAD_PROBE -I include -I lsp-dsp-units_amd/csrc \
//        tests/experiments/biquad_roles_probe.hip lsp-dsp-units_amd/csrc/runtime.hip -o tests/experiments/biquad_roles_probe
#include "../../lsp-dsp-units_amd/csrc/biquad.hip"
#include <algorithm>
#include <cstdio>

int main(int argc, char **argv)
{
    const uint32_t C = 1024, WAVES = 2 * C, NS = (argc > 1) ? atoi(argv[1]) : 8;
    const size_t n = (argc > 2) ? atoi(argv[2]) : 4096;
    mi_biquad_bank_t *bank = nullptr;
    if (mi_biquad_bank_create(&bank, C, NS ? NS : 1) != MI_OK) { printf("create: %s\n", mi_dspu_last_error()); return 1; }
    std::vector<mi_biquad_x1_t> ch(size_t(C) * (NS ? NS : 1));
    for (auto &q : ch) { q.b0 = 0.2f; q.b1 = 0.4f; q.b2 = 0.2f; q.a1 = 0.5f; q.a2 = -0.3f; q.p0 = q.p1 = q.p2 = 0.0f; }
    mi_biquad_bank_set_all_chains(bank, ch.data(), NS, 1);
    float *in, *out;
    const int ring = 16;
    (void)hipMalloc(&in, ring * C * n * sizeof(float)); (void)hipMalloc(&out, ring * C * n * sizeof(float));
    (void)hipMemset(in, 0, ring * C * n * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 240; ++rep)
    {
        if (rep == 40) (void)hipEventRecord(e0, nullptr);
        mi_biquad_bank_process(bank, out + size_t(rep % ring) * C * n, in + size_t(rep % ring) * C * n, n, n, n, nullptr);
    }
    (void)hipEventRecord(e1, nullptr);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("sections %u, n %zu: %.2f us per call (back-to-back launches, probe build)\n", NS, n, ms * 1000.0f / 200.0f);
    std::vector<unsigned long long> h(4096 * 16 * 2);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_probe), h.size() * sizeof(h[0]));
    unsigned long long t0 = ~0ull;
    for (uint32_t b = 0; b < WAVES; ++b) t0 = std::min(t0, h[(b * 16) * 2]);
    const int tiles = int((n + 2047) / 2048);
    for (int role = 0; role < 2; ++role)
    {
        printf("role %c: wall-clock us since the first wave started (min / median / max over %u waves)\n", role ? 'B' : 'A', C);
        std::vector<int> slots = {0};
        for (int i = 0; i < tiles && i < 3; ++i) { slots.push_back(1 + 4 * i); slots.push_back(2 + 4 * i); slots.push_back(3 + 4 * i); }
        slots.push_back(15);
        for (int s : slots)
        {
            std::vector<double> v;
            for (uint32_t b = role; b < WAVES; b += 2) v.push_back((h[(b * 16 + s) * 2] - t0) / 100.0);
            std::sort(v.begin(), v.end());
            printf("  slot %2d: %7.2f %7.2f %7.2f\n", s, v.front(), v[v.size() / 2], v.back());
        }
        for (int i = 0; i < tiles && i < 3; ++i)
        {
            std::vector<double> v;
            for (uint32_t b = role; b < WAVES; b += 2)
                v.push_back(double(h[(b * 16 + 2 + 4 * i) * 2 + 1] - h[(b * 16 + 1 + 4 * i) * 2 + 1]));
            std::sort(v.begin(), v.end());
            printf("  sections of tile %d, shader cycles: %8.0f %8.0f %8.0f\n", i, v.front(), v[v.size() / 2], v.back());
        }
    }
    return 0;
}
