// Phase timing of conv_frame_kernel<12> (thread 0 of every workgroup, 100 MHz wall clock):
//   0 entry, 1 operand prefetch issued, 2 forward transform done, 3 product + ring store done, 4 inverse done, 5 exit.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMI_CONV_PROBE -I include -I lsp-dsp-units_amd/csrc \
//        tests/experiments/conv_frame_probe.hip lsp-dsp-units_amd/csrc/runtime.hip -o tests/experiments/conv_frame_probe
#include "../../lsp-dsp-units_amd/csrc/convolver.hip"
#include <algorithm>
#include <cstdio>

int main(int argc, char **argv)
{
    const uint32_t C = 256, taps = (argc > 1) ? atoi(argv[1]) : 65536, frame = 4096;
    std::vector<float> ir(size_t(C) * taps);
    for (size_t i = 0; i < ir.size(); ++i) ir[i] = float((i * 7919) % 1000) * 1e-6f;
    mi_convolver_bank_t *bank = nullptr;
    if (mi_convolver_bank_create(&bank, C, ir.data(), taps, nullptr, taps, 13, 0.0f, nullptr) != MI_OK) { printf("create: %s\n", mi_dspu_last_error()); return 1; }
    float *in, *out;
    (void)hipMalloc(&in, size_t(C) * frame * 4); (void)hipMalloc(&out, size_t(C) * frame * 4);
    (void)hipMemset(in, 0, size_t(C) * frame * 4);
    for (int rep = 0; rep < 20; ++rep)
        if (mi_convolver_bank_process(bank, out, in, frame, frame, frame, nullptr) != MI_OK) { printf("process: %s\n", mi_dspu_last_error()); return 1; }
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024 * 8);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_conv_probe), h.size() * 8);
    unsigned long long t0 = ~0ull;
    for (uint32_t b = 0; b < C; ++b) t0 = std::min(t0, h[b * 8]);
    for (int s = 0; s < 6; ++s)
    {
        std::vector<double> v;
        for (uint32_t b = 0; b < C; ++b) v.push_back((h[b * 8 + s] - t0) / 100.0);
        std::sort(v.begin(), v.end());
        printf("  slot %d: %7.2f %7.2f %7.2f us (min / median / max over %u workgroups)\n", s, v.front(), v[v.size() / 2], v.back(), C);
    }
    return 0;
}
