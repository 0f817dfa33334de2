// The Convolver's batches of frames (C3 geometry: 256 channels, 65 536 taps, 4096-sample frames, batches of 16): the timeline
// inside conv_batch_tail_kernel<16> (thread 0 of every workgroup, 100 MHz wall clock): 7 entry, 0 first bins formed (first
// piece of a channel only), 1 window of frames in its
// registers, 2 first step over the partitions done, 3 eighth step done, 4 last step done, 5 operands of the p = 1 term in,
// 6 exit (stores issued).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMI_CONV_PROBE -I include -I lsp-dsp-units_amd/csrc \
//        tests/experiments/conv_tail_probe.hip lsp-dsp-units_amd/csrc/runtime.hip -o tests/experiments/conv_tail_probe
#include "../../lsp-dsp-units_amd/csrc/convolver.hip"
#include <algorithm>
#include <cstdio>

int main(int argc, char **argv)
{
    const uint32_t C = (argc > 1) ? atoi(argv[1]) : 256, taps = 65536, frame = 4096, K = 16;
    std::vector<float> ir(size_t(C) * taps);
    for (size_t i = 0; i < ir.size(); ++i) ir[i] = float((i * 7919) % 1000) * 1e-6f;
    mi_convolver_bank_t *bank = nullptr;
    if (mi_convolver_bank_create(&bank, C, ir.data(), taps, nullptr, taps, 13, 0.0f, nullptr) != MI_OK) { printf("create: %s\n", mi_dspu_last_error()); return 1; }
    float *in, *out;
    (void)hipMalloc(&in, size_t(K) * C * frame * 4); (void)hipMalloc(&out, size_t(K) * C * frame * 4);
    (void)hipMemset(in, 0, size_t(K) * C * frame * 4);
    std::vector<float *> outs(K); std::vector<const float *> ins(K);
    for (uint32_t k = 0; k < K; ++k) { outs[k] = out + size_t(k) * C * frame; ins[k] = in + size_t(k) * C * frame; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i)
        if (mi_convolver_bank_process_blocks(bank, outs.data(), ins.data(), K, frame, frame, frame, nullptr) != MI_OK) { printf("process: %s\n", mi_dspu_last_error()); return 1; }
    (void)hipDeviceSynchronize();
    const int batches = 20;
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < batches; ++i)
        (void)mi_convolver_bank_process_blocks(bank, outs.data(), ins.data(), K, frame, frame, frame, nullptr);
    (void)hipEventRecord(e1, nullptr);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%u channels, batches of %u frames: %.2f us per frame\n", C, K, ms * 1000.0f / (batches * K));
    const uint32_t W = C * 8;                               // workgroups of the tail kernel
    std::vector<unsigned long long> h(4096 * 8);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_tail_probe), h.size() * 8);
    unsigned long long t0 = ~0ull;
    for (uint32_t b = 0; b < W && b < 4096; ++b) t0 = std::min(t0, h[b * 8 + 7]);
    static const char *names[8] = { "first bins formed", "window in registers", "step 1 done", "step 8 done", "last step done", "p = 1 operands in", "exit", "entry" };
    printf("conv_batch_tail_kernel<16>: us since the first workgroup's entry, min / quartile / median / 3rd quartile / max over %u workgroups\n", W);
    for (int s0 = -1; s0 < 7; ++s0)
    {
        const int s = (s0 < 0) ? 7 : s0;
        std::vector<double> v;
        for (uint32_t b = 0; b < W && b < 4096; ++b) v.push_back((h[b * 8 + s] - t0) / 100.0);
        std::sort(v.begin(), v.end());
        printf("  %-22s %7.2f %7.2f %7.2f %7.2f %7.2f\n", names[s], v.front(), v[v.size() / 4], v[v.size() / 2], v[3 * v.size() / 4], v.back());
    }
    printf("workgroups inside the kernel at t (entered - left):");
    for (double t = 2.0; t < 130.0; t += 8.0)
    {
        int n = 0;
        for (uint32_t b = 0; b < W && b < 4096; ++b)
            n += ((h[b * 8 + 7] - t0) / 100.0 <= t) - ((h[b * 8 + 6] - t0) / 100.0 <= t);
        printf(" %.0f us: %d |", t, n);
    }
    printf("\n");
    printf("workgroups that enter after 100 us (index = channel * 8 + piece):");
    for (uint32_t b = 0; b < W && b < 4096; ++b)
        if ((h[b * 8 + 7] - t0) / 100.0 > 100.0) printf(" %u", b);
    printf("\nentry of workgroups 0, 1, 2, ... in steps of 64:");
    for (uint32_t b = 0; b < W && b < 4096; b += 64) printf(" %.1f", (h[b * 8 + 7] - t0) / 100.0);
    printf("\n");
    printf("per workgroup (us), median: ");
    for (int s = 0; s < 7; ++s)
    {
        std::vector<double> v;
        for (uint32_t b = 0; b < W && b < 4096; ++b) v.push_back((h[b * 8 + s] - h[b * 8 + (s ? s - 1 : 7)]) / 100.0);
        std::sort(v.begin(), v.end());
        printf(" %s %.2f |", names[s], v[v.size() / 2]);
    }
    printf("\n");
    return 0;
}
