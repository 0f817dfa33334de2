// The Convolver fed in 256-sample calls (C3 geometry: 256 channels, 65 536 taps, 4096-sample frame): where a call's time
// goes.  (a) the rate of the call stream from the host's side (events around whole frames of sixteen calls), (b) the rate of
// an EMPTY kernel launched the same way (what sixteen launches cost before they do anything), (c) the timeline inside
// conv_small_kernel<true> (lane 0 of both waves of every workgroup, 100 MHz wall clock) for the block given on the command
// line:  0 entry, 1 operand loads issued, 2 forward transform done, 3 split + products + merge done, 4 inverse done,
// 5 hand-over barrier passed, 6 exit.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMI_CONV_PROBE -I include -I lsp-dsp-units_amd/csrc \
//        tests/experiments/conv_small_probe.hip lsp-dsp-units_amd/csrc/runtime.hip -o tests/experiments/conv_small_probe
// Run:   conv_small_probe [block of the frame the timeline is taken from = 8] [call size = 256]
#include "../../lsp-dsp-units_amd/csrc/convolver.hip"
#include <algorithm>
#include <cstdio>

__global__ void empty_kernel(float *p) { if (p == (float *)1) *p = 0.0f; }

int main(int argc, char **argv)
{
    const uint32_t C = 256, taps = 65536, frame = 4096;
    const int kblk = (argc > 1) ? atoi(argv[1]) : 8;
    const size_t call = (argc > 2) ? atoi(argv[2]) : 256;
    std::vector<float> ir(size_t(C) * taps);
    for (size_t i = 0; i < ir.size(); ++i) ir[i] = float((i * 7919) % 1000) * 1e-6f;
    mi_convolver_bank_t *bank = nullptr;
    if (mi_convolver_bank_create(&bank, C, ir.data(), taps, nullptr, taps, 13, 0.0f, nullptr) != MI_OK) { printf("create: %s\n", mi_dspu_last_error()); return 1; }
    float *in, *out;
    (void)hipMalloc(&in, size_t(C) * frame * 4); (void)hipMalloc(&out, size_t(C) * frame * 4);
    (void)hipMemset(in, 0, size_t(C) * frame * 4);
    auto feed = [&](size_t samples) -> bool {              // `samples` of the stream in calls of `call`
        for (size_t d = 0; d < samples; d += call)
            if (mi_convolver_bank_process(bank, out + (d % frame), in + (d % frame), call, frame, frame, nullptr) != MI_OK)
            { printf("process: %s\n", mi_dspu_last_error()); return false; }
        return true;
    };
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    if (!feed(4 * frame)) return 1;
    (void)hipDeviceSynchronize();
    const int frames = 50;
    (void)hipEventRecord(e0, nullptr);
    if (!feed(size_t(frames) * frame)) return 1;
    (void)hipEventRecord(e1, nullptr);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("calls of %zu samples: %.2f us per frame of 4096 (%d frames, %zu calls each)\n", call, ms * 1000.0f / frames, frames, frame / call);

    // whole-frame calls, for the ratio
    (void)hipEventRecord(e0, nullptr);
    for (int f = 0; f < frames; ++f)
        (void)mi_convolver_bank_process(bank, out, in, frame, frame, frame, nullptr);
    (void)hipEventRecord(e1, nullptr);
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("whole-frame calls: %.2f us per frame\n", ms * 1000.0f / frames);

    // sixteen empty launches in a row
    for (int i = 0; i < 64; ++i) hipLaunchKernelGGL(empty_kernel, dim3(C), dim3(128), 0, nullptr, (float *)nullptr);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < 16 * frames; ++i) hipLaunchKernelGGL(empty_kernel, dim3(C), dim3(128), 0, nullptr, (float *)nullptr);
    (void)hipEventRecord(e1, nullptr);
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("empty kernel, same grid, back to back: %.2f us per launch\n", ms * 1000.0f / (16 * frames));

    // the small-block launches of one frame alone (blocks 0 .. 14: the sixteenth call also completes the frame)
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, nullptr);
    if (!feed(15 * call)) return 1;
    (void)hipEventRecord(e1, nullptr);
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("fifteen block calls of a frame: %.2f us per call\n", ms * 1000.0f / 15);
    if (!feed(call)) return 1;                              // completes the frame

    // the timeline of block `kblk`: stop the stream right behind it
    if (!feed(size_t(kblk + 1) * 256)) return 1;
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024 * 2 * 8);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_small_probe), h.size() * 8);
    unsigned long long t0 = ~0ull;
    for (uint32_t b = 0; b < 2 * C; ++b) t0 = std::min(t0, h[b * 8]);
    static const char *names[7] = { "entry", "operand loads issued", "forward transform done", "split + products + merge", "inverse done",
                                    "hand-over barrier passed", "exit" };
    for (int role = 0; role < 2; ++role)
    {
        printf("block %d of the frame, %s wave: us since the first wave's entry, min / median / max over %u workgroups\n", kblk,
               role ? "debt" : "output", C);
        for (int s = 0; s < 7; ++s)
        {
            std::vector<double> v;
            for (uint32_t b = 0; b < C; ++b) v.push_back((h[(b * 2 + role) * 8 + s] - t0) / 100.0);
            std::sort(v.begin(), v.end());
            printf("  %-28s %7.2f %7.2f %7.2f\n", names[s], v.front(), v[v.size() / 2], v.back());
        }
    }
    return 0;
}
