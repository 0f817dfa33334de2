// Where do the workgroups of a 1024 x 128-thread launch land?  Records HW_ID / XCC_ID / start time of every wave
// (same LDS footprint as biquad_bank_kernel<16,2>), prints which workgroups share a CU and a SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tests/experiments/census.hip -o tests/experiments/census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
__global__ __launch_bounds__(128, 2) void census(unsigned *rec, int spin)
{
    __shared__ float pad[6144];                      // 24.5 KB like the biquad kernel
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    const unsigned long long t = wall_clock64();
    pad[threadIdx.x] = float(hw);
    const unsigned long long until = t + spin;
    while (wall_clock64() < until) __builtin_amdgcn_s_sleep(4);
    if ((threadIdx.x & 63) == 0)
    {
        unsigned *r = rec + (blockIdx.x * 2 + (threadIdx.x >> 6)) * 4;
        r[0] = hw; r[1] = xcc; r[2] = unsigned(t); r[3] = unsigned(pad[threadIdx.x] != 0.f);
    }
}
int main()
{
    const int G = 1024;
    unsigned *d; hipMalloc(&d, G * 2 * 4 * sizeof(unsigned));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(census, dim3(G), dim3(128), 0, 0, d, 500);
    hipDeviceSynchronize();
    std::vector<unsigned> h(G * 2 * 4);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    unsigned t0 = ~0u; for (int i = 0; i < G * 2; ++i) t0 = std::min(t0, h[i * 4 + 2]);
    std::map<unsigned, std::vector<int>> cu, simd;
    for (int b = 0; b < G; ++b) for (int w = 0; w < 2; ++w)
    {
        const unsigned hw = h[(b * 2 + w) * 4], xcc = h[(b * 2 + w) * 4 + 1] & 0xf;
        const unsigned wave = hw & 15, sm = (hw >> 4) & 3, cuid = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cuid;
        cu[key].push_back(b * 2 + w);
        simd[(key << 2) | sm].push_back(b * 2 + w);
        if (b < 24 || (b % 128) == 0) printf("wg %4d wave %d: xcc %u se %u sh %u cu %2u simd %u slot %2u start +%u ticks\n", b, w, xcc, se, sh, cuid, sm, wave, h[(b * 2 + w) * 4 + 2] - t0);
    }
    printf("distinct CUs %zu, distinct SIMDs %zu\n", cu.size(), simd.size());
    std::map<size_t, int> hist; for (auto &kv : simd) hist[kv.second.size()]++;
    for (auto &kv : hist) printf("  SIMDs hosting %zu waves: %d\n", kv.first, kv.second);
    int shown = 0;
    for (auto &kv : simd) { if (shown++ >= 12) break; printf("  simd %05x:", kv.first); for (int v : kv.second) printf(" wg%d.w%d", v / 2, v & 1); printf("\n"); }
    // which cohort functions put two different cohorts on every SIMD?
    const char *names[] = {"(b>>8)&1", "b&1", "(b>>9)&1", "(b>>3)&1", "(b>>7)&1", "(b>>6)&1", "(b>>5)&1", "(b>>4)&1", "wave"};
    for (int f = 0; f < 9; ++f)
    {
        int mixed = 0, total = 0;
        for (auto &kv : simd)
        {
            if (kv.second.size() != 2) continue;
            ++total;
            auto c = [&](int v) { const int b = v / 2; switch (f) { case 0: return (b >> 8) & 1; case 1: return b & 1; case 2: return (b >> 9) & 1; case 3: return (b >> 3) & 1;
                                   case 4: return (b >> 7) & 1; case 5: return (b >> 6) & 1; case 6: return (b >> 5) & 1; case 7: return (b >> 4) & 1; default: return v & 1; } };
            if (c(kv.second[0]) != c(kv.second[1])) ++mixed;
        }
        printf("cohort %-10s: %d of %d two-wave SIMDs host one wave of each cohort\n", names[f], mixed, total);
    }
    return 0;
}
