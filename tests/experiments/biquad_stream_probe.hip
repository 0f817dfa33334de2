// Account of biquad_stream_kernel from inside the waves (lane 0 of every wave): where the shader cycles of a K-block
// launch go -- waiting for the tile's loads, the sections, the transposition and store issue -- and how often the
// hand-over wait loop turns.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMI_BIQUAD_PROBE -I include -I lsp-dsp-units_amd/csrc \
//        tests/experiments/biquad_stream_probe.hip lsp-dsp-units_amd/csrc/runtime.hip -o tests/experiments/biquad_stream_probe
// Run:   biquad_stream_probe [sections = 8] [samples = 4096] [blocks = 20]
#include "../../lsp-dsp-units_amd/csrc/biquad.hip"
#include <algorithm>
#include <cstdio>
#include <cmath>
#include <map>

int main(int argc, char **argv)
{
    const uint32_t C = 1024, NS = (argc > 1) ? atoi(argv[1]) : 8;
    const size_t n = (argc > 2) ? atoi(argv[2]) : 4096;
    const int K = (argc > 3) ? atoi(argv[3]) : 20;
    const int NWV = getenv("MI_BIQUAD_STREAM_WAVES") ? atoi(getenv("MI_BIQUAD_STREAM_WAVES")) : 4;
    mi_biquad_bank_t *bank = nullptr;
    if (mi_biquad_bank_create(&bank, C, NS ? NS : 1) != MI_OK) { printf("create: %s\n", mi_dspu_last_error()); return 1; }
    std::vector<mi_biquad_x1_t> ch(size_t(C) * (NS ? NS : 1));
    // per-channel low-passes with cut-offs log-uniform in 200 Hz .. 18 kHz (RBJ form, Q 0.75; a1, a2 with the reference's
    // sign convention: already negated) and noise as input, so that the arithmetic draws the power of the real workload
    for (uint32_t c = 0; c < C; ++c)
    {
        const double fc = 200.0 * std::pow(90.0, double((c * 2654435761u) % 1024) / 1024.0), w = 2.0 * M_PI * fc / 48000.0;
        const double al = std::sin(w) / 1.5, a0 = 1.0 + al, cs = std::cos(w);
        for (uint32_t k = 0; k < (NS ? NS : 1); ++k)
        {
            mi_biquad_x1_t &q = ch[size_t(c) * (NS ? NS : 1) + k];
            q.b0 = float((1.0 - cs) / 2.0 / a0); q.b1 = float((1.0 - cs) / a0); q.b2 = q.b0;
            q.a1 = float(2.0 * cs / a0); q.a2 = float(-(1.0 - al) / a0); q.p0 = q.p1 = q.p2 = 0.0f;
        }
    }
    mi_biquad_bank_set_all_chains(bank, ch.data(), NS, 1);
    float *in, *out;
    const int ring = 16;
    hipMalloc(&in, ring * C * n * sizeof(float)); hipMalloc(&out, ring * C * n * sizeof(float));
    {
        std::vector<float> hx(size_t(ring) * C * n);
        uint32_t r = 12345u;
        for (auto &v : hx) { r = r * 1664525u + 1013904223u; v = (float(r >> 8) / 8388608.0f - 1.0f) * 0.5f; }
        hipMemcpy(in, hx.data(), hx.size() * sizeof(float), hipMemcpyHostToDevice);
    }
    std::vector<float *> po(K);
    std::vector<const float *> pi(K);
    for (int k = 0; k < K; ++k) { po[k] = out + size_t(k % ring) * C * n; pi[k] = in + size_t(k % ring) * C * n; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 40;
    for (int rep = 0; rep < reps + 5; ++rep)
    {
        if (rep == 5) hipEventRecord(e0, nullptr);
        if (mi_biquad_bank_process_blocks(bank, po.data(), pi.data(), K, n, n, n, nullptr) != MI_OK) { printf("%s\n", mi_dspu_last_error()); return 1; }
    }
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%d blocks per launch, %d waves per channel: %.2f us per block (back-to-back launches)\n", K, NWV, ms * 1000.0f / reps / K);
    const uint32_t WAVES = C * NWV;
    std::vector<unsigned long long> h(4096 * 16 * 2);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_probe), h.size() * sizeof(h[0]));
    unsigned long long t0 = ~0ull;
    for (uint32_t b = 0; b < WAVES; ++b) t0 = std::min(t0, h[b * 32]);
    auto stat = [&](const char *name, int slot, bool wall, double scale) {
        std::vector<double> v;
        for (uint32_t b = 0; b < WAVES; ++b) v.push_back(wall ? (h[b * 32 + slot] - t0) / 100.0 : double(h[b * 32 + slot]) * scale);
        std::sort(v.begin(), v.end());
        printf("  %-44s %9.2f %9.2f %9.2f\n", name, v.front(), v[v.size() / 2], v.back());
    };
    printf("last launch, min / median / max over %u waves\n", WAVES);
    stat("entry (us since the first wave)", 0, true, 1);
    stat("first tile ready (us)", 7, true, 1);
    stat("first store issued (us)", 8, true, 1);
    stat("exit (us)", 1, true, 1);
    const double per = 1.0 / K;
    stat("shader cycles per block, all", 2, false, per);
    stat("  waiting for the tile's loads", 3, false, per);
    stat("  sections", 4, false, per);
    stat("  transposition, store issue, loop", 5, false, per);
    stat("turns of the hand-over wait loop per block", 6, false, per);
    {
        std::vector<double> v;
        for (uint32_t b = 0; b < WAVES; ++b) v.push_back(double(h[b * 32 + 2]) / ((h[b * 32 + 1] - h[b * 32 + 0]) / 100.0) / 1000.0);
        std::sort(v.begin(), v.end());
        printf("  %-44s %9.3f %9.3f %9.3f\n", "shader clock (GHz: cycles / wall time)", v.front(), v[v.size() / 2], v.back());
    }
    // the stream's pace: when the waves left their i-th sub-block, and the shader clock over that sub-block
    printf("wave iteration i done (us): min / median / max over the waves, the median's step, median clock (GHz) over the iteration\n");
    double prev = 0;
    for (int it = 0; it < 20 && it * NWV < K * int((n + 2047) / 2048); ++it)
    {
        std::vector<double> v, ck;
        for (uint32_t b = 0; b < WAVES; ++b)
        {
            const unsigned long long cur = h[b * 32 + 11 + it];
            if (!cur) continue;
            v.push_back(double(uint32_t(cur) - uint32_t(t0)) / 100.0);
            if (it > 0 && h[b * 32 + 10 + it])
            {
                const unsigned long long pr = h[b * 32 + 10 + it];
                const double dw = double(uint32_t(cur) - uint32_t(pr)) / 100.0, dc = double(uint32_t(cur >> 32) - uint32_t(pr >> 32));
                if (dw > 0) ck.push_back(dc / dw / 1000.0);
            }
        }
        if (v.empty()) break;
        std::sort(v.begin(), v.end());
        std::sort(ck.begin(), ck.end());
        printf("  %2d: %8.2f %8.2f %8.2f   +%.2f   %.3f\n", it, v.front(), v[v.size() / 2], v.back(), v[v.size() / 2] - prev, ck.empty() ? 0.0 : ck[ck.size() / 2]);
        prev = v[v.size() / 2];
    }
    // age inside a CU: the workgroups of a CU in the order of their entry, and when each one left
    std::map<unsigned, std::vector<std::pair<unsigned long long, double>>> cu;
    std::map<unsigned, std::vector<uint32_t>> members;
    for (uint32_t b = 0; b < C; ++b)
    {
        const unsigned hw = unsigned(h[b * NWV * 32 + 9]), xcc = unsigned(h[b * NWV * 32 + 10]) & 0xf;
        const unsigned key = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
        double ex = 0;
        for (int w = 0; w < NWV; ++w) ex = std::max(ex, (h[(b * NWV + w) * 32 + 1] - t0) / 100.0);
        cu[key].push_back({h[b * NWV * 32], ex});
        members[key].push_back(b);
    }
    int regular = 0;
    for (auto &kv : members)
    {
        std::sort(kv.second.begin(), kv.second.end());
        bool ok = kv.second.size() == 4;
        for (size_t i = 0; ok && i < 4; ++i) ok = kv.second[i] == kv.second[0] + 256 * i;
        regular += ok;
    }
    printf("CUs whose workgroups are b, b + 256, b + 512, b + 768: %d\n", regular);
    std::map<size_t, int> hist;
    std::vector<std::vector<double>> by_rank(8);
    for (auto &kv : cu)
    {
        hist[kv.second.size()]++;
        std::sort(kv.second.begin(), kv.second.end());
        for (size_t r = 0; r < kv.second.size() && r < 8; ++r) by_rank[r].push_back(kv.second[r].second);
    }
    // where the slow workgroups sit: exit of the LAST workgroup of every CU, by XCD and by shader engine
    {
        std::map<unsigned, std::vector<double>> by_xcc, by_se;
        for (auto &kv : cu)
        {
            double last = 0;
            for (auto &e : kv.second) last = std::max(last, e.second);
            by_xcc[kv.first >> 12].push_back(last);
            by_se[kv.first >> 8].push_back(last);
        }
        printf("exit of a CU's last workgroup (us), min / median / max over the CUs of an XCD:\n");
        for (auto &kv : by_xcc)
        {
            auto &v = kv.second; std::sort(v.begin(), v.end());
            printf("  xcd %u (%zu CUs): %8.2f %8.2f %8.2f\n", kv.first, v.size(), v.front(), v[v.size() / 2], v.back());
        }
        printf("  ... by shader engine (median):");
        for (auto &kv : by_se) { auto &v = kv.second; std::sort(v.begin(), v.end()); printf(" %.0f", v[v.size() / 2]); }
        printf("\n");
    }
    printf("CUs %zu;", cu.size());
    for (auto &kv : hist) printf(" %d CUs with %zu workgroups;", kv.second, kv.first);
    printf("\nexit time (us) of a CU's workgroups by order of entry: min / median / max over the CUs\n");
    for (size_t r = 0; r < 8; ++r)
    {
        auto &v = by_rank[r];
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("  entered %zu.: %9.2f %9.2f %9.2f\n", r + 1, v.front(), v[v.size() / 2], v.back());
    }
    return 0;
}
