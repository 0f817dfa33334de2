// Stand-alone timing + sanity harness for the biquad bank (C2 shape by default): back-to-back calls over a ring of
// buffers larger than the Infinity Cache, time per call from HIP events around the run, and a float64 check of a few
// channels.  Build (any -D experiment switches of biquad.hip may be added):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -I include -I lsp-dsp-units_amd/csrc \
//         tests/experiments/bq_time.hip lsp-dsp-units_amd/csrc/runtime.hip -o tests/experiments/bq_time
// Usage: bq_time [sections=8] [samples=4096] [channels=1024] [calls=400]
#include "../../lsp-dsp-units_amd/csrc/biquad.hip"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>

int main(int argc, char **argv)
{
    const uint32_t NS = (argc > 1) ? atoi(argv[1]) : 8;
    const size_t n = (argc > 2) ? atoi(argv[2]) : 4096;
    const uint32_t C = (argc > 3) ? atoi(argv[3]) : 1024;
    const int calls = (argc > 4) ? atoi(argv[4]) : 400;
    mi_biquad_bank_t *bank = nullptr;
    if (mi_biquad_bank_create(&bank, C, NS ? NS : 1) != MI_OK) { printf("create: %s\n", mi_dspu_last_error()); return 1; }
    // 2nd-order Butterworth low-pass sections, cutoff log-uniform 200 Hz .. 18 kHz per channel (bilinear, 48 kHz)
    std::mt19937 rng(3);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::vector<mi_biquad_x1_t> ch(size_t(C) * (NS ? NS : 1));
    for (uint32_t c = 0; c < C; ++c)
    {
        const double f = 200.0 * std::pow(90.0, U(rng));
        const double k = std::tan(M_PI * f / 48000.0), q = 0.7071 + 0.3 * U(rng);
        const double norm = 1.0 / (1.0 + k / q + k * k);
        for (uint32_t s = 0; s < NS; ++s)
        {
            mi_biquad_x1_t &b = ch[size_t(c) * NS + s];
            b.b0 = float(k * k * norm); b.b1 = float(2.0 * k * k * norm); b.b2 = b.b0;
            b.a1 = float(-2.0 * (k * k - 1.0) * norm); b.a2 = float(-(1.0 - k / q + k * k) * norm);     // signs pre-negated
            b.p0 = b.p1 = b.p2 = 0.0f;
        }
    }
    mi_biquad_bank_set_all_chains(bank, ch.data(), NS, 1);
    const int ring = 12;
    float *in, *out;
    const size_t blk = size_t(C) * n;
    if (hipMalloc(&in, ring * blk * sizeof(float)) != hipSuccess || hipMalloc(&out, ring * blk * sizeof(float)) != hipSuccess) return 2;
    std::vector<float> hx(ring * blk);
    std::normal_distribution<float> N01(0.0f, 0.25f);
    for (auto &v : hx) v = N01(rng);
    hipMemcpy(in, hx.data(), hx.size() * sizeof(float), hipMemcpyHostToDevice);

    // correctness: two consecutive calls from cleared state against float64, first and last channels
    std::vector<float> hy(2 * blk);
    for (int b = 0; b < 2; ++b)
        if (mi_biquad_bank_process(bank, out + b * blk, in + b * blk, n, n, n, nullptr) != MI_OK) { printf("process: %s\n", mi_dspu_last_error()); return 1; }
    hipMemcpy(hy.data(), out, 2 * blk * sizeof(float), hipMemcpyDeviceToHost);
    double worst = 0, worst32 = 0;
    for (uint32_t c : {0u, 1u, C / 2, C - 1})
    {
        std::vector<double> y(2 * n);
        std::vector<float> y32(2 * n);
        for (int b = 0; b < 2; ++b) for (size_t i = 0; i < n; ++i) { y[b * n + i] = hx[b * blk + c * n + i]; y32[b * n + i] = hx[b * blk + c * n + i]; }
        for (uint32_t s = 0; s < NS; ++s)
        {
            const mi_biquad_x1_t &q = ch[size_t(c) * NS + s];
            double d0 = 0, d1 = 0;
            float f0 = 0, f1 = 0;
            for (size_t i = 0; i < 2 * n; ++i)
            {
                const double xx = y[i], yy = q.b0 * xx + d0;
                d0 = q.b1 * xx + d1 + q.a1 * yy; d1 = q.b2 * xx + q.a2 * yy; y[i] = yy;
                const float xf = y32[i], tq = fmaf(q.b1, xf, f1), u = q.b2 * xf, yf = fmaf(q.b0, xf, f0);
                f0 = fmaf(q.a1, yf, tq); f1 = fmaf(q.a2, yf, u); y32[i] = yf;
            }
        }
        double peak = 0, err = 0, e32 = 0;
        for (size_t i = 0; i < 2 * n; ++i) peak = std::max(peak, std::fabs(y[i]));
        for (int b = 0; b < 2; ++b) for (size_t i = 0; i < n; ++i)
        {
            err = std::max(err, std::fabs(hy[b * blk + c * n + i] - y[b * n + i]));
            e32 = std::max(e32, std::fabs(double(y32[b * n + i]) - y[b * n + i]));
        }
        printf("  channel %4u: gpu vs float64 %.2e, float32 recurrence vs float64 %.2e (relative to peak %.3g)\n", c, err / peak, e32 / peak, peak);
        worst = std::max(worst, err / peak); worst32 = std::max(worst32, e32 / peak);
    }
    printf("check: worst gpu error %.2e (float32 serial recurrence %.2e)%s\n", worst, worst32, (worst < 1e-3) ? "" : "  ** FAIL **");

    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < calls + 40; ++rep)
    {
        if (rep == 40) hipEventRecord(e0, nullptr);
        mi_biquad_bank_process(bank, out + size_t(rep % ring) * blk, in + size_t(rep % ring) * blk, n, n, n, nullptr);
    }
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000.0 / calls;
    printf("sections %u, %u ch x %zu: %.2f us per call back to back  => %.0f Msamples/s, %.3f of 8 TB/s\n", NS, C, n, us,
           double(blk) / us, double(blk) * 8.0 / (us * 1e-6) / 8e12);
    return (worst < 1e-3) ? 0 : 3;
}
