// CPU run of the index arithmetic of lsp-dsp-units_amd/csrc/fft16.h: all T threads of a workgroup in lock step, phase by
// phase (the barriers of fft16_regs are the boundaries between the loops below), against a double-precision DFT.
// Build (host code only runs; hipcc is used because the header speaks HIP):
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -I lsp-dsp-units_amd/csrc tests/cpp/fft16_host.cpp -o tests/cpp/fft16_host
#include "fft16.h"

#include <cmath>
#include <complex>
#include <cstdio>
#include <random>
#include <vector>

using namespace mi_fft16;
typedef std::complex<double> cd;

static void fft_ref(std::vector<cd> &a, bool inv)
{
    const size_t n = a.size();
    if (n == 1) return;
    std::vector<cd> e(n / 2), o(n / 2);
    for (size_t i = 0; i < n / 2; ++i) { e[i] = a[2 * i]; o[i] = a[2 * i + 1]; }
    fft_ref(e, inv); fft_ref(o, inv);
    for (size_t k = 0; k < n / 2; ++k)
    {
        const cd w = std::polar(1.0, (inv ? 2.0 : -2.0) * M_PI * double(k) / double(n)) * o[k];
        a[k] = e[k] + w; a[k + n / 2] = e[k] - w;
    }
}

template <int LOGN, bool INV, int I>
static void twiddled_stage(std::vector<std::array<v2f, 16>> &x, std::vector<float2> &img, const std::vector<tw16<LOGN>> &tw)
{
    using P = plan16<LOGN>;
    for (int t = 0; t < P::T; ++t)
    {
        v2f r[16];
        for (int u = 0; u < 16; ++u) r[u] = x[t][u];
        pass_twiddled<LOGN, INV, I>(r, tw[t]);
        exchange_store<LOGN, I>(img.data(), r, t);
    }
    for (int t = 0; t < P::T; ++t)
    {
        v2f r[16];
        exchange_load<LOGN>(img.data(), r, t);
        for (int u = 0; u < 16; ++u) x[t][u] = r[u];
    }
}

template <int LOGN, bool INV>
static double run(const std::vector<float> &table)
{
    using P = plan16<LOGN>;
    std::mt19937 rng(100 + LOGN + (INV ? 7 : 0));
    std::normal_distribution<float> g(0.0f, 1.0f);
    std::vector<float2> z(P::N);
    for (auto &v : z) v = make_float2(g(rng), g(rng));
    std::vector<cd> ref(P::N);
    for (int i = 0; i < P::N; ++i) ref[i] = cd(z[i].x, z[i].y);
    fft_ref(ref, INV);

    const float4 *tab = reinterpret_cast<const float4 *>(table.data()) + table16_offset(LOGN);
    std::vector<tw16<LOGN>> tw(P::T);
    for (int t = 0; t < P::T; ++t) load_tw16<LOGN>(tw[t], tab, t);
    std::vector<std::array<v2f, 16>> x(P::T);
    std::vector<float2> buf(P::LDS, make_float2(NAN, NAN));          // a cell read before it is written shows up as NaN
    for (int i = 0; i < P::N; ++i) buf[i] = z[i];
    for (int t = 0; t < P::T; ++t)
    {
        v2f r[16];
        natural_load<LOGN>(buf.data(), r, t);
        for (int u = 0; u < 16; ++u) x[t][u] = r[u];
    }
    std::vector<float2> img(P::LDS, make_float2(NAN, NAN));
    if (P::NTW >= 1) twiddled_stage<LOGN, INV, 0>(x, img, tw);
    if (P::NTW >= 2) twiddled_stage<LOGN, INV, (P::NTW >= 2) ? 1 : 0>(x, img, tw);
    if (P::NTW >= 3) twiddled_stage<LOGN, INV, (P::NTW >= 3) ? 2 : 0>(x, img, tw);
    // the image must have been written and read cell for cell: positions used = image_cell(0 .. N-1), all below LDS
    int top = 0;
    for (int pos = 0; pos < P::N; ++pos) top = std::max(top, image_cell(pos));
    if (top >= P::LDS) { printf("LOGN %d: image cell %d beyond the %d cells of the plan\n", LOGN, top, P::LDS); return 1.0; }
    std::vector<float2> out(P::N);
    for (int t = 0; t < P::T; ++t)
    {
        v2f r[16];
        for (int u = 0; u < 16; ++u) r[u] = x[t][u];
        pass_last<LOGN, INV>(r);
        natural_store<LOGN>(out.data(), r, t);
    }
    double err = 0.0, peak = 0.0;
    for (int i = 0; i < P::N; ++i)
    {
        peak = std::max(peak, std::abs(ref[i]));
        const double e = std::abs(cd(out[i].x, out[i].y) - ref[i]);
        err = std::max(err, std::isfinite(e) ? e : 1e30);
    }
    printf("LOGN %2d %s: T %3d, passes %d, max error %.3e of peak %.3e (relative %.2e)\n", LOGN, INV ? "inverse" : "forward",
           P::T, P::NP, err, peak, err / peak);
    return err / peak;
}

int main()
{
    std::vector<float> table(4 * size_t(table16_total()));
    table16_build(table.data());
    double worst = 0.0;
    worst = std::max(worst, run<10, false>(table)); worst = std::max(worst, run<10, true>(table));
    worst = std::max(worst, run<11, false>(table)); worst = std::max(worst, run<11, true>(table));
    worst = std::max(worst, run<12, false>(table)); worst = std::max(worst, run<12, true>(table));
    worst = std::max(worst, run<13, false>(table)); worst = std::max(worst, run<13, true>(table));
    printf("table: %d float4 entries; worst relative error %.2e\n", table16_total(), worst);
    return (worst < 2e-6) ? 0 : 1;
}
