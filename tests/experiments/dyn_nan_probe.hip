// probe: the sections the device builds for one matched-Z type, beside the host's (same source, both compilers).
// Built WITH the SLP vectorizer (the default) the per-sample loop of probe_loop returns NaN numerators for
// FLT_MT_BWC_HIPASS while the straight-line kernel and the separately evaluated intermediates are right; built with
// -fno-slp-vectorize (what the Makefile does for dynfilter.hip) the loop is right as well:
//   hipcc -O3 --offload-arch=gfx950 -Iinclude -Ilsp-dsp-units_amd/csrc -Ilsp-dsp-units_amd/include [-fno-slp-vectorize]
//         tests/experiments/dyn_nan_probe.hip -Llsp-dsp-units_amd -lmi_dspu
#include "../../lsp-dsp-units_amd/csrc/dynfilter.hip"
#include <cstdio>
__global__ void probe(dyn_filter f, float g, float *out)
{
    const cascade c = dyn_cascade(f.p, 0, g);
    const section5 s = dyn_section(f, 0, g);
    out[0] = c.t[0]; out[1] = c.t[1]; out[2] = c.t[2]; out[3] = c.b[0]; out[4] = c.b[1]; out[5] = c.b[2];
    out[6] = s.b0; out[7] = s.b1; out[8] = s.b2; out[9] = s.a1; out[10] = s.a2;
    float P[3];
    matched_poly(c.t, f.f0, f.kf, P); out[11] = P[0]; out[12] = P[1]; out[13] = P[2];
    matched_poly(c.b, f.f0, f.kf, P); out[14] = P[0]; out[15] = P[1]; out[16] = P[2];
}
__global__ void probe_loop(dyn_filter f, const float *g, float *out, uint32_t nc)
{
    __shared__ float gs[8][64];
    for (int k = 0; k < 8; ++k) gs[k][threadIdx.x] = g[k];
    for (uint32_t J = 0; J < nc; ++J)
    {
        #pragma unroll 1
        for (int k = 0; k < 8; ++k)
        {
            const section5 s = dyn_section(f, J, gs[k][threadIdx.x]);
            out[k * 5 + 0] = s.b0; out[k * 5 + 1] = s.b1; out[k * 5 + 2] = s.b2; out[k * 5 + 3] = s.a1; out[k * 5 + 4] = s.a2;
            if (k == 0)
            {
                const cascade c = dyn_cascade(f.p, J, gs[k][threadIdx.x]);
                const double w = 0.1 * double(f.f0) * double(f.kf);
                const matched_side n = matched_one(c.t, f.f0, f.kf, w), d = matched_one(c.b, f.f0, f.kf, w);
                out[40] = n.P[0]; out[41] = n.P[1]; out[42] = n.P[2]; out[43] = float(n.A); out[44] = float(n.I);
                out[45] = d.P[0]; out[46] = d.P[1]; out[47] = d.P[2]; out[48] = float(d.A); out[49] = float(d.I);
                out[50] = c.t[0]; out[51] = c.t[1]; out[52] = c.t[2];
            }
        }
    }
}
int main()
{
    for (uint32_t type : { 30u, 32u, 4u })
    {
        dyn_filter f;
        f.p.base = base_type(type); f.p.slope = 2; f.p.xf = 1.0f; f.p.Q = 0.6f;
        f.nc = cascade_count(type, 2); f.bilinear = 0; f.kf = kTwoPi / 48000.0f; f.f0 = 1200.0f;
        float *d; hipMalloc(&d, 17 * 4);
        hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, 0, f, 1.3f, d);
        float h[17]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        const cascade c = dyn_cascade(f.p, 0, 1.3f);
        const section5 s = dyn_section(f, 0, 1.3f);
        printf("type %u\n dev t %g %g %g b %g %g %g | sec %g %g %g %g %g | Pt %g %g %g Pb %g %g %g\n", type, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15], h[16]);
        {
            float *dg, *dl; hipMalloc(&dg, 32); hipMalloc(&dl, 256);
            float hg[8] = { 1.8f, 1.8f, 0.5f, 1.0f, 2.0f, 3.0f, 0.3f, 1.1f }, hl[64];
            hipMemcpy(dg, hg, 32, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(probe_loop, dim3(1), dim3(64), 0, 0, f, dg, dl, f.nc);
            hipMemcpy(hl, dl, 256, hipMemcpyDeviceToHost);
            printf("  in loop: Pn %g %g %g A %g I %g | Pd %g %g %g A %g I %g | t %g %g %g\n", hl[40],hl[41],hl[42],hl[43],hl[44],hl[45],hl[46],hl[47],hl[48],hl[49],hl[50],hl[51],hl[52]);
            for (int k = 0; k < 8; ++k) printf("  loop g %g: %g %g %g %g %g\n", hg[k], hl[k*5], hl[k*5+1], hl[k*5+2], hl[k*5+3], hl[k*5+4]);
        }
        printf(" host t %g %g %g b %g %g %g | sec %g %g %g %g %g\n", c.t[0], c.t[1], c.t[2], c.b[0], c.b[1], c.b[2], s.b0, s.b1, s.b2, s.a1, s.a2);
    }
    return 0;
}
