// Experiment (round 3, VERDICT r2 item 1, second paragraph): the biquad cascade with FOUR chunks of 16 samples per lane --
// one wave per channel, one 4096-sample tile, no hand-off between waves, one cross-lane scan per section instead of two --
// next to the product's <16,2> kernel (two waves per channel, two chunks per lane) on the C2 shape.
// The probe is the real arithmetic (weights pass, composition, DPP scan with powers of P^4, start states, exact recurrence,
// transposed loads and stores); its output is checked against the product's.
//   hipcc -O3 --offload-arch=gfx950 -Iinclude -Ilsp-dsp-units_amd/csrc -Ilsp-dsp-units_amd/include
//         tests/experiments/biquad_quad_probe.hip -Llsp-dsp-units_amd -lmi_dspu -Wl,-rpath,$PWD/lsp-dsp-units_amd
#include "../../lsp-dsp-units_amd/csrc/biquad.hip"
#include <cstdio>
#include <cmath>

namespace
{
    constexpr int QL = 16, QTAB = 144;      // per section: 8 coefs | 8 matrices x 4 | 16 x (p, q) | 16 lanes x (P^4)^(i+1)
    // [0..4] b0 b1 b2 a1 a2 | [8..39] P P^2 P^4 P^8 P^16 P^32 P^64 P^3 | [40..71] pq | [72..135] ql4

    void quad_row(float *row, const float *q)
    {
        const double b0 = q[0], b1 = q[1], b2 = q[2], a1 = q[3], a2 = q[4];
        for (int i = 0; i < 8; ++i)
            row[i] = (i < 5) ? q[i] : 0.0f;
        const mat2 A = { a1, 1.0, a2, 0.0 };
        double v0 = b1 + a1 * b0, v1 = b2 + a2 * b0;
        float *pq = row + 40;
        for (int k = QL - 1; k >= 0; --k)
        {
            pq[2 * k] = float(v0); pq[2 * k + 1] = float(v1);
            const double n0 = A.a * v0 + A.b * v1, n1 = A.c * v0 + A.d * v1;
            v0 = n0; v1 = n1;
        }
        const auto put = [](float *m, const mat2 &M) { m[0] = float(M.a); m[1] = float(M.c); m[2] = float(M.b); m[3] = float(M.d); };
        mat2 P = { 1.0, 0.0, 0.0, 1.0 };
        for (int k = 0; k < QL; ++k)
            P = mul(P, A);
        mat2 M = P;
        for (int i = 0; i < 7; ++i)             // P, P^2, P^4, P^8, P^16, P^32, P^64
        {
            put(row + 8 + 4 * i, M);
            M = mul(M, M);
        }
        const mat2 P2 = mul(P, P), P4 = mul(P2, P2);
        put(row + 36, mul(P2, P));
        mat2 Qi = P4;
        for (int i = 0; i < 16; ++i)
        {
            put(row + 72 + 4 * i, Qi);
            Qi = mul(Qi, P4);
        }
    }

    __global__ __launch_bounds__(64)
    void biquad_quad_kernel(float *out, const float *in, size_t stride, int n /* 4096 */, const float *tab, float *state, int nsec)
    {
        constexpr int W = 64, PITCH = W + 4, LPT = W / 4;
        __shared__ __attribute__((aligned(16))) float sx[64 * PITCH];
        const int ch = blockIdx.x, t = threadIdx.x, l16 = t & 15;
        const bool lane0 = (t == 0), row3 = (t >= 48);
        const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in + size_t(ch) * stride), 0, n * 4, BUFFER_DWORD3);
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out + size_t(ch) * stride, 0, n * 4, BUFFER_DWORD3);
        float4 ld[LPT];
        #pragma unroll
        for (int k = 0; k < LPT; ++k)
        {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(irsrc, (4 * (k * 64 + t)) * 4, 0, 0);
            ld[k] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
        #pragma unroll
        for (int k = 0; k < LPT; ++k)
        {
            const int i = 4 * (k * 64 + t);
            *reinterpret_cast<float4 *>(&sx[i + (i / W) * 4]) = ld[k];
        }
        __builtin_amdgcn_wave_barrier();
        v2f xa[QL], xb[QL];                     // xa: chunks A (x) and B (y) of the lane, xb: chunks C and D
        #pragma unroll
        for (int k = 0; k < QL / 4; ++k)
        {
            const float4 a = *reinterpret_cast<const float4 *>(&sx[t * PITCH + 4 * k]);
            const float4 b = *reinterpret_cast<const float4 *>(&sx[t * PITCH + 16 + 4 * k]);
            const float4 c = *reinterpret_cast<const float4 *>(&sx[t * PITCH + 32 + 4 * k]);
            const float4 d = *reinterpret_cast<const float4 *>(&sx[t * PITCH + 48 + 4 * k]);
            xa[4 * k + 0] = v2f{a.x, b.x}; xa[4 * k + 1] = v2f{a.y, b.y}; xa[4 * k + 2] = v2f{a.z, b.z}; xa[4 * k + 3] = v2f{a.w, b.w};
            xb[4 * k + 0] = v2f{c.x, d.x}; xb[4 * k + 1] = v2f{c.y, d.y}; xb[4 * k + 2] = v2f{c.z, d.z}; xb[4 * k + 3] = v2f{c.w, d.w};
        }
        typedef const __attribute__((address_space(4))) float cfloat;
        typedef const __attribute__((address_space(4))) v16f cv16f;
        typedef const __attribute__((address_space(4))) v8f cv8f;
        float2 *st = reinterpret_cast<float2 *>(state + size_t(ch) * nsec * 2);
        // the section's table: wave-uniform parts in SGPRs (constant address space), reloaded for the NEXT section as soon as
        // this one is done with them -- as the product kernel does -- so that the scalar loads fly under the recurrence
        v8f cf; v16f m0, m1, pq0, pq1; float4 ql;
        auto load_tab = [&](int s, bool coefs, bool rest)
        {
            const float *T = tab + (size_t(ch) * nsec + s) * QTAB;
            cfloat *U = reinterpret_cast<cfloat *>(reinterpret_cast<uint64_t>(T));
            if (coefs)
                cf = *reinterpret_cast<cv8f *>(U);
            if (rest)
            {
                m0 = *reinterpret_cast<cv16f *>(U + 8); m1 = *reinterpret_cast<cv16f *>(U + 24);
                pq0 = *reinterpret_cast<cv16f *>(U + 40); pq1 = *reinterpret_cast<cv16f *>(U + 56);
                ql = *reinterpret_cast<const float4 *>(T + 72 + 4 * l16);
            }
        };
        load_tab(0, true, true);
        for (int s = 0; s < nsec; ++s)
        {
            // 1. zero-state end states of the four chunks
            v2f a0 = splat(0.0f), a1_ = splat(0.0f), b0_ = splat(0.0f), b1_ = splat(0.0f);
            v2f c0 = splat(0.0f), c1_ = splat(0.0f), d0_ = splat(0.0f), d1_ = splat(0.0f);
            #pragma unroll
            for (int k = 0; k < QL; k += 2)
            {
                const v16f &r = (k < 8) ? pq0 : pq1;
                const v2f w0 = v2f{r[(2 * k) % 16], r[(2 * k + 1) % 16]}, w1 = v2f{r[(2 * k + 2) % 16], r[(2 * k + 3) % 16]};
                a0 = pk_fma(w0, splat(xa[k].x), a0);      b0_ = pk_fma(w0, splat(xa[k].y), b0_);
                a1_ = pk_fma(w1, splat(xa[k + 1].x), a1_); b1_ = pk_fma(w1, splat(xa[k + 1].y), b1_);
                c0 = pk_fma(w0, splat(xb[k].x), c0);      d0_ = pk_fma(w0, splat(xb[k].y), d0_);
                c1_ = pk_fma(w1, splat(xb[k + 1].x), c1_); d1_ = pk_fma(w1, splat(xb[k + 1].y), d1_);
            }
            const v2f zA = a0 + a1_, zB = b0_ + b1_, zC = c0 + c1_, zD = d0_ + d1_;
            const v2f P0 = v2f{m0[0], m0[1]}, P1 = v2f{m0[2], m0[3]};          // P
            const v2f Q0 = v2f{m0[4], m0[5]}, Q1 = v2f{m0[6], m0[7]};          // P^2
            const v2f R0 = v2f{m0[8], m0[9]}, R1 = v2f{m0[10], m0[11]};        // P^4
            // 2. end state of the quad for a zero start: P^2 (P zA + zB) + (P zC + zD)
            const v2f e2 = mat_fma(P0, P1, zA, zB), e4 = mat_fma(P0, P1, zC, zD);
            v2f e = mat_fma(Q0, Q1, e2, e4);
            const float2 cs = st[s];
            const v2f cvec = lane0 ? v2f{cs.x, cs.y} : splat(0.0f);
            e = mat_fma(R0, R1, cvec, e);
            const v2f zero = splat(0.0f);
            e = mat_fma(R0, R1, dpp_zero<DPP_ROW_SHR1>(e), e);
            e = mat_fma(v2f{m0[12], m0[13]}, v2f{m0[14], m0[15]}, dpp_zero<DPP_ROW_SHR2>(e), e);    // P^8
            e = mat_fma(v2f{m1[0], m1[1]}, v2f{m1[2], m1[3]}, dpp_zero<DPP_ROW_SHR4>(e), e);        // P^16
            e = mat_fma(v2f{m1[4], m1[5]}, v2f{m1[6], m1[7]}, dpp_zero<DPP_ROW_SHR8>(e), e);        // P^32
            const v2f QLc0 = v2f{ql.x, ql.y}, QLc1 = v2f{ql.z, ql.w};
            e = mat_fma(QLc0, QLc1, dpp_or<DPP_ROW_BCAST15, 0xa>(zero, e), e);
            {
                const v2f sv  = dpp_or<DPP_ROW_BCAST31, 0xc>(zero, e);
                const v2f s2 = mat_fma(v2f{m1[8], m1[9]}, v2f{m1[10], m1[11]}, sv, zero);           // P^64
                e = mat_fma(QLc0, QLc1, row3 ? s2 : sv, e);
            }
            // 3. start states of the four chunks
            const v2f S  = dpp_or<DPP_WAVE_SHR1, 0xf>(cvec, e);
            const v2f SB = mat_fma(P0, P1, S, zA), SC = mat_fma(P0, P1, SB, zB), SD = mat_fma(P0, P1, SC, zC);
            v2f d0 = v2f{S.x, SB.x}, d1 = v2f{S.y, SB.y}, g0 = v2f{SC.x, SD.x}, g1 = v2f{SC.y, SD.y};
            const v2f b0 = splat(cf[0]), b1 = splat(cf[1]), b2 = splat(cf[2]), a1 = splat(cf[3]), a2 = splat(cf[4]);
            __builtin_amdgcn_sched_barrier(0);
            load_tab((s + 1 < nsec) ? s + 1 : 0, false, true);
            __builtin_amdgcn_sched_barrier(0);
            #pragma unroll
            for (int k = 0; k < QL; ++k)
            {
                const v2f xx = xa[k], yy = xb[k];
                const v2f tq = pk_fma(b1, xx, d1), tr = pk_fma(b1, yy, g1);
                const v2f u  = b2 * xx, w = b2 * yy;
                const v2f y  = pk_fma(b0, xx, d0), z = pk_fma(b0, yy, g0);
                d0 = pk_fma(a1, y, tq); g0 = pk_fma(a1, z, tr);
                d1 = pk_fma(a2, y, u);  g1 = pk_fma(a2, z, w);
                xa[k] = y; xb[k] = z;
            }
            __builtin_amdgcn_sched_barrier(0);
            load_tab((s + 1 < nsec) ? s + 1 : 0, true, false);
            if (t == 63)
                st[s] = make_float2(g0.y, g1.y);        // (the tile ends with lane 63's chunk D: n is a whole tile)
        }
        #pragma unroll
        for (int k = 0; k < QL / 4; ++k)
        {
            *reinterpret_cast<float4 *>(&sx[t * PITCH + 4 * k])      = make_float4(xa[4 * k].x, xa[4 * k + 1].x, xa[4 * k + 2].x, xa[4 * k + 3].x);
            *reinterpret_cast<float4 *>(&sx[t * PITCH + 16 + 4 * k]) = make_float4(xa[4 * k].y, xa[4 * k + 1].y, xa[4 * k + 2].y, xa[4 * k + 3].y);
            *reinterpret_cast<float4 *>(&sx[t * PITCH + 32 + 4 * k]) = make_float4(xb[4 * k].x, xb[4 * k + 1].x, xb[4 * k + 2].x, xb[4 * k + 3].x);
            *reinterpret_cast<float4 *>(&sx[t * PITCH + 48 + 4 * k]) = make_float4(xb[4 * k].y, xb[4 * k + 1].y, xb[4 * k + 2].y, xb[4 * k + 3].y);
        }
        __builtin_amdgcn_wave_barrier();
        #pragma unroll
        for (int k = 0; k < LPT; ++k)
        {
            const int i = 4 * (k * 64 + t);
            store_through(orsrc, i, *reinterpret_cast<const float4 *>(&sx[i + (i / W) * 4]));
        }
    }
}

int main()
{
    const int C = 1024, n = 4096, NS = 8, K = 400;
    std::vector<float> hx(size_t(C) * n), coef(size_t(C) * NS * 5), tab(size_t(C) * NS * QTAB);
    uint64_t sd = 88172645463325252ull;
    auto rnd = [&]() { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; return double(sd >> 11) / 9007199254740992.0; };
    for (float &v : hx) v = float(rnd() - 0.5);
    mi_biquad_bank_t *bank = nullptr;
    if (mi_biquad_bank_create(&bank, C, NS) != 0) { printf("no device\n"); return 1; }
    for (int c = 0; c < C; ++c)
    {
        mi_biquad_x1_t sec[NS];
        for (int s = 0; s < NS; ++s)
        {
            // a bell around 1 kHz .. 8 kHz at 48 kHz, radius 0.9 .. 0.98
            const double w = 2.0 * M_PI * (1000.0 + 7000.0 * rnd()) / 48000.0, r = 0.9 + 0.08 * rnd(), g = 0.5 + rnd();
            const double a1 = 2.0 * r * cos(w), a2 = -r * r;
            float *q = &coef[(size_t(c) * NS + s) * 5];
            q[0] = float(g); q[1] = float(-g * a1 * 0.9); q[2] = float(-g * a2 * 0.8); q[3] = float(a1); q[4] = float(a2);
            sec[s].b0 = q[0]; sec[s].b1 = q[1]; sec[s].b2 = q[2]; sec[s].a1 = q[3]; sec[s].a2 = q[4]; sec[s].p0 = sec[s].p1 = sec[s].p2 = 0.0f;
            quad_row(&tab[(size_t(c) * NS + s) * QTAB], q);
        }
        mi_biquad_bank_set_chains(bank, c, sec, NS, 1);
    }
    float *dx, *dy, *dz, *dtab, *dstate;
    hipMalloc(&dx, hx.size() * 4); hipMalloc(&dy, hx.size() * 4); hipMalloc(&dz, hx.size() * 4);
    hipMalloc(&dtab, tab.size() * 4); hipMalloc(&dstate, size_t(C) * NS * 2 * 4);
    hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dstate, 0, size_t(C) * NS * 2 * 4);
    hipStream_t st; hipStreamCreate(&st);
    // correctness: two consecutive blocks (carried state) through both
    std::vector<float> y1(hx.size()), y2(hx.size());
    for (int rep = 0; rep < 2; ++rep)
    {
        mi_biquad_bank_process(bank, dy, dx, n, n, n, st);
        hipLaunchKernelGGL(biquad_quad_kernel, dim3(C), dim3(64), 0, st, dz, dx, size_t(n), n, dtab, dstate, NS);
    }
    hipStreamSynchronize(st);
    hipMemcpy(y1.data(), dy, hx.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(y2.data(), dz, hx.size() * 4, hipMemcpyDeviceToHost);
    double peak = 0, err = 0;
    for (size_t i = 0; i < y1.size(); ++i) { peak = fmax(peak, fabs(y1[i])); err = fmax(err, fabs(double(y1[i]) - y2[i])); }
    printf("second block: |quad - product| = %.3g of the peak %.3g\n", err / peak, peak);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 4; ++which)
    {
        for (int i = 0; i < 20; ++i)
            if (which & 1) hipLaunchKernelGGL(biquad_quad_kernel, dim3(C), dim3(64), 0, st, dz, dx, size_t(n), n, dtab, dstate, NS);
            else mi_biquad_bank_process(bank, dy, dx, n, n, n, st);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int i = 0; i < K; ++i)
            if (which & 1) hipLaunchKernelGGL(biquad_quad_kernel, dim3(C), dim3(64), 0, st, dz, dx, size_t(n), n, dtab, dstate, NS);
            else mi_biquad_bank_process(bank, dy, dx, n, n, n, st);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-46s %6.2f us per call\n", (which & 1) ? "four chunks per lane, one wave per channel" : "product <16,2>: two waves, two chunks per lane", ms / K * 1e3);
    }
    return 0;
}
