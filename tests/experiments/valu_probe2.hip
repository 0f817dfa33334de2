// Round-2 issue-rate probe for the biquad kernel's building blocks on gfx950:
//   plain / packed FMA chains, the TDF-II recurrence body (plain and packed, coefficients in SGPRs),
//   the zero-state weights pass, one DPP scan step (packed mov_dpp + pk_fma vs v_fmac_f32_dpp).
// Prints shader cycles per loop body per wave at 1, 2 and 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tests/experiments/valu_probe2.hip -o tests/experiments/valu_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float v2f __attribute__((ext_vector_type(2)));

enum { K_FMA, K_PKFMA, K_REC, K_REC_PK, K_W, K_W_PK, K_SCAN_PK, K_SCAN_DPP, K_SECTION, K_SECTION_PK };

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float a) { return v2f{a, a}; }

template <int CTRL>
__device__ __forceinline__ float dpp0(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// one plain-FMA scan step: e += M * shr(e), fused DPP operands (matrix in VGPRs)
#define FMAC_DPP(dst, src, m, ctrl) asm volatile("v_fmac_f32_dpp %0, %1, %2 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(dst) : "v"(src), "v"(m))

template <int KIND, int CH>
__global__ __launch_bounds__(256)      // (without it the 128-VGPR default spills the 4-chunk packed bodies: 25 and 53 registers)
void probe(float *out, unsigned long long *stamps, const float *coef, int iters)
{
    // coefficients: wave-uniform, kept in SGPRs
    float c[40];
    #pragma unroll
    for (int i = 0; i < 40; ++i) c[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(coef[i])));
    float x[CH][16];
    #pragma unroll
    for (int q = 0; q < CH; ++q)
        #pragma unroll
        for (int k = 0; k < 16; ++k) x[q][k] = threadIdx.x * 1e-3f + k + q;
    float d0[CH], d1[CH];
    for (int q = 0; q < CH; ++q) { d0[q] = 0.1f * q; d1[q] = 0.2f; }
    float mv[4] = { coef[threadIdx.x & 3], coef[4], coef[5], coef[6] };

    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it)
    {
        if (KIND == K_FMA)
        {
            #pragma unroll
            for (int r = 0; r < 16; ++r)
                #pragma unroll
                for (int q = 0; q < CH; ++q) d0[q] = fmaf(d0[q], c[0], c[1]);
        }
        if (KIND == K_PKFMA)
        {
            #pragma unroll
            for (int r = 0; r < 16; ++r)
                #pragma unroll
                for (int q = 0; q < CH; ++q)
                {
                    v2f a = pk_fma(v2f{d0[q], d1[q]}, v2f{c[0], c[1]}, v2f{c[2], c[3]});
                    d0[q] = a.x; d1[q] = a.y;
                }
        }
        if (KIND == K_REC || KIND == K_SECTION)        // TDF-II recurrence over 16 samples, CH independent chunks, plain FMA
        {
            if (KIND == K_SECTION)
            {
                // weights pass: (z, w) += (p_k, q_k) x_k
                float z[CH], w[CH];
                #pragma unroll
                for (int q = 0; q < CH; ++q) { z[q] = 0; w[q] = 0; }
                #pragma unroll
                for (int k = 0; k < 16; ++k)
                    #pragma unroll
                    for (int q = 0; q < CH; ++q) { z[q] = fmaf(c[8 + 2 * (k & 7)], x[q][k], z[q]); w[q] = fmaf(c[9 + 2 * (k & 7)], x[q][k], w[q]); }
                #pragma unroll
                for (int q = 0; q < CH; ++q) { d0[q] += z[q]; d1[q] += w[q]; }
            }
            #pragma unroll
            for (int k = 0; k < 16; ++k)
                #pragma unroll
                for (int q = 0; q < CH; ++q)
                {
                    const float xx = x[q][k];
                    const float tq = fmaf(c[1], xx, d1[q]);
                    const float u  = c[2] * xx;
                    const float y  = fmaf(c[0], xx, d0[q]);
                    d0[q] = fmaf(c[3], y, tq);
                    d1[q] = fmaf(c[4], y, u);
                    x[q][k] = y;
                }
        }
        if (KIND == K_REC_PK || KIND == K_SECTION_PK)  // the same, chunks paired into packed FMAs (CH must be even)
        {
            #pragma unroll
            for (int q = 0; q < CH; q += 2)
            {
                if (KIND == K_SECTION_PK)
                {
                    v2f a0 = splat(0), b0 = splat(0);
                    #pragma unroll
                    for (int k = 0; k < 16; ++k)
                    {
                        const v2f pq = v2f{c[8 + 2 * (k & 7)], c[9 + 2 * (k & 7)]};
                        a0 = pk_fma(pq, splat(x[q][k]), a0);
                        b0 = pk_fma(pq, splat(x[q + 1][k]), b0);
                    }
                    d0[q] += a0.x; d1[q] += a0.y; d0[q + 1] += b0.x; d1[q + 1] += b0.y;
                }
            }
            #pragma unroll
            for (int k = 0; k < 16; ++k)
                #pragma unroll
                for (int q = 0; q < CH; q += 2)
                {
                    const v2f xx = v2f{x[q][k], x[q + 1][k]};
                    v2f D0 = v2f{d0[q], d0[q + 1]}, D1 = v2f{d1[q], d1[q + 1]};
                    const v2f tq = pk_fma(splat(c[1]), xx, D1);
                    const v2f u  = splat(c[2]) * xx;
                    const v2f y  = pk_fma(splat(c[0]), xx, D0);
                    D0 = pk_fma(splat(c[3]), y, tq);
                    D1 = pk_fma(splat(c[4]), y, u);
                    x[q][k] = y.x; x[q + 1][k] = y.y;
                    d0[q] = D0.x; d0[q + 1] = D0.y; d1[q] = D1.x; d1[q + 1] = D1.y;
                }
        }
        if (KIND == K_W)
        {
            #pragma unroll
            for (int k = 0; k < 16; ++k)
                #pragma unroll
                for (int q = 0; q < CH; ++q) { d0[q] = fmaf(c[8 + 2 * (k & 7)], x[q][k], d0[q]); d1[q] = fmaf(c[9 + 2 * (k & 7)], x[q][k], d1[q]); }
        }
        if (KIND == K_W_PK)
        {
            #pragma unroll
            for (int k = 0; k < 16; ++k)
                #pragma unroll
                for (int q = 0; q < CH; ++q)
                {
                    v2f a = pk_fma(v2f{c[8 + 2 * (k & 7)], c[9 + 2 * (k & 7)]}, splat(x[q][k]), v2f{d0[q], d1[q]});
                    d0[q] = a.x; d1[q] = a.y;
                }
        }
        if (KIND == K_SCAN_PK)                          // 4 row steps of the 2x2 scan, as the round-1 kernel does them
        {
            #pragma unroll
            for (int q = 0; q < CH; ++q)
            {
                v2f e = v2f{d0[q], d1[q]};
                e = pk_fma(v2f{c[0], c[1]}, splat(dpp0<0x111>(e.x)), pk_fma(v2f{c[2], c[3]}, splat(dpp0<0x111>(e.y)), e));
                e = pk_fma(v2f{c[4], c[5]}, splat(dpp0<0x112>(e.x)), pk_fma(v2f{c[6], c[7]}, splat(dpp0<0x112>(e.y)), e));
                e = pk_fma(v2f{c[8], c[9]}, splat(dpp0<0x114>(e.x)), pk_fma(v2f{c[10], c[11]}, splat(dpp0<0x114>(e.y)), e));
                e = pk_fma(v2f{c[12], c[13]}, splat(dpp0<0x118>(e.x)), pk_fma(v2f{c[14], c[15]}, splat(dpp0<0x118>(e.y)), e));
                d0[q] = e.x; d1[q] = e.y;
            }
        }
        if (KIND == K_SCAN_DPP)                         // the same 4 steps with v_fmac_f32_dpp (matrix entries in VGPRs)
        {
            #pragma unroll
            for (int q = 0; q < CH; ++q)
            {
                float ex = d0[q], ey = d1[q], nx, ny;
                #define STEP(ctrl) \
                    nx = ex; ny = ey; \
                    asm volatile("s_nop 1"); \
                    FMAC_DPP(nx, ex, mv[0], ctrl); FMAC_DPP(ny, ex, mv[1], ctrl); \
                    FMAC_DPP(nx, ey, mv[2], ctrl); FMAC_DPP(ny, ey, mv[3], ctrl); \
                    ex = nx; ey = ny;
                STEP("row_shr:1") STEP("row_shr:2") STEP("row_shr:4") STEP("row_shr:8")
                #undef STEP
                d0[q] = ex; d1[q] = ey;
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int q = 0; q < CH; ++q) { s += d0[q] + d1[q]; for (int k = 0; k < 16; ++k) s += x[q][k]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int CH>
void run(const char *name, int instr_per_body)
{
    float *out, *coef; unsigned long long *st;
    const int max_blocks = 256 * 4, threads = 256;
    hipMalloc(&out, max_blocks * threads * sizeof(float)); hipMalloc(&st, max_blocks * 4 * sizeof(unsigned long long));
    std::vector<float> hc(40);
    for (int i = 0; i < 40; ++i) hc[i] = 0.01f * (i + 1);
    hc[0] = 0.2f; hc[1] = 0.4f; hc[2] = 0.2f; hc[3] = 0.5f; hc[4] = -0.3f;
    hipMalloc(&coef, 40 * sizeof(float)); hipMemcpy(coef, hc.data(), 40 * sizeof(float), hipMemcpyHostToDevice);
    const int iters = 400;
    printf("%-34s", name);
    for (int wps : {1, 2, 4})
    {
        const int blocks = 256 * wps;
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<KIND, CH>), dim3(blocks), dim3(threads), 0, 0, out, st, coef, iters);
        hipDeviceSynchronize();
        const int waves = blocks * threads / 64;
        std::vector<unsigned long long> h(waves);
        hipMemcpy(h.data(), st, waves * sizeof(h[0]), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double body = double(h[waves / 2]) / iters;
        printf("  w/SIMD %d: %7.1f cyc/body (%5.2f per instr, %5.2f per SIMD slot)", wps, body, body / instr_per_body, body / instr_per_body / wps);
    }
    printf("\n");
    hipFree(out); hipFree(st); hipFree(coef);
}

int main()
{
    run<K_FMA, 1>("v_fma_f32 1 chain x16", 16);
    run<K_FMA, 2>("v_fma_f32 2 chains x16", 32);
    run<K_FMA, 4>("v_fma_f32 4 chains x16", 64);
    run<K_PKFMA, 1>("v_pk_fma_f32 1 chain x16", 16);
    run<K_PKFMA, 2>("v_pk_fma_f32 2 chains x16", 32);
    run<K_PKFMA, 4>("v_pk_fma_f32 4 chains x16", 64);
    run<K_W, 1>("weights plain 1 chunk (32 fmac)", 32);
    run<K_W, 2>("weights plain 2 chunks (64 fmac)", 64);
    run<K_W_PK, 2>("weights packed 2 chunks (32 pk)", 32);
    run<K_W_PK, 4>("weights packed 4 chunks (64 pk)", 64);
    run<K_REC, 1>("recurrence plain 1 chunk (80)", 80);
    run<K_REC, 2>("recurrence plain 2 chunks (160)", 160);
    run<K_REC, 4>("recurrence plain 4 chunks (320)", 320);
    run<K_REC_PK, 2>("recurrence packed 2 chunks (80)", 80);
    run<K_REC_PK, 4>("recurrence packed 4 chunks (160)", 160);
    run<K_SECTION, 2>("weights+rec plain 2 chunks (224)", 224);
    run<K_SECTION, 4>("weights+rec plain 4 chunks (448)", 448);
    run<K_SECTION_PK, 2>("weights+rec packed 2 chunks (112)", 112);
    run<K_SECTION_PK, 4>("weights+rec packed 4 chunks (224)", 224);
    run<K_SCAN_PK, 1>("scan 4 row steps packed (16)", 16);
    run<K_SCAN_DPP, 1>("scan 4 row steps fmac_dpp (24)", 24);
    run<K_SCAN_PK, 2>("scan 4 row steps packed x2 (32)", 32);
    run<K_SCAN_DPP, 2>("scan 4 row steps fmac_dpp x2 (48)", 48);
    return 0;
}
