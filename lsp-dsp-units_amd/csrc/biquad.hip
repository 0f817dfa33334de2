// Biquad cascade bank for gfx950: the GPU side of lsp::dspu::FilterBank::process
// (reference: src/main/filters/FilterBank.cpp:256-291, which calls
// dsp::biquad_process_x8/x4/x2/x1 of lsp-dsp-lib once per packed bank).
//
// Why this is not "one channel per lane, serial in time": a block of N samples
// through one section is a chain of 2N dependent FMAs; at N = 4096 that chain
// alone is longer than the whole HBM budget of the block.  The kernel therefore
// cuts every channel's block into chunks of L samples, one chunk per lane, and
// runs every section in three steps (state s = {d0,d1}, s' = A s + B x):
//
//   1. zero-state response of the chunk's end state:  z = sum_k A^(L-1-k) B x[k]
//      -> two dot products with per-section tables p[],q[] (wave-uniform, SGPR);
//   2. inclusive scan over chunks  E_t = P E_(t-1) + z_t  with P = A^L
//      (Hillis-Steele, level j uses P^(2^j), also wave-uniform); the state the
//      channel carried in from the previous call enters at chunk 0;
//   3. the exact TDF-II recurrence over the chunk, started from E_(t-1):
//         y = b0 x + d0;  d0 = (b1 x + d1) + a1 y;  d1 = b2 x + a2 y
//      -- the reference's own per-sample arithmetic; only the chunk start state
//      carries the (float32 round-off sized) difference of steps 1-2.
//
// All sections of a channel run back to back on samples held in registers, so
// HBM sees each sample once in and once out (8 B per channel-sample).
// A workgroup owns one channel: it loads the block with coalesced 16-B loads,
// transposes it through a padded LDS tile so each lane gets its L consecutive
// samples (conflict-free ds_read_b128: row pitch L+4 dwords, (L+4)/4 odd), and
// stores the result the same way back.
#include "mi_common.h"

#include <cmath>
#include <cstdint>
#include <vector>

namespace
{
    constexpr int ilog2(int v) { return (v <= 1) ? 0 : 1 + ilog2(v >> 1); }

    template <int L, int NT>
    struct geom
    {
        static constexpr int NLEV   = ilog2(NT);            // scan levels
        static constexpr int TAB    = 8 + 4 * NLEV + 2 * L; // floats per (channel, section)
        static constexpr int PITCH  = L + 4;                // LDS dwords per chunk
        static constexpr int BLOCK  = L * NT;               // samples per launch and channel
        static_assert(((PITCH / 4) & 1) == 1, "LDS pitch must be an odd number of 16-B slots");
        static_assert((TAB % 4) == 0, "table rows stay 16-B aligned");
    };

    // Table row of one section (all wave-uniform):
    //   [0..4]  b0 b1 b2 a1 a2          [5..7] unused
    //   [8 + 4j ..]  P^(2^j) row-major, j = 0..NLEV-1
    //   then p[L], q[L]
    template <int L, int NT, bool ALIGNED, bool FULL>
    __global__ __launch_bounds__(NT)
    void biquad_bank_kernel(float *out, const float *in, size_t out_stride, size_t in_stride,
                            int cnt, const float *__restrict__ tab, float *state,
                            const uint32_t *__restrict__ nsec, int max_sec)
    {
        using G = geom<L, NT>;
        constexpr int NLEV = G::NLEV;
        constexpr int TAB  = G::TAB;
        constexpr int PITCH = G::PITCH;
        constexpr int WLEV = (NLEV < 6) ? NLEV : 6;         // levels that stay inside a wave

        __shared__ __attribute__((aligned(16))) float sx[NT * PITCH];
        __shared__ float2 sscan[(NT > 64) ? 2 * NT : 2];

        const int ch    = blockIdx.x;
        const int t     = threadIdx.x;
        const int lane  = t & 63;
        const int ns    = int(nsec[ch]);
        const float *xin = in + size_t(ch) * in_stride;
        float *yout      = out + size_t(ch) * out_stride;

        // ---- coalesced load, transposed through LDS -------------------------------------
        float x[L];
        #pragma unroll
        for (int k = 0; k < L / 4; ++k)
        {
            const int i = 4 * (k * NT + t);
            float4 v;
            if (ALIGNED && (FULL || i + 4 <= cnt))
                v = *reinterpret_cast<const float4 *>(xin + i);
            else
            {
                v.x = (i + 0 < cnt) ? xin[i + 0] : 0.0f;
                v.y = (i + 1 < cnt) ? xin[i + 1] : 0.0f;
                v.z = (i + 2 < cnt) ? xin[i + 2] : 0.0f;
                v.w = (i + 3 < cnt) ? xin[i + 3] : 0.0f;
            }
            *reinterpret_cast<float4 *>(&sx[i + (i / L) * 4]) = v;
        }
        __syncthreads();
        #pragma unroll
        for (int k = 0; k < L / 4; ++k)
        {
            const float4 v = *reinterpret_cast<const float4 *>(&sx[t * PITCH + 4 * k]);
            x[4 * k + 0] = v.x; x[4 * k + 1] = v.y; x[4 * k + 2] = v.z; x[4 * k + 3] = v.w;
        }

        // Lane that owns the last valid sample of the block, and how many it owns.
        const int t_last = FULL ? (NT - 1) : ((cnt - 1) / L);
        const int m_last = FULL ? L : (cnt - t_last * L);

        // ---- sections, strictly in series (FilterBank.cpp:267-290) ----------------------
        for (int s = 0; s < ns; ++s)
        {
            const float *T  = tab + (size_t(ch) * max_sec + s) * TAB;
            const float *M  = T + 8;
            const float *P  = T + 8 + 4 * NLEV;
            const float *Q  = P + L;
            float *st       = state + (size_t(ch) * max_sec + s) * 2;
            const float b0 = T[0], b1 = T[1], b2 = T[2], a1 = T[3], a2 = T[4];
            const float c0 = st[0], c1 = st[1];          // state carried in from the previous call

            // 1. end state of the chunk for zero start state
            float z0 = 0.0f, z1 = 0.0f, w0 = 0.0f, w1 = 0.0f;
            #pragma unroll
            for (int k = 0; k < L; k += 2)
            {
                z0 = fmaf(P[k], x[k], z0);
                w0 = fmaf(Q[k], x[k], w0);
                z1 = fmaf(P[k + 1], x[k + 1], z1);
                w1 = fmaf(Q[k + 1], x[k + 1], w1);
            }
            float z = z0 + z1, w = w0 + w1;
            if (t == 0)
            {
                z = fmaf(M[0], c0, fmaf(M[1], c1, z));
                w = fmaf(M[2], c0, fmaf(M[3], c1, w));
            }

            // 2. inclusive scan over chunks: E_t += P^(2^j) E_(t - 2^j)
            #pragma unroll
            for (int j = 0; j < WLEV; ++j)
            {
                const int d = 1 << j;
                const float zs = __shfl_up(z, d, 64);
                const float ws = __shfl_up(w, d, 64);
                if (lane >= d)
                {
                    z = fmaf(M[4 * j + 0], zs, fmaf(M[4 * j + 1], ws, z));
                    w = fmaf(M[4 * j + 2], zs, fmaf(M[4 * j + 3], ws, w));
                }
            }
            float d0 = __shfl_up(z, 1, 64);
            float d1 = __shfl_up(w, 1, 64);
            if (NT > 64)
            {
                float2 *sc = sscan + (s & 1) * NT;
                sc[t] = make_float2(z, w);
                __syncthreads();
                #pragma unroll
                for (int j = 6; j < NLEV; ++j)
                {
                    const int d = 1 << j;
                    if (t >= d)
                    {
                        const float2 e = sc[t - d];
                        z = fmaf(M[4 * j + 0], e.x, fmaf(M[4 * j + 1], e.y, z));
                        w = fmaf(M[4 * j + 2], e.x, fmaf(M[4 * j + 3], e.y, w));
                    }
                    if (NLEV > 7)       // more than two waves: republish before the next level / the hand-over
                    {
                        __syncthreads();
                        sc[t] = make_float2(z, w);
                        __syncthreads();
                    }
                }
                // hand E_(t-1) to the first lane of every wave but the first (for two waves
                // the first wave's entries were final when they were published)
                d0 = __shfl_up(z, 1, 64);
                d1 = __shfl_up(w, 1, 64);
                if (lane == 0 && t != 0)
                {
                    const float2 e = sc[t - 1];
                    d0 = e.x;
                    d1 = e.y;
                }
            }
            if (t == 0)
            {
                d0 = c0;
                d1 = c1;
            }

            // 3. exact recurrence over the chunk
            float f0 = d0, f1 = d1;
            #pragma unroll
            for (int k = 0; k < L; ++k)
            {
                const float xx = x[k];
                const float y  = fmaf(b0, xx, d0);
                const float tt = fmaf(b1, xx, d1);
                d0   = fmaf(a1, y, tt);
                d1   = fmaf(a2, y, b2 * xx);
                x[k] = y;
                if (!FULL && (k + 1 == m_last))
                {
                    f0 = d0;
                    f1 = d1;
                }
            }
            if (FULL)
            {
                f0 = d0;
                f1 = d1;
            }
            if (t == t_last)
            {
                st[0] = f0;
                st[1] = f1;
            }
        }

        // ---- transposed back through LDS, coalesced store -------------------------------
        __syncthreads();
        #pragma unroll
        for (int k = 0; k < L / 4; ++k)
            *reinterpret_cast<float4 *>(&sx[t * PITCH + 4 * k]) =
                make_float4(x[4 * k + 0], x[4 * k + 1], x[4 * k + 2], x[4 * k + 3]);
        __syncthreads();
        #pragma unroll
        for (int k = 0; k < L / 4; ++k)
        {
            const int i = 4 * (k * NT + t);
            const float4 v = *reinterpret_cast<const float4 *>(&sx[i + (i / L) * 4]);
            if (ALIGNED && (FULL || i + 4 <= cnt))
                *reinterpret_cast<float4 *>(yout + i) = v;
            else
            {
                if (i + 0 < cnt) yout[i + 0] = v.x;
                if (i + 1 < cnt) yout[i + 1] = v.y;
                if (i + 2 < cnt) yout[i + 2] = v.z;
                if (i + 3 < cnt) yout[i + 3] = v.w;
            }
        }
    }

    __global__ void impulse_kernel(float *out, size_t stride, size_t samples, uint32_t channels)
    {
        // FilterBank.cpp:316-318: zero the buffer, out[0] = 1
        const size_t total = size_t(channels) * samples;
        for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
             i += size_t(gridDim.x) * blockDim.x)
        {
            const size_t c = i / samples, k = i - c * samples;
            out[c * stride + k] = (k == 0) ? 1.0f : 0.0f;
        }
    }

    // ---- host-side tables -------------------------------------------------------------------
    struct mat2 { double a, b, c, d; };
    inline mat2 mul(const mat2 &x, const mat2 &y)
    {
        return { x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d,
                 x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d };
    }

    template <int L, int NT>
    void fill_row(float *row, const float *q /* b0 b1 b2 a1 a2 */)
    {
        using G = geom<L, NT>;
        const double b0 = q[0], b1 = q[1], b2 = q[2], a1 = q[3], a2 = q[4];
        for (int i = 0; i < 8; ++i)
            row[i] = (i < 5) ? q[i] : 0.0f;
        // s' = A s + B x  with  A = [a1 1; a2 0],  B = [b1 + a1 b0, b2 + a2 b0]
        const mat2 A = { a1, 1.0, a2, 0.0 };
        double v0 = b1 + a1 * b0, v1 = b2 + a2 * b0;
        float *p = row + 8 + 4 * G::NLEV, *qq = p + L;
        for (int k = L - 1; k >= 0; --k)        // p[k],q[k] = A^(L-1-k) B
        {
            p[k]  = float(v0);
            qq[k] = float(v1);
            const double n0 = A.a * v0 + A.b * v1, n1 = A.c * v0 + A.d * v1;
            v0 = n0;
            v1 = n1;
        }
        mat2 Pm = { 1.0, 0.0, 0.0, 1.0 };
        for (int k = 0; k < L; ++k)
            Pm = mul(Pm, A);
        for (int j = 0; j < G::NLEV; ++j)
        {
            float *m = row + 8 + 4 * j;
            m[0] = float(Pm.a); m[1] = float(Pm.b); m[2] = float(Pm.c); m[3] = float(Pm.d);
            Pm = mul(Pm, Pm);
        }
    }

    using big   = geom<32, 128>;    // blocks of up to 4096 samples per launch
    using small = geom<8, 64>;      // blocks of up to 512 samples per launch
} // namespace

struct mi_biquad_bank
{
    uint32_t                channels    = 0;
    uint32_t                max_sec     = 0;
    std::vector<uint32_t>   nsec;           // FilterBank::nItems per channel
    std::vector<int64_t>    last_nsec;      // FilterBank::nLastItems (-1 after init)
    std::vector<float>      coef;           // [channels][max_sec][5]
    std::vector<uint8_t>    dirty;          // tables of the channel need a rebuild
    std::vector<uint8_t>    clear;          // delay memory of the channel must be cleared
    bool                    pending     = false;
    std::vector<float>      h_big, h_small; // host images of the device tables
    float                  *d_big       = nullptr;
    float                  *d_small     = nullptr;
    float                  *d_state     = nullptr;
    float                  *d_backup    = nullptr;
    uint32_t               *d_nsec      = nullptr;
};

namespace
{
    template <int L, int NT>
    hipError_t launch(mi_biquad_bank *b, float *out, const float *in, size_t out_stride,
                      size_t in_stride, int cnt, bool aligned, const float *tab, hipStream_t st)
    {
        const dim3 grid(b->channels), block(NT);
        const bool full = (cnt == L * NT);
        #define MI_LAUNCH(A, F)                                                                   \
            hipLaunchKernelGGL((biquad_bank_kernel<L, NT, A, F>), grid, block, 0, st, out, in,    \
                               out_stride, in_stride, cnt, tab, b->d_state, b->d_nsec, int(b->max_sec))
        if (aligned) { if (full) MI_LAUNCH(true, true); else MI_LAUNCH(true, false); }
        else         { if (full) MI_LAUNCH(false, true); else MI_LAUNCH(false, false); }
        #undef MI_LAUNCH
        return hipGetLastError();
    }

    int commit(mi_biquad_bank *b, hipStream_t st)
    {
        if (!b->pending)
            return MI_OK;
        const size_t row_big = size_t(b->max_sec) * big::TAB, row_small = size_t(b->max_sec) * small::TAB;
        size_t n_dirty = 0;
        for (uint32_t c = 0; c < b->channels; ++c)
        {
            if (!b->dirty[c])
                continue;
            ++n_dirty;
            for (uint32_t s = 0; s < b->nsec[c]; ++s)
            {
                const float *q = &b->coef[(size_t(c) * b->max_sec + s) * 5];
                fill_row<32, 128>(&b->h_big[c * row_big + size_t(s) * big::TAB], q);
                fill_row<8, 64>(&b->h_small[c * row_small + size_t(s) * small::TAB], q);
            }
        }
        if (n_dirty > 0)
        {
            if (n_dirty * 4 >= b->channels)     // mostly dirty: one transfer per table
            {
                MI_HIP_CHECK(hipMemcpyAsync(b->d_big, b->h_big.data(), b->h_big.size() * sizeof(float),
                                            hipMemcpyHostToDevice, st));
                MI_HIP_CHECK(hipMemcpyAsync(b->d_small, b->h_small.data(), b->h_small.size() * sizeof(float),
                                            hipMemcpyHostToDevice, st));
            }
            else
            {
                for (uint32_t c = 0; c < b->channels; ++c)
                {
                    if (!b->dirty[c] || b->nsec[c] == 0)
                        continue;
                    MI_HIP_CHECK(hipMemcpyAsync(b->d_big + c * row_big, &b->h_big[c * row_big],
                                                size_t(b->nsec[c]) * big::TAB * sizeof(float),
                                                hipMemcpyHostToDevice, st));
                    MI_HIP_CHECK(hipMemcpyAsync(b->d_small + c * row_small, &b->h_small[c * row_small],
                                                size_t(b->nsec[c]) * small::TAB * sizeof(float),
                                                hipMemcpyHostToDevice, st));
                }
            }
            MI_HIP_CHECK(hipMemcpyAsync(b->d_nsec, b->nsec.data(), b->channels * sizeof(uint32_t),
                                        hipMemcpyHostToDevice, st));
        }
        // delay memory clears (FilterBank.cpp:233-235)
        uint32_t c = 0;
        while (c < b->channels)
        {
            if (!b->clear[c]) { ++c; continue; }
            uint32_t e = c;
            while (e < b->channels && b->clear[e]) ++e;
            MI_HIP_CHECK(hipMemsetAsync(b->d_state + size_t(c) * b->max_sec * 2, 0,
                                        size_t(e - c) * b->max_sec * 2 * sizeof(float), st));
            c = e;
        }
        // pageable host memory: the runtime has consumed the sources when the calls return
        std::fill(b->dirty.begin(), b->dirty.end(), uint8_t(0));
        std::fill(b->clear.begin(), b->clear.end(), uint8_t(0));
        b->pending = false;
        return MI_OK;
    }
} // namespace

extern "C" {

int mi_biquad_bank_create(mi_biquad_bank_t **bank, uint32_t channels, uint32_t max_sections)
{
    MI_REQUIRE(bank != nullptr, MI_EINVAL, "mi_biquad_bank_create: NULL result pointer");
    *bank = nullptr;
    MI_REQUIRE(channels > 0, MI_EINVAL, "mi_biquad_bank_create: channels must be > 0");
    if (max_sections == 0)
        max_sections = 1;               // FilterBank::init(0) still allocates 3 banks (FilterBank.cpp:67)
    MI_REQUIRE(mi_dspu_device_count() > 0, MI_ENODEV, "no HIP device available (there is no CPU fallback)");

    mi_biquad_bank *b = new (std::nothrow) mi_biquad_bank();
    MI_REQUIRE(b != nullptr, MI_ENOMEM, "mi_biquad_bank_create: out of host memory");
    b->channels = channels;
    b->max_sec  = max_sections;
    const size_t cs = size_t(channels) * max_sections;
    try
    {
        b->nsec.assign(channels, 0);
        b->last_nsec.assign(channels, -1);
        b->coef.assign(cs * 5, 0.0f);
        b->dirty.assign(channels, 0);
        b->clear.assign(channels, 0);
        b->h_big.assign(cs * big::TAB, 0.0f);
        b->h_small.assign(cs * small::TAB, 0.0f);
    }
    catch (...)
    {
        delete b;
        return mi::fail(MI_ENOMEM, "mi_biquad_bank_create: out of host memory");
    }
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_big), cs * big::TAB * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_small), cs * small::TAB * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_state), cs * 2 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_backup), cs * 2 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_nsec), channels * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(b->d_state, 0, cs * 2 * sizeof(float));
    if (e == hipSuccess) e = hipMemset(b->d_nsec, 0, channels * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(b->d_big, 0, cs * big::TAB * sizeof(float));
    if (e == hipSuccess) e = hipMemset(b->d_small, 0, cs * small::TAB * sizeof(float));
    if (e != hipSuccess)
    {
        mi_biquad_bank_destroy(b);
        return mi::fail(e == hipErrorOutOfMemory ? MI_ENOMEM : MI_EHIP,
                        "mi_biquad_bank_create: %s", hipGetErrorString(e));
    }
    *bank = b;
    return MI_OK;
}

int mi_biquad_bank_destroy(mi_biquad_bank_t *b)
{
    if (b == nullptr)
        return MI_OK;
    (void)hipFree(b->d_big);
    (void)hipFree(b->d_small);
    (void)hipFree(b->d_state);
    (void)hipFree(b->d_backup);
    (void)hipFree(b->d_nsec);
    delete b;
    return MI_OK;
}

int mi_biquad_bank_set_chains(mi_biquad_bank_t *b, uint32_t channel,
                              const mi_biquad_x1_t *chains, uint32_t count, int clear)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_biquad_bank_set_chains: NULL bank");
    MI_REQUIRE(channel < b->channels, MI_EINVAL, "mi_biquad_bank_set_chains: channel %u out of range", channel);
    MI_REQUIRE(count == 0 || chains != nullptr, MI_EINVAL, "mi_biquad_bank_set_chains: NULL chains");
    float *dst = &b->coef[size_t(channel) * b->max_sec * 5];
    for (uint32_t i = 0; i < count; ++i)
    {
        // add_chain() beyond the capacity hands out the last slot again (FilterBank.cpp:94-99)
        const uint32_t slot = (i < b->max_sec) ? i : b->max_sec - 1;
        dst[slot * 5 + 0] = chains[i].b0;
        dst[slot * 5 + 1] = chains[i].b1;
        dst[slot * 5 + 2] = chains[i].b2;
        dst[slot * 5 + 3] = chains[i].a1;
        dst[slot * 5 + 4] = chains[i].a2;
    }
    const uint32_t items = (count < b->max_sec) ? count : b->max_sec;
    b->nsec[channel]  = items;
    b->dirty[channel] = 1;
    if (clear || int64_t(items) != b->last_nsec[channel])     // FilterBank.cpp:233-235
        b->clear[channel] = 1;
    b->last_nsec[channel] = items;
    b->pending = true;
    return MI_OK;
}

int mi_biquad_bank_set_all_chains(mi_biquad_bank_t *b, const mi_biquad_x1_t *chains, uint32_t count, int clear)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_biquad_bank_set_all_chains: NULL bank");
    for (uint32_t c = 0; c < b->channels; ++c)
    {
        const int r = mi_biquad_bank_set_chains(b, c, chains + size_t(c) * count, count, clear);
        if (r != MI_OK)
            return r;
    }
    return MI_OK;
}

int mi_biquad_bank_size(const mi_biquad_bank_t *b, uint32_t channel, uint32_t *count)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_biquad_bank_size: NULL bank");
    MI_REQUIRE(channel < b->channels && count != nullptr, MI_EINVAL, "mi_biquad_bank_size: bad argument");
    *count = b->nsec[channel];
    return MI_OK;
}

int mi_biquad_bank_commit(mi_biquad_bank_t *b, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_biquad_bank_commit: NULL bank");
    return commit(b, mi::as_stream(stream));
}

int mi_biquad_bank_reset(mi_biquad_bank_t *b, uint32_t channel, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_biquad_bank_reset: NULL bank");
    if (channel == UINT32_MAX)
        std::fill(b->clear.begin(), b->clear.end(), uint8_t(1));
    else
    {
        MI_REQUIRE(channel < b->channels, MI_EINVAL, "mi_biquad_bank_reset: channel %u out of range", channel);
        b->clear[channel] = 1;
    }
    b->pending = true;
    return commit(b, mi::as_stream(stream));
}

int mi_biquad_bank_process(mi_biquad_bank_t *b, float *out, const float *in, size_t samples,
                           size_t out_stride, size_t in_stride, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_biquad_bank_process: NULL bank");
    if (samples == 0)
        return MI_OK;
    MI_REQUIRE(out != nullptr && in != nullptr, MI_EINVAL, "mi_biquad_bank_process: NULL buffer");
    MI_REQUIRE(out_stride >= samples && in_stride >= samples, MI_EINVAL,
               "mi_biquad_bank_process: stride shorter than the block");
    hipStream_t st = mi::as_stream(stream);
    int r = commit(b, st);
    if (r != MI_OK)
        return r;

    const bool aligned = ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(in)) % 16 == 0) &&
                         (out_stride % 4 == 0) && (in_stride % 4 == 0);
    size_t done = 0;
    while (done < samples)
    {
        const size_t left = samples - done;
        hipError_t e;
        size_t step;
        if (left > size_t(small::BLOCK))
        {
            step = (left < size_t(big::BLOCK)) ? left : size_t(big::BLOCK);
            e = launch<32, 128>(b, out + done, in + done, out_stride, in_stride, int(step),
                                aligned && (done % 4 == 0), b->d_big, st);
        }
        else
        {
            step = left;
            e = launch<8, 64>(b, out + done, in + done, out_stride, in_stride, int(step),
                              aligned && (done % 4 == 0), b->d_small, st);
        }
        MI_HIP_CHECK(e);
        done += step;
    }
    return MI_OK;
}

int mi_biquad_bank_impulse_response(mi_biquad_bank_t *b, float *out, size_t samples, size_t out_stride, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_biquad_bank_impulse_response: NULL bank");
    if (samples == 0)
        return MI_OK;
    MI_REQUIRE(out != nullptr && out_stride >= samples, MI_EINVAL, "mi_biquad_bank_impulse_response: bad buffer");
    hipStream_t st = mi::as_stream(stream);
    int r = commit(b, st);
    if (r != MI_OK)
        return r;
    const size_t bytes = size_t(b->channels) * b->max_sec * 2 * sizeof(float);
    MI_HIP_CHECK(hipMemcpyAsync(b->d_backup, b->d_state, bytes, hipMemcpyDeviceToDevice, st));
    MI_HIP_CHECK(hipMemsetAsync(b->d_state, 0, bytes, st));
    hipLaunchKernelGGL(impulse_kernel, dim3(1024), dim3(256), 0, st, out, out_stride, samples, b->channels);
    MI_HIP_CHECK(hipGetLastError());
    r = mi_biquad_bank_process(b, out, out, samples, out_stride, out_stride, stream);
    if (r != MI_OK)
        return r;
    MI_HIP_CHECK(hipMemcpyAsync(b->d_state, b->d_backup, bytes, hipMemcpyDeviceToDevice, st));
    return MI_OK;
}

int mi_biquad_section_tables(const mi_biquad_x1_t *chain, int variant, float *table, uint32_t *geometry)
{
    MI_REQUIRE(chain != nullptr && geometry != nullptr, MI_EINVAL, "mi_biquad_section_tables: bad argument");
    MI_REQUIRE(variant == 0 || variant == 1, MI_EINVAL, "mi_biquad_section_tables: variant must be 0 or 1");
    const float q[5] = { chain->b0, chain->b1, chain->b2, chain->a1, chain->a2 };
    if (variant == 0)
    {
        geometry[0] = 32; geometry[1] = 128; geometry[2] = big::NLEV; geometry[3] = big::TAB;
        if (table != nullptr)
            fill_row<32, 128>(table, q);
    }
    else
    {
        geometry[0] = 8; geometry[1] = 64; geometry[2] = small::NLEV; geometry[3] = small::TAB;
        if (table != nullptr)
            fill_row<8, 64>(table, q);
    }
    return MI_OK;
}

int mi_biquad_bank_get_state(mi_biquad_bank_t *b, float *host_state, void *stream)
{
    MI_REQUIRE(b != nullptr && host_state != nullptr, MI_EINVAL, "mi_biquad_bank_get_state: bad argument");
    hipStream_t st = mi::as_stream(stream);
    int r = commit(b, st);
    if (r != MI_OK)
        return r;
    MI_HIP_CHECK(hipMemcpyAsync(host_state, b->d_state, size_t(b->channels) * b->max_sec * 2 * sizeof(float),
                                hipMemcpyDeviceToHost, st));
    MI_HIP_CHECK(hipStreamSynchronize(st));
    return MI_OK;
}

int mi_biquad_bank_set_state(mi_biquad_bank_t *b, const float *host_state, void *stream)
{
    MI_REQUIRE(b != nullptr && host_state != nullptr, MI_EINVAL, "mi_biquad_bank_set_state: bad argument");
    hipStream_t st = mi::as_stream(stream);
    int r = commit(b, st);
    if (r != MI_OK)
        return r;
    MI_HIP_CHECK(hipMemcpyAsync(b->d_state, host_state, size_t(b->channels) * b->max_sec * 2 * sizeof(float),
                                hipMemcpyHostToDevice, st));
    MI_HIP_CHECK(hipStreamSynchronize(st));
    return MI_OK;
}

} // extern "C"
