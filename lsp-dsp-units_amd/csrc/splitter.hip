// mi_splitter_bank -- lsp::dspu::SpectralSplitter for many channels (reference: src/main/util/SpectralSplitter.cpp:62-361),
// the engine under lsp::dspu::FFTCrossover (src/main/util/FFTCrossover.cpp).
//
// Per channel: the last N = 2^rank input samples form the analysis buffer; every `frame` = 2^(chunk_rank-1) samples it is
// transformed ONCE, every handler shapes that spectrum its own way, goes back to the time domain, and the last 2*frame
// samples -- windowed with sin^2 -- are overlap-added into the handler's output line.  One workgroup owns one channel's
// hop: the half spectrum of the real transform stays in registers while the handlers take turns in LDS, so the input is
// read once and transformed once however many bands there are ("multiple masks per frame ->
// multi-band output in one pass").
//
// Handler kinds:
//   COPY      no spectral function: the first 2*frame samples of the analysis buffer go to the output line
//             (SpectralSplitter.cpp:330);
//   MASK      spectrum[k] *= gain[k] with N real gains in FFT order (what FFTCrossover::spectral_func does with its vFFT,
//             FFTCrossover.cpp:124-140), fused in the hop kernel;
//   CALLBACK  the full complex spectrum is handed to a host function as a device pointer (it may enqueue anything on the
//             stream); a complex inverse transform brings its result back.  This is the compatibility path of the
//             lsp::dspu::SpectralSplitter class mirror.
#include "mi_common.h"
#include "fft_device.h"
#include "fft16.h"
#include "fft_wave.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

namespace mi
{
    void make_window(float *dst, size_t n, int type);                       // host/windows.cpp
}

namespace
{
    using namespace mi_fft;
    // radix-16 core (fft16.h) for 1024 .. 8192-point transforms, radix-8 core (fft_device.h) below that
    template <int L_> using fplan = mi_fft16::fsel<L_>;

    enum { H_OFF = 0, H_COPY = 1, H_MASK = 2, H_CALLBACK = 3 };

    // the caller's output rows travel by value with the launch (no device copy of a pointer table to keep in step with the
    // stream); banks with more handlers fall back to the table
    constexpr uint32_t OUTS_BY_VALUE = 16;
    struct out_table
    {
        float *p[OUTS_BY_VALUE];
        float *const *more;             // handlers > OUTS_BY_VALUE: device array
        __device__ __forceinline__ float *at(uint32_t h) const { return (more != nullptr) ? more[h] : p[h]; }
    };

    struct handler_desc
    {
        const float *mask;              // H_MASK: [channels or 1][mask_stride] real gains, N per row
        uint64_t     mask_stride;       // 0: one row for every channel
        uint32_t     mode;
        uint32_t     has_sink;          // overlap-add only where somebody listens (SpectralSplitter.cpp:333)
    };

    // overlap-add of 2*frame windowed samples y (as pairs) into a handler's line: the line moves on by one frame.
    // emit (or NULL): where the frame that is finished now -- the first half of the line -- goes in the caller's buffer.
    template <typename GET>
    __device__ __forceinline__ void overlap_add(float2 *line, const float2 *__restrict__ wnd, uint32_t frame, int tid, int T,
                                                float scale, float *emit, GET y)
    {
        const uint32_t hp = frame >> 1;                                 // a frame of samples is frame/2 pairs
        for (uint32_t m = tid; m < hp; m += T)
        {
            // a thread owns pair m of both frames of the line: the old second frame is read before it is overwritten
            const float2 y0 = y(m), y1 = y(m + hp), w0 = wnd[m], w1 = wnd[m + hp];
            const float2 prev = line[m + hp];
            const float2 done = make_float2(fmaf(y0.x * scale, w0.x, prev.x), fmaf(y0.y * scale, w0.y, prev.y));
            line[m]      = done;
            line[m + hp] = make_float2(y1.x * scale * w1.x, y1.y * scale * w1.y);
            if (emit != nullptr)
            {
                // (the caller's row: global memory, whatever the pointer's history -- see the note in splitter_hop_kernel)
                typedef __attribute__((address_space(1))) float gwfloat;
                gwfloat *const e = reinterpret_cast<gwfloat *>(reinterpret_cast<uint64_t>(emit));
                e[2 * m]     = done.x;
                e[2 * m + 1] = done.y;
            }
        }
    }

    // One hop of one channel.  in_cur / in_next: [channels][in_pitch], the analysis buffer in the first N floats -- read
    // from one, written (moved on by one frame) into the other, so that several workgroups may work on a channel; lines:
    // [handlers][channels][line_pitch]; spec (only when WRITE_SPEC): [channels][N] complex.
    // gridDim.y == 1: the workgroup serves every handler of its channel (the input is read and transformed once);
    // gridDim.y == handlers: one handler each (few channels: more workgroups than CUs matter more than the repeated
    // forward transform; ceil(handlers / MULTI) in the several-hops form); workgroup y == 0 moves the analysis buffer on.
    // Fused streaming (ingest_n > 0): the `frame` samples that follow the hop are taken from `src` (NULL: silence) into the
    // new analysis buffer, and the frame every handler finishes goes straight to the caller's buffers.
    // MULTI > 0: the several-hops form, MULTI handlers per workgroup.  One handler per workgroup: held to four waves per
    // SIMD (128 VGPRs) -- 1024 workgroups of 256 threads are then ONE round on the device; left to itself the compiler
    // lands on either side of that line from one change of the source to the next (102 / 135 / 146 VGPRs seen: 30 against
    // 36 us per block).
    // Runs of blocks (mi_splitter_bank_process_blocks): the hops of SEVERAL calls in one launch -- every block a buffer of its
    // own (and a set of output buffers of its own), `per` hops each, the addresses in the kernel arguments.
    constexpr uint32_t SPLIT_BLOCKS_MAX = 64, SPLIT_PTRS_MAX = 256;
    struct split_blocks
    {
        uint32_t        per;                    // hops per block
        const float    *src[SPLIT_BLOCKS_MAX];
        float          *out[SPLIT_PTRS_MAX];    // [block * handlers + handler]
    };

    template <int LOGH, bool WRITE_SPEC, bool PER_BAND, int MULTI, bool TAB>
    __device__ __forceinline__
    void splitter_hop_body(const float *in_cur, float *in_next, size_t in_pitch, float *lines, size_t line_pitch,
                           uint32_t channels, const handler_desc *__restrict__ hd, uint32_t handlers,
                           const float *__restrict__ wnd, uint32_t frame, float2 *spec, const float2 *__restrict__ tw,
                           const float *src, size_t src_stride, uint32_t ingest_n, const out_table &outs,
                           size_t out_stride, size_t out_pos, uint32_t hops /* > 1: that many hops of a streaming call at once */,
                           const split_blocks *tab)
    {
        static_assert(!TAB || MULTI > 0, "runs of blocks: the several-hops form");
        using PL = fplan<LOGH>;
        constexpr int H = PL::N, T = PL::T, N = 2 * H, PER = (H + T - 1) / T;
        __shared__ float2 lds_[fplan<LOGH>::LDS];
        float2 *const buf = lds_, *const scr = lds_ + fplan<LOGH>::SCR;
        const int ch = blockIdx.x, tid = threadIdx.x;
        // a handler's output rows (runs of blocks: out of the table at every block's first hop -- every listening handler is a mask there)
        auto out_row = [&](uint32_t h) -> float * { return TAB ? nullptr : outs.at(h); };
        constexpr bool all = !PER_BAND;
        constexpr uint32_t BPW = (MULTI > 0) ? MULTI : 1;          // handlers per workgroup
        // (one handler per workgroup: the compiler must SEE that the handler loops run once -- as a min() with `handlers`
        // the one-hop kernel took 152 registers instead of 102)
        const uint32_t h0 = all ? 0 : blockIdx.y * BPW;
        const uint32_t h1 = all ? handlers : (BPW == 1) ? h0 + 1 : (h0 + BPW < handlers ? h0 + BPW : handlers);
        const bool owner = all || (blockIdx.y == 0);
        // the frame is asked for before the handler descriptors are looked at (their little dependent loads would otherwise
        // cost an exposed latency before the frame's own: tests/experiments/analyzer_probe.hip found that pattern)
        const float2 *x2 = reinterpret_cast<const float2 *>(in_cur + size_t(ch) * in_pitch);
        float2 xr[PER];
        #pragma unroll
        for (int i = 0; i < PER; ++i)
            xr[i] = (tid + i * T < H) ? x2[tid + i * T] : make_float2(0.0f, 0.0f);
        bool masks = WRITE_SPEC && owner;
        for (uint32_t h = h0; h < h1; ++h)
            masks = masks || (hd[h].mode == H_MASK && hd[h].has_sink);
        bool copies = false;
        for (uint32_t h = h0; h < h1; ++h)
            copies = copies || (hd[h].mode == H_COPY && hd[h].has_sink);
        if (!owner && !masks && !copies)
            return;
        typename fplan<LOGH>::real rf;
        if (masks)
            rf.load(tw, TWN, tid);
        const float2 *w2 = reinterpret_cast<const float2 *>(wnd);
        if (masks)
            rf.prepare();
        #pragma unroll
        for (int i = 0; i < PER; ++i)
            if (tid + i * T < H)
                buf[tid + i * T] = xr[i];
        __syncthreads();

        // handlers without a spectral function see the head of the analysis buffer as it is
        for (uint32_t h = h0; h < h1; ++h)
        {
            if (hd[h].mode != H_COPY || !hd[h].has_sink)
                continue;
            float2 *line = reinterpret_cast<float2 *>(lines + (size_t(h) * channels + ch) * line_pitch);
            float *const orow = out_row(h);
            float *emit = (ingest_n > 0 && orow != nullptr) ? orow + size_t(ch) * out_stride + out_pos : nullptr;
            overlap_add(line, w2, frame, tid, T, 1.0f, emit, [&](uint32_t m) { return buf[m]; });
        }
        if (owner)
        {
            // the analysis buffer moves on by one frame, then takes the samples of the frame that follows
            float *nx = in_next + size_t(ch) * in_pitch;
            float2 *n2 = reinterpret_cast<float2 *>(nx);
            #pragma unroll
            for (int i = 0; i < PER; ++i)
            {
                const int m = tid + i * T;
                if (m >= int(frame >> 1) && m < H)
                    n2[m - (frame >> 1)] = xr[i];
            }
            if (ingest_n > 0)
            {
                // the caller's `frame` samples behind hop q (TAB: block q / per of the run, hop q % per inside it)
                auto hop_src = [&](uint32_t q) -> const float * {
                    if (TAB)
                    {
                        const uint32_t k = q / tab->per;
                        return tab->src[k] + size_t(ch) * src_stride + size_t(q - k * tab->per) * frame;
                    }
                    return (src != nullptr) ? src + size_t(ch) * src_stride + size_t(q) * frame : nullptr;
                };
                const float *s1 = hop_src(hops - 1);
                for (uint32_t i = tid; i < ingest_n; i += T)
                    nx[N - frame + i] = (s1 != nullptr) ? s1[i] : 0.0f;
                if (hops > 1)                               // (frame == N / 2 then) the buffer after the last hop: the call's last two blocks
                {
                    const float *s0 = hop_src(hops - 2);
                    for (uint32_t i = tid; i < frame; i += T)
                        nx[i] = (s0 != nullptr) ? s0[i] : 0.0f;
                }
            }
        }
        if (!masks)
            return;
        __syncthreads();

        // 512 .. 8192-point transforms whose hop is the whole half frame (frame == H: the line's 2 * frame samples are the
        // transform's output as it stands): the frame goes into the forward transform in registers, the thread's pairs
        // (Z_k, Z_(H-k)) wait in registers while the handlers take turns, each handler's split + gains + merge is ONE pass
        // (real_fft::pairs_mask_store) and its inverse hands the samples back in registers for the overlap-add
        // (round 3: 104 + 96 h LDS instructions per thread and hop -> 72 + 64 h).
        if constexpr (!WRITE_SPEC && LOGH <= 12 && !fplan<LOGH>::radix16 && (mi_fft::plan<LOGH>::T == mi_fft::plan<LOGH>::TB))
        {
            if (frame == uint32_t(H))
            {
                static_assert(PER * T == H && (PER % 2) == 0, "whole pairs per thread");
                constexpr int IT = mi_fft::real_fft<LOGH>::PAIRS;
                const float scale = 1.0f / float(N);
                const uint32_t hp = frame >> 1;
                typedef const __attribute__((address_space(1))) float gfloat;
                typedef __attribute__((address_space(1))) float gwfloat;
                typedef const __attribute__((address_space(1))) v2f gv2f;
                typedef __attribute__((address_space(1))) v2f gwv2f;
                if constexpr (MULTI > 0)
                {
                    // Several hops of a streaming call in ONE launch (host: every listening handler a mask; MULTI handlers per
                    // workgroup, which share the forward transform): between two hops nothing goes through memory -- the half
                    // of the frame the next hop starts with, and the tail the overlap-add leaves in a handler's line, stay in
                    // registers; the caller's samples that complete the next frame come straight from its block.
                    // (the same for every lane, and told so: the addresses live in SGPRs)
                    auto one = [](const void *q) -> uint64_t {
                        const uint64_t v = reinterpret_cast<uint64_t>(q);
                        return uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v)))))
                             | (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v >> 32))))) << 32);
                    };
                    float2 lo[PER / 2], hi[PER / 2], tail[MULTI][PER / 2];
                    bool on[MULTI];
                    uint64_t gp[MULTI], ep[MULTI], lp[MULTI];
                    #pragma unroll
                    for (int b = 0; b < MULTI; ++b)
                    {
                        const uint32_t h = h0 + b;
                        on[b] = h < h1 && hd[h].mode == H_MASK && hd[h].has_sink;
                        const uint32_t hh = on[b] ? h : h0;
                        gp[b] = one(hd[hh].mask + size_t(ch) * hd[hh].mask_stride);
                        float *const orow = out_row(hh);
                        ep[b] = one((orow != nullptr) ? orow + size_t(ch) * out_stride + out_pos : nullptr);
                        lp[b] = one(lines + (size_t(hh) * channels + ch) * line_pitch);
                        #pragma unroll
                        for (int i = 0; i < PER / 2; ++i)
                        {
                            const v2f t = on[b] ? reinterpret_cast<gv2f *>(lp[b])[tid + i * T + hp] : v2f{0.0f, 0.0f};
                            tail[b][i] = make_float2(t.x, t.y);
                        }
                    }
                    #pragma unroll
                    for (int i = 0; i < PER / 2; ++i)
                    {
                        lo[i] = xr[i];
                        hi[i] = xr[i + PER / 2];
                    }
                    uint64_t sp = one((src != nullptr) ? src + size_t(ch) * src_stride : nullptr);
                    uint64_t wp = one(wnd);
                    for (uint32_t hop = 0; hop < hops; ++hop)
                    {
                        // TAB: hop `hop` of the run is hop `jh` of block `blk`; at a block's first hop the addresses of its
                        // input rows and of the handlers' output rows are taken from the table
                        uint32_t jh = hop;
                        if (TAB)
                        {
                            const uint32_t blk = hop / tab->per;
                            jh = hop - blk * tab->per;
                            if (jh == 0)
                            {
                                sp = one(tab->src[blk] + size_t(ch) * src_stride);
                                #pragma unroll
                                for (int b = 0; b < MULTI; ++b)
                                {
                                    const uint32_t hh = on[b] ? h0 + b : h0;
                                    float *o = tab->out[blk * handlers + hh];
                                    ep[b] = one((o != nullptr) ? o + size_t(ch) * out_stride : nullptr);
                                }
                            }
                        }
                        // (what does not change from hop to hop is laundered once per hop: hoisted out of the loop its loads
                        // would be kept across the transforms and spill the twiddles, spectral.hip's stft_stream_kernel)
                        asm volatile("" : "+s"(sp), "+s"(wp));
                        int tix = tid;
                        asm volatile("" : "+v"(tix));
                        gv2f *const s2 = reinterpret_cast<gv2f *>(sp);
                        gv2f *const wg = reinterpret_cast<gv2f *>(wp);
                        v2f io[PER];
                        #pragma unroll
                        for (int i = 0; i < PER / 2; ++i)
                        {
                            io[i] = v2f{lo[i].x, lo[i].y};
                            io[i + PER / 2] = v2f{hi[i].x, hi[i].y};
                            lo[i] = hi[i];                              // the frame moves on by half
                            const v2f nx = (sp != 0) ? s2[size_t(jh) * hp + tix + i * T] : v2f{0.0f, 0.0f};
                            hi[i] = make_float2(nx.x, nx.y);
                        }
                        mi_fft::fft_lds<LOGH, false, true, false>(buf, scr, rf.ft, tix, io);
                        float2 zk[IT], zm[IT];
                        rf.pairs_load(buf, zk, zm, tix);
                        #pragma unroll
                        for (int b = 0; b < MULTI; ++b)
                        {
                            if (!on[b])
                                continue;
                            asm volatile("" : "+s"(gp[b]), "+s"(ep[b]), "+s"(lp[b]));
                            gfloat *const g = reinterpret_cast<gfloat *>(gp[b]);
                            gwfloat *const emit = reinterpret_cast<gwfloat *>(ep[b]);
                            __syncthreads();                            // everybody holds its pairs / is done with the handler before
                            rf.pairs_mask_store(buf, zk, zm, [&](int k) -> float { return g[k]; /* the even part already, bind_mask */ }, tix);
                            mi_fft::fft_lds<LOGH, true, false, true>(buf, scr, rf.ft, tix, io);
                            #pragma unroll
                            for (int i = 0; i < PER / 2; ++i)
                            {
                                const uint32_t m = tix + i * T;
                                const v2f y0 = io[i], y1 = io[i + PER / 2], w0 = wg[m], w1 = wg[m + hp];
                                const float2 done = make_float2(fmaf(y0.x * scale, w0.x, tail[b][i].x), fmaf(y0.y * scale, w0.y, tail[b][i].y));
                                tail[b][i] = make_float2(y1.x * scale * w1.x, y1.y * scale * w1.y);
                                if (hop + 1 == hops)                    // the handler's line as the call leaves it
                                    reinterpret_cast<gwv2f *>(lp[b])[m] = v2f{done.x, done.y};
                                if (ep[b] != 0)
                                {
                                    emit[size_t(jh) * frame + 2 * m]     = done.x;
                                    emit[size_t(jh) * frame + 2 * m + 1] = done.y;
                                }
                            }
                        }
                        if (hop + 1 < hops)
                            __syncthreads();                            // buf is refilled by the next hop
                    }
                    #pragma unroll
                    for (int b = 0; b < MULTI; ++b)
                        if (on[b])
                        {
                            #pragma unroll
                            for (int i = 0; i < PER / 2; ++i)
                                reinterpret_cast<gwv2f *>(lp[b])[tid + i * T + hp] = v2f{tail[b][i].x, tail[b][i].y};
                        }
                    return;
                }
                v2f io[PER];
                #pragma unroll
                for (int i = 0; i < PER; ++i)
                    io[i] = v2f{xr[i].x, xr[i].y};
                mi_fft::fft_lds<LOGH, false, true, false>(buf, scr, rf.ft, tid, io);
                float2 zk[IT], zm[IT];
                rf.pairs_load(buf, zk, zm, tid);
                for (uint32_t h = h0; h < h1; ++h)
                {
                    if (hd[h].mode != H_MASK || !hd[h].has_sink)
                        continue;
                    // (pointers that come out of memory -- the handler's gains, the caller's output rows -- are generic to the
                    // compiler: read or written through them every access is a FLAT instruction, which counts against
                    // lgkmcnt too and ties the waits for LDS data to it.  They are global memory and are told so.)
                    gfloat *const g = reinterpret_cast<gfloat *>(reinterpret_cast<uint64_t>(hd[h].mask + size_t(ch) * hd[h].mask_stride));
                    __syncthreads();                                    // everybody holds its pairs / is done with the handler before
                    // only the real part of the inverse is kept (pcomplex_c2r): a real gain acts through its even part
                    // (the gains are asked for here, not ahead of the forward transform: with their addresses known early the
                    // compiler fetches them all at the top and the kernel needs 146 registers instead of 102)
                    int tix = tid;
                    asm volatile("" : "+v"(tix));
                    rf.pairs_mask_store(buf, zk, zm, [&](int k) -> float { return g[k]; /* the even part already, bind_mask */ }, tix);
                    mi_fft::fft_lds<LOGH, true, false, true>(buf, scr, rf.ft, tid, io);
                    float2 *line = reinterpret_cast<float2 *>(lines + (size_t(h) * channels + ch) * line_pitch);
                    float *emit_ = (ingest_n > 0 && out_row(h) != nullptr) ? out_row(h) + size_t(ch) * out_stride + out_pos : nullptr;
                    gwfloat *const emit = reinterpret_cast<gwfloat *>(reinterpret_cast<uint64_t>(emit_));
                    #pragma unroll
                    for (int i = 0; i < PER / 2; ++i)
                    {
                        const uint32_t m = tid + i * T;                 // pair m of the first frame, pair m + hp of the second
                        const v2f y0 = io[i], y1 = io[i + PER / 2];
                        const float2 w0 = w2[m], w1 = w2[m + hp], prev = line[m + hp];
                        const float2 done = make_float2(fmaf(y0.x * scale, w0.x, prev.x), fmaf(y0.y * scale, w0.y, prev.y));
                        line[m]      = done;
                        line[m + hp] = make_float2(y1.x * scale * w1.x, y1.y * scale * w1.y);
                        if (emit != nullptr)
                        {
                            emit[2 * m]     = done.x;
                            emit[2 * m + 1] = done.y;
                        }
                    }
                }
                return;
            }
        }

        rf.forward(buf, scr, tid);
        // several handlers per workgroup: the spectrum waits in registers while they take turns in LDS; one handler per
        // workgroup shapes it where it is
        constexpr int KEEP = PER_BAND ? 1 : PER;
        float2 sp[KEEP];
        if (!PER_BAND || WRITE_SPEC)
        {
            #pragma unroll
            for (int i = 0; i < KEEP; ++i)
                sp[i] = (tid + i * T < H) ? buf[tid + i * T] : make_float2(0.0f, 0.0f);
        }

        if (WRITE_SPEC && owner)
        {
            float2 *so = spec + size_t(ch) * N;
            #pragma unroll
            for (int i = 0; i < PER; ++i)
            {
                const int k = tid + i * T;
                if (k >= H)
                    continue;
                if (k == 0)
                {
                    so[0] = make_float2(sp[i].x, 0.0f);
                    so[H] = make_float2(sp[i].y, 0.0f);
                }
                else
                {
                    so[k]     = sp[i];
                    so[N - k] = cconj(sp[i]);
                }
            }
        }

        const float scale = 1.0f / float(N);
        for (uint32_t h = h0; h < h1; ++h)
        {
            if (hd[h].mode != H_MASK || !hd[h].has_sink)
                continue;
            typedef const __attribute__((address_space(1))) float gfloat;
            gfloat *const g = reinterpret_cast<gfloat *>(reinterpret_cast<uint64_t>(hd[h].mask + size_t(ch) * hd[h].mask_stride));
            if (!PER_BAND)
                __syncthreads();                                        // the previous handler is done with buf
            #pragma unroll
            for (int i = 0; i < PER; ++i)
            {
                const int k = tid + i * T;
                if (k >= H)
                    continue;
                float2 v = PER_BAND ? buf[k] : sp[PER_BAND ? 0 : i];
                if (k == 0)
                {
                    v.x *= g[0];
                    v.y *= g[H];
                }
                else
                {
                    // only the real part of the inverse is kept (pcomplex_c2r): a real gain acts through its even part
                    const float gk = 0.5f * (g[k] + g[N - k]);
                    v.x *= gk;
                    v.y *= gk;
                }
                buf[k] = v;
            }
            __syncthreads();
            rf.inverse(buf, scr, tid);
            float2 *line = reinterpret_cast<float2 *>(lines + (size_t(h) * channels + ch) * line_pitch);
            float *emit = (ingest_n > 0 && out_row(h) != nullptr) ? out_row(h) + size_t(ch) * out_stride + out_pos : nullptr;
            const uint32_t first = uint32_t(H) - frame;                // the last 2*frame samples, in pairs
            overlap_add(line, w2, frame, tid, T, scale, emit, [&](uint32_t m) { return buf[first + m]; });
        }
    }

    // ---- runs of 4096-sample blocks at rank 12, every listening handler a mask shared by the channels: on the wave-resident
    // transform (fft_wave.h) ------------------------------------------------------------------------------------------------------
    // splitter_hops_blocks_kernel runs at the rate of its transforms through LDS, and at 256 channels it is one workgroup per channel
    // AND band, each redoing the forward transform.  Here a WAVE owns a channel's consecutive blocks and takes the two frames of a
    // block -- frame 2u = block u - 1, frame 2u + 1 = [second half of block u - 1 | first half of block u] -- as ONE complex
    // sequence z = A + i B: a real, even gain acts on the spectra of A and B alike, so g Z is the spectrum of the two shaped frames:
    // ONE forward transform per block and one inverse per band (NB + 1 transforms of 4096 complex points for what was 2 + 2 NB real
    // transforms of 4096 points, with their splits and merges), no barrier.  The spectrum waits in registers while the bands take
    // turns, each band's overlap-add tail stays in 32 registers of the wave: 128 (work) + 128 (spectrum) + 32 NB registers -- the
    // wave has its SIMD to itself (four waves per CU) and the whole register file with it.  Window and gains (their halves: bind_mask
    // stores the even part) sit in LDS.  A channel's run is cut into 1, 2 or 4 segments, the waves of one workgroup; a segment that
    // does not start the run first redoes the block in front of it (without storing) for the tails it starts from.
    // The same sums through another transform and in another order of roundings: within 1e-6 of splitter_hops_blocks_kernel, not its
    // bits.  The host sends a run this way only if no buffer of the run overlaps another (the segments run side by side).
    constexpr int SPW = 4;                                  // waves of a workgroup: one per SIMD
    struct wave_bands
    {
        const float    *gain[4];                            // the listening handlers' gains (N floats, even)
        float          *line[4];                            // ... their lines [channels][pitch]
        uint32_t        handler[4];                         // ... their numbers (columns of split_blocks::out)
    };
    template <int NB>
    __global__ __launch_bounds__(64 * SPW, 1)
    void splitter_wave_blocks_kernel(const float *in_cur, float *in_next, size_t in_pitch, size_t line_pitch, const wave_bands wb,
                                     uint32_t handlers, const float *__restrict__ wnd, const float2 *__restrict__ tw,
                                     const split_blocks tab, size_t src_stride, size_t out_stride, int blocks, int channels, int segs)
    {
        using namespace mi_fftw;
        constexpr int HALF = R / 2, HOP = N / 2;
        __shared__ float areas[SPW][AREA];
        __shared__ float2 pl[16 * R];
        // window and gains the way a lane reads them: entry lane + 64 r of a table at float4 cell [r / 4][lane], component r % 4 --
        // sixteen 16-byte LDS reads per table and use instead of sixty-four 4-byte ones (one wave per SIMD: every read's round trip
        // is the wave's own time)
        __shared__ float4 wnd_l[N / 4];
        __shared__ float4 gain_l[NB][N / 4];
        const int tid = threadIdx.x, lane = tid & 63;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        fill_table_pq(pl, tw, tid, 64 * SPW);
        for (int i = tid; i < N; i += 64 * SPW)
        {
            const int l = i & 63, r = i >> 6;               // entry i = l + 64 r
            reinterpret_cast<float *>(wnd_l)[((r >> 2) * 64 + l) * 4 + (r & 3)] = wnd[i] * (1.0f / float(N));   // (the pair's 1 / N rides on the window)
            #pragma unroll
            for (int b = 0; b < NB; ++b)
                reinterpret_cast<float *>(gain_l[b])[((r >> 2) * 64 + l) * 4 + (r & 3)] = wb.gain[b][i];
        }
        __syncthreads();
        const int gid = blockIdx.x * SPW + wv;              // wave of the launch: (channel, segment); segs divides SPW
        const bool idle = gid >= channels * segs;
        const int ch = idle ? 0 : gid / segs, seg = gid - ch * segs;
        const int per = (blocks + segs - 1) / segs, u0 = seg * per, u1 = idle ? u0 : ((u0 + per < blocks) ? u0 + per : blocks);
        auto at = [](__amdgpu_buffer_rsrc_t r, int lane_off, int row_off) -> float {
            return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, lane_off, row_off, 0));
        };
        // block u of the run as the caller gave it (u = -1: the N samples the object holds)
        auto block_in = [&](int u) -> __amdgpu_buffer_rsrc_t {
            return mi::wt_buffer((u < 0) ? const_cast<float *>(in_cur) + size_t(ch) * in_pitch
                                         : const_cast<float *>(tab.src[u]) + size_t(ch) * src_stride, unsigned(N * sizeof(float)));
        };
        float tail[NB][HALF];                               // what the next frame adds to, band by band: sample lane + 64 j
        #pragma unroll
        for (int b = 0; b < NB; ++b)
        {
            const __amdgpu_buffer_rsrc_t rl = mi::wt_buffer(wb.line[b] + size_t(ch) * line_pitch, (u0 == 0 && u0 < u1) ? unsigned(N * sizeof(float)) : 0u);
            #pragma unroll
            for (int j = 0; j < HALF; ++j)
                tail[b][j] = at(rl, lane * 4, (HOP + 64 * j) * 4);      // (0 where the segment starts inside the run: out of range)
        }
        for (int u = (u0 == 0) ? 0 : u0 - 1; u < u1 && u0 < u1; ++u)
        {
            // z[n] = A[n] + i B[n], n = lane + 64 j: A = block u - 1, B = [its second half | first half of block u]
            const __amdgpu_buffer_rsrc_t ra = block_in(u - 1), rb = block_in(u);
            v2f x[R], z[R];
            #pragma unroll
            for (int j = 0; j < R; ++j)
                x[j].x = at(ra, lane * 4, 256 * j);
            #pragma unroll
            for (int j = 0; j < HALF; ++j)
            {
                x[j].y = x[j + HALF].x;
                x[j + HALF].y = at(rb, lane * 4, 256 * j);
            }
            fft4096_t<false>(x, pl, areas[wv], lane);
            #pragma unroll
            for (int r = 0; r < R; ++r)
                z[r] = x[r];
            const bool store = u >= u0;                      // (the block in front of the segment: only its tails are wanted)
            // The bands take turns in ONE copy of the code (an inverse transform is 37 KB of it): the loop is kept rolled and the
            // tails ROTATE through tail[0] -- indexed by the loop's counter they would live in scratch memory.
            #pragma nounroll
            for (int b = 0; b < NB; ++b)
            {
                // the gain of bin k = lane + 64 r (the even part of the handler's gains: N of them)
                #pragma unroll
                for (int r4 = 0; r4 < R / 4; ++r4)
                {
                    const float4 g = gain_l[b][r4 * 64 + lane];
                    x[4 * r4 + 0] = z[4 * r4 + 0] * v2f{g.x, g.x};
                    x[4 * r4 + 1] = z[4 * r4 + 1] * v2f{g.y, g.y};
                    x[4 * r4 + 2] = z[4 * r4 + 2] * v2f{g.z, g.z};
                    x[4 * r4 + 3] = z[4 * r4 + 3] * v2f{g.w, g.w};
                }
                fft4096_t<true>(x, pl, areas[wv], lane);
                #pragma unroll
                for (int j4 = 0; j4 < R / 4; ++j4)
                {
                    const float4 w = wnd_l[j4 * 64 + lane];
                    x[4 * j4 + 0] = x[4 * j4 + 0] * v2f{w.x, w.x};
                    x[4 * j4 + 1] = x[4 * j4 + 1] * v2f{w.y, w.y};
                    x[4 * j4 + 2] = x[4 * j4 + 2] * v2f{w.z, w.z};
                    x[4 * j4 + 3] = x[4 * j4 + 3] * v2f{w.w, w.w};
                }
                float *const o = tab.out[size_t(u) * handlers + wb.handler[b]];
                const __amdgpu_buffer_rsrc_t rout = mi::wt_buffer((store && o != nullptr) ? o + size_t(ch) * out_stride : nullptr,
                                                                  (store && o != nullptr) ? unsigned(N * sizeof(float)) : 0u);
                #pragma unroll
                for (int j = 0; j < HALF; ++j)
                {
                    const float done_a = x[j].x + tail[0][j];
                    const float done_b = x[j].y + x[j + HALF].x;
                    const float mine = x[j + HALF].y;                       // this band's new tail: to the back of the queue
                    #pragma unroll
                    for (int q = 0; q + 1 < NB; ++q)
                        tail[q][j] = tail[q + 1][j];
                    tail[NB - 1][j] = mine;
                    mi::wt_store<mi::CPOL_NT_SC1>(rout, lane * 4 + 256 * j, done_a);             // (dropped by the bounds check where nothing is stored)
                    mi::wt_store<mi::CPOL_NT_SC1>(rout, lane * 4 + 256 * (j + HALF), done_b);
                }
            }
        }
        // The object's state as the call leaves it -- behind a barrier: the wave of the channel's first segment has read the state the
        // call found.  Lines: [the frame finished last | the tail]; the analysis buffer (the other one of the pair): the last block.
        __syncthreads();
        if (u1 == blocks && u0 < u1)
        {
            const __amdgpu_buffer_rsrc_t rlast = block_in(blocks - 1);
            const __amdgpu_buffer_rsrc_t rx = mi::wt_buffer(in_next + size_t(ch) * in_pitch, unsigned(N * sizeof(float)));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (this wave's own stores of the last block)
            // (all the loads of a piece, then its stores: alternating, every store waits for its load's round trip)
            {
                float t[R];
                #pragma unroll
                for (int j = 0; j < R; ++j)
                    t[j] = at(rlast, lane * 4, 256 * j);
                #pragma unroll
                for (int j = 0; j < R; ++j)
                    mi::wt_store(rx, lane * 4 + 256 * j, t[j]);
            }
            #pragma unroll
            for (int b = 0; b < NB; ++b)
            {
                const __amdgpu_buffer_rsrc_t rl = mi::wt_buffer(wb.line[b] + size_t(ch) * line_pitch, unsigned(N * sizeof(float)));
                float *const o = tab.out[size_t(blocks - 1) * handlers + wb.handler[b]];
                const __amdgpu_buffer_rsrc_t rd = mi::wt_buffer(o + size_t(ch) * out_stride, unsigned(N * sizeof(float)));
                float d[HALF];
                #pragma unroll
                for (int j = 0; j < HALF; ++j)
                    d[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, lane * 4, 256 * (j + HALF), mi::CPOL_SC1));
                #pragma unroll
                for (int j = 0; j < HALF; ++j)
                {
                    mi::wt_store(rl, lane * 4 + 256 * j, d[j]);
                    mi::wt_store(rl, lane * 4 + 256 * (j + HALF), tail[b][j]);
                }
            }
        }
    }

    template <int LOGH, bool WRITE_SPEC, bool PER_BAND, int MULTI = 0>
    __global__ __launch_bounds__(fplan<LOGH>::T, (PER_BAND && MULTI <= 1) ? 4 : (MULTI == 2) ? 2 : 1)
    void splitter_hop_kernel(const float *in_cur, float *in_next, size_t in_pitch, float *lines, size_t line_pitch,
                             uint32_t channels, const handler_desc *__restrict__ hd, uint32_t handlers,
                             const float *__restrict__ wnd, uint32_t frame, float2 *spec, const float2 *__restrict__ tw,
                             const float *src, size_t src_stride, uint32_t ingest_n, const out_table outs,
                             size_t out_stride, size_t out_pos, uint32_t hops /* > 1: that many hops of a streaming call at once */)
    {
        splitter_hop_body<LOGH, WRITE_SPEC, PER_BAND, MULTI, false>(in_cur, in_next, in_pitch, lines, line_pitch, channels, hd, handlers, wnd,
                                                                   frame, spec, tw, src, src_stride, ingest_n, outs, out_stride, out_pos, hops, nullptr);
    }

    // the several-hops form over a run of blocks, one handler per workgroup
    template <int LOGH>
    __global__ __launch_bounds__(fplan<LOGH>::T, (fplan<LOGH>::T <= 64) ? 2 : 4)     // (one-wave workgroups: no need to squeeze into 128 registers)
    void splitter_hops_blocks_kernel(const float *in_cur, float *in_next, size_t in_pitch, float *lines, size_t line_pitch,
                                     uint32_t channels, const handler_desc *__restrict__ hd, uint32_t handlers,
                                     const float *__restrict__ wnd, uint32_t frame, const float2 *__restrict__ tw,
                                     size_t src_stride, size_t out_stride, uint32_t hops, const split_blocks tab)
    {
        const out_table none = {};                          // (the rows come out of the table, block by block)
        splitter_hop_body<LOGH, false, true, 1, true>(in_cur, in_next, in_pitch, lines, line_pitch, channels, hd, handlers, wnd, frame,
                                                      nullptr, tw, tab.src[0], src_stride, frame, none, out_stride, 0, hops, &tab);
    }

    // CALLBACK handlers: only the real part of the inverse of what the function left in `spec` is kept (pcomplex_c2r),
    // and Re ifft(S) is the inverse of S's Hermitian part (S[k] + conj S[N-k]) / 2 -- the same half-size real transform
    // as on the way there.  The last 2*frame samples are windowed into the handler's line.
    template <int LOGH>
    __global__ __launch_bounds__(fplan<LOGH>::T)
    void splitter_inverse_kernel(float *line0, size_t line_pitch, const float2 *__restrict__ spec,
                                 const float *__restrict__ wnd, uint32_t frame, const float2 *__restrict__ tw)
    {
        using PL = fplan<LOGH>;
        constexpr int H = PL::N, T = PL::T, N = 2 * H;
        __shared__ float2 lds_[fplan<LOGH>::LDS];
        float2 *const buf = lds_, *const scr = lds_ + fplan<LOGH>::SCR;
        const int ch = blockIdx.x, tid = threadIdx.x;
        typename fplan<LOGH>::real rf;
        rf.load(tw, TWN, tid);
        rf.prepare();
        const float2 *sp = spec + size_t(ch) * N;
        for (int k = tid; k < H; k += T)
        {
            if (k == 0)
                buf[0] = make_float2(sp[0].x, sp[H].x);
            else
            {
                const float2 a = sp[k], c = sp[N - k];
                buf[k] = make_float2(0.5f * (a.x + c.x), 0.5f * (a.y - c.y));
            }
        }
        __syncthreads();
        rf.inverse(buf, scr, tid);
        float2 *line = reinterpret_cast<float2 *>(line0 + size_t(ch) * line_pitch);
        const uint32_t first = uint32_t(H) - frame;                    // the last 2*frame samples, in pairs
        overlap_add(line, reinterpret_cast<const float2 *>(wnd), frame, tid, T, 1.0f / float(N), nullptr,
                    [&](uint32_t m) { return buf[first + m]; });
    }

    // the streaming side of process(): new samples into the analysis buffers, finished samples out of the lines
    // grid (pieces of 256, channels, handlers + 1): z == 0 is the input, z - 1 the handler
    __global__ __launch_bounds__(256)
    void splitter_io_kernel(float *in_buf, size_t in_pitch, uint32_t in_pos, const float *__restrict__ src, size_t src_stride,
                            const float *__restrict__ lines, size_t line_pitch, uint32_t line_pos, const out_table outs,
                            size_t out_stride, size_t out_pos, uint32_t channels, uint32_t n)
    {
        const uint32_t i = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y, z = blockIdx.z;
        if (i >= n)
            return;
        if (z == 0)
            in_buf[size_t(ch) * in_pitch + in_pos + i] = (src != nullptr) ? src[size_t(ch) * src_stride + i] : 0.0f;
        else
        {
            float *o = outs.at(z - 1);
            if (o != nullptr)
                o[size_t(ch) * out_stride + out_pos + i] = lines[(size_t(z - 1) * channels + ch) * line_pitch + line_pos + i];
        }
    }
} // namespace

struct mi_splitter_bank
{
    struct handler_t
    {
        int          mode = H_OFF;
        float       *d_mask = nullptr;      // owned copy of the gains
        size_t       mask_stride = 0, mask_cap = 0;
        mi_splitter_func_t func = nullptr;
        void        *object = nullptr, *subject = nullptr;
    };
    uint32_t    channels = 0, handlers = 0, max_rank = 0, rank = 0, chunk_rank = 0;
    int32_t     user_chunk_rank = 0;
    float       phase = 0.0f;
    bool        update = true, desc_dirty = true;
    uint32_t    fill = 0;                   // nFrameSize
    uint32_t    bindings = 0;
    std::vector<handler_t> h;
    std::vector<uint8_t> has_sink;
    float      *d_in = nullptr, *d_in2 = nullptr, *d_lines = nullptr, *d_wnd = nullptr;    // d_in: current analysis buffers, d_in2: the other half of the pair
    float2     *d_spec = nullptr, *d_tmp = nullptr, *d_big = nullptr;     // d_big: scratch of the four-step transform (ranks >= 15)
    handler_desc *d_desc = nullptr;
    float     **d_outs = nullptr;
    std::vector<float *> outs_shadow;
    const float2 *d_tw = nullptr;
    size_t      pitch = 0;                  // 2^max_rank
};

namespace
{
    uint32_t effective_chunk_rank(const mi_splitter_bank *b, uint32_t rank)
    {
        if (b->user_chunk_rank <= 0)
            return rank;
        const int32_t r = b->user_chunk_rank;
        return uint32_t((r < 5) ? 5 : (r > int32_t(rank)) ? int32_t(rank) : r);          // lsp_limit(user, 5, rank)
    }

    int splitter_clear(mi_splitter_bank *b, hipStream_t st)                              // SpectralSplitter.cpp:246-258
    {
        MI_HIP_CHECK(hipMemsetAsync(b->d_in, 0, size_t(b->channels) * b->pitch * sizeof(float), st));
        MI_HIP_CHECK(hipMemsetAsync(b->d_in2, 0, size_t(b->channels) * b->pitch * sizeof(float), st));
        // every line: a handler nobody listens to never adds to its line, and bind() clears it anyway
        MI_HIP_CHECK(hipMemsetAsync(b->d_lines, 0, size_t(b->handlers) * b->channels * b->pitch * sizeof(float), st));
        return MI_OK;
    }

    int splitter_apply_settings(mi_splitter_bank *b, hipStream_t st)                     // update_settings(), :224-244
    {
        b->rank = std::min(b->rank, b->max_rank);
        b->chunk_rank = effective_chunk_rank(b, b->rank);
        const size_t frame = size_t(1) << (b->chunk_rank - 1);
        std::vector<float> w(frame * 2);
        mi::make_window(w.data(), frame * 2, MI_WINDOW_SQR_COSINE);
        MI_HIP_CHECK(hipMemcpyAsync(b->d_wnd, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice, st));
        MI_HIP_CHECK(hipStreamSynchronize(st));
        const int r = splitter_clear(b, st);
        if (r != MI_OK)
            return r;
        b->fill = uint32_t(float(frame) * (b->phase * 0.5f));
        b->update = false;
        return MI_OK;
    }

    int splitter_upload_desc(mi_splitter_bank *b, hipStream_t st)
    {
        std::vector<handler_desc> d(b->handlers);
        for (uint32_t i = 0; i < b->handlers; ++i)
        {
            d[i].mask = b->h[i].d_mask;
            d[i].mask_stride = b->h[i].mask_stride;
            d[i].mode = uint32_t(b->h[i].mode);
            d[i].has_sink = b->has_sink[i];
        }
        MI_HIP_CHECK(hipMemcpyAsync(b->d_desc, d.data(), d.size() * sizeof(handler_desc), hipMemcpyHostToDevice, st));
        MI_HIP_CHECK(hipStreamSynchronize(st));
        b->desc_dirty = false;
        return MI_OK;
    }

    #define MI_LOGH_SWITCH(lh, CALL)                    \
        switch (lh)                                     \
        {                                               \
            case 4:  { CALL(4);  break; }               \
            case 5:  { CALL(5);  break; }               \
            case 6:  { CALL(6);  break; }               \
            case 7:  { CALL(7);  break; }               \
            case 8:  { CALL(8);  break; }               \
            case 9:  { CALL(9);  break; }               \
            case 10: { CALL(10); break; }               \
            case 11: { CALL(11); break; }               \
            case 12: { CALL(12); break; }               \
            default: { CALL(13); break; }               \
        }

    out_table splitter_outs(const mi_splitter_bank *b)
    {
        out_table t;
        for (uint32_t i = 0; i < OUTS_BY_VALUE; ++i)
            t.p[i] = (i < b->handlers) ? b->outs_shadow[i] : nullptr;
        t.more = (b->handlers > OUTS_BY_VALUE) ? b->d_outs : nullptr;
        return t;
    }

    template <int LH> constexpr bool hop_in_registers = (LH <= 12) && !fplan<LH>::radix16 && (mi_fft::plan<LH>::T == mi_fft::plan<LH>::TB);

    // May the hops of one streaming call share a launch (splitter_hop_kernel, hops > 1)?  The register path must exist for
    // the size, the hop must be the whole half frame, and every handler that is listened to must be a mask.
    bool splitter_hops_fuse(const mi_splitter_bank *b, const float *src, size_t src_stride)
    {
        if (b->chunk_rank != b->rank || mi::test_path("splitter_hop_launches"))     // (a launch per hop: what chunked frames take anyway)
            return false;
        bool fast = false;
        #define MI_CALL(LH) fast = hop_in_registers<LH>
        MI_LOGH_SWITCH(int(b->rank) - 1, MI_CALL)
        #undef MI_CALL
        if (!fast)
            return false;
        for (uint32_t i = 0; i < b->handlers; ++i)
            if (b->has_sink[i] && b->h[i].mode != H_MASK)
                return false;
        // the blocks that follow the first come in as pairs of samples
        return src == nullptr || ((reinterpret_cast<uintptr_t>(src) % 8) == 0 && (src_stride % 2) == 0);
    }

    bool splitter_has_callbacks(const mi_splitter_bank *b)
    {
        for (uint32_t i = 0; i < b->handlers; ++i)
            if (b->h[i].mode == H_CALLBACK)
                return true;
        return false;
    }

    // src / src_stride / ingest_n / out_stride / out_pos: the fused streaming of splitter_hop_kernel (ingest_n == 0: none)
    int splitter_hop(mi_splitter_bank *b, hipStream_t st, const float *src, size_t src_stride, uint32_t ingest_n,
                     size_t out_stride, size_t out_pos, uint32_t hops = 1)
    {
        const int lh = int(b->rank) - 1;
        const uint32_t frame = 1u << (b->chunk_rank - 1);
        const bool callbacks = splitter_has_callbacks(b);
        // few channels: one workgroup per handler (the forward transform is repeated, the device is filled)
        // (the several-hops form at any channel count: 82 against 105 us per 4096-sample call at 1024 channels x 4 bands,
        // tests/experiments/splitter_rate.py)
        dim3 grid(b->channels, (!callbacks && b->handlers > 1 && (b->channels <= 512 || hops > 1)) ? b->handlers : 1);
        // (several hops per launch with TWO handlers per workgroup sharing the forward transform -- a quarter less arithmetic, half the
        // waves per SIMD -- measured 27.7 against 24.4 us per block at 256 channels x 4 bands, rank 12:
        // profiles/r03_experiments/splitter_hops_per_launch.txt; removed in round 6)
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        mi::take_profile_events(&ev0, &ev1);
        if (!callbacks)
        {
            #define MI_ARGS b->d_in, b->d_in2, b->pitch, b->d_lines, b->pitch, b->channels, b->d_desc, b->handlers, b->d_wnd, frame, \
                (float2 *)nullptr, b->d_tw, src, src_stride, ingest_n, splitter_outs(b), out_stride, out_pos, hops
            if (hops > 1 && grid.y > 1)
            {
                #define MI_CALL(LH) if constexpr (hop_in_registers<LH>) \
                    MI_LAUNCH((splitter_hop_kernel<LH, false, true, 1>), grid, dim3(fplan<LH>::T), 0, st, ev0, ev1, MI_ARGS)
                MI_LOGH_SWITCH(lh, MI_CALL)
                #undef MI_CALL
            }
            else if (hops > 1)
            {
                #define MI_CALL(LH) if constexpr (hop_in_registers<LH>) \
                    MI_LAUNCH((splitter_hop_kernel<LH, false, false, 1>), grid, dim3(fplan<LH>::T), 0, st, ev0, ev1, MI_ARGS)
                MI_LOGH_SWITCH(lh, MI_CALL)
                #undef MI_CALL
            }
            else if (grid.y > 1)
            {
                #define MI_CALL(LH) MI_LAUNCH((splitter_hop_kernel<LH, false, true>), grid, dim3(fplan<LH>::T), 0, st, ev0, ev1, MI_ARGS)
                MI_LOGH_SWITCH(lh, MI_CALL)
                #undef MI_CALL
            }
            else
            {
                #define MI_CALL(LH) MI_LAUNCH((splitter_hop_kernel<LH, false, false>), grid, dim3(fplan<LH>::T), 0, st, ev0, ev1, MI_ARGS)
                MI_LOGH_SWITCH(lh, MI_CALL)
                #undef MI_CALL
            }
            #undef MI_ARGS
            MI_HIP_CHECK(hipGetLastError());
            std::swap(b->d_in, b->d_in2);
            return MI_OK;
        }
        #define MI_CALL(LH) MI_LAUNCH((splitter_hop_kernel<LH, true, false>), grid, dim3(fplan<LH>::T), 0, st, ev0, ev1, \
            b->d_in, b->d_in2, b->pitch, b->d_lines, b->pitch, b->channels, b->d_desc, b->handlers, b->d_wnd, frame, b->d_spec, \
            b->d_tw, (const float *)nullptr, size_t(0), 0u, splitter_outs(b), size_t(0), size_t(0), 1u)
        MI_LOGH_SWITCH(lh, MI_CALL)
        #undef MI_CALL
        MI_HIP_CHECK(hipGetLastError());
        std::swap(b->d_in, b->d_in2);
        for (uint32_t i = 0; i < b->handlers; ++i)
        {
            mi_splitter_bank::handler_t &h = b->h[i];
            if (h.mode != H_CALLBACK)
                continue;
            h.func(h.object, h.subject, reinterpret_cast<float *>(b->d_tmp), reinterpret_cast<const float *>(b->d_spec), b->rank,
                   b->channels, st);
            if (!b->has_sink[i])
                continue;
            float *line0 = b->d_lines + size_t(i) * b->channels * b->pitch;
            #define MI_CALL(LH) hipLaunchKernelGGL((splitter_inverse_kernel<LH>), grid, dim3(fplan<LH>::T), 0, st, \
                line0, b->pitch, b->d_tmp, b->d_wnd, frame, b->d_tw)
            MI_LOGH_SWITCH(lh, MI_CALL)
            #undef MI_CALL
            MI_HIP_CHECK(hipGetLastError());
        }
        return MI_OK;
    }

    int splitter_bind_common(mi_splitter_bank *b, uint32_t handler, int mode, hipStream_t st)
    {
        mi_splitter_bank::handler_t &h = b->h[handler];
        if (h.mode == H_OFF)
            ++b->bindings;
        h.mode = mode;
        b->desc_dirty = true;
        // bind() clears the handler's output line (SpectralSplitter.cpp:159-160)
        MI_HIP_CHECK(hipMemsetAsync(b->d_lines + size_t(handler) * b->channels * b->pitch, 0,
                                    size_t(b->channels) * b->pitch * sizeof(float), st));
        return MI_OK;
    }

    // ---- frames above 2^14 samples (ranks 15 .. 18): the hop as plain launches around the four-step transform of spectral.hip
    // (mi::big_fft_run), complex all the way as in the reference (pcomplex_r2c ... packed_reverse_fft, real part kept)

    // spec[n] = (in[n], 0): the analysis buffer as a complex sequence
    __global__ __launch_bounds__(256)
    void splitter_big_load_kernel(float2 *spec, const float *__restrict__ in_buf, size_t in_pitch, uint32_t N)
    {
        const uint32_t n = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
        if (n < N)
            spec[size_t(ch) * N + n] = make_float2(in_buf[size_t(ch) * in_pitch + n], 0.0f);
    }

    // prod[k] = spec[k] * g[k], the N real gains in FFT order (FFTCrossover::spectral_func, FFTCrossover.cpp:137-139)
    __global__ __launch_bounds__(256)
    void splitter_big_mask_kernel(float2 *prod, const float2 *__restrict__ spec, const float *__restrict__ mask, size_t mask_stride,
                                  uint32_t N)
    {
        const uint32_t k = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
        if (k < N)
        {
            const float g = mask[size_t(ch) * mask_stride + k];
            const float2 v = spec[size_t(ch) * N + k];
            prod[size_t(ch) * N + k] = make_float2(v.x * g, v.y * g);
        }
    }

    // the handler's line moves on by one frame and takes the windowed last 2*frame samples of the way back (res != NULL:
    // real parts, scaled) or the first 2*frame samples of the analysis buffer (a handler without a spectral function, :330)
    __global__ __launch_bounds__(256)
    void splitter_big_ola_kernel(float *line0, size_t line_pitch, const float2 *__restrict__ res, const float *__restrict__ in_buf,
                                 size_t in_pitch, const float *__restrict__ wnd, uint32_t frame, uint32_t N, float scale)
    {
        const uint32_t m = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
        if (m >= frame)
            return;
        float y0, y1;
        if (res != nullptr)
        {
            const float2 *r = res + size_t(ch) * N + (N - 2 * frame);
            y0 = r[m].x * scale;
            y1 = r[m + frame].x * scale;
        }
        else
        {
            y0 = in_buf[size_t(ch) * in_pitch + m];
            y1 = in_buf[size_t(ch) * in_pitch + m + frame];
        }
        float *line = line0 + size_t(ch) * line_pitch;
        const float prev = line[m + frame];
        line[m]         = fmaf(y0, wnd[m], prev);
        line[m + frame] = y1 * wnd[m + frame];
    }

    int splitter_hop_big(mi_splitter_bank *b, hipStream_t st)
    {
        const uint32_t N = 1u << b->rank, frame = 1u << (b->chunk_rank - 1);
        const dim3 gN((N + 255) / 256, b->channels), gF((frame + 255) / 256, b->channels);
        hipLaunchKernelGGL(splitter_big_load_kernel, gN, dim3(256), 0, st, b->d_spec, b->d_in, b->pitch, N);
        MI_HIP_CHECK(hipGetLastError());
        int r = mi::big_fft_run(false, b->d_spec, b->d_spec, b->d_big, b->rank, b->channels, b->d_tw, st);
        if (r != MI_OK)
            return r;
        for (uint32_t i = 0; i < b->handlers; ++i)
        {
            mi_splitter_bank::handler_t &h = b->h[i];
            if (h.mode == H_OFF)
                continue;
            const float2 *res = nullptr;
            if (h.mode == H_CALLBACK)                       // the function runs whether or not somebody listens (:316-327)
                h.func(h.object, h.subject, reinterpret_cast<float *>(b->d_tmp), reinterpret_cast<const float *>(b->d_spec), b->rank,
                       b->channels, st);
            if (!b->has_sink[i])
                continue;
            if (h.mode == H_MASK)
            {
                hipLaunchKernelGGL(splitter_big_mask_kernel, gN, dim3(256), 0, st, b->d_tmp, b->d_spec, h.d_mask, h.mask_stride, N);
                MI_HIP_CHECK(hipGetLastError());
            }
            if (h.mode != H_COPY)
            {
                r = mi::big_fft_run(true, b->d_tmp, b->d_tmp, b->d_big, b->rank, b->channels, b->d_tw, st);
                if (r != MI_OK)
                    return r;
                res = b->d_tmp;
            }
            hipLaunchKernelGGL(splitter_big_ola_kernel, gF, dim3(256), 0, st, b->d_lines + size_t(i) * b->channels * b->pitch, b->pitch,
                               res, b->d_in, b->pitch, b->d_wnd, frame, N, 1.0f / float(N));
            MI_HIP_CHECK(hipGetLastError());
        }
        // the analysis buffer moves on by one frame, into the other buffer of the pair
        MI_HIP_CHECK(hipMemcpy2DAsync(b->d_in2, b->pitch * sizeof(float), b->d_in + frame, b->pitch * sizeof(float),
                                      size_t(N - frame) * sizeof(float), b->channels, hipMemcpyDeviceToDevice, st));
        std::swap(b->d_in, b->d_in2);
        return MI_OK;
    }
} // namespace

extern "C" {

int mi_splitter_bank_create(mi_splitter_bank_t **bank, uint32_t channels, uint32_t max_rank, uint32_t handlers)
{
    MI_REQUIRE(bank != nullptr, MI_EINVAL, "mi_splitter_bank_create: NULL result pointer");
    *bank = nullptr;
    MI_REQUIRE(channels > 0 && handlers > 0, MI_EINVAL, "mi_splitter_bank_create: no channels or handlers");
    MI_REQUIRE(max_rank >= 5 && max_rank <= 18, MI_EINVAL, "mi_splitter_bank_create: max_rank %u outside 5..18", max_rank);
    MI_REQUIRE(mi_dspu_device_count() > 0, MI_ENODEV, "no HIP device available (there is no CPU fallback)");
    mi_splitter_bank *b = new (std::nothrow) mi_splitter_bank();
    MI_REQUIRE(b != nullptr, MI_ENOMEM, "mi_splitter_bank_create: out of host memory");
    b->channels = channels;
    b->handlers = handlers;
    b->rank = b->max_rank = max_rank;
    b->pitch = size_t(1) << max_rank;
    b->h.resize(handlers);
    b->has_sink.assign(handlers, 1);
    b->outs_shadow.assign(handlers, nullptr);
    int twn = 0;
    int r = mi::fft_twiddles(&b->d_tw, &twn);
    hipError_t e = hipSuccess;
    if (r == MI_OK)
    {
        const size_t row = size_t(channels) * b->pitch;
        e = hipMalloc(reinterpret_cast<void **>(&b->d_in), row * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_in2), row * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_lines), row * handlers * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_wnd), b->pitch * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_desc), handlers * sizeof(handler_desc));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_outs), handlers * sizeof(float *));
        if (e == hipSuccess) e = hipMemset(b->d_in, 0, row * sizeof(float));
        if (e == hipSuccess) e = hipMemset(b->d_in2, 0, row * sizeof(float));
        if (e == hipSuccess) e = hipMemset(b->d_lines, 0, row * handlers * sizeof(float));
        if (e == hipSuccess) e = hipMemset(b->d_outs, 0, handlers * sizeof(float *));
        if (e == hipSuccess && max_rank > 14)               // frames above 2^14: spectrum, product / result, transform scratch
        {
            const size_t n = size_t(channels) << max_rank;
            e = hipMalloc(reinterpret_cast<void **>(&b->d_spec), n * sizeof(float2));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_tmp), n * sizeof(float2));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_big), n * sizeof(float2));
        }
    }
    if (r != MI_OK || e != hipSuccess)
    {
        mi_splitter_bank_destroy(b);
        return (r != MI_OK) ? r : mi::fail(MI_EHIP, "mi_splitter_bank_create: %s", hipGetErrorString(e));
    }
    *bank = b;
    return MI_OK;
}

int mi_splitter_bank_destroy(mi_splitter_bank_t *b)
{
    if (b == nullptr)
        return MI_OK;
    for (mi_splitter_bank::handler_t &h : b->h)
        (void)hipFree(h.d_mask);
    (void)hipFree(b->d_in); (void)hipFree(b->d_in2); (void)hipFree(b->d_lines); (void)hipFree(b->d_wnd); (void)hipFree(b->d_desc);
    (void)hipFree(b->d_outs); (void)hipFree(b->d_spec); (void)hipFree(b->d_tmp); (void)hipFree(b->d_big);
    delete b;
    return MI_OK;
}

int mi_splitter_bank_set_rank(mi_splitter_bank_t *b, uint32_t rank)                  // SpectralSplitter.cpp:266-273
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_splitter_bank_set_rank: NULL bank");
    if (rank == b->rank || rank > b->max_rank)
        return MI_OK;
    MI_REQUIRE(rank >= 5, MI_EINVAL, "mi_splitter_bank_set_rank: rank %u below 5", rank);
    b->rank = rank;
    b->update = true;
    return MI_OK;
}

int mi_splitter_bank_set_chunk_rank(mi_splitter_bank_t *b, int32_t rank)             // :275-282
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_splitter_bank_set_chunk_rank: NULL bank");
    if (rank == b->user_chunk_rank)
        return MI_OK;
    b->user_chunk_rank = rank;
    b->update = true;
    return MI_OK;
}

int mi_splitter_bank_set_phase(mi_splitter_bank_t *b, float phase)                   // :260-264
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_splitter_bank_set_phase: NULL bank");
    b->phase = (phase < 0.0f) ? 0.0f : (phase > 1.0f) ? 1.0f : phase;
    b->update = true;
    return MI_OK;
}

int mi_splitter_bank_get(const mi_splitter_bank_t *b, uint32_t *rank, uint32_t *chunk_rank, uint32_t *latency, uint32_t *remaining)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_splitter_bank_get: NULL bank");
    const uint32_t r = std::min(b->rank, b->max_rank);
    // latency() with settings pending takes the chunk rank from the new settings but falls back to nRank as it is (:284-293)
    const uint32_t cr = b->update ? ((b->user_chunk_rank > 0) ? effective_chunk_rank(b, r) : b->rank) : b->chunk_rank;
    if (rank != nullptr) *rank = b->rank;
    if (chunk_rank != nullptr) *chunk_rank = cr;
    if (latency != nullptr) *latency = 1u << cr;
    if (remaining != nullptr)
    {
        const uint32_t frame = 1u << (cr - 1);
        const uint32_t fill = b->update ? uint32_t(float(frame) * (b->phase * 0.5f)) : b->fill;
        *remaining = (fill >= frame) ? frame : frame - fill;           // a full frame is transformed first, then refilled
    }
    return MI_OK;
}

int mi_splitter_bank_bind_copy(mi_splitter_bank_t *b, uint32_t handler, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_splitter_bank_bind_copy: NULL bank");
    MI_REQUIRE(handler < b->handlers, MI_EINVAL, "mi_splitter_bank_bind_copy: handler %u out of range", handler);
    return splitter_bind_common(b, handler, H_COPY, mi::as_stream(stream));
}

int mi_splitter_bank_bind_mask(mi_splitter_bank_t *b, uint32_t handler, const float *mask, size_t mask_stride, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_splitter_bank_bind_mask: NULL bank");
    MI_REQUIRE(handler < b->handlers, MI_EINVAL, "mi_splitter_bank_bind_mask: handler %u out of range", handler);
    MI_REQUIRE(mask != nullptr, MI_EINVAL, "mi_splitter_bank_bind_mask: NULL mask");
    const size_t N = size_t(1) << b->rank;
    MI_REQUIRE(mask_stride == 0 || mask_stride >= N, MI_EINVAL, "mi_splitter_bank_bind_mask: stride %zu below 2^rank", mask_stride);
    hipStream_t st = mi::as_stream(stream);
    mi_splitter_bank::handler_t &h = b->h[handler];
    const size_t rows = (mask_stride == 0) ? 1 : b->channels;
    const size_t need = rows * N;
    if (need > h.mask_cap)
    {
        MI_HIP_CHECK(hipStreamSynchronize(st));                         // a hop in flight may still read the old gains
        (void)hipFree(h.d_mask);
        h.d_mask = nullptr;
        h.mask_cap = 0;
        MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&h.d_mask), need * sizeof(float)));
        h.mask_cap = need;
        b->desc_dirty = true;
    }
    // Only the real part of the inverse is kept (pcomplex_c2r), so a real gain acts through its even part
    // (g[k] + g[N - k]) / 2: that is what goes to the device (the same float the kernels used to form per bin and hop; a
    // table that is even already comes out as it went in, x + x and the halving are exact).
    std::vector<float> even;
    try { even.resize(need); } catch (...) { return MI_ENOMEM; }
    for (size_t r = 0; r < rows; ++r)
    {
        const float *g = mask + r * ((mask_stride == 0) ? N : mask_stride);
        float *e = even.data() + r * N;
        e[0] = g[0];
        e[N / 2] = g[N / 2];
        for (size_t k = 1; k < N / 2; ++k)
            e[k] = e[N - k] = 0.5f * (g[k] + g[N - k]);
    }
    MI_HIP_CHECK(hipMemcpyAsync(h.d_mask, even.data(), need * sizeof(float), hipMemcpyHostToDevice, st));
    MI_HIP_CHECK(hipStreamSynchronize(st));
    const size_t new_stride = (mask_stride == 0) ? 0 : N;
    if (h.mask_stride != new_stride)
    {
        h.mask_stride = new_stride;
        b->desc_dirty = true;
    }
    if (h.mode == H_MASK)
        return MI_OK;                                                   // new gains for a bound handler: the line lives on
    return splitter_bind_common(b, handler, H_MASK, st);
}

int mi_splitter_bank_bind_callback(mi_splitter_bank_t *b, uint32_t handler, mi_splitter_func_t func, void *object, void *subject,
                                   void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_splitter_bank_bind_callback: NULL bank");
    MI_REQUIRE(handler < b->handlers, MI_EINVAL, "mi_splitter_bank_bind_callback: handler %u out of range", handler);
    MI_REQUIRE(func != nullptr, MI_EINVAL, "mi_splitter_bank_bind_callback: NULL function");
    if (b->d_spec == nullptr)
    {
        const size_t n = size_t(b->channels) * b->pitch;
        MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_spec), n * sizeof(float2)));
        MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_tmp), n * sizeof(float2)));
    }
    mi_splitter_bank::handler_t &h = b->h[handler];
    h.func = func;
    h.object = object;
    h.subject = subject;
    return splitter_bind_common(b, handler, H_CALLBACK, mi::as_stream(stream));
}

int mi_splitter_bank_unbind(mi_splitter_bank_t *b, uint32_t handler)                 // :165-180
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_splitter_bank_unbind: NULL bank");
    MI_REQUIRE(handler < b->handlers, MI_EINVAL, "mi_splitter_bank_unbind: handler %u out of range", handler);
    mi_splitter_bank::handler_t &h = b->h[handler];
    if (h.mode == H_OFF)
        return mi::fail(MI_ESTATE, "mi_splitter_bank_unbind: handler %u is not bound", handler);
    h.mode = H_OFF;
    h.func = nullptr;
    --b->bindings;
    b->desc_dirty = true;
    return MI_OK;
}

int mi_splitter_bank_clear(mi_splitter_bank_t *b, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_splitter_bank_clear: NULL bank");
    return splitter_clear(b, mi::as_stream(stream));
}

// what the launches of a call take by value from the host: the fill of the frame, which analysis buffer of the pair is current
static uint64_t splitter_bank_positions(const void *bank)
{
    const mi_splitter_bank *b = static_cast<const mi_splitter_bank *>(bank);
    uint64_t h = mi::position_mix(b->fill, uint64_t(reinterpret_cast<uintptr_t>(b->d_in)));
    return mi::position_mix(h, (uint64_t(b->update) << 1) | uint64_t(b->desc_dirty));
}

// who listens: a handler whose output pointer is NULL has no sink (the descriptors on the device follow)
static int splitter_listeners(mi_splitter_bank_t *b, float *const *outs, hipStream_t st)
{
    bool outs_changed = false;
    for (uint32_t i = 0; i < b->handlers; ++i)
    {
        float *o = (outs != nullptr) ? outs[i] : nullptr;
        const uint8_t sink = (o != nullptr && b->h[i].mode != H_OFF) ? 1 : 0;
        if (sink != b->has_sink[i])
        {
            b->has_sink[i] = sink;
            b->desc_dirty = true;
        }
        if (b->h[i].mode == H_OFF)
            o = nullptr;
        if (o != b->outs_shadow[i])
        {
            b->outs_shadow[i] = o;
            outs_changed = true;
        }
    }
    if (outs_changed && b->handlers > OUTS_BY_VALUE)
    {
        MI_HIP_CHECK(hipMemcpyAsync(b->d_outs, b->outs_shadow.data(), b->handlers * sizeof(float *), hipMemcpyHostToDevice, st));
        MI_HIP_CHECK(hipStreamSynchronize(st));
    }
    if (b->desc_dirty)
    {
        const int r = splitter_upload_desc(b, st);
        if (r != MI_OK)
            return r;
    }
    return MI_OK;
}

static int splitter_wave_try(mi_splitter_bank_t *b, const split_blocks &tab, size_t run, size_t count,
                             size_t in_stride, size_t out_stride, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);

int mi_splitter_bank_process(mi_splitter_bank_t *b, float *const *outs, const float *in, size_t count, size_t out_stride,
                             size_t in_stride, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_splitter_bank_process: NULL bank");
    hipStream_t st = mi::as_stream(stream);
    {
        const int rc = mi::capture_touch(st, b, "spectral splitter", splitter_bank_positions);
        if (rc != MI_OK)
            return rc;
    }
    if (b->update)
    {
        const int r = splitter_apply_settings(b, st);
        if (r != MI_OK)
            return r;
    }
    if (b->bindings == 0 || count == 0)                                // SpectralSplitter.cpp:299-300
        return MI_OK;
    // The reference hands its bands to sink functions (SpectralSplitter.cpp:344-356): a band cannot be the block it reads.  Here
    // the bands are buffers, and the kernels write a band's finished samples before they take the caller's samples of the
    // same place: an output row that overlaps the input rows is refused rather than filled with something else.
    if (in != nullptr && outs != nullptr)
    {
        const uintptr_t i0 = reinterpret_cast<uintptr_t>(in), in_bytes = ((size_t(b->channels) - 1) * in_stride + count) * sizeof(float);
        const uintptr_t out_bytes = ((size_t(b->channels) - 1) * out_stride + count) * sizeof(float);
        for (uint32_t i = 0; i < b->handlers; ++i)
        {
            const uintptr_t o0 = reinterpret_cast<uintptr_t>(outs[i]);
            MI_REQUIRE(outs[i] == nullptr || o0 + out_bytes <= i0 || i0 + in_bytes <= o0, MI_EINVAL,
                       "mi_splitter_bank_process: the output rows of handler %u overlap the input rows", i);
        }
    }
    {
        const int r = splitter_listeners(b, outs, st);
        if (r != MI_OK)
            return r;
    }
    const uint32_t N = 1u << b->rank, frame = 1u << (b->chunk_rank - 1), gap = N - frame;
    size_t done = 0;
    const bool callbacks = splitter_has_callbacks(b);
    while (done < count)
    {
        if (b->fill >= frame)                                           // a frame is complete: transform (:311-356)
        {
            // a whole frame follows in this call: the transform kernel takes it in and hands the finished frame out itself
            const bool big = b->rank > 14;
            const bool fused = !big && !callbacks && (count - done >= frame);
            const float *src = (in != nullptr) ? in + done : nullptr;
            // EIGHT or more whole blocks of N samples follow in this call (rank 12, listening masks shared by the channels): a wave
            // per channel and segment on the wave-resident transform, the blocks being column slices of the caller's buffers
            // (splitter_wave_blocks_kernel; shorter calls stay on the workgroup kernels: the launch has 25 us of its own)
            if (fused && src != nullptr && outs != nullptr && b->rank == 12 && (count - done) % N == 0 && (count - done) / N >= 8 &&
                b->handlers <= OUTS_BY_VALUE && splitter_hops_fuse(b, src, in_stride) && (out_stride % 2) == 0)
            {
                bool went = true;
                while (went && done < count)
                {
                    const size_t cap = std::min<size_t>(SPLIT_BLOCKS_MAX, SPLIT_PTRS_MAX / b->handlers);
                    const size_t run = std::min<size_t>((count - done) / N, cap);
                    split_blocks tab;
                    tab.per = 2;
                    for (size_t q = 0; q < run; ++q)
                    {
                        tab.src[q] = in + done + q * N;
                        for (uint32_t i = 0; i < b->handlers; ++i)
                            tab.out[q * b->handlers + i] = (b->h[i].mode == H_OFF || outs[i] == nullptr) ? nullptr : outs[i] + done + q * N;
                    }
                    hipEvent_t ev0 = nullptr, ev1 = nullptr;
                    mi::take_profile_events(&ev0, &ev1);
                    const int rw = splitter_wave_try(b, tab, run, N, in_stride, out_stride, st, ev0, ev1);
                    if (rw < 0)
                        return rw;
                    went = rw == 1;
                    if (went)
                    {
                        std::swap(b->d_in, b->d_in2);
                        b->fill = frame;
                        done += run * N;
                    }
                }
                if (done >= count)
                    return MI_OK;
                if (went)
                    continue;
            }
            const uint32_t hops = (fused && count - done >= 2 * size_t(frame) && splitter_hops_fuse(b, src, in_stride))
                                ? uint32_t(std::min<size_t>((count - done) / frame, 1u << 20)) : 1;
            const int r = big   ? splitter_hop_big(b, st)
                        : fused ? splitter_hop(b, st, src, in_stride, frame, out_stride, done, hops)
                                : splitter_hop(b, st, nullptr, 0, 0, 0, 0);
            if (r != MI_OK)
                return r;
            b->fill = 0;
            if (fused)
            {
                b->fill = frame;
                done += size_t(hops) * frame;
                continue;
            }
        }
        const uint32_t n = uint32_t(std::min<size_t>(frame - b->fill, count - done));
        const dim3 grid((n + 255) / 256, b->channels, b->handlers + 1);
        hipLaunchKernelGGL(splitter_io_kernel, grid, dim3(256), 0, st, b->d_in, b->pitch, gap + b->fill,
                           (in != nullptr) ? in + done : (const float *)nullptr, in_stride, b->d_lines, b->pitch, b->fill,
                           splitter_outs(b), out_stride, done, b->channels, n);
        MI_HIP_CHECK(hipGetLastError());
        b->fill += n;
        done += n;
    }
    return MI_OK;
}

// A run of `run` blocks of `count` samples (tab: their input rows and the handlers' output rows) on splitter_wave_blocks_kernel if it
// qualifies: rank 12, blocks of exactly one frame, one to four
// listening masks shared by the channels, every output buffer of the run apart from every other (the segments of a channel's run
// go side by side).  Returns 1 if the run went out this way (the caller swaps the analysis buffers), 0 if it does not qualify.
static int splitter_wave_try(mi_splitter_bank_t *b, const split_blocks &tab, size_t run, size_t count,
                             size_t in_stride, size_t out_stride, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1)
{
    const uint32_t nh = b->handlers;
    wave_bands wb{};
    uint32_t nb = 0;
    bool waves = b->rank == 12 && b->chunk_rank == b->rank && count == (size_t(1) << b->rank) && !mi::compat_bits();
    for (uint32_t i = 0; i < nh && waves; ++i)
    {
        if (!b->has_sink[i])
            continue;
        if (b->h[i].mode != H_MASK || b->h[i].mask_stride != 0 || nb == 4)
        {
            waves = false;
            break;
        }
        wb.gain[nb] = b->h[i].d_mask;
        wb.line[nb] = b->d_lines + size_t(i) * b->channels * b->pitch;
        wb.handler[nb] = i;
        ++nb;
    }
    waves = waves && nb >= 1;
    if (waves)
    {
        // Every output of the run apart from every other, ROW BY ROW: an output is `channels` rows of `count` samples, out_stride
        // apart -- the column slices of one [channels][blocks x count] buffer (what a long process() call hands over) interleave
        // without touching, although their whole spans overlap (the span test sent every such call to the workgroup
        // kernels).  Rows a + c S and b + c' S meet iff |(b - a) + k S| < L for a k with |k| < channels.
        std::vector<uintptr_t> outs_;
        for (size_t q = 0; q < run; ++q)
            for (uint32_t i = 0; i < nb; ++i)
                outs_.push_back(reinterpret_cast<uintptr_t>(tab.out[q * nh + wb.handler[i]]));
        const int64_t S = int64_t(out_stride * sizeof(float)), L = int64_t(count * sizeof(float)), C1 = int64_t(b->channels) - 1;
        auto meet = [&](uintptr_t a, uintptr_t c) -> bool {
            const int64_t d = int64_t(c) - int64_t(a);
            if (S <= 0 || C1 == 0)
                return d < L && -d < L;
            int64_t k0 = d / S;
            if (d - k0 * S < 0)
                --k0;                                       // floor
            const int64_t m = d - k0 * S;                   // 0 <= m < S: row c' of one lies m bytes behind row c' + k0 of the other
            return (m < L && k0 <= C1 && -k0 <= C1) || (S - m < L && k0 + 1 <= C1 && -(k0 + 1) <= C1);
        };
        std::sort(outs_.begin(), outs_.end());
        for (size_t i = 1; i < outs_.size() && waves; ++i)
            waves = outs_[i] != outs_[i - 1];
        // (sorted: an output can only meet the ones that start within its own span)
        const uintptr_t span = uintptr_t(C1 * S + L);
        for (size_t i = 0; i < outs_.size() && waves; ++i)
            for (size_t j = i + 1; j < outs_.size() && waves && outs_[j] < outs_[i] + span; ++j)
                waves = !meet(outs_[i], outs_[j]);          // (outputs against inputs: the run was formed that way)
    }
    if (!waves)
        return 0;
    const int want = int((1024 + b->channels - 1) / b->channels);
    int segs = 1;
    while (segs < SPW && 2 * segs <= want && 2 * segs <= int(run) / 4)
        segs *= 2;
    const dim3 grid((b->channels * unsigned(segs) + SPW - 1) / SPW);
    #define MI_WAVE(NB_) MI_LAUNCH((splitter_wave_blocks_kernel<NB_>), grid, dim3(64 * SPW), 0, st, ev0, ev1, b->d_in, b->d_in2, b->pitch, \
                                   b->pitch, wb, nh, b->d_wnd, b->d_tw, tab, in_stride, out_stride, int(run), int(b->channels), segs)
    switch (nb)
    {
        case 1:  { MI_WAVE(1); break; }
        case 2:  { MI_WAVE(2); break; }
        case 3:  { MI_WAVE(3); break; }
        default: { MI_WAVE(4); break; }
    }
    #undef MI_WAVE
    MI_HIP_CHECK(hipGetLastError());
    return 1;
}

// `blocks` consecutive process() calls in one C call (SpectralSplitter.cpp:295-361 per block): block k reads in[k] and hands
// band i to outs[k * handlers + i].  Runs of blocks of whole frames go out as ONE launch of the several-hops kernel
// (splitter_hops_blocks_kernel: between two hops nothing goes through memory, whichever block they belong to) where that kernel
// applies -- every listening handler a mask, the chunk the whole half frame, more than one handler -- with the samples and the
// state of the calls one by one.
int mi_splitter_bank_process_blocks(mi_splitter_bank_t *b, float *const *outs, const float *const *in, size_t blocks, size_t count,
                                    size_t out_stride, size_t in_stride, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_splitter_bank_process_blocks: NULL bank");
    if (count == 0 || blocks == 0)
        return MI_OK;
    MI_REQUIRE(in != nullptr, MI_EINVAL, "mi_splitter_bank_process_blocks: NULL pointer table");
    hipStream_t st = mi::as_stream(stream);
    const uint32_t nh = b->handlers;
    auto one_block = [&](size_t k) -> int {
        return mi_splitter_bank_process(b, (outs != nullptr) ? outs + k * nh : nullptr, in[k], count, out_stride, in_stride, stream);
    };
    size_t k = 0;
    while (k < blocks)
    {
        {
            const int rc = mi::capture_touch(st, b, "spectral splitter", splitter_bank_positions);
            if (rc != MI_OK)
                return rc;
        }
        if (b->update)
        {
            const int r = splitter_apply_settings(b, st);
            if (r != MI_OK)
                return r;
        }
        const uint32_t frame = 1u << (b->chunk_rank - 1);
        size_t run = 0;
        const bool steady = b->bindings != 0 && outs != nullptr && b->fill >= frame && b->rank <= 14 && !splitter_has_callbacks(b) && nh > 1 &&
                            nh <= OUTS_BY_VALUE && (count % frame) == 0 && in[k] != nullptr && (out_stride % 2) == 0;
        if (steady)
        {
            const int r = splitter_listeners(b, outs + k * nh, st);
            if (r != MI_OK)
                return r;
        }
        if (steady && splitter_hops_fuse(b, in[k], in_stride))
        {
            const size_t cap = std::min<size_t>(SPLIT_BLOCKS_MAX, SPLIT_PTRS_MAX / nh);
            const size_t ob = ((size_t(b->channels) - 1) * out_stride + count) * sizeof(float), ib = ((size_t(b->channels) - 1) * in_stride + count) * sizeof(float);
            auto overlap = [](const void *p, size_t pn, const void *q, size_t qn) -> bool {
                const uintptr_t a0 = reinterpret_cast<uintptr_t>(p), b0 = reinterpret_cast<uintptr_t>(q);
                return p != nullptr && q != nullptr && a0 < b0 + qn && b0 < a0 + pn;
            };
            while (k + run < blocks && run < cap)
            {
                const size_t j = k + run;
                bool ok = in[j] != nullptr && (reinterpret_cast<uintptr_t>(in[j]) % 8) == 0;
                for (uint32_t i = 0; ok && i < nh; ++i)    // the same handlers listen in every block of the run; rows of pairs
                    ok = ((outs[j * nh + i] != nullptr) == (outs[k * nh + i] != nullptr)) && (reinterpret_cast<uintptr_t>(outs[j * nh + i]) % 8) == 0;
                for (size_t q = k; ok && q <= j; ++q)       // nothing the run writes is read by it
                    for (uint32_t i = 0; ok && i < nh; ++i)
                        ok = !overlap(outs[q * nh + i], ob, in[j], ib) && !overlap(outs[j * nh + i], ob, in[q], ib);
                if (!ok)
                    break;
                ++run;
            }
        }
        if (run < 2)
        {
            const int r = one_block(k);
            if (r != MI_OK)
                return r;
            ++k;
            continue;
        }
        split_blocks tab;
        tab.per = uint32_t(count / frame);
        for (size_t q = 0; q < run; ++q)
        {
            tab.src[q] = in[k + q];
            for (uint32_t i = 0; i < nh; ++i)
                tab.out[q * nh + i] = (b->h[i].mode == H_OFF) ? nullptr : outs[(k + q) * nh + i];
        }
        const uint32_t hops = uint32_t(run) * tab.per;
        const int lh = int(b->rank) - 1;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        mi::take_profile_events(&ev0, &ev1);
        // rank 12, blocks of exactly one frame, up to four listening masks shared by the channels, every buffer of the run apart from
        // every other: a wave per channel and segment of the run on the wave-resident transform (splitter_wave_blocks_kernel)
        {
            const int rw = splitter_wave_try(b, tab, run, count, in_stride, out_stride, st, ev0, ev1);
            if (rw < 0)
                return rw;
            if (rw == 1)
            {
                std::swap(b->d_in, b->d_in2);
                b->fill = frame;
                k += run;
                continue;
            }
        }
        #define MI_CALL(LH) if constexpr (hop_in_registers<LH>) \
            MI_LAUNCH((splitter_hops_blocks_kernel<LH>), dim3(b->channels, nh), dim3(fplan<LH>::T), 0, st, ev0, ev1, b->d_in, b->d_in2, b->pitch, \
                      b->d_lines, b->pitch, b->channels, b->d_desc, nh, b->d_wnd, frame, b->d_tw, in_stride, out_stride, hops, tab)
        MI_LOGH_SWITCH(lh, MI_CALL)
        #undef MI_CALL
        MI_HIP_CHECK(hipGetLastError());
        std::swap(b->d_in, b->d_in2);
        b->fill = frame;
        k += run;
    }
    return MI_OK;
}

} // extern "C"
