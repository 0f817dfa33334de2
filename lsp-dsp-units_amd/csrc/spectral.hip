// Short-time spectral units for gfx950:
//   mi_spectral_bank  -- lsp::dspu::SpectralProcessor / MultiSpectralProcessor
//                        (reference: src/main/util/SpectralProcessor.cpp:138-199,
//                         src/main/util/MultiSpectralProcessor.cpp:288-393)
//   mi_analyzer_bank  -- lsp::dspu::Analyzer (reference: src/main/util/Analyzer.cpp:251-409,443-456)
//
// One workgroup owns one channel's frame.  A frame of N = 2^rank real samples goes through an N/2-point
// complex transform held in LDS (fft_device.h): window -> pack pairs -> FFT -> split into the half spectrum.
// What happens between the two transforms is an "operation":
//   NONE      the reference's unbound processor: no transform at all, just window * window overlap-add
//             (SpectralProcessor.cpp:171-172);
//   MASK      spectrum[k] *= mask[k] (real, Hermitian-symmetric gain, shared or per channel) fused in-kernel;
//   CALLBACK  the full N-bin complex spectrum of every channel is written to device memory, the user's
//             function is called on the host with that device pointer and the stream (it may enqueue any
//             kernel -- this is where a cross-channel reduction or an RCCL collective lives), and a full
//             complex inverse transform brings it back (the callback may break Hermitian symmetry, only the
//             real part is kept, SpectralProcessor.cpp:168-169).
#include "mi_common.h"
#include "fft_device.h"
#include "fft16.h"
#include "fft_wave.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

namespace
{
    // The magnitude's square root (pcomplex_mod, Analyzer.cpp:359): the hardware's v_sqrt_f32 (1 ulp) instead of the correctly
    // rounded sqrtf the compiler builds around it (a dozen instructions more per value: a quarter of the analysis kernels'
    // arithmetic) -- 6e-8 of the value against the 1e-5 the results are held to.
    __device__ __forceinline__ float mag_root(float x) { return __builtin_amdgcn_sqrtf(x); }
    // mix2(vAmp, |X|, 1 - tau, tau) (Analyzer.cpp:361): two products and their sum, each rounded -- the arithmetic of the reference's generic dsp::mix2, and
    // ONE form for every kernel that smooths (a contraction chosen per kernel by the compiler would make the bits of a run of
    // strobes depend on the launch that took them)
    __device__ __forceinline__ float mix2(float a, float m, float keep, float tau) { return __fadd_rn(__fmul_rn(a, keep), __fmul_rn(m, tau)); }

    using namespace mi_fft;
    // radix-16 core (fft16.h) for 1024 .. 8192-point transforms, radix-8 core (fft_device.h) below that
    template <int L_> using fplan = mi_fft16::fsel<L_>;

    // ---- hop transform of the spectral processor ----------------------------------------------------------
    // in_buf/out_buf: [channels][N] state of the reference object (pInBuf/pOutBuf); wnd: N window samples.
    // MODE 0: NONE, 1: MASK (fused), 2: forward half of the CALLBACK path (writes spec), 3: inverse half.
    template <int LOGH, int MODE>
    __global__ __launch_bounds__(fplan<LOGH>::T)
    void stft_hop_kernel(float *in_buf, float *out_buf, const float *__restrict__ wnd_in,
                         const float *__restrict__ wnd_out, const float *__restrict__ mask, size_t mask_stride,
                         float2 *spec, const uint8_t *__restrict__ active, const float2 *__restrict__ tw,
                         const float *io_src, size_t io_src_stride, float *io_dst, size_t io_dst_stride)
    {
        using PL = fplan<LOGH>;
        constexpr int H = PL::N, T = PL::T, N = 2 * H;
        __shared__ float2 lds_[fplan<LOGH>::LDS];
        float2 *const buf = lds_, *const scr = lds_ + fplan<LOGH>::SCR;
        const int ch = blockIdx.x, tid = threadIdx.x;
        typename fplan<LOGH>::real rf;
        if (MODE == 1 || MODE == 2)
        {
            rf.load(tw, TWN, tid);
            rf.prepare();
        }
        float2 *x2 = reinterpret_cast<float2 *>(in_buf + size_t(ch) * N);
        float2 *o2 = reinterpret_cast<float2 *>(out_buf + size_t(ch) * N);
        const float2 *wi = reinterpret_cast<const float2 *>(wnd_in);
        const float2 *wo = reinterpret_cast<const float2 *>(wnd_out);
        const bool on = (active == nullptr) || (active[ch] != 0);

        if (MODE == 2 && io_src != nullptr)
        {
            // MultiSpectralProcessor's timing with a whole frame in one call (process() would do this with two strided
            // copies before the hop): the frame the last overlap-add finished goes to the caller, the caller's frame
            // becomes the second half of the input buffer.  Pair-aligned rows.
            const float2 *s2 = reinterpret_cast<const float2 *>(io_src + size_t(ch) * io_src_stride);
            float2 *d2 = reinterpret_cast<float2 *>(io_dst + size_t(ch) * io_dst_stride);
            for (int m = tid; m < H / 2; m += T)
            {
                d2[m] = o2[m];
                x2[m + H / 2] = s2[m];
            }
            __syncthreads();                                // other threads of the workgroup read these cells below
        }
        if (MODE != 3)
        {
            // frame * input window, packed as z[m] = x[2m] + i x[2m+1]
            for (int m = tid; m < H; m += T)
            {
                float2 v = x2[m];
                if (wi != nullptr)
                {
                    const float2 w = wi[m];
                    v.x *= w.x;
                    v.y *= w.y;
                }
                buf[m] = v;
            }
            __syncthreads();
            if (MODE != 0 && on)
                rf.forward(buf, scr, tid);
        }
        if (MODE == 1 && on)
        {
            const float *mk = mask + size_t(ch) * mask_stride;          // H + 1 real gains
            for (int k = tid; k < H; k += T)
            {
                float2 v = buf[k];
                if (k == 0) { v.x *= mk[0]; v.y *= mk[H]; }
                else        { v.x *= mk[k]; v.y *= mk[k]; }
                buf[k] = v;
            }
            __syncthreads();
            rf.inverse(buf, scr, tid);
        }
        if (MODE == 2)
        {
            // full N-bin spectrum for the callback: X[k], k <= H from the image, the rest by symmetry
            float2 *sp = spec + size_t(ch) * N;
            if (on)
            {
                for (int k = tid; k < H; k += T)
                {
                    const float2 v = buf[k];
                    if (k == 0)
                    {
                        sp[0] = make_float2(v.x, 0.0f);
                        sp[H] = make_float2(v.y, 0.0f);
                    }
                    else
                    {
                        sp[k]     = v;
                        sp[N - k] = cconj(v);
                    }
                }
            }
            else        // unbound channel: the windowed frame itself travels on (MultiSpectralProcessor.cpp:346-350)
            {
                float *sr = reinterpret_cast<float *>(sp);
                for (int m = tid; m < H; m += T)
                {
                    sr[2 * m]     = buf[m].x;
                    sr[2 * m + 1] = buf[m].y;
                }
            }
            return;
        }
        if (MODE == 3)
            return;     // handled by stft_inverse_kernel

        // overlap-add: shift the output buffer by half a frame, add frame * output window
        // (a thread owns the pair m, m + H/2, so the old second half is read before it is overwritten)
        const float scale = (MODE == 1 && on) ? 1.0f / float(N) : 1.0f;
        for (int m = tid; m < H / 2; m += T)
        {
            const float2 y0 = buf[m], w0 = wo[m], y1 = buf[m + H / 2], w1 = wo[m + H / 2];
            const float2 prev = o2[m + H / 2];
            o2[m]         = make_float2(fmaf(y0.x * scale, w0.x, prev.x), fmaf(y0.y * scale, w0.y, prev.y));
            o2[m + H / 2] = make_float2(y1.x * scale * w1.x, y1.y * scale * w1.y);
        }
        // shift the input buffer by half a frame
        for (int m = tid; m < H / 2; m += T)
        {
            const float2 v = x2[m + H / 2];
            x2[m] = v;
        }
    }

    // A whole hop of a streaming call in ONE launch (operations NONE and MASK, SpectralProcessor's timing): the hop of
    // stft_hop_kernel, then what process() would do with two more launches -- the frame that the overlap-add has just
    // finished goes to the caller's `dst`, and the caller's next `frame` samples enter the input buffer behind the shift.
    // Every global operand is requested up front (twiddles, frame, window, the output buffer's second half, the new
    // samples, the gains); the second half of the frame stays in registers for the shift instead of being read again.
    // Pairs: a frame of N = 2H samples is H float2 pairs, a hop is H/2 pairs; thread t owns pairs t + i T, so pair m and
    // pair m + H/2 belong to the same thread (KPT = H / T is even: LOGH >= 7).
    // Runs of blocks (mi_spectral_bank_process_blocks): the hops of SEVERAL calls in one launch -- every block a buffer of its
    // own, `per` hops each, the addresses in the kernel arguments.
    constexpr int STFT_BLOCKS_MAX = 64;
    struct stft_blocks
    {
        int             per;            // hops per block
        const float    *src[STFT_BLOCKS_MAX];
        float          *dst[STFT_BLOCKS_MAX];
    };

    template <int LOGH, bool MASKED, bool TAB>
    __device__ __forceinline__
    void stft_stream_body(float *in_buf, float *out_buf, const float *__restrict__ wnd_in, const float *__restrict__ wnd_out,
                          const float *__restrict__ mask, size_t mask_stride, const float2 *__restrict__ tw,
                          const float *__restrict__ src, size_t src_stride, float *dst, size_t dst_stride, int hops,
                          const stft_blocks *tab)
    {
        // `hops` consecutive hops of the call in ONE launch (round 3): between two hops nothing goes through memory -- the second
        // half of the frame and the caller's samples that complete the next frame are in registers already, and so is the tail
        // the overlap-add leaves for the next hop.  Per channel and hop that is 8 KiB read + 8 KiB written at rank 12 (plus the
        // object's two state buffers once per CALL) instead of 40 KiB read + 48 KiB written per hop as one launch each.
        using PL = fplan<LOGH>;
        constexpr int H = PL::N, T = PL::T, N = 2 * H, KPT = H / T, HPT = KPT / 2;
        static_assert(KPT >= 2 && (KPT & 1) == 0 && KPT * T == H, "stft_stream_kernel needs an even number of pairs per thread");
        __shared__ float2 lds_[fplan<LOGH>::LDS];
        float2 *const buf = lds_, *const scr = lds_ + fplan<LOGH>::SCR;
        const int ch = blockIdx.x, tid = threadIdx.x;
        typename fplan<LOGH>::real rf;
        if (MASKED)
            rf.load(tw, TWN, tid);
        float2 *x2 = reinterpret_cast<float2 *>(in_buf + size_t(ch) * N);
        float2 *o2 = reinterpret_cast<float2 *>(out_buf + size_t(ch) * N);
        const float2 *wi = reinterpret_cast<const float2 *>(wnd_in);
        const float2 *wo = reinterpret_cast<const float2 *>(wnd_out);
        const float2 *s2 = TAB ? nullptr : reinterpret_cast<const float2 *>(src + size_t(ch) * src_stride);
        float2 *d2 = TAB ? nullptr : reinterpret_cast<float2 *>(dst + size_t(ch) * dst_stride);
        // the caller's samples of hop q / where hop q's finished frame goes (TAB: block q / per of the run, hop q % per inside it)
        auto src_hop = [&](int q) -> const float2 * {
            if (!TAB)
                return s2 + q * (H / 2);
            const int k = q / tab->per;
            return reinterpret_cast<const float2 *>(tab->src[k] + size_t(ch) * src_stride) + (q - k * tab->per) * (H / 2);
        };
        auto dst_hop = [&](int q) -> float2 * {
            if (!TAB)
                return d2 + q * (H / 2);
            const int k = q / tab->per;
            return reinterpret_cast<float2 *>(tab->dst[k] + size_t(ch) * dst_stride) + (q - k * tab->per) * (H / 2);
        };
        // live across the transforms: the half of the frame that the next hop starts with and the pending tail (the twiddles take
        // most of the rest of the 128 registers: a spill to scratch costs this kernel a factor of three); everything else is
        // read where it is used -- the frame's new half from the caller's block, windows and gains from L2
        float2 lo[HPT], hi[HPT], prev[HPT];
        #pragma unroll
        for (int i = 0; i < HPT; ++i)
        {
            lo[i]   = x2[tid + i * T];
            hi[i]   = x2[tid + i * T + H / 2];
            prev[i] = o2[tid + i * T + H / 2];
        }
        const float *mk = MASKED ? mask + size_t(ch) * mask_stride : nullptr;   // H + 1 real gains
        if (MASKED)
            rf.prepare();
        const float scale = MASKED ? 1.0f / float(N) : 1.0f;
        for (int h = 0; h < hops; ++h)
        {
            // The windows and the gains do not change from hop to hop, and the compiler knows: left alone it hoists their 40
            // loads per thread out of this loop, keeps them in registers across the transforms and spills the twiddles to
            // scratch (measured: 35 -> 91 us per step).  The pointers are laundered once per hop instead.
            asm volatile("" : "+s"(wi), "+s"(wo), "+s"(mk));
            // (a laundered pointer has lost its address space: read through it the windows and gains would be FLAT loads,
            // which count against lgkmcnt as well and so tie every wait for LDS data to them -- back to global they go)
            typedef const __attribute__((address_space(1))) v2f gv2f;
            typedef const __attribute__((address_space(1))) float gfloat;
            gv2f *const wig = reinterpret_cast<gv2f *>(reinterpret_cast<uint64_t>(wi));
            gv2f *const wog = reinterpret_cast<gv2f *>(reinterpret_cast<uint64_t>(wo));
            gfloat *const mkg = reinterpret_cast<gfloat *>(reinterpret_cast<uint64_t>(mk));
            // (the same goes for everything derived from the thread index: the swizzled LDS addresses of the eight passes)
            int tix = tid;
            asm volatile("" : "+v"(tix));
            if (h > 0)                                                      // frame h = [second half of frame h - 1 | caller's samples h - 1]
            {
                const float2 *sh = src_hop(h - 1);
                #pragma unroll
                for (int i = 0; i < HPT; ++i)
                    hi[i] = sh[tix + i * T];
            }
            // 512 .. 8192-point transforms take the windowed frame in registers and hand the result back in registers (fft_lds
            // REG_IN / REG_OUT: pair tix + i T is exactly what thread tix's first butterfly reads and its last one writes),
            // and the split into the half spectrum, the gains and the merge are ONE pass over LDS (real_fft::mask_pairs)
            constexpr bool REGS = MASKED && !fplan<LOGH>::radix16 && (mi_fft::plan<LOGH>::T == mi_fft::plan<LOGH>::TB);
            v2f io[KPT];
            #pragma unroll
            for (int i = 0; i < HPT; ++i)
            {
                const int m = tix + i * T;
                const v2f w0 = (wi != nullptr) ? wig[m] : v2f{1.0f, 1.0f};
                const v2f w1 = (wi != nullptr) ? wig[m + H / 2] : v2f{1.0f, 1.0f};
                io[i]       = v2f{lo[i].x * w0.x, lo[i].y * w0.y};
                io[i + HPT] = v2f{hi[i].x * w1.x, hi[i].y * w1.y};
                if (!REGS)
                {
                    buf[m]         = make_float2(io[i].x, io[i].y);
                    buf[m + H / 2] = make_float2(io[i + HPT].x, io[i + HPT].y);
                }
                lo[i] = hi[i];                                              // the frame moves on by half
            }
            if (!REGS)
                __syncthreads();
            if constexpr (REGS)
            {
                mi_fft::fft_lds<LOGH, false, true, false>(buf, scr, rf.ft, tix, io);
                rf.mask_pairs(buf, mkg, tix);
                mi_fft::fft_lds<LOGH, true, false, true>(buf, scr, rf.ft, tix, io);
            }
            else if (MASKED)
            {
                rf.forward(buf, scr, tix);
                #pragma unroll
                for (int i = 0; i < KPT; ++i)
                {
                    const int k = tix + i * T;
                    const float g = mkg[k];
                    float2 v = buf[k];
                    if (k == 0) { v.x *= g; v.y *= mkg[H]; }
                    else        { v.x *= g; v.y *= g; }
                    buf[k] = v;
                }
                __syncthreads();
                rf.inverse(buf, scr, tix);
            }
            if (!REGS)
            {
                #pragma unroll
                for (int i = 0; i < HPT; ++i)
                {
                    const float2 a = buf[tix + i * T], c = buf[tix + i * T + H / 2];
                    io[i] = v2f{a.x, a.y};
                    io[i + HPT] = v2f{c.x, c.y};
                }
            }
            const bool last = (h + 1 == hops);
            float2 *const dh = dst_hop(h);
            const float2 *const sl = last ? src_hop(h) : nullptr;
            #pragma unroll
            for (int i = 0; i < HPT; ++i)
            {
                const int m = tix + i * T;
                const v2f w0 = wog[m], w1 = wog[m + H / 2];
                const v2f y0 = io[i], y1 = io[i + HPT];
                const float2 done = make_float2(fmaf(y0.x * scale, w0.x, prev[i].x), fmaf(y0.y * scale, w0.y, prev[i].y));
                prev[i] = make_float2(y1.x * scale * w1.x, y1.y * scale * w1.y);     // the tail the next hop adds to
                dh[m] = done;                                               // the finished frame, straight to the caller
                if (last)                                                   // the object's state as the call leaves it
                {
                    o2[m]         = done;
                    o2[m + H / 2] = prev[i];
                    x2[m]         = lo[i];
                    x2[m + H / 2] = sl[m];
                }
            }
            if (!last)
                __syncthreads();                                            // buf is refilled by the next hop
        }
    }

    template <int LOGH, bool MASKED>
    // (radix-8 core: <= 128 VGPRs, four waves per SIMD; radix-16 core: sixteen points and two passes of twiddles per thread,
    //  two waves per SIMD -- a bank of 1024 channels is 2048 waves, two per SIMD, either way)
    __global__ __launch_bounds__(fplan<LOGH>::T, fplan<LOGH>::radix16 ? 2 : (fplan<LOGH>::T <= 256) ? 4 : (fplan<LOGH>::T <= 512) ? 2 : 1)
    void stft_stream_kernel(float *in_buf, float *out_buf, const float *__restrict__ wnd_in, const float *__restrict__ wnd_out,
                            const float *__restrict__ mask, size_t mask_stride, const float2 *__restrict__ tw,
                            const float *__restrict__ src, size_t src_stride, float *dst, size_t dst_stride, int hops)
    {
        stft_stream_body<LOGH, MASKED, false>(in_buf, out_buf, wnd_in, wnd_out, mask, mask_stride, tw, src, src_stride, dst, dst_stride, hops, nullptr);
    }

    template <int LOGH, bool MASKED>
    __global__ __launch_bounds__(fplan<LOGH>::T, fplan<LOGH>::radix16 ? 2 : (fplan<LOGH>::T <= 256) ? 4 : (fplan<LOGH>::T <= 512) ? 2 : 1)
    void stft_stream_blocks_kernel(float *in_buf, float *out_buf, const float *__restrict__ wnd_in, const float *__restrict__ wnd_out,
                                   const float *__restrict__ mask, size_t mask_stride, const float2 *__restrict__ tw,
                                   const stft_blocks tab, size_t src_stride, size_t dst_stride, int hops)
    {
        stft_stream_body<LOGH, MASKED, true>(in_buf, out_buf, wnd_in, wnd_out, mask, mask_stride, tw, nullptr, src_stride, nullptr, dst_stride, hops, &tab);
    }

    // ---- runs of 4096-sample blocks at rank 12 with a fused mask, on the wave-resident transform (fft_wave.h) -------------------------
    // stft_stream_blocks_kernel runs at the rate of its transforms through LDS (0.24 of the HBM roofline, 0.40 of the issue rate).
    // Here a WAVE owns a channel's consecutive blocks and takes the two frames of a block -- frame 2u = block u - 1, frame 2u + 1
    // = [second half of block u - 1 | first half of block u] -- as ONE complex sequence z = w (A + i B): a real, symmetric gain
    // acts on the spectrum of A and on the spectrum of B alike, so G Z is the spectrum of the two shaped frames: one forward and one
    // inverse 4096-point complex transform per BLOCK (two real frames), no split, no merge, no barrier.  The inverse's real part is
    // frame A's result, its imaginary part frame B's; the output window and the overlap-add are the lane's own registers, and the
    // tail the next block adds to stays in 32 registers of the wave.  The shape of splitter_wave_blocks_kernel (splitter.hip) with one
    // band and a window in front: four waves per workgroup, one per SIMD; the windows and the gains sit in LDS the way a lane reads them
    // (16-byte reads).  A first version with two waves per SIMD, 4-byte reads of half tables and a hundred spilled registers measured
    // 21.8 - 26 us per block against the workgroup kernel's 19.4 (profiles/r05_experiments/stft_wave_blocks.txt).
    // A channel's run is cut into 1, 2 or 4 segments, the waves of one workgroup; a segment that does not start the run first redoes
    // the block in front of it (without storing) for the tail it starts from.
    // The same sums through another transform: within 1e-6 of stft_stream_blocks_kernel, not its bits.  The host sends a run this way
    // only if no buffer of the run overlaps another (the segments run side by side).
    // SHARED: one row of gains for every channel -- in LDS; otherwise a row per channel, which the wave asks for (64 loads of its
    // own channel's row, the gain of bin k at min(k, N - k)) in front of the forward transform and finds in registers behind it.
    constexpr int STFT_WAVES = 4;
    template <bool SHARED>
    __global__ __launch_bounds__(64 * STFT_WAVES, 1)
    void stft_wave_blocks_kernel(float *in_buf, float *out_buf, const float *__restrict__ wnd_in /* or NULL: none */,
                                 const float *__restrict__ wnd_out, const float *__restrict__ mask /* rows of N / 2 + 1 gains */,
                                 size_t mask_stride /* 0 (SHARED) or the rows' pitch */,
                                 const float2 *__restrict__ tw, const stft_blocks tab, size_t src_stride, size_t dst_stride,
                                 int blocks, int channels, int segs)
    {
        using namespace mi_fftw;
        constexpr int HALF = R / 2, HOP = N / 2;            // registers of half a frame; samples of a hop
        __shared__ float areas[STFT_WAVES][AREA];
        __shared__ float2 pl[16 * R];
        // entry lane + 64 r of a table at float4 cell [r / 4][lane], component r % 4
        __shared__ float4 win_l[N / 4], wout_l[N / 4], gain_l[SHARED ? N / 4 : 1];
        const int tid = threadIdx.x, lane = tid & 63;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        fill_table_pq(pl, tw, tid, 64 * STFT_WAVES);
        for (int i = tid; i < N; i += 64 * STFT_WAVES)
        {
            const int l = i & 63, r = i >> 6, cell = ((r >> 2) * 64 + l) * 4 + (r & 3);
            reinterpret_cast<float *>(win_l)[cell] = (wnd_in != nullptr) ? wnd_in[i] : 1.0f;
            reinterpret_cast<float *>(wout_l)[cell] = wnd_out[i] * (1.0f / float(N));    // (the transform pair's 1 / N rides on the window)
            if (SHARED)
                reinterpret_cast<float *>(gain_l)[cell] = mask[(i <= HOP) ? i : N - i];   // the N / 2 + 1 gains act on k and N - k alike
        }
        __syncthreads();
        const int gid = blockIdx.x * STFT_WAVES + wv;       // wave of the launch: (channel, segment); segs divides STFT_WAVES
        const bool idle = gid >= channels * segs;
        const int ch = idle ? 0 : gid / segs, seg = gid - ch * segs;
        const int per = (blocks + segs - 1) / segs, u0 = seg * per, u1 = idle ? u0 : ((u0 + per < blocks) ? u0 + per : blocks);
        float *const xs = in_buf + size_t(ch) * N, *const os = out_buf + size_t(ch) * N;
        auto at = [](__amdgpu_buffer_rsrc_t r, int lane_off, int row_off) -> float {
            return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, lane_off, row_off, 0));
        };
        // block u of the run as the caller gave it (u = -1: the N samples the object holds)
        auto block_in = [&](int u) -> __amdgpu_buffer_rsrc_t {
            return mi::wt_buffer((u < 0) ? xs : const_cast<float *>(tab.src[u]) + size_t(ch) * src_stride, unsigned(N * sizeof(float)));
        };
        auto times = [&](v2f (&x)[R], const float4 *t) {
            #pragma unroll
            for (int r4 = 0; r4 < R / 4; ++r4)
            {
                const float4 g = t[r4 * 64 + lane];
                x[4 * r4 + 0] = x[4 * r4 + 0] * v2f{g.x, g.x};
                x[4 * r4 + 1] = x[4 * r4 + 1] * v2f{g.y, g.y};
                x[4 * r4 + 2] = x[4 * r4 + 2] * v2f{g.z, g.z};
                x[4 * r4 + 3] = x[4 * r4 + 3] * v2f{g.w, g.w};
            }
        };
        float prev[HALF];                                   // the tail the next frame adds to: sample lane + 64 j
        {
            const __amdgpu_buffer_rsrc_t ro = mi::wt_buffer(os, (u0 == 0 && u0 < u1) ? unsigned(N * sizeof(float)) : 0u);
            #pragma unroll
            for (int j = 0; j < HALF; ++j)
                prev[j] = at(ro, lane * 4, (HOP + 64 * j) * 4);         // (0 where the segment starts inside the run: out of range)
        }
        // half blocks in hand: h0, h1 = block u - 1, h2 = first half of block u (sample lane + 64 j each); the two the next block
        // needs are asked for before this block's transforms and arrive underneath them (one wave per SIMD: nothing else hides them;
        // the wave has the registers for it)
        const int ustart = (u0 == 0) ? 0 : u0 - 1;
        float h0[HALF], h1[HALF], h2[HALF], n1[HALF], n2[HALF];
        if (u0 < u1)
        {
            const __amdgpu_buffer_rsrc_t ra = block_in(ustart - 1), rb = block_in(ustart);
            #pragma unroll
            for (int j = 0; j < HALF; ++j)
            {
                h0[j] = at(ra, lane * 4, 256 * j);
                h1[j] = at(ra, lane * 4, 256 * (j + HALF));
                h2[j] = at(rb, lane * 4, 256 * j);
            }
        }
        for (int u = ustart; u < u1 && u0 < u1; ++u)
        {
            // z[n] = w[n] (A[n] + i B[n]), n = lane + 64 j: A = block u - 1 = [h0 | h1], B = [h1 | h2]
            v2f x[R];
            #pragma unroll
            for (int j = 0; j < HALF; ++j)
            {
                x[j] = v2f{h0[j], h1[j]};
                x[j + HALF] = v2f{h1[j], h2[j]};
                h0[j] = h2[j];
            }
            {
                const bool more = u + 1 < u1;
                const __amdgpu_buffer_rsrc_t ra = mi::wt_buffer(const_cast<float *>(tab.src[more ? u : 0]) + size_t(ch) * src_stride, more ? unsigned(N * sizeof(float)) : 0u);
                const __amdgpu_buffer_rsrc_t rb = mi::wt_buffer(const_cast<float *>(tab.src[more ? u + 1 : 0]) + size_t(ch) * src_stride, more ? unsigned(N * sizeof(float)) : 0u);
                #pragma unroll
                for (int j = 0; j < HALF; ++j)
                {
                    n1[j] = at(ra, lane * 4, 256 * (j + HALF));
                    n2[j] = at(rb, lane * 4, 256 * j);
                }
            }
            times(x, win_l);
            if constexpr (SHARED)
            {
                fft4096_t<false>(x, pl, areas[wv], lane);
                times(x, gain_l);
            }
            else
            {
                const __amdgpu_buffer_rsrc_t rmk = mi::wt_buffer(const_cast<float *>(mask) + size_t(ch) * mask_stride, unsigned((HOP + 1) * sizeof(float)));
                float g[R];
                #pragma unroll
                for (int r = 0; r < R; ++r)                 // bin lane + 64 r: row entry lane + 64 r, or 64 (64 - r) - lane behind N / 2
                    g[r] = (r < HALF) ? at(rmk, lane * 4, 256 * r) : at(rmk, (64 - lane) * 4, 256 * (R - 1 - r));
                fft4096_t<false>(x, pl, areas[wv], lane);
                #pragma unroll
                for (int r = 0; r < R; ++r)
                    x[r] = x[r] * v2f{g[r], g[r]};
            }
            fft4096_t<true>(x, pl, areas[wv], lane);
            times(x, wout_l);
            const bool store = u >= u0;                      // (the block in front of the segment: only its tail is wanted)
            const __amdgpu_buffer_rsrc_t rout = mi::wt_buffer(store ? tab.dst[u] + size_t(ch) * dst_stride : nullptr,
                                                              store ? unsigned(N * sizeof(float)) : 0u);
            #pragma unroll
            for (int j = 0; j < HALF; ++j)
            {
                const float done_a = x[j].x + prev[j];
                const float done_b = x[j].y + x[j + HALF].x;
                prev[j] = x[j + HALF].y;
                mi::wt_store(rout, lane * 4 + 256 * j, done_a);                 // (dropped by the bounds check where nothing is stored)
                mi::wt_store(rout, lane * 4 + 256 * (j + HALF), done_b);
                h1[j] = n1[j];
                h2[j] = n2[j];
            }
        }
        // The object's state as the call leaves it -- behind a barrier: the wave of the channel's first segment has read the state
        // the call found.  Output buffer: [the frame finished last | the tail]; input buffer: the last block.
        __syncthreads();
        if (u1 == blocks && u0 < u1)
        {
            const __amdgpu_buffer_rsrc_t rl = block_in(blocks - 1), rx = mi::wt_buffer(xs, unsigned(N * sizeof(float)));
            const __amdgpu_buffer_rsrc_t ro = mi::wt_buffer(os, unsigned(N * sizeof(float)));
            const __amdgpu_buffer_rsrc_t rd = mi::wt_buffer(tab.dst[blocks - 1] + size_t(ch) * dst_stride, unsigned(N * sizeof(float)));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (this wave's own stores of the last block)
            // (all the loads, then all the stores: alternating, every store waited for its load's round trip -- the compiler keeps
            // them in program order -- and the 96 of them were 45 us of every launch, three quarters of a one-block call)
            float t[R], d[HALF];
            #pragma unroll
            for (int j = 0; j < R; ++j)
                t[j] = at(rl, lane * 4, 256 * j);
            #pragma unroll
            for (int j = 0; j < HALF; ++j)
                d[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, lane * 4, 256 * (j + HALF), mi::CPOL_SC1));
            #pragma unroll
            for (int j = 0; j < R; ++j)
                mi::wt_store(rx, lane * 4 + 256 * j, t[j]);
            #pragma unroll
            for (int j = 0; j < HALF; ++j)
            {
                mi::wt_store(ro, lane * 4 + 256 * j, d[j]);
                mi::wt_store(ro, lane * 4 + 256 * (j + HALF), prev[j]);
            }
        }
    }

    // CALLBACK path, second half.  The function may have broken the Hermitian symmetry of the spectrum and only the real
    // part of the inverse is kept (SpectralProcessor.cpp:168-169): Re ifft(S) = ifft of S's Hermitian part
    // (S[k] + conj S[N-k]) / 2, so the way back is the same half-size real transform as the way there.
    // Then window, overlap-add, input shift.
    template <int LOGH>
    __global__ __launch_bounds__(fplan<LOGH>::T)
    void stft_inverse_kernel(float *in_buf, float *out_buf, const float *__restrict__ wnd_out,
                             const float2 *__restrict__ spec, const uint8_t *__restrict__ active,
                             const uint8_t *__restrict__ has_out, const float2 *__restrict__ tw)
    {
        using PL = fplan<LOGH>;
        constexpr int H = PL::N, T = PL::T, N = 2 * H;
        __shared__ float2 lds_[fplan<LOGH>::LDS];
        float2 *const buf = lds_, *const scr = lds_ + fplan<LOGH>::SCR;
        const int ch = blockIdx.x, tid = threadIdx.x;
        const bool on = ((active == nullptr) || (active[ch] != 0)) && ((has_out == nullptr) || (has_out[ch] != 0));
        const float2 *sp = spec + size_t(ch) * N;
        float2 *o2 = reinterpret_cast<float2 *>(out_buf + size_t(ch) * N);
        float2 *x2 = reinterpret_cast<float2 *>(in_buf + size_t(ch) * N);
        const float2 *wo = reinterpret_cast<const float2 *>(wnd_out);
        if (on)
        {
            typename fplan<LOGH>::real rf;
            rf.load(tw, TWN, tid);
            rf.prepare();
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
        }
        const float scale = 1.0f / float(N);
        for (int m = tid; m < H / 2; m += T)
        {
            // bound channels: the inverse; others: the windowed frame the forward kernel left in the spectrum buffer
            float2 y0, y1;
            if (on)
            {
                y0 = make_float2(buf[m].x * scale, buf[m].y * scale);
                y1 = make_float2(buf[m + H / 2].x * scale, buf[m + H / 2].y * scale);
            }
            else
            {
                y0 = sp[m];
                y1 = sp[m + H / 2];
            }
            const float2 w0 = wo[m], w1 = wo[m + H / 2], prev = o2[m + H / 2];
            o2[m]         = make_float2(fmaf(y0.x, w0.x, prev.x), fmaf(y0.y, w0.y, prev.y));
            o2[m + H / 2] = make_float2(y1.x * w1.x, y1.y * w1.y);
        }
        for (int m = tid; m < H / 2; m += T)
        {
            const float2 v = x2[m + H / 2];
            x2[m] = v;
        }
    }

    // ---- frames above 2^14 samples (ranks 15 .. BIG_MAX_RANK) ------------------------------------------------------
    // Such a frame does not fit the LDS of a workgroup.  Its transform goes through global memory as a four-step complex
    // transform N = N1 x N2: N2 = N / 8192 interleaved sub-sequences of N1 = 8192 points are transformed in LDS with the
    // core above and multiplied by W_N^(n2 k1); N1 columns of N2 <= 32 points are then summed directly.  With
    // n = n1 N2 + n2 and k = k1 + N1 k2:  X[k] = sum_n2 W_N2^(n2 k2) [ W_N^(n2 k1) sum_n1 x[n1 N2 + n2] W_N1^(n1 k1) ].
    // The frame is treated as complex with a zero imaginary part and only the real part of the way back is kept, which is
    // what the reference does (pcomplex_r2c ... pcomplex_c2r, SpectralProcessor.cpp:164-169).  These frames are rare and
    // long: the path is several plain launches per hop, not a fused kernel.
    constexpr int BIG_LOG1 = 13, BIG_N1 = 1 << BIG_LOG1, BIG_MAX_RANK = 18;

    // work[n] = (in[n] * w[n], 0)
    __global__ __launch_bounds__(256)
    void big_window_kernel(float2 *work, const float *__restrict__ in_buf, const float *__restrict__ wnd, uint32_t N)
    {
        const uint32_t n = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
        if (n < N)
        {
            const float v = in_buf[size_t(ch) * N + n];
            work[size_t(ch) * N + n] = make_float2((wnd != nullptr) ? v * wnd[n] : v, 0.0f);
        }
    }

    template <bool INVERSE>
    __global__ __launch_bounds__(plan<BIG_LOG1>::T)
    void big_rows_kernel(float2 *dst /* [ch][N2][N1] */, const float2 *__restrict__ src /* [ch][N] */, uint32_t N2, uint32_t N,
                         const float2 *__restrict__ tw)
    {
        constexpr int T = plan<BIG_LOG1>::T;
        __shared__ float2 buf[plan<BIG_LOG1>::BUF], scr[plan<BIG_LOG1>::BUF];
        const uint32_t n2 = blockIdx.x, ch = blockIdx.y;
        const int tid = threadIdx.x;
        fft_tw<BIG_LOG1> ft;
        load_fft_tw<BIG_LOG1>(ft, tw, TWN / BIG_N1, tid);
        finish_fft_tw<BIG_LOG1>(ft);
        const float2 *x = src + size_t(ch) * N;
        for (int n1 = tid; n1 < BIG_N1; n1 += T)
            buf[n1] = x[size_t(n1) * N2 + n2];
        __syncthreads();
        fft_lds<BIG_LOG1, INVERSE>(buf, scr, ft, tid);
        float2 *y = dst + size_t(ch) * N + size_t(n2) * BIG_N1;
        const float unit = (INVERSE ? 2.0f : -2.0f) / float(N);     // n2 k1 < N <= 2^18: exact in float32
        for (int k1 = tid; k1 < BIG_N1; k1 += T)
        {
            float sn, cs;
            sincospif(unit * float(n2 * uint32_t(k1)), &sn, &cs);
            y[k1] = cmul(buf[k1], make_float2(cs, sn));
        }
    }

    template <int LOG2, bool INVERSE>
    __global__ __launch_bounds__(256)
    void big_cols_kernel(float2 *dst /* [ch][N] natural order */, const float2 *__restrict__ src /* [ch][N2][N1] */, uint32_t N)
    {
        constexpr int N2 = 1 << LOG2;
        const uint32_t k1 = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
        float2 y[N2];
        #pragma unroll
        for (int n2 = 0; n2 < N2; ++n2)
            y[n2] = src[size_t(ch) * N + size_t(n2) * BIG_N1 + k1];
        #pragma unroll
        for (int k2 = 0; k2 < N2; ++k2)
        {
            float2 acc = y[0];
            #pragma unroll
            for (int n2 = 1; n2 < N2; ++n2)
            {
                const int j = (n2 * k2) & (N2 - 1);                    // W_N2^(n2 k2): folded at compile time
                const double a = (INVERSE ? 2.0 : -2.0) * 3.14159265358979323846 * double(j) / double(N2);
                const float2 w = make_float2(float(__builtin_cos(a)), float(__builtin_sin(a)));
                acc = cadd(acc, cmul(y[n2], w));
            }
            dst[size_t(ch) * N + k1 + size_t(BIG_N1) * k2] = acc;
        }
    }

    // S[k] *= mask[min(k, N - k)]  (the N/2 + 1 real gains of bind_mask act on k and N - k alike)
    __global__ __launch_bounds__(256)
    void big_mask_kernel(float2 *spec, const float *__restrict__ mask, size_t mask_stride, uint32_t N)
    {
        const uint32_t k = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
        if (k < N)
        {
            const float g = mask[size_t(ch) * mask_stride + ((k <= N / 2) ? k : N - k)];
            float2 &v = spec[size_t(ch) * N + k];
            v = make_float2(v.x * g, v.y * g);
        }
    }

    // channels without a bound input: their windowed frame travels on in the spectrum buffer (MultiSpectralProcessor.cpp:346-350)
    __global__ __launch_bounds__(256)
    void big_unbound_kernel(float2 *spec, const float2 *__restrict__ work, const uint8_t *__restrict__ active, uint32_t N)
    {
        const uint32_t n = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
        if (n < N && active[ch] == 0)
            reinterpret_cast<float *>(spec + size_t(ch) * N)[n] = work[size_t(ch) * N + n].x;
    }

    // overlap-add with the output window (out shifted by half a frame), then the input buffer moves on by half a frame
    __global__ __launch_bounds__(256)
    void big_ola_kernel(float *in_buf, float *out_buf, const float2 *__restrict__ work, const float2 *__restrict__ spec,
                        const float *__restrict__ wnd_out, float scale, uint32_t N, const uint8_t *__restrict__ active,
                        const uint8_t *__restrict__ has_out)
    {
        const uint32_t m = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y, half = N >> 1;
        if (m >= half)
            return;
        const bool on = (spec == nullptr) || (((active == nullptr) || active[ch] != 0) && ((has_out == nullptr) || has_out[ch] != 0));
        float y0, y1;
        if (on)
        {
            y0 = work[size_t(ch) * N + m].x * scale;
            y1 = work[size_t(ch) * N + m + half].x * scale;
        }
        else
        {
            const float *f = reinterpret_cast<const float *>(spec + size_t(ch) * N);
            y0 = f[m];
            y1 = f[m + half];
        }
        float *o = out_buf + size_t(ch) * N, *x = in_buf + size_t(ch) * N;
        const float prev = o[m + half];
        o[m]        = fmaf(y0, wnd_out[m], prev);
        o[m + half] = y1 * wnd_out[m + half];
        x[m] = x[m + half];
    }

    // process(src, count) of the reference -- analysis only (SpectralProcessor.cpp:201-249): no inverse transform and
    // nothing is added to the output buffer; it is shifted by half a frame and its tail zeroed, the input buffer shifted.
    __global__ __launch_bounds__(256)
    void stft_shift_kernel(float *in_buf, float *out_buf, uint32_t frame)
    {
        float *ib = in_buf + size_t(blockIdx.y) * 2 * frame, *ob = out_buf + size_t(blockIdx.y) * 2 * frame;
        const uint32_t i = blockIdx.x * 256 + threadIdx.x;
        if (i >= frame)
            return;
        ob[i] = ob[i + frame];          // every cell is read and written by the same thread or by exactly one other
        ob[i + frame] = 0.0f;           // one, in this order: thread i reads i + frame before it writes it
        ib[i] = ib[i + frame];
    }

#ifdef MI_AN_PROBE
    // phase timestamps of thread 0 of every workgroup (tests/experiments/analyzer_probe.hip)
    __device__ unsigned long long g_an_probe[4096 * 8];
    #define MI_APROBE(slot) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) \
        g_an_probe[blockIdx.x * 8 + (slot)] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
    // ... and of lane 0 of every wave of analyzer_frames_wave_kernel (tests/experiments/analyzer_wave_probe.hip)
    __device__ unsigned long long g_anw_probe[256 * 8 * 32];
    #define MI_WPROBE(slot) do { __builtin_amdgcn_sched_barrier(0); if ((threadIdx.x & 63) == 0 && blockIdx.x < 256 && (slot) < 32) \
        g_anw_probe[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 32 + (slot)] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
    #define MI_APROBE(slot) do { } while (0)
    #define MI_WPROBE(slot) do { } while (0)
#endif

    // Per-bin reduction over channels (the C5 callback), in one launch and in an order that does not depend on the launch
    // geometry AND composes across channel shards: blocks of REDUCE_BLOCK consecutive channels are summed in channel
    // order, the block sums are then added along a binary tree aligned to powers of two (element j takes in element
    // j + s for s = 1, 2, 4, ...).  The sum over channels [0, 2^k * REDUCE_BLOCK) is a node of that tree, so the
    // reductions of two banks that hold the halves of a channel set add up to the reduction of the whole set bit for
    // bit -- which is what lets the per-bin sum be sharded over GPUs and all-reduced (tests/test_spectral_gpu.py).
    // A workgroup owns 16 bins (one 64-byte segment of every channel row): a wave reads that segment of four channels at
    // once, WAVES x 4 groups of 16 lanes work on as many blocks at a time with the sixteen loads of a block in flight.
    // Narrow workgroups instead of 64-bin ones because 2049 bins would otherwise be 33 workgroups on 256 CUs.
    // The body serves two callers: bin_reduce_kernel (its own launch, 16 waves) and the reduce role that rides on the
    // analysis launch (analyzer_kernel, round 3: as many waves as the analysis workgroups have; DEVICE: the rows were
    // written by other workgroups of the SAME launch, so they are read with device-scope loads past this CU's L1).
    constexpr uint32_t REDUCE_BINS = 16, REDUCE_WAVES = 16, REDUCE_BLOCK = 16, REDUCE_MAX_BLOCKS = 1024;

    template <uint32_t WAVES, bool DEVICE, uint32_t REDUCE_BINS = 16>
    __device__ __forceinline__
    void bin_reduce_body(float *out, const float *src, uint32_t stride, uint32_t channels, uint32_t bins,
                         const float *__restrict__ env, uint32_t block /* channels per block, multiple of 16 */,
                         float (*part)[REDUCE_BINS], uint32_t group /* which REDUCE_BINS bins */)
    {
        const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const uint32_t b = lane & (REDUCE_BINS - 1), r = lane / REDUCE_BINS;
        const uint32_t k = group * REDUCE_BINS + b;
        constexpr uint32_t GROUPS = WAVES * (64 / REDUCE_BINS);                 // blocks in flight
        const uint32_t nblocks = (channels + block - 1) / block;
        // DEVICE: sc1 loads (past this CU's L1, coherent at device scope) through a buffer descriptor -- the builtin, not an
        // atomic load: the compiler keeps sixteen of them in flight, sixteen relaxed atomic loads it waits for one by one
        // (measured: 45 us instead of 3 for the 8.4 MB of C5)
        const __amdgpu_buffer_rsrc_t rsrc = mi::wt_buffer(const_cast<float *>(src), unsigned(size_t(channels) * stride * sizeof(float)));
        auto row = [&](uint32_t c) -> float {
            if (DEVICE)
                return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, int((size_t(c) * stride + k) * sizeof(float)), 0, mi::CPOL_SC1));
            return src[size_t(c) * stride + k];
        };
        for (uint32_t j = w * (64 / REDUCE_BINS) + r; j < nblocks; j += GROUPS)
        {
            float s = 0.0f;
            if (k < bins)
            {
                const uint32_t c0 = j * block, c1 = (c0 + block < channels) ? c0 + block : channels;
                uint32_t c = c0;
                for (; c + 16 <= c1; c += 16)
                {
                    float v[16];
                    #pragma unroll
                    for (int i = 0; i < 16; ++i)
                        v[i] = row(c + i);
                    #pragma unroll
                    for (int i = 0; i < 16; ++i)
                        s += v[i];
                }
                for (; c < c1; ++c)
                    s += row(c);
            }
            part[j][b] = s;
        }
        __syncthreads();
        if (nblocks <= 64)
        {
            // Up to 64 block sums per bin (1024 channels): the same binary tree -- element j, a multiple of 2 s, takes in element
            // j + s -- inside ONE wave per bin by xor shuffles (lane j holds block j; a block that is not there enters as + 0,
            // which changes no sum of non-negative terms), instead of six rounds over LDS with a barrier behind each.
            for (uint32_t bb = w; bb < REDUCE_BINS; bb += WAVES)
            {
                float v = (lane < nblocks) ? part[lane][bb] : 0.0f;
                #pragma unroll
                for (int s = 1; s < 64; s <<= 1)
                    v += __shfl_xor(v, s);
                const uint32_t kk = group * REDUCE_BINS + bb;
                if (lane == 0 && kk < bins)
                    out[kk] = (env != nullptr) ? v * env[kk] : v;
            }
            return;
        }
        for (uint32_t s = 1; s < nblocks; s <<= 1)
        {
            // element j (a multiple of 2s) takes in element j + s
            const uint32_t pairs = (nblocks + 2 * s - 1) / (2 * s);
            for (uint32_t i = threadIdx.x; i < pairs * REDUCE_BINS; i += 64 * WAVES)
            {
                const uint32_t j = (i / REDUCE_BINS) * 2 * s, bb = i & (REDUCE_BINS - 1);
                if (j + s < nblocks)
                    part[j][bb] += part[j + s][bb];
            }
            __syncthreads();
        }
        if (threadIdx.x < REDUCE_BINS && k < bins)
        {
            const float t = part[0][b];
            out[k] = (env != nullptr) ? t * env[k] : t;
        }
    }

    __global__ __launch_bounds__(64 * REDUCE_WAVES)
    void bin_reduce_kernel(float *out, const float *__restrict__ src, uint32_t stride, uint32_t channels, uint32_t bins,
                           const float *__restrict__ env, uint32_t block /* channels per block, multiple of 16 */)
    {
        __shared__ float part[REDUCE_MAX_BLOCKS][REDUCE_BINS];
        bin_reduce_body<REDUCE_WAVES, false>(out, src, stride, channels, bins, env, block, part, blockIdx.x);
    }

    // The reductions of several analyses in ONE launch: blockIdx.y picks the frame, whose rows stand in a plane of their own
    // (mi_analyzer_bank_process_reduce_frames).  A reduction alone is 129 workgroups of one per CU on a 256-CU part and mostly
    // latency; eight frames' worth fill the chip.  Same body, same order of summation, same bits per frame.
    // With several frames to fill the chip a workgroup owns BINS = 32 bins, a whole 128-byte line of every channel row (16 bins
    // are half a line: the other half went to the neighbouring workgroup, i.e. to another XCD's L2 -- every line fetched twice);
    // the 16-bin form serves banks of more than BINS_32_BLOCKS blocks (its LDS holds twice the block sums).
    constexpr uint32_t REDUCE_FRAMES_MAX = 16, BINS_32_BLOCKS = 512;
    struct reduce_planes { const float *rows[REDUCE_FRAMES_MAX]; };
    template <uint32_t BINS>
    __global__ __launch_bounds__(64 * REDUCE_WAVES)
    void bin_reduce_frames_kernel(float *out, size_t out_stride, const reduce_planes planes, uint32_t stride, uint32_t channels,
                                  uint32_t bins, const float *__restrict__ env, uint32_t block)
    {
        __shared__ float part[REDUCE_MAX_BLOCKS * 16 / BINS][BINS];
        bin_reduce_body<REDUCE_WAVES, false, BINS>(out + size_t(blockIdx.y) * out_stride, planes.rows[blockIdx.y], stride, channels, bins,
                                                   env, block, part, blockIdx.x);
    }

    // ---- analyzer -------------------------------------------------------------------------------------------
    // ring: [channels][buf_size]; the frame of channel c ends `delay[c]` samples before `head`.
    // amp_old: vAmp as of the strobe (what get_spectrum() shows until the next strobe: the reference copies vAmp to vData
    // at the strobe, Analyzer.cpp:321-326 -- here the two buffers swap roles instead); amp_new: vAmp after this analysis.
    // ingest: `ingest_n` new samples per channel go into the ring behind `head` in the same launch (the "fill the
    // buffer" half of Analyzer::process, Analyzer.cpp:371-398; they lie outside every analysis window of this strobe).
    template <int LOGH>
    __device__ __forceinline__
    void analyzer_role(float2 *lds_, float *ring, uint32_t buf_size, uint32_t head,
                       const uint32_t *__restrict__ delay, const uint8_t *__restrict__ flags,
                       const float *__restrict__ wnd, const float *__restrict__ amp_old, float *amp_new,
                       uint32_t amp_stride, float tau, const float2 *__restrict__ tw,
                       const float *ingest, size_t ingest_stride, uint32_t ingest_n, int ingest_zero)
    {
        using PL = fplan<LOGH>;
        constexpr int H = PL::N, T = PL::T, N = 2 * H;
        float2 *const buf = lds_, *const scr = lds_ + fplan<LOGH>::SCR;
        const int ch = blockIdx.x, tid = threadIdx.x;
        MI_APROBE(0);
        // Everything the launch needs is requested before anything is waited for (in-kernel timeline,
        // tests/experiments/analyzer_probe.hip: the two dependent little loads -- the channel's flags, then its delay for the
        // frame's address -- used to cost two exposed latencies before the frame was even asked for, and the new samples were
        // only asked for when the spectrum had been stored): twiddles, flags AND delay, then the frame, the spectrum being
        // smoothed and the samples to ingest.
        typename fplan<LOGH>::real rf;
        rf.load(tw, TWN, tid);
        const uint8_t fl = flags[ch];                       // bit0: active, bit1: frozen
        const uint32_t dly = delay[ch];
        const float *a = amp_old + size_t(ch) * amp_stride;
        float *an = amp_new + size_t(ch) * amp_stride;
        float *rbw = ring + size_t(ch) * buf_size;
        // the new samples are requested now and stored at the very end
        constexpr int IPT = 4;                              // up to 4*T samples per pass
        if (ingest_n > 0 && ((fl & 2) || !(fl & 1)))        // channels that skip the analysis still take their samples
        {
            for (uint32_t i = tid; i < ingest_n; i += T)
            {
                uint32_t w = head + i;
                if (w >= buf_size) w -= buf_size;
                rbw[w] = ingest_zero ? 0.0f : ingest[size_t(ch) * ingest_stride + i];
            }
        }
        if (fl & 2)                                         // frozen: vAmp stays (Analyzer.cpp:334)
        {
            const __amdgpu_buffer_rsrc_t rrow = mi::wt_buffer(an, unsigned((H + 1) * sizeof(float)));
            for (int k = tid; k <= H; k += T)
                mi::wt_store(rrow, 4 * k, a[k]);
            return;
        }
        if (!(fl & 1))                                      // inactive: vAmp = 0 (Analyzer.cpp:363-364)
        {
            const __amdgpu_buffer_rsrc_t rrow = mi::wt_buffer(an, unsigned((H + 1) * sizeof(float)));
            for (int k = tid; k <= H; k += T)
                mi::wt_store(rrow, 4 * k, 0.0f);
            return;
        }
        constexpr int KPT = (H + T - 1) / T;
        // Analyzer.cpp:339-353: doff = head - (fft_size + delay), wrapped into the ring
        int64_t doff = int64_t(head) - int64_t(N) - int64_t(dly);
        while (doff < 0)
            doff += buf_size;
        const float *rb = rbw;
        const float2 *w2 = reinterpret_cast<const float2 *>(wnd);
        float2 xin[KPT], win[KPT];
        #pragma unroll
        for (int i = 0; i < KPT; ++i)
        {
            const int m = (tid + i * T) & (H - 1);
            uint32_t i0 = uint32_t(doff) + 2 * m;
            if (i0 >= buf_size) i0 -= buf_size;
            if (i0 + 1 < buf_size)
                xin[i] = *reinterpret_cast<const float2 *>(rb + i0);     // 4-byte aligned 8-byte load
            else
                xin[i] = make_float2(rb[i0], rb[0]);
            win[i] = w2[m];
        }
        float aold[KPT], aold_h = 0.0f;
        #pragma unroll
        for (int i = 0; i < KPT; ++i)
            aold[i] = (tid + i * T < H) ? a[tid + i * T] : 0.0f;
        if (tid == 0)
            aold_h = a[H];
        // the first IPF * T samples to ingest wait in registers
        constexpr int IPF = 8;
        float ing[IPF];
        #pragma unroll
        for (int j = 0; j < IPF; ++j)
        {
            const uint32_t i = tid + j * T;
            ing[j] = (i < ingest_n && !ingest_zero) ? ingest[size_t(ch) * ingest_stride + i] : 0.0f;
        }
        MI_APROBE(1);
        rf.prepare();
        MI_APROBE(2);
        constexpr bool REGS = !fplan<LOGH>::radix16 && (mi_fft::plan<LOGH>::T == mi_fft::plan<LOGH>::TB);
        if constexpr (REGS)
        {
            // the windowed frame goes into the transform in registers (fft_lds REG_IN): one LDS round trip and a barrier less
            v2f io[KPT];
            #pragma unroll
            for (int i = 0; i < KPT; ++i)
                io[i] = v2f{xin[i].x * win[i].x, xin[i].y * win[i].y};
            MI_APROBE(3);
            mi_fft::fft_lds<LOGH, false, true, false>(buf, scr, rf.ft, tid, io);
            mi_fft::real_split<LOGH>(buf, rf.rt, tid);
        }
        else
        {
            #pragma unroll
            for (int i = 0; i < KPT; ++i)
                if (tid + i * T < H)
                    buf[tid + i * T] = make_float2(xin[i].x * win[i].x, xin[i].y * win[i].y);
            __syncthreads();
            MI_APROBE(3);
            rf.forward(buf, scr, tid);
        }
        MI_APROBE(4);
        // pcomplex_mod over N/2+1 bins, then mix2(vAmp, mod, 1 - tau, tau) (Analyzer.cpp:359-361)
        const float keep = 1.0f - tau;
        const __amdgpu_buffer_rsrc_t ramp = mi::wt_buffer(an, unsigned((H + 1) * sizeof(float)));
        #pragma unroll
        for (int i = 0; i < KPT; ++i)
        {
            const int k = tid + i * T;
            if (k >= H)
                break;
            const float2 v = buf[k];
            const float mag = (k == 0) ? fabsf(v.x) : mag_root(v.x * v.x + v.y * v.y);
            mi::wt_store(ramp, 4 * k, mix2(aold[i], mag, keep, tau));
        }
        if (tid == 0)
            mi::wt_store(ramp, 4 * H, mix2(aold_h, fabsf(buf[0].y), keep, tau));
        MI_APROBE(5);
        // ring ingest: cells head .. head + ingest_n - 1 (mod size), none of them inside a window read above
        if (ingest_n > 0)
        {
            const __amdgpu_buffer_rsrc_t rring = mi::wt_buffer(rbw, unsigned(buf_size * sizeof(float)));
            #pragma unroll
            for (int j = 0; j < IPF; ++j)
            {
                const uint32_t i = tid + j * T;
                uint32_t w = head + i;
                if (w >= buf_size) w -= buf_size;
                if (i < ingest_n)
                    mi::wt_store(rring, int(w * sizeof(float)), ing[j]);
            }
            for (uint32_t i0 = IPF * T; i0 < ingest_n; i0 += IPT * T)
            {
                float v[IPT];
                #pragma unroll
                for (int j = 0; j < IPT; ++j)
                {
                    const uint32_t i = i0 + tid + j * T;
                    v[j] = (i < ingest_n && !ingest_zero) ? ingest[size_t(ch) * ingest_stride + i] : 0.0f;
                }
                #pragma unroll
                for (int j = 0; j < IPT; ++j)
                {
                    const uint32_t i = i0 + tid + j * T;
                    uint32_t w = head + i;
                    if (w >= buf_size) w -= buf_size;
                    if (i < ingest_n)
                        mi::wt_store(rring, int(w * sizeof(float)), v[j]);
                }
            }
        }
        MI_APROBE(6);
    }

    // The analysis of every channel at a strobe (one workgroup per channel).  (Round 3 also let the per-bin reduction ride on this
    // launch as a second role -- reduce workgroups waiting inside the launch for the rows; measured slower than two launches,
    // profiles/HISTORY.md, and removed in round 6.)
    template <int LOGH>
    __global__ __launch_bounds__(fplan<LOGH>::T)
    void analyzer_kernel(float *ring, uint32_t buf_size, uint32_t head,
                         const uint32_t *__restrict__ delay, const uint8_t *__restrict__ flags,
                         const float *__restrict__ wnd, const float *__restrict__ amp_old, float *amp_new,
                         uint32_t amp_stride, float tau, const float2 *__restrict__ tw,
                         const float *ingest, size_t ingest_stride, uint32_t ingest_n, int ingest_zero)
    {
        __shared__ float2 lds_[fplan<LOGH>::LDS];
        analyzer_role<LOGH>(lds_, ring, buf_size, head, delay, flags, wnd, amp_old, amp_new, amp_stride, tau, tw,
                            ingest, ingest_stride, ingest_n, ingest_zero);
    }


    // ---- several strobes of the analyzer in ONE launch (mi_analyzer_bank_process_reduce_frames) --------------------------------
    // The C5 shape: a strobe per call, the hop half a frame (period = N / 2), no user delays.  The frame of strobe f is then
    // the two hops BEFORE the call's block f -- hops the launch has just had in its hands.  Round 6: the launch leaves the RAW
    // magnitudes of strobe f in plane rows[f]; the smoothing vAmp = mix2(vAmp, |X|, 1 - tau, tau) (Analyzer.cpp:359-361) -- an
    // elementwise first-order recurrence over the strobes -- is walked by the reduction launch that reads every plane anyway
    // (bin_smooth_reduce_kernel), in the reference's order.  With that the strobes of a channel owe each other nothing:
    // channels x frames independent units instead of `channels` sequential walks.  What the launch leaves in the ring is what
    // `frames` launches of analyzer_kernel leave: every hop at its place.  Frozen and inactive channels only take their samples
    // (their rows are the reduction's business: kept / zero, Analyzer.cpp:334,363-364).
    constexpr int AN_FRAMES_MAX = 16;
    struct an_frames_args
    {
        int             frames;
        const float    *in[AN_FRAMES_MAX];      // block f of the call: ingested at strobe f
        float          *rows[AN_FRAMES_MAX];    // plane that takes |X| of strobe f (raw: not smoothed)
    };

    // Workgroup form (transforms of 1024 .. 8192 points other than 4096: the wave form below takes rank 12): a channel's
    // workgroup walks the frames with the window's halves in registers (the second half of frame f is the first half of
    // frame f + 1), the next block's samples asked for before the transform.  Same window products, same transform, same
    // magnitude per strobe as analyzer_kernel: with the reduction's mix2 the bits of frame-by-frame calls.
    template <int LOGH>
    __global__ __launch_bounds__(fplan<LOGH>::T, (fplan<LOGH>::T <= 256) ? 4 : 2)   // four workgroups of 256 threads per CU: 1024 channels in one round
    void analyzer_frames_kernel(const an_frames_args fa, size_t in_stride, bool aligned, float *ring, uint32_t buf_size, uint32_t head,
                                const uint8_t *__restrict__ flags, const float *__restrict__ wnd, uint32_t amp_stride,
                                const float2 *__restrict__ tw)
    {
        using PL = fplan<LOGH>;
        constexpr int H = PL::N, T = PL::T, N = 2 * H, KPT = H / T, HALF = KPT / 2, HOP = H;      // HOP samples = HOP / 2 pairs
        static_assert(!PL::radix16 && mi_fft::plan<LOGH>::T == mi_fft::plan<LOGH>::TB && (KPT % 2) == 0, "register hand-over of the transform");
        __shared__ float2 lds_[PL::LDS];
        float2 *const buf = lds_, *const scr = lds_ + PL::SCR;
        const int ch = blockIdx.x, tid = threadIdx.x;
        MI_APROBE(0);
        typename PL::real rf;
        rf.load(tw, TWN, tid);
        const uint8_t fl = flags[ch];                       // bit0: active, bit1: frozen
        float *rbw = ring + size_t(ch) * buf_size;
        const __amdgpu_buffer_rsrc_t rring = mi::wt_buffer(rbw, unsigned(buf_size * sizeof(float)));
        // block f's samples of this thread: pairs p = tid + j T of the hop
        auto load_hop = [&](int f, float2 (&v)[HALF]) {
            const float *x = fa.in[f] + size_t(ch) * in_stride;
            #pragma unroll
            for (int j = 0; j < HALF; ++j)
            {
                const int p = tid + j * T;
                v[j] = aligned ? *reinterpret_cast<const float2 *>(x + 2 * p) : make_float2(x[2 * p], x[2 * p + 1]);
            }
        };
        // ... into the ring behind the head of strobe f (Analyzer.cpp:371-398)
        // (an even head in a ring of even size: a pair never straddles the wrap and goes out as ONE 8-byte store -- 512 contiguous
        // bytes per wave instruction.  As two 4-byte stores every instruction wrote every other word of four cache lines and the
        // write-through traffic of the ingest counted twice: round 4's 1.40 x of the launch's algorithmic bytes,
        // profiles/r04_pmc_hbm_raw.json WRITE_SIZE 403 MB against 279 MB of rows and samples)
        auto ingest_hop = [&](int f, const float2 (&v)[HALF]) {
            uint64_t hc = uint64_t(head) + uint64_t(f) * HOP;           // (inside one turn of the ring: see ring_place)
            while (hc >= buf_size)
                hc -= buf_size;
            const uint32_t h0 = uint32_t(hc);
            const bool pairs = ((h0 | buf_size) & 1u) == 0u;
            #pragma unroll
            for (int j = 0; j < HALF; ++j)
            {
                uint32_t w0 = h0 + 2 * (tid + j * T);
                if (w0 >= buf_size) w0 -= buf_size;
                if (pairs)
                {
                    mi::wt_store(rring, int(w0 * sizeof(float)), v[j]);
                    continue;
                }
                uint32_t w1 = w0 + 1;
                if (w1 >= buf_size) w1 -= buf_size;
                mi::wt_store(rring, int(w0 * sizeof(float)), v[j].x);
                mi::wt_store(rring, int(w1 * sizeof(float)), v[j].y);
            }
        };
        if ((fl & 2) || !(fl & 1))
        {
            // frozen: vAmp stays (Analyzer.cpp:334); inactive: vAmp = 0 (:363-364) -- the reduction launch sees to that; the
            // samples go in all the same
            for (int f = 0; f < fa.frames; ++f)
            {
                float2 v[HALF];
                load_hop(f, v);
                ingest_hop(f, v);
            }
            return;
        }
        // the frame of the first strobe: the N samples in front of the head (Analyzer.cpp:339-353 with no delay)
        int64_t doff = int64_t(head) - int64_t(N);
        while (doff < 0)
            doff += buf_size;
        const float2 *w2 = reinterpret_cast<const float2 *>(wnd);
        float2 xin[KPT];
        #pragma unroll
        for (int i = 0; i < KPT; ++i)
        {
            const int m = tid + i * T;
            uint32_t i0 = uint32_t(doff) + 2 * m;
            if (i0 >= buf_size) i0 -= buf_size;
            if (i0 + 1 < buf_size)
                xin[i] = *reinterpret_cast<const float2 *>(rbw + i0);
            else
                xin[i] = make_float2(rbw[i0], rbw[0]);
        }
        MI_APROBE(1);
        rf.prepare();
        for (int f = 0; f < fa.frames; ++f)
        {
            // (the window comes out of the L2 at every strobe instead of occupying 16 registers: four workgroups fit a CU.  Asked
            // for one strobe ahead -- behind the transform, next to the block's request below -- it costs four spilled registers
            // and 8 % of the launch: measured, round 5)
            v2f io[KPT];
            #pragma unroll
            for (int i = 0; i < KPT; ++i)
            {
                const float2 w = w2[tid + i * T];
                io[i] = v2f{xin[i].x * w.x, xin[i].y * w.y};
            }
            #pragma unroll
            for (int j = 0; j < HALF; ++j)
                xin[j] = xin[j + HALF];
            if (f == 3) MI_APROBE(2);
            mi_fft::fft_lds<LOGH, false, true, false>(buf, scr, rf.ft, tid, io);
            mi_fft::real_split<LOGH>(buf, rf.rt, tid);
            if (f == 3) MI_APROBE(3);
            // this strobe's block is asked for NOW -- the transform's registers are free again, and the request is in flight
            // underneath the magnitudes and the row's stores instead of behind them (the stores to the rows and the loads of the
            // caller's block cannot be told apart by the compiler, which keeps them in program order)
            float2 blk[HALF];
            load_hop(f, blk);
            // pcomplex_mod over N/2+1 bins (Analyzer.cpp:359); mix2 follows in the reduction launch
            float *const row = fa.rows[f] + size_t(ch) * amp_stride;
            const __amdgpu_buffer_rsrc_t ramp = mi::wt_buffer(row, unsigned((H + 1) * sizeof(float)));
            #pragma unroll
            for (int i = 0; i < KPT; ++i)
            {
                const int k = tid + i * T;
                const float2 v = buf[k];
                mi::wt_store(ramp, 4 * k, (k == 0) ? fabsf(v.x) : mag_root(v.x * v.x + v.y * v.y));
            }
            if (tid == 0)
                mi::wt_store(ramp, 4 * H, fabsf(buf[0].y));
            if (f == 3) MI_APROBE(4);
            // this strobe's block: into the ring, and the second half of the next strobe's frame
            ingest_hop(f, blk);
            #pragma unroll
            for (int j = 0; j < HALF; ++j)
                xin[j + HALF] = blk[j];
            __syncthreads();                                // the transform's buffers are free for the next strobe
            if (f == 3) MI_APROBE(5);
            if (f == 2) MI_APROBE(7);
        }
        MI_APROBE(6);
    }

    // ---- the same run of strobes at rank 12 on the wave-resident transform (fft_wave.h; round 6) --------------------------------
    // A WAVE owns a PAIR of consecutive strobes of a channel: the frames of strobes 2p and 2p + 1 -- hops [h(2p-2) | h(2p-1)] and
    // [h(2p-1) | h(2p)], h(q) = block q of the call for q >= 0, what the ring holds in front of the head for q < 0 -- are the real
    // and the imaginary part of ONE 4096-point complex sequence z = w (A + i B); the spectra come apart as
    // X_A[k] = (Z[k] + conj Z[N-k]) / 2, X_B[k] = (Z[k] - conj Z[N-k]) / 2i, the partner bin by ds_bpermute (lane 64 - l, register
    // 63 - r; lane 0: register (64 - r) & 63), and only the magnitudes of bins 0 .. N/2 are wanted: one forward transform per TWO
    // strobes, no workgroup-wide LDS pass, no barrier behind the tables.  The magnitudes go out RAW (the smoothing walks the planes
    // in the reduction launch), so the eight waves of a workgroup -- the sixteen strobes of a channel -- owe each other nothing, two
    // waves per SIMD: round 5's form of this kernel kept vAmp in registers from strobe to strobe, i.e. ONE wave per channel and
    // SIMD, and a lone wave issued at half the rate (profiles/r05_experiments/analyzer_frames_wave.txt).
    // One workgroup per CU walks the channels blockIdx.x + i gridDim.x (tables filled once); wave p takes pair p of channel after
    // channel with the NEXT channel's hops in flight underneath this one's transform: h(2p-2) and h(2p-1) by LDS-DMA into the
    // wave's exchange area as soon as the transform's one exchange has left it (16-byte pieces: the area's image of a hop is its
    // samples in order, 64 j + lane = the sample a lane wants in register j), h(2p) into 32 registers.  In-kernel timeline
    // (tests/experiments/analyzer_wave_probe.hip): a unit never waits for its hops; what a wave spends is its own instructions, a
    // third of them memory instructions that issue against the queue's back-pressure -- hence wide pieces wherever the layout
    // allows: the DMA, and the ring ingest (Analyzer.cpp:371-398) of the two hops that lie in the area (wave p files hops 2p - 2
    // and 2p - 1 from there, 16 bytes per lane; the run's last one or two hops, which no area holds, are shared out among the
    // eight waves by rows).  Stores are non-temporal (nothing here is read again soon: 92 -> 80 us per launch).
    // The host sends a run this way only while the ring holds a frame AND the run's hops side by side (no wave overwrites what
    // another still reads).  Same products, another transform and another order of roundings: within 1e-6 of analyzer_kernel<11>,
    // not its bits.
    constexpr int ANW_WAVES = 8, ANW_NT = 2 /* cache policy: non-temporal */;
    __global__ __launch_bounds__(64 * ANW_WAVES, 2)
    void analyzer_frames_wave_kernel(const an_frames_args fa, size_t in_stride, int wide /* blocks and rows 16-byte aligned */, float *ring,
                                     uint32_t buf_size, uint32_t head, const uint8_t *__restrict__ flags, const float *__restrict__ wnd,
                                     uint32_t amp_stride, const float2 *__restrict__ tw, int channels)
    {
        using namespace mi_fftw;
        constexpr int HALF = R / 2, HOP = N / 2;            // registers of a hop (sample lane + 64 j); samples of a hop
        __shared__ float areas[ANW_WAVES][AREA];
        __shared__ float2 pl[16 * R];
        __shared__ float wnd_l[N];
        const int tid = threadIdx.x, lane = tid & 63;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        MI_WPROBE(30);
        fill_table_pq(pl, tw, tid, 64 * ANW_WAVES);
        for (int i = tid; i < N; i += 64 * ANW_WAVES)
            wnd_l[i] = wnd[i];
        // the flags (bit0: active, bit1: frozen) of the channels this workgroup walks, one per lane: a v_readlane away
        const int walk = (channels - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);      // channels blockIdx.x + i gridDim.x, i < walk
        auto flags_of = [&](int i0) -> uint32_t {
            const int c = int(blockIdx.x) + (i0 + lane) * int(gridDim.x);
            return (c < channels) ? uint32_t(flags[c]) : 0u;
        };
        uint32_t flv = flags_of(0);
        __syncthreads();
        MI_WPROBE(31);
        [[maybe_unused]] int unit_no = 0;                   // (the probe's stamps are numbered by it)
        const int pairs = (fa.frames + 1) / 2;
        auto at = [](__amdgpu_buffer_rsrc_t r, int lane_off, int row_off) -> float {
            return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, lane_off, row_off, 0));
        };
        auto put = [](float v, __amdgpu_buffer_rsrc_t r, int lane_off, int row_off) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, lane_off, row_off, ANW_NT);
        };
        // Where hop q of channel ch is read from (q >= 0: the caller's block q; q < 0: the ring in front of the head,
        // Analyzer.cpp:339-353 with no delay) and where it is filed (the ring behind the head of strobe q, :371-398): a buffer of
        // the hop's 8 KiB when it lies in one piece -- row j at 256 j, one lane offset in a VGPR and the row in the scalar offset --
        // or the whole ring and a first cell (the hop straddles the ring's end: once per turn, offsets per element there)
        struct place { __amdgpu_buffer_rsrc_t rsrc; uint32_t first; bool straight, quads; };
        auto ring_place = [&](int ch, int64_t cell) -> place {
            float *rbw = ring + size_t(ch) * buf_size;
            // (the host keeps a run inside one turn of the ring -- frame + run <= buf_size --: a step either way at most.  As
            // `cell % buf_size` every place cost a 64-bit division in scalar code: 500 of a unit's 650 scalar instructions)
            cell += (cell < 0) ? int64_t(buf_size) : 0;
            cell -= (cell >= int64_t(buf_size)) ? int64_t(buf_size) : 0;
            const uint32_t c0 = uint32_t(cell);
            if (c0 + HOP <= buf_size)
                return place{mi::wt_buffer(rbw + c0, unsigned(HOP * sizeof(float))), 0u, true, (c0 & 3u) == 0u};      // (rows of the ring start on 16 bytes)
            return place{mi::wt_buffer(rbw, unsigned(buf_size * sizeof(float))), c0, false, false};
        };
        auto source = [&](int ch, int q) -> place {
            if (q >= 0)
                return place{mi::wt_buffer(const_cast<float *>(fa.in[q]) + size_t(ch) * in_stride, unsigned(HOP * sizeof(float))), 0u, true, wide != 0};
            return ring_place(ch, int64_t(head) + int64_t(q) * HOP);
        };
        auto wrapped = [&](uint32_t first, int cell) -> int {             // byte offset of cell first + `cell` of a ring
            uint32_t c = first + uint32_t(cell);
            if (c >= buf_size) c -= buf_size;
            return int(c * sizeof(float));
        };
        // rows j0 .. j0 + ROWS - 1 of a hop into registers (sample lane + 64 j), and from registers into the ring
        auto load_rows = [&](const place &s, int j0, auto &v) {
            constexpr int ROWS = int(sizeof(v) / sizeof(v[0]));
            if (s.straight)
            {
                #pragma unroll
                for (int j = 0; j < ROWS; ++j)
                    v[j] = at(s.rsrc, lane * 4, 256 * (j0 + j));
                return;
            }
            int ln = lane;
            asm volatile("" : "+v"(ln));                    // (the wrapped offsets are made here, not ahead of the loops and spilled)
            #pragma unroll
            for (int j = 0; j < ROWS; ++j)
                v[j] = at(s.rsrc, wrapped(s.first, ln + 64 * (j0 + j)), 0);
        };
        auto file_rows = [&](const place &d, int j0, const auto &v) {
            constexpr int ROWS = int(sizeof(v) / sizeof(v[0]));
            if (d.straight)
            {
                #pragma unroll
                for (int j = 0; j < ROWS; ++j)
                    put(v[j], d.rsrc, lane * 4, 256 * (j0 + j));
                return;
            }
            int ln = lane;
            asm volatile("" : "+v"(ln));
            #pragma unroll
            for (int j = 0; j < ROWS; ++j)
                put(v[j], d.rsrc, wrapped(d.first, ln + 64 * (j0 + j)), 0);
        };
        auto flags_at = [&](int i) -> uint32_t {           // channel i of the walk (its flags in lane i & 63 of flv)
            if ((i & 63) == 0 && i > 0)
                flv = flags_of(i);
            return uint32_t(__builtin_amdgcn_readlane(int(flv), i & 63));
        };
        // the frozen and inactive channels of the walk only take their samples (the reduction launch keeps / zeroes their rows):
        // a pass of its own in front of the analyses, the hops shared out among the waves
        for (int i = 0; i < walk; ++i)
        {
            if ((flags_at(i) & 3u) == 1u)
                continue;
            const int ch = int(blockIdx.x) + i * int(gridDim.x);
            for (int q = wv; q < fa.frames; q += ANW_WAVES)
            {
                float v[HALF];
                load_rows(source(ch, q), 0, v);
                file_rows(ring_place(ch, int64_t(head) + int64_t(q) * HOP), 0, v);
            }
        }
        if (walk > 64)
            flv = flags_of(0);
        auto next_active = [&](int i) -> int {             // the first analysed channel of the walk at or behind position i (walk: none)
            while (i < walk && (flags_at(i) & 3u) != 1u)
                ++i;
            return i;
        };
        // The run's last hops -- 2 (pairs - 1) and, with an even count of strobes, the one behind it -- lie in no wave's area: their
        // 32 or 64 rows are filed by the eight waves, XROWS each (asked for with a unit's third hop, filed by the unit behind it)
        const int first_loose = 2 * (pairs - 1), loose_rows = (fa.frames - first_loose) * HALF / ANW_WAVES;          // 4 or 8
        constexpr int XROWS = 2 * HALF / ANW_WAVES;
        const int loose_q = first_loose + (wv * loose_rows) / HALF, loose_j0 = (wv * loose_rows) % HALF;
        auto load_loose = [&](int ch, float (&e)[XROWS]) {
            const place s = source(ch, loose_q);
            float (&lo)[XROWS / 2] = reinterpret_cast<float (&)[XROWS / 2]>(e[0]);
            float (&hi)[XROWS / 2] = reinterpret_cast<float (&)[XROWS / 2]>(e[XROWS / 2]);
            load_rows(s, loose_j0, lo);
            if (loose_rows == XROWS)
                load_rows(s, loose_j0 + XROWS / 2, hi);
        };
        auto file_loose = [&](int ch, const float (&e)[XROWS]) {
            const place d = ring_place(ch, int64_t(head) + int64_t(loose_q) * HOP);
            const float (&lo)[XROWS / 2] = reinterpret_cast<const float (&)[XROWS / 2]>(e[0]);
            const float (&hi)[XROWS / 2] = reinterpret_cast<const float (&)[XROWS / 2]>(e[XROWS / 2]);
            file_rows(d, loose_j0, lo);
            if (loose_rows == XROWS)
                file_rows(d, loose_j0 + XROWS / 2, hi);
        };
        const int p = wv;
        if (p >= pairs)
        {
            // (a run of fewer than fifteen strobes: a wave without a pair still files its share of the last hops)
            for (int i = next_active(0); i < walk; i = next_active(i + 1))
            {
                const int ch = int(blockIdx.x) + i * int(gridDim.x);
                float e[XROWS];
                load_loose(ch, e);
                file_loose(ch, e);
            }
            return;
        }
        const bool second = 2 * p + 1 < fa.frames;
        float *const area = areas[wv];
        const unsigned area_at = unsigned(uintptr_t(area));                // LDS byte address (wave-uniform)
        // LDS-DMA, written in asm: the compiler knows nothing of these loads (the wait at the head of a unit is ours).  A piece:
        // memory rsrc + voff + soff + offset -> LDS m0 + lane x (4 | 16) + offset; 4 KiB per statement.
        auto dma_quads = [&](__amdgpu_buffer_rsrc_t rsrc, int soff, unsigned m0v) __attribute__((always_inline)) {
            unsigned keep;
            const int voff = lane * 16;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                         "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
                         "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds\n\t"
                         "buffer_load_dwordx4 %1, %2, %3 offen offset:2048 lds\n\t"
                         "buffer_load_dwordx4 %1, %2, %3 offen offset:3072 lds\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(m0v) : "memory");
        };
        auto dma_words = [&](__amdgpu_buffer_rsrc_t rsrc, int soff, unsigned m0v) __attribute__((always_inline)) {
            unsigned keep;
            const int voff = lane * 4;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                         "buffer_load_dword %1, %2, %3 offen lds\n\t"
                         "buffer_load_dword %1, %2, %3 offen offset:256 lds\n\t"
                         "buffer_load_dword %1, %2, %3 offen offset:512 lds\n\t"
                         "buffer_load_dword %1, %2, %3 offen offset:768 lds\n\t"
                         "buffer_load_dword %1, %2, %3 offen offset:1024 lds\n\t"
                         "buffer_load_dword %1, %2, %3 offen offset:1280 lds\n\t"
                         "buffer_load_dword %1, %2, %3 offen offset:1536 lds\n\t"
                         "buffer_load_dword %1, %2, %3 offen offset:1792 lds\n\t"
                         "buffer_load_dword %1, %2, %3 offen offset:2048 lds\n\t"
                         "buffer_load_dword %1, %2, %3 offen offset:2304 lds\n\t"
                         "buffer_load_dword %1, %2, %3 offen offset:2560 lds\n\t"
                         "buffer_load_dword %1, %2, %3 offen offset:2816 lds\n\t"
                         "buffer_load_dword %1, %2, %3 offen offset:3072 lds\n\t"
                         "buffer_load_dword %1, %2, %3 offen offset:3328 lds\n\t"
                         "buffer_load_dword %1, %2, %3 offen offset:3584 lds\n\t"
                         "buffer_load_dword %1, %2, %3 offen offset:3840 lds\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(m0v) : "memory");
        };
        // hop q of channel ch -> the 8 KiB of LDS at byte address `to`, its samples in order
        auto dma_hop = [&](int ch, int q, unsigned to) {
            const place s = source(ch, q);
            if (s.straight && s.quads)
            {
                dma_quads(s.rsrc, 0, to);
                dma_quads(s.rsrc, 4096, to + 4096);
            }
            else if (s.straight)
            {
                dma_words(s.rsrc, 0, to);
                dma_words(s.rsrc, 4096, to + 4096);
            }
            else
                for (int j = 0; j < HALF; ++j)
                {
                    const int voff = wrapped(s.first, lane + 64 * j);
                    const unsigned m0v = to + 256u * unsigned(j);
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(voff), "s"(s.rsrc), "s"(m0v) : "memory");
                }
        };
        // ... and from the area into the ring: hop q (its image at word `from` of the area), 16 bytes per lane where its place allows
        auto file_area = [&](int ch, int q, int from) {
            const place d = ring_place(ch, int64_t(head) + int64_t(q) * HOP);
            if (d.straight && d.quads)
            {
                #pragma unroll
                for (int k = 0; k < HOP / 256; ++k)
                {
                    const float4 v = *reinterpret_cast<const float4 *>(area + from + 256 * k + 4 * lane);
                    const mi::u32x4 u = { __float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w) };
                    __builtin_amdgcn_raw_buffer_store_b128(u, d.rsrc, lane * 16, 1024 * k, ANW_NT);
                }
                return;
            }
            // (its place is not 16-byte aligned, or it straddles the ring's end: a word per lane; the cell's offset is formed per
            // row -- nothing of this rare path is kept in registers across the transform)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            #pragma unroll 1
            for (int j = 0; j < HALF; ++j)
                put(area[from + 64 * j + ln], d.rsrc, d.straight ? 4 * (ln + 64 * j) : wrapped(d.first, ln + 64 * j), 0);
        };
        int i = next_active(0);
        float h2[HALF], e[XROWS];
        if (i < walk)
        {
            const int ch = int(blockIdx.x) + i * int(gridDim.x);
            dma_hop(ch, 2 * p - 2, area_at);
            dma_hop(ch, 2 * p - 1, area_at + unsigned(HOP * sizeof(float)));
            load_rows(source(ch, 2 * p), 0, h2);
            load_loose(ch, e);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        while (i < walk)
        {
            const int ch = int(blockIdx.x) + i * int(gridDim.x);
            v2f x[R];
            // The unit's hops are here: every load is older than the 66 stores of the rows that followed them (completion in
            // order: all but the newest 63 operations done = every load done)
            asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
            MI_WPROBE(unit_no * 6 + 0);
            // strobes 2p and 2p + 1: A = [h(2p-2) | h(2p-1)], B = [h(2p-1) | h(2p)], times the window
            #pragma unroll
            for (int j = 0; j < HALF; ++j)
            {
                const float w0 = wnd_l[lane + 64 * j], w1 = wnd_l[lane + 64 * (j + HALF)];
                const float a = area[64 * j + lane], b = area[HOP + 64 * j + lane];
                x[j] = v2f{a * w0, b * w0};
                x[j + HALF] = v2f{b * w1, h2[j] * w1};
            }
            if (p > 0)
            {
                file_area(ch, 2 * p - 2, 0);
                file_area(ch, 2 * p - 1, HOP);
            }
            MI_WPROBE(unit_no * 6 + 1);
            file_loose(ch, e);
            MI_WPROBE(unit_no * 6 + 2);
            i = next_active(i + 1);
            const int chn = int(blockIdx.x) + i * int(gridDim.x);
            fft4096_t<false>(x, pl, area, lane, [&]() {
                // the exchange is through (and the area's hops filed: the LDS pipe takes a wave's accesses in order): the area takes
                // the next unit's first two hops, 32 registers the third (the transform's first half has none to spare)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                MI_WPROBE(unit_no * 6 + 3);
                if (i < walk)
                {
                    dma_hop(chn, 2 * p - 2, area_at);
                    dma_hop(chn, 2 * p - 1, area_at + unsigned(HOP * sizeof(float)));
                    load_rows(source(chn, 2 * p), 0, h2);
                    load_loose(chn, e);
                }
            });
            MI_WPROBE(unit_no * 6 + 4);
            // pcomplex_mod over N/2 + 1 bins (Analyzer.cpp:359) of strobe 2p and of strobe 2p + 1
            const unsigned row_bytes = unsigned((HOP + 1) * sizeof(float));
            const __amdgpu_buffer_rsrc_t row_a = mi::wt_buffer(fa.rows[2 * p] + size_t(ch) * amp_stride, row_bytes);
            const __amdgpu_buffer_rsrc_t row_b = mi::wt_buffer(fa.rows[second ? 2 * p + 1 : 2 * p] + size_t(ch) * amp_stride, second ? row_bytes : 0u);
            const int paddr = ((64 - lane) & 63) * 4;
            const bool l0 = lane == 0;
            // 2 X_A = Z + conj P, 2i X_B = Z - conj P with P = Z[N - k]: lane 64 - l, register 63 - r (lane 0: register (64 - r) & 63)
            auto mags = [&](v2f z, v2f pz, float &ma, float &mb) {
                const float ex = z.x + pz.x, ey = z.y - pz.y, ox = z.x - pz.x, oy = z.y + pz.y;
                ma = 0.5f * mag_root(ex * ex + ey * ey);
                mb = 0.5f * mag_root(ox * ox + oy * oy);
            };
            constexpr int CH = 8;                           // partners asked for together (registers: the next channel's hop is waiting)
            #pragma unroll
            for (int r0 = 0; r0 < HALF; r0 += CH)
            {
                v2f part[CH];
                #pragma unroll
                for (int r = 0; r < CH; ++r)
                    part[r] = from_partner(paddr, x[63 - (r0 + r)]);
                #pragma unroll
                for (int rr = 0; rr < CH; ++rr)
                {
                    const int r = r0 + rr;
                    const v2f own = x[(64 - r) & 63];
                    const v2f pz = v2f{l0 ? own.x : part[rr].x, l0 ? own.y : part[rr].y};
                    float ma, mb;
                    mags(x[r], pz, ma, mb);
                    put(ma, row_a, lane * 4, 256 * r);
                    put(mb, row_b, lane * 4, 256 * r);
                }
            }
            if (l0)                                         // bin N / 2: lane 0's register HALF, its own partner
            {
                float ma, mb;
                mags(x[HALF], x[HALF], ma, mb);
                put(ma, row_a, 4 * HOP, 0);
                put(mb, row_b, 4 * HOP, 0);
            }
            MI_WPROBE(unit_no * 6 + 5);
            ++unit_no;
        }
        MI_WPROBE(29);
    }

    // ---- smoothing + per-bin reduction of a run of strobes (round 6) ----------------------------------------------------------------
    // The planes hold the RAW magnitudes of strobes 0 .. frames - 1 (analyzer_frames_*_kernel).  A thread owns one bin of a block
    // of `block` consecutive channels: channel after channel it walks vAmp = mix2(vAmp, |X_f|, 1 - tau, tau) over the strobes in
    // the reference's order (Analyzer.cpp:355-361; frozen: vAmp stays, :334; inactive: 0, :363-364) starting from the bank's vAmp,
    // adds every strobe's smoothed value to that strobe's block sum -- in channel order: the order of bin_reduce_body -- and leaves
    // vAmp after the last strobe in `amp` and after the one before it in `data` (what the strobe publishes, Analyzer.cpp:321-326).
    // A wave owns 64 bins (two 128-byte lines of every row) of ONE block, so the channel -- its flags, every plane's row -- is a
    // scalar; a workgroup owns SMR_BLOCKS blocks, whose sums go through the first levels of bin_reduce_body's binary tree (element
    // j, a multiple of 2 s, takes in element j + s) into `partial` [slice][frame][bins_stride]; bin_combine_kernel walks the
    // tree's upper levels over the slices.  Same additions in the same order as frames x (analyzer_kernel, bin_reduce_kernel): the
    // sums' bits depend on the magnitudes alone.  SMR_AHEAD channels' rows (17 loads each) are in flight together -- a thread's
    // channels are a chain of round trips otherwise; nothing is loaded under a condition a load decides (a frozen or inactive
    // channel's planes are read and not used).
        constexpr uint32_t SMR_BINS = 64, SMR_BLOCKS = 4, SMR_AHEAD = 4;      // (2, 4, 8 channels ahead measure 39, 38, 43 us at C5)
    struct smooth_planes { const float *raw[AN_FRAMES_MAX]; };
    template <bool FULL /* all AN_FRAMES_MAX strobes */>
    __global__ __launch_bounds__(64 * SMR_BLOCKS)
    void bin_smooth_reduce_kernel(float *__restrict__ partial, const smooth_planes sp, int frames_, float *__restrict__ amp,
                                  float *__restrict__ data, uint32_t stride, uint32_t channels, uint32_t bins,
                                  const uint8_t *__restrict__ flags, float tau, uint32_t block)
    {
        __shared__ float part[AN_FRAMES_MAX][SMR_BLOCKS][SMR_BINS];
        const int frames = FULL ? AN_FRAMES_MAX : frames_;
        const uint32_t lane = threadIdx.x & 63;
        const uint32_t w = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));
        const uint32_t k = blockIdx.x * SMR_BINS + lane;
        const uint32_t j = blockIdx.y * SMR_BLOCKS + w;                    // the block of channels
        const float keep = 1.0f - tau;
        float s[AN_FRAMES_MAX];
        #pragma unroll
        for (int f = 0; f < AN_FRAMES_MAX; ++f)
            s[f] = 0.0f;
        const uint32_t c0 = j * block;
        if (c0 < channels)
        {
            const uint32_t c1 = (c0 + block < channels) ? c0 + block : channels;
            const uint32_t kk = (k < bins) ? k : bins - 1;                  // (lanes past the last bin: its value again, not kept)
            for (uint32_t cb = c0; cb < c1; cb += 64)
            {
                // the flags of up to 64 channels, one per lane: a channel's flags are then a v_readlane away, not a round trip
                // in front of its rows' requests
                const uint32_t flv = flags[(cb + lane < c1) ? cb + lane : c1 - 1];
                const uint32_t ce = (cb + 64 < c1) ? cb + 64 : c1;
                for (uint32_t cc = cb; cc < ce; cc += SMR_AHEAD)
                {
                    float m[SMR_AHEAD][AN_FRAMES_MAX], a0[SMR_AHEAD];
                    #pragma unroll
                    for (uint32_t u = 0; u < SMR_AHEAD; ++u)
                    {
                        const uint32_t c = (cc + u < ce) ? cc + u : ce - 1;    // (past the end: the last channel again, not used)
                        const size_t at = size_t(c) * stride + kk;
                        a0[u] = amp[at];
                        #pragma unroll
                        for (int f = 0; f < AN_FRAMES_MAX; ++f)
                        {
                            // (a run of fewer strobes: plane 0 again -- an unconditional request; its value is not used)
                            const float *__restrict__ plane = sp.raw[(f < frames) ? f : 0];
                            m[u][f] = plane[at];
                        }
                    }
                    #pragma unroll
                    for (uint32_t u = 0; u < SMR_AHEAD; ++u)
                    {
                        if (cc + u >= ce)
                            break;
                        const size_t at = size_t(cc + u) * stride + kk;
                        const uint32_t fl = uint32_t(__builtin_amdgcn_readlane(int(flv), int(cc + u - cb)));
                        const bool active = (fl & 3u) == 1u;
                        float a = (fl == 0u) ? 0.0f : a0[u];                // inactive (and not frozen): vAmp = 0 from the first strobe on
                        #pragma unroll
                        for (int f = 0; f < AN_FRAMES_MAX; ++f)
                        {
                            if (f < frames)
                            {
                                a = active ? mix2(a, m[u][f], keep, tau) : a;
                                s[f] += a;
                                if (f == frames - 2 && k < bins)
                                    data[at] = a;
                            }
                        }
                        if (k < bins)
                            amp[at] = a;
                    }
                }
            }
        }
        #pragma unroll
        for (int f = 0; f < AN_FRAMES_MAX; ++f)
            part[f][w][lane] = s[f];
        __syncthreads();
        // tree levels s = 1, 2 over the four block sums of the slice
        static_assert(SMR_BLOCKS == 4, "the tree below is written out for four block sums");
        for (uint32_t i = threadIdx.x; i < uint32_t(frames) * SMR_BINS; i += 64 * SMR_BLOCKS)
        {
            const uint32_t f = i / SMR_BINS, bb = i & (SMR_BINS - 1);
            const float v = (part[f][0][bb] + part[f][1][bb]) + (part[f][2][bb] + part[f][3][bb]);
            if (blockIdx.x * SMR_BINS + bb < stride)
                partial[(size_t(blockIdx.y) * AN_FRAMES_MAX + f) * stride + blockIdx.x * SMR_BINS + bb] = v;
        }
    }

    // The upper levels of the tree: out[f][k] = tree over the slices' partial sums, times the envelope where one is asked for.  The
    // slices come in order; st[b] holds the sum of a complete aligned run of 2^b slices waiting for its right-hand neighbour (a
    // binary counter: slice i's carries are the set low bits of i, all scalars) -- element j, a multiple of 2 s, takes in element
    // j + s, a slice that is not there enters as + 0: bin_reduce_body's tree.  (As the last-arriving workgroup's job inside the
    // launch above it measured 10 us more than this launch's 4.8: a release, an add, an acquire and the reads at the end of
    // every group of bins at once.)
    constexpr uint32_t CMB_AHEAD = 16, CMB_LEVELS = 9;                      // up to 256 slices (REDUCE_MAX_BLOCKS / SMR_BLOCKS)
    __global__ __launch_bounds__(256)
    void bin_combine_kernel(float *out, size_t out_stride, const float *__restrict__ partial, uint32_t slices, uint32_t stride,
                            uint32_t bins, const float *__restrict__ env)
    {
        const uint32_t k = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
        if (k >= bins)
            return;
        float st[CMB_LEVELS];
        #pragma unroll
        for (uint32_t b = 0; b < CMB_LEVELS; ++b)
            st[b] = 0.0f;
        for (uint32_t i0 = 0; i0 < slices; i0 += CMB_AHEAD)
        {
            float v[CMB_AHEAD];
            #pragma unroll
            for (uint32_t u = 0; u < CMB_AHEAD; ++u)
                v[u] = (i0 + u < slices) ? partial[(size_t(i0 + u) * AN_FRAMES_MAX + f) * stride + k] : 0.0f;
            #pragma unroll
            for (uint32_t u = 0; u < CMB_AHEAD; ++u)
            {
                const uint32_t i = i0 + u;
                if (i >= slices)
                    break;
                float x = v[u];
                bool carry = true;
                #pragma unroll
                for (uint32_t b = 0; b < CMB_LEVELS; ++b)
                {
                    if (carry && ((i >> b) & 1u))
                        x = st[b] + x;
                    else if (carry)
                    {
                        st[b] = x;
                        carry = false;
                    }
                }
            }
        }
        // the runs that are left are the set bits of `slices`; a run enters the level above with + 0 at its right until it meets one
        float acc = 0.0f;
        bool have = false;
        #pragma unroll
        for (uint32_t b = 0; b < CMB_LEVELS; ++b)
            if ((slices >> b) & 1u)
            {
                acc = have ? st[b] + acc : st[b];
                have = true;
            }
        out[size_t(f) * out_stride + k] = (env != nullptr) ? acc * env[k] : acc;
    }

    // ---- analyzer frames above 2^14 samples: the steps of analyzer_kernel as plain launches around the four-step transform
    // work[n] = (ring[(head - N - delay + n) mod size] * w[n], 0) for the channels that are analysed
    __global__ __launch_bounds__(256)
    void big_an_gather_kernel(float2 *work, const float *__restrict__ ring, uint32_t buf_size, uint32_t head,
                              const uint32_t *__restrict__ delay, const uint8_t *__restrict__ flags,
                              const float *__restrict__ wnd, uint32_t N)
    {
        const uint32_t n = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
        if (n >= N || flags[ch] != 1)                       // frozen or inactive: nothing is transformed
            return;
        int64_t doff = int64_t(head) - int64_t(N) - int64_t(delay[ch]);     // Analyzer.cpp:339-353
        while (doff < 0)
            doff += buf_size;
        uint32_t i = uint32_t((uint64_t(doff) + n) % buf_size);
        work[size_t(ch) * N + n] = make_float2(ring[size_t(ch) * buf_size + i] * wnd[n], 0.0f);
    }

    // vAmp = mix2(vAmp, |X|, 1 - tau, tau) over N/2 + 1 bins (Analyzer.cpp:359-361); frozen: kept (:334); inactive: 0 (:363-364)
    __global__ __launch_bounds__(256)
    void big_an_mag_kernel(const float2 *__restrict__ spec, const float *__restrict__ amp_old, float *amp_new, uint32_t amp_stride,
                           float tau, const uint8_t *__restrict__ flags, uint32_t N)
    {
        const uint32_t k = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
        if (k > N / 2)
            return;
        const uint8_t fl = flags[ch];
        const float a = amp_old[size_t(ch) * amp_stride + k];
        float r;
        if (fl & 2)
            r = a;
        else if (!(fl & 1))
            r = 0.0f;
        else
        {
            const float2 v = spec[size_t(ch) * N + k];
            r = mix2(a, mag_root(v.x * v.x + v.y * v.y), 1.0f - tau, tau);
        }
        amp_new[size_t(ch) * amp_stride + k] = r;
    }

    // the samples that follow the strobe go into the ring behind head (Analyzer.cpp:371-398)
    __global__ __launch_bounds__(256)
    void an_ingest_kernel(float *ring, uint32_t buf_size, uint32_t head, const float *__restrict__ in, size_t in_stride,
                          uint32_t n, int zero)
    {
        const uint32_t i = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
        if (i >= n)
            return;
        uint32_t w = head + i;
        if (w >= buf_size) w -= buf_size;
        ring[size_t(ch) * buf_size + w] = zero ? 0.0f : in[size_t(ch) * in_stride + i];
    }

    // out[c][i] = data[c][idx[i]] * env[idx[i]]   (Analyzer::get_spectrum, Analyzer.cpp:443-456)
    __global__ __launch_bounds__(256)
    void spectrum_gather_kernel(float *out, size_t out_stride, const float *__restrict__ data, uint32_t stride,
                                const float *__restrict__ env, const uint32_t *__restrict__ idx, uint32_t count)
    {
        const uint32_t i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
        if (i >= count)
            return;
        const uint32_t j = idx[i];
        out[size_t(c) * out_stride + i] = data[size_t(c) * stride + j] * env[j];
    }

    // ---- host side window / envelope generators (misc/windows.cpp, misc/envelope.cpp) live in host/windows.cpp
} // namespace

namespace mi
{
    void make_window(float *dst, size_t n, int type);                       // host/windows.cpp
    void make_reverse_noise_lin(float *dst, float first, float last, float center, size_t n, int type);
}

// =============================================================================================================
struct mi_spectral_bank
{
    uint32_t    channels = 0, max_rank = 0, rank = 0;
    float       phase = 0.0f;
    bool        update = true;              // bUpdate: settings not applied yet
    bool        eager = false;              // transform when the frame fills (Multi...) / when the next sample comes
    uint32_t    offset = 0;                 // nOffset
    int         op = MI_SPECTRAL_OP_NONE;
    mi_spectral_func_t func = nullptr;
    void       *object = nullptr, *subject = nullptr;
    float      *d_in = nullptr, *d_out = nullptr, *d_wnd = nullptr, *d_wnd_out = nullptr, *d_mask = nullptr;
    int         wnd_in = MI_WINDOW_COSINE, wnd_out = MI_WINDOW_COSINE;     // < 0: no window
    float2     *d_spec = nullptr;
    uint8_t    *d_active = nullptr, *d_has_out = nullptr;   // MultiSpectralProcessor bindings (NULL: all bound)
    size_t      mask_stride = 0;
    const float2 *d_tw = nullptr;
    float2     *d_big_work = nullptr, *d_big_tmp = nullptr;    // frames above 2^14 samples: the transform's scratch
};

namespace mi
{
    // what the launches of a call take by value from the host: the fill of the frame being received
    uint64_t spectral_bank_positions(const void *bank)
    {
        const mi_spectral_bank *b = static_cast<const mi_spectral_bank *>(bank);
        return position_mix(b->offset, uint64_t(b->update));
    }
}

namespace
{
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

    int spectral_apply_settings(mi_spectral_bank *b, hipStream_t st)
    {
        // SpectralProcessor::update_settings (SpectralProcessor.cpp:107-125)
        const size_t N = size_t(1) << b->rank;
        std::vector<float> w(N);
        if (b->wnd_in >= 0)
        {
            mi::make_window(w.data(), N, b->wnd_in);
            MI_HIP_CHECK(hipMemcpyAsync(b->d_wnd, w.data(), N * sizeof(float), hipMemcpyHostToDevice, st));
            MI_HIP_CHECK(hipStreamSynchronize(st));
        }
        if (b->wnd_out >= 0)
            mi::make_window(w.data(), N, b->wnd_out);
        else
            std::fill(w.begin(), w.end(), 1.0f);
        MI_HIP_CHECK(hipMemcpyAsync(b->d_wnd_out, w.data(), N * sizeof(float), hipMemcpyHostToDevice, st));
        MI_HIP_CHECK(hipStreamSynchronize(st));
        MI_HIP_CHECK(hipMemsetAsync(b->d_in, 0, size_t(b->channels) * N * sizeof(float), st));
        MI_HIP_CHECK(hipMemsetAsync(b->d_out, 0, size_t(b->channels) * N * sizeof(float), st));
        b->offset = uint32_t(N * (b->phase * 0.5f));
        b->update = false;
        return MI_OK;
    }

    template <bool INVERSE>
    int big_fft_run(float2 *dst, const float2 *src, float2 *tmp, uint32_t rank, uint32_t channels, const float2 *tw, hipStream_t st)
    {
        const uint32_t N = 1u << rank, N2 = N / uint32_t(BIG_N1);
        hipLaunchKernelGGL((big_rows_kernel<INVERSE>), dim3(N2, channels), dim3(plan<BIG_LOG1>::T), 0, st, tmp, src, N2, N, tw);
        MI_HIP_CHECK(hipGetLastError());
        const dim3 grid(BIG_N1 / 256, channels);
        switch (rank - BIG_LOG1)
        {
            case 2:  hipLaunchKernelGGL((big_cols_kernel<2, INVERSE>), grid, dim3(256), 0, st, dst, tmp, N); break;
            case 3:  hipLaunchKernelGGL((big_cols_kernel<3, INVERSE>), grid, dim3(256), 0, st, dst, tmp, N); break;
            case 4:  hipLaunchKernelGGL((big_cols_kernel<4, INVERSE>), grid, dim3(256), 0, st, dst, tmp, N); break;
            default: hipLaunchKernelGGL((big_cols_kernel<5, INVERSE>), grid, dim3(256), 0, st, dst, tmp, N); break;
        }
        MI_HIP_CHECK(hipGetLastError());
        return MI_OK;
    }

    template <bool INVERSE>
    int big_fft(mi_spectral_bank *b, float2 *dst, const float2 *src, hipStream_t st)
    {
        return big_fft_run<INVERSE>(dst, src, b->d_big_tmp, b->rank, b->channels, b->d_tw, st);
    }

    // a hop of a frame above 2^14 samples: the same steps as stft_hop_kernel / stft_inverse_kernel, one launch each
    int spectral_hop_big(mi_spectral_bank *b, hipStream_t st, bool analyze_only)
    {
        const uint32_t N = 1u << b->rank, frame = N >> 1;
        const dim3 gN((N + 255) / 256, b->channels), gH((frame + 255) / 256, b->channels);
        const bool callback = (b->op == MI_SPECTRAL_OP_CALLBACK) && (b->func != nullptr);
        const bool transform = callback || (b->op == MI_SPECTRAL_OP_MASK);
        if (analyze_only && !callback)
        {
            hipLaunchKernelGGL(stft_shift_kernel, gH, dim3(256), 0, st, b->d_in, b->d_out, frame);
            MI_HIP_CHECK(hipGetLastError());
            return MI_OK;
        }
        hipLaunchKernelGGL(big_window_kernel, gN, dim3(256), 0, st, b->d_big_work, b->d_in,
                           (b->wnd_in >= 0) ? b->d_wnd : (const float *)nullptr, N);
        MI_HIP_CHECK(hipGetLastError());
        if (transform)
        {
            int r = big_fft<false>(b, b->d_spec, b->d_big_work, st);
            if (r != MI_OK)
                return r;
            if (callback)
            {
                if (b->d_active != nullptr)
                {
                    hipLaunchKernelGGL(big_unbound_kernel, gN, dim3(256), 0, st, b->d_spec, b->d_big_work, b->d_active, N);
                    MI_HIP_CHECK(hipGetLastError());
                }
                b->func(b->object, b->subject, reinterpret_cast<float *>(b->d_spec), b->rank, b->channels, st);
            }
            else
            {
                hipLaunchKernelGGL(big_mask_kernel, gN, dim3(256), 0, st, b->d_spec, b->d_mask, b->mask_stride, N);
                MI_HIP_CHECK(hipGetLastError());
            }
            if (analyze_only)
            {
                hipLaunchKernelGGL(stft_shift_kernel, gH, dim3(256), 0, st, b->d_in, b->d_out, frame);
                MI_HIP_CHECK(hipGetLastError());
                return MI_OK;
            }
            r = big_fft<true>(b, b->d_big_work, b->d_spec, st);
            if (r != MI_OK)
                return r;
        }
        hipLaunchKernelGGL(big_ola_kernel, gH, dim3(256), 0, st, b->d_in, b->d_out, b->d_big_work,
                           callback ? b->d_spec : (const float2 *)nullptr, b->d_wnd_out, transform ? 1.0f / float(N) : 1.0f, N,
                           callback ? b->d_active : (const uint8_t *)nullptr, callback ? b->d_has_out : (const uint8_t *)nullptr);
        MI_HIP_CHECK(hipGetLastError());
        return MI_OK;
    }

    // io_src / io_dst (callback operation only): the whole frame the hop transforms is taken from the caller and the frame
    // finished by the previous hop handed out inside the hop's first launch (see stft_hop_kernel)
    int spectral_hop(mi_spectral_bank *b, hipStream_t st, bool analyze_only = false, const float *io_src = nullptr,
                     size_t io_src_stride = 0, float *io_dst = nullptr, size_t io_dst_stride = 0)
    {
        if (b->rank > 14)
            return spectral_hop_big(b, st, analyze_only);
        const int lh = int(b->rank) - 1;
        const dim3 grid(b->channels);
        if (analyze_only)
        {
            if (b->op == MI_SPECTRAL_OP_CALLBACK && b->func != nullptr)
            {
                #define MI_CALL(LH) hipLaunchKernelGGL((stft_hop_kernel<LH, 2>), grid, dim3(fplan<LH>::T), 0, st, \
                    b->d_in, b->d_out, (b->wnd_in >= 0) ? b->d_wnd : (const float *)nullptr, b->d_wnd_out, (const float *)nullptr, size_t(0), b->d_spec, b->d_active, b->d_tw, (const float *)nullptr, size_t(0), (float *)nullptr, size_t(0))
                MI_LOGH_SWITCH(lh, MI_CALL)
                #undef MI_CALL
                MI_HIP_CHECK(hipGetLastError());
                b->func(b->object, b->subject, reinterpret_cast<float *>(b->d_spec), b->rank, b->channels, st);
            }
            const uint32_t frame = 1u << (b->rank - 1);
            hipLaunchKernelGGL(stft_shift_kernel, dim3((frame + 255) / 256, b->channels), dim3(256), 0, st, b->d_in, b->d_out, frame);
            MI_HIP_CHECK(hipGetLastError());
            return MI_OK;
        }
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        mi::take_profile_events(&ev0, &ev1);
        const bool bound = (b->op == MI_SPECTRAL_OP_CALLBACK) ? (b->func != nullptr) : true;
        if (b->op == MI_SPECTRAL_OP_NONE || !bound)
        {
            #define MI_CALL(LH) MI_LAUNCH((stft_hop_kernel<LH, 0>), grid, dim3(fplan<LH>::T), 0, st, ev0, ev1, \
                b->d_in, b->d_out, (b->wnd_in >= 0) ? b->d_wnd : (const float *)nullptr, b->d_wnd_out, (const float *)nullptr, size_t(0), (float2 *)nullptr, \
                (const uint8_t *)nullptr, b->d_tw, (const float *)nullptr, size_t(0), (float *)nullptr, size_t(0))
            MI_LOGH_SWITCH(lh, MI_CALL)
            #undef MI_CALL
        }
        else if (b->op == MI_SPECTRAL_OP_MASK)
        {
            #define MI_CALL(LH) MI_LAUNCH((stft_hop_kernel<LH, 1>), grid, dim3(fplan<LH>::T), 0, st, ev0, ev1, \
                b->d_in, b->d_out, (b->wnd_in >= 0) ? b->d_wnd : (const float *)nullptr, b->d_wnd_out, b->d_mask, b->mask_stride, (float2 *)nullptr, \
                (const uint8_t *)nullptr, b->d_tw, (const float *)nullptr, size_t(0), (float *)nullptr, size_t(0))
            MI_LOGH_SWITCH(lh, MI_CALL)
            #undef MI_CALL
        }
        else
        {
            #define MI_CALL(LH) MI_LAUNCH((stft_hop_kernel<LH, 2>), grid, dim3(fplan<LH>::T), 0, st, ev0, ev1, \
                b->d_in, b->d_out, (b->wnd_in >= 0) ? b->d_wnd : (const float *)nullptr, b->d_wnd_out, (const float *)nullptr, size_t(0), b->d_spec, b->d_active, b->d_tw, io_src, io_src_stride, io_dst, io_dst_stride)
            MI_LOGH_SWITCH(lh, MI_CALL)
            #undef MI_CALL
            MI_HIP_CHECK(hipGetLastError());
            b->func(b->object, b->subject, reinterpret_cast<float *>(b->d_spec), b->rank, b->channels, st);
            #define MI_CALL(LH) hipLaunchKernelGGL((stft_inverse_kernel<LH>), grid, dim3(fplan<LH>::T), 0, st, \
                b->d_in, b->d_out, b->d_wnd_out, b->d_spec, b->d_active, b->d_has_out, b->d_tw)
            MI_LOGH_SWITCH(lh, MI_CALL)
            #undef MI_CALL
        }
        MI_HIP_CHECK(hipGetLastError());
        return MI_OK;
    }
} // namespace

namespace mi
{
    int big_fft_run(bool inverse, float2 *dst, const float2 *src, float2 *tmp, uint32_t rank, uint32_t channels, const float2 *tw,
                    hipStream_t st)
    {
        if (rank <= uint32_t(BIG_LOG1) + 1 || rank > uint32_t(BIG_MAX_RANK))
            return fail(MI_EINVAL, "big_fft_run: rank %u outside 15..%d", rank, BIG_MAX_RANK);
        return inverse ? ::big_fft_run<true>(dst, src, tmp, rank, channels, tw, st)
                       : ::big_fft_run<false>(dst, src, tmp, rank, channels, tw, st);
    }
}

extern "C" {

int mi_spectral_bank_create(mi_spectral_bank_t **bank, uint32_t channels, uint32_t max_rank)
{
    MI_REQUIRE(bank != nullptr, MI_EINVAL, "mi_spectral_bank_create: NULL result pointer");
    *bank = nullptr;
    MI_REQUIRE(channels > 0, MI_EINVAL, "mi_spectral_bank_create: channels must be > 0");
    MI_REQUIRE(max_rank >= 5 && max_rank <= uint32_t(BIG_MAX_RANK), MI_EINVAL,
               "mi_spectral_bank_create: max_rank %u outside the supported 5..%d (frames of 32..%d samples)", max_rank,
               BIG_MAX_RANK, 1 << BIG_MAX_RANK);
    MI_REQUIRE(mi_dspu_device_count() > 0, MI_ENODEV, "no HIP device available (there is no CPU fallback)");
    mi_spectral_bank *b = new (std::nothrow) mi_spectral_bank();
    MI_REQUIRE(b != nullptr, MI_ENOMEM, "mi_spectral_bank_create: out of host memory");
    b->channels = channels;
    b->max_rank = b->rank = max_rank;                       // SpectralProcessor::init sets nRank = max_rank
    const size_t N = size_t(1) << max_rank;
    int twn = 0;
    int r = mi::fft_twiddles(&b->d_tw, &twn);
    hipError_t e = hipSuccess;
    if (r == MI_OK)
    {
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_in), size_t(channels) * N * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_out), size_t(channels) * N * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_wnd), N * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_wnd_out), N * sizeof(float));
        if (max_rank > 14)                                  // frames that go through global memory: two complex scratch frames
        {                                                   // and the spectrum buffer per channel
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_big_work), size_t(channels) * N * sizeof(float2));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_big_tmp), size_t(channels) * N * sizeof(float2));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_spec), size_t(channels) * N * sizeof(float2));
        }
        if (e != hipSuccess)
            r = mi::fail(e == hipErrorOutOfMemory ? MI_ENOMEM : MI_EHIP, "mi_spectral_bank_create: %s", hipGetErrorString(e));
    }
    if (r != MI_OK)
    {
        mi_spectral_bank_destroy(b);
        return r;
    }
    *bank = b;
    return MI_OK;
}

int mi_spectral_bank_destroy(mi_spectral_bank_t *b)
{
    if (b == nullptr)
        return MI_OK;
    (void)hipFree(b->d_in); (void)hipFree(b->d_out); (void)hipFree(b->d_wnd); (void)hipFree(b->d_wnd_out); (void)hipFree(b->d_mask);
    (void)hipFree(b->d_spec); (void)hipFree(b->d_active); (void)hipFree(b->d_has_out);
    (void)hipFree(b->d_big_work); (void)hipFree(b->d_big_tmp);
    delete b;
    return MI_OK;
}

int mi_spectral_bank_set_rank(mi_spectral_bank_t *b, uint32_t rank)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_spectral_bank_set_rank: NULL bank");
    if (rank == b->rank || rank > b->max_rank)              // SpectralProcessor.cpp:140-141: silently ignored
        return MI_OK;
    MI_REQUIRE(rank >= 5, MI_EINVAL, "mi_spectral_bank_set_rank: rank %u below the supported minimum 5", rank);
    b->rank = rank;
    b->update = true;
    return MI_OK;
}

int mi_spectral_bank_set_phase(mi_spectral_bank_t *b, float phase)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_spectral_bank_set_phase: NULL bank");
    b->phase = (phase < 0.0f) ? 0.0f : (phase > 1.0f) ? 1.0f : phase;
    b->update = true;
    return MI_OK;
}

int mi_spectral_bank_set_windows(mi_spectral_bank_t *b, int in_window, int out_window)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_spectral_bank_set_windows: NULL bank");
    MI_REQUIRE(in_window < MI_WINDOW_TOTAL && out_window < MI_WINDOW_TOTAL, MI_EINVAL, "mi_spectral_bank_set_windows: unknown window");
    b->wnd_in = in_window;
    b->wnd_out = out_window;
    b->update = true;
    return MI_OK;
}

int mi_spectral_bank_get(const mi_spectral_bank_t *b, uint32_t *rank, uint32_t *latency, uint32_t *remaining)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_spectral_bank_get: NULL bank");
    if (rank)      *rank = b->rank;
    if (latency)   *latency = 1u << b->rank;                                   // SpectralProcessor.h:142
    if (remaining) *remaining = (1u << (b->rank - 1)) - b->offset;             // SpectralProcessor.cpp:251-255
    return MI_OK;
}

int mi_spectral_bank_unbind(mi_spectral_bank_t *b)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_spectral_bank_unbind: NULL bank");
    b->op = MI_SPECTRAL_OP_NONE;
    b->func = nullptr;
    b->object = b->subject = nullptr;
    return MI_OK;
}

int mi_spectral_bank_bind(mi_spectral_bank_t *b, mi_spectral_func_t func, void *object, void *subject)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_spectral_bank_bind: NULL bank");
    if (func == nullptr)
        return mi_spectral_bank_unbind(b);
    if (b->d_spec == nullptr)
    {
        const size_t N = size_t(1) << b->max_rank;
        MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_spec), size_t(b->channels) * N * sizeof(float2)));
    }
    b->op = MI_SPECTRAL_OP_CALLBACK;
    b->func = func;
    b->object = object;
    b->subject = subject;
    return MI_OK;
}

int mi_spectral_bank_bind_mask(mi_spectral_bank_t *b, const float *mask, size_t mask_stride, void *stream)
{
    MI_REQUIRE(b != nullptr && mask != nullptr, MI_EINVAL, "mi_spectral_bank_bind_mask: bad argument");
    const size_t bins = (size_t(1) << (b->rank - 1)) + 1;
    MI_REQUIRE(mask_stride == 0 || mask_stride >= bins, MI_EINVAL, "mi_spectral_bank_bind_mask: stride shorter than N/2+1");
    const size_t rows = (mask_stride == 0) ? 1 : b->channels;
    if (b->d_mask == nullptr)
    {
        const size_t maxbins = (size_t(1) << (b->max_rank - 1)) + 1;
        MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_mask), size_t(b->channels) * maxbins * sizeof(float)));
    }
    MI_HIP_CHECK(hipMemcpy2DAsync(b->d_mask, bins * sizeof(float), mask, (mask_stride ? mask_stride : bins) * sizeof(float),
                                  bins * sizeof(float), rows, hipMemcpyHostToDevice, mi::as_stream(stream)));
    MI_HIP_CHECK(hipStreamSynchronize(mi::as_stream(stream)));
    b->mask_stride = (mask_stride == 0) ? 0 : bins;
    b->op = MI_SPECTRAL_OP_MASK;
    return MI_OK;
}

int mi_spectral_bank_bind_channels(mi_spectral_bank_t *b, const uint8_t *has_in, const uint8_t *has_out, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_spectral_bank_bind_channels: NULL bank");
    hipStream_t st = mi::as_stream(stream);
    if (has_in != nullptr)
    {
        if (b->d_active == nullptr)
            MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_active), b->channels));
        MI_HIP_CHECK(hipMemcpyAsync(b->d_active, has_in, b->channels, hipMemcpyHostToDevice, st));
    }
    if (has_out != nullptr)
    {
        if (b->d_has_out == nullptr)
            MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_has_out), b->channels));
        MI_HIP_CHECK(hipMemcpyAsync(b->d_has_out, has_out, b->channels, hipMemcpyHostToDevice, st));
    }
    MI_HIP_CHECK(hipStreamSynchronize(st));
    return MI_OK;
}

int mi_spectral_bank_reset(mi_spectral_bank_t *b, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_spectral_bank_reset: NULL bank");
    if (b->update)                                          // SpectralProcessor.cpp:257-266
        return MI_OK;
    const size_t N = size_t(1) << b->rank;
    hipStream_t st = mi::as_stream(stream);
    // reference clears pOutBuf + pInBuf (buf_size * 2 floats starting at pOutBuf)
    MI_HIP_CHECK(hipMemsetAsync(b->d_in, 0, size_t(b->channels) * N * sizeof(float), st));
    MI_HIP_CHECK(hipMemsetAsync(b->d_out, 0, size_t(b->channels) * N * sizeof(float), st));
    return MI_OK;
}

// `run` blocks of N samples at rank 12 with a mask shared by the channels, every buffer apart from every other: stft_wave_blocks_kernel
static int stft_wave_launch(mi_spectral_bank_t *b, const stft_blocks &tab, size_t run, size_t in_stride, size_t out_stride, hipStream_t st,
                            hipEvent_t ev0, hipEvent_t ev1)
{
    const float *wi = (b->wnd_in >= 0) ? b->d_wnd : (const float *)nullptr;
    // segments per channel: a wave per SIMD on the device (1024 waves) if the run is long enough to be cut
    int want = int((1024 + b->channels - 1) / b->channels), segs = 1;
    while (segs < STFT_WAVES && 2 * segs <= want && 2 * segs <= int(run) / 4)
        segs *= 2;
    const unsigned total = b->channels * unsigned(segs);
    if (b->mask_stride == 0)
        MI_LAUNCH(stft_wave_blocks_kernel<true>, dim3((total + STFT_WAVES - 1) / STFT_WAVES), dim3(64 * STFT_WAVES), 0, st, ev0, ev1,
                  b->d_in, b->d_out, wi, b->d_wnd_out, b->d_mask, size_t(0), b->d_tw, tab, in_stride, out_stride, int(run), int(b->channels), segs);
    else
        MI_LAUNCH(stft_wave_blocks_kernel<false>, dim3((total + STFT_WAVES - 1) / STFT_WAVES), dim3(64 * STFT_WAVES), 0, st, ev0, ev1,
                  b->d_in, b->d_out, wi, b->d_wnd_out, b->d_mask, b->mask_stride, b->d_tw, tab, in_stride, out_stride, int(run), int(b->channels), segs);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

int mi_spectral_bank_process(mi_spectral_bank_t *b, float *out, const float *in, size_t count,
                             size_t out_stride, size_t in_stride, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_spectral_bank_process: NULL bank");
    if (count == 0)
        return MI_OK;
    MI_REQUIRE(in != nullptr, MI_EINVAL, "mi_spectral_bank_process: NULL input");
    hipStream_t st = mi::as_stream(stream);
    {
        const int rc = mi::capture_touch(st, b, "spectral processor", mi::spectral_bank_positions);
        if (rc != MI_OK)
            return rc;
    }
    if (b->update)
    {
        const int r = spectral_apply_settings(b, st);
        if (r != MI_OK)
            return r;
    }
    const size_t N = size_t(1) << b->rank, frame = N >> 1;
    const bool analyze_only = (out == nullptr) && !b->eager;       // process(src, count), SpectralProcessor.cpp:201-249
    size_t done = 0;
    while (done < count)                                    // SpectralProcessor.cpp:156-198, MultiSpectralProcessor.cpp:297-392
    {
        // SpectralProcessor transforms a complete frame when the NEXT sample arrives (:159), MultiSpectralProcessor as
        // soon as the frame is complete (:324): the samples are the same, the moment the function is called is not
        if (!b->eager && b->offset >= frame)
        {
            // a whole frame follows in this call: the hop, the emission of the frame it finishes and the intake of the next
            // one in a single launch (stft_stream_kernel) instead of a hop and two strided copies
            const bool bound = (b->op == MI_SPECTRAL_OP_CALLBACK) ? (b->func != nullptr) : true;
            const bool plain = (b->op == MI_SPECTRAL_OP_NONE) || !bound, masked = bound && (b->op == MI_SPECTRAL_OP_MASK);
            const bool aligned = ((reinterpret_cast<uintptr_t>(in + done) | reinterpret_cast<uintptr_t>(out + done)) % 8 == 0) &&
                                 (in_stride % 2 == 0) && (out_stride % 2 == 0);
            // (the stream launch hands a finished frame to the caller BEFORE it takes the caller's samples at the same place:
            // a call whose output rows overlap its input rows -- in place, SpectralProcessor.cpp:188-189 copies in before it
            // copies out -- keeps the hop and the two copies)
            const uintptr_t o0 = reinterpret_cast<uintptr_t>(out), i0 = reinterpret_cast<uintptr_t>(in);
            const bool apart = out == nullptr || o0 + ((size_t(b->channels) - 1) * out_stride + count) * sizeof(float) <= i0 ||
                               i0 + ((size_t(b->channels) - 1) * in_stride + count) * sizeof(float) <= o0;
            if (out != nullptr && (plain || masked) && b->rank >= 8 && b->rank <= 13 && count - done >= frame && aligned &&
                b->d_active == nullptr && apart)
            {
                // A call of EIGHT or more whole blocks of N samples at rank 12 with a mask: a wave per channel (and segment) on the
                // wave-resident transform, the blocks being column slices of the caller's two buffers, which lie apart
                // (stft_wave_blocks_kernel: 14.6 us per block at eight blocks, 12 at 64, against the workgroup kernel's 17 - 19).
                // Shorter calls stay where they are: the kernel has 20 us of its own per launch -- its tables into LDS, a first block
                // with nothing in flight, the state written back -- and a one-block call measured 32.5 against 25 - 30 us.
                if (masked && b->rank == 12 && (count - done) % N == 0 && (count - done) / N >= 8 && !mi::compat_bits())
                {
                    while (done < count)
                    {
                        const size_t run = std::min<size_t>((count - done) / N, size_t(STFT_BLOCKS_MAX));
                        stft_blocks tab;
                        tab.per = 2;
                        for (size_t q = 0; q < run; ++q)
                        {
                            tab.src[q] = in + done + q * N;
                            tab.dst[q] = out + done + q * N;
                        }
                        hipEvent_t ev0 = nullptr, ev1 = nullptr;
                        mi::take_profile_events(&ev0, &ev1);
                        const int r = stft_wave_launch(b, tab, run, in_stride, out_stride, st, ev0, ev1);
                        if (r != MI_OK)
                            return r;
                        done += run * N;
                    }
                    b->offset = uint32_t(frame);
                    continue;
                }
                const int lh = int(b->rank) - 1;
                hipEvent_t ev0 = nullptr, ev1 = nullptr;
                mi::take_profile_events(&ev0, &ev1);
                const float *wi = (b->wnd_in >= 0) ? b->d_wnd : (const float *)nullptr;
                // every whole frame that follows in this call rides on the same launch
                const int hops = int(std::min<size_t>((count - done) / frame, size_t(1) << 20));
                #define MI_CALL(LH) \
                    if (masked) MI_LAUNCH((stft_stream_kernel<(LH < 7 ? 7 : LH > 12 ? 12 : LH), true>), dim3(b->channels), dim3(fplan<(LH < 7 ? 7 : LH > 12 ? 12 : LH)>::T), 0, st, ev0, ev1, \
                                          b->d_in, b->d_out, wi, b->d_wnd_out, b->d_mask, b->mask_stride, b->d_tw, in + done, in_stride, out + done, out_stride, hops); \
                    else        MI_LAUNCH((stft_stream_kernel<(LH < 7 ? 7 : LH > 12 ? 12 : LH), false>), dim3(b->channels), dim3(fplan<(LH < 7 ? 7 : LH > 12 ? 12 : LH)>::T), 0, st, ev0, ev1, \
                                          b->d_in, b->d_out, wi, b->d_wnd_out, (const float *)nullptr, size_t(0), b->d_tw, in + done, in_stride, out + done, out_stride, hops)
                MI_LOGH_SWITCH(lh, MI_CALL)
                #undef MI_CALL
                MI_HIP_CHECK(hipGetLastError());
                b->offset = uint32_t(frame);
                done += frame * size_t(hops);
                continue;
            }
            const int r = spectral_hop(b, st, analyze_only);
            if (r != MI_OK)
                return r;
            b->offset = 0;
        }
        const size_t n = (count - done < frame - b->offset) ? count - done : frame - b->offset;
        // MultiSpectralProcessor's timing, a bound function and a whole frame in this piece: the hand-out of the finished frame
        // and the intake of the new one ride on the hop's first launch instead of two strided copies
        // (rows that lie apart only: see above)
        const uintptr_t eo0 = reinterpret_cast<uintptr_t>(out), ei0 = reinterpret_cast<uintptr_t>(in);
        const bool rows_apart = out == nullptr || eo0 + ((size_t(b->channels) - 1) * out_stride + count) * sizeof(float) <= ei0 ||
                                ei0 + ((size_t(b->channels) - 1) * in_stride + count) * sizeof(float) <= eo0;
        if (b->eager && b->offset == 0 && n == frame && out != nullptr && b->op == MI_SPECTRAL_OP_CALLBACK && b->func != nullptr &&
            b->rank <= 14 && (reinterpret_cast<uintptr_t>(in + done) | reinterpret_cast<uintptr_t>(out + done)) % 8 == 0 &&
            in_stride % 2 == 0 && out_stride % 2 == 0 && rows_apart)
        {
            const int r = spectral_hop(b, st, false, in + done, in_stride, out + done, out_stride);
            if (r != MI_OK)
                return r;
            done += frame;                                      // offset stays 0: the frame was complete and is transformed
            continue;
        }
        if (n > 0)
        {
            MI_HIP_CHECK(hipMemcpy2DAsync(b->d_in + frame + b->offset, N * sizeof(float), in + done, in_stride * sizeof(float),
                                          n * sizeof(float), b->channels, hipMemcpyDeviceToDevice, st));
            if (out != nullptr)
                MI_HIP_CHECK(hipMemcpy2DAsync(out + done, out_stride * sizeof(float), b->d_out + b->offset, N * sizeof(float),
                                              n * sizeof(float), b->channels, hipMemcpyDeviceToDevice, st));
        }
        b->offset += uint32_t(n);
        done += n;
        if (b->eager && b->offset >= frame)
        {
            const int r = spectral_hop(b, st);
            if (r != MI_OK)
                return r;
            b->offset = 0;
        }
    }
    return MI_OK;
}

// `blocks` consecutive process() calls in one C call (SpectralProcessor.cpp:143-199 per block).  Runs of blocks of whole frames
// whose buffers lie apart go out as ONE launch (stft_stream_blocks_kernel: between two hops nothing goes through memory, whichever
// block they belong to) -- the samples and the state of the calls one by one.
int mi_spectral_bank_process_blocks(mi_spectral_bank_t *b, float *const *out, const float *const *in, size_t blocks, size_t count,
                                    size_t out_stride, size_t in_stride, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_spectral_bank_process_blocks: NULL bank");
    if (count == 0 || blocks == 0)
        return MI_OK;
    MI_REQUIRE(out != nullptr && in != nullptr, MI_EINVAL, "mi_spectral_bank_process_blocks: NULL pointer table");
    for (size_t k = 0; k < blocks; ++k)
        MI_REQUIRE(in[k] != nullptr, MI_EINVAL, "mi_spectral_bank_process_blocks: NULL input of block %zu", k);
    hipStream_t st = mi::as_stream(stream);
    auto one_by_one = [&](size_t first, size_t n) -> int {
        for (size_t k = first; k < first + n; ++k)
        {
            const int r = mi_spectral_bank_process(b, out[k], in[k], count, out_stride, in_stride, stream);
            if (r != MI_OK)
                return r;
        }
        return MI_OK;
    };
    size_t k = 0;
    while (k < blocks)
    {
        {
            const int rc = mi::capture_touch(st, b, "spectral processor", mi::spectral_bank_positions);
            if (rc != MI_OK)
                return rc;
        }
        if (b->update)
        {
            const int r = spectral_apply_settings(b, st);
            if (r != MI_OK)
                return r;
        }
        const size_t N = size_t(1) << b->rank, frame = N >> 1;
        const bool bound = (b->op == MI_SPECTRAL_OP_CALLBACK) ? (b->func != nullptr) : true;
        const bool plain = (b->op == MI_SPECTRAL_OP_NONE) || !bound, masked = bound && (b->op == MI_SPECTRAL_OP_MASK);
        const bool steady = !b->eager && b->offset == frame && (plain || masked) && b->rank >= 8 && b->rank <= 13 &&
                            (count % frame) == 0 && (in_stride % 2) == 0 && (out_stride % 2) == 0 && b->d_active == nullptr;
        size_t run = 0;
        if (steady)
        {
            const size_t ob = ((size_t(b->channels) - 1) * out_stride + count) * sizeof(float), ib = ((size_t(b->channels) - 1) * in_stride + count) * sizeof(float);
            auto overlap = [](const void *p, size_t pn, const void *q, size_t qn) -> bool {
                const uintptr_t a0 = reinterpret_cast<uintptr_t>(p), b0 = reinterpret_cast<uintptr_t>(q);
                return a0 < b0 + qn && b0 < a0 + pn;
            };
            while (k + run < blocks && run < size_t(STFT_BLOCKS_MAX))
            {
                const size_t j = k + run;
                bool ok = out[j] != nullptr && ((reinterpret_cast<uintptr_t>(out[j]) | reinterpret_cast<uintptr_t>(in[j])) % 8) == 0;
                for (size_t i = k; ok && i <= j; ++i)       // nothing the run writes is read by it (its own rows included)
                    ok = !overlap(out[i], ob, in[j], ib) && !overlap(out[j], ob, in[i], ib);
                if (!ok)
                    break;
                ++run;
            }
        }
        if (run < 2)
        {
            const int r = one_by_one(k, 1);
            if (r != MI_OK)
                return r;
            ++k;
            continue;
        }
        stft_blocks tab;
        tab.per = int(count / frame);
        for (size_t i = 0; i < run; ++i)
        {
            tab.src[i] = in[k + i];
            tab.dst[i] = out[k + i];
        }
        const int hops = int(run) * tab.per, lh = int(b->rank) - 1;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        mi::take_profile_events(&ev0, &ev1);
        const float *wi = (b->wnd_in >= 0) ? b->d_wnd : (const float *)nullptr;
        // rank 12, a fused mask, blocks of exactly one frame whose buffers all lie apart: a wave per channel and segment of the run
        // on the wave-resident transform (stft_wave_blocks_kernel: two frames per complex transform)
        bool waves = masked && b->rank == 12 && count == N && !mi::compat_bits();
        if (waves)
        {
            std::vector<std::pair<uintptr_t, uintptr_t>> iv;
            const size_t ob = ((size_t(b->channels) - 1) * out_stride + count) * sizeof(float);
            for (size_t i = 0; i < run; ++i)
                iv.emplace_back(reinterpret_cast<uintptr_t>(out[k + i]), reinterpret_cast<uintptr_t>(out[k + i]) + ob);
            std::sort(iv.begin(), iv.end());
            for (size_t i = 1; i < run && waves; ++i)
                waves = iv[i].first >= iv[i - 1].second;     // (outputs against inputs: the run was formed that way)
        }
        if (waves)
        {
            const int rw = stft_wave_launch(b, tab, run, in_stride, out_stride, st, ev0, ev1);
            if (rw != MI_OK)
                return rw;
            b->offset = uint32_t(frame);
            k += run;
            continue;
        }
        #define MI_CALL(LH) \
            if (masked) MI_LAUNCH((stft_stream_blocks_kernel<(LH < 7 ? 7 : LH > 12 ? 12 : LH), true>), dim3(b->channels), dim3(fplan<(LH < 7 ? 7 : LH > 12 ? 12 : LH)>::T), 0, st, ev0, ev1, \
                                  b->d_in, b->d_out, wi, b->d_wnd_out, b->d_mask, b->mask_stride, b->d_tw, tab, in_stride, out_stride, hops); \
            else        MI_LAUNCH((stft_stream_blocks_kernel<(LH < 7 ? 7 : LH > 12 ? 12 : LH), false>), dim3(b->channels), dim3(fplan<(LH < 7 ? 7 : LH > 12 ? 12 : LH)>::T), 0, st, ev0, ev1, \
                                  b->d_in, b->d_out, wi, b->d_wnd_out, (const float *)nullptr, size_t(0), b->d_tw, tab, in_stride, out_stride, hops)
        MI_LOGH_SWITCH(lh, MI_CALL)
        #undef MI_CALL
        MI_HIP_CHECK(hipGetLastError());
        b->offset = uint32_t(frame);
        k += run;
    }
    return MI_OK;
}

int mi_spectral_bank_set_timing(mi_spectral_bank_t *b, int eager)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_spectral_bank_set_timing: NULL bank");
    b->eager = (eager != 0);
    return MI_OK;
}

} // extern "C"

// =============================================================================================================
struct mi_analyzer_bank
{
    uint32_t    channels = 0, max_rank = 0, rank = 0;
    uint32_t    sample_rate = 0, max_sample_rate = 0, max_delay = 0;
    uint32_t    buf_size = 0, head = 0, counter = 0, period = 0, step = 0;
    uint32_t    bins_stride = 0;
    int         window = MI_WINDOW_HANN, envelope = MI_ENVELOPE_PINK_NOISE;
    float       reactivity = 0.0f, tau = 1.0f, rate = 1.0f, min_rate = 1.0f, shift = 1.0f;
    bool        active = true;
    uint32_t    reconfigure = 0x1f;
    std::vector<uint32_t> user_delay, delay;
    std::vector<uint8_t>  ch_active, ch_freeze;
    float      *d_ring = nullptr, *d_amp = nullptr, *d_data = nullptr, *d_wnd = nullptr, *d_env = nullptr;
    float2     *d_big_work = nullptr, *d_big_tmp = nullptr, *d_big_spec = nullptr;    // frames above 2^14 samples
    uint32_t   *d_delay = nullptr;
    uint8_t    *d_flags = nullptr;
    const float2 *d_tw = nullptr;
    bool        meta_dirty = true;
    bool        analysed = false;           // a strobe pass has run: vAmp / vData hold a period's results
    // per-bin reduction riding on the analysis launch (mi_analyzer_bank_process_reduce)
    // mi_analyzer_bank_process_reduce_frames: planes that keep the spectra of the frames of a call until their reductions
    // run as one launch; d_amp and d_data are always two of the planes the bank owns (these and the two it was made with)
    std::vector<float *> planes;
    float      *d_partial = nullptr;        // [slices][REDUCE_FRAMES_MAX][bins_stride]: the slices' partial sums of bin_smooth_reduce_kernel

    uint32_t max_user_delay() const
    {
        uint32_t m = 0;
        for (uint32_t d : user_delay)
            m = (d > m) ? d : m;
        return m;
    }
};

namespace
{
    enum { R_ENVELOPE = 1, R_WINDOW = 2, R_ANALYSIS = 4, R_TAU = 8, R_COUNTERS = 16, R_ALL = 31 };

    int analyzer_reconfigure(mi_analyzer_bank *b, hipStream_t st)           // Analyzer.cpp:251-297
    {
        if (!b->reconfigure)
            return MI_OK;
        const size_t fft_size = size_t(1) << b->rank, csize = (fft_size >> 1) + 1;
        const size_t fft_period = size_t(float(b->sample_rate) / b->rate);
        b->step   = uint32_t(fft_period / b->channels);
        b->period = b->step * b->channels;
        MI_REQUIRE(b->step > 0, MI_EINVAL,
                   "analyzer: refresh rate %.3f Hz needs at least one sample per channel and period (sample_rate / rate >= channels); "
                   "the reference divides by zero here (Analyzer.cpp:258-260,315)", double(b->rate));
        if (b->reconfigure & R_ENVELOPE)
        {
            std::vector<float> env(csize);
            mi::make_reverse_noise_lin(env.data(), 0.0f, b->sample_rate * 0.5f, 100.0f, csize, b->envelope);
            const float k = b->shift / fft_size;
            for (float &v : env)
                v *= k;
            MI_HIP_CHECK(hipMemcpyAsync(b->d_env, env.data(), csize * sizeof(float), hipMemcpyHostToDevice, st));
            MI_HIP_CHECK(hipStreamSynchronize(st));
        }
        if (b->reconfigure & R_ANALYSIS)
        {
            MI_HIP_CHECK(hipMemsetAsync(b->d_amp, 0, size_t(b->channels) * b->bins_stride * sizeof(float), st));
            MI_HIP_CHECK(hipMemsetAsync(b->d_data, 0, size_t(b->channels) * b->bins_stride * sizeof(float), st));
        }
        if (b->reconfigure & R_WINDOW)
        {
            std::vector<float> w(fft_size);
            mi::make_window(w.data(), fft_size, b->window);
            MI_HIP_CHECK(hipMemcpyAsync(b->d_wnd, w.data(), fft_size * sizeof(float), hipMemcpyHostToDevice, st));
            MI_HIP_CHECK(hipStreamSynchronize(st));
        }
        if (b->reconfigure & R_TAU)
            b->tau = 1.0f - expf(logf(1.0f - float(M_SQRT1_2)) / (b->reactivity * b->rate));   // seconds_to_samples(fRate, fReactivity)
        if (b->reconfigure & R_COUNTERS)
        {
            for (uint32_t i = 0; i < b->channels; ++i)
                b->delay[i] = i * b->step;
            b->meta_dirty = true;
        }
        b->reconfigure = 0;
        return MI_OK;
    }

    // One analysis pass for all channels at the strobe instant (the reference staggers the
    // channels over the period but every channel reads the window that ends at the strobe).
    // One launch per strobe: the analysis of every channel plus the ingest of the `n` samples that follow the strobe.
    // the analysis of channels [first, channels) with the delays and flags on the device, then the ingest of `n` samples
    // (strobe pass only): one fused launch for frames up to 2^14 samples, the four-step path above that
    int analyzer_launch(mi_analyzer_bank *b, hipStream_t st, uint32_t first, const float *in, size_t in_stride, uint32_t n,
                        bool zero, hipEvent_t ev0, hipEvent_t ev1)
    {
        const uint32_t count = b->channels - first;
        float *ring = b->d_ring + size_t(first) * b->buf_size;
        const size_t row = size_t(first) * b->bins_stride;
        if (b->rank <= 14)
        {
            #define MI_CALL(LH) MI_LAUNCH((analyzer_kernel<LH>), dim3(count), dim3(fplan<LH>::T), 0, st, ev0, ev1, \
                ring, b->buf_size, b->head, b->d_delay + first, b->d_flags + first, b->d_wnd, b->d_data + row, b->d_amp + row, \
                b->bins_stride, b->tau, b->d_tw, in, in_stride, n, zero ? 1 : 0)
            MI_LOGH_SWITCH(int(b->rank) - 1, MI_CALL)
            #undef MI_CALL
            MI_HIP_CHECK(hipGetLastError());
            return MI_OK;
        }
        const uint32_t N = 1u << b->rank;
        hipLaunchKernelGGL(big_an_gather_kernel, dim3((N + 255) / 256, count), dim3(256), 0, st, b->d_big_work, ring, b->buf_size,
                           b->head, b->d_delay + first, b->d_flags + first, b->d_wnd, N);
        MI_HIP_CHECK(hipGetLastError());
        const int r = big_fft_run<false>(b->d_big_spec, b->d_big_work, b->d_big_tmp, b->rank, count, b->d_tw, st);
        if (r != MI_OK)
            return r;
        hipLaunchKernelGGL(big_an_mag_kernel, dim3((N / 2 + 1 + 255) / 256, count), dim3(256), 0, st, b->d_big_spec, b->d_data + row,
                           b->d_amp + row, b->bins_stride, b->tau, b->d_flags + first, N);
        MI_HIP_CHECK(hipGetLastError());
        if (n > 0)
        {
            hipLaunchKernelGGL(an_ingest_kernel, dim3((n + 255) / 256, count), dim3(256), 0, st, ring, b->buf_size, b->head, in,
                               in_stride, n, zero ? 1 : 0);
            MI_HIP_CHECK(hipGetLastError());
        }
        return MI_OK;
    }

    int analyzer_strobe(mi_analyzer_bank *b, hipStream_t st, const float *in, size_t in_stride, uint32_t n, bool zero)
    {
        if (b->meta_dirty)
        {
            std::vector<uint32_t> d(b->channels);
            std::vector<uint8_t> f(b->channels);
            for (uint32_t i = 0; i < b->channels; ++i)
            {
                // the reference analyses channel i `delay[i]` samples after the strobe and looks back that much
                // further, so relative to the strobe only the user delay remains
                d[i] = b->user_delay[i];
                f[i] = uint8_t(((b->active && b->ch_active[i]) ? 1 : 0) | (b->ch_freeze[i] ? 2 : 0));
            }
            MI_HIP_CHECK(hipMemcpyAsync(b->d_delay, d.data(), d.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            MI_HIP_CHECK(hipMemcpyAsync(b->d_flags, f.data(), f.size(), hipMemcpyHostToDevice, st));
            MI_HIP_CHECK(hipStreamSynchronize(st));
            b->meta_dirty = false;
        }
        // Analyzer.cpp:321-326 (vData <- vAmp at the strobe): the buffers swap roles, the kernel reads the old vAmp
        // (now vData) and writes the new one
        std::swap(b->d_amp, b->d_data);
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        mi::take_profile_events(&ev0, &ev1);
        const int rl = analyzer_launch(b, st, 0, in, in_stride, n, zero, ev0, ev1);
        if (rl != MI_OK)
            return rl;
        b->analysed = true;
        return MI_OK;
    }

    // Settings changed in the middle of a period.  The reference analyses channel i when the period counter reaches
    // i * nStep, with the settings of that moment, looking back to the strobe (Analyzer.cpp:314-366); the strobe pass
    // above has already analysed every channel with the settings of the strobe.  The channels whose turn is still to
    // come are analysed again with the settings now in force: same window of samples (it ends at the strobe), same
    // previous spectrum (the published copy, vData), result over the strobe pass's.
    int analyzer_redo(mi_analyzer_bank *b, hipStream_t st)
    {
        const uint32_t first = (b->counter + b->step - 1) / b->step;
        if (first >= b->channels)
            return MI_OK;
        std::vector<uint32_t> d(b->channels);
        std::vector<uint8_t> f(b->channels);
        for (uint32_t i = 0; i < b->channels; ++i)
        {
            d[i] = b->counter + b->user_delay[i];               // back to the strobe, then the user delay
            f[i] = uint8_t(((b->active && b->ch_active[i]) ? 1 : 0) | (b->ch_freeze[i] ? 2 : 0));
        }
        MI_HIP_CHECK(hipMemcpyAsync(b->d_delay, d.data(), d.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        MI_HIP_CHECK(hipMemcpyAsync(b->d_flags, f.data(), f.size(), hipMemcpyHostToDevice, st));
        MI_HIP_CHECK(hipStreamSynchronize(st));
        b->meta_dirty = true;                                   // the next strobe wants its own delays back
        const int rl = analyzer_launch(b, st, first, nullptr, 0, 0, false, nullptr, nullptr);
        if (rl != MI_OK)
            return rl;
        MI_HIP_CHECK(hipGetLastError());
        return MI_OK;
    }
} // namespace

extern "C" {

int mi_analyzer_bank_create(mi_analyzer_bank_t **bank, uint32_t channels, uint32_t max_rank, uint32_t max_sample_rate,
                            float min_rate, uint32_t max_delay)
{
    MI_REQUIRE(bank != nullptr, MI_EINVAL, "mi_analyzer_bank_create: NULL result pointer");
    *bank = nullptr;
    MI_REQUIRE(channels > 0 && max_sample_rate > 0 && min_rate > 0.0f, MI_EINVAL, "mi_analyzer_bank_create: bad argument");
    MI_REQUIRE(max_rank >= 5 && max_rank <= uint32_t(BIG_MAX_RANK), MI_EINVAL,
               "mi_analyzer_bank_create: max_rank %u outside the supported 5..%d", max_rank, BIG_MAX_RANK);
    MI_REQUIRE(mi_dspu_device_count() > 0, MI_ENODEV, "no HIP device available (there is no CPU fallback)");
    mi_analyzer_bank *b = new (std::nothrow) mi_analyzer_bank();
    MI_REQUIRE(b != nullptr, MI_ENOMEM, "mi_analyzer_bank_create: out of host memory");
    b->channels = channels;
    b->max_rank = b->rank = max_rank;
    b->max_sample_rate = max_sample_rate;
    b->max_delay = max_delay;
    b->min_rate = float(uint32_t(min_rate));                // Analyzer.cpp:120: fMinRate = uint32_t(min_rate)
    const size_t fft_items = size_t(1) << max_rank;
    // Analyzer.cpp:90-96: ring length = fft + 2*sr/min_rate + max_delay + DEFAULT_ALIGN, aligned to DEFAULT_ALIGN (0x40)
    size_t bs = fft_items + size_t(float(max_sample_rate * 2) / min_rate) + max_delay + 0x40;
    bs = (bs + 0x3f) & ~size_t(0x3f);
    b->buf_size = uint32_t(bs);
    b->bins_stride = uint32_t((((fft_items >> 1) + 1) + 31) & ~size_t(31));    // rows start on 128-byte lines (see analyzer_kernel)
    b->user_delay.assign(channels, 0);
    b->delay.assign(channels, 0);
    b->ch_active.assign(channels, 1);
    b->ch_freeze.assign(channels, 0);
    int twn = 0;
    int r = mi::fft_twiddles(&b->d_tw, &twn);
    hipError_t e = hipSuccess;
    if (r == MI_OK)
    {
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_ring), size_t(channels) * bs * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_amp), size_t(channels) * b->bins_stride * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_data), size_t(channels) * b->bins_stride * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_wnd), fft_items * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_env), b->bins_stride * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_delay), channels * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_flags), channels);
        if (max_rank > 14)                                  // frames that go through global memory: three complex frames per channel
        {
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_big_work), size_t(channels) * fft_items * sizeof(float2));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_big_tmp), size_t(channels) * fft_items * sizeof(float2));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_big_spec), size_t(channels) * fft_items * sizeof(float2));
        }
        if (e == hipSuccess) e = hipMemset(b->d_ring, 0, size_t(channels) * bs * sizeof(float));
        if (e == hipSuccess) e = hipMemset(b->d_amp, 0, size_t(channels) * b->bins_stride * sizeof(float));
        if (e == hipSuccess) e = hipMemset(b->d_data, 0, size_t(channels) * b->bins_stride * sizeof(float));
        if (e != hipSuccess)
            r = mi::fail(e == hipErrorOutOfMemory ? MI_ENOMEM : MI_EHIP, "mi_analyzer_bank_create: %s", hipGetErrorString(e));
    }
    if (r != MI_OK)
    {
        mi_analyzer_bank_destroy(b);
        return r;
    }
    *bank = b;
    return MI_OK;
}

int mi_analyzer_bank_destroy(mi_analyzer_bank_t *b)
{
    if (b == nullptr)
        return MI_OK;
    if (b->planes.empty())
    {
        (void)hipFree(b->d_amp);
        (void)hipFree(b->d_data);
    }
    for (float *p : b->planes)                              // (d_amp and d_data are among them)
        (void)hipFree(p);
    (void)hipFree(b->d_ring); (void)hipFree(b->d_wnd);
    (void)hipFree(b->d_env); (void)hipFree(b->d_delay); (void)hipFree(b->d_flags);
    (void)hipFree(b->d_big_work); (void)hipFree(b->d_big_tmp); (void)hipFree(b->d_big_spec);
    (void)hipFree(b->d_partial);
    delete b;
    return MI_OK;
}

int mi_analyzer_bank_reset(mi_analyzer_bank_t *b)                         // Analyzer::reset(), util/Analyzer.h:296
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_analyzer_bank_reset: NULL bank");
    b->reconfigure |= R_ANALYSIS;                           // the spectra are cleared at the next reconfigure (Analyzer.cpp:274-281)
    return MI_OK;
}

int mi_analyzer_bank_configure(mi_analyzer_bank_t *b, int what, double value)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_analyzer_bank_configure: NULL bank");
    switch (what)                                           // setters of Analyzer.cpp:154-231
    {
        case MI_ANALYZER_SAMPLE_RATE:
        {
            uint32_t sr = uint32_t(value);
            sr = (sr < b->max_sample_rate) ? sr : b->max_sample_rate;
            if (sr != b->sample_rate) { b->sample_rate = sr; b->reconfigure |= R_ALL; }
            break;
        }
        case MI_ANALYZER_RATE:
        {
            const float rate = (float(value) > b->min_rate) ? float(value) : b->min_rate;
            if (rate != b->rate) { b->rate = rate; b->reconfigure |= R_COUNTERS; }
            break;
        }
        case MI_ANALYZER_WINDOW:
            if (int(value) != b->window) { b->window = int(value); b->reconfigure |= R_WINDOW; }
            break;
        case MI_ANALYZER_ENVELOPE:
            if (int(value) != b->envelope) { b->envelope = int(value); b->reconfigure |= R_ENVELOPE; }
            break;
        case MI_ANALYZER_SHIFT:
            if (float(value) != b->shift) { b->shift = float(value); b->reconfigure |= R_ENVELOPE; }
            break;
        case MI_ANALYZER_REACTIVITY:
            if (float(value) != b->reactivity) { b->reactivity = float(value); b->reconfigure |= R_TAU; }
            break;
        case MI_ANALYZER_RANK:
        {
            const uint32_t rank = uint32_t(value);
            MI_REQUIRE(rank >= 2 && rank <= b->max_rank, MI_EINVAL, "analyzer: rank %u out of range", rank);   // Analyzer.cpp:204-205
            MI_REQUIRE(rank >= 5, MI_EINVAL, "analyzer: rank %u below the supported minimum 5", rank);
            if (rank != b->rank) { b->rank = rank; b->reconfigure |= R_ALL; }
            break;
        }
        case MI_ANALYZER_ACTIVE:
            b->active = (value != 0.0);
            b->meta_dirty = true;
            break;
        default:
            return mi::fail(MI_EINVAL, "mi_analyzer_bank_configure: unknown setting %d", what);
    }
    return MI_OK;
}

int mi_analyzer_bank_channel(mi_analyzer_bank_t *b, uint32_t channel, int what, uint32_t value)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_analyzer_bank_channel: NULL bank");
    MI_REQUIRE(channel < b->channels, MI_EINVAL, "mi_analyzer_bank_channel: channel %u out of range", channel);
    switch (what)                                           // Analyzer.cpp:213-249
    {
        case MI_ANALYZER_CH_FREEZE: b->ch_freeze[channel] = uint8_t(value != 0); break;
        case MI_ANALYZER_CH_ENABLE:
            if (b->ch_active[channel] != uint8_t(value != 0)) { b->ch_active[channel] = uint8_t(value != 0); b->reconfigure |= R_COUNTERS; }
            break;
        case MI_ANALYZER_CH_DELAY:
            MI_REQUIRE(value <= b->max_delay, MI_EINVAL, "mi_analyzer_bank_channel: delay %u above the maximum", value);
            b->user_delay[channel] = value;
            break;
        default:
            return mi::fail(MI_EINVAL, "mi_analyzer_bank_channel: unknown setting %d", what);
    }
    b->meta_dirty = true;
    return MI_OK;
}

// what the launches of a call take by value from the host: ring head, position in the period, which of the two spectrum
// buffers is vAmp
static uint64_t analyzer_bank_positions(const void *bank)
{
    const mi_analyzer_bank *b = static_cast<const mi_analyzer_bank *>(bank);
    uint64_t h = mi::position_mix(b->head, b->counter);
    h = mi::position_mix(h, uint64_t(reinterpret_cast<uintptr_t>(b->d_amp)));
    return mi::position_mix(h, (uint64_t(b->analysed) << 1) | uint64_t(b->reconfigure != 0 || b->meta_dirty));
}

int mi_analyzer_bank_process(mi_analyzer_bank_t *b, const float *in, size_t samples, size_t in_stride, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_analyzer_bank_process: NULL bank");
    hipStream_t st = mi::as_stream(stream);
    {
        const int rc = mi::capture_touch(st, b, "analyzer", analyzer_bank_positions);
        if (rc != MI_OK)
            return rc;
    }
    const bool changed = (b->reconfigure != 0) || b->meta_dirty;
    int r = analyzer_reconfigure(b, st);
    if (r != MI_OK)
        return r;
    if (changed && b->counter > 0 && b->analysed)
    {
        r = analyzer_redo(b, st);
        if (r != MI_OK)
            return r;
    }
    size_t offset = 0;
    while (offset < samples)                                // Analyzer.cpp:309-408
    {
        // run to the next strobe (channel analyses inside the period are folded into the strobe pass)
        size_t n = b->period - b->counter;
        n = (samples - offset < n) ? samples - offset : n;
        // the window of a strobe ends `delay` samples before head and is N long: the n cells behind head are free
        const bool fits = size_t(b->buf_size) >= (size_t(1) << b->rank) + b->max_user_delay() + n;
        if (b->counter == 0 && fits)
        {
            r = analyzer_strobe(b, st, (in != nullptr) ? in + offset : nullptr, in_stride, uint32_t(n), in == nullptr);
            if (r != MI_OK)
                return r;
            b->head = uint32_t((b->head + n) % b->buf_size);
        }
        else
        {
            if (b->counter == 0)
            {
                r = analyzer_strobe(b, st, nullptr, 0, 0, false);
                if (r != MI_OK)
                    return r;
            }
            size_t left = n, src = offset;
            while (left > 0)                                // ring ingest with wrap (Analyzer.cpp:371-398)
            {
                const size_t piece = (left < b->buf_size - b->head) ? left : b->buf_size - b->head;
                if (in != nullptr)
                    MI_HIP_CHECK(hipMemcpy2DAsync(b->d_ring + b->head, size_t(b->buf_size) * sizeof(float), in + src,
                                                  in_stride * sizeof(float), piece * sizeof(float), b->channels,
                                                  hipMemcpyDeviceToDevice, st));
                else
                    MI_HIP_CHECK(hipMemset2DAsync(b->d_ring + b->head, size_t(b->buf_size) * sizeof(float), 0,
                                                  piece * sizeof(float), b->channels, st));
                b->head = uint32_t((b->head + piece) % b->buf_size);
                left -= piece;
                src += piece;
            }
        }
        offset += n;
        b->counter += uint32_t(n);
        if (b->counter >= b->period)
            b->counter -= b->period;
    }
    return MI_OK;
}

int mi_analyzer_bank_get_spectrum(mi_analyzer_bank_t *b, float *out, size_t out_stride, const uint32_t *idx,
                                  uint32_t count, void *stream)
{
    MI_REQUIRE(b != nullptr && out != nullptr && idx != nullptr, MI_EINVAL, "mi_analyzer_bank_get_spectrum: bad argument");
    hipStream_t st = mi::as_stream(stream);
    // the reference reads vData and vEnvelope as they are (Analyzer.cpp:443-456): pending settings wait for process();
    // only a bank that has never been configured builds its envelope here
    if (b->period == 0)
    {
        const int r = analyzer_reconfigure(b, st);
        if (r != MI_OK)
            return r;
    }
    hipLaunchKernelGGL(spectrum_gather_kernel, dim3((count + 255) / 256, b->channels), dim3(256), 0, st,
                       out, out_stride, b->d_data, b->bins_stride, b->d_env, idx, count);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

int mi_analyzer_bank_reduce_bins(mi_analyzer_bank_t *b, float *out, int with_envelope, void *stream)
{
    MI_REQUIRE(b != nullptr && out != nullptr, MI_EINVAL, "mi_analyzer_bank_reduce_bins: bad argument");
    hipStream_t st = mi::as_stream(stream);
    const uint32_t bins = (1u << (b->rank - 1)) + 1;
    // blocks of 16 channels up to 16 384 channels; beyond that the block grows so that the block sums still fit the LDS
    uint32_t block = REDUCE_BLOCK;
    while ((b->channels + block - 1) / block > REDUCE_MAX_BLOCKS)
        block *= 2;
    hipLaunchKernelGGL(bin_reduce_kernel, dim3((bins + REDUCE_BINS - 1) / REDUCE_BINS), dim3(64 * REDUCE_WAVES), 0, st,
                       out, b->d_amp, b->bins_stride, b->channels, bins, with_envelope ? b->d_env : nullptr, block);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

int mi_analyzer_bank_process_reduce(mi_analyzer_bank_t *b, const float *in, size_t samples, size_t in_stride,
                                    float *out, int with_envelope, void *stream)
{
    MI_REQUIRE(b != nullptr && out != nullptr, MI_EINVAL, "mi_analyzer_bank_process_reduce: bad argument");
    const int r = mi_analyzer_bank_process(b, in, samples, in_stride, stream);
    if (r != MI_OK)
        return r;
    return mi_analyzer_bank_reduce_bins(b, out, with_envelope, stream);
}

int mi_analyzer_bank_process_reduce_frames(mi_analyzer_bank_t *b, const float *const *in, size_t frames, size_t samples, size_t in_stride,
                                           float *out, size_t out_stride, int with_envelope, void *stream)
{
    MI_REQUIRE(b != nullptr && out != nullptr && in != nullptr, MI_EINVAL, "mi_analyzer_bank_process_reduce_frames: bad argument");
    const uint32_t bins = (1u << (b->rank - 1)) + 1;
    MI_REQUIRE(out_stride >= bins, MI_EINVAL, "mi_analyzer_bank_process_reduce_frames: out_stride shorter than a row of bins");
    hipStream_t st = mi::as_stream(stream);
    // (test knobs, read once per call -- not once per turn of the loop: ADVICE r04)
    // (test hook: the run as a launch per strobe + the planes' reductions in one launch -- what runs with user delays or another
    // hop take anyway)
    const bool knob_strobe_per_launch = mi::test_path("analyzer_strobe_per_launch");
    size_t f = 0;
    while (f < frames)
    {
        // pending settings take effect at the first call as always; the batched form needs every call to be exactly one
        // strobe: a period per call, starting at a strobe, transforms of up to 2^14 points
        int r = analyzer_reconfigure(b, st);
        if (r != MI_OK)
            return r;
        const bool batch = b->rank <= 14 && samples == size_t(b->period) && b->counter == 0 && !b->meta_dirty && frames - f >= 2 &&
                           size_t(b->buf_size) >= (size_t(1) << b->rank) + b->max_user_delay() + samples;
        if (!batch)
        {
            r = mi_analyzer_bank_process_reduce(b, in[f], samples, in_stride, out + f * out_stride, with_envelope, stream);
            if (r != MI_OK)
                return r;
            ++f;
            continue;
        }
        const size_t plane_bytes = size_t(b->channels) * b->bins_stride * sizeof(float);
        uint32_t block = REDUCE_BLOCK;
        while ((b->channels + block - 1) / block > REDUCE_MAX_BLOCKS)
            block *= 2;
        const uint32_t nblocks = (b->channels + block - 1) / block, slices = (nblocks + SMR_BLOCKS - 1) / SMR_BLOCKS;
        if (b->planes.empty())
        {
            // (all the spare planes or none: a list that is short would leave NULL rows for a later call to write through --
            // ADVICE r04)  REDUCE_FRAMES_MAX spare planes (a run's raw magnitudes) next to vAmp and vData, and the slices' partial
            // sums of the smoothing reduction
            std::vector<float *> made;
            made.push_back(b->d_amp);
            made.push_back(b->d_data);
            hipError_t e = hipSuccess;
            for (uint32_t k = 0; k < REDUCE_FRAMES_MAX && e == hipSuccess; ++k)
            {
                float *p = nullptr;
                e = hipMalloc(reinterpret_cast<void **>(&p), plane_bytes);
                if (e == hipSuccess)
                    made.push_back(p);
            }
            float *part = nullptr;
            if (e == hipSuccess)
                e = hipMalloc(reinterpret_cast<void **>(&part), size_t(slices) * REDUCE_FRAMES_MAX * b->bins_stride * sizeof(float));

            if (e != hipSuccess)
            {
                for (size_t j = 2; j < made.size(); ++j)
                    (void)hipFree(made[j]);
                (void)hipFree(part);
                MI_HIP_CHECK(e);
            }
            b->d_partial = part;

            b->planes.swap(made);
        }
        size_t cnt = (frames - f < size_t(REDUCE_FRAMES_MAX)) ? frames - f : size_t(REDUCE_FRAMES_MAX);
        const float *const env = with_envelope ? b->d_env : nullptr;
        // the strobes themselves as ONE launch (analyzer_frames_*_kernel) where the hop is half a frame, nothing is delayed and
        // every block is there; otherwise a launch per strobe
        const int lh = int(b->rank) - 1;
        bool one_launch = lh >= 9 && lh <= 12 && 2 * size_t(b->period) == (size_t(1) << b->rank) && b->max_user_delay() == 0 &&
                          !knob_strobe_per_launch;
        for (size_t k = 0; one_launch && k < cnt; ++k)
            one_launch = in[f + k] != nullptr;
        if (one_launch)
        {
            // the launch's units take the run's hops in no particular order: the ring must hold the first strobe's frame and all
            // the run's hops side by side (a shorter ring: a shorter run)
            const size_t room = (size_t(b->buf_size) - (size_t(1) << b->rank)) / samples;
            cnt = (cnt < room) ? cnt : room;
            one_launch = cnt >= 2;
            if (!one_launch)
                cnt = (frames - f < size_t(REDUCE_FRAMES_MAX)) ? frames - f : size_t(REDUCE_FRAMES_MAX);
        }
        if (one_launch)
        {
            const int rc = mi::capture_touch(st, b, "analyzer", analyzer_bank_positions);
            if (rc != MI_OK)
                return rc;
            // The launch leaves strobe k's RAW magnitudes in spare plane k; bin_smooth_reduce_kernel walks the smoothing over
            // them from vAmp (d_amp, in place: vAmp after the run) and leaves vAmp as of the last strobe but one -- the copy that
            // strobe published (Analyzer.cpp:321-326) -- in d_data: the planes stay where the run found them (a captured run of
            // batches repeats with the ring alone).
            an_frames_args fa;
            smooth_planes sp;
            fa.frames = int(cnt);
            bool aligned = (in_stride % 2) == 0;
            for (size_t k = 0; k < size_t(AN_FRAMES_MAX); ++k)
            {
                fa.in[k] = (k < cnt) ? in[f + k] : nullptr;
                sp.raw[k] = fa.rows[k] = b->planes[2 + k];
                aligned = aligned && (k >= cnt || (reinterpret_cast<uintptr_t>(in[f + k]) % 8) == 0);
            }
            hipEvent_t ev0 = nullptr, ev1 = nullptr;
            mi::take_profile_events(&ev0, &ev1);
            if (lh == 11)
            {
                // rank 12: a wave per pair of strobes on the wave-resident transform; one workgroup per CU walks the channels
                // (its tables are filled once, a wave's next unit is in flight underneath its transform; any grid is a correct one)
                static const uint32_t cus = []() {
                    int dev = 0, n = 0;
                    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
                        n = 256;
                    return uint32_t(n);
                }();
                bool wide = (in_stride % 4) == 0;
                for (size_t k = 0; k < cnt; ++k)
                    wide = wide && (reinterpret_cast<uintptr_t>(in[f + k]) % 16) == 0;
                MI_LAUNCH(analyzer_frames_wave_kernel, dim3(b->channels < cus ? b->channels : cus), dim3(64 * ANW_WAVES), 0, st, ev0, ev1,
                          fa, in_stride, wide ? 1 : 0, b->d_ring, b->buf_size, b->head, b->d_flags, b->d_wnd, b->bins_stride, b->d_tw, int(b->channels));
            }
            else
            {
                #define MI_CALL(LH) MI_LAUNCH((analyzer_frames_kernel<LH>), dim3(b->channels), dim3(fplan<LH>::T), 0, st, ev0, ev1, \
                    fa, in_stride, aligned, b->d_ring, b->buf_size, b->head, b->d_flags, b->d_wnd, b->bins_stride, b->d_tw)
                switch (lh)
                {
                    case 9:  { MI_CALL(9);  break; }
                    case 10: { MI_CALL(10); break; }
                    default: { MI_CALL(12); break; }
                }
                #undef MI_CALL
            }
            MI_HIP_CHECK(hipGetLastError());
            if (cnt == size_t(AN_FRAMES_MAX))
                hipLaunchKernelGGL((bin_smooth_reduce_kernel<true>), dim3((bins + SMR_BINS - 1) / SMR_BINS, slices), dim3(64 * SMR_BLOCKS), 0, st,
                                   b->d_partial, sp, int(cnt), b->d_amp, b->d_data, b->bins_stride, b->channels, bins, b->d_flags, b->tau, block);
            else
                hipLaunchKernelGGL((bin_smooth_reduce_kernel<false>), dim3((bins + SMR_BINS - 1) / SMR_BINS, slices), dim3(64 * SMR_BLOCKS), 0, st,
                                   b->d_partial, sp, int(cnt), b->d_amp, b->d_data, b->bins_stride, b->channels, bins, b->d_flags, b->tau, block);
            MI_HIP_CHECK(hipGetLastError());
            hipLaunchKernelGGL(bin_combine_kernel, dim3((bins + 255) / 256, uint32_t(cnt)), dim3(256), 0, st,
                               out + f * out_stride, out_stride, b->d_partial, slices, b->bins_stride, bins, env);
            MI_HIP_CHECK(hipGetLastError());
            b->head = uint32_t((uint64_t(b->head) + uint64_t(cnt) * samples) % b->buf_size);
            b->analysed = true;
            f += cnt;
            continue;
        }
        // A launch per strobe: frame k's analysis reads the spectrum of the frame before (vAmp) and leaves its own in a plane that
        // nothing reads any more.  The last two frames of the batch take the planes the bank came in with -- the last one the
        // plane that was vAmp (only the batch's first frame reads it), the one before it the published copy's (vData: the strobe
        // replaces it, Analyzer.cpp:321-326) -- so that a batch leaves vAmp and vData where it found them; the frames before
        // those two go through the spare planes.  The reductions of the batch's strobes then run as ONE launch.
        float *const amp0 = b->d_amp, *const data0 = b->d_data;
        std::vector<float *> spare(cnt);
        for (size_t k = 0; k + 2 < cnt; ++k)
            spare[k] = b->planes[2 + k];
        spare[cnt - 2] = data0;
        spare[cnt - 1] = amp0;
        reduce_planes rp;
        for (size_t k = 0; k < cnt; ++k)
        {
            b->d_data = spare[k];                           // analyzer_strobe swaps: d_amp <- this plane, d_data <- the spectrum so far
            r = mi_analyzer_bank_process(b, in[f + k], samples, in_stride, stream);
            if (r != MI_OK)
                return r;
            rp.rows[k] = b->d_amp;
        }
        if (nblocks <= BINS_32_BLOCKS)
            hipLaunchKernelGGL((bin_reduce_frames_kernel<32>), dim3((bins + 31) / 32, uint32_t(cnt)), dim3(64 * REDUCE_WAVES), 0, st,
                               out + f * out_stride, out_stride, rp, b->bins_stride, b->channels, bins, env, block);
        else
            hipLaunchKernelGGL((bin_reduce_frames_kernel<16>), dim3((bins + 15) / 16, uint32_t(cnt)), dim3(64 * REDUCE_WAVES), 0, st,
                               out + f * out_stride, out_stride, rp, b->bins_stride, b->channels, bins, env, block);
        MI_HIP_CHECK(hipGetLastError());
        f += cnt;
    }
    return MI_OK;
}

int mi_analyzer_bank_info(const mi_analyzer_bank_t *b, uint32_t *rank, uint32_t *bins, uint32_t *period, uint32_t *step)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_analyzer_bank_info: NULL bank");
    if (rank)   *rank = b->rank;
    if (bins)   *bins = (1u << (b->rank - 1)) + 1;
    if (period) *period = b->period;
    if (step)   *step = b->step;
    return MI_OK;
}

} // extern "C"
