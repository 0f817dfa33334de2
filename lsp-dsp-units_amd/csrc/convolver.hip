// Partitioned FFT convolver bank for gfx950: the GPU side of lsp::dspu::Convolver
// (reference: src/main/util/Convolver.cpp:77-215 init, :217-313 process).
//
// The reference is a zero-latency NON-uniform partitioned convolver (128-tap head, doubling levels, equal
// tail blocks, 15 inverse transforms per frame at rank 13, a load-spreading schedule) because a CPU thread
// has to bound the work of every 128-sample step.  On the GPU the same input/output contract -- exact linear
// convolution, zero added latency, arbitrary call sizes -- is met with a UNIFORM partition of B = frame
// samples and a frequency-domain delay line:
//
//   state per channel:  acc[2B]   contributions of everything received so far to the current and the next frame
//                       ring[P-1] spectra of the last P-1 complete input frames
//                       H[P]      spectra of the IR partitions (2B-point real transforms, packed to B complex)
//
//                       Yt        pending spectrum: sum_{p>=1} H_p * X_(c+1-p), the tail owed to the next frame
//
//   whole frame (off == 0, >= B samples):                                       [conv_frame_kernel]
//       X = FFT(frame);  y = IFFT(H0 * X + Yt);  out = acc[0:B] + y[0:B];  acc = shift(acc) + y[B:2B]
//   then, off the output's critical path, the tail owed to the NEXT frame:      [conv_mac_kernel]
//       Yt = sum_{p>=1} H_p * X_(c+1-p)                  <- the HBM-bound part: streams H and the ring once
//   partial call (anything else): a pending Yt is first folded into acc         [conv_tail_kernel]
//     and the head partition is applied directly in the time domain
//       acc[off+i] += sum_j x[j] h[i-j]  (zero latency for any chunking)        [conv_direct_kernel]
//     and when the frame completes its spectrum enters the ring                 [conv_commit_kernel]
//
// Spectra use the packed real-FFT image of fft_device.h (B complex per 2B-point real transform).
#include "mi_common.h"
#include "fft_device.h"
#include "fft16.h"
#include "fft_wave.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>

namespace
{
    using namespace mi_fft;
    // radix-16 core (fft16.h) for 1024 .. 8192-point transforms, radix-8 core (fft_device.h) below that
    template <int L_> using fplan = mi_fft16::fsel<L_>;

    constexpr int LOGM_MIN = 7, LOGM_MAX = 12;

#ifdef MI_CONV_PROBE
    // phase timestamps of thread 0 of every workgroup (tests/experiments/conv_frame_probe.hip)
    __device__ unsigned long long g_conv_probe[1024 * 8];
    #define MI_CPROBE(slot) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) \
        g_conv_probe[blockIdx.x * 8 + (slot)] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
    // conv_small_kernel: lane 0 of each of a workgroup's two waves (tests/experiments/conv_small_probe.hip)
    __device__ unsigned long long g_small_probe[1024 * 2 * 8];
    #define MI_SPROBE(slot) do { __builtin_amdgcn_sched_barrier(0); if ((threadIdx.x & 63) == 0) \
        g_small_probe[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 8 + (slot)] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
    // conv_batch_tail_kernel: thread 0 of every workgroup (tests/experiments/conv_tail_probe.hip)
    __device__ unsigned long long g_tail_probe[4096 * 8];
    #define MI_TPROBE(slot) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) \
        g_tail_probe[((blockIdx.y * gridDim.x + blockIdx.x) & 4095) * 8 + (slot)] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
    #define MI_CPROBE(slot) do { } while (0)
    #define MI_SPROBE(slot) do { } while (0)
    #define MI_TPROBE(slot) do { } while (0)
#endif

    // ---- forward transform of a real block of `valid` samples zero-padded to 2M, into buf ---------------
    template <int LOGM>
    __device__ void load_and_forward(float2 *buf, float2 *scr, const float *src, int valid, bool aligned,
                                     const typename fplan<LOGM>::real &rf, int tid)
    {
        using PL = fplan<LOGM>;
        constexpr int M = PL::N, T = PL::T;
        for (int n = tid; n < M; n += T)
        {
            float2 v = make_float2(0.0f, 0.0f);
            const int i = 2 * n;
            if (i + 1 < valid)
                v = aligned ? *reinterpret_cast<const float2 *>(src + i) : make_float2(src[i], src[i + 1]);
            else if (i < valid)
                v.x = src[i];
            buf[n] = v;
        }
        __syncthreads();
        rf.forward(buf, scr, tid);
    }

    __device__ __forceinline__ float2 image_mul(float2 x, float2 h, int k)
    {
        return (k == 0) ? make_float2(x.x * h.x, x.y * h.y) : cmul(x, h);      // bin 0 packs DC and Nyquist
    }

    // ---- IR partition -> image (Convolver::init, Convolver.cpp:152-197) -----------------------------------
    template <int LOGM>
    __global__ __launch_bounds__(fplan<LOGM>::T)
    void conv_parse_kernel(float2 *H, const float *ir, size_t ir_stride, const uint32_t *__restrict__ counts,
                           int P, const float2 *__restrict__ tw, const uint8_t *__restrict__ only /* or NULL: every channel */)
    {
        constexpr int M = fplan<LOGM>::N;
        __shared__ float2 lds_[fplan<LOGM>::LDS];
        float2 *const buf = lds_, *const scr = lds_ + fplan<LOGM>::SCR;
        const int p = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
        if (only != nullptr && only[ch] == 0)
            return;
        typename fplan<LOGM>::real rf;
        rf.load(tw, TWN, tid);
        rf.prepare();
        const int count = (counts != nullptr) ? int(counts[ch]) : P * M;
        int valid = count - p * M;
        valid = (valid < 0) ? 0 : (valid > M ? M : valid);
        load_and_forward<LOGM>(buf, scr, ir + size_t(ch) * ir_stride + size_t(p) * M, valid, false, rf, tid);
        float2 *dst = H + (size_t(ch) * P + p) * M;
        for (int k = tid; k < M; k += fplan<LOGM>::T)
            dst[k] = buf[k];
    }

    // ---- whole frame --------------------------------------------------------------------------------------
    // One workgroup, one channel.  `done` (or NULL): a per-channel counter the workgroup bumps once the frame's image is in
    // the ring and the pending tail has been consumed -- the tail role of conv_step_kernel waits for it.
    // LEAN: operands of the later phases are loaded where they are used instead of up front -- for conv_step_kernel, whose
    // frame role is not on the critical path but must leave half of the register file to the tail role's waves.
    template <int LOGM, bool LEAN>
    __device__ __forceinline__
    void frame_role(float2 *buf, float2 *scr, const int ch,
                    float *out, const float *in, size_t out_stride, size_t in_stride, bool aligned,
                    float2 *ring, int R, int slot, const float2 *__restrict__ H, int P,
                    float *acc, const float2 *__restrict__ Yt /* pending tail or NULL */,
                    const float2 *__restrict__ tw,
                    float *dl_ring /* or NULL */, uint32_t dl_size, uint32_t dl_tail, uint32_t dl_head,
                    bool upper_zero /* acc[B:2B] is known to hold zeros: neither read nor re-zeroed */,
                    uint32_t *done, bool twice = false /* bump `done` once more when acc is written (the tail role folds into it) */)
    {
        using PL = fplan<LOGM>;
        constexpr int M = PL::N, T = PL::T, B = M;
        const int tid = threadIdx.x;
        MI_CPROBE(0);
        // Request order = order of use (vmcnt counts in order): twiddles, the frame, then the operands of the later
        // phases -- the head partition's image, the pending tail and the overlap-add accumulator wait in registers.
        // One exposed HBM latency instead of four.
        typename fplan<LOGM>::real rf;
        rf.load(tw, TWN, tid);
        constexpr int KPT = M / T, NPT = (M / 2) / T;
        const float *x = in + size_t(ch) * in_stride;
        float2 xin[KPT];
        #pragma unroll
        for (int i = 0; i < KPT; ++i)
        {
            const int n = tid + i * T;                      // B samples, zero-padded to 2B
            if (dl_ring != nullptr)
            {
                // Equalizer FIR path: the frame is what a delay line of dl_size cells gives back (cells dl_tail ...,
                // even offsets, so a pair never straddles the end), and the call's own samples go into the line
                // at dl_head ... -- the two ranges do not meet (delay.hip, delay_exchange_kernel)
                float *line = dl_ring + size_t(ch) * dl_size;
                xin[i] = make_float2(0.0f, 0.0f);
                if (n < B / 2)
                {
                    uint32_t r = dl_tail + 2 * n, w = dl_head + 2 * n;
                    if (r >= dl_size) r -= dl_size;
                    if (w >= dl_size) w -= dl_size;
                    xin[i] = *reinterpret_cast<const float2 *>(line + r);
                    *reinterpret_cast<float2 *>(line + w) =
                        aligned ? *reinterpret_cast<const float2 *>(x + 2 * n) : make_float2(x[2 * n], x[2 * n + 1]);
                }
                continue;
            }
            xin[i] = (n >= B / 2) ? make_float2(0.0f, 0.0f)
                   : aligned ? *reinterpret_cast<const float2 *>(x + 2 * n) : make_float2(x[2 * n], x[2 * n + 1]);
        }
        const float2 *h0 = H + size_t(ch) * P * M;
        const float2 *yt = (Yt != nullptr) ? Yt + size_t(ch) * M : nullptr;
        float *a = acc + size_t(ch) * 2 * B;
        // 512-point transforms and up: the frame goes into the forward transform in registers and comes out of the inverse in
        // registers (fft_lds REG_IN / REG_OUT), and split + product + merge are ONE pass over LDS (real_split_filter_merge):
        // four LDS round trips and four barriers less per frame
        constexpr bool REGS = !fplan<LOGM>::radix16 && (mi_fft::plan<LOGM>::T == mi_fft::plan<LOGM>::TB);
        // bin of register slot i: in order (REGS: by pairs -- slot i < NPT holds bin k = tid + i T, slot i + NPT its partner
        // M - k, the partner of the packed bin 0 being M / 2)
        auto bin_of = [&](int i) -> int {
            if (!REGS)
                return tid + i * T;
            const int k = tid + (i % NPT) * T;
            return (i < NPT) ? k : (k == 0) ? M / 2 : M - k;
        };
        float2 hreg[LEAN ? 1 : KPT], yreg[LEAN ? 1 : KPT], a0[LEAN ? 1 : NPT], a1[LEAN ? 1 : NPT];
        if (!LEAN)
        {
            #pragma unroll
            for (int i = 0; i < KPT; ++i)
            {
                hreg[i] = h0[bin_of(i)];
                yreg[i] = (yt != nullptr) ? yt[bin_of(i)] : make_float2(0.0f, 0.0f);
            }
            #pragma unroll
            for (int i = 0; i < NPT; ++i)
            {
                a0[i] = *reinterpret_cast<const float2 *>(a + 2 * (tid + i * T));
                a1[i] = upper_zero ? make_float2(0.0f, 0.0f) : *reinterpret_cast<const float2 *>(a + B + 2 * (tid + i * T));
            }
        }
        MI_CPROBE(1);
        rf.prepare();
        v2f io[KPT];
        if constexpr (REGS)
        {
            #pragma unroll
            for (int i = 0; i < KPT; ++i)
                io[i] = v2f{xin[i].x, xin[i].y};
            mi_fft::fft_lds<LOGM, false, true, false>(buf, scr, rf.ft, tid, io);
        }
        else
        {
            #pragma unroll
            for (int i = 0; i < KPT; ++i)
                buf[tid + i * T] = xin[i];
            __syncthreads();
            rf.forward(buf, scr, tid);
        }
        MI_CPROBE(2);

        // the frame's image enters the ring; its product with the head partition plus the pending tail goes back
        const __amdgpu_buffer_rsrc_t rring = mi::wt_buffer((R > 0) ? ring + (size_t(ch) * R + slot) * M : nullptr,
                                                           (R > 0) ? unsigned(M * sizeof(float2)) : 0u);
        // bin k of the image: into the ring, times the head partition's, plus the pending tail's
        auto through = [&](int slot_i, int k, float2 xk) -> float2 {
            mi::wt_store(rring, k * int(sizeof(float2)), xk);        // dropped by the bounds check when there is no ring
            const float2 hk = LEAN ? h0[k] : hreg[slot_i];
            const float2 yk = LEAN ? ((yt != nullptr) ? yt[k] : make_float2(0.0f, 0.0f)) : yreg[slot_i];
            return cadd(image_mul(xk, hk, k), yk);
        };
        if constexpr (!REGS)
        {
            #pragma unroll
            for (int i = 0; i < KPT; ++i)
            {
                const int k = tid + i * T;
                buf[k] = through(i, k, buf[k]);
            }
        }
        // The image went to the ring with write-through (sc1) stores; EVERY storing wave drains them (vmcnt 0) before it
        // arrives at the barrier behind which thread 0 moves the counter.  The wait is written out: a workgroup-scope
        // release only lowers to lgkmcnt(0) on gfx950 and s_barrier adds no vmcnt wait of its own, so without it the
        // counter could overtake another wave's image stores (MI355X_MICROARCH.md, valid hand-off forms, condition 3).
        // Inline asm: the waitcnt-insertion pass cannot drop it (build check: tests/test_abi.py greps the ISA for it).
        // NOT a device-scope release fence: that writes back the whole L2 of the XCD.
        // Split, product and merge: ONE pass over LDS where the frame is the launch (conv_frame_kernel: equalizer FIR 9.8 ->
        // 9.2 us per block), three passes in the frame ROLE of conv_step_kernel, whose operands are loaded where they are used
        // (LEAN) -- fused there the step measured 44.6 against 44.1 us (profiles/r03_experiments/conv_frame_one_pass.txt)
        if constexpr (REGS && LEAN)
        {
            mi_fft::real_split<LOGM>(buf, rf.rt, tid);
            #pragma unroll
            for (int i = 0; i < KPT; ++i)
            {
                const int k = bin_of(i);
                buf[k] = through(i, k, buf[k]);
            }
            if (done != nullptr)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (done != nullptr && tid == 0)                // (the tail role may go on while this one merges)
                __hip_atomic_fetch_add(done + ch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mi_fft::real_merge<LOGM>(buf, rf.rt, tid);
        }
        else if constexpr (REGS)
        {
            // (the pass ends with the barrier; the wait for the image stores stands in front of it)
            mi_fft::real_split_filter_merge<LOGM>(buf, rf.rt, tid,
                [&](int i, int k, float2 xk, bool partner) -> float2 { return through(partner ? i + NPT : i, k, xk); },
                [&]() { if (done != nullptr) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); });
        }
        else
        {
            if (done != nullptr)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (!(REGS && LEAN) && done != nullptr && tid == 0)
            __hip_atomic_fetch_add(done + ch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        MI_CPROBE(3);
        if constexpr (REGS)
            mi_fft::fft_lds<LOGM, true, false, true>(buf, scr, rf.ft, tid, io);
        else
            rf.inverse(buf, scr, tid);
        MI_CPROBE(4);

        const float scale = 1.0f / float(2 * M);
        float *o = (out != nullptr) ? out + size_t(ch) * out_stride : nullptr;
        const __amdgpu_buffer_rsrc_t racc = mi::wt_buffer(a, unsigned(2 * B * sizeof(float)));
        const __amdgpu_buffer_rsrc_t rout = mi::wt_buffer(o, (out != nullptr) ? unsigned(B * sizeof(float)) : 0u);
        #pragma unroll
        for (int i = 0; i < NPT; ++i)
        {
            const int n = tid + i * T;
            const float2 y0 = REGS ? make_float2(io[i].x, io[i].y) : buf[n];
            const float2 y1 = REGS ? make_float2(io[i + NPT].x, io[i + NPT].y) : buf[n + M / 2];
            const float2 p0 = LEAN ? *reinterpret_cast<const float2 *>(a + 2 * n) : a0[i];
            const float2 p1 = !LEAN ? a1[i] : upper_zero ? make_float2(0.0f, 0.0f) : *reinterpret_cast<const float2 *>(a + B + 2 * n);
            const float2 r = make_float2(fmaf(y0.x, scale, p0.x), fmaf(y0.y, scale, p0.y));
            if (out == nullptr)
                ;                                           // a frame received in blocks: its own half has gone out block by block
            else if (aligned)
                mi::wt_store<mi::CPOL_NT_SC1>(rout, 8 * n, r);
            else
            {
                mi::wt_store<mi::CPOL_NT_SC1>(rout, 8 * n, r.x);
                mi::wt_store<mi::CPOL_NT_SC1>(rout, 8 * n + 4, r.y);
            }
            mi::wt_store(racc, 8 * n, make_float2(fmaf(y1.x, scale, p1.x), fmaf(y1.y, scale, p1.y)));
            if (!upper_zero)
                mi::wt_store(racc, 4 * B + 8 * n, make_float2(0.0f, 0.0f));
        }
        MI_CPROBE(5);
        if (done != nullptr && twice)
        {
            // the accumulator is in memory (write-through stores, drained by every wave before the barrier, as above): the
            // tail role of this channel may add its inverse transform to it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0)
                __hip_atomic_fetch_add(done + ch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    template <int LOGM>
    __global__ __launch_bounds__(fplan<LOGM>::T)
    void conv_frame_kernel(float *out, const float *in, size_t out_stride, size_t in_stride, bool aligned,
                           float2 *ring, int R, int slot, const float2 *__restrict__ H, int P,
                           float *acc, const float2 *__restrict__ Yt, const float2 *__restrict__ tw,
                           float *dl_ring, uint32_t dl_size, uint32_t dl_tail, uint32_t dl_head, bool upper_zero)
    {
        __shared__ float2 lds_[fplan<LOGM>::LDS];
        float2 *const buf = lds_, *const scr = lds_ + fplan<LOGM>::SCR;
        frame_role<LOGM, false>(buf, scr, blockIdx.x, out, in, out_stride, in_stride, aligned, ring, R, slot, H, P, acc, Yt, tw,
                                dl_ring, dl_size, dl_tail, dl_head, upper_zero, nullptr);
    }


    // ---- several whole frames of a single-partition bank in ONE launch (the Equalizer's FIR path, block after block) ------
    // A launch per block (conv_frame_kernel) fetches, per channel and block, the response's image (8 B per sample), the
    // overlap-add tail (4 + 4 B) and the frame out of the delay line (4 B) next to the 4 + 4 B of samples in and out, and does
    // nothing else while they arrive.  Here a channel's workgroup walks K consecutive blocks: the image and the tail stay in
    // registers from block to block, the delay line is read once (block f's frame IS block f - 1 of the call: the line is a
    // delay of exactly one block) and written once, and block f + 1's samples are asked for before block f's transforms.
    // Same transforms, same products, same sums as K launches of conv_frame_kernel: bit-identical results.
    constexpr int FRAMES_MAX_BLOCKS = 128;
    struct frames_args
    {
        int             blocks;
        float          *out[FRAMES_MAX_BLOCKS];
        const float    *in[FRAMES_MAX_BLOCKS];
    };

    template <int LOGM>
    __global__ __launch_bounds__(fplan<LOGM>::T)
    void conv_frames_kernel(const frames_args fa, size_t out_stride, size_t in_stride, bool aligned,
                            const float2 *__restrict__ H /* [channels][M]: one partition */, float *acc,
                            const float2 *__restrict__ Yt /* always NULL (a single partition owes no tail): see `through` */,
                            const float2 *__restrict__ tw,
                            float *dl_ring, uint32_t dl_size, uint32_t dl_tail, uint32_t dl_head, bool upper_zero)
    {
        using PL = fplan<LOGM>;
        constexpr int M = PL::N, T = PL::T, B = M, KPT = M / T, NPT = KPT / 2;
        static_assert(!PL::radix16 && mi_fft::plan<LOGM>::T == mi_fft::plan<LOGM>::TB, "register hand-over of the transforms (512 .. 8192 points)");
        __shared__ float2 lds_[PL::LDS];
        float2 *const buf = lds_, *const scr = lds_ + PL::SCR;
        const int ch = blockIdx.x, tid = threadIdx.x;
        typename PL::real rf;
        rf.load(tw, TWN, tid);
        float *const line = dl_ring + size_t(ch) * dl_size;
        float *const a = acc + size_t(ch) * 2 * B;
        const float2 *const h0 = H + size_t(ch) * M;
        // the first frame: what the delay line gives back (cells dl_tail ..., even offsets: a pair never straddles the end)
        float2 xin[NPT];
        #pragma unroll
        for (int i = 0; i < NPT; ++i)
        {
            uint32_t r = dl_tail + 2 * (tid + i * T);
            if (r >= dl_size) r -= dl_size;
            xin[i] = *reinterpret_cast<const float2 *>(line + r);
        }
        // the image by pairs of bins (slot i: bin k = tid + i T, slot i + NPT its partner M - k; bin 0 packs DC and Nyquist,
        // its partner slot holds bin M / 2) and the overlap-add tail: in registers for the whole launch
        // (yreg: the pending tail frame_role adds to the product -- zeros here, but kept as the same run-time operand so that
        // the compiler contracts product and sum into the same fused multiply-adds as in conv_frame_kernel: same bits)
        float2 hreg[KPT], yreg[KPT], a0[NPT], a1[NPT];
        const float2 *const yt = (Yt != nullptr) ? Yt + size_t(ch) * M : nullptr;
        #pragma unroll
        for (int i = 0; i < NPT; ++i)
        {
            const int k = tid + i * T, km = (k == 0) ? M / 2 : M - k;
            hreg[i] = h0[k];
            hreg[i + NPT] = h0[km];
            yreg[i] = (yt != nullptr) ? yt[k] : make_float2(0.0f, 0.0f);
            yreg[i + NPT] = (yt != nullptr) ? yt[km] : make_float2(0.0f, 0.0f);
            a0[i] = *reinterpret_cast<const float2 *>(a + 2 * k);
            a1[i] = upper_zero ? make_float2(0.0f, 0.0f) : *reinterpret_cast<const float2 *>(a + B + 2 * k);
        }
        rf.prepare();
        const float scale = 1.0f / float(2 * M);
        for (int f = 0; f < fa.blocks; ++f)
        {
            // the call's block f: the frame of block f + 1 (or what the delay line holds after the call)
            const float *x = fa.in[f] + size_t(ch) * in_stride;
            float2 xnext[NPT];
            #pragma unroll
            for (int i = 0; i < NPT; ++i)
            {
                const int n = tid + i * T;
                xnext[i] = aligned ? *reinterpret_cast<const float2 *>(x + 2 * n) : make_float2(x[2 * n], x[2 * n + 1]);
            }
            v2f io[KPT];
            #pragma unroll
            for (int i = 0; i < NPT; ++i)
            {
                io[i] = v2f{xin[i].x, xin[i].y};
                io[i + NPT] = v2f{0.0f, 0.0f};              // B samples zero-padded to 2 B
            }
            mi_fft::fft_lds<LOGM, false, true, false>(buf, scr, rf.ft, tid, io);
            mi_fft::real_split_filter_merge<LOGM>(buf, rf.rt, tid,
                [&](int i, int k, float2 xk, bool partner) -> float2 {
                    const int slot_i = partner ? i + NPT : i;
                    return cadd(image_mul(xk, hreg[slot_i], k), yreg[slot_i]);
                },
                []() { });
            mi_fft::fft_lds<LOGM, true, false, true>(buf, scr, rf.ft, tid, io);
            float *o = fa.out[f] + size_t(ch) * out_stride;
            const __amdgpu_buffer_rsrc_t rout = mi::wt_buffer(o, unsigned(B * sizeof(float)));
            #pragma unroll
            for (int i = 0; i < NPT; ++i)
            {
                const int n = tid + i * T;
                const float2 r = make_float2(fmaf(io[i].x, scale, a0[i].x), fmaf(io[i].y, scale, a0[i].y));
                if (aligned)
                    mi::wt_store<mi::CPOL_NT_SC1>(rout, 8 * n, r);
                else
                {
                    mi::wt_store<mi::CPOL_NT_SC1>(rout, 8 * n, r.x);
                    mi::wt_store<mi::CPOL_NT_SC1>(rout, 8 * n + 4, r.y);
                }
                a0[i] = make_float2(fmaf(io[i + NPT].x, scale, a1[i].x), fmaf(io[i + NPT].y, scale, a1[i].y));
                a1[i] = make_float2(0.0f, 0.0f);
                xin[i] = xnext[i];
            }
            __syncthreads();                                // the transforms' buffers are free for the next frame
        }
        // what the next call starts from: the last block in the delay line (where K single-block calls would have left it),
        // the tail in the accumulator, its upper half zero
        const __amdgpu_buffer_rsrc_t racc = mi::wt_buffer(a, unsigned(2 * B * sizeof(float)));
        #pragma unroll
        for (int i = 0; i < NPT; ++i)
        {
            const int n = tid + i * T;
            uint32_t w = uint32_t((uint64_t(dl_head) + uint64_t(fa.blocks - 1) * B + 2 * n) % dl_size);
            *reinterpret_cast<float2 *>(line + w) = xin[i];
            mi::wt_store(racc, 8 * n, a0[i]);
            if (!upper_zero)
                mi::wt_store(racc, 4 * B + 8 * n, make_float2(0.0f, 0.0f));
        }
    }

    // ---- the same run of blocks at B = 4096 on the wave-resident transform (fft_wave.h) ------------------------------------------
    // conv_frames_kernel<12> runs AT the rate of its transforms (1.9 us per 4096-point transform and CU: their passes through LDS
    // and their butterflies add up, profiles/r05_experiments/fft_cores_probe.txt).  Here a WAVE owns a block: 64 registers per lane
    // of samples, forward transform, split-product-merge as one step per bin (table (alpha, beta) of the response,
    // conv_wave_table_kernel), inverse -- no workgroup-wide pass and no barrier.
    // Overlap-SAVE, so that the waves owe each other nothing: with u_g the frame of block g (block g - 1 of the call: the delay line
    // is a delay of exactly one block; u_0 is what the line holds), unit g transforms the 2 B samples [u_(g-1) | u_g] and the upper
    // half of the circular convolution IS block g's output (the response has at most B taps).  u_(-1) := 0 makes unit 0's upper half
    // the lower half of u_0 * h, to which the accumulator the call found is added; u_K := 0 makes unit K's upper half the tail the
    // call leaves in the accumulator: K + 1 units for K blocks, each input block read by two waves (the second time out of the L2),
    // no sums between the waves, no LDS besides the waves' own exchange areas.  (The first version handed overlap-add tails from
    // wave to wave through LDS behind two barriers per eight blocks: 441 us per 128 blocks at 256 channels, a quarter of it the
    // exposed latency of loads and stores that the lockstep of the barriers put in the same place for all eight waves.)
    // The host sends a run this way only if no output buffer of the call overlaps an input buffer and everything is 8-byte aligned.
    // The same convolution in another order of roundings: within 1e-6 of conv_frames_kernel<12>, not bit-identical.
    constexpr int WAVE_FRAMES = 8;                          // waves of a workgroup
    __global__ __launch_bounds__(64 * WAVE_FRAMES, 2)
    void conv_frames_wave_kernel(const frames_args fa, size_t out_stride, size_t in_stride,
                                 const float4 *__restrict__ tabs /* [channels][4096]: (alpha, beta) per bin */, float *acc,
                                 const float2 *__restrict__ tw,
                                 float *dl_ring, uint32_t dl_size, uint32_t dl_tail, uint32_t dl_head, bool upper_zero)
    {
        using namespace mi_fftw;
        constexpr int B = N, HALF = R / 2;                  // samples of a block; registers of a half frame
        __shared__ float areas[WAVE_FRAMES][AREA];
        __shared__ float2 pl[16 * R];
        const int ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);           // (a scalar: the buffers of a wave's unit are uniform)
        fill_table_pq(pl, tw, tid, 64 * WAVE_FRAMES);
        __syncthreads();
        const __amdgpu_buffer_rsrc_t tab = table_of(tabs + size_t(ch) * N);
        float *const line = dl_ring + size_t(ch) * dl_size;
        float *const a = acc + size_t(ch) * 2 * B;
        const __amdgpu_buffer_rsrc_t racc = mi::wt_buffer(a, unsigned(2 * B * sizeof(float)));
        // two floats at byte `lane_off` (a VGPR) + `row_off` (a scalar): one base per buffer, not a 64-bit pointer per row
        auto pair_at = [](__amdgpu_buffer_rsrc_t r, int lane_off, int row_off) -> v2f {
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            const u2 d = __builtin_amdgcn_raw_buffer_load_b64(r, lane_off, row_off, 0);
            return v2f{__uint_as_float(d.x), __uint_as_float(d.y)};
        };
        v2f x[R];
        for (int g = wv; g <= fa.blocks; g += WAVE_FRAMES)
        {
            // [u_(g-1) | u_g]: sample pair n = lane + 64 j of the first half in register j, of the second in register HALF + j
            #pragma unroll
            for (int half = 0; half < 2; ++half)
            {
                const int u = g - 1 + half;                 // frame u: block u - 1 of the call
                if (u < 0 || u >= fa.blocks)
                {
                    #pragma unroll
                    for (int j = 0; j < HALF; ++j)
                        x[half * HALF + j] = v2f{0.0f, 0.0f};
                }
                else if (u == 0)
                {
                    const __amdgpu_buffer_rsrc_t rline = mi::wt_buffer(line, unsigned(dl_size * sizeof(float)));
                    int ln = lane;
                    asm volatile("" : "+v"(ln));            // (the 32 wrapped offsets are made here, once per launch, not ahead of the loop)
                    #pragma unroll
                    for (int j = 0; j < HALF; ++j)
                    {
                        uint32_t r = dl_tail + 2 * (ln + 64 * j);       // even offsets: a pair never straddles the end
                        if (r >= dl_size) r -= dl_size;
                        x[half * HALF + j] = pair_at(rline, int(r * sizeof(float)), 0);
                    }
                }
                else
                {
                    const __amdgpu_buffer_rsrc_t rin = mi::wt_buffer(const_cast<float *>(fa.in[u - 1]) + size_t(ch) * in_stride,
                                                                     unsigned(B * sizeof(float)));
                    #pragma unroll
                    for (int j = 0; j < HALF; ++j)
                        x[half * HALF + j] = pair_at(rin, lane * 8, j * 512);
                }
            }
            float4 q[2 * AHEAD];
            fft4096_t<false>(x, pl, areas[wv], lane, [&]() { table_ahead(q, tab, lane); });
            split_filter_merge(x, q, tab, lane);
            fft4096_t<true>(x, pl, areas[wv], lane);
            // register HALF + i: samples 2 n, 2 n + 1 of block g (n = lane + 64 i); the lower half is the circular wrap: not used
            if (g == 0)
            {
                #pragma unroll
                for (int i = 0; i < HALF; ++i)
                    x[HALF + i] += pair_at(racc, lane * 8, i * 512);
            }
            else if (g == 1 && !upper_zero)
            {
                #pragma unroll
                for (int i = 0; i < HALF; ++i)
                    x[HALF + i] += pair_at(racc, lane * 8, 4 * B + i * 512);
            }
            if (g < fa.blocks)
            {
                const __amdgpu_buffer_rsrc_t rout = mi::wt_buffer(fa.out[g] + size_t(ch) * out_stride, unsigned(B * sizeof(float)));
                #pragma unroll
                for (int i = 0; i < HALF; ++i)
                    mi::wt_store<mi::CPOL_NT_SC1>(rout, 8 * (lane + 64 * i), make_float2(x[HALF + i].x, x[HALF + i].y));
            }
        }
        // What the next call starts from: the tail (unit K's result: the last one of its wave, still in its registers) in the
        // accumulator, its upper half zero -- behind a barrier: units 0 and 1 have read the accumulator the call found.
        __syncthreads();
        if (wv == fa.blocks % WAVE_FRAMES)
        {
            #pragma unroll
            for (int i = 0; i < HALF; ++i)
            {
                mi::wt_store(racc, 8 * (lane + 64 * i), make_float2(x[HALF + i].x, x[HALF + i].y));
                if (!upper_zero)
                    mi::wt_store(racc, 4 * B + 8 * (lane + 64 * i), make_float2(0.0f, 0.0f));
            }
        }
        // ... and the last block in the delay line (where as many single-block calls would have left it)
        // (the loads first, then the stores: alternating, every store waits for its load's round trip)
        const float *last = fa.in[fa.blocks - 1] + size_t(ch) * in_stride;
        constexpr int COPIES = B / 2 / (64 * WAVE_FRAMES);
        float2 v[COPIES];
        #pragma unroll
        for (int i = 0; i < COPIES; ++i)
            v[i] = *reinterpret_cast<const float2 *>(last + 2 * (tid + i * 64 * WAVE_FRAMES));
        #pragma unroll
        for (int i = 0; i < COPIES; ++i)
        {
            const int n = tid + i * 64 * WAVE_FRAMES;
            const uint32_t w = uint32_t((uint64_t(dl_head) + uint64_t(fa.blocks - 1) * B + 2 * n) % dl_size);
            *reinterpret_cast<float2 *>(line + w) = v[i];
        }
    }

    // (alpha, beta) of fft_wave.h's split_filter_merge from a single-partition bank's images, bin by bin
    __global__ __launch_bounds__(256)
    void conv_wave_table_kernel(float4 *tabs, const float2 *__restrict__ H, const float2 *__restrict__ tw)
    {
        constexpr int M = mi_fftw::N;
        const int ch = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
        const float2 *h = H + size_t(ch) * M;
        const float2 h0 = h[0];                             // (DC, Nyquist)
        const float2 hk = (k == 0) ? make_float2(h0.x, 0.0f) : h[k];
        const float2 hp = (k == 0) ? make_float2(h0.y, 0.0f) : h[M - k];
        const float sx = 0.5f * (hk.x + hp.x), sy = 0.5f * (hk.y - hp.y);      // S, D = (H[k] +- conj H[M - k]) / 2
        const float dx = 0.5f * (hk.x - hp.x), dy = 0.5f * (hk.y + hp.y);
        const float2 w = tw[k * (TWN / (2 * M))];           // e^{-i pi k / M}
        const float sc = 1.0f / float(M);
        tabs[size_t(ch) * M + k] = make_float4((sx + dx * w.y) * sc, (sy + dy * w.y) * sc, -dy * w.x * sc, dx * w.x * sc);
    }

    // ---- K whole frames of a partitioned bank in ONE call (mi_convolver_bank_process_blocks) ------------------------------------
    // A frame per launch streams, frame after frame, all partitions' images and the whole ring (272 B per channel-sample at
    // C3): every H_p is read once per frame.  With K frames in hand the tails of all of them come out of ONE pass over the
    // partitions -- a thread owns a pair of bins, keeps the K sums and a window of K frames' values of that pair in registers
    // and takes the partitions in the order the tail role of conv_step_kernel takes them (p = 2 .. P - 1, then p = 1), so every
    // sum goes through the same additions in the same order: the same floats as K one-launch steps.  Three launches:
    //   conv_batch_forward_kernel   the K frames' images (forward transform + split, as frame_role does it) -> staging
    //   conv_batch_tail_kernel<K>   Yt_f = sum_{p >= 1} H_p X_(f + 1 - p) for the K frames: H once, the ring's and the staged
    //                               images once (2 K + 2 P loads of 16 bytes per thread instead of 2 K (P - 1))
    //   conv_batch_frames_kernel    per channel, frame after frame: image times H_0 plus the tail owed to it, merge, inverse,
    //                               overlap-add, emission; the image enters the ring; the accumulator stays in registers
    constexpr int BATCH_MAX = 16;
    struct batch_args
    {
        int             frames;
        float          *out[BATCH_MAX];
        const float    *in[BATCH_MAX];
    };

    template <int LOGM>
    __global__ __launch_bounds__(fplan<LOGM>::T)
    void conv_batch_forward_kernel(const batch_args ba, size_t in_stride, bool aligned, float2 *ring, int R, int slot0,
                                   const float2 *__restrict__ tw)
    {
        using PL = fplan<LOGM>;
        constexpr int M = PL::N, T = PL::T, B = M, KPT = M / T, NPT = KPT / 2;
        static_assert(!PL::radix16 && mi_fft::plan<LOGM>::T == mi_fft::plan<LOGM>::TB, "register hand-over of the transforms (512 .. 8192 points)");
        __shared__ float2 lds_[PL::LDS];
        float2 *const buf = lds_, *const scr = lds_ + PL::SCR;
        const int ch = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
        typename PL::real rf;
        rf.load(tw, TWN, tid);
        const float *x = ba.in[f] + size_t(ch) * in_stride;
        v2f io[KPT];
        #pragma unroll
        for (int i = 0; i < KPT; ++i)
        {
            const int n = tid + i * T;                      // B samples, zero-padded to 2 B
            const float2 v = (n >= B / 2) ? make_float2(0.0f, 0.0f)
                           : aligned ? *reinterpret_cast<const float2 *>(x + 2 * n) : make_float2(x[2 * n], x[2 * n + 1]);
            io[i] = v2f{v.x, v.y};
        }
        rf.prepare();
        mi_fft::fft_lds<LOGM, false, true, false>(buf, scr, rf.ft, tid, io);
        mi_fft::real_split<LOGM>(buf, rf.rt, tid);
        // (the ring has room for a batch next to the P - 2 frames the batch's tails still need: see launch_batch)
        float2 *dst = ring + (size_t(ch) * R + (slot0 + 1 + f) % R) * M;
        #pragma unroll
        for (int i = 0; i < KPT; ++i)                       // (the pairs this thread split itself)
        {
            const int k0 = tid + (i % NPT) * T;
            const int k = (i < NPT) ? k0 : (k0 == 0) ? M / 2 : M - k0;
            dst[k] = buf[k];
        }
    }

    #ifndef MI_TAIL_DEPTH
    #define MI_TAIL_DEPTH 2
    #endif
    template <int K, bool KEEP>
    __global__ __launch_bounds__(256, 2)                    // two workgroups per CU (the staged frames in LDS allow no more at K = 16)
    void conv_batch_tail_kernel(float2 *yps /* [channels][K][M]: H_0 X_f + Yt_(f-1), what frame f's inverse transform takes */,
                                float2 *yt /* [channels][M]: in, the tail pending before the call (if `pending`); out, Yt_(K-1) */,
                                bool pending, const float2 *__restrict__ ring, int R, int slot0,
                                const float2 *__restrict__ H, int P, int M)
    {
        typedef float f4 __attribute__((ext_vector_type(4)));
        // (a channel's first piece, which also forms the packed first bins, moves from XCD to XCD with the channel: a launch's
        // workgroups go to the XCDs in turn, and with piece = blockIdx.x all first pieces met on one of them)
        const int ch = blockIdx.y, piece = (blockIdx.x + blockIdx.y) % gridDim.x;
        const int idx = piece * 256 + threadIdx.x, M4 = M / 2;                            // (M4 is a multiple of 256: M >= 512)
        const f4 *Hc = reinterpret_cast<const f4 *>(H + size_t(ch) * P * M);
        const f4 *Xr = reinterpret_cast<const f4 *>(ring + size_t(ch) * R * M);
        // frame m of the call (m >= 0) or before it (m < 0): frame -1 sits in ring slot slot0, frame 0 in the one behind it, ...
        auto image = [&](int m) -> const f4 * {
            int r = (slot0 + 1 + m) % R;
            r = (r < 0) ? r + R : r;
            return Xr + size_t(r) * M4;
        };
        // the tail role's sums, term for term: per bin  t = (fma(-x.y, h.y, s.x), fma(x.y, h.x, s.y)),  s = (fma(x.x, h.x, t.x),
        // fma(x.x, h.y, t.y)).  The first pair is ONE v_pk_fma_f32 whose operand selectors take x's and h's halves crosswise
        // (written out: the compiler copies x.y into a register of its own first, a move per multiply-add pair -- a quarter
        // of the walk's instructions)
        typedef float f2 __attribute__((ext_vector_type(2)));
        auto cmac = [](const f2 s, const f2 h, const f2 x) __attribute__((always_inline)) -> f2 {
            f2 t;
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(t) : "v"(x), "v"(h), "v"(s));
            return f2{fmaf(x.x, h.x, t.x), fmaf(x.x, h.y, t.y)};
        };
        auto mac = [&](f4 &s, const f4 h, const f4 x) __attribute__((always_inline))
        {
            s.hi = cmac(s.hi, h.hi, x.hi);
            s.lo = cmac(s.lo, h.lo, x.lo);
        };
        // bin 0 packs (DC, Nyquist): two real products instead of a complex one -- lane f of the first workgroup forms the first
        // pair's first bin of frame f's tail (the same chain of multiply-adds as the tail role's dc / ny) and of what frame f's
        // inverse transform takes, NOW (the ring still holds the frames before the call), and files them at the end
        float2 fix_y = make_float2(0.0f, 0.0f), fix_t = make_float2(0.0f, 0.0f);
        MI_TPROBE(7);
        if (piece == 0 && threadIdx.x < 64)
        {
            const int f = (threadIdx.x < K) ? int(threadIdx.x) : K - 1;
            float dc = 0.0f, ny = 0.0f;
            // (sixteen partitions' operands asked for at once, then the chain: one at a time it was P - 2 memory round trips, 19 us
            // in front of everything else this workgroup does)
            for (int q0 = 2; q0 < P; q0 += 16)
            {
                float2 hh[16], xx[16];
                #pragma unroll
                for (int j = 0; j < 16; ++j)
                {
                    const int q = (q0 + j < P) ? q0 + j : P - 1;
                    hh[j] = *reinterpret_cast<const float2 *>(Hc + size_t(q) * M4);
                    xx[j] = *reinterpret_cast<const float2 *>(image(f + 1 - q));
                }
                #pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (q0 + j < P)
                    {
                        dc = fmaf(xx[j].x, hh[j].x, dc);
                        ny = fmaf(xx[j].y, hh[j].y, ny);
                    }
            }
            const f4 h = Hc[size_t(1) * M4], x = image(f)[0];
            dc = fmaf(x.x, h.x, dc);
            ny = fmaf(x.y, h.y, ny);
            // the tail owed to frame f: lane f - 1's (lane 0: the pending one from before the call)
            float pdc = __shfl_up(dc, 1), pny = __shfl_up(ny, 1);
            if (threadIdx.x == 0)
            {
                const float2 before = pending ? (yt + size_t(ch) * M)[0] : make_float2(0.0f, 0.0f);
                pdc = before.x;
                pny = before.y;
            }
            const f4 h0 = Hc[0];
            const int k0 = 2 * idx;                         // 0 in lane 0 only; lanes 1 .. K - 1 pass 0 as a run-time value as well
            fix_y = cadd(image_mul(make_float2(x.x, x.y), make_float2(h0.x, h0.y), k0 - 2 * int(threadIdx.x)), make_float2(pdc, pny));
            fix_t = make_float2(dc, ny);
        }
        MI_TPROBE(0);
        f4 s[K], xw[K];
        // the window for p = 2: frames f - 1, f = 0 .. K - 1; frame m lives in register m mod K throughout.  The staged frames
        // among them (0 .. K - 2) are needed once more, by the p = 1 term at the end: they wait in LDS instead of being read twice
        // (one block of LDS, the request queue of the walk over the partitions first: its addresses go through M0)
        constexpr int D = (K >= MI_TAIL_DEPTH) ? MI_TAIL_DEPTH : K;
        __shared__ f4 lds_[(2 * D + (KEEP ? K - 1 : 0)) * 256];
        f4 (*const qh)[256] = reinterpret_cast<f4 (*)[256]>(lds_), (*const qx)[256] = qh + D, (*const own)[256] = qx + D;
        #pragma unroll
        for (int f = 0; f < K; ++f)
        {
            s[f] = f4{0.0f, 0.0f, 0.0f, 0.0f};
            const f4 v = image(f - 1)[idx];
            xw[(f - 1 + K) % K] = v;
            if (KEEP && f >= 1)
                own[f - 1][threadIdx.x] = v;
        }
        // The partitions' images and the frames the window takes in are asked for D steps ahead of the sums that take them (a
        // step's 16 K multiply-adds are a quarter of a memory round trip) and land in LDS by themselves (LDS-DMA: a wave's 64
        // lanes write 1 KiB in lane order, each lane reads back its own 16 bytes -- no barrier, the issuing wave's counted
        // vmcnt covers it): a queue in registers went from step to step through copies of registers whose loads had just
        // been issued, i.e. waited for them at every step.  Every step issues its two requests, past the last partition a
        // repeat of it, so that the count the wait rests on never changes.
        static_assert(K % D == 0, "the queue's phase survives the loop's back edge");
        if (P > 2)
        {
            const f4 *hp = Hc + size_t(2) * M4 + idx;       // H_q, this thread's pair of bins
            const f4 *const xl = Xr + idx;
            int q = 2, rq = (slot0 + 1 - 2) % R;            // the next partition asked for and the ring slot of frame -q
            rq = __builtin_amdgcn_readfirstlane((rq < 0) ? rq + R : rq);
            const int w0 = threadIdx.x & ~63;
            // (written out: the compiler puts a vmcnt(0) in front of every LDS read that follows an LDS-DMA builtin of its own)
            auto dma = [](const f4 *src, const f4 *dst) __attribute__((always_inline)) {
                unsigned keep;
                const unsigned at = __builtin_amdgcn_readfirstlane(unsigned(uintptr_t(dst)));
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(at) : "memory");
            };
            auto ask = [&](int slot) __attribute__((always_inline)) {
                dma(hp, &qh[slot][w0]);
                dma(xl + uint32_t(rq) * uint32_t(M4), &qx[slot][w0]);    // what the window takes in for q + 1 (frame -q), in place of frame K - q
                if (q < P - 1)
                {
                    ++q;
                    hp += M4;
                    rq = (rq == 0) ? R - 1 : rq - 1;
                }
            };
            // (the window is in its registers before the first request goes out: the compiler's own wait for it would otherwise
            // sit inside the loop, a vmcnt(0) at every step)
            #pragma unroll
            for (int f = 0; f < K; ++f)
                asm volatile("" :: "v"(xw[f]));
            MI_TPROBE(1);
            #pragma unroll
            for (int d = 0; d < D; ++d)
                ask(d);
            int p = 2;
            while (p < P)
            {
                #pragma unroll
                for (int u = 0; u < K; ++u)                 // p = 2 + c K + u: frame f + 1 - p sits in register (f - 1 - u) mod K
                {
                    if (p < P)
                    {
                        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (D - 1)) : "memory");     // this step's pair has landed
                        const f4 h = qh[(K * 4 + u) % D][threadIdx.x], xn = qx[(K * 4 + u) % D][threadIdx.x];
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // ... and is in registers: the slot is free
                        ask((K * 4 + u) % D);
                        #pragma unroll
                        for (int f = 0; f < K; ++f)
                            mac(s[f], h, xw[(f - 1 - u + 2 * K) % K]);
                        xw[(2 * K - 2 - u) % K] = xn;       // (behind the last partition: a value no sum takes)
                        if (u == 0) MI_TPROBE(2);
                        if (u == 7 % K) MI_TPROBE(3);
                        ++p;
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                  // (the repeats still on their way)
        }
        MI_TPROBE(4);
        // p = 1 last: the frames' own images.  With frame f's image in hand and the tail of frame f - 1 complete, what frame f's
        // inverse transform takes is formed right here (frame_role's `through`: image times H_0 plus the tail owed to the frame)
        {
            const f4 h1 = Hc[size_t(1) * M4 + idx], h0 = Hc[idx];
            f4 *const ytc = reinterpret_cast<f4 *>(yt + size_t(ch) * M);
            f4 prev = pending ? ytc[idx] : f4{0.0f, 0.0f, 0.0f, 0.0f};
            #ifdef MI_CONV_PROBE
            asm volatile("" :: "v"(h1), "v"(h0), "v"(prev));
            MI_TPROBE(5);
            #endif
            #pragma unroll
            for (int f = 0; f < K; ++f)
            {
                const f4 x = (KEEP && f < K - 1) ? own[f][threadIdx.x] : image(f)[idx];
                mac(s[f], h1, x);
                const float2 y0 = cadd(image_mul(make_float2(x.x, x.y), make_float2(h0.x, h0.y), 2 * idx), make_float2(prev.x, prev.y));
                const float2 y1 = cadd(image_mul(make_float2(x.z, x.w), make_float2(h0.z, h0.w), 2 * idx + 1), make_float2(prev.z, prev.w));
                reinterpret_cast<f4 *>(yps + (size_t(ch) * K + f) * M)[idx] = f4{y0.x, y0.y, y1.x, y1.y};
                prev = s[f];
            }
            ytc[idx] = s[K - 1];
        }
        MI_TPROBE(6);
        // (the first workgroup: the packed first bins, formed before anything was written)
        if (piece != 0)
            return;
        __syncthreads();
        if (threadIdx.x < K)
        {
            const int f = threadIdx.x;
            (yps + (size_t(ch) * K + f) * M)[0] = fix_y;
            if (f == K - 1)
                (yt + size_t(ch) * M)[0] = fix_t;
        }
    }

    // The frames' outputs: a channel's K frames are shared out among G workgroups, each walking `per` consecutive frames with the
    // accumulator in registers and the next frame's spectrum asked for before the current frame's transforms.  The overlap-add
    // couples a frame to the one before it only through the upper half of that frame's inverse transform (acc = fma(y1, scale,
    // 0) once the upper half of the accumulator is zero), so a workgroup whose run starts inside the batch runs the inverse of
    // the frame before its first one as well and needs no other workgroup.  (One workgroup per channel: a chain of memory
    // latencies on 256 workgroups; one per frame: every spectrum read twice.)  Nothing the launch reads is written by it: the
    // accumulator the last frame leaves goes to acc_new and conv_batch_finish_kernel files it where the bank keeps it (one run per
    // channel: acc_new IS the accumulator, see the end of the kernel).
    template <int LOGM>
    __global__ __launch_bounds__(fplan<LOGM>::T)
    void conv_batch_frames_kernel(const batch_args ba, int per, size_t out_stride, bool aligned, const float2 *__restrict__ yps,
                                  const float *acc, float *acc_new, int acc_new_stride, bool zero_upper, const float2 *__restrict__ tw,
                                  bool upper_zero)
    {
        using PL = fplan<LOGM>;
        constexpr int M = PL::N, T = PL::T, B = M, KPT = M / T, NPT = KPT / 2;
        static_assert(!PL::radix16 && mi_fft::plan<LOGM>::T == mi_fft::plan<LOGM>::TB, "register hand-over of the transforms (512 .. 8192 points)");
        __shared__ float2 lds_[PL::LDS];
        float2 *const buf = lds_, *const scr = lds_ + PL::SCR;
        const int ch = blockIdx.x, tid = threadIdx.x, K = ba.frames;
        const int f0 = int(blockIdx.y) * per, f1 = (f0 + per < K) ? f0 + per : K;
        typename PL::real rf;
        rf.load(tw, TWN, tid);
        const float *const a = acc + size_t(ch) * 2 * B;
        auto bin_of = [&](int i) -> int {
            const int k = tid + (i % NPT) * T;
            return (i < NPT) ? k : (k == 0) ? M / 2 : M - k;
        };
        float2 yr[KPT];                                      // frame g's spectrum (H_0 X_g + Yt_(g-1)) by this thread's bins
        auto fetch = [&](int g)
        {
            const float2 *Y = yps + (size_t(ch) * K + g) * M;
            #pragma unroll
            for (int i = 0; i < KPT; ++i)
                yr[i] = Y[bin_of(i)];
        };
        fetch((f0 > 0) ? f0 - 1 : 0);
        float2 a0[NPT], a1[NPT];
        // what the accumulator held before the call concerns frames 0 (both halves) and 1 (the upper half under frame 0's spill)
        #pragma unroll
        for (int i = 0; i < NPT; ++i)
        {
            a0[i] = (f0 == 0) ? *reinterpret_cast<const float2 *>(a + 2 * (tid + i * T)) : make_float2(0.0f, 0.0f);
            a1[i] = (f0 <= 1 && !upper_zero) ? *reinterpret_cast<const float2 *>(a + B + 2 * (tid + i * T)) : make_float2(0.0f, 0.0f);
        }
        rf.prepare();
        const float scale = 1.0f / float(2 * M);
        v2f io[KPT];
        // the fetched spectrum, merged and through the inverse transform: io = the frame's 2 B samples (times 2 M); `next`: the
        // frame to fetch meanwhile
        auto inverse = [&](int next)
        {
            #pragma unroll
            for (int i = 0; i < KPT; ++i)
                buf[bin_of(i)] = yr[i];
            if (next >= 0)
                fetch(next);
            __syncthreads();
            mi_fft::real_merge<LOGM>(buf, rf.rt, tid);
            mi_fft::fft_lds<LOGM, true, false, true>(buf, scr, rf.ft, tid, io);
        };
        if (f0 > 0)
        {
            inverse(f0);                                    // the frame before the run: its upper half is what the run's first output starts from
            #pragma unroll
            for (int i = 0; i < NPT; ++i)
            {
                a0[i] = make_float2(fmaf(io[i + NPT].x, scale, a1[i].x), fmaf(io[i + NPT].y, scale, a1[i].y));
                a1[i] = make_float2(0.0f, 0.0f);
            }
            __syncthreads();                                // the transforms' buffers are free again
        }
        for (int f = f0; f < f1; ++f)
        {
            inverse((f + 1 < f1) ? f + 1 : -1);
            float *o = ba.out[f] + size_t(ch) * out_stride;
            const __amdgpu_buffer_rsrc_t rout = mi::wt_buffer(o, unsigned(B * sizeof(float)));
            #pragma unroll
            for (int i = 0; i < NPT; ++i)
            {
                const int n = tid + i * T;
                const float2 r = make_float2(fmaf(io[i].x, scale, a0[i].x), fmaf(io[i].y, scale, a0[i].y));
                if (aligned)
                    mi::wt_store<mi::CPOL_NT_SC1>(rout, 8 * n, r);
                else
                {
                    mi::wt_store<mi::CPOL_NT_SC1>(rout, 8 * n, r.x);
                    mi::wt_store<mi::CPOL_NT_SC1>(rout, 8 * n + 4, r.y);
                }
                a0[i] = make_float2(fmaf(io[i + NPT].x, scale, a1[i].x), fmaf(io[i + NPT].y, scale, a1[i].y));
                a1[i] = make_float2(0.0f, 0.0f);
            }
            __syncthreads();                                // the transforms' buffers are free for the next frame
        }
        if (f1 == K)                                        // the last frame's spill: the accumulator after the call
        {
            // (one run per channel: this workgroup is the accumulator's only reader in the launch, so acc_new is the accumulator
            // itself -- rows of 2 B, the upper half zeroed here if it was not -- and there is no finish launch)
            const __amdgpu_buffer_rsrc_t racc = mi::wt_buffer(acc_new + size_t(ch) * acc_new_stride, unsigned((zero_upper ? 2 * B : B) * sizeof(float)));
            #pragma unroll
            for (int i = 0; i < NPT; ++i)
            {
                mi::wt_store(racc, 8 * (tid + i * T), a0[i]);
                if (zero_upper)
                    mi::wt_store(racc, int(B * sizeof(float)) + 8 * (tid + i * T), make_float2(0.0f, 0.0f));
            }
        }
    }

    // the accumulator a batch leaves, where the bank keeps it: acc[0, B) = the last frame's spill, acc[B, 2 B) = 0
    __global__ __launch_bounds__(256)
    void conv_batch_finish_kernel(float *acc, const float *__restrict__ acc_new, int B, bool upper_zero)
    {
        const int ch = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;      // n < 2 B
        if (n < B)
            acc[size_t(ch) * 2 * B + n] = acc_new[size_t(ch) * B + n];
        else if (n < 2 * B && !upper_zero)
            acc[size_t(ch) * 2 * B + n] = 0.0f;
    }

    // ---- whole frame AND the tail owed to the next one, in one launch (P >= 2) ------------------------------------------
    // Workgroups 0 .. C-1 are the frame role above (latency bound: load, two transforms, store); workgroups C .. 2C-1
    // stream the channel's tail  Yt' = sum_{p>=1} H_p X_(k+1-p)  (bandwidth bound: H and the ring once) at the same
    // time -- one of each fits a CU (8 + 8 waves at <= 128 VGPRs, 2 x 64 KiB of LDS), so the frame role's 12 us disappear
    // under the stream instead of preceding it: 43.7 us per step at C3 against 50.4 us as two launches.
    // Partitions p >= 2 only need frames that were in the ring before the launch; the p = 1 term needs THIS frame's image:
    // the tail role takes it last, after the frame workgroup of its channel has bumped `done` (which also says that the
    // pending Yt has been consumed, so Yt can be overwritten).  The hand-over uses device-scope ATOMIC accesses only
    // (relaxed counter, write-through stores of the image, device-scope loads of it) -- a device-scope release / acquire
    // fence writes back / invalidates the whole L2 of the XCD, once per workgroup and once per poll: the first version of
    // this kernel took 118 us per step for that reason alone.
    // Frame workgroups have the lower indices and every XCD's dispatcher hands out its share of the grid in index order: a
    // tail workgroup is only placed after every frame workgroup of its XCD has been placed, frame workgroups never wait for
    // anything, so no cycle of waiting workgroups can form whatever the channel count.  That index-ordered dispatch is OBSERVED
    // behaviour of gfx950 (MI355X_MICROARCH.md, "Workgroup dispatch": blocks are dealt round-robin over the 8 XCDs, each XCD
    // places its share in index order), not a HIP guarantee; the rounding to 8 below only keeps both roles of a channel on
    // one XCD (speed).  Hence three safeguards: (1) the one-launch step is only used on gfx950 and can be switched off
    // (MI_CONV_TWO_LAUNCH=1: conv_frame_kernel + conv_mac_kernel, no in-launch waiting at all); (2) a wait that does not end
    // (about a second) bumps `fault` and raises the host-mapped `fault_host` and gives up instead of hanging the device;
    // (3) the NEXT mi_convolver_bank_process / _faults / _destroy of the bank reports MI_EHIP once that flag is up (the frame
    // in which it happened and everything after it is invalid until mi_convolver_bank_reset).  `seen` is the tail role's
    // private count of the frames it has taken.
    template <int LOGM, bool NT>
    // (one frame and one tail workgroup share a CU at LOGM = 12: 4 + 4 waves of <= 256 VGPRs; the smaller transforms have one- or
    //  two-wave workgroups, several of which fit a CU whatever they allocate)
    __global__ __launch_bounds__(fplan<LOGM>::T, (fplan<LOGM>::T >= 256) ? 2 : 1)
    void conv_step_kernel(float *out, const float *in, size_t out_stride, size_t in_stride, bool aligned,
                          float2 *ring, int R, int slot, const float2 *__restrict__ H, int P,
                          float *acc, float2 *Yt, bool yt_pending, const float2 *__restrict__ tw, bool upper_zero,
                          int channels /* of this launch */, int first /* its first channel */,
                          uint32_t *done, uint32_t *seen, uint32_t *fault, uint32_t *fault_host,
                          bool fold /* the tail goes straight into acc (time domain) instead of into Yt: a frame received in blocks */)
    {
        using PL = fplan<LOGM>;
        constexpr int M = PL::N, T = PL::T, M4 = M / 2, J = (M4 + T - 1) / T;
        __shared__ float2 lds_[fplan<LOGM>::LDS];
        float2 *const buf = lds_, *const scr = lds_ + fplan<LOGM>::SCR;
        // (the role boundary is `channels` rounded up to the 8 XCDs the workgroups are dealt to in turn: both workgroups of a
        // channel then sit on the same XCD, whose dispatcher hands out its share of the grid in index order -- frame before tail)
        const int boundary = (channels + 7) & ~7;
        if (int(blockIdx.x) < boundary)
        {
            if (int(blockIdx.x) >= channels)
                return;
            frame_role<LOGM, true>(buf, scr, first + int(blockIdx.x), out, in, out_stride, in_stride, aligned, ring, R, slot, H, P, acc,
                                   yt_pending ? Yt : nullptr, tw, nullptr, 0u, 0u, 0u, upper_zero, done, fold);
            return;
        }
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int ch = first + int(blockIdx.x) - boundary, tid = threadIdx.x;
        const f4 *Hc = reinterpret_cast<const f4 *>(H + size_t(ch) * P * M);
        const f4 *Xc = reinterpret_cast<const f4 *>(ring + size_t(ch) * R * M);
        f4 s[J];
        float dc = 0.0f, ny = 0.0f;                             // bin 0 packs (DC, Nyquist): thread 0, j = 0
        #pragma unroll
        for (int j = 0; j < J; ++j)
            s[j] = f4{0.0f, 0.0f, 0.0f, 0.0f};
        auto mac = [&](const f4 h, const f4 x, int j)
        {
            s[j].z = fmaf(x.z, h.z, fmaf(-x.w, h.w, s[j].z));
            s[j].w = fmaf(x.z, h.w, fmaf(x.w, h.z, s[j].w));
            s[j].x = fmaf(x.x, h.x, fmaf(-x.y, h.y, s[j].x));
            s[j].y = fmaf(x.x, h.y, fmaf(x.y, h.x, s[j].y));
            if (j == 0)
            {
                dc = fmaf(x.x, h.x, dc);
                ny = fmaf(x.y, h.y, ny);
            }
        };
        int r = (slot == 0) ? R - 1 : slot - 1;                 // the frame before this one
        #pragma unroll 2
        for (int p = 2; p < P; ++p)
        {
            f4 h[J], x[J];
            #pragma unroll
            for (int j = 0; j < J; ++j)
            {
                const int idx = tid + j * T;
                if (idx < M4)
                {
                    h[j] = NT ? __builtin_nontemporal_load(&Hc[size_t(p) * M4 + idx]) : Hc[size_t(p) * M4 + idx];
                    x[j] = NT ? __builtin_nontemporal_load(&Xc[size_t(r) * M4 + idx]) : Xc[size_t(r) * M4 + idx];
                }
            }
            r = (r == 0) ? R - 1 : r - 1;
            #pragma unroll
            for (int j = 0; j < J; ++j)
                if (tid + j * T < M4)
                    mac(h[j], x[j], j);
        }
        // this frame's image
        f4 h1[J];
        #pragma unroll
        for (int j = 0; j < J; ++j)
            if (tid + j * T < M4)
                h1[j] = NT ? __builtin_nontemporal_load(&Hc[size_t(1) * M4 + tid + j * T]) : Hc[size_t(1) * M4 + tid + j * T];
        auto await = [&]()                                     // thread 0: the frame workgroup's next bump of `done`
        {
            const uint32_t target = seen[ch] + 1u;
            uint32_t spins = 0;
            while (int32_t(__hip_atomic_load(done + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0)
            {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 22))
                {
                    atomicAdd(fault, 1u);
                    // host-mapped word: the next process() call of the bank sees it without a synchronisation
                    __hip_atomic_store(fault_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
            seen[ch] = target;
        };
        if (tid == 0)
            await();
        __syncthreads();
        // this frame's image: device-scope loads (past the L2 of this XCD, which never held these lines in this launch anyway)
        #pragma unroll
        for (int j = 0; j < J; ++j)
        {
            const int idx = tid + j * T;
            if (idx < M4)
            {
                const unsigned long long *q = reinterpret_cast<const unsigned long long *>(&Xc[size_t(slot) * M4 + idx]);
                const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                f4 xk;
                xk.x = __uint_as_float(unsigned(lo)); xk.y = __uint_as_float(unsigned(lo >> 32));
                xk.z = __uint_as_float(unsigned(hi)); xk.w = __uint_as_float(unsigned(hi >> 32));
                mac(h1[j], xk, j);
            }
        }
        if (tid == 0)
        {
            s[0].x = dc;
            s[0].y = ny;
        }
        if (fold)
        {
            // acc += IFFT(Yt) right here (conv_tail_kernel's work, one launch and 16 MB of Yt less): the tail's image goes
            // through this workgroup's LDS, and the accumulator -- written by the frame workgroup of the channel, on
            // another CU, in this same launch -- is read past the L1 once that workgroup has said it is there
            typename fplan<LOGM>::real rf;
            rf.load(tw, TWN, tid);
            f4 *image = reinterpret_cast<f4 *>(buf);
            #pragma unroll
            for (int j = 0; j < J; ++j)
                if (tid + j * T < M4)
                    image[tid + j * T] = s[j];
            rf.prepare();
            __syncthreads();
            rf.inverse(buf, scr, tid);
            if (tid == 0)
                await();
            __syncthreads();
            const float scale = 1.0f / float(2 * M);
            const __amdgpu_buffer_rsrc_t racc = mi::wt_buffer(acc + size_t(ch) * 2 * M, unsigned(2 * M * sizeof(float)));
            for (int n = tid; n < M; n += T)
            {
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                const u2 raw = __builtin_amdgcn_raw_buffer_load_b64(racc, n * int(sizeof(float2)), 0, mi::CPOL_SC1);
                const float2 y = buf[n], v = make_float2(__uint_as_float(raw.x), __uint_as_float(raw.y));
                mi::wt_store(racc, n * int(sizeof(float2)), make_float2(fmaf(y.x, scale, v.x), fmaf(y.y, scale, v.y)));
            }
            return;
        }
        f4 *dst = reinterpret_cast<f4 *>(Yt + size_t(ch) * M);
        #pragma unroll
        for (int j = 0; j < J; ++j)
            if (tid + j * T < M4)
                dst[tid + j * T] = s[j];
    }

    // ---- tail of the next frame: Yt = sum_{p=1..P-1} H_p * X_(newest - (p-1)) ------------------------------
    // One thread owns two neighbouring bins (16-B loads); grid = (M / 512, channels).
    template <bool NT>
    __global__ __launch_bounds__(256)
    void conv_mac_kernel(float2 *Yt, const float2 *__restrict__ ring, int R, int newest,
                         const float2 *__restrict__ H, int P, int M)
    {
        const int ch = blockIdx.y;
        const int idx = blockIdx.x * 256 + threadIdx.x;          // float4 index inside the image
        if (2 * idx >= M)
            return;
        const float4 *Hc = reinterpret_cast<const float4 *>(H + size_t(ch) * P * M);
        const float4 *Xc = reinterpret_cast<const float4 *>(ring + size_t(ch) * R * M);
        const int M4 = M / 2;
        float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float dc = 0.0f, ny = 0.0f;
        int r = newest;
        #pragma unroll 4
        for (int p = 1; p < P; ++p)
        {
            // NT: a working set beyond the Infinity Cache is read exactly once per frame -- non-temporal loads keep the
            // stream from displacing itself on the way (MI355X_MICROARCH.md, "nt-weights")
            typedef float f4 __attribute__((ext_vector_type(4)));
            float4 h, x;
            if (NT)
            {
                const f4 hv = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(&Hc[size_t(p) * M4 + idx]));
                const f4 xv = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(&Xc[size_t(r) * M4 + idx]));
                h = make_float4(hv.x, hv.y, hv.z, hv.w);
                x = make_float4(xv.x, xv.y, xv.z, xv.w);
            }
            else
            {
                h = Hc[size_t(p) * M4 + idx];
                x = Xc[size_t(r) * M4 + idx];
            }
            r = (r == 0) ? R - 1 : r - 1;
            // second bin of the pair is always an ordinary complex bin
            s.z = fmaf(x.z, h.z, fmaf(-x.w, h.w, s.z));
            s.w = fmaf(x.z, h.w, fmaf(x.w, h.z, s.w));
            // first bin: complex product; for idx == 0 it is the packed (DC, Nyquist) pair instead
            s.x = fmaf(x.x, h.x, fmaf(-x.y, h.y, s.x));
            s.y = fmaf(x.x, h.y, fmaf(x.y, h.x, s.y));
            dc  = fmaf(x.x, h.x, dc);
            ny  = fmaf(x.y, h.y, ny);
        }
        if (idx == 0)
        {
            s.x = dc;
            s.y = ny;
        }
        reinterpret_cast<float4 *>(Yt + size_t(ch) * M)[idx] = s;
    }

    template <int LOGM>
    __global__ __launch_bounds__(fplan<LOGM>::T)
    void conv_tail_kernel(float *acc, const float2 *__restrict__ Yt, const float2 *__restrict__ tw)
    {
        using PL = fplan<LOGM>;
        constexpr int M = PL::N, T = PL::T;
        __shared__ float2 lds_[fplan<LOGM>::LDS];
        float2 *const buf = lds_, *const scr = lds_ + fplan<LOGM>::SCR;
        const int ch = blockIdx.x, tid = threadIdx.x;
        typename fplan<LOGM>::real rf;
        rf.load(tw, TWN, tid);
        const float2 *src = Yt + size_t(ch) * M;
        for (int k = tid; k < M; k += T)
            buf[k] = src[k];
        rf.prepare();
        __syncthreads();
        rf.inverse(buf, scr, tid);
        const float scale = 1.0f / float(2 * M);
        float2 *a = reinterpret_cast<float2 *>(acc + size_t(ch) * 2 * M);
        for (int n = tid; n < M; n += T)
        {
            const float2 y = buf[n], v = a[n];
            a[n] = make_float2(fmaf(y.x, scale, v.x), fmaf(y.y, scale, v.y));
        }
    }

    // a call's samples into the frame (hipMemcpy2DAsync does the same behind 10 us and more of host work per call)
    __global__ __launch_bounds__(256)
    void conv_file_kernel(float *frame, int B, int off, const float *in, size_t in_stride, int cnt)
    {
        const int j = blockIdx.x * 256 + threadIdx.x;
        if (j < cnt)
            frame[size_t(blockIdx.y) * B + off + j] = in[size_t(blockIdx.y) * in_stride + j];
    }

    // ---- partial call: time-domain head (Convolver.cpp:292-296 does the same with dsp::convolve) ---------
    // The call's samples have already been copied into frame[off..off+cnt) (so `out` may alias the caller's
    // input: several workgroups of one channel read all of them).
    // `in` != NULL: the samples have NOT been copied -- they are read from the caller's rows and the first workgroup of a
    // channel files them in the frame (the host passes `in` only when `out` does not overlap it: one launch per call
    // instead of a copy and a launch).
    __global__ __launch_bounds__(256)
    void conv_direct_kernel(float *out, size_t out_stride, float *acc, float *frame,
                            const float *__restrict__ h0, int B, int off, int cnt, int taps /* <= B: h0[0 .. taps) only */,
                            int limit /* results at off + i >= limit are not produced (2 B: all of them) */,
                            const float *in = nullptr, size_t in_stride = 0)
    {
        extern __shared__ float sxin[];                         // cnt samples of this call
        const int ch = blockIdx.y, tid = threadIdx.x;
        float *fr = frame + size_t(ch) * B + off;
        const float *x = (in != nullptr) ? in + size_t(ch) * in_stride : fr;
        for (int j = tid; j < cnt; j += 256)
        {
            const float v = x[j];
            sxin[j] = v;
            if (in != nullptr && blockIdx.x == 0)
                fr[j] = v;
        }
        // the taps this workgroup's 256 results meet, h[i - j] for j in [0, cnt): a window of at most cnt + 255, in LDS as well
        // (read from memory inside the loop every multiply-add waited for its own load: 22 us for 256 samples against 512
        // taps, 1 us of arithmetic)
        float *const sh = sxin + cnt;
        const float *h = h0 + size_t(ch) * B;
        const int i0 = blockIdx.x * 256;
        const int wlo = (i0 - (cnt - 1) > 0) ? i0 - (cnt - 1) : 0;
        const int whi = (i0 + 255 < taps - 1) ? i0 + 255 : taps - 1;
        for (int k = tid; k <= whi - wlo; k += 256)
            sh[k] = h[wlo + k];
        __syncthreads();
        const int i = i0 + tid;                                 // output position relative to `off`
        if (i >= cnt + taps - 1 || off + i >= limit)
            return;
        const int jlo = (i - (taps - 1) > 0) ? i - (taps - 1) : 0;
        const int jhi = (i < cnt - 1) ? i : cnt - 1;
        float s = 0.0f;
        const float *hw = sh + (i - wlo);
        #pragma unroll 8
        for (int j = jlo; j <= jhi; ++j)
            s = fmaf(sxin[j], hw[-j], s);
        float *a = acc + size_t(ch) * 2 * B + off + i;
        const float v = *a + s;
        *a = v;
        if (i < cnt)
            out[size_t(ch) * out_stride + i] = v;
    }


    // ---- sub-frame calls of partitioned banks: the head partition as a delay line of SMALL blocks -------------------------
    // (round 3; the reference bounds the work of every call with its 128-tap head and doubling levels, Convolver.cpp:144-210,
    // 230-296.)  A call that is not a whole frame used to convolve its samples with all B taps of the head partition in
    // the time domain (conv_direct_kernel): O(n B) per call, 4096 multiply-adds per output sample at rank 13.  For banks
    // with a frame of 1024 samples or more the head partition is itself partitioned, uniformly, into blocks of SB = 256,
    // INSIDE the frame that is being received:
    //     taps [0, SB)       in the time domain, as the samples arrive (zero latency for any chunking)    [conv_direct_kernel]
    //     taps [SB, B)       a frequency-domain delay line over the frame's own blocks: when block k of the frame is
    //                        complete its image enters the small ring (slot k) and what the frame's blocks owe block
    //                        k + 1, sum_{p = 1 .. k+1} Hs_p Xs_(k+1-p), goes into acc                        [conv_small_kernel]
    //     across the frame boundary nothing is paid piecewise: every result that would land beyond the frame's end is
    //     dropped, and when the frame completes its whole spill through the head partition, IFFT(H_0 X_F)[B, 2B), is
    //     added at once by the commit kernel, which has the frame's image X_F in its hands anyway  [conv_commit_kernel<.., true>]
    //     taps [B, ...)      the frame-sized delay line, as before
    // So a frame received in pieces settles with the next frame exactly like a frame received whole, the small ring starts
    // empty with every frame, and whole-frame calls and sub-frame calls mix freely.  A piece that is exactly one aligned
    // small block takes all of its work in ONE launch (FULL: its own partition-0 term through the transform as well).
    // Work per sample is bounded by B / SB complex multiply-adds and three 512-point transforms per 256 samples.
    constexpr int LOGS = 8, SB = 1 << LOGS;

    // Block k of the frame (`off` = k SB its first sample).
    //   FULL:      the block arrives whole in this call: in -> frame, image -> ring, out = acc + (Hs_0 x)[0:SB]
    //   otherwise: the block has been received piecewise (frame holds it, its partition-0 term went through the direct
    //              kernel): image -> ring
    // then acc[off + SB, off + 3 SB) += IFFT(sum_{p = 1 .. k+1} Hs_p Xs_(k+1-p)), as far as that lies inside the frame.
    // FULL runs as TWO waves per channel (round 3): both transform the block (each into its own half of the LDS -- the
    // forward transform is on either wave's path anyway), then wave 0 takes the output (Hs_0 x, inverse, out) and wave 1
    // the debt (the sum over the small partitions, inverse, acc); wave 0 hands the upper half of its inverse, which the
    // debt's store adds, over through LDS.  One wave did the three transforms one after the other: 8.2 us per call, most
    // of it a lone wave's instruction latency.  Same arithmetic in the same order: the same floats.
    template <bool FULL, int LS = LOGS, int PMAX = (1 << (LOGM_MAX - LOGS)) - 1 /* B / SBK - 1 at the largest frame */>
    __global__ __launch_bounds__(fplan<LS>::T * (FULL ? 2 : 1))
    void conv_small_kernel(float *out, const float *in, size_t out_stride, size_t in_stride, float *frame, int B,
                           int off, float2 *sring, int Ps, const float2 *__restrict__ Hs,
                           float *acc, const float2 *__restrict__ tw)
    {
        constexpr int M = fplan<LS>::N, T = fplan<LS>::T, KPT = M / T, SBK = 1 << LS;   // SBK: samples of a block
        static_assert(KPT * T == M && (KPT % 2) == 0, "small blocks: whole pairs per thread");
        static_assert(T == 64, "one wave per role");
        __shared__ float2 lds_[FULL ? 2 : 1][fplan<LS>::LDS];
        __shared__ float2 yx[FULL ? M / 2 : 1];                             // (Hs_0 x)[SBK, 2 SBK), from the output wave to the debt wave
        const int role = FULL ? int(threadIdx.x >> 6) : 0;
        const bool outs = FULL && role == 0, debt = !FULL || role == 1;
        float2 *const buf = lds_[role], *const scr = lds_[role] + fplan<LS>::SCR;
        const int ch = blockIdx.x, tid = threadIdx.x & (T - 1), kblk = off / SBK;
        const bool next1 = (off + SBK < B), next2 = (off + 2 * SBK < B);     // the two blocks after this one, if the frame has them
        MI_SPROBE(0);
        if (FULL && role == 1 && !next1)
            return;                                                         // the frame's last block owes nothing inside the frame
        // Memory first, in the order of need: the block's samples, the twiddles, then everything else (a wave's loads return in
        // the order of their issue: what the forward transform waits for must not stand behind the 130 loads of the debt --
        // with the twiddles prepared first and the samples asked for last the kernel paid the latency of memory twice, 2.0 of
        // its 5.1 us before the first butterfly, profiles/r04_experiments/conv_small_timeline.txt)
        float *fr = frame + size_t(ch) * B + off;
        // SBK real samples = SBK / 2 pairs, zero-padded to 2 SBK: straight into the transform's registers (fft_lds REG_IN)
        v2f io[KPT];
        #pragma unroll
        for (int i = 0; i < KPT; ++i)
        {
            const int n = tid + i * T;
            float2 v = make_float2(0.0f, 0.0f);
            if (n < SBK / 2)
            {
                if (FULL)
                {
                    const float *x = in + size_t(ch) * in_stride;
                    v = make_float2(x[2 * n], x[2 * n + 1]);
                }
                else
                    v = *reinterpret_cast<const float2 *>(fr + 2 * n);
            }
            io[i] = v2f{v.x, v.y};
        }
        typename fplan<LS>::real rf;
        rf.load(tw, TWN, tid);
        float *a  = acc + size_t(ch) * 2 * B + off;
        const float2 *hs = Hs + size_t(ch) * Ps * M;
        float2 *rg = sring + size_t(ch) * Ps * M;
        // Register slots hold the image's bins by the PAIRS of the real transform's split (fft_device.h
        // real_split_filter_merge): slot i < KPT / 2 is bin k = tid + i T, slot i + KPT / 2 its partner M - k (M / 2 for the
        // packed bin 0) -- the thread that holds a pair splits it, takes it through the products and merges it again.
        auto bin_of = [&](int slot) -> int {
            const int k = tid + (slot % (KPT / 2)) * T;
            return (slot < KPT / 2) ? k : (k == 0) ? M / 2 : M - k;
        };
        // Everything the ring's debt needs that does not depend on this block is asked for NOW: the small partitions' images
        // and the images of the frame's earlier blocks (a wave has 512 registers to itself: up to 2 x 15 x 4
        // values in flight under the transforms instead of fifteen dependent round trips to L2 behind them).
        const int pmax = (!next1 || !debt) ? 0 : (kblk + 1 < Ps - 1) ? kblk + 1 : Ps - 1;
        float2 hreg[PMAX][KPT], xreg[PMAX][KPT], h0reg[KPT];
        #pragma unroll
        for (int i = 0; i < KPT; ++i)
            h0reg[i] = outs ? hs[bin_of(i)] : make_float2(0.0f, 0.0f);
        #pragma unroll
        for (int p = 1; p <= PMAX; ++p)
            if (p <= pmax)
            {
                #pragma unroll
                for (int i = 0; i < KPT; ++i)
                {
                    hreg[p - 1][i] = hs[size_t(p) * M + bin_of(i)];
                    if (p >= 2)
                        xreg[p - 1][i] = rg[size_t(kblk + 1 - p) * M + bin_of(i)];
                }
            }
        // ... and so is what the results are added to (acc: written by earlier launches of the stream only)
        float2 accv[KPT / 2], accw[KPT / 2];
        #pragma unroll
        for (int i = 0; i < KPT / 2; ++i)
        {
            const int n = tid + i * T;
            accv[i] = accw[i] = make_float2(0.0f, 0.0f);
            if (outs)
                accv[i] = *reinterpret_cast<const float2 *>(a + 2 * n);
            else if (next1)
            {
                accv[i] = *reinterpret_cast<const float2 *>(a + SBK + 2 * n);
                if (next2)
                    accw[i] = *reinterpret_cast<const float2 *>(a + 2 * SBK + 2 * n);
            }
        }
        MI_SPROBE(1);
        rf.prepare();
        if (outs)                                                           // the frame keeps its samples for the commit
        {
            #pragma unroll
            for (int i = 0; i < KPT; ++i)
                if (tid + i * T < SBK / 2)
                    *reinterpret_cast<float2 *>(fr + 2 * (tid + i * T)) = make_float2(io[i].x, io[i].y);
        }
        // (128-point transforms: their widest pass has fewer butterflies than the wave has lanes, no register hand-over)
        constexpr bool REG = (T == mi_fft::plan<LS>::TB);
        if constexpr (REG)
            mi_fft::fft_lds<LS, false, true, false>(buf, scr, rf.ft, tid, io);
        else
        {
            #pragma unroll
            for (int i = 0; i < KPT; ++i)
                buf[tid + i * T] = make_float2(io[i].x, io[i].y);
            __syncthreads();
            mi_fft::fft_lds<LS, false>(buf, scr, rf.ft, tid);
        }
        MI_SPROBE(2);
        const float scale = 1.0f / float(2 * M);
        const bool keep = next2 && (outs || !FULL);                         // (nobody reads the last two blocks' images)
        // ONE pass over the transform's output: the pair is split into the image's bins, the image goes to the ring, and what
        // goes through the inverse transform -- the output wave's Hs_0 x, or what the frame's blocks owe block k + 1 (p = 1
        // is this block's own image) -- is formed and merged in place.  (Round 3, first form: forward with its split pass,
        // the image read back, the products, a pass to put them into LDS, inverse with its merge pass -- five barriers and
        // five trips through LDS more on a lone wave's path.)
        mi_fft::real_split_filter_merge<LS>(buf, rf.rt, tid,
            [&](int i, int k, float2 x0, bool partner) -> float2 {
                const int slot = partner ? i + KPT / 2 : i;
                if (keep)
                    rg[size_t(kblk) * M + k] = x0;
                if (outs)
                    return image_mul(x0, h0reg[slot], k);
                float2 t = make_float2(0.0f, 0.0f);
                #pragma unroll
                for (int p = 1; p <= PMAX; ++p)
                    if (p <= pmax)
                    {
                        const float2 h = hreg[p - 1][slot];
                        const float2 x = (p == 1) ? x0 : xreg[p - 1][slot];
                        if (k == 0)                                         // bin 0 packs (DC, Nyquist)
                        {
                            t.x = fmaf(x.x, h.x, t.x);
                            t.y = fmaf(x.y, h.y, t.y);
                        }
                        else
                        {
                            t.x = fmaf(x.x, h.x, fmaf(-x.y, h.y, t.x));
                            t.y = fmaf(x.x, h.y, fmaf(x.y, h.x, t.y));
                        }
                    }
                return t;
            }, [] {});
        MI_SPROBE(3);
        if (!FULL && !next1)
            return;                                                         // the frame's last block: the commit settles the rest
        if constexpr (REG)
            mi_fft::fft_lds<LS, true, false, true>(buf, scr, rf.ft, tid, io);
        else
        {
            mi_fft::fft_lds<LS, true>(buf, scr, rf.ft, tid);
            #pragma unroll
            for (int i = 0; i < KPT; ++i)
                io[i] = v2f{buf[tid + i * T].x, buf[tid + i * T].y};
        }
        MI_SPROBE(4);
        // (io[i]: pair n = tid + i T of the first half, io[i + KPT / 2]: pair n + M / 2 of the second)
        if (outs)
        {
            float *o = out + size_t(ch) * out_stride;
            #pragma unroll
            for (int i = 0; i < KPT / 2; ++i)
            {
                const int n = tid + i * T;
                const float2 p0 = accv[i];
                o[2 * n]     = fmaf(io[i].x, scale, p0.x);
                o[2 * n + 1] = fmaf(io[i].y, scale, p0.y);
                yx[n] = make_float2(io[i + KPT / 2].x, io[i + KPT / 2].y);
            }
        }
        if (FULL)
            __syncthreads();                                                // yx is there (a lone output wave passes at once)
        MI_SPROBE(5);
        if (!debt)
            return;
        #pragma unroll
        for (int i = 0; i < KPT / 2; ++i)
        {
            const int n = tid + i * T;
            const float2 t0 = make_float2(io[i].x, io[i].y), t1 = make_float2(io[i + KPT / 2].x, io[i + KPT / 2].y);
            const float2 yhi = FULL ? yx[n] : make_float2(0.0f, 0.0f);
            float2 *a1 = reinterpret_cast<float2 *>(a + SBK + 2 * n), *a2 = reinterpret_cast<float2 *>(a + 2 * SBK + 2 * n);
            const float2 v1 = accv[i];
            *a1 = make_float2(fmaf(t0.x + yhi.x, scale, v1.x), fmaf(t0.y + yhi.y, scale, v1.y));
            if (next2)
            {
                const float2 v2 = accw[i];
                *a2 = make_float2(fmaf(t1.x, scale, v2.x), fmaf(t1.y, scale, v2.y));
            }
        }
        MI_SPROBE(6);
    }

    // frame complete after partial calls: its spectrum enters the ring, acc moves on by one frame
    // SETTLE (banks whose sub-frame calls go through conv_small_kernel): nothing of this frame has been added beyond its end
    // yet -- its whole spill through the head partition, IFFT(H_0 X_F)[B, 2B), is added here.
    template <int LOGM, bool SETTLE>
    __global__ __launch_bounds__(fplan<LOGM>::T)
    void conv_commit_kernel(const float *frame, float2 *ring, int R, int slot, float *acc,
                            const float2 *__restrict__ tw, const float2 *__restrict__ H, int P)
    {
        using PL = fplan<LOGM>;
        constexpr int M = PL::N, T = PL::T, B = M;
        __shared__ float2 lds_[fplan<LOGM>::LDS];
        float2 *const buf = lds_, *const scr = lds_ + fplan<LOGM>::SCR;
        const int ch = blockIdx.x, tid = threadIdx.x;
        if (R > 0 || SETTLE)
        {
            typename fplan<LOGM>::real rf;
            rf.load(tw, TWN, tid);
            rf.prepare();
            load_and_forward<LOGM>(buf, scr, frame + size_t(ch) * B, B, true, rf, tid);
            float2 *rdst = ring + (size_t(ch) * R + slot) * M;
            const float2 *h0 = H + size_t(ch) * P * M;
            for (int k = tid; k < M; k += T)
            {
                const float2 xk = buf[k];
                if (R > 0)
                    rdst[k] = xk;
                if (SETTLE)
                    buf[k] = image_mul(xk, h0[k], k);
            }
            if (SETTLE)
            {
                __syncthreads();
                rf.inverse(buf, scr, tid);
            }
        }
        const float scale = 1.0f / float(2 * M);
        float2 *a = reinterpret_cast<float2 *>(acc + size_t(ch) * 2 * B);
        for (int n = tid; n < M / 2; n += T)
        {
            float2 v = a[n + M / 2];
            if (SETTLE)
            {
                const float2 y1 = buf[n + M / 2];
                v = make_float2(fmaf(y1.x, scale, v.x), fmaf(y1.y, scale, v.y));
            }
            a[n] = v;
            a[n + M / 2] = make_float2(0.0f, 0.0f);
        }
    }

    // rows of the named channels (all when `only` is NULL) from one [channels][n] array to another
    __global__ __launch_bounds__(256)
    void conv_copy_rows_kernel(float *dst, const float *src, size_t dst_pitch, size_t src_pitch, uint32_t n,
                               const uint8_t *__restrict__ only)
    {
        const uint32_t ch = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
        if (i < n && (only == nullptr || only[ch] != 0))
            dst[size_t(ch) * dst_pitch + i] = src[size_t(ch) * src_pitch + i];
    }

    // ---- one-frame cross-fade between two impulse responses (Equalizer "smooth" retune, Equalizer.cpp:486-501) ----------
    // Weight of the NEW response at position n of the frame's 2B-long result: 0 before B/2, a linear ramp i/B over the
    // next B positions (dsp::lramp1 / lramp_add2 with delta = 1/B), 1 from 3B/2 on; the old response gets 1 - that.
    __device__ __forceinline__ float xfade_new_weight(int n, int B)
    {
        const int i = n - B / 2;
        return (i <= 0) ? 0.0f : (i >= B) ? 1.0f : float(i) * (1.0f / float(B));
    }

    // whole frame, single partition (P == 1): y = w_old IFFT(X H_old) + w_new IFFT(X H_new), then the usual overlap-add
    template <int LOGM>
    __global__ __launch_bounds__(fplan<LOGM>::T)
    void conv_xfade_frame_kernel(float *out, const float *in, size_t out_stride, size_t in_stride, bool aligned,
                                 const float2 *__restrict__ Hold, const float2 *__restrict__ Hnew, float *acc,
                                 const float2 *__restrict__ tw, const uint8_t *__restrict__ xmask)
    {
        using PL = fplan<LOGM>;
        constexpr int M = PL::N, T = PL::T, B = M, KPT = M / T;
        __shared__ float2 lds_[fplan<LOGM>::LDS];
        float2 *const buf = lds_, *const scr = lds_ + fplan<LOGM>::SCR;
        const int ch = blockIdx.x, tid = threadIdx.x;
        typename fplan<LOGM>::real rf;
        rf.load(tw, TWN, tid);
        rf.prepare();
        load_and_forward<LOGM>(buf, scr, in + size_t(ch) * in_stride, B, aligned, rf, tid);
        float2 X[KPT], yo[KPT];
        #pragma unroll
        for (int i = 0; i < KPT; ++i)
        {
            const int k = tid + i * T;
            X[i] = buf[k];
            buf[k] = image_mul(X[i], Hold[size_t(ch) * M + k], k);
        }
        __syncthreads();
        rf.inverse(buf, scr, tid);
        #pragma unroll
        for (int i = 0; i < KPT; ++i)
        {
            const int k = tid + i * T;
            yo[i] = buf[k];
        }
        __syncthreads();
        #pragma unroll
        for (int i = 0; i < KPT; ++i)
        {
            const int k = tid + i * T;
            buf[k] = image_mul(X[i], Hnew[size_t(ch) * M + k], k);
        }
        __syncthreads();
        rf.inverse(buf, scr, tid);
        const float scale = 1.0f / float(2 * M);
        float *a = acc + size_t(ch) * 2 * B;
        float *o = out + size_t(ch) * out_stride;
        #pragma unroll
        for (int i = 0; i < KPT; ++i)
        {
            const int m = tid + i * T;                          // samples 2m, 2m+1 of the 2B-long result
            const float2 yn = buf[m];
            // a channel that takes no part in this cross-fade (its response did not change) keeps the old weights
            const bool fades = xmask[ch] != 0;
            const float w0 = fades ? xfade_new_weight(2 * m, B) : 0.0f, w1 = fades ? xfade_new_weight(2 * m + 1, B) : 0.0f;
            if (m < M / 2)                                      // first half: out = acc + y, acc <- next half
            {
                // the reference ramps its output buffer down, i.e. the old result TOGETHER WITH the overlap tail of the
                // blocks before it (Equalizer.cpp:496): same here
                const float2 a0 = *reinterpret_cast<const float2 *>(a + 2 * m);
                o[2 * m]     = (a0.x + yo[i].x * scale) * (1.0f - w0) + yn.x * scale * w0;
                o[2 * m + 1] = (a0.y + yo[i].y * scale) * (1.0f - w1) + yn.y * scale * w1;
            }
            else
            {
                const float y0 = (yo[i].x * (1.0f - w0) + yn.x * w0) * scale;
                const float y1 = (yo[i].y * (1.0f - w1) + yn.y * w1) * scale;
                const int n = 2 * m - B;
                const float2 a1 = *reinterpret_cast<const float2 *>(a + B + n);
                *reinterpret_cast<float2 *>(a + n)     = make_float2(a1.x + y0, a1.y + y1);
                *reinterpret_cast<float2 *>(a + B + n) = make_float2(0.0f, 0.0f);
            }
        }
    }

    // start of a cross-fade frame that arrives in pieces: the overlap tail already in acc fades with the old response
    // (see conv_xfade_frame_kernel), once
    __global__ __launch_bounds__(256)
    void conv_xfade_prescale_kernel(float *acc, int B, const uint8_t *__restrict__ xmask)
    {
        const int ch = blockIdx.y, n = B / 2 + blockIdx.x * 256 + threadIdx.x;
        if (n < B && xmask[ch] != 0)
            acc[size_t(ch) * 2 * B + n] *= 1.0f - xfade_new_weight(n, B);
    }

    // partial call inside the cross-fade frame: both head responses in the time domain, mixed by output position
    __global__ __launch_bounds__(256)
    void conv_direct_xfade_kernel(float *out, size_t out_stride, float *acc, const float *frame,
                                  const float *__restrict__ h0_old, const float *__restrict__ h0_new, int B, int off, int cnt,
                                  const uint8_t *__restrict__ xmask)
    {
        extern __shared__ float sxin[];                         // cnt samples of this call
        const int ch = blockIdx.y, tid = threadIdx.x;
        const float *x = frame + size_t(ch) * B + off;
        for (int j = tid; j < cnt; j += 256)
            sxin[j] = x[j];
        __syncthreads();
        const int i = blockIdx.x * 256 + tid;                   // output position relative to `off`
        if (i >= cnt + B - 1)
            return;
        const float *ho = h0_old + size_t(ch) * B, *hn = h0_new + size_t(ch) * B;
        const int jlo = (i - (B - 1) > 0) ? i - (B - 1) : 0;
        const int jhi = (i < cnt - 1) ? i : cnt - 1;
        float so = 0.0f, sn = 0.0f;
        for (int j = jlo; j <= jhi; ++j)
        {
            so = fmaf(sxin[j], ho[i - j], so);
            sn = fmaf(sxin[j], hn[i - j], sn);
        }
        const float w = (xmask[ch] != 0) ? xfade_new_weight(off + i, B) : 0.0f;
        float *a = acc + size_t(ch) * 2 * B + off + i;
        const float v = *a + (so * (1.0f - w) + sn * w);
        *a = v;
        if (i < cnt)
            out[size_t(ch) * out_stride + i] = v;
    }

    // ---- twiddle table, one per device -------------------------------------------------------------------
    float2 *g_tw[64] = { nullptr };
    std::mutex g_tw_lock;               // banks may be created from several host threads

    int twiddle_table(const float2 **out)
    {
        int dev = 0;
        MI_HIP_CHECK(hipGetDevice(&dev));
        MI_REQUIRE(dev >= 0 && dev < 64, MI_EINVAL, "device index %d out of range", dev);
        std::lock_guard<std::mutex> guard(g_tw_lock);
        if (g_tw[dev] == nullptr)
        {
            // TWN entries exp(-2 pi i j / TWN), then the per-pass twiddle tables of the radix-16 core (fft16.h)
            const size_t total = size_t(TWN) + 2 * size_t(mi_fft16::table16_total());
            std::vector<float2> h(total);
            for (int j = 0; j < TWN; ++j)
            {
                const double a = -2.0 * M_PI * double(j) / double(TWN);
                h[j] = make_float2(float(std::cos(a)), float(std::sin(a)));
            }
            mi_fft16::table16_build(reinterpret_cast<float *>(h.data() + TWN));
            float2 *d = nullptr;
            MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d), total * sizeof(float2)));
            MI_HIP_CHECK(hipMemcpy(d, h.data(), total * sizeof(float2), hipMemcpyHostToDevice));
            g_tw[dev] = d;
        }
        *out = g_tw[dev];
        return MI_OK;
    }
} // namespace

namespace mi
{
    int fft_twiddles(const float2 **tw, int *twn)
    {
        *twn = TWN;
        return twiddle_table(tw);
    }
}

struct mi_convolver_bank
{
    uint32_t    channels = 0;
    uint32_t    rank = 0;           // reference rank after clamping (Convolver.cpp:87)
    int         logm = 0;           // log2 of the partition (= complex transform) size
    int         B = 0, P = 0, R = 0;
    uint32_t    taps = 0;           // longest IR of the bank (Convolver::data_size of that channel)
    float       phase = 0.0f;
    int         slot = 0;           // ring slot of the newest complete frame
    int         off = 0;            // samples already received of the current frame
    bool        live = false;       // false: count == 0, process() emits zeros (Convolver.cpp:219-223)
    bool        yt_pending = false; // d_yt holds a tail spectrum that has not been folded into acc yet
    bool        upper_zero = false; // acc[B:2B] holds zeros (true between whole frames: the frame kernel skips that half)
    uint32_t   *d_sync = nullptr;   // [done[channels] | seen[channels] | fault]: hand-over between the roles of conv_step_kernel
    // sub-frame calls of partitioned banks (conv_small_kernel): the head partition as a delay line of SB-sample blocks
    bool        small = false;      // P >= 2 and B >= 1024
    int         Ps = 0;             // small partitions of the head: B / SB
    int         logs = LOGS;        // log2 SB: 256-sample blocks; a frame of 256 (rank 9) takes two blocks of 128
    float2     *d_Hs = nullptr, *d_sring = nullptr;     // [channels][Ps][SB]: images of the small partitions / of the frame's blocks
    uint32_t   *h_fault = nullptr;  // host-mapped flag raised by a hand-over that timed out (read by process() without a sync)
    uint32_t   *d_fault_host = nullptr;     // its device address
    int         cus = 256;          // compute units of the device the bank lives on
    bool        one_launch = true;  // whole-frame steps as conv_step_kernel (gfx950, not switched off) or as two launches
    float2     *d_H = nullptr, *d_ring = nullptr, *d_yt = nullptr;
    float2     *d_ring_before = nullptr;    // the ring the bank was made with, once its first batch of frames has grown it (kept: see there)
    float      *d_acc = nullptr, *d_frame = nullptr, *d_h0 = nullptr;
    float2     *d_yts = nullptr;                        // [channels][BATCH_MAX][B]: what the inverse transforms of a batch of frames take (process_blocks)
    float      *d_acc_new = nullptr;                    // [channels][B]: the accumulator a batch leaves, on its way into d_acc
    // Single-partition banks (the equalizer's FIR) can change their responses while streaming, channel by channel, the
    // way a reference Equalizer object does (Equalizer.cpp:339-345,481-501): every object has a response in force (vConv),
    // a cross-fade target (vNewConv) and a flag that the target waits for the block that completes next (EF_XFADE).
    //   set_irs      -> vConv of the named channels            crossfade_irs -> vNewConv of the named channels + flag
    //   a block completes: it is convolved with vConv; flagged channels fade to vNewConv, which becomes their vConv.
    // A frame of this convolver is the reference's block on its way through, so it uses what was in force when it began:
    // responses live in a pool of four buffers, `cv` / `nv` are the ones standing for vConv / vNewConv, and the frame being
    // received keeps d_H / d_h0 (old) and d_Hx / d_h0x (new, cross-fade frames only) whatever arrives meanwhile.
    struct response
    {
        float2 *H = nullptr; float *h0 = nullptr;
        float4 *W = nullptr;        // single-partition banks of 4096-sample frames: the images as (alpha, beta) per bin (fft_wave.h);
                                    // made wherever H is written (refresh_wave_table), so it is never older than H
    };
    response    pool[4];
    int         cv = 0, nv = 0;             // pool index of vConv / vNewConv
    int         fr_old = 0, fr_new = -1;    // pool indices the open frame uses
    bool        frame_open = false;         // the frame's first samples have been taken (or its whole)
    std::vector<uint8_t> xf_wait;           // EF_XFADE per channel
    bool        xf_any = false;
    float2     *d_Hx = nullptr;             // = pool[fr_new] in a cross-fade frame
    float      *d_h0x = nullptr;
    bool        xfade_active = false;       // the frame being received is a cross-fade frame
    uint8_t    *d_xmask = nullptr;          // ... for these channels
    uint8_t    *d_only = nullptr;           // scratch: channel selection of a parse
    const float2 *d_tw = nullptr;
    std::vector<uint32_t> counts;
};

namespace
{
    #define MI_LOGM_SWITCH(logm, CALL)                  \
        switch (logm)                                   \
        {                                               \
            case 7:  { CALL(7);  break; }               \
            case 8:  { CALL(8);  break; }               \
            case 9:  { CALL(9);  break; }               \
            case 10: { CALL(10); break; }               \
            case 11: { CALL(11); break; }               \
            default: { CALL(12); break; }               \
        }

    // tail owed to the next frame, from the ring (the dominant, HBM-bound step)
    int launch_mac(mi_convolver_bank *b, hipStream_t st)
    {
        if (b->P <= 1)
            return MI_OK;
        const int M = b->B;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        mi::take_profile_events(&ev0, &ev1);
        // the images this launch reads: (P - 1) partitions of H and of the ring, M complex each, per channel
        const size_t working_set = size_t(b->channels) * size_t(b->P - 1) * size_t(M) * sizeof(float2) * 2;
        const bool nt = working_set > (size_t(256) << 20);
        if (nt)
            MI_LAUNCH(conv_mac_kernel<true>, dim3((M / 2 + 255) / 256, b->channels), dim3(256), 0, st, ev0, ev1,
                                  b->d_yt, b->d_ring, b->R, b->slot, b->d_H, b->P, M);
        else
            MI_LAUNCH(conv_mac_kernel<false>, dim3((M / 2 + 255) / 256, b->channels), dim3(256), 0, st, ev0, ev1,
                                  b->d_yt, b->d_ring, b->R, b->slot, b->d_H, b->P, M);
        MI_HIP_CHECK(hipGetLastError());
        b->yt_pending = true;
        return MI_OK;
    }

    // A whole frame from the caller's block: frame and tail roles in one launch when there is a tail (P >= 2)
    int launch_frame(mi_convolver_bank *b, float *o, const float *x, size_t out_stride, size_t in_stride, bool aligned, hipStream_t st,
                     bool fold = false)
    {
        if (b->R > 0)
            b->slot = (b->slot + 1) % b->R;
        if (b->P >= 2 && b->one_launch)
        {
            hipEvent_t ev0 = nullptr, ev1 = nullptr;
            mi::take_profile_events(&ev0, &ev1);
            const size_t working_set = size_t(b->channels) * size_t(b->P - 1) * size_t(b->B) * sizeof(float2) * 2;
            const bool nt = working_set > (size_t(256) << 20);
            uint32_t *done = b->d_sync, *seen = b->d_sync + b->channels, *fault = b->d_sync + 2 * size_t(b->channels);
            // One frame and one tail workgroup per CU run side by side; with more channels than CUs the frame workgroups
            // (lower indices, dispatched first) would take both places of every CU and the tail role would follow them
            // instead of overlapping: such banks go in launches of one CU-count of channels each.
            const int per_launch = (b->logm >= 12) ? b->cus : int(b->channels);   // (smaller transforms: several workgroups fit a CU anyway)
            for (int first = 0; first < int(b->channels); first += per_launch)
            {
                const int cnt = std::min(per_launch, int(b->channels) - first);
                hipEvent_t e0 = (first == 0) ? ev0 : nullptr, e1 = (first + cnt >= int(b->channels)) ? ev1 : nullptr;
                #define MI_CALL(LM) \
                    if (nt) MI_LAUNCH((conv_step_kernel<LM, true>), dim3(((cnt + 7) & ~7) + cnt), dim3(fplan<LM>::T), 0, st, e0, e1, \
                                      o, x, out_stride, in_stride, aligned, b->d_ring, b->R, b->slot, b->d_H, b->P, b->d_acc, b->d_yt, \
                                      b->yt_pending, b->d_tw, b->upper_zero, cnt, first, done, seen, fault, b->d_fault_host, fold); \
                    else    MI_LAUNCH((conv_step_kernel<LM, false>), dim3(((cnt + 7) & ~7) + cnt), dim3(fplan<LM>::T), 0, st, e0, e1, \
                                      o, x, out_stride, in_stride, aligned, b->d_ring, b->R, b->slot, b->d_H, b->P, b->d_acc, b->d_yt, \
                                      b->yt_pending, b->d_tw, b->upper_zero, cnt, first, done, seen, fault, b->d_fault_host, fold)
                MI_LOGM_SWITCH(b->logm, MI_CALL)
                #undef MI_CALL
                MI_HIP_CHECK(hipGetLastError());
            }
            b->yt_pending = !fold;
            b->upper_zero = !fold;                                      // (folded: the tail's inverse fills both halves of acc)
            return MI_OK;
        }
        hipEvent_t fe0 = nullptr, fe1 = nullptr;
        mi::take_profile_events(&fe0, &fe1);
        #define MI_CALL(LM) MI_LAUNCH((conv_frame_kernel<LM>), dim3(b->channels), dim3(fplan<LM>::T), 0, st, fe0, fe1, \
                                      o, x, out_stride, in_stride, aligned, b->d_ring, b->R, b->slot, \
                                      b->d_H, b->P, b->d_acc, b->yt_pending ? b->d_yt : nullptr, b->d_tw, \
                                      static_cast<float *>(nullptr), 0u, 0u, 0u, b->upper_zero)
        MI_LOGM_SWITCH(b->logm, MI_CALL)
        #undef MI_CALL
        MI_HIP_CHECK(hipGetLastError());
        b->yt_pending = false;
        b->upper_zero = true;
        return launch_mac(b, st);                                   // (nothing to do for P == 1)
    }

    // acc += IFFT(Yt): only the partial-call path needs the tail in the time domain
    int fold_pending(mi_convolver_bank *b, hipStream_t st)
    {
        if (!b->yt_pending)
            return MI_OK;
        #define MI_CALL(LM) hipLaunchKernelGGL((conv_tail_kernel<LM>), dim3(b->channels), dim3(fplan<LM>::T), 0, st, \
                                               b->d_acc, b->d_yt, b->d_tw)
        MI_LOGM_SWITCH(b->logm, MI_CALL)
        #undef MI_CALL
        MI_HIP_CHECK(hipGetLastError());
        b->yt_pending = false;
        b->upper_zero = false;
        return MI_OK;
    }

    // K whole frames (2 <= K <= BATCH_MAX, a power of two) of a partitioned bank at a frame boundary: three launches
    int launch_batch(mi_convolver_bank *b, float *const *out, const float *const *in, int K, size_t out_stride, size_t in_stride,
                     hipStream_t st)
    {
        const size_t cells = size_t(b->channels) * BATCH_MAX * size_t(b->B);
        if (b->d_yts == nullptr)
        {
            MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_yts), cells * sizeof(float2)));
            MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_acc_new), size_t(b->channels) * size_t(b->B) * sizeof(float)));
        }
        // The frames of a batch enter the ring BEFORE their tails are formed: the ring needs room for them next to the P - 2
        // frames those tails still take.  Grown once, at the first batch: the frames it holds move to the end of the new one.
        if (b->R < b->P - 1 + BATCH_MAX)
        {
            const int newR = b->P - 1 + BATCH_MAX;
            const size_t img = size_t(b->B) * sizeof(float2);
            float2 *grown = nullptr;
            MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&grown), size_t(b->channels) * newR * img));
            hipError_t e = hipMemsetAsync(grown, 0, size_t(b->channels) * newR * img, st);
            for (int j = 0; e == hipSuccess && j < b->R; ++j)              // frame -1 - j: slot (slot - j) mod R -> slot newR - 1 - j
                e = hipMemcpy2DAsync(grown + size_t(newR - 1 - j) * b->B, size_t(newR) * img,
                                     b->d_ring + size_t(((b->slot - j) % b->R + b->R) % b->R) * b->B, size_t(b->R) * img,
                                     img, b->channels, hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess)
                e = hipStreamSynchronize(st);
            if (e != hipSuccess)
            {
                (void)hipFree(grown);
                MI_HIP_CHECK(e);
            }
            // A graph captured on this bank earlier holds the old ring, its size and slot in its launches: the epoch tells
            // mi_dspu_graph_launch to refuse it (MI_ESTATE).  The old ring itself stays allocated until the bank goes (one
            // allocation, once per bank): a graph launched some other way, or one still in flight on another stream, then
            // reads and writes memory that is still the bank's instead of freed addresses.
            mi::bank_epoch_bump(b);
            b->d_ring_before = b->d_ring;
            b->d_ring = grown;
            b->R = newR;
            b->slot = newR - 1;
        }
        batch_args ba;
        ba.frames = K;
        bool aligned = (out_stride % 2 == 0) && (in_stride % 2 == 0);
        for (int k = 0; k < K; ++k)
        {
            ba.out[k] = out[k];
            ba.in[k] = in[k];
            aligned = aligned && ((reinterpret_cast<uintptr_t>(out[k]) | reinterpret_cast<uintptr_t>(in[k])) % 8 == 0);
        }
        const int M = b->B;
        #define MI_CALL(LM) hipLaunchKernelGGL((conv_batch_forward_kernel<LM>), dim3(b->channels, K), dim3(fplan<LM>::T), 0, st, \
                                               ba, in_stride, aligned, b->d_ring, b->R, b->slot, b->d_tw)
        switch (b->logm) { case 9: { MI_CALL(9); break; } case 10: { MI_CALL(10); break; } case 11: { MI_CALL(11); break; } default: { MI_CALL(12); break; } }
        #undef MI_CALL
        MI_HIP_CHECK(hipGetLastError());
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        mi::take_profile_events(&ev0, &ev1);                // the pass over the partitions is what the batch is about
        const dim3 tgrid(M / 2 / 256, b->channels);
        #define MI_TAIL(KK) MI_LAUNCH((conv_batch_tail_kernel<KK, true>), tgrid, dim3(256), 0, st, ev0, ev1, b->d_yts, b->d_yt, \
                                      b->yt_pending, b->d_ring, b->R, b->slot, b->d_H, b->P, M)
        switch (K) { case 2: { MI_TAIL(2); break; } case 4: { MI_TAIL(4); break; } case 8: { MI_TAIL(8); break; } default: { MI_TAIL(16); break; } }
        #undef MI_TAIL
        MI_HIP_CHECK(hipGetLastError());
        // (a workgroup per CU is enough for these: one run per channel from 256 channels on -- 16.9 against 17.2 / 17.5 / 18.2 us per
        // frame with 2 / 4 / 8 runs at C3 --, up to four per channel below that)
        const int want = std::max(1, std::min(4, int(256 / b->channels)));
        const int groups = std::min(K, want), per = (K + groups - 1) / groups;
        // one run per channel: the workgroup that takes the accumulator is the one that leaves it -- straight into d_acc, no finish launch
        const bool always_finish = mi::test_path("conv_batch_finish");      // (the finish launch banks of fewer than 256 channels take anyway)
        const bool direct = per >= K && !always_finish;
        float *const acc_out = direct ? b->d_acc : b->d_acc_new;
        const int acc_out_stride = direct ? 2 * M : M;
        const bool zero_upper = direct && !b->upper_zero;
        #define MI_CALL(LM) hipLaunchKernelGGL((conv_batch_frames_kernel<LM>), dim3(b->channels, (K + per - 1) / per), dim3(fplan<LM>::T), 0, st, \
                                               ba, per, out_stride, aligned, b->d_yts, b->d_acc, acc_out, acc_out_stride, zero_upper, b->d_tw, \
                                               b->upper_zero)
        switch (b->logm) { case 9: { MI_CALL(9); break; } case 10: { MI_CALL(10); break; } case 11: { MI_CALL(11); break; } default: { MI_CALL(12); break; } }
        #undef MI_CALL
        MI_HIP_CHECK(hipGetLastError());
        if (!direct)
        {
            hipLaunchKernelGGL(conv_batch_finish_kernel, dim3((2 * M + 255) / 256, b->channels), dim3(256), 0, st, b->d_acc, b->d_acc_new, M,
                               b->upper_zero);
            MI_HIP_CHECK(hipGetLastError());
        }
        b->slot = (b->slot + K) % b->R;
        b->yt_pending = true;
        b->upper_zero = true;
        return MI_OK;
    }
} // namespace

namespace mi
{
    bool convolver_takes_delayed_frame(const mi_convolver_bank_t *b, size_t samples)
    {
        return b != nullptr && b->live && b->off == 0 && samples == size_t(b->B) && !b->xfade_active && !b->xf_any;
    }

    // EF_XFADE is dropped without having happened (Equalizer.cpp:250,264,356: the equalizer left its FIR modes)
    void convolver_cancel_crossfade(mi_convolver_bank_t *b, const uint8_t *channels)
    {
        if (b == nullptr || !b->xf_any)
            return;
        bool left = false;
        for (uint32_t c = 0; c < b->channels; ++c)
        {
            if (channels == nullptr || channels[c] != 0)
                b->xf_wait[c] = 0;
            left = left || (b->xf_wait[c] != 0);
        }
        b->xf_any = left;
    }

    // what the launches of a call take by value from the host: ring slot, fill of the open frame, the responses in force
    uint64_t convolver_bank_positions(const void *bank)
    {
        const mi_convolver_bank *b = static_cast<const mi_convolver_bank *>(bank);
        uint64_t h = position_mix(uint64_t(b->slot), uint64_t(b->off));
        h = position_mix(h, (uint64_t(b->live) << 0) | (uint64_t(b->yt_pending) << 1) | (uint64_t(b->frame_open) << 2) |
                            (uint64_t(b->xf_any) << 3) | (uint64_t(b->xfade_active) << 4) | (uint64_t(b->upper_zero) << 5));
        h = position_mix(h, (uint64_t(uint32_t(b->cv)) << 0) | (uint64_t(uint32_t(b->nv)) << 8) | (uint64_t(uint32_t(b->fr_old)) << 16) |
                            (uint64_t(uint32_t(b->fr_new) & 0xff) << 24));
        return h;
    }

    int convolver_process_delayed_frame(mi_convolver_bank_t *b, float *out, const float *in, size_t out_stride,
                                        size_t in_stride, const delay_view &dl, hipStream_t st)
    {
        MI_REQUIRE(convolver_takes_delayed_frame(b, size_t(b->B)), MI_ESTATE, "convolver_process_delayed_frame: not at a plain frame boundary");
        b->d_H = b->pool[b->cv].H;                                      // the response in force as the frame begins
        b->d_h0 = b->pool[b->cv].h0;
        const uint32_t tail = (dl.head + dl.size - dl.delay) % dl.size;
        MI_REQUIRE((dl.size % 2 == 0) && (tail % 2 == 0) && (dl.head % 2 == 0) && dl.delay >= uint32_t(b->B) &&
                   size_t(dl.size - dl.delay) >= size_t(b->B), MI_EINVAL, "convolver_process_delayed_frame: delay line geometry");
        const bool aligned = ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(in)) % 8 == 0) &&
                             (out_stride % 2 == 0) && (in_stride % 2 == 0);
        if (b->R > 0)
            b->slot = (b->slot + 1) % b->R;
        hipEvent_t fe0 = nullptr, fe1 = nullptr;
        mi::take_profile_events(&fe0, &fe1);
        #define MI_CALL(LM) MI_LAUNCH((conv_frame_kernel<LM>), dim3(b->channels), dim3(fplan<LM>::T), 0, st, fe0, fe1, \
                                      out, in, out_stride, in_stride, aligned, b->d_ring, b->R, b->slot, \
                                      b->d_H, b->P, b->d_acc, b->yt_pending ? b->d_yt : nullptr, b->d_tw, \
                                      dl.ring, dl.size, tail, dl.head, b->upper_zero)
        MI_LOGM_SWITCH(b->logm, MI_CALL)
        #undef MI_CALL
        MI_HIP_CHECK(hipGetLastError());
        b->yt_pending = false;
        b->upper_zero = true;
        return launch_mac(b, st);
    }

    // pool[idx].H has just been written (stream order): its (alpha, beta) table follows.  Every channel: rows a parse did not
    // name were copied from the buffer it started from, and so is what is derived from them.
    int refresh_wave_table(mi_convolver_bank_t *b, int idx, hipStream_t st)
    {
        if (b->P != 1 || b->logm != 12)
            return MI_OK;
        mi_convolver_bank::response &r = b->pool[idx];
        if (r.W == nullptr)
            MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&r.W), size_t(b->channels) * mi_fftw::N * sizeof(float4)));
        hipLaunchKernelGGL(conv_wave_table_kernel, dim3(mi_fftw::N / 256, b->channels), dim3(256), 0, st, r.W, r.H, b->d_tw);
        MI_HIP_CHECK(hipGetLastError());
        return MI_OK;
    }

    // May a run of blocks ride conv_frames_wave_kernel?  Its waves take the blocks in no particular order, so: no output overlaps
    // an input; two outputs either do not overlap or are the SAME buffer at a distance of a multiple of WAVE_FRAMES blocks (a ring
    // of buffers: those blocks belong to one wave, which writes them in the blocks' order).
    static bool wave_run_ok(float *const *out, const float *const *in, size_t blocks, size_t span_o, size_t span_i)
    {
        struct iv { uintptr_t a, e; int idx; bool is_out; };
        std::vector<iv> v;
        v.reserve(2 * blocks);
        for (size_t k = 0; k < blocks; ++k)
        {
            v.push_back({reinterpret_cast<uintptr_t>(out[k]), reinterpret_cast<uintptr_t>(out[k] + span_o), int(k), true});
            v.push_back({reinterpret_cast<uintptr_t>(in[k]), reinterpret_cast<uintptr_t>(in[k] + span_i), int(k), false});
        }
        std::sort(v.begin(), v.end(), [](const iv &x, const iv &y) { return x.a != y.a ? x.a < y.a : x.idx < y.idx; });
        uintptr_t end_out = 0, end_in = 0, last_out = 0;
        int last_idx = 0;
        for (const iv &x : v)
        {
            if (x.is_out)
            {
                if (x.a < end_in)
                    return false;
                if (x.a < end_out && !(x.a == last_out && (x.idx - last_idx) % WAVE_FRAMES == 0))
                    return false;
                if (x.a != last_out || end_out == 0)
                {
                    last_out = x.a;
                    last_idx = x.idx;
                }
                end_out = std::max(end_out, x.e);
            }
            else
            {
                if (x.a < end_out)
                    return false;
                end_in = std::max(end_in, x.e);
            }
        }
        return true;
    }

    // true if `blocks` blocks of `samples` samples each can go as one launch of conv_frames_kernel: a single-partition bank at a
    // plain frame boundary, blocks of exactly one frame, transforms of 512 .. 8192 points, no cross-fade waiting
    bool convolver_takes_delayed_frames(const mi_convolver_bank_t *b, size_t samples)
    {
        return convolver_takes_delayed_frame(b, samples) && b->P == 1 && b->R == 0 && !b->yt_pending && b->logm >= 9 && b->logm <= 12;
    }

    int convolver_process_delayed_frames(mi_convolver_bank_t *b, float *const *out, const float *const *in, size_t blocks,
                                         size_t out_stride, size_t in_stride, const delay_view &dl, hipStream_t st, bool apart)
    {
        MI_REQUIRE(convolver_takes_delayed_frames(b, size_t(b->B)) && blocks >= 1 && blocks <= size_t(FRAMES_MAX_BLOCKS), MI_ESTATE,
                   "convolver_process_delayed_frames: not at a plain frame boundary of a single-partition bank");
        b->d_H = b->pool[b->cv].H;                                      // the response in force as the frames begin
        b->d_h0 = b->pool[b->cv].h0;
        const uint32_t tail = (dl.head + dl.size - dl.delay) % dl.size;
        MI_REQUIRE((dl.size % 2 == 0) && (tail % 2 == 0) && (dl.head % 2 == 0) && (b->B % 2 == 0) && dl.delay == uint32_t(b->B) &&
                   size_t(dl.size) >= 2 * size_t(b->B), MI_EINVAL, "convolver_process_delayed_frames: delay line geometry");
        frames_args fa;
        fa.blocks = int(blocks);
        bool aligned = (out_stride % 2 == 0) && (in_stride % 2 == 0);
        for (size_t k = 0; k < blocks; ++k)
        {
            fa.out[k] = out[k];
            fa.in[k] = in[k];
            aligned = aligned && ((reinterpret_cast<uintptr_t>(out[k]) | reinterpret_cast<uintptr_t>(in[k])) % 8 == 0);
        }
        hipEvent_t fe0 = nullptr, fe1 = nullptr;
        mi::take_profile_events(&fe0, &fe1);
        // 4096-sample blocks: a wave per block on the wave-resident transform (conv_frames_wave_kernel) -- if the buffers of the run
        // allow its blocks to be taken in any order (wave_run_ok: a run in place goes the workgroup's way)
        bool waves = b->logm == 12 && aligned && blocks >= 2 && b->pool[b->cv].W != nullptr && !mi::compat_bits();
        const size_t span_o = (b->channels - 1) * out_stride + size_t(b->B), span_i = (b->channels - 1) * in_stride + size_t(b->B);
        if (waves && (apart || wave_run_ok(out, in, blocks, span_o, span_i)))
        {
            MI_LAUNCH(conv_frames_wave_kernel, dim3(b->channels), dim3(64 * WAVE_FRAMES), 0, st, fe0, fe1,
                      fa, out_stride, in_stride, b->pool[b->cv].W, b->d_acc, b->d_tw, dl.ring, dl.size, tail, dl.head, b->upper_zero);
            MI_HIP_CHECK(hipGetLastError());
            b->upper_zero = true;
            return MI_OK;
        }
        #define MI_CALL(LM) MI_LAUNCH((conv_frames_kernel<LM>), dim3(b->channels), dim3(fplan<LM>::T), 0, st, fe0, fe1, \
                                      fa, out_stride, in_stride, aligned, b->d_H, b->d_acc, static_cast<const float2 *>(nullptr), \
                                      b->d_tw, dl.ring, dl.size, tail, dl.head, \
                                      b->upper_zero)
        switch (b->logm)
        {
            case 9:  { MI_CALL(9);  break; }
            case 10: { MI_CALL(10); break; }
            case 11: { MI_CALL(11); break; }
            default: { MI_CALL(12); break; }
        }
        #undef MI_CALL
        MI_HIP_CHECK(hipGetLastError());
        b->upper_zero = true;
        return MI_OK;
    }
} // namespace mi

extern "C" {

int mi_convolver_bank_create(mi_convolver_bank_t **bank, uint32_t channels, const float *irs, size_t ir_stride,
                             const uint32_t *counts, uint32_t count, uint32_t rank, float phase, void *stream)
{
    MI_REQUIRE(bank != nullptr, MI_EINVAL, "mi_convolver_bank_create: NULL result pointer");
    *bank = nullptr;
    MI_REQUIRE(channels > 0, MI_EINVAL, "mi_convolver_bank_create: channels must be > 0");
    MI_REQUIRE(mi_dspu_device_count() > 0, MI_ENODEV, "no HIP device available (there is no CPU fallback)");
    hipStream_t st = mi::as_stream(stream);

    mi_convolver_bank *b = new (std::nothrow) mi_convolver_bank();
    MI_REQUIRE(b != nullptr, MI_ENOMEM, "mi_convolver_bank_create: out of host memory");
    b->channels = channels;
    b->rank     = (rank < 8) ? 8 : (rank > 16) ? 16 : rank;             // CONVOLVER_RANK_MIN/MAX
    b->phase    = phase;
    b->counts.assign(channels, count);
    b->xf_wait.assign(channels, 0);
    uint32_t longest = 0;
    for (uint32_t c = 0; c < channels; ++c)
    {
        if (counts != nullptr)
            b->counts[c] = counts[c];
        longest = (b->counts[c] > longest) ? b->counts[c] : longest;
    }
    b->taps = longest;
    if (longest == 0)                                                    // Convolver.cpp:80-84
    {
        *bank = b;
        return MI_OK;
    }
    MI_REQUIRE(irs != nullptr && ir_stride >= longest, MI_EINVAL, "mi_convolver_bank_create: bad impulse responses");

    b->logm = int(b->rank) - 1;
    if (b->logm > LOGM_MAX) b->logm = LOGM_MAX;                          // partitions above 4096 are cut to 4096
    if (b->logm < LOGM_MIN) b->logm = LOGM_MIN;
    b->B = 1 << b->logm;
    b->P = int((longest + b->B - 1) / b->B);
    b->R = b->P - 1;
    b->live = true;

    int twn = 0;
    const int r = mi::fft_twiddles(&b->d_tw, &twn);
    if (r != MI_OK) { mi_convolver_bank_destroy(b); return r; }

    const size_t M = size_t(b->B);
    hipError_t e = hipSuccess;
    float *d_ir = nullptr;
    uint32_t *d_counts = nullptr;
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_H), size_t(channels) * b->P * M * sizeof(float2));
    if (e == hipSuccess && b->R > 0) e = hipMalloc(reinterpret_cast<void **>(&b->d_ring), size_t(channels) * b->R * M * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_yt), size_t(channels) * M * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_acc), size_t(channels) * 2 * M * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_frame), size_t(channels) * M * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_sync), (2 * size_t(channels) + 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(b->d_sync, 0, (2 * size_t(channels) + 1) * sizeof(uint32_t), st);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&b->h_fault), sizeof(uint32_t), hipHostMallocMapped);
    if (e == hipSuccess)
    {
        *b->h_fault = 0u;
        e = hipHostGetDevicePointer(reinterpret_cast<void **>(&b->d_fault_host), b->h_fault, 0);
    }
    if (e == hipSuccess)
    {
        // The one-launch step leans on gfx950's index-ordered dispatch (see conv_step_kernel): other parts, and anyone who
        // asks (MI_CONV_TWO_LAUNCH=1), get the frame kernel followed by the tail kernel.
        int dev = 0, n = 0;
        hipDeviceProp_t prop;
        e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
        if (e == hipSuccess)
        {
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
                b->cus = n;
            const char *knob = getenv("MI_CONV_TWO_LAUNCH");
            b->one_launch = (std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) && !(knob != nullptr && atoi(knob) != 0);
        }
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_h0), size_t(channels) * M * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_ir), size_t(channels) * b->P * M * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_counts), channels * sizeof(uint32_t));
    // host staging: every row zero-padded to P*B, only the channel's own `count` taps copied
    std::vector<float> stage;
    try { stage.assign(size_t(channels) * b->P * M, 0.0f); }
    catch (...) { (void)hipFree(d_ir); (void)hipFree(d_counts); mi_convolver_bank_destroy(b);
                  return mi::fail(MI_ENOMEM, "mi_convolver_bank_create: out of host memory"); }
    for (uint32_t c = 0; c < channels; ++c)
        std::memcpy(&stage[size_t(c) * b->P * M], irs + size_t(c) * ir_stride, size_t(b->counts[c]) * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(d_ir, stage.data(), stage.size() * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_counts, b->counts.data(), channels * sizeof(uint32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpy2DAsync(b->d_h0, M * sizeof(float), d_ir, size_t(b->P) * M * sizeof(float),
                                              M * sizeof(float), channels, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess)
    {
        #define MI_CALL(LM) hipLaunchKernelGGL((conv_parse_kernel<LM>), dim3(b->P, channels), dim3(fplan<LM>::T), 0, st, \
                                               b->d_H, d_ir, size_t(b->P) * M, d_counts, b->P, b->d_tw, (const uint8_t *)nullptr)
        MI_LOGM_SWITCH(b->logm, MI_CALL)
        #undef MI_CALL
        e = hipGetLastError();
    }
    if (e == hipSuccess && b->P >= 2 && b->B >= SB)
    {
        // (round 6: frames of 512 and 256 samples too -- ranks 10 and 9, Convolver.cpp:251-262 runs its doubling levels there:
        // two blocks of half a frame, so that a host that calls with half frames pays one launch per call)
        b->small = true;
        b->logs = (b->B >= 2 * SB) ? LOGS : LOGS - 1;
        b->Ps = b->B >> b->logs;
        const size_t cells = size_t(channels) * b->B;                       // Ps blocks of SB cells
        e = hipMalloc(reinterpret_cast<void **>(&b->d_Hs), cells * sizeof(float2));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_sring), cells * sizeof(float2));
        if (e == hipSuccess)
        {
            // the head partition's taps (zero padded rows of d_h0) as Ps small partitions
            if (b->logs == LOGS)
                hipLaunchKernelGGL((conv_parse_kernel<LOGS>), dim3(b->Ps, channels), dim3(fplan<LOGS>::T), 0, st,
                                   b->d_Hs, b->d_h0, M, (const uint32_t *)nullptr, b->Ps, b->d_tw, (const uint8_t *)nullptr);
            else
                hipLaunchKernelGGL((conv_parse_kernel<LOGS - 1>), dim3(b->Ps, channels), dim3(fplan<LOGS - 1>::T), 0, st,
                                   b->d_Hs, b->d_h0, M, (const uint32_t *)nullptr, b->Ps, b->d_tw, (const uint8_t *)nullptr);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_ir);
    (void)hipFree(d_counts);
    b->pool[0].H = b->d_H;
    b->pool[0].h0 = b->d_h0;
    if (e == hipSuccess && mi::refresh_wave_table(b, 0, st) != MI_OK)
        e = hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess)
    {
        mi_convolver_bank_destroy(b);
        return mi::fail(e == hipErrorOutOfMemory ? MI_ENOMEM : MI_EHIP, "mi_convolver_bank_create: %s", hipGetErrorString(e));
    }
    *bank = b;
    return mi_convolver_bank_reset(b, stream);
}

// parse `count` taps per channel (device rows) into partition images at dst_H and the head taps at dst_h0; `only`
// (HOST flags, NULL = every channel) limits the update to the named channels, the others keep their rows
static int parse_irs(mi_convolver_bank_t *b, const float *d_irs, size_t ir_stride, uint32_t count, float2 *dst_H,
                     float *dst_h0, const uint8_t *only, hipStream_t st)
{
    const size_t M = size_t(b->B);
    const uint8_t *d_only = nullptr;
    if (only != nullptr)
    {
        if (b->d_only == nullptr)
            MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_only), b->channels));
        MI_HIP_CHECK(hipMemcpyAsync(b->d_only, only, b->channels, hipMemcpyHostToDevice, st));
        MI_HIP_CHECK(hipStreamSynchronize(st));
        d_only = b->d_only;
    }
    // zero-padded staging rows [channels][P*B], then the same parse kernel as init
    float *d_ir = nullptr;
    uint32_t *d_counts = nullptr;
    MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d_ir), size_t(b->channels) * b->P * M * sizeof(float)));
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_counts), b->channels * sizeof(uint32_t));
    std::vector<uint32_t> cnt(b->channels, count);
    if (e == hipSuccess) e = hipMemsetAsync(d_ir, 0, size_t(b->channels) * b->P * M * sizeof(float), st);
    if (e == hipSuccess) e = hipMemcpy2DAsync(d_ir, size_t(b->P) * M * sizeof(float), d_irs, ir_stride * sizeof(float),
                                              size_t(count) * sizeof(float), b->channels, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_counts, cnt.data(), cnt.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
    {
        hipLaunchKernelGGL(conv_copy_rows_kernel, dim3(unsigned((M + 255) / 256), b->channels), dim3(256), 0, st,
                           dst_h0, d_ir, M, size_t(b->P) * M, uint32_t(M), d_only);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
    {
        #define MI_CALL(LM) hipLaunchKernelGGL((conv_parse_kernel<LM>), dim3(b->P, b->channels), dim3(fplan<LM>::T), 0, st, \
                                               dst_H, d_ir, size_t(b->P) * M, d_counts, b->P, b->d_tw, d_only)
        MI_LOGM_SWITCH(b->logm, MI_CALL)
        #undef MI_CALL
        e = hipGetLastError();
    }
    for (int i = 0; i < 4 && e == hipSuccess; ++i)
        if (b->pool[i].H == dst_H && mi::refresh_wave_table(b, i, st) != MI_OK)
            e = hipErrorOutOfMemory;
    if (e == hipSuccess && b->small && dst_h0 == b->d_h0)
    {
        if (b->logs == LOGS)
            hipLaunchKernelGGL((conv_parse_kernel<LOGS>), dim3(b->Ps, b->channels), dim3(fplan<LOGS>::T), 0, st,
                               b->d_Hs, dst_h0, M, (const uint32_t *)nullptr, b->Ps, b->d_tw, d_only);
        else
            hipLaunchKernelGGL((conv_parse_kernel<LOGS - 1>), dim3(b->Ps, b->channels), dim3(fplan<LOGS - 1>::T), 0, st,
                               b->d_Hs, dst_h0, M, (const uint32_t *)nullptr, b->Ps, b->d_tw, d_only);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_ir);
    (void)hipFree(d_counts);
    if (e != hipSuccess)
        return mi::fail(e == hipErrorOutOfMemory ? MI_ENOMEM : MI_EHIP, "convolver: parsing impulse responses: %s", hipGetErrorString(e));
    return MI_OK;
}

namespace
{
    bool used_by_frame(const mi_convolver_bank *b, int idx)
    {
        return b->frame_open && (idx == b->fr_old || (b->xfade_active && idx == b->fr_new));
    }

    // A pool buffer that vConv (which == 0) or vNewConv (which == 1) may be written into: its own one when nobody else
    // holds it, otherwise a free one that starts as a copy of `base` (so that the channels a call does not name keep
    // their rows).  Four buffers always leave one free: the open frame holds at most two, the other slot one.
    int writable_response(mi_convolver_bank *b, int which, int base, hipStream_t st, int *out)
    {
        const int own = which ? b->nv : b->cv, other = which ? b->cv : b->nv;
        int idx = own;
        if (own == other || used_by_frame(b, own))
        {
            idx = -1;
            for (int i = 0; i < 4 && idx < 0; ++i)
                if (i != own && i != other && !used_by_frame(b, i))
                    idx = i;
            MI_REQUIRE(idx >= 0, MI_ESTATE, "convolver: no free response buffer");
        }
        const size_t M = size_t(b->B), nH = size_t(b->channels) * b->P * M, nh = size_t(b->channels) * M;
        if (b->pool[idx].H == nullptr)
        {
            MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->pool[idx].H), nH * sizeof(float2)));
            MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->pool[idx].h0), nh * sizeof(float)));
        }
        if (idx != base)
        {
            MI_HIP_CHECK(hipMemcpyAsync(b->pool[idx].H, b->pool[base].H, nH * sizeof(float2), hipMemcpyDeviceToDevice, st));
            MI_HIP_CHECK(hipMemcpyAsync(b->pool[idx].h0, b->pool[base].h0, nh * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        *out = idx;
        return MI_OK;
    }
}

int mi_convolver_bank_set_irs_device(mi_convolver_bank_t *b, const float *d_irs, size_t ir_stride, uint32_t count,
                                     const uint8_t *channels, void *stream)
{
    MI_REQUIRE(b != nullptr && b->live, MI_ESTATE, "mi_convolver_bank_set_irs_device: bank is not initialised");
    MI_REQUIRE(d_irs != nullptr && count > 0 && ir_stride >= count, MI_EINVAL, "mi_convolver_bank_set_irs_device: bad argument");
    MI_REQUIRE(count <= uint32_t(b->P) * uint32_t(b->B), MI_EINVAL,
               "mi_convolver_bank_set_irs_device: %u taps exceed the %d x %d the bank was created for", count, b->P, b->B);
    hipStream_t st = mi::as_stream(stream);
    for (uint32_t c = 0; c < b->channels; ++c)
        if (channels == nullptr || channels[c] != 0)
            b->counts[c] = count;
    b->taps = std::max(b->taps, count);
    if (b->P > 1)                                                       // partitioned banks: in place, input history kept
        return parse_irs(b, d_irs, ir_stride, count, b->d_H, b->d_h0, channels, st);
    // vConv of the named channels.  The frame being received keeps the response it started with (the reference convolves a
    // block as a whole with the response in force when the block completes, i.e. when this frame began).
    int idx = 0;
    int r = writable_response(b, 0, b->cv, st, &idx);
    if (r == MI_OK)
        r = parse_irs(b, d_irs, ir_stride, count, b->pool[idx].H, b->pool[idx].h0, channels, st);
    if (r != MI_OK)
        return r;
    b->cv = idx;
    if (b->xf_any)
    {
        // A cross-fade is waiting.  For the channels it is waiting for, vNewConv stays what it is -- after the fade that
        // OLDER response is back in force (Equalizer.cpp:491: vConv <- vNewConv); the other channels have no fade ahead,
        // their row of the fade target has to follow vConv
        std::vector<uint8_t> follow(b->channels, 0);
        bool any = false;
        for (uint32_t c = 0; c < b->channels; ++c)
            if ((channels == nullptr || channels[c] != 0) && !b->xf_wait[c])
                any = follow[c] = 1;
        if (any)
            r = parse_irs(b, d_irs, ir_stride, count, b->pool[b->nv].H, b->pool[b->nv].h0, follow.data(), st);
    }
    else
        b->nv = b->cv;
    return r;
}

int mi_convolver_bank_crossfade_irs_device(mi_convolver_bank_t *b, const float *d_irs, size_t ir_stride, uint32_t count,
                                           const uint8_t *channels, void *stream)
{
    MI_REQUIRE(b != nullptr && b->live, MI_ESTATE, "mi_convolver_bank_crossfade_irs_device: bank is not initialised");
    MI_REQUIRE(d_irs != nullptr && count > 0 && ir_stride >= count, MI_EINVAL, "mi_convolver_bank_crossfade_irs_device: bad argument");
    MI_REQUIRE(b->P == 1, MI_EINVAL, "mi_convolver_bank_crossfade_irs_device: only for single-partition banks (taps <= frame)");
    MI_REQUIRE(count <= uint32_t(b->B), MI_EINVAL, "mi_convolver_bank_crossfade_irs_device: %u taps exceed the frame of %d", count, b->B);
    hipStream_t st = mi::as_stream(stream);
    // vNewConv of the named channels; with no fade waiting yet the target starts as a copy of what is in force
    int idx = 0;
    int r = writable_response(b, 1, b->xf_any ? b->nv : b->cv, st, &idx);
    if (r == MI_OK)
        r = parse_irs(b, d_irs, ir_stride, count, b->pool[idx].H, b->pool[idx].h0, channels, st);
    if (r != MI_OK)
        return r;
    b->nv = idx;
    for (uint32_t c = 0; c < b->channels; ++c)
        if (channels == nullptr || channels[c] != 0)
        {
            b->counts[c] = count;
            b->xf_wait[c] = 1;
        }
    b->taps = std::max(b->taps, count);
    b->xf_any = true;
    if (b->d_xmask == nullptr)
        MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_xmask), b->channels));
    return MI_OK;
}

int mi_convolver_bank_faults(mi_convolver_bank_t *b, uint32_t *count, void *stream)
{
    MI_REQUIRE(b != nullptr && count != nullptr, MI_EINVAL, "mi_convolver_bank_faults: bad argument");
    *count = 0;
    if (b->d_sync == nullptr)
        return MI_OK;
    hipStream_t st = mi::as_stream(stream);
    MI_HIP_CHECK(hipMemcpyAsync(count, b->d_sync + 2 * size_t(b->channels), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    MI_HIP_CHECK(hipStreamSynchronize(st));
    if (*count != 0)
        return mi::fail(MI_EHIP, "mi_convolver_bank_faults: %u hand-over(s) between the roles of a frame step timed out; "
                                 "the bank's output is invalid from that frame on until mi_convolver_bank_reset", *count);
    return MI_OK;
}

int mi_convolver_bank_destroy(mi_convolver_bank_t *b)
{
    if (b == nullptr)
        return MI_OK;
    if (b->pool[0].H == nullptr)                                        // creation failed before the pool took them over
    {
        (void)hipFree(b->d_H);
        (void)hipFree(b->d_h0);
    }
    for (mi_convolver_bank::response &r : b->pool)
    {
        (void)hipFree(r.H);
        (void)hipFree(r.h0);
        (void)hipFree(r.W);
    }
    (void)hipFree(b->d_ring); (void)hipFree(b->d_ring_before); (void)hipFree(b->d_yt); (void)hipFree(b->d_acc); (void)hipFree(b->d_frame);
    (void)hipFree(b->d_xmask); (void)hipFree(b->d_only); (void)hipFree(b->d_sync);
    (void)hipFree(b->d_Hs); (void)hipFree(b->d_sring);
    (void)hipFree(b->d_yts); (void)hipFree(b->d_acc_new);
    mi::bank_epoch_forget(b);
    const bool faulted = (b->h_fault != nullptr) && (*static_cast<volatile uint32_t *>(b->h_fault) != 0u);
    (void)hipHostFree(b->h_fault);
    delete b;
    if (faulted)
        return mi::fail(MI_EHIP, "mi_convolver_bank_destroy: a hand-over between the roles of a frame step had timed out; "
                                 "output of the bank after that frame was invalid");
    return MI_OK;
}

int mi_convolver_bank_reset(mi_convolver_bank_t *b, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_convolver_bank_reset: NULL bank");
    if (!b->live)
        return MI_OK;
    hipStream_t st = mi::as_stream(stream);
    const size_t M = size_t(b->B);
    if (b->R > 0)
        MI_HIP_CHECK(hipMemsetAsync(b->d_ring, 0, size_t(b->channels) * b->R * M * sizeof(float2), st));
    MI_HIP_CHECK(hipMemsetAsync(b->d_acc, 0, size_t(b->channels) * 2 * M * sizeof(float), st));
    b->upper_zero = true;
    MI_HIP_CHECK(hipMemsetAsync(b->d_frame, 0, size_t(b->channels) * M * sizeof(float), st));
    b->slot = 0;
    b->off  = 0;
    b->yt_pending = false;
    if (b->d_sync != nullptr)                                           // hand-over counters and fault records start over
        MI_HIP_CHECK(hipMemsetAsync(b->d_sync, 0, (2 * size_t(b->channels) + 1) * sizeof(uint32_t), st));
    if (b->h_fault != nullptr)
    {
        // a launch that is still waiting could raise the flag after this point: let the stream drain first (only when it is up)
        if (*static_cast<volatile uint32_t *>(b->h_fault) != 0u)
            MI_HIP_CHECK(hipStreamSynchronize(st));
        *static_cast<volatile uint32_t *>(b->h_fault) = 0u;
    }
    // the frame being received is dropped; what is in force (and a fade that is still waiting) stays as it is: a fade that
    // had begun was taken at the reference's block boundary, its target is the response in force already
    b->frame_open = false;
    b->xfade_active = false;
    b->d_H = b->pool[b->cv].H;
    b->d_h0 = b->pool[b->cv].h0;
    return MI_OK;
}

int mi_convolver_bank_info(const mi_convolver_bank_t *b, uint32_t *rank, uint32_t *frame, uint32_t *partitions,
                           uint32_t *data_size)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_convolver_bank_info: NULL bank");
    if (rank)       *rank = b->live ? b->rank : 0;              // Convolver::rank() is 0 when uninitialised
    if (frame)      *frame = uint32_t(b->B);
    if (partitions) *partitions = uint32_t(b->P);
    if (data_size)  *data_size = b->taps;
    return MI_OK;
}

// whether the rows a call writes lie apart from the rows it reads (as whole address ranges)
static bool rows_apart(const float *o, size_t out_stride, const float *x, size_t in_stride, size_t cnt, uint32_t channels)
{
    const uintptr_t o0 = reinterpret_cast<uintptr_t>(o), x0 = reinterpret_cast<uintptr_t>(x);
    const uintptr_t on = (size_t(channels - 1) * out_stride + cnt) * sizeof(float), xn = (size_t(channels - 1) * in_stride + cnt) * sizeof(float);
    return o0 + on <= x0 || x0 + xn <= o0;
}

int mi_convolver_bank_process(mi_convolver_bank_t *b, float *out, const float *in, size_t samples,
                              size_t out_stride, size_t in_stride, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_convolver_bank_process: NULL bank");
    if (samples == 0)
        return MI_OK;
    MI_REQUIRE(out != nullptr && in != nullptr, MI_EINVAL, "mi_convolver_bank_process: NULL buffer");
    MI_REQUIRE(out_stride >= samples && in_stride >= samples, MI_EINVAL, "mi_convolver_bank_process: stride shorter than the block");
    hipStream_t st = mi::as_stream(stream);
    {
        const int rc = mi::capture_touch(st, b, "convolver", mi::convolver_bank_positions);
        if (rc != MI_OK)
            return rc;
    }
    if (!b->live)                                                // Convolver.cpp:219-223
    {
        MI_HIP_CHECK(hipMemset2DAsync(out, out_stride * sizeof(float), 0, samples * sizeof(float), b->channels, st));
        return MI_OK;
    }
    // a hand-over inside an earlier frame step gave up (conv_step_kernel): what the bank has produced since is invalid
    MI_REQUIRE(b->h_fault == nullptr || *static_cast<volatile uint32_t *>(b->h_fault) == 0u, MI_EHIP,
               "mi_convolver_bank_process: a hand-over between the roles of an earlier frame step timed out "
               "(mi_convolver_bank_faults); the bank's output is invalid from that frame on until mi_convolver_bank_reset");
    const int B = b->B;
    size_t done = 0;
    while (done < samples)
    {
        const size_t left = samples - done;
        float *o = out + done;
        const float *x = in + done;
        if (b->P == 1 && b->off == 0 && !b->frame_open)                     // a frame begins: the reference's block completes
        {
            b->fr_old = b->cv;
            b->d_H = b->pool[b->cv].H;
            b->d_h0 = b->pool[b->cv].h0;
            if (b->xf_any)                                                  // flagged channels fade to vNewConv across this frame,
            {                                                               // which is their vConv from now on
                b->fr_new = b->nv;
                b->d_Hx = b->pool[b->nv].H;
                b->d_h0x = b->pool[b->nv].h0;
                MI_HIP_CHECK(hipMemcpyAsync(b->d_xmask, b->xf_wait.data(), b->channels, hipMemcpyHostToDevice, st));
                MI_HIP_CHECK(hipStreamSynchronize(st));
                b->xfade_active = true;
                b->cv = b->nv;
                std::fill(b->xf_wait.begin(), b->xf_wait.end(), uint8_t(0));
                b->xf_any = false;
            }
            b->frame_open = true;
        }
        if (b->off == 0 && left >= size_t(B) && b->xfade_active)
        {
            const bool aligned = ((reinterpret_cast<uintptr_t>(o) | reinterpret_cast<uintptr_t>(x)) % 8 == 0) &&
                                 (out_stride % 2 == 0) && (in_stride % 2 == 0);
            #define MI_CALL(LM) hipLaunchKernelGGL((conv_xfade_frame_kernel<LM>), dim3(b->channels), dim3(fplan<LM>::T), 0, st, \
                                                   o, x, out_stride, in_stride, aligned, b->d_H, b->d_Hx, b->d_acc, b->d_tw, b->d_xmask)
            MI_LOGM_SWITCH(b->logm, MI_CALL)
            #undef MI_CALL
            MI_HIP_CHECK(hipGetLastError());
            b->upper_zero = true;
            b->xfade_active = false;
            b->frame_open = false;
            done += size_t(B);
        }
        else if (b->off == 0 && left >= size_t(B))
        {
            const bool aligned = ((reinterpret_cast<uintptr_t>(o) | reinterpret_cast<uintptr_t>(x)) % 8 == 0) &&
                                 (out_stride % 2 == 0) && (in_stride % 2 == 0);
            const int r = launch_frame(b, o, x, out_stride, in_stride, aligned, st);
            if (r != MI_OK)
                return r;
            b->frame_open = false;
            done += size_t(B);
        }
        else
        {
            const int rf = fold_pending(b, st);
            if (rf != MI_OK)
                return rf;
            if (b->small)
            {
                // the head partition as a delay line of sb-sample blocks inside the frame (conv_small_kernel): bounded work
                const int sb = 1 << b->logs;
                int cnt;
                if ((b->off % sb) == 0 && left >= size_t(sb))
                {
                    cnt = sb;                                               // an aligned whole block: one launch
                    mi::note_launch("conv_small_kernel<true>");
                    if (b->logs == LOGS)
                        hipLaunchKernelGGL((conv_small_kernel<true>), dim3(b->channels), dim3(2 * fplan<LOGS>::T), 0, st,
                                           o, x, out_stride, in_stride, b->d_frame, B, b->off, b->d_sring, b->Ps, b->d_Hs, b->d_acc, b->d_tw);
                    else
                        hipLaunchKernelGGL((conv_small_kernel<true, LOGS - 1, 1>), dim3(b->channels), dim3(2 * fplan<LOGS - 1>::T), 0, st,
                                           o, x, out_stride, in_stride, b->d_frame, B, b->off, b->d_sring, b->Ps, b->d_Hs, b->d_acc, b->d_tw);
                    MI_HIP_CHECK(hipGetLastError());
                    b->off += cnt;
                }
                else
                {
                    const size_t room = size_t(sb - (b->off % sb));
                    cnt = int((left < room) ? left : room);
                    const bool apart = rows_apart(o, out_stride, x, in_stride, size_t(cnt), b->channels);
                    if (!apart)
                        hipLaunchKernelGGL(conv_file_kernel, dim3((cnt + 255) / 256, b->channels), dim3(256), 0, st, b->d_frame, B, b->off,
                                           x, in_stride, cnt);
                    mi::note_launch("conv_direct_kernel");
                    hipLaunchKernelGGL(conv_direct_kernel, dim3((cnt + sb - 1 + 255) / 256, b->channels), dim3(256),
                                       size_t(2 * cnt + 256) * sizeof(float), st, o, out_stride, b->d_acc, b->d_frame, b->d_h0, B, b->off, cnt, sb, B,
                                       apart ? x : nullptr, in_stride);
                    MI_HIP_CHECK(hipGetLastError());
                    b->off += cnt;
                    if ((b->off % sb) == 0)                                 // the block is complete: its image, and what the frame owes the next
                    {
                        if (b->logs == LOGS)
                            hipLaunchKernelGGL((conv_small_kernel<false>), dim3(b->channels), dim3(fplan<LOGS>::T), 0, st,
                                               (float *)nullptr, (const float *)nullptr, size_t(0), size_t(0), b->d_frame, B, b->off - sb,
                                               b->d_sring, b->Ps, b->d_Hs, b->d_acc, b->d_tw);
                        else
                            hipLaunchKernelGGL((conv_small_kernel<false, LOGS - 1, 1>), dim3(b->channels), dim3(fplan<LOGS - 1>::T), 0, st,
                                               (float *)nullptr, (const float *)nullptr, size_t(0), size_t(0), b->d_frame, B, b->off - sb,
                                               b->d_sring, b->Ps, b->d_Hs, b->d_acc, b->d_tw);
                        MI_HIP_CHECK(hipGetLastError());
                    }
                }
                b->upper_zero = false;
                done += size_t(cnt);
            }
            else
            {
            const int cnt = int((left < size_t(B - b->off)) ? left : size_t(B - b->off));
            const dim3 grid((cnt + B - 1 + 255) / 256, b->channels);
            const bool apart = !b->xfade_active && rows_apart(o, out_stride, x, in_stride, size_t(cnt), b->channels);
            if (!apart)
                hipLaunchKernelGGL(conv_file_kernel, dim3((cnt + 255) / 256, b->channels), dim3(256), 0, st, b->d_frame, B, b->off,
                                   x, in_stride, cnt);
            if (b->xfade_active && b->off == 0)
                hipLaunchKernelGGL(conv_xfade_prescale_kernel, dim3((B / 2 + 255) / 256, b->channels), dim3(256), 0, st, b->d_acc, B, b->d_xmask);
            if (b->xfade_active)
                hipLaunchKernelGGL(conv_direct_xfade_kernel, grid, dim3(256), size_t(cnt) * sizeof(float), st,
                                   o, out_stride, b->d_acc, b->d_frame, b->d_h0, b->d_h0x, B, b->off, cnt, b->d_xmask);
            else
            {
                mi::note_launch("conv_direct_kernel");
                hipLaunchKernelGGL(conv_direct_kernel, grid, dim3(256), size_t(2 * cnt + 256) * sizeof(float), st,
                                   o, out_stride, b->d_acc, b->d_frame, b->d_h0, B, b->off, cnt, B, 2 * B, apart ? x : nullptr, in_stride);
            }
            MI_HIP_CHECK(hipGetLastError());
            b->upper_zero = false;
            b->off += cnt;
            done += size_t(cnt);
            }
            if (b->off == B && b->small && b->P >= 2 && b->one_launch && !b->yt_pending)
            {
                // A frame received in blocks completes like a whole frame minus its output (which went out block by
                // block): the frame role of the one-launch step transforms d_frame, hands the image to the tail role and
                // leaves IFFT(H_0 X)[B, 2B) + acc[B, 2B) as the new accumulator -- commit and tail side by side in one
                // launch instead of one after the other (14.7 + 38.6 us as two launches at C3).
                const int r = launch_frame(b, nullptr, b->d_frame, 0, size_t(B), true, st, true);
                if (r != MI_OK)
                    return r;
                b->off = 0;
                b->xfade_active = false;
                b->frame_open = false;
            }
            else if (b->off == B)
            {
                if (b->R > 0)
                    b->slot = (b->slot + 1) % b->R;
                #define MI_CALL(LM) \
                    if (b->small) hipLaunchKernelGGL((conv_commit_kernel<LM, true>), dim3(b->channels), dim3(fplan<LM>::T), 0, st, \
                                                     b->d_frame, b->d_ring, b->R, b->slot, b->d_acc, b->d_tw, b->d_H, b->P); \
                    else          hipLaunchKernelGGL((conv_commit_kernel<LM, false>), dim3(b->channels), dim3(fplan<LM>::T), 0, st, \
                                                     b->d_frame, b->d_ring, b->R, b->slot, b->d_acc, b->d_tw, b->d_H, b->P)
                MI_LOGM_SWITCH(b->logm, MI_CALL)
                #undef MI_CALL
                MI_HIP_CHECK(hipGetLastError());
                b->upper_zero = true;
                const int r = launch_mac(b, st);
                if (r != MI_OK)
                    return r;
                b->off = 0;
                b->xfade_active = false;
                b->frame_open = false;
            }
        }
    }
    return MI_OK;
}

// `blocks` consecutive process() calls in one C call.  Whole frames of a partitioned bank at a frame boundary go in batches of
// 16 / 8 / 4 / 2 frames (launch_batch: every partition's image is read once per batch instead of once per frame) -- the samples
// and the state left behind are those of the calls one by one (the one-launch frame step's, sum for sum).  Anything else
// (other block sizes, single-partition banks, a cross-fade under way, a block that reads what an earlier block of the call
// writes, captures) is what it says: a loop of process() calls.
int mi_convolver_bank_process_blocks(mi_convolver_bank_t *b, float *const *out, const float *const *in, size_t blocks, size_t samples,
                                     size_t out_stride, size_t in_stride, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_convolver_bank_process_blocks: NULL bank");
    if (samples == 0 || blocks == 0)
        return MI_OK;
    MI_REQUIRE(out != nullptr && in != nullptr, MI_EINVAL, "mi_convolver_bank_process_blocks: NULL pointer table");
    for (size_t k = 0; k < blocks; ++k)
        MI_REQUIRE(out[k] != nullptr && in[k] != nullptr, MI_EINVAL, "mi_convolver_bank_process_blocks: NULL buffer of block %zu", k);
    MI_REQUIRE(out_stride >= samples && in_stride >= samples, MI_EINVAL, "mi_convolver_bank_process_blocks: stride shorter than the block");
    hipStream_t st = mi::as_stream(stream);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = st != nullptr && hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    const bool per_frame = mi::test_path("conv_frame_per_launch");      // (the loop of calls that blocks which are not whole frames take anyway)
    const size_t ob = (size_t(b->channels - 1) * out_stride + samples) * sizeof(float), ib = (size_t(b->channels - 1) * in_stride + samples) * sizeof(float);
    auto overlap = [](const void *p, size_t pn, const void *q, size_t qn) -> bool {
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(p), b0 = reinterpret_cast<uintptr_t>(q);
        return a0 < b0 + qn && b0 < a0 + pn;
    };
    size_t k = 0;
    while (k < blocks)
    {
        size_t run = 0;
        // (one_launch: the batch is bit for bit the ONE-LAUNCH frame step's sums; the two-launch fall-back stays frame by frame)
        const bool batchable = b->live && b->one_launch && b->P >= 2 && b->R >= 1 && samples == size_t(b->B) && b->off == 0 && !b->frame_open &&
                               !b->xfade_active && !b->xf_any && b->logm >= 9 && b->logm <= 12 && !capturing && !per_frame &&
                               (b->h_fault == nullptr || *static_cast<volatile uint32_t *>(b->h_fault) == 0u);
        if (batchable)
        {
            // the frames of a batch are all read before any of them is written, and written side by side: a block joins unless
            // it reads or writes what an earlier block of the batch writes (its own output may be its input)
            run = 1;
            while (k + run < blocks && run < size_t(BATCH_MAX))
            {
                bool ok = true;
                for (size_t i = k; ok && i < k + run; ++i)
                    ok = !overlap(out[i], ob, in[k + run], ib) && !overlap(out[i], ob, out[k + run], ob);
                if (!ok)
                    break;
                ++run;
            }
            size_t K = 1;
            while (2 * K <= run)
                K *= 2;
            run = K;
        }
        if (run >= 2)
        {
            const int r = launch_batch(b, out + k, in + k, int(run), out_stride, in_stride, st);
            if (r != MI_OK)
                return r;
            k += run;
            continue;
        }
        const int r = mi_convolver_bank_process(b, out[k], in[k], samples, out_stride, in_stride, stream);
        if (r != MI_OK)
            return r;
        ++k;
    }
    return MI_OK;
}

} // extern "C"
