// Shared host-side plumbing for the C-ABI implementation (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "mi_dspu.h"

namespace mi_meters { struct ilufs_epilogue; }

namespace mi
{
    // Thread-local message behind mi_dspu_last_error().
    char       *error_buffer();
    int         fail(int code, const char *fmt, ...);

    inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

    // Events armed by mi_dspu_profile_next_launch(); the next hot-path kernel launch of this thread
    // consumes them (hipExtLaunchKernelGGL records them at the kernel's own begin/end).
    void        take_profile_events(hipEvent_t *start, hipEvent_t *stop);
    // the hot-path kernel the calling thread launched last (MI_LAUNCH notes its name): mi_dspu_last_launch(), so that a test can
    // tell WHICH launch a call took, not only that its result is right
    void        note_launch(const char *kernel);
    // The library's environment switches, ALL of them (read at every call; documented in include/mi_dspu.h):
    //   MI_DSPU_COMPAT_BITS=1     runs of 4096-point blocks stay on the workgroup kernels -- the bits of block-by-block calls --
    //                             instead of the wave-resident transform kernels (within 1e-6 of them): compat_bits()
    //   MI_CONV_TWO_LAUNCH=1, MI_ILUFS_TWO_LAUNCHES=1   fall-backs behind two in-launch hand-overs that rest on gfx950 behaviour
    //   MI_DSPU_TEST_PATH=a,b,..  test hook: sends a call down ANOTHER LIVE path of the library (one that other inputs take anyway)
    //                             so that a differential test can hold the two against each other: test_path("a")
    bool        compat_bits();
    bool        test_path(const char *name);

    // hipGraph capture of a bank that keeps ring positions on the host (runtime.hip): called at the top of
    // its process() with a function that packs those positions; on a stream that is being captured the positions are
    // noted at the bank's first call and compared again at mi_dspu_graph_end_capture.  MI_OK when the stream is not
    // capturing; MI_ESTATE when it is captured behind the library's back.
    typedef uint64_t (*position_fn)(const void *bank);
    int         capture_touch(hipStream_t st, const void *bank, const char *what, position_fn fn);
    // A bank that re-makes device buffers its launches take by value (the convolver's ring at its first batch of frames) bumps
    // its epoch: graphs captured on it before are refused at mi_dspu_graph_launch (MI_ESTATE) instead of replaying stale addresses.
    void        bank_epoch_bump(const void *bank);
    void        bank_epoch_forget(const void *bank);        // at the bank's destruction
    uint64_t    delay_bank_positions(const void *bank);         // delay.hip
    uint64_t    convolver_bank_positions(const void *bank);     // convolver.hip
    uint64_t    spectral_bank_positions(const void *bank);      // spectral.hip
    inline uint64_t position_mix(uint64_t h, uint64_t v) { return (h ^ v) * 0x100000001b3ull + 0x9e3779b97f4a7c15ull; }

    // Launch of a hot-path kernel: the extended launch (which records the armed events at the kernel's own begin and end)
    // only when events are armed -- it is not allowed on a capturing stream; the plain launch is, so a steady-state
    // process() call can be captured into a hipGraph.
    #define MI_LAUNCH(kernel, grid, block, lds, st, ev0, ev1, ...) \
        do { \
            ::mi::note_launch(#kernel); \
            if ((ev0) != nullptr || (ev1) != nullptr) \
                hipExtLaunchKernelGGL(kernel, grid, block, lds, st, ev0, ev1, 0, __VA_ARGS__); \
            else \
                hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__); \
        } while (0)

    // A chain of biquad banks on the same block in one launch (biquad.hip): stage k runs bank k's sections in place on
    // the travelling signal, or on a branch of it, and writes its result to `out` if there is one.  Returns MI_OK when
    // the fused launch was issued, 1 when the call does not qualify (run the banks one by one then), < 0 on errors.
    struct biquad_chain_stage
    {
        mi_biquad_bank_t   *bank;
        float              *out;            // NULL: nothing written
        size_t              out_stride;
        int                 branch;         // 1: the travelling signal goes on unchanged, the result only goes to `out`
    };
    int         biquad_chain_process(const biquad_chain_stage *stages, int count, const float *in, size_t in_stride,
                                     size_t samples, hipStream_t st, bool long_calls_as_streams = true);
    // The same chain over `blocks` consecutive blocks (buffers of their own): runs of blocks go out as ONE launch
    // (biquad_stream_chain_kernel) where the blocks allow it, single blocks as biquad_chain_process.  slot[k]: which of a
    // block's `outs` outputs stage k writes (-1: none; stages[k].out is not looked at); block i's output s is
    // out[i * outs + s], all outputs with the row stride out_stride.  Returns like biquad_chain_process -- 1 before
    // anything has been issued.
    int         biquad_chain_process_blocks(const biquad_chain_stage *stages, const int *slot, int count, int outs,
                                            float *const *out, const float *const *in, size_t blocks, size_t samples,
                                            size_t out_stride, size_t in_stride, hipStream_t st);

    // A biquad bank over a block without an output (biquad.hip): sums[channel * 4 + s] += the sum of the squares of the
    // filtered samples of segment s = [seg_end[s - 1], seg_end[s]), seg_end[3] = samples.  The meters' weighting filter.
    // ep != NULL: the integrated loudness meter's bookkeeping of this call (ilufs_device.h) goes with the launch when the
    // call qualifies (*rode = true: the last workgroup of every meter does it), otherwise the caller's own kernel follows.
    void        biquad_bank_output_reread(mi_biquad_bank_t *bank, bool yes);   // process()'s output feeds the owner's next launch
    int         biquad_bank_sumsq(mi_biquad_bank_t *bank, const float *in, size_t in_stride, size_t samples,
                                  const uint32_t seg_end[3], float *sums, hipStream_t st,
                                  const mi_meters::ilufs_epilogue *ep = nullptr, bool *rode = nullptr);

    // The impulse response of every channel's cascade in the reference's own operation order (unfused, sample after
    // sample): what the Equalizer synthesises its FIR from.  biquad.hip.
    int         biquad_bank_reference_impulse_response(mi_biquad_bank_t *bank, float *out, size_t samples, size_t out_stride,
                                                       hipStream_t st);

    // Device twiddle table exp(-2 pi i j / twn), one per device, created on first use (convolver.hip).
    int         fft_twiddles(const float2 **tw, int *twn);
    // Complex transform of `channels` sequences of 2^rank points (rank 15 .. 18) through global memory (spectral.hip, the
    // four-step form N = 8192 x N2): src -> tmp -> dst, unnormalised either way; dst may be src.
    int         big_fft_run(bool inverse, float2 *dst, const float2 *src, float2 *tmp, uint32_t rank, uint32_t channels,
                            const float2 *tw, hipStream_t st);

    // Library-internal coupling of a delay line bank and a convolver bank (the Equalizer's FIR path): the convolver's
    // frame kernel pulls its frame straight out of the delay line and pushes the new samples into it, one launch
    // instead of two.  delay.hip / convolver.hip.
    struct delay_view
    {
        float      *ring;           // [channels][size]
        uint32_t    size, head;
        uint32_t    delay;          // common delay of all channels, UINT32_MAX if they differ
    };
    int         delay_bank_view(mi_delay_bank_t *bank, delay_view *view);
    void        delay_bank_advance(mi_delay_bank_t *bank, size_t samples);
    // true if the next `samples` of the bank are exactly one whole frame handled by the plain frame kernel
    void        convolver_cancel_crossfade(mi_convolver_bank_t *bank, const uint8_t *channels /* host flags or NULL */);
    bool        convolver_takes_delayed_frame(const mi_convolver_bank_t *bank, size_t samples);
    int         convolver_process_delayed_frame(mi_convolver_bank_t *bank, float *out, const float *in, size_t out_stride,
                                                size_t in_stride, const delay_view &dl, hipStream_t st);
    // the same for a run of blocks of one frame each in ONE launch (single-partition banks: the Equalizer's FIR); the delay
    // line is advanced by the caller by blocks x frame samples
    bool        convolver_takes_delayed_frames(const mi_convolver_bank_t *bank, size_t samples);
    int         convolver_process_delayed_frames(mi_convolver_bank_t *bank, float *const *out, const float *const *in, size_t blocks,
                                                 size_t out_stride, size_t in_stride, const delay_view &dl, hipStream_t st,
                                                 bool apart = false /* the caller vouches: the blocks' outputs are distinct and none overlaps an input */);
    constexpr size_t CONV_FRAMES_MAX = 128;         // blocks per launch
    // blocks of the next launch when `left` are left: a run of K blocks is K + 1 units of work for the eight waves of
    // conv_frames_wave_kernel's workgroups, so a full launch takes 127 (128 units: sixteen rounds and no seventeenth for one wave)
    inline size_t conv_frames_chunk(size_t left) { return (left <= CONV_FRAMES_MAX) ? left : CONV_FRAMES_MAX - 1; }
} // namespace mi

#if defined(__HIPCC__)
namespace mi
{
    // Write-through (sc1) stores through a raw buffer descriptor: output that nobody on this device reads again soon
    // leaves the XCD's L2 while the kernel is still running instead of in one write-back burst when it ends
    // (MI355X_MICROARCH.md, "publish-large").  Builtins, not inline asm: the compiler's vmcnt bookkeeping sees them.
    constexpr int BUFFER_DWORD3 = 0x00020000;               // raw buffer, 32-bit data format (gfx9 family)
    constexpr int CPOL_SC1 = 16;
    // ... and non-temporal on top (nt|sc1) for output that nobody on the DEVICE reads again at all (the caller's output
    // planes): the lines do not displace the state and tables the next launch wants from L2 / MALL.  Measured both ways
    // (profiles/r06_experiments/store_policy.txt): it pays for the biquad and splitter outputs; planes that the next
    // launch re-reads (C5's amp/data, the meters' sums) stay sc1.
    constexpr int CPOL_NT_SC1 = 18;
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

    __device__ __forceinline__ __amdgpu_buffer_rsrc_t wt_buffer(void *base, unsigned bytes)
    {
        return __builtin_amdgcn_make_buffer_rsrc(base, 0, int(bytes), BUFFER_DWORD3);
    }
    template <int POL = CPOL_SC1>
    __device__ __forceinline__ void wt_store(__amdgpu_buffer_rsrc_t rsrc, int byte_offset, float v)
    {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc, byte_offset, 0, POL);
    }
    template <int POL = CPOL_SC1>
    __device__ __forceinline__ void wt_store(__amdgpu_buffer_rsrc_t rsrc, int byte_offset, float2 v)
    {
        const u32x2 d = { __float_as_uint(v.x), __float_as_uint(v.y) };
        __builtin_amdgcn_raw_buffer_store_b64(d, rsrc, byte_offset, 0, POL);
    }
    template <int POL = CPOL_SC1>
    __device__ __forceinline__ void wt_store(__amdgpu_buffer_rsrc_t rsrc, int byte_offset, float4 v)
    {
        const u32x4 d = { __float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w) };
        __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, byte_offset, 0, POL);
    }
} // namespace mi
#endif

#define MI_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t mi_err__ = (expr);                                                   \
        if (mi_err__ != hipSuccess)                                                     \
            return ::mi::fail((mi_err__ == hipErrorNoDevice || mi_err__ == hipErrorInvalidDevice) \
                                  ? MI_ENODEV : MI_EHIP,                                \
                              "%s failed: %s (%s:%d)", #expr, hipGetErrorString(mi_err__), \
                              __FILE__, __LINE__);                                      \
    } while (0)

#define MI_REQUIRE(cond, code, ...)                                                     \
    do { if (!(cond)) return ::mi::fail((code), __VA_ARGS__); } while (0)
