// TEST INFRASTRUCTURE (not product; nothing under lsp-dsp-units_amd/ links or loads this file).
//
// The oracle's recurrence (oracle/biquad_oracle.c: orc_biquad_cascade, the transposed direct form II section of
// FilterBank::process, /root/reference/src/main/filters/FilterBank.cpp:256-291) run SERIALLY on the device: one channel per
// lane, sample after sample, section after section, with the oracle's own operation order and without contraction into fused
// multiply-adds (-ffp-contract=off, like oracle/Makefile).  tests/test_biquad_gpu.py::test_device_twin_equals_the_oracle
// asserts its output and filter memory equal the CPU oracle's BIT FOR BIT: IEEE float32 multiplication and addition on
// gfx950 are the reference's arithmetic, and what separates the product's time-parallel kernel from the oracle is the
// ORDER of its operations only -- the product is then held to the noise rule against this on-device twin as well.
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace
{
    __global__ void serial_cascade_kernel(float *dst, const float *src, size_t n, size_t stride, const float *coef, float *state,
                                          const uint32_t *nsec, uint32_t max_sec, uint32_t channels)
    {
        const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
        if (c >= channels)
            return;
        float *out = dst + size_t(c) * stride;
        const float *in0 = src + size_t(c) * stride;
        const uint32_t ns = nsec[c];
        if (ns == 0)
        {
            for (size_t i = 0; i < n; ++i)
                out[i] = in0[i];
            return;
        }
        for (uint32_t s = 0; s < ns; ++s)
        {
            const float *q = coef + (size_t(c) * max_sec + s) * 5;
            float *st = state + (size_t(c) * max_sec + s) * 2;
            const float b0 = q[0], b1 = q[1], b2 = q[2], a1 = q[3], a2 = q[4];
            float d0 = st[0], d1 = st[1];
            const float *in = (s == 0) ? in0 : out;
            for (size_t i = 0; i < n; ++i)
            {
                const float x  = in[i];
                const float y  = b0 * x + d0;
                const float p1 = b1 * x + a1 * y;
                const float p2 = b2 * x + a2 * y;
                d0 = d1 + p1;
                d1 = p2;
                out[i] = y;
            }
            st[0] = d0;
            st[1] = d1;
        }
    }
}

// Host buffers in, host buffers out: dst/src [channels][n], coef [channels][max_sec][5], state [channels][max_sec][2] (updated).
// Returns 0, or the HIP error code.
extern "C" int twin_biquad_bank(float *dst, const float *src, size_t channels, size_t n, const float *coef, float *state,
                                const uint32_t *nsec, size_t max_sec)
{
    float *d_dst = nullptr, *d_src = nullptr, *d_coef = nullptr, *d_state = nullptr;
    uint32_t *d_nsec = nullptr;
    const size_t bytes = channels * n * sizeof(float), cb = channels * max_sec * 5 * sizeof(float), sb = channels * max_sec * 2 * sizeof(float);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_dst), bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_src), bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_coef), cb);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_state), sb);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_nsec), channels * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(d_src, src, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_coef, coef, cb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_state, state, sb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_nsec, nsec, channels * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess)
    {
        hipLaunchKernelGGL(serial_cascade_kernel, dim3(uint32_t((channels + 63) / 64)), dim3(64), 0, nullptr, d_dst, d_src, n, n,
                           d_coef, d_state, d_nsec, uint32_t(max_sec), uint32_t(channels));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(dst, d_dst, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(state, d_state, sb, hipMemcpyDeviceToHost);
    (void)hipFree(d_dst); (void)hipFree(d_src); (void)hipFree(d_coef); (void)hipFree(d_state); (void)hipFree(d_nsec);
    return int(e);
}
