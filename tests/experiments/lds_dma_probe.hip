// Probe (round 6): LDS-DMA on gfx950 into LDS addresses above 64 KiB, and what the instruction offset of a buffer_load ... lds
// adds to.  Build: hipcc --offload-arch=gfx950 -O2 lds_dma_probe.hip -o lds_dma_probe; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int WORDS = 38 * 1024;            // 152 KiB of LDS
__global__ __launch_bounds__(64) void probe(const float *src, float *out, int base_words, int mode)
{
    __shared__ float lds[WORDS];
    const int lane = threadIdx.x;
    for (int i = lane; i < WORDS; i += 64)
        lds[i] = -1.0f;
    __syncthreads();
    const unsigned dst = unsigned(uintptr_t(&lds[base_words]));      // LDS byte address of the target
    if (mode == 0)
    {
        // global_load_lds_dword: lane's own source address; LDS = M0 + lane * 4
        const float *g = src + lane;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
        const float *g2 = src + 64 + lane;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off offset:256\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g2 - 64), "s"(dst) : "memory");
    }
    else
    {
        // buffer_load_dword ... offen offset:256 lds: does the instruction offset move the LDS address as well?
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 4096, 0x00020000);
        unsigned keep;
        const int voff = lane * 4;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\t"
                     "buffer_load_dword %1, %2, 0 offen offset:256 lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(r), "s"(dst) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < WORDS; i += 64)
        out[i] = lds[i];
}

int main()
{
    float *src, *out;
    hipMalloc(&src, 4096);
    hipMalloc(&out, WORDS * 4);
    std::vector<float> h(1024), o(WORDS);
    for (int i = 0; i < 1024; ++i) h[i] = float(i + 1);
    hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    const int bases[] = {0, 15 * 1024, 17 * 1024, 30 * 1024, 37 * 1024};   // words: 0, 60 K, 68 K, 120 K, 148 KiB
    for (int mode = 0; mode < 2; ++mode)
        for (int b : bases)
        {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out, b, mode);
            if (hipDeviceSynchronize() != hipSuccess) { printf("mode %d base %d: launch failed\n", mode, b); return 1; }
            hipMemcpy(o.data(), out, WORDS * 4, hipMemcpyDeviceToHost);
            int first = -1, count = 0, ok1 = 1, ok2 = 1;
            for (int i = 0; i < WORDS; ++i)
                if (o[i] != -1.0f) { if (first < 0) first = i; ++count; }
            for (int i = 0; i < 64; ++i) { ok1 &= (o[b + i] == float(i + 1)); ok2 &= (o[b + 64 + i] == float(64 + i + 1)); }
            printf("mode %d base word %6d (%3d KiB): %d words written, first at %d; row0 %s, row1 at +256 B %s\n", mode, b, b * 4 / 1024, count, first,
                   ok1 ? "ok" : "WRONG", ok2 ? "ok" : "WRONG");
        }
    return 0;
}
