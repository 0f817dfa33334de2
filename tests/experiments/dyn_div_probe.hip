// probe: how often dv() of dynfilter.hip (reciprocal + Newton step + one correction) differs from the IEEE quotient on
// the device, over operands in the range the cascade builders see
#include "../../lsp-dsp-units_amd/csrc/dynfilter.hip"
#include <cstdio>
#include <cstring>
__global__ void probe(const float *a, const float *b, unsigned long long *stats, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float q0 = a[i] / b[i], q1 = dv(a[i], b[i]);
    if (q0 != q1)
    {
        atomicAdd(&stats[0], 1ull);
        const int d = abs(__float_as_int(q0) - __float_as_int(q1));
        atomicMax(&stats[1], (unsigned long long)d);
    }
}
int main()
{
    const int n = 1 << 24;
    std::vector<float> a(n), b(n);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return double(st >> 11) / 9007199254740992.0; };
    for (int i = 0; i < n; ++i)
    {
        a[i] = float(exp((rnd() * 2.0 - 1.0) * log(1.0e4)));
        b[i] = float(exp((rnd() * 2.0 - 1.0) * log(1.0e4)));
        if (i & 1)
            a[i] = 1.0f;                        // half of the sample: reciprocals
    }
    float *da, *db; unsigned long long *ds, hs[2] = { 0, 0 };
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&ds, 16);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(ds, 0, 16);
    hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, da, db, ds, n);
    hipMemcpy(hs, ds, 16, hipMemcpyDeviceToHost);
    printf("dv() vs IEEE division on gfx950: %d pairs in [1e-4, 1e4]^2, %llu differ (%.3g per million), largest distance %llu ulp\n",
           n, hs[0], 1e6 * double(hs[0]) / n, hs[1]);
    return 0;
}
