// development probe: does v_rcp_f32 (quarter rate alone) overlap with plain VALU work of the same / other waves?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NF, int NR>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float a[8], r[4];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 0.001f + i;
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] = seed + 2.0f + i;
    const float m = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16; ++rep) {
#pragma unroll
            for (int i = 0; i < NF; i++) a[i % 8] = __builtin_fmaf(a[i % 8], m, c);
#pragma unroll
            for (int i = 0; i < NR; i++) r[i % 4] = __builtin_amdgcn_rcpf(r[i % 4]);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
#pragma unroll
    for (int i = 0; i < 4; i++) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NF, int NR>
void run(int wg_per_cu)
{
    float *out;
    const int grid = 256 * wg_per_cu;
    (void)hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 400;
    hipLaunchKernelGGL((k<NF, NR>), dim3(grid), dim3(256), 0, 0, out, 10, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NF, NR>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double groups = (double)wg_per_cu * 16 * iters;
    printf("fma %2d + rcp %d per group, waves/SIMD %d: %.1f cycles per group per SIMD (2.4 GHz)\n", NF, NR, wg_per_cu,
           ms * 1e-3 * 2.4e9 / groups);
    (void)hipFree(out);
}
int main()
{
    for (int w : {1, 4}) {
        run<16, 0>(w); run<0, 2>(w); run<0, 4>(w); run<16, 2>(w); run<16, 4>(w); run<8, 2>(w); run<24, 2>(w); run<32, 4>(w);
    }
    return 0;
}
