// development probe: issue rates of the instructions K6's band kernel is made of (one MI355X, all CUs, W waves per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP 512
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 0.001f + i;
    double d0 = seed, d1 = seed + 1;
    const float m = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) a[i] = __builtin_fmaf(a[i], m, c);                       // v_fma_f32, 8 independent chains
                if (KIND == 1) a[i] = __builtin_amdgcn_rcpf(a[i]);                       // v_rcp_f32
                if (KIND == 2) a[i] = (a[i] > c) ? a[i] : m;                             // v_cmp + v_cndmask (vcc)
                if (KIND == 3) a[i] = a[i] - m;                                          // v_sub_f32
                if (KIND == 4) a[i] = fminf(a[i], m);                                    // v_min (+ canonicalize?)
            }
            if (KIND == 5) { a[0] = __builtin_fmaf(a[0], m, c); }                         // one dependent chain
            if (KIND == 6) {                                                              // f64 4x4x4 MFMA, two chains
                d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(d0, 1.0, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(d1, 1.0, d1, 0, 0, 0);
            }
            if (KIND == 7) { d0 = d0 + (double)a[0]; d1 = d1 + (double)a[1]; a[0] += 1.0f; a[1] += 1.0f; }  // cvt + add_f64
        }
    }
    float s = (float)(d0 + d1);
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, int per_iter, int wg_per_cu)
{
    float *out;
    const int grid = 256 * wg_per_cu;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200;
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: wg_per_cu waves (a 256-thread workgroup puts one wave on each SIMD), each per_iter * iters instructions
    const double instr_per_simd = (double)wg_per_cu * per_iter * iters;
    printf("%-28s waves/SIMD %d  %.3f ms  %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", name, wg_per_cu, ms,
           ms * 1e-3 * 2.4e9 / instr_per_simd);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32 x8 indep", REP, w);
        run<1>("v_rcp_f32 x8 indep", REP, w);
        run<2>("v_cmp+v_cndmask x8", REP * 2, w);
        run<3>("v_sub_f32 x8", REP, w);
        run<4>("v_min_f32 x8", REP, w);
        run<5>("v_fma_f32 dependent", REP / 8, w);
        run<6>("mfma_f64_4x4x4 x2", REP / 8 * 2, w);
        run<7>("cvt_f64+add_f64 x2", REP / 8 * 6, w);
    }
    return 0;
}
