// Development probe: v_mfma_f32_4x4x1_16b_f32 as 16 independent 4 x 4 outer products per wave.
//   (1) layout + arithmetic: D_i of lane (block b, column j) against fmaf(A[lane 4 b + i], B[lane 4 b + j], C_i) -- bit for bit?
//   (2) issue rate: cycles per instruction per SIMD, alone and under VALU load.
//   hipcc --offload-arch=gfx950 -O2 -o build_dev/mfma_f32_4x4x1_probe scripts/dev/mfma_f32_4x4x1_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k_layout(const float *a, const float *b, const float *c, float *d)
{
    const int l = threadIdx.x;
    v4f acc = {c[4 * l], c[4 * l + 1], c[4 * l + 2], c[4 * l + 3]};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[4 * l + i] = acc[i];
}

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float *out, int iters)
{
    const int l = threadIdx.x;
    v4f acc0 = {0, 0, 0, 0}, acc1 = {1, 1, 1, 1}, acc2 = {2, 2, 2, 2}, acc3 = {3, 3, 3, 3};
    float x = (float)l * 1e-3f, y = 1.0f + x, f0 = x, f1 = y, f2 = x + 1, f3 = y + 1, f4 = 0.5f, f5 = 0.25f, f6 = 3.f, f7 = 4.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 1) {
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, x, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, y, acc3, 0, 0, 0);
        }
        if (MODE & 2) {
            f0 = __builtin_fmaf(f0, x, y); f1 = __builtin_fmaf(f1, x, y); f2 = __builtin_fmaf(f2, x, y); f3 = __builtin_fmaf(f3, x, y);
            f4 = __builtin_fmaf(f4, x, y); f5 = __builtin_fmaf(f5, x, y); f6 = __builtin_fmaf(f6, x, y); f7 = __builtin_fmaf(f7, x, y);
            f0 = __builtin_fmaf(f0, y, x); f1 = __builtin_fmaf(f1, y, x); f2 = __builtin_fmaf(f2, y, x); f3 = __builtin_fmaf(f3, y, x);
            f4 = __builtin_fmaf(f4, y, x); f5 = __builtin_fmaf(f5, y, x); f6 = __builtin_fmaf(f6, y, x); f7 = __builtin_fmaf(f7, y, x);
        }
    }
    out[blockIdx.x * 256 + l] = acc0[0] + acc1[1] + acc2[2] + acc3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

int main()
{
    float ha[64], hb[64], hc[256], hd[256], *a, *b, *c, *d;
    srand(7);
    int bad = 0, unfused = 0;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&c, 1024); hipMalloc(&d, 1024);
    for (int trial = 0; trial < 200; ++trial) {
        for (int i = 0; i < 64; ++i) { ha[i] = (float)rand() / RAND_MAX * 2 - 1; hb[i] = (float)rand() / RAND_MAX * 2 - 1; }
        for (int i = 0; i < 256; ++i) hc[i] = ((float)rand() / RAND_MAX * 2 - 1) * (trial & 1 ? 1.0f : 1e-3f);
        if (trial == 0) { ha[0] = 1e-30f; hb[0] = 1e-10f; hc[0] = 0.0f; }  // a denormal product
        hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice); hipMemcpy(c, hc, 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, a, b, c, d);
        hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) {
                const int blk = l / 4, j = l % 4;
                const float fused = fmaf(ha[4 * blk + i], hb[4 * blk + j], hc[4 * l + i]);
                const float prod = ha[4 * blk + i] * hb[4 * blk + j];
                const float two = prod + hc[4 * l + i];
                if (memcmp(&fused, &hd[4 * l + i], 4) != 0) {
                    ++bad;
                    if (memcmp(&two, &hd[4 * l + i], 4) == 0) ++unfused;
                    if (bad < 6) printf("trial %d lane %d i %d: mfma %.9g fma %.9g mul+add %.9g\n", trial, l, i, hd[4 * l + i], fused, two);
                }
            }
    }
    printf("layout D_i(lane 4b+j) = A[4b+i] * B[4b+j] + C_i: %d of %d values differ from fmaf (%d of them equal mul-then-add)\n", bad, 200 * 256, unfused);
    float *out;
    hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, wgs = 256 * 4;  // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    auto time = [&](auto kern, const char *name, double instr_per_iter) {
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: 4 waves x iters x instr_per_iter instructions in ms
        printf("%-28s %.3f ms -> %.2f cycles per wave instruction per SIMD at 2.4 GHz (4 waves per SIMD)\n", name, ms,
               ms * 1e-3 * 2.4e9 / (4.0 * iters * instr_per_iter));
    };
    time(k_rate<1>, "4 mfma_f32_4x4x1", 4);
    time(k_rate<2>, "16 v_fma_f32", 16);
    time(k_rate<3>, "4 mfma + 16 fma", 20);
    return 0;
}
