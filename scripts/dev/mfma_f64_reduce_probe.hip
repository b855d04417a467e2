#include <hip/hip_runtime.h>
__global__ void k(const float *x, double *out)
{
    const int l = threadIdx.x;
    double v = (double)x[l];
    double t1 = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
    double t2 = __builtin_amdgcn_mfma_f64_4x4x4f64(t1, 1.0, 0.0, 0, 0, 0);
    out[l] = t2;
    out[64 + l] = t1;
}
int main()
{
    float hx[64]; for (int i = 0; i < 64; i++) hx[i] = (float)(1 << (i % 16)) + 100000.0f * (i / 16);
    float *dx; double *dout; hipMalloc(&dx, 256); hipMalloc(&dout, 1024);
    hipMemcpy(dx, hx, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dout);
    double ho[128]; hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
    for (int r = 0; r < 4; r++) { printf("row %d:", r); for (int i = 0; i < 16; i++) printf(" %.0f", ho[16 * r + i]); printf("\n  t1:"); for (int i = 0; i < 16; i++) printf(" %.0f", ho[64 + 16 * r + i]); printf("\n"); }
    return 0;
}
