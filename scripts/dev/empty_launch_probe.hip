// Development probe: what a launch whose workgroups exit at once costs in a stream, by grid size -- the overflow-only launch
// behind k_bpm_row, k_backward_big and k_large_raster are such launches at the headline shape.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/empty_launch_probe scripts/dev/empty_launch_probe.hip && /tmp/empty_launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_exit(const int *flag, int *out)
{
    if (flag[blockIdx.y] == 0) return;  // (one scalar load per workgroup, like lines_ok[image])
    out[blockIdx.x * 256 + threadIdx.x] = 1;
}

__global__ __launch_bounds__(256) void k_work(float *p, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}

int main()
{
    int *flag, *out;
    float *buf;
    const int n = 1 << 24;
    hipMalloc(&flag, 4096 * sizeof(int));
    hipMemset(flag, 0, 4096 * sizeof(int));
    hipMalloc(&out, 1 << 20);
    hipMalloc(&buf, n * sizeof(float));
    hipMemset(buf, 0, n * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 200;
    float base = 0.0f;
    for (int pass = 0; pass < 2; ++pass)
        for (int wgs : {0, 1, 64, 256, 1024, 4096, 16384, 65536}) {
            // a working kernel (~10 us) followed by the empty one, as in the step
            for (int warm = 0; warm < 2; ++warm) {
                hipEventRecord(e0, 0);
                for (int i = 0; i < iters; ++i) {
                    hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, 0, buf, n);
                    if (wgs) hipLaunchKernelGGL(k_exit, dim3((wgs + 63) / 64, wgs < 64 ? wgs : 64), dim3(256), 0, 0, flag, out);
                }
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (!wgs) base = ms;
            if (pass) printf("workgroups %6d: %.2f us per pair, empty launch = %.2f us\n", wgs, ms * 1e3f / iters, (ms - base) * 1e3f / iters);
        }
    return 0;
}
