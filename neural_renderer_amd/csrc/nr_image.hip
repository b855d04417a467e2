// nr_image.hip -- the callee immediately behind the rasterizer: the image epilogue of
// `rasterize_rgbad` (reference neural_renderer/rasterize.py:953-969): NHWC -> NCHW transposition of rgb, vertical flip
// of every output, and the 0.5x down-sampling of anti-aliasing (`average_pooling_2d(x, 2, 2)`), plus its backward.
//
// In the reference (and in stock torch) that is transpose + flip + pooling per output and the three matching backward
// ops: up to 9 full-image passes forward and 9 backward at the super-sampled size.  Here each direction is one
// bandwidth-bound kernel that reads every map once and writes every image once.
//
// Arithmetic: the 2x2 mean is ((a + b) + c) + d, times 0.25f, with a, b the upper row of the block in the flipped
// image (left, right) and c, d the lower row -- the order of oracle._avg_pool2.  The backward multiplies by 0.25f.
#include "nr_device.h"

using namespace nr;

namespace {

// one thread per OUTPUT pixel (b, r, c) of the [is, is] images; r = 0 is the TOP row (raster row S-1)
template <bool AA>
__global__ __launch_bounds__(256) void k_image_epilogue(const float *__restrict__ rgb_map,
                                                        const float *__restrict__ alpha_map,
                                                        const float *__restrict__ depth_map, float *__restrict__ rgb_out,
                                                        float *__restrict__ alpha_out, float *__restrict__ depth_out,
                                                        int S, int is, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % is);
    const int r = (int)((i / is) % is);
    const size_t b = i / ((size_t)is * is);
    const size_t plane = (size_t)is * is;
    if (AA) {
        // flipped rows 2r, 2r+1 are raster rows S-1-2r, S-2-2r
        const size_t top = (b * S + (size_t)(S - 1 - 2 * r)) * S + 2 * c;
        const size_t bot = top - S;
        if (rgb_map) {
            const float *t = rgb_map + top * 3, *u = rgb_map + bot * 3;
#pragma unroll
            for (int k = 0; k < 3; k++)
                rgb_out[(b * 3 + k) * plane + (size_t)r * is + c] = (((t[k] + t[3 + k]) + u[k]) + u[3 + k]) * 0.25f;
        }
        if (alpha_map) alpha_out[i] = (((alpha_map[top] + alpha_map[top + 1]) + alpha_map[bot]) + alpha_map[bot + 1]) * 0.25f;
        if (depth_map) depth_out[i] = (((depth_map[top] + depth_map[top + 1]) + depth_map[bot]) + depth_map[bot + 1]) * 0.25f;
    } else {
        const size_t src = (b * S + (size_t)(S - 1 - r)) * S + c;
        if (rgb_map) {
#pragma unroll
            for (int k = 0; k < 3; k++) rgb_out[(b * 3 + k) * plane + (size_t)r * is + c] = rgb_map[src * 3 + k];
        }
        if (alpha_map) alpha_out[i] = alpha_map[src];
        if (depth_map) depth_out[i] = depth_map[src];
    }
}

// one thread per RASTER pixel (b, y, x) of the [S, S] maps; y = 0 is the BOTTOM row
template <bool AA>
__global__ __launch_bounds__(256) void k_image_epilogue_backward(const float *__restrict__ g_rgb_out,
                                                                 const float *__restrict__ g_alpha_out,
                                                                 const float *__restrict__ g_depth_out,
                                                                 float *__restrict__ g_rgb_map,
                                                                 float *__restrict__ g_alpha_map,
                                                                 float *__restrict__ g_depth_map, int S, int is, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % S);
    const int y = (int)((i / S) % S);
    const size_t b = i / ((size_t)S * S);
    const size_t plane = (size_t)is * is;
    const int r = AA ? (S - 1 - y) >> 1 : S - 1 - y;
    const int c = AA ? x >> 1 : x;
    const size_t src = (size_t)r * is + c;
    if (g_rgb_map) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float g = g_rgb_out[(b * 3 + k) * plane + src];
            g_rgb_map[i * 3 + k] = AA ? g * 0.25f : g;
        }
    }
    if (g_alpha_map) {
        const float g = g_alpha_out[b * plane + src];
        g_alpha_map[i] = AA ? g * 0.25f : g;
    }
    if (g_depth_map) {
        const float g = g_depth_out[b * plane + src];
        g_depth_map[i] = AA ? g * 0.25f : g;
    }
}

inline int epilogue_args(const void *a0, const void *a1, const void *b0, const void *b1, const void *c0, const void *c1,
                         int B, int S, int aa)
{
    if ((a0 == nullptr) != (a1 == nullptr) || (b0 == nullptr) != (b1 == nullptr) || (c0 == nullptr) != (c1 == nullptr))
        return NR_E_MODE;
    if (!a0 && !b0 && !c0) return NR_E_MODE;
    if (B < 1 || S < 1 || S > 16384) return NR_E_SIZE;
    if (aa && (S & 1)) return NR_E_SIZE;
    return 0;
}

}  // namespace

NR_API int nr_image_epilogue(const float *rgb_map, const float *alpha_map, const float *depth_map, float *rgb_out,
                             float *alpha_out, float *depth_out, int32_t B, int32_t S, int32_t anti_aliasing, void *stream)
{
    const int rc = epilogue_args(rgb_map, rgb_out, alpha_map, alpha_out, depth_map, depth_out, B, S, anti_aliasing);
    if (rc) return rc;
    const int is = anti_aliasing ? S / 2 : S;
    const size_t n = (size_t)B * is * is;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (anti_aliasing)
        hipLaunchKernelGGL(k_image_epilogue<true>, grid, block, 0, (hipStream_t)stream, rgb_map, alpha_map, depth_map,
                           rgb_out, alpha_out, depth_out, S, is, n);
    else
        hipLaunchKernelGGL(k_image_epilogue<false>, grid, block, 0, (hipStream_t)stream, rgb_map, alpha_map, depth_map,
                           rgb_out, alpha_out, depth_out, S, is, n);
    return launch_status();
}

NR_API int nr_image_epilogue_backward(const float *grad_rgb_out, const float *grad_alpha_out, const float *grad_depth_out,
                                      float *grad_rgb_map, float *grad_alpha_map, float *grad_depth_map, int32_t B,
                                      int32_t S, int32_t anti_aliasing, void *stream)
{
    const int rc = epilogue_args(grad_rgb_out, grad_rgb_map, grad_alpha_out, grad_alpha_map, grad_depth_out,
                                 grad_depth_map, B, S, anti_aliasing);
    if (rc) return rc;
    const int is = anti_aliasing ? S / 2 : S;
    const size_t n = (size_t)B * S * S;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (anti_aliasing)
        hipLaunchKernelGGL(k_image_epilogue_backward<true>, grid, block, 0, (hipStream_t)stream, grad_rgb_out,
                           grad_alpha_out, grad_depth_out, grad_rgb_map, grad_alpha_map, grad_depth_map, S, is, n);
    else
        hipLaunchKernelGGL(k_image_epilogue_backward<false>, grid, block, 0, (hipStream_t)stream, grad_rgb_out,
                           grad_alpha_out, grad_depth_out, grad_rgb_map, grad_alpha_map, grad_depth_map, S, is, n);
    return launch_status();
}
