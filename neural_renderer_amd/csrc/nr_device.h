// nr_device.h -- device helpers shared by the gfx950 kernels of libnr_hip.so.
//
// Numerics contract (DESIGN.md "Numerics"): float32 IEEE arithmetic in the operation order of the reference
// source (neural_renderer/rasterize.py), double promotion where the reference's CUDA text has a double
// literal, NO multiply-add contraction (-ffp-contract=off), correctly rounded division.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/nr_hip.h"

#define NR_API extern "C" __attribute__((visibility("default")))

namespace nr {

constexpr int WAVE = 64;

// --------------------------------------------------------------------------------------------------
// shared device helpers

// back-face test: rasterize.py:252 / :306 / :540
__device__ __forceinline__ bool is_backside(float x0, float y0, float x1, float y1, float x2, float y2)
{
    return (y2 - y0) * (x1 - x0) < (y1 - y0) * (x2 - x0);
}

// NDC -> pixel units: 0.5 * (v * is + is - 1), rasterize.py:258 / :549 (the 0.5 scaling is exact in f32)
__device__ __forceinline__ float to_pixel(float v, float fs) { return 0.5f * (v * fs + fs - 1.0f); }

// inverse of [[p0x,p1x,p2x],[p0y,p1y,p2y],[1,1,1]]: rasterize.py:261-269
__device__ __forceinline__ void compute_face_inv(const float px[3], const float py[3], float inv[9])
{
    inv[0] = py[1] - py[2];
    inv[1] = px[2] - px[1];
    inv[2] = px[1] * py[2] - px[2] * py[1];
    inv[3] = py[2] - py[0];
    inv[4] = px[0] - px[2];
    inv[5] = px[2] * py[0] - px[0] * py[2];
    inv[6] = py[0] - py[1];
    inv[7] = px[1] - px[0];
    inv[8] = px[0] * py[1] - px[1] * py[0];
    const float den = px[2] * (py[0] - py[1]) + px[0] * (py[1] - py[2]) + px[1] * (py[2] - py[0]);
#pragma unroll
    for (int k = 0; k < 9; k++) inv[k] /= den;
}

// pixel centre in NDC: (2. * i + 1 - is) / is evaluated in double, rasterize.py:291-292
__device__ __forceinline__ float pixel_center(int i, int S) { return (float)((2.0 * i + 1 - S) / S); }

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ float bcast_f(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}


// --------------------------------------------------------------------------------------------------
// Screen-space bounding box of a face in pixel units (4 x int16: x_lo, x_hi, y_lo, y_hi), shared by the
// forward tile rasterizer and the face-centric backward kernels (they MUST agree: a pixel the forward
// could give to a face has to be inside the box the backward scans).
// An empty box (x_lo > x_hi) marks faces that can never own a pixel: back faces (rasterize.py:306),
// off-screen faces, faces whose three vertices coincide (their barycentric weights are NaN for every
// pixel, so `zp < depth_min` never holds, :322-334).  Degenerate faces (zero / non-finite determinant but
// distinct vertices) keep the full image: the reference's inside test can accept pixels anywhere on their
// supporting line.  Regular faces get the exact pixel range of the triangle plus a guard band of
// BBOX_GUARD pixels: the inside test runs on rounded NDC floats, whose rounding moves an edge by
// ~1e-6 pixel, so 0.25 pixel is conservative for anything but needles thinner than ~1e-5 pixel.
struct __attribute__((aligned(8))) BBox {
    short x_lo, x_hi, y_lo, y_hi;
};
constexpr float BBOX_GUARD = 0.25f;

__device__ __forceinline__ BBox face_bbox(float x0, float y0, float x1, float y1, float x2, float y2, int S)
{
    BBox bb;
    bb.x_lo = 1; bb.x_hi = 0; bb.y_lo = 1; bb.y_hi = 0;
    if (is_backside(x0, y0, x1, y1, x2, y2)) return bb;
    if ((x0 == x1) && (x1 == x2) && (y0 == y1) && (y1 == y2)) return bb;
    const float fs = (float)S;
    const float px[3] = {to_pixel(x0, fs), to_pixel(x1, fs), to_pixel(x2, fs)};
    const float py[3] = {to_pixel(y0, fs), to_pixel(y1, fs), to_pixel(y2, fs)};
    const float den = px[2] * (py[0] - py[1]) + px[0] * (py[1] - py[2]) + px[1] * (py[2] - py[0]);
    if (!(fabsf(den) > 0.0f) || !(fabsf(den) < __builtin_inff())) {
        bb.x_lo = 0; bb.x_hi = (short)(S - 1); bb.y_lo = 0; bb.y_hi = (short)(S - 1);
        return bb;
    }
    const float xmin = fminf(fminf(px[0], px[1]), px[2]), xmax = fmaxf(fmaxf(px[0], px[1]), px[2]);
    const float ymin = fminf(fminf(py[0], py[1]), py[2]), ymax = fmaxf(fmaxf(py[0], py[1]), py[2]);
    const float lo_c = -2.0f, hi_c = (float)S + 1.0f;  // clamp before the int conversion (huge / NaN coordinates)
    const int xl = max((int)ceilf(fminf(fmaxf(xmin - BBOX_GUARD, lo_c), hi_c)), 0);
    const int xh = min((int)floorf(fminf(fmaxf(xmax + BBOX_GUARD, lo_c), hi_c)), S - 1);
    const int yl = max((int)ceilf(fminf(fmaxf(ymin - BBOX_GUARD, lo_c), hi_c)), 0);
    const int yh = min((int)floorf(fminf(fmaxf(ymax + BBOX_GUARD, lo_c), hi_c)), S - 1);
    if (xl <= xh && yl <= yh) {
        bb.x_lo = (short)xl; bb.x_hi = (short)xh; bb.y_lo = (short)yl; bb.y_hi = (short)yh;
    }
    return bb;
}

// --------------------------------------------------------------------------------------------------
// texture taps shared by F3 (forward) and B2 (backward recompute): rasterize.py:398-425
struct Taps {
    int isc[8];
    float w[8];
};

// The eight trilinear taps of a pixel (rasterize.py:398-421).  When an index float reaches ts - 1 exactly (eps too small to
// survive the float32 rounding of :402, or eps = 0) the "upper" corner of that dimension has index ts and weight exactly 0;
// its flattened index can then leave the face's cube (isc >= ts^3).  The reference multiplies whatever lies there by 0 /
// adds 0 to it; consumers here skip such taps instead of touching memory outside the cube.
__device__ __forceinline__ void compute_taps(const float *__restrict__ face, const float *__restrict__ weight,
                                             float depth, int ts, double eps, Taps &t)
{
    float tif[3];
    int ti[3];
    float fr[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float v = weight[k] * (float)(ts - 1) * (depth / face[3 * k + 2]);  // :400
        v = fmaxf(v, 0.0f);                                                 // :401
        v = (float)fmin((double)v, (double)(ts - 1) - eps);                 // :402 (double min, then rounded)
        tif[k] = v;
        ti[k] = (int)v;
        fr[k] = v - (float)ti[k];
    }
#pragma unroll
    for (int pn = 0; pn < 8; pn++) {
        float w = 1.0f;
        int idx[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (((pn >> k) & 1) == 0) {
                w *= 1.0f - fr[k];
                idx[k] = ti[k];
            } else {
                w *= fr[k];
                idx[k] = ti[k] + 1;
            }
        }
        t.isc[pn] = idx[0] * ts * ts + idx[1] * ts + idx[2];
        t.w[pn] = w;
    }
}


// --------------------------------------------------------------------------------------------------
// host side helpers
// XCD-aware workgroup placement.  MI355X has 8 accelerator dies (XCDs), each with a private L2; the hardware deals
// consecutive workgroup ids of a launch round-robin to them.  Work items that share data (the faces, maps and textures of one
// image) should therefore NOT have consecutive ids: a 1-D grid of xcd_grid(n) workgroups is launched and the kernel uses
// xcd_block(n) instead of blockIdx.x, which makes each XCD walk one contiguous 1/8 of the logical range.
constexpr unsigned NUM_XCD = 8;
inline unsigned xcd_grid(size_t n_blocks) { return (unsigned)((n_blocks + NUM_XCD - 1) / NUM_XCD * NUM_XCD); }
#ifdef __HIPCC__
// logical block id in [0, n_blocks), or n_blocks (= "no work") for the padding blocks
__device__ __forceinline__ unsigned xcd_block(unsigned n_blocks)
{
    const unsigned chunk = (n_blocks + NUM_XCD - 1) / NUM_XCD;
    const unsigned logical = (blockIdx.x % NUM_XCD) * chunk + blockIdx.x / NUM_XCD;
    return logical < n_blocks ? logical : n_blocks;
}
#endif

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline int check_sizes(int B, int F, int S)
{
    if (B < 1 || F < 1 || S < 1 || S > 16384) return NR_E_SIZE;
    if ((size_t)B * (size_t)F > 0x7fffffffull / 9) return NR_E_SIZE;  // int32 face indexing inside kernels
    return 0;
}

inline int launch_status()
{
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}


// --------------------------------------------------------------------------------------------------
// stage runners shared between the per-stage ABI entry points and the fused nr_backward_rasterize.
// vis_list / vis_count (optional): per-image sorted lists of the faces that own at least one pixel, as built
// by the K6 band pipeline ([B][F] ints, [B] counts); when given, the gather kernels visit only those faces.
int run_backward_pixel_map(const float *faces, const int32_t *face_index_map, const float *rgb_map,
                           const float *alpha_map, const float *grad_rgb_map, const float *grad_alpha_map,
                           float *grad_faces, int B, int F, int S, double eps, int return_rgb, int return_alpha,
                           void *workspace, size_t workspace_bytes, hipStream_t st, const int **vis_list_out,
                           const int **vis_count_out);
int run_backward_textures(const int32_t *face_index_map, const float *sampling_weight_map,
                          const int32_t *sampling_index_map, const float *faces, const float *weight_map,
                          const float *depth_map, const float *grad_rgb_map, float *grad_textures, int B, int F, int S,
                          int ts, double eps, int flags, const int *vis_list, const int *vis_count, hipStream_t st,
                          const float *g_depth_fused, float *grad_faces_fused, int *depth_done);
int run_backward_depth_map(const float *faces, const float *depth_map, const int32_t *face_index_map,
                           const float *face_inv_map, const float *weight_map, const float *grad_depth_map,
                           float *grad_faces, int B, int F, int S, const int *vis_list, const int *vis_count,
                           hipStream_t st);

}  // namespace nr
