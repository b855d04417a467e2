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
// Candidate pixels of a face: the only pixels the forward needs to test and the backward gathers need to scan (they MUST
// agree: a pixel the forward could give to a face has to be among the candidates the backward visits).
//   * none: back faces (rasterize.py:306), off-screen faces, faces whose three vertices coincide (their barycentric weights
//     are NaN for every pixel, so `zp < depth_min` never holds, :322-334);
//   * a BOX: the exact pixel range of the triangle plus a guard band of BBOX_GUARD pixels -- the inside test (:310-312) runs
//     on rounded NDC floats, whose rounding moves an edge by ~1e-6 pixel, so 0.25 pixel is conservative;
//   * a STRIP along the longest edge for needles and degenerate (collinear) faces.  The inside test compares rounded
//     products of NDC differences; its rounding error is a relative 2^-22 or so of |p - a| * |b - a|, i.e. every edge line is
//     accepted within an ANGLE of ~5e-7 rad at any distance.  Beyond a vertex whose interior angle is below twice that, the
//     wedges of the two adjacent edges overlap and the reference accepts pixels far outside the triangle, along its axis
//     (found by fuzzing: micro-triangles of a few ulps through pixel centres).  Faces that thin -- 2 * area <= 2^-18 *
//     (longest edge)^2 in NDC, the coordinates the test works with -- lie within 4e-6 of their longest edge's line, and so does
//     everything the test can accept: the candidates are the STRIP_W pixels around that line in every row (or column);
//   * the whole image when nothing better can be said (non-finite or astronomically large coordinates).
struct Cand {
    int x_lo, y_lo, bw;  // box: origin and width
    int n;               // number of candidates, 0 = none
    int strip;           // 0 box, 1 strip stepping through rows (x = a * y + b), 2 strip stepping through columns
    float a, b;
};
constexpr float BBOX_GUARD = 0.25f;
constexpr int STRIP_W = 4;  // floor(line) - 1 .. floor(line) + 2

__device__ __forceinline__ Cand face_candidates(float x0, float y0, float x1, float y1, float x2, float y2, int S)
{
    Cand c;
    c.x_lo = c.y_lo = 0; c.bw = 1; c.n = 0; c.strip = 0; c.a = c.b = 0.0f;
    if (is_backside(x0, y0, x1, y1, x2, y2)) return c;
    if ((x0 == x1) && (x1 == x2) && (y0 == y1) && (y1 == y2)) return c;
    const float fs = (float)S;
    const float px[3] = {to_pixel(x0, fs), to_pixel(x1, fs), to_pixel(x2, fs)};
    const float py[3] = {to_pixel(y0, fs), to_pixel(y1, fs), to_pixel(y2, fs)};
    const float den = px[2] * (py[0] - py[1]) + px[0] * (py[1] - py[2]) + px[1] * (py[2] - py[0]);
    const float e0x = x1 - x0, e0y = y1 - y0, e1x = x2 - x1, e1y = y2 - y1, e2x = x0 - x2, e2y = y0 - y2;
    const float area2 = fabsf(e0x * e1y - e0y * e1x);
    const float l0 = e0x * e0x + e0y * e0y, l1 = e1x * e1x + e1y * e1y, l2 = e2x * e2x + e2y * e2y;
    const float len2 = fmaxf(fmaxf(l0, l1), l2);
    const bool thin = !(area2 > 0x1p-18f * len2) || !(fabsf(den) > 0.0f);
    const float big = 1.0e6f;  // pixel coordinates beyond this make the strip arithmetic meaningless
    const bool wild = !(len2 < __builtin_inff()) || !(fabsf(den) < __builtin_inff()) ||
                      !(fmaxf(fmaxf(fabsf(px[0]), fabsf(px[1])), fabsf(px[2])) < big) ||
                      !(fmaxf(fmaxf(fabsf(py[0]), fabsf(py[1])), fabsf(py[2])) < big);
    // A face is "thin" relative to its longest edge; with vertices far off-screen (a perspective division by z near 0) that
    // edge can be 10^5 pixels long and the face still several pixels high on screen.  The 4-pixel strip is only conservative
    // while 2^-18 * (longest edge in pixels) stays below half a pixel; beyond that the whole image is the candidate set.
    const bool strip_ok = len2 * fs * fs * 0.25f * 0x1p-36f <= 0.25f;
    if (thin && !wild && strip_ok) {
        // the longest edge: its direction from the NDC differences (exact for close vertices -- in pixel units a
        // micro-triangle's edge would be a few ulps of noise), its position from one endpoint in pixel units
        const int k = (l0 >= l1 && l0 >= l2) ? 0 : (l1 >= l2 ? 1 : 2);
        const float ax = px[k], ay = py[k];
        const float dx = k == 0 ? e0x : (k == 1 ? e1x : e2x), dy = k == 0 ? e0y : (k == 1 ? e1y : e2y);
        if (fabsf(dy) >= fabsf(dx) && dy != 0.0f) {
            c.strip = 1; c.a = dx / dy; c.b = ax - c.a * ay;
        } else if (dx != 0.0f) {
            c.strip = 2; c.a = dy / dx; c.b = ay - c.a * ax;
        }
        if (c.strip) { c.n = S * STRIP_W; return c; }
    }
    if (thin || wild) {  // the whole image
        c.bw = S; c.n = S * S;
        return c;
    }
    const float xmin = fminf(fminf(px[0], px[1]), px[2]), xmax = fmaxf(fmaxf(px[0], px[1]), px[2]);
    const float ymin = fminf(fminf(py[0], py[1]), py[2]), ymax = fmaxf(fmaxf(py[0], py[1]), py[2]);
    const float lo_c = -2.0f, hi_c = (float)S + 1.0f;  // clamp before the int conversion
    const int xl = max((int)ceilf(fminf(fmaxf(xmin - BBOX_GUARD, lo_c), hi_c)), 0);
    const int xh = min((int)floorf(fminf(fmaxf(xmax + BBOX_GUARD, lo_c), hi_c)), S - 1);
    const int yl = max((int)ceilf(fminf(fmaxf(ymin - BBOX_GUARD, lo_c), hi_c)), 0);
    const int yh = min((int)floorf(fminf(fmaxf(ymax + BBOX_GUARD, lo_c), hi_c)), S - 1);
    if (xl <= xh && yl <= yh) {
        c.x_lo = xl; c.y_lo = yl; c.bw = xh - xl + 1; c.n = c.bw * (yh - yl + 1);
    }
    return c;
}

// i-th candidate -> pixel (x, y); false when it falls outside the image (strips only)
__device__ __forceinline__ bool cand_pixel(const Cand &c, int i, int S, int &x, int &y)
{
    if (c.strip == 0) {
        const int yy = i / c.bw;
        x = c.x_lo + (i - yy * c.bw);
        y = c.y_lo + yy;
        return true;
    }
    const int m = i / STRIP_W, j = i - m * STRIP_W;
    const int o = (int)floorf(c.a * (float)m + c.b) - 1 + j;
    if (c.strip == 1) { y = m; x = o; } else { x = m; y = o; }
    return o >= 0 && o < S;
}

// --------------------------------------------------------------------------------------------------
// Per-face light colours (nr_face_light in include/nr_hip.h): instead of textures that were multiplied by the
// light colour of their face and duplicated for the fill_back copy in front of the rasterizer (lighting.py:50-51,
// renderer.py:79: 2 x B x Nf x ts^3 x 3 floats written, read, and the same again for the gradient), the kernels read the
// ORIGINAL cubes [B, tex_faces, ts^3, 3] and multiply the sampled colour by light[b, f, :].  Face f >= tex_faces is the
// reversed copy of face f - tex_faces and reads its cube with the first and the third axis exchanged.
struct FaceLight {
    const float *light = nullptr;     // [B, F, 3]; NULL = off: textures are [B, F, ts^3, 3], sampled as they are
    int tex_faces = 0;                // Nf (F == Nf or F == 2 * Nf)
    const float *textures = nullptr;  // backward only: the cubes, for the gradient of `light`
    float *grad_light = nullptr;      // backward only: [B, F, 3] or NULL
};

// texel (i, j, k) -> (k, j, i) of a ts^3 cube, flattened (renderer.py:79)
__device__ __forceinline__ int transpose_texel(int t, int ts)
{
    const int k = t % ts, j = (t / ts) % ts, i = t / (ts * ts);
    return (k * ts + j) * ts + i;
}

// texture taps shared by F3 (forward) and B2 (backward recompute): rasterize.py:398-425
struct Taps {
    int isc[8];
    float w[8];
};

// The eight trilinear taps of a pixel (rasterize.py:398-421).  When an index float reaches ts - 1 exactly (eps too small to
// survive the float32 rounding of :402, or eps = 0) the "upper" corner of that dimension has index ts and weight exactly 0;
// its flattened index can then leave the face's cube (isc >= ts^3).  The reference multiplies whatever lies there by 0 /
// adds 0 to it; consumers here skip such taps instead of touching memory outside the cube.
// z: the face's three vertex depths; flip: flatten the taps of the cube with axes 0 and 2 exchanged (FaceLight: the
// reversed copy of a face reads the original cube transposed)
__device__ __forceinline__ void compute_taps(const float *__restrict__ z, const float *__restrict__ weight,
                                             float depth, int ts, double eps, Taps &t, bool flip = false)
{
    float tif[3];
    int ti[3];
    float fr[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float v = weight[k] * (float)(ts - 1) * (depth / z[k]);  // :400
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
        t.isc[pn] = (flip ? idx[2] : idx[0]) * ts * ts + idx[1] * ts + (flip ? idx[0] : idx[2]);
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

#ifdef __HIPCC__
// Fills of the library's own instead of hipMemsetAsync: a captured HIP graph that holds a memset node on memory from outside
// the graph's pool misbehaves on replay on ROCm 7.2 (the 0xff fill of the z-buffer faults on the second replay, the zero
// fill of grad_textures leaves garbage) -- found with the operator's graph-replay mode.  Kernel nodes replay fine.
static __global__ __launch_bounds__(256) void k_fill_bytes(unsigned char *__restrict__ dst, size_t bytes, unsigned value32)
{
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i0 >= bytes) return;
    if (i0 + 16 <= bytes && ((size_t)(dst + i0) & 15) == 0) {
        *reinterpret_cast<uint4 *>(dst + i0) = make_uint4(value32, value32, value32, value32);
    } else {
        for (size_t i = i0; i < bytes && i < i0 + 16; ++i) dst[i] = (unsigned char)value32;
    }
}
// every byte of [dst, dst + bytes) = byte; returns a hipError_t as int (0 = ok)
inline int fill_bytes(void *dst, int byte, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return 0;
    const unsigned b = (unsigned)(byte & 0xff), v = b | (b << 8) | (b << 16) | (b << 24);
    hipLaunchKernelGGL(k_fill_bytes, dim3((unsigned)((bytes + 4095) / 4096)), dim3(256), 0, st, (unsigned char *)dst, bytes, v);
    return (int)hipGetLastError();
}
#endif

inline int check_sizes(int B, int F, int S)
{
    if (B < 1 || B > 65535 || F < 1 || S < 1 || S > 16384) return NR_E_SIZE;  // B: several kernels put the image on grid.y
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
struct SetupHook;      // nr_band_lines.h
struct LineSetupArgs;  // nr_band_lines.h
int run_backward_pixel_map(const float *faces, const int32_t *face_index_map, const float *rgb_map,
                           const float *alpha_map, const float *grad_rgb_map, const float *grad_alpha_map,
                           float *grad_faces, int B, int F, int S, double eps, int return_rgb, int return_alpha,
                           int flags, const unsigned char *visible_faces, void *workspace, size_t workspace_bytes,
                           hipStream_t st, const int **vis_list_out, const int **vis_count_out,
                           const double **defer_scratch = nullptr, const int **defer_slot_of = nullptr,
                           void *zero_ptr = nullptr, size_t zero_bytes = 0, int *zeroed = nullptr,
                           const SetupHook *hook = nullptr);
// zero_ptr / zero_bytes: a buffer the caller wants zero-filled before its next kernel (the fused backward's grad_textures);
// *zeroed = 1 when the band kernel did it on the side (default kernel, 16-byte aligned, <= 256 MB), else the caller fills
// defer_scratch / defer_slot_of (both or none): the caller will finish K6 itself for the LISTED faces -- rounding the double
// sums of their list positions into grad_faces (run_backward_textures does, or run_bpm_finalize for all faces) -- so
// k_bpm_finalize is not launched and the compaction kernel stores the zeros of the unlisted faces; NULLs come back when
// the band pipeline did not run (global-memory fallback: grad_faces are complete).
void run_bpm_finalize(const double *scratch, const int *slot_of, float *grad_faces, int B, int F, hipStream_t st,
                      bool add = false);  // add: on top of what grad_faces holds, listed faces only (see the kernel)
int run_backward_textures(const int32_t *face_index_map, const float *sampling_weight_map,
                          const int32_t *sampling_index_map, const float *faces, const float *faces_z_ref,
                          const float *weight_map,
                          const float *depth_map, const float *grad_rgb_map, float *grad_textures, int B, int F, int S,
                          int ts, double eps, int flags, const int *vis_list, const int *vis_count, hipStream_t st,
                          const float *g_depth_fused, float *grad_faces_fused, int *depth_done,
                          const double *k6_scratch, const int *slot_of, int *k6_finalized, const FaceLight &lit = FaceLight(),
                          bool prefilled = false, int phase = 0, const struct LineSetupArgs *ls = nullptr,
                          const int *zero_slot_of = nullptr);
// prefilled: grad_textures is already zero (no fill launch).  phase: 0 everything; 1 the fills and the gathers; 2 what they leave
// out (k_backward_big) -- the fused backward runs K6's band kernel between the two.  ls: K6's line-setup launch, to go into
// the gather's launch (phase 0 / 1; launched here in any case); zero_slot_of: K6's face -> list position table, with which
// that launch also stores grad_textures' zeros (the cubes of unlisted faces) instead of a fill in front
// lit.light given: grad_textures is [B, lit.tex_faces, ts^3, 3] (zero-filled here; a face stores only when it owns a
// pixel -- of a face and its reversed copy at most one does), lit.grad_light receives [B, F, 3]
int face_light_args(const nr_face_light *lit, int F, bool backward, FaceLight &out);  // nr_forward.hip
int run_backward_depth_map(const float *faces, const float *depth_map, const int32_t *face_index_map,
                           const float *face_inv_map, const float *weight_map, const float *grad_depth_map,
                           float *grad_faces, int B, int F, int S, const int *vis_list, const int *vis_count,
                           hipStream_t st, const unsigned char *visible);  // visible: the forward's per-face flags or NULL

}  // namespace nr
