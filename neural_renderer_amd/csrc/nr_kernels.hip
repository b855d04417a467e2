// nr_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels + C ABI (include/nr_hip.h) for the
// differentiable rasterizer hot path of neural_renderer (reference: neural_renderer/rasterize.py).
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
//              -munsafe-fp-atomics -fPIC -shared  (see neural_renderer_amd/_build.py)
//
// Numerics contract (DESIGN.md "Numerics"): float32 IEEE arithmetic in the operation order of the
// reference source, double promotion where the reference's CUDA text has a double literal, NO
// multiply-add contraction, correctly rounded division.  face_index_map is bit-exact w.r.t. the oracle.
//
// Kernels
//   k_face_setup           F1  per face: back-face cull, inverse barycentric matrix, screen bbox   (ref K1)
//   k_raster_tiles         F2  per 32x32 tile: bbox-scan of the image's faces into an LDS list, then one
//                              pixel per lane (16x4 blocks per wave) resolves min-depth over the list      (ref K2)
//   k_shade                F3  per pixel: trilinear texture sampling + background + alpha             (ref K4+K5)
//   k_backward_pixel_map   B1  per (face, edge, axis) lane; short sweeps serial, long sweeps by the whole
//                              wave (lane = pixel), DPP/bpermute reduction, plain stores             (ref K6)
//   k_backward_textures    B2  per pixel: recompute or read the 8 sampling taps, hardware f32 atomics   (ref K7)
//   k_backward_depth_map   B3  per pixel: analytic depth gradient, hardware f32 atomics                 (ref K8)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nr_hip.h"

#define NR_API extern "C" __attribute__((visibility("default")))

namespace {

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
// F1: per-face setup.  workspace = inv[B*F*9] floats, then bbox[B*F] (4 x int16: x_lo, x_hi, y_lo, y_hi).
// An empty box (x_lo > x_hi) marks faces that can never own a pixel (back faces, off-screen, all three
// vertices coincident).  Degenerate faces (zero / non-finite determinant) get the full-image box so that
// culling never changes what the reference's brute-force loop would have produced.
struct __attribute__((aligned(8))) BBox {
    short x_lo, x_hi, y_lo, y_hi;
};

__global__ __launch_bounds__(256) void k_face_setup(const float *__restrict__ faces, float *__restrict__ ws_inv,
                                                    BBox *__restrict__ ws_bbox, int n_faces_total, int S)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_faces_total) return;
    const float *f = faces + (size_t)i * 9;
    const float x0 = f[0], y0 = f[1], x1 = f[3], y1 = f[4], x2 = f[6], y2 = f[7];
    float inv[9];
    BBox bb;
    bb.x_lo = 1; bb.x_hi = 0; bb.y_lo = 1; bb.y_hi = 0;
    if (is_backside(x0, y0, x1, y1, x2, y2)) {
#pragma unroll
        for (int k = 0; k < 9; k++) inv[k] = 0.0f;  // rasterize.py:240 zeros_like + :253 continue
    } else {
        const float fs = (float)S;
        const float px[3] = {to_pixel(x0, fs), to_pixel(x1, fs), to_pixel(x2, fs)};
        const float py[3] = {to_pixel(y0, fs), to_pixel(y1, fs), to_pixel(y2, fs)};
        compute_face_inv(px, py, inv);
        const float den = px[2] * (py[0] - py[1]) + px[0] * (py[1] - py[2]) + px[1] * (py[2] - py[0]);
        const bool coincident = (x0 == x1) && (x1 == x2) && (y0 == y1) && (y1 == y2);
        if (coincident) {
            // w = NaN for every pixel -> zp = NaN -> never selected (rasterize.py:322-334): cull.
        } else if (!(fabsf(den) > 0.0f) || !(fabsf(den) < __builtin_inff())) {
            bb.x_lo = 0; bb.x_hi = (short)(S - 1); bb.y_lo = 0; bb.y_hi = (short)(S - 1);
        } else {
            // conservative box with a one-pixel guard band (the inside test runs in NDC floats)
            const float xmin = fminf(fminf(px[0], px[1]), px[2]), xmax = fmaxf(fmaxf(px[0], px[1]), px[2]);
            const float ymin = fminf(fminf(py[0], py[1]), py[2]), ymax = fmaxf(fmaxf(py[0], py[1]), py[2]);
            const float hi = (float)S + 1.0f;
            const int xl = max((int)floorf(fminf(fmaxf(xmin, -2.0f), hi)) - 1, 0);
            const int xh = min((int)ceilf(fminf(fmaxf(xmax, -2.0f), hi)) + 1, S - 1);
            const int yl = max((int)floorf(fminf(fmaxf(ymin, -2.0f), hi)) - 1, 0);
            const int yh = min((int)ceilf(fminf(fmaxf(ymax, -2.0f), hi)) + 1, S - 1);
            if (xl <= xh && yl <= yh) {
                bb.x_lo = (short)xl; bb.x_hi = (short)xh; bb.y_lo = (short)yl; bb.y_hi = (short)yh;
            }
        }
    }
    float *o = ws_inv + (size_t)i * 9;
#pragma unroll
    for (int k = 0; k < 9; k++) o[k] = inv[k];
    ws_bbox[i] = bb;
}

// --------------------------------------------------------------------------------------------------
// F2: tile rasterizer.  One workgroup (4 waves) per 32x32-pixel tile of one image.  Faces are consumed in
// rounds of ROUND faces: every thread tests ROUND/256 boxes against the tile and appends survivors to an
// LDS list (wave-aggregated LDS atomic); then each wave walks its four 16x4 pixel blocks, 64 list entries
// per step (one box test per lane, __ballot), and for every surviving face all 64 lanes run the
// reference's inside / barycentric / depth test for their own pixel.  Face data of a candidate is
// wave-uniform, so it is fetched through the scalar path.  Winner rule: smaller zp, ties -> lower face
// index (the reference scans faces in ascending order with a strict `<`, rasterize.py:300,334).
constexpr int TILE = 32;
constexpr int BLK_W = 16, BLK_H = 4;
constexpr int ROUND = 1024;
constexpr int RASTER_THREADS = 256;

struct PixelState {
    float z, w0, w1, w2;
    int fn;
};

__device__ __forceinline__ void test_face(const float *__restrict__ f, const float *__restrict__ iv, int fn,
                                          float xp, float yp, float xif, float yif, double near_d, double far_d,
                                          PixelState &st)
{
    const float x0 = f[0], y0 = f[1], z0 = f[2], x1 = f[3], y1 = f[4], z1 = f[5], x2 = f[6], y2 = f[7], z2 = f[8];
    // rasterize.py:310-312 (back faces never reach here: their box is empty)
    if (((yp - y0) * (x1 - x0) < (xp - x0) * (y1 - y0)) || ((yp - y1) * (x2 - x1) < (xp - x1) * (y2 - y1)) ||
        ((yp - y2) * (x0 - x2) < (xp - x2) * (y0 - y2)))
        return;
    // :317-327
    float w0 = iv[0] * xif + iv[1] * yif + iv[2];
    float w1 = iv[3] * xif + iv[4] * yif + iv[5];
    float w2 = iv[6] * xif + iv[7] * yif + iv[8];
    w0 = fminf(fmaxf(w0, 0.0f), 1.0f);
    w1 = fminf(fmaxf(w1, 0.0f), 1.0f);
    w2 = fminf(fmaxf(w2, 0.0f), 1.0f);
    const float w_sum = (0.0f + w0) + w1 + w2;
    w0 /= w_sum;
    w1 /= w_sum;
    w2 /= w_sum;
    // :330 -- double reciprocal of a float rounded to float == correctly rounded float division
    const float zp = 1.0f / (w0 / z0 + w1 / z1 + w2 / z2);
    if ((double)zp <= near_d || far_d <= (double)zp) return;  // :331
    if (zp < st.z || (zp == st.z && fn < st.fn)) {            // :334 (+ explicit tie rule, see header)
        st.z = zp;
        st.fn = fn;
        st.w0 = w0;
        st.w1 = w1;
        st.w2 = w2;
    }
}

__global__ __launch_bounds__(RASTER_THREADS) void k_raster_tiles(
    const float *__restrict__ faces, const float *__restrict__ ws_inv, const BBox *__restrict__ ws_bbox,
    int32_t *__restrict__ face_index_map, float *__restrict__ weight_map, float *__restrict__ depth_map,
    float *__restrict__ face_inv_map, int F, int S, int tiles_x, double near_d, double far_d)
{
    __shared__ int s_fn[ROUND];
    __shared__ BBox s_bb[ROUND];
    __shared__ int s_cnt;

    const int b = blockIdx.y;
    const int tile_x0 = (blockIdx.x % tiles_x) * TILE;
    const int tile_y0 = (blockIdx.x / tiles_x) * TILE;
    const int tile_x1 = min(tile_x0 + TILE, S) - 1, tile_y1 = min(tile_y0 + TILE, S) - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t face_base = (size_t)b * F;

    // this lane's pixels: block r of wave w sits at column block (r & 1), row block (2 * w + (r >> 1))
    const int lx = lane & (BLK_W - 1), ly = lane >> 4;
    const int pxa[2] = {tile_x0 + lx, tile_x0 + BLK_W + lx};
    const int pya[2] = {tile_y0 + (2 * wave) * BLK_H + ly, tile_y0 + (2 * wave + 1) * BLK_H + ly};
    const float xpa[2] = {pixel_center(pxa[0], S), pixel_center(pxa[1], S)};
    const float ypa[2] = {pixel_center(pya[0], S), pixel_center(pya[1], S)};

    const float far_f = (float)far_d;  // rasterize.py:296
    PixelState st[4];
#pragma unroll
    for (int r = 0; r < 4; r++) { st[r].z = far_f; st[r].fn = -1; st[r].w0 = st[r].w1 = st[r].w2 = 0.0f; }

    for (int base = 0; base < F; base += ROUND) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        // ---- scan: bbox vs tile
#pragma unroll
        for (int j = 0; j < ROUND / RASTER_THREADS; j++) {
            const int fn = base + j * RASTER_THREADS + tid;
            bool hit = false;
            BBox bb;
            if (fn < F) {
                bb = ws_bbox[face_base + fn];
                hit = (bb.x_lo <= tile_x1) && (bb.x_hi >= tile_x0) && (bb.y_lo <= tile_y1) && (bb.y_hi >= tile_y0);
            }
            const unsigned long long m = __ballot(hit);
            if (m) {
                int off = 0;
                if (lane == 0) off = atomicAdd(&s_cnt, __popcll(m));
                off = rfl(off);
                if (hit) {
                    const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
                    s_fn[pos] = fn;
                    s_bb[pos] = bb;
                }
            }
        }
        __syncthreads();
        const int n = s_cnt;
        // ---- raster: lane = pixel
        if (n > 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int cx = r & 1, cy = r >> 1;
                const int bx0 = tile_x0 + cx * BLK_W, by0 = tile_y0 + (2 * wave + cy) * BLK_H;
                const int bx1 = bx0 + BLK_W - 1, by1 = by0 + BLK_H - 1;
                const float xp = xpa[cx], yp = ypa[cy];
                const float xif = (float)pxa[cx], yif = (float)pya[cy];
                for (int j0 = 0; j0 < n; j0 += WAVE) {
                    const int j = j0 + lane;
                    bool hit = false;
                    if (j < n) {
                        const BBox bb = s_bb[j];
                        hit = (bb.x_lo <= bx1) && (bb.x_hi >= bx0) && (bb.y_lo <= by1) && (bb.y_hi >= by0);
                    }
                    unsigned long long m = __ballot(hit);
                    while (m) {
                        const int t = __builtin_ctzll(m);
                        m &= m - 1;
                        const int fn = rfl(s_fn[j0 + t]);
                        const float *f = faces + (face_base + fn) * 9;
                        const float *iv = ws_inv + (face_base + fn) * 9;
                        test_face(f, iv, fn, xp, yp, xif, yif, near_d, far_d, st[r]);
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: every pixel is written (init values where no face was found, rasterize.py:478-496)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int px = pxa[r & 1], py = pya[r >> 1];
        if (px < S && py < S) {
            const size_t i = ((size_t)b * S + py) * S + px;
            face_index_map[i] = st[r].fn;
            if (depth_map) depth_map[i] = st[r].z;
            if (weight_map) {
                float *w = weight_map + 3 * i;
                w[0] = st[r].w0;
                w[1] = st[r].w1;
                w[2] = st[r].w2;
            }
            if (face_inv_map) {
                float *o = face_inv_map + 9 * i;
                if (st[r].fn >= 0) {
                    const float *iv = ws_inv + (face_base + st[r].fn) * 9;
#pragma unroll
                    for (int k = 0; k < 9; k++) o[k] = iv[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 9; k++) o[k] = 0.0f;
                }
            }
        }
    }
}

// --------------------------------------------------------------------------------------------------
// texture taps shared by F3 (forward) and B2 (backward recompute): rasterize.py:398-425
struct Taps {
    int isc[8];
    float w[8];
};

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

// F3: shading, one pixel per thread (linear pixel index: fully coalesced map traffic).
__global__ __launch_bounds__(256) void k_shade(const float *__restrict__ faces, const float *__restrict__ textures,
                                               const int32_t *__restrict__ face_index_map,
                                               const float *__restrict__ weight_map,
                                               const float *__restrict__ depth_map, float *__restrict__ rgb_map,
                                               int32_t *__restrict__ sampling_index_map,
                                               float *__restrict__ sampling_weight_map,
                                               const float *__restrict__ background, int bg_per_batch,
                                               float *__restrict__ alpha_map, int F, int S, int ts, double eps,
                                               int fix_batch_z, size_t n_pixels)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    const int fi = face_index_map[i];
    const int b = (int)(i / ((size_t)S * S));
    if (alpha_map) alpha_map[i] = (fi >= 0) ? 1.0f : 0.0f;  // :449
    if (!rgb_map) return;
    float rgb[3];
    Taps t;
    if (fi >= 0) {
        const float *face = faces + ((size_t)(fix_batch_z ? b : 0) * F + fi) * 9;  // :389 (Q1)
        const float *texture = textures + ((size_t)b * F + fi) * ts * ts * ts * 3;   // :390
        const float w[3] = {weight_map[3 * i], weight_map[3 * i + 1], weight_map[3 * i + 2]};
        compute_taps(face, w, depth_map[i], ts, eps, t);
        rgb[0] = rgb[1] = rgb[2] = 0.0f;
#pragma unroll
        for (int pn = 0; pn < 8; pn++) {
            const float *tx = texture + t.isc[pn] * 3;
            rgb[0] += t.w[pn] * tx[0];
            rgb[1] += t.w[pn] * tx[1];
            rgb[2] += t.w[pn] * tx[2];
        }
        // :463 with mask = 1: rgb * 1 + 0 * bg (kept literal: it maps -0 to +0 and NaN backgrounds to NaN)
        const float *bg = background + (bg_per_batch ? 3 * b : 0);
#pragma unroll
        for (int k = 0; k < 3; k++) rgb[k] = rgb[k] * 1.0f + 0.0f * bg[k];
    } else {
        const float *bg = background + (bg_per_batch ? 3 * b : 0);
#pragma unroll
        for (int k = 0; k < 3; k++) rgb[k] = 0.0f * 0.0f + 1.0f * bg[k];
#pragma unroll
        for (int pn = 0; pn < 8; pn++) { t.isc[pn] = 0; t.w[pn] = 0.0f; }
    }
    float *o = rgb_map + 3 * i;
    o[0] = rgb[0];
    o[1] = rgb[1];
    o[2] = rgb[2];
    if (sampling_index_map) {
        int4 *p = reinterpret_cast<int4 *>(sampling_index_map + 8 * i);
        p[0] = make_int4(t.isc[0], t.isc[1], t.isc[2], t.isc[3]);
        p[1] = make_int4(t.isc[4], t.isc[5], t.isc[6], t.isc[7]);
    }
    if (sampling_weight_map) {
        float4 *p = reinterpret_cast<float4 *>(sampling_weight_map + 8 * i);
        p[0] = make_float4(t.w[0], t.w[1], t.w[2], t.w[3]);
        p[1] = make_float4(t.w[4], t.w[5], t.w[6], t.w[7]);
    }
}

// --------------------------------------------------------------------------------------------------
// B1: backward_pixel_map (rasterize.py:517-748).
//
// Work decomposition.  The reference runs ONE thread per face through 3 edges x 2 axes x every integer
// column/row d0 crossed by the edge x two pixel sweeps along d1 (an "out" sweep from the edge to the image
// border, an "in" sweep from the edge to the opposite edge).  Here a lane owns one (face, edge, axis)
// item -- 10 faces x 6 items per wave -- and walks its d0 range; for each line it sets the sweep up
// (crossing point, in/out pixels, reference colours, the two distance coefficients) and then
//   * sweeps of at most SERIAL_MAX pixels are walked by the owning lane (divergent but short),
//   * longer sweeps are queued and served one at a time by the WHOLE wave: lane l visits pixel
//     d1_from + l, + 64, ...; the two partial sums are reduced across the wave and credited to the owner.
// Every per-pixel term is computed with the reference's arithmetic (same operations, same precision);
// only the order of the float additions differs (per-sweep tree + per-item running sum instead of one
// serial sum per face).  The six (vertex, x|y) results of a face are exchanged between its six lanes
// and STORED (no atomics): grad_faces needs no zero fill and the result is run-to-run deterministic.
constexpr int FACES_PER_WAVE = 10;
constexpr int BPM_THREADS = 256;
constexpr int SERIAL_MAX = 4;

template <bool RGB, bool ALPHA>
struct SweepCtx {
    const int32_t *__restrict__ fi_map;
    const float *__restrict__ rgb_map;
    const float *__restrict__ alpha_map;
    const float *__restrict__ g_rgb;
    const float *__restrict__ g_alpha;
    double eps;
    double two_over_s;  // 2/S, exact when S is a power of two
    double s_d;
    bool s_pow2;
};

// one pixel visit: rasterize.py:630-657 (out) / :697-728 (in)
template <bool RGB, bool ALPHA>
__device__ __forceinline__ void visit(const SweepCtx<RGB, ALPHA> &c, size_t idx, int d1, bool mode_in, int fn,
                                      float ref_a, float ref_r, float ref_g, float ref_b, float d1_cross, float c0,
                                      float c1, bool has0, bool has1, float &acc0, float &acc1)
{
    if (mode_in && c.fi_map[idx] != fn) return;  // :707
    float diff = 0.0f;
    if (ALPHA) diff += (c.alpha_map[idx] - ref_a) * c.g_alpha[idx];
    if (RGB) {
        const float *p = c.rgb_map + 3 * idx;
        const float *g = c.g_rgb + 3 * idx;
        diff += (p[0] - ref_r) * g[0];
        diff += (p[1] - ref_g) * g[1];
        diff += (p[2] - ref_b) * g[2];
    }
    if (diff <= 0.0f) return;  // :647 / :717
    const float t = (float)d1 - d1_cross;
    if (has0) {  // :648-652
        const double q = (double)(c0 * t) * 2.0;
        float dist = (float)(c.s_pow2 ? (double)(c0 * t) * c.two_over_s : q / c.s_d);
        dist = (0.0f < dist) ? (float)((double)dist + c.eps) : (float)((double)dist - c.eps);
        acc0 -= diff / dist;
    }
    if (has1) {  // :653-657
        const double q = (double)(c1 * t) * 2.0;
        float dist = (float)(c.s_pow2 ? (double)(c1 * t) * c.two_over_s : q / c.s_d);
        dist = (0.0f < dist) ? (float)((double)dist + c.eps) : (float)((double)dist - c.eps);
        acc1 -= diff / dist;
    }
}

template <bool RGB, bool ALPHA>
__global__ __launch_bounds__(BPM_THREADS) void k_backward_pixel_map(
    const float *__restrict__ faces, const int32_t *__restrict__ fi_map, const float *__restrict__ rgb_map,
    const float *__restrict__ alpha_map, const float *__restrict__ g_rgb, const float *__restrict__ g_alpha,
    float *__restrict__ grad_faces, int n_faces_total, int F, int S, double eps)
{
    SweepCtx<RGB, ALPHA> c;
    c.fi_map = fi_map; c.rgb_map = rgb_map; c.alpha_map = alpha_map; c.g_rgb = g_rgb; c.g_alpha = g_alpha;
    c.eps = eps; c.s_d = (double)S; c.two_over_s = 2.0 / (double)S; c.s_pow2 = (S & (S - 1)) == 0;

    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * (BPM_THREADS / WAVE) + (threadIdx.x >> 6);
    const int fl = lane / 6, item = lane - fl * 6;
    const int edge = item >> 1, axis = item & 1;
    const int gi = wave_global * FACES_PER_WAVE + fl;  // global face index b * F + fn
    const bool lane_valid = (fl < FACES_PER_WAVE) && (gi < n_faces_total);

    float g0 = 0.0f, g1 = 0.0f;  // running sums for vertex pi[0] / pi[1], coordinate (1 - axis)

    // ---- item setup: rasterize.py:536-569
    int b = 0, fn = 0, n_lines = 0, d0_from = 0, direction = 1;
    float p0x = 0, p0y = 0, p1x = 0, p1y = 0, p2x = 0, p2y = 0, slope = 0;
    if (lane_valid) {
        b = gi / F;
        fn = gi - b * F;
        const float *f = faces + (size_t)gi * 9;
        const float fx[3] = {f[0], f[3], f[6]}, fy[3] = {f[1], f[4], f[7]};
        if (!is_backside(fx[0], fy[0], fx[1], fy[1], fx[2], fy[2])) {
            const float fs = (float)S;
            const int i0 = edge, i1 = (edge + 1) % 3, i2 = (edge + 2) % 3;
            const float ppx[3] = {to_pixel(fx[i0], fs), to_pixel(fx[i1], fs), to_pixel(fx[i2], fs)};
            const float ppy[3] = {to_pixel(fy[i0], fs), to_pixel(fy[i1], fs), to_pixel(fy[i2], fs)};
            // p[num][dim] = pp[num][(dim + axis) % 2]: axis 1 swaps the roles of x and y (:556)
            p0x = axis ? ppy[0] : ppx[0]; p0y = axis ? ppx[0] : ppy[0];
            p1x = axis ? ppy[1] : ppx[1]; p1y = axis ? ppx[1] : ppy[1];
            p2x = axis ? ppy[2] : ppx[2]; p2y = axis ? ppx[2] : ppy[2];
            if (axis == 0) direction = (p0x < p1x) ? -1 : 1; else direction = (p0x < p1x) ? 1 : -1;  // :559-564
            d0_from = (int)fmax((double)ceilf(fminf(p0x, p1x)), 0.0);        // :568
            const int d0_to = (int)fmin((double)fmaxf(p0x, p1x), S - 1.0);   // :569
            // p0x == p1x: the only possible d0 equals both -> both contributions are skipped (:648,:653)
            if (p0x != p1x && d0_to >= d0_from) n_lines = d0_to - d0_from + 1;
            slope = (p1y - p0y) / (p1x - p0x);  // :573, invariant along the edge
        }
    }
    // strides of d0 / d1 in the row-major maps (:587-593)
    const size_t sd0 = axis ? (size_t)S : 1, sd1 = axis ? 1 : (size_t)S;
    const size_t img_base = (size_t)b * S * S;

    int max_lines = n_lines;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) max_lines = max(max_lines, __shfl_xor(max_lines, o, WAVE));

    for (int it = 0; it < max_lines; ++it) {
        // per-line state
        bool ok = false, is_in_fn = false, has0 = false, has1 = false;
        int d1_in = 0, d1_out = 0, in_from = 0, in_to = -1, out_from = 0, out_to = -1;
        float d1_cross = 0, c0 = 0, c1 = 0;
        float in_a = 0, in_r = 0, in_g = 0, in_b = 0, out_a = 0, out_r = 0, out_g = 0, out_b = 0;
        size_t line_base = 0;
        if (it < n_lines) {
            const int d0 = d0_from + it;
            const float d0f = (float)d0;
            d1_cross = slope * (d0f - p0x) + p0y;                                            // :573
            d1_in = (0 < direction) ? (int)floorf(d1_cross) : (int)ceilf(d1_cross);          // :574
            d1_out = d1_in + direction;                                                      // :575
            ok = !(d1_in < 0 || S <= d1_in) && !(d1_out < 0 || S <= d1_out);                 // :578-579
            if (ok) {
                line_base = img_base + (size_t)d0 * sd0;
                const size_t idx_in = line_base + (size_t)d1_in * sd1, idx_out = line_base + (size_t)d1_out * sd1;
                if (ALPHA) { in_a = alpha_map[idx_in]; out_a = alpha_map[idx_out]; }          // :594-597
                if (RGB) {                                                                   // :598-601
                    in_r = rgb_map[3 * idx_in]; in_g = rgb_map[3 * idx_in + 1]; in_b = rgb_map[3 * idx_in + 2];
                    out_r = rgb_map[3 * idx_out]; out_g = rgb_map[3 * idx_out + 1]; out_b = rgb_map[3 * idx_out + 2];
                }
                is_in_fn = (fi_map[idx_in] == fn);                                           // :604
                has0 = (p1x != d0f);
                has1 = (p0x != d0f);
                c0 = (p1x - p0x) / (p1x - d0f);   // :649 leading factor, invariant along the sweep
                c1 = (p1x - p0x) / (d0f - p0x);   // :654
                if (is_in_fn) {                    // :606-609
                    const int lim = (0 < direction) ? S - 1 : 0;
                    out_from = max(min(d1_out, lim), 0);
                    out_to = min(max(d1_out, lim), S - 1);
                }
                float d0_cross2;                   // :665-672
                if ((d0f - p0x) * (d0f - p2x) < 0)
                    d0_cross2 = (p2y - p0y) / (p2x - p0x) * (d0f - p0x) + p0y;
                else
                    d0_cross2 = (p1y - p2y) / (p1x - p2x) * (d0f - p2x) + p2y;
                const int lim2 = (0 < direction) ? (int)ceilf(d0_cross2) : (int)floorf(d0_cross2);
                in_from = max(min(d1_in, lim2), 0);
                in_to = min(max(d1_in, lim2), S - 1);
            }
        }

        // ---- short sweeps: walked by the owning lane
        const int out_len = out_to - out_from + 1, in_len = in_to - in_from + 1;
        const bool out_serial = ok && out_len > 0 && out_len <= SERIAL_MAX;
        const bool in_serial = ok && in_len > 0 && in_len <= SERIAL_MAX;
        if (out_serial)
            for (int d1 = out_from; d1 <= out_to; ++d1)
                visit<RGB, ALPHA>(c, line_base + (size_t)d1 * sd1, d1, false, fn, in_a, in_r, in_g, in_b, d1_cross,
                                  c0, c1, has0, has1, g0, g1);
        if (in_serial)
            for (int d1 = in_from; d1 <= in_to; ++d1)
                visit<RGB, ALPHA>(c, line_base + (size_t)d1 * sd1, d1, true, fn, out_a, out_r, out_g, out_b,
                                  d1_cross, c0, c1, has0, has1, g0, g1);

        // ---- long sweeps: served by the whole wave, one sweep at a time
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const bool mode_in = (pass == 1);
            const bool want = ok && (mode_in ? (in_len > SERIAL_MAX) : (out_len > SERIAL_MAX));
            unsigned long long m = __ballot(want);
            while (m) {
                const int src = __builtin_ctzll(m);
                m &= m - 1;
                const int s_from = bcast_i(mode_in ? in_from : out_from, src);
                const int s_to = bcast_i(mode_in ? in_to : out_to, src);
                const int s_fn = bcast_i(fn, src);
                const int s_axis = bcast_i(axis, src);
                const unsigned lb_lo = (unsigned)bcast_i((int)(unsigned)(line_base & 0xffffffffu), src);
                const unsigned lb_hi = (unsigned)bcast_i((int)(unsigned)(line_base >> 32), src);
                const size_t s_base = ((size_t)lb_hi << 32) | lb_lo;
                const size_t s_sd1 = s_axis ? 1 : (size_t)S;
                const float s_cross = bcast_f(d1_cross, src);
                const float s_c0 = bcast_f(c0, src), s_c1 = bcast_f(c1, src);
                const bool s_has0 = bcast_i((int)has0, src) != 0, s_has1 = bcast_i((int)has1, src) != 0;
                float ra = 0, rr = 0, rg = 0, rb = 0;
                if (ALPHA) ra = bcast_f(mode_in ? out_a : in_a, src);
                if (RGB) {
                    rr = bcast_f(mode_in ? out_r : in_r, src);
                    rg = bcast_f(mode_in ? out_g : in_g, src);
                    rb = bcast_f(mode_in ? out_b : in_b, src);
                }
                float a0 = 0.0f, a1 = 0.0f;
                for (int d1 = s_from + lane; d1 <= s_to; d1 += WAVE)
                    visit<RGB, ALPHA>(c, s_base + (size_t)d1 * s_sd1, d1, mode_in, s_fn, ra, rr, rg, rb, s_cross, s_c0,
                                      s_c1, s_has0, s_has1, a0, a1);
                a0 = wave_sum(a0);
                a1 = wave_sum(a1);
                if (lane == src) { g0 += a0; g1 += a1; }
            }
        }
    }

    // ---- combine the six items of a face and store: component (vertex v, coord 1 - axis) =
    //      g0 of item (edge v, axis) + g1 of item (edge v + 2 mod 3, axis)      (pi[] of :547, :651, :656)
    const int partner = fl * 6 + 2 * ((edge + 2) % 3) + axis;
    const float g1_partner = __shfl(g1, partner & 63, WAVE);
    if (lane_valid) {
        float *o = grad_faces + (size_t)gi * 9 + 3 * edge;
        o[1 - axis] = g0 + g1_partner;
        if (axis == 0) o[2] = 0.0f;  // K6 never touches z
    }
}

// --------------------------------------------------------------------------------------------------
// B2: backward_textures (rasterize.py:750-792).  -munsafe-fp-atomics => global_atomic_add_f32.
__global__ __launch_bounds__(256) void k_backward_textures(
    const int32_t *__restrict__ face_index_map, const float *__restrict__ sampling_weight_map,
    const int32_t *__restrict__ sampling_index_map, const float *__restrict__ faces,
    const float *__restrict__ weight_map, const float *__restrict__ depth_map, const float *__restrict__ g_rgb,
    float *__restrict__ grad_textures, int F, int S, int ts, double eps, int fix_batch_z, size_t n_pixels)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    const int fi = face_index_map[i];
    if (fi < 0) return;
    const int b = (int)(i / ((size_t)S * S));
    Taps t;
    if (sampling_weight_map) {
#pragma unroll
        for (int pn = 0; pn < 8; pn++) {
            t.w[pn] = sampling_weight_map[8 * i + pn];
            t.isc[pn] = sampling_index_map[8 * i + pn];
        }
    } else {
        const float *face = faces + ((size_t)(fix_batch_z ? b : 0) * F + fi) * 9;
        const float w[3] = {weight_map[3 * i], weight_map[3 * i + 1], weight_map[3 * i + 2]};
        compute_taps(face, w, depth_map[i], ts, eps, t);
    }
    const float g[3] = {g_rgb[3 * i], g_rgb[3 * i + 1], g_rgb[3 * i + 2]};
    float *gt = grad_textures + ((size_t)b * F + fi) * ts * ts * ts * 3;
#pragma unroll
    for (int pn = 0; pn < 8; pn++) {
        float *p = gt + t.isc[pn] * 3;
        atomicAdd(p + 0, t.w[pn] * g[0]);  // :780
        atomicAdd(p + 1, t.w[pn] * g[1]);
        atomicAdd(p + 2, t.w[pn] * g[2]);
    }
}

// --------------------------------------------------------------------------------------------------
// B3: backward_depth_map (rasterize.py:794-847).
__global__ __launch_bounds__(256) void k_backward_depth_map(
    const float *__restrict__ faces, const float *__restrict__ depth_map, const int32_t *__restrict__ face_index_map,
    const float *__restrict__ face_inv_map, const float *__restrict__ weight_map, const float *__restrict__ g_depth,
    float *__restrict__ grad_faces, int F, int S, size_t n_pixels)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    const int fn = face_index_map[i];
    if (fn < 0) return;
    const int b = (int)(i / ((size_t)S * S));
    const float *face = faces + ((size_t)b * F + fn) * 9;
    float f[9];
#pragma unroll
    for (int k = 0; k < 9; k++) f[k] = face[k];
    float inv[9];
    if (face_inv_map) {
#pragma unroll
        for (int k = 0; k < 9; k++) inv[k] = face_inv_map[9 * i + k];
    } else {
        const float fs = (float)S;
        const float px[3] = {to_pixel(f[0], fs), to_pixel(f[3], fs), to_pixel(f[6], fs)};
        const float py[3] = {to_pixel(f[1], fs), to_pixel(f[4], fs), to_pixel(f[7], fs)};
        compute_face_inv(px, py, inv);
    }
    const float depth = depth_map[i];
    const float depth2 = depth * depth;
    const float w[3] = {weight_map[3 * i], weight_map[3 * i + 1], weight_map[3 * i + 2]};
    const float gd = g_depth[i];
    float *gf = grad_faces + ((size_t)b * F + fn) * 9;
    // :824-827
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float z_k = f[3 * k + 2];
        atomicAdd(gf + 3 * k + 2, gd * w[k] * depth2 / (z_k * z_k));
    }
    // :830-837
    float tmp[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int l = 0; l < 3; l++) tmp[k] += -inv[3 * l + k] / f[3 * l + 2];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int l = 0; l < 2; l++) atomicAdd(gf + 3 * k + l, -gd * tmp[l] * w[k] * depth2 * (float)S / 2.0f);
}

// --------------------------------------------------------------------------------------------------
// host side
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

}  // namespace

// ====================================================================================================
NR_API int nr_version(void) { return NR_VERSION; }

NR_API const char *nr_error_string(int code)
{
    switch (code) {
        case 0: return "success";
        case NR_E_NULL: return "nr: a required pointer is NULL";
        case NR_E_SIZE: return "nr: size out of range";
        case NR_E_WORKSPACE: return "nr: workspace missing or too small";
        case NR_E_MODE: return "nr: nothing to do / inconsistent optional arguments";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "nr: unknown error";
    }
}

NR_API size_t nr_forward_workspace_bytes(int32_t B, int32_t F, int32_t S)
{
    if (check_sizes(B, F, S)) return 0;
    const size_t n = (size_t)B * F;
    return align_up(n * 9 * sizeof(float), 256) + align_up(n * sizeof(BBox), 256);
}

NR_API size_t nr_backward_workspace_bytes(int32_t B, int32_t F, int32_t S, int32_t return_rgb, int32_t return_alpha)
{
    (void)return_rgb; (void)return_alpha;
    if (check_sizes(B, F, S)) return 0;
    return 0;  // the first-generation kernel sweeps the row-major maps directly
}

NR_API int nr_forward_face_index_map(const float *faces, int32_t *face_index_map, float *weight_map, float *depth_map,
                                     float *face_inv_map, int32_t B, int32_t F, int32_t S, double near, double far,
                                     void *workspace, size_t workspace_bytes, void *stream)
{
    if (!faces || !face_index_map) return NR_E_NULL;
    if (int e = check_sizes(B, F, S)) return e;
    if (!workspace || workspace_bytes < nr_forward_workspace_bytes(B, F, S)) return NR_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)B * F;
    float *ws_inv = (float *)workspace;
    BBox *ws_bbox = (BBox *)((char *)workspace + align_up(n * 9 * sizeof(float), 256));

    hipLaunchKernelGGL(k_face_setup, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, faces, ws_inv, ws_bbox,
                       (int)n, S);
    const int tiles = (S + TILE - 1) / TILE;
    hipLaunchKernelGGL(k_raster_tiles, dim3(tiles * tiles, B), dim3(RASTER_THREADS), 0, st, faces, ws_inv, ws_bbox,
                       face_index_map, weight_map, depth_map, face_inv_map, F, S, tiles, near, far);
    return launch_status();
}

NR_API int nr_forward_texture_sampling(const float *faces, const float *textures, const int32_t *face_index_map,
                                       const float *weight_map, const float *depth_map, float *rgb_map,
                                       int32_t *sampling_index_map, float *sampling_weight_map,
                                       const float *background, int32_t bg_per_batch, float *alpha_map, int32_t B,
                                       int32_t F, int32_t S, int32_t ts, double eps, int32_t flags, void *stream)
{
    if (!face_index_map) return NR_E_NULL;
    if (!rgb_map && !alpha_map) return NR_E_MODE;
    if (int e = check_sizes(B, F, S)) return e;
    if (rgb_map) {
        if (!faces || !textures || !weight_map || !depth_map || !background) return NR_E_NULL;
        if (ts < 2 || ts > 1024) return NR_E_SIZE;
        if ((sampling_index_map == nullptr) != (sampling_weight_map == nullptr)) return NR_E_MODE;
    }
    const size_t n = (size_t)B * S * S;
    hipLaunchKernelGGL(k_shade, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, faces, textures,
                       face_index_map, weight_map, depth_map, rgb_map, sampling_index_map, sampling_weight_map,
                       background, bg_per_batch, alpha_map, F, S, ts, eps,
                       (flags & NR_FLAG_FIX_TEXTURE_BATCH_Z) ? 1 : 0, n);
    return launch_status();
}

NR_API int nr_backward_pixel_map(const float *faces, const int32_t *face_index_map, const float *rgb_map,
                                 const float *alpha_map, const float *grad_rgb_map, const float *grad_alpha_map,
                                 float *grad_faces, int32_t B, int32_t F, int32_t S, double eps, int32_t return_rgb,
                                 int32_t return_alpha, void *workspace, size_t workspace_bytes, void *stream)
{
    (void)workspace; (void)workspace_bytes;
    if (!faces || !face_index_map || !grad_faces) return NR_E_NULL;
    if (!return_rgb && !return_alpha) return NR_E_MODE;  // rasterize.py:523-524 returns early; callers skip the call
    if (return_rgb && (!rgb_map || !grad_rgb_map)) return NR_E_NULL;
    if (return_alpha && (!alpha_map || !grad_alpha_map)) return NR_E_NULL;
    if (int e = check_sizes(B, F, S)) return e;
    const int n = B * F;
    const int waves = (n + FACES_PER_WAVE - 1) / FACES_PER_WAVE;
    const dim3 grid((unsigned)((waves + BPM_THREADS / WAVE - 1) / (BPM_THREADS / WAVE))), block(BPM_THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (return_rgb && return_alpha)
        hipLaunchKernelGGL((k_backward_pixel_map<true, true>), grid, block, 0, st, faces, face_index_map, rgb_map,
                           alpha_map, grad_rgb_map, grad_alpha_map, grad_faces, n, F, S, eps);
    else if (return_rgb)
        hipLaunchKernelGGL((k_backward_pixel_map<true, false>), grid, block, 0, st, faces, face_index_map, rgb_map,
                           alpha_map, grad_rgb_map, grad_alpha_map, grad_faces, n, F, S, eps);
    else
        hipLaunchKernelGGL((k_backward_pixel_map<false, true>), grid, block, 0, st, faces, face_index_map, rgb_map,
                           alpha_map, grad_rgb_map, grad_alpha_map, grad_faces, n, F, S, eps);
    return launch_status();
}

NR_API int nr_backward_textures(const int32_t *face_index_map, const float *sampling_weight_map,
                                const int32_t *sampling_index_map, const float *faces, const float *weight_map,
                                const float *depth_map, const float *grad_rgb_map, float *grad_textures, int32_t B,
                                int32_t F, int32_t S, int32_t ts, double eps, int32_t flags, void *stream)
{
    if (!face_index_map || !grad_rgb_map || !grad_textures) return NR_E_NULL;
    if ((sampling_index_map == nullptr) != (sampling_weight_map == nullptr)) return NR_E_MODE;
    if (!sampling_weight_map && (!faces || !weight_map || !depth_map)) return NR_E_NULL;
    if (int e = check_sizes(B, F, S)) return e;
    if (ts < 2 || ts > 1024) return NR_E_SIZE;
    const size_t n = (size_t)B * S * S;
    hipLaunchKernelGGL(k_backward_textures, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       face_index_map, sampling_weight_map, sampling_index_map, faces, weight_map, depth_map,
                       grad_rgb_map, grad_textures, F, S, ts, eps, (flags & NR_FLAG_FIX_TEXTURE_BATCH_Z) ? 1 : 0, n);
    return launch_status();
}

NR_API int nr_backward_depth_map(const float *faces, const float *depth_map, const int32_t *face_index_map,
                                 const float *face_inv_map, const float *weight_map, const float *grad_depth_map,
                                 float *grad_faces, int32_t B, int32_t F, int32_t S, void *stream)
{
    if (!faces || !depth_map || !face_index_map || !weight_map || !grad_depth_map || !grad_faces) return NR_E_NULL;
    if (int e = check_sizes(B, F, S)) return e;
    const size_t n = (size_t)B * S * S;
    hipLaunchKernelGGL(k_backward_depth_map, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       faces, depth_map, face_index_map, face_inv_map, weight_map, grad_depth_map, grad_faces, F, S,
                       n);
    return launch_status();
}
