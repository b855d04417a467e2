// nr_forward.hip -- forward kernels of the gfx950 rasterizer + their C-ABI entry points.
//
//   k_face_setup    F1  per face: back-face cull, inverse barycentric matrix, screen bbox        (ref K1, :240-277)
//   k_raster_tiles  F2  per 32x32 tile: bbox-scan of the image's faces, survivors' geometry staged in LDS,
//                       then one pixel per lane (16x4 blocks per wave) resolves min-depth over the list
//                                                                                               (ref K2, :279-359)
//   k_shade         F3  per pixel: trilinear texture sampling + background + alpha        (ref K4+K5, :361-465)
#include "nr_device.h"

using namespace nr;

namespace {

// --------------------------------------------------------------------------------------------------
// F1: workspace = inv[B*F*9] floats (the reference's `faces_inv`, zeros for back faces), then bbox[B*F].
__global__ __launch_bounds__(256) void k_face_setup(const float *__restrict__ faces, float *__restrict__ ws_inv,
                                                    BBox *__restrict__ ws_bbox, int n_faces_total, int S)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_faces_total) return;
    const float *f = faces + (size_t)i * 9;
    const float x0 = f[0], y0 = f[1], x1 = f[3], y1 = f[4], x2 = f[6], y2 = f[7];
    float inv[9];
    if (is_backside(x0, y0, x1, y1, x2, y2)) {
#pragma unroll
        for (int k = 0; k < 9; k++) inv[k] = 0.0f;  // rasterize.py:240 zeros_like + :253 continue
    } else {
        const float fs = (float)S;
        const float px[3] = {to_pixel(x0, fs), to_pixel(x1, fs), to_pixel(x2, fs)};
        const float py[3] = {to_pixel(y0, fs), to_pixel(y1, fs), to_pixel(y2, fs)};
        compute_face_inv(px, py, inv);
    }
    float *o = ws_inv + (size_t)i * 9;
#pragma unroll
    for (int k = 0; k < 9; k++) o[k] = inv[k];
    ws_bbox[i] = face_bbox(x0, y0, x1, y1, x2, y2, S);
}

// --------------------------------------------------------------------------------------------------
// F2: tile rasterizer.  One workgroup (4 waves) per 32x32-pixel tile of one image.
//   scan   every thread tests one face box per round against the tile (boxes of 4 rounds are fetched together
//          so their memory latency overlaps); a hit is appended to an LDS list (wave-aggregated LDS atomic) and
//          the hitting thread copies the face's 9 + 9 floats (vertices, inverse matrix) into the list entry;
//   raster when the list is nearly full (or the faces are exhausted) the entries are rasterized two ways:
//          SMALL faces (box-in-tile area <= SMALL_AREA pixels; the bulk of a fine mesh) are face-parallel: one
//          lane per face walks the face's few pixels and publishes (depth bits << 32 | face index) with a
//          64-bit LDS atomic min into the tile's z-buffer -- dense clusters of tiny faces (teapot knob: 185
//          faces over one 16x4 block) cost passes of 64 faces instead of 185 serial wave-wide tests;
//          LARGE faces are pixel-parallel: each wave walks its four 16x4 pixel blocks, 64 list entries per step
//          (one box test per lane, __ballot), and for every surviving entry all 64 lanes read the entry from
//          LDS (same address: broadcast) and test their own pixel, keeping the winner in registers;
//   resolve at the end each pixel merges the register winner with the LDS z-buffer winner and re-evaluates the
//          weights of an LDS winner (same function, same inputs -> same bits).
// Both paths evaluate the reference's inside / barycentric / depth arithmetic through eval_pixel(); the
// winner rule is "smaller zp, ties -> lower face index" (the reference scans faces in ascending order with
// a strict `<`, rasterize.py:300,334), which the packed 64-bit min reproduces because near > 0 makes the
// float bit pattern order-preserving.
constexpr int TILE = 32;
constexpr int BLK_W = 16, BLK_H = 4;
constexpr int RASTER_THREADS = 256;
constexpr int LIST_CAP = 384;       // entries; flushed when fewer than 256 slots remain
constexpr int ENTRY_F = 20;         // floats per entry: 9 vertices + 9 inverse + 2 pad (80 B)
constexpr int SMALL_AREA = 128;     // box-in-tile pixels up to which a face takes the face-parallel path
constexpr unsigned long long ZEMPTY = ~0ull;

struct FaceGeo {
    float x0, y0, z0, x1, y1, z1, x2, y2, z2, i0, i1, i2, i3, i4, i5, i6, i7, i8;
};

__device__ __forceinline__ FaceGeo load_geo(const float *__restrict__ e)
{
    const float4 a = *reinterpret_cast<const float4 *>(e);       // x0 y0 z0 x1
    const float4 b = *reinterpret_cast<const float4 *>(e + 4);   // y1 z1 x2 y2
    const float4 c = *reinterpret_cast<const float4 *>(e + 8);   // z2 i0 i1 i2
    const float4 d = *reinterpret_cast<const float4 *>(e + 12);  // i3 i4 i5 i6
    const float2 g = *reinterpret_cast<const float2 *>(e + 16);  // i7 i8
    FaceGeo q;
    q.x0 = a.x; q.y0 = a.y; q.z0 = a.z; q.x1 = a.w; q.y1 = b.x; q.z1 = b.y; q.x2 = b.z; q.y2 = b.w; q.z2 = c.x;
    q.i0 = c.y; q.i1 = c.z; q.i2 = c.w; q.i3 = d.x; q.i4 = d.y; q.i5 = d.z; q.i6 = d.w; q.i7 = g.x; q.i8 = g.y;
    return q;
}

// One (face, pixel) evaluation of the reference's K2 body.  Returns false when the pixel is rejected.
__device__ __forceinline__ bool eval_pixel(const FaceGeo &q, float xp, float yp, float xif, float yif, double near_d,
                                           double far_d, float &zp, float &w0, float &w1, float &w2)
{
    // rasterize.py:310-312 (back faces never reach here: their box is empty)
    if (((yp - q.y0) * (q.x1 - q.x0) < (xp - q.x0) * (q.y1 - q.y0)) ||
        ((yp - q.y1) * (q.x2 - q.x1) < (xp - q.x1) * (q.y2 - q.y1)) ||
        ((yp - q.y2) * (q.x0 - q.x2) < (xp - q.x2) * (q.y0 - q.y2)))
        return false;
    // :317-327
    w0 = q.i0 * xif + q.i1 * yif + q.i2;
    w1 = q.i3 * xif + q.i4 * yif + q.i5;
    w2 = q.i6 * xif + q.i7 * yif + q.i8;
    w0 = fminf(fmaxf(w0, 0.0f), 1.0f);
    w1 = fminf(fmaxf(w1, 0.0f), 1.0f);
    w2 = fminf(fmaxf(w2, 0.0f), 1.0f);
    const float w_sum = (0.0f + w0) + w1 + w2;
    w0 /= w_sum;
    w1 /= w_sum;
    w2 /= w_sum;
    // :330 -- double reciprocal of a float rounded to float == correctly rounded float division
    zp = 1.0f / (w0 / q.z0 + w1 / q.z1 + w2 / q.z2);
    if ((double)zp <= near_d || far_d <= (double)zp) return false;  // :331
    return true;
}

struct PixelState {
    float z, w0, w1, w2;
    int fn;
};

__global__ __launch_bounds__(RASTER_THREADS) void k_raster_tiles(
    const float *__restrict__ faces, const float *__restrict__ ws_inv, const BBox *__restrict__ ws_bbox,
    int32_t *__restrict__ face_index_map, float *__restrict__ weight_map, float *__restrict__ depth_map,
    float *__restrict__ face_inv_map, int F, int S, int tiles_x, double near_d, double far_d)
{
    __shared__ __attribute__((aligned(16))) float s_geo[LIST_CAP * ENTRY_F];
    __shared__ unsigned long long s_zbuf[TILE * TILE];
    __shared__ int s_fn[LIST_CAP];
    __shared__ BBox s_bb[LIST_CAP];
    __shared__ float s_xp[TILE], s_yp[TILE];
    __shared__ int s_cnt;

    const int b = blockIdx.y;
    const int tile_x0 = (blockIdx.x % tiles_x) * TILE;
    const int tile_y0 = (blockIdx.x / tiles_x) * TILE;
    const int tile_x1 = min(tile_x0 + TILE, S) - 1, tile_y1 = min(tile_y0 + TILE, S) - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t face_base = (size_t)b * F;

    // pixel centres of the tile's columns / rows (rasterize.py:291-292)
    if (tid < TILE) s_xp[tid] = pixel_center(tile_x0 + tid, S);
    else if (tid < 2 * TILE) s_yp[tid - TILE] = pixel_center(tile_y0 + tid - TILE, S);
    for (int i = tid; i < TILE * TILE; i += RASTER_THREADS) s_zbuf[i] = ZEMPTY;
    if (tid == 0) s_cnt = 0;

    // this lane's pixels: block r of wave w sits at column block (r & 1), row block (2 * w + (r >> 1))
    const int lx = lane & (BLK_W - 1), ly = lane >> 4;
    const int lxa[2] = {lx, BLK_W + lx};
    const int lya[2] = {(2 * wave) * BLK_H + ly, (2 * wave + 1) * BLK_H + ly};

    const float far_f = (float)far_d;  // rasterize.py:296
    PixelState st[4];
#pragma unroll
    for (int r = 0; r < 4; r++) { st[r].z = far_f; st[r].fn = -1; st[r].w0 = st[r].w1 = st[r].w2 = 0.0f; }
    __syncthreads();

    constexpr int PF = 4;  // scan rounds whose boxes are fetched together
    for (int base0 = 0; base0 < F; base0 += PF * RASTER_THREADS) {
        BBox bbs[PF];
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int fn = base0 + u * RASTER_THREADS + tid;
            bbs[u].x_lo = 1; bbs[u].x_hi = 0; bbs[u].y_lo = 1; bbs[u].y_hi = 0;
            if (fn < F) bbs[u] = ws_bbox[face_base + fn];
        }
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int base = base0 + u * RASTER_THREADS;
            if (base >= F) break;  // uniform
            // ---- scan: one box per thread
            const int fn = base + tid;
            const BBox bb = bbs[u];
            const bool hit = (bb.x_lo <= bb.x_hi) && (bb.x_lo <= tile_x1) && (bb.x_hi >= tile_x0) &&
                             (bb.y_lo <= tile_y1) && (bb.y_hi >= tile_y0);
            const unsigned long long m = __ballot(hit);
            if (m) {
                int off = 0;
                if (lane == 0) off = atomicAdd(&s_cnt, __popcll(m));
                off = rfl(off);
                if (hit) {
                    const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
                    s_fn[pos] = fn;
                    s_bb[pos] = bb;
                    const float *f = faces + (face_base + fn) * 9;
                    const float *iv = ws_inv + (face_base + fn) * 9;
                    float *e = s_geo + pos * ENTRY_F;
#pragma unroll
                    for (int k = 0; k < 9; k++) e[k] = f[k];
#pragma unroll
                    for (int k = 0; k < 9; k++) e[9 + k] = iv[k];
                }
            }
            __syncthreads();
            const int n = s_cnt;
            __syncthreads();  // everybody has read n before the next round's appends can move s_cnt
            const bool last = base + RASTER_THREADS >= F;
            if (n > LIST_CAP - RASTER_THREADS || (last && n > 0)) {
                // ---- small faces: lane = face, LDS z-buffer
                for (int j = tid; j < n; j += RASTER_THREADS) {
                    const BBox q = s_bb[j];
                    const int x_lo = max((int)q.x_lo, tile_x0), x_hi = min((int)q.x_hi, tile_x1);
                    const int y_lo = max((int)q.y_lo, tile_y0), y_hi = min((int)q.y_hi, tile_y1);
                    if ((x_hi - x_lo + 1) * (y_hi - y_lo + 1) > SMALL_AREA) continue;
                    const FaceGeo g = load_geo(s_geo + j * ENTRY_F);
                    const unsigned fnu = (unsigned)s_fn[j];
                    for (int py = y_lo; py <= y_hi; py++) {
                        const float yp = s_yp[py - tile_y0], yif = (float)py;
                        for (int px = x_lo; px <= x_hi; px++) {
                            float zp, w0, w1, w2;
                            if (eval_pixel(g, s_xp[px - tile_x0], yp, (float)px, yif, near_d, far_d, zp, w0, w1, w2))
                                atomicMin(&s_zbuf[(py - tile_y0) * TILE + (px - tile_x0)],
                                          ((unsigned long long)__float_as_uint(zp) << 32) | fnu);
                        }
                    }
                }
                // ---- large faces: lane = pixel
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int cx = r & 1, cy = r >> 1;
                    const int bx0 = tile_x0 + cx * BLK_W, by0 = tile_y0 + (2 * wave + cy) * BLK_H;
                    const int bx1 = bx0 + BLK_W - 1, by1 = by0 + BLK_H - 1;
                    const float xp = s_xp[lxa[cx]], yp = s_yp[lya[cy]];
                    const float xif = (float)(tile_x0 + lxa[cx]), yif = (float)(tile_y0 + lya[cy]);
                    for (int j0 = 0; j0 < n; j0 += WAVE) {
                        const int j = j0 + lane;
                        bool h2 = false;
                        if (j < n) {
                            const BBox q = s_bb[j];
                            const int x_lo = max((int)q.x_lo, tile_x0), x_hi = min((int)q.x_hi, tile_x1);
                            const int y_lo = max((int)q.y_lo, tile_y0), y_hi = min((int)q.y_hi, tile_y1);
                            h2 = ((x_hi - x_lo + 1) * (y_hi - y_lo + 1) > SMALL_AREA) && (q.x_lo <= bx1) &&
                                 (q.x_hi >= bx0) && (q.y_lo <= by1) && (q.y_hi >= by0);
                        }
                        unsigned long long mm = __ballot(h2);
                        while (mm) {
                            const int t = __builtin_ctzll(mm);
                            mm &= mm - 1;
                            const int idx = j0 + t;  // wave-uniform
                            const FaceGeo g = load_geo(s_geo + idx * ENTRY_F);
                            const int fn2 = s_fn[idx];
                            float zp, w0, w1, w2;
                            if (eval_pixel(g, xp, yp, xif, yif, near_d, far_d, zp, w0, w1, w2) &&
                                (zp < st[r].z || (zp == st[r].z && fn2 < st[r].fn))) {  // :334 + explicit tie rule
                                st[r].z = zp; st[r].fn = fn2; st[r].w0 = w0; st[r].w1 = w1; st[r].w2 = w2;
                            }
                        }
                    }
                }
                __syncthreads();
                if (tid == 0) s_cnt = 0;
                __syncthreads();
            }
        }
    }

    // ---- resolve + epilogue: every pixel is written (init values where no face was found, :478-496)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int lxx = lxa[r & 1], lyy = lya[r >> 1];
        const int px = tile_x0 + lxx, py = tile_y0 + lyy;
        if (px < S && py < S) {
            const unsigned long long pk = s_zbuf[lyy * TILE + lxx];
            if (pk != ZEMPTY) {
                const float zs = __uint_as_float((unsigned)(pk >> 32));
                const int fs_ = (int)(unsigned)(pk & 0xffffffffu);
                if (zs < st[r].z || (zs == st[r].z && fs_ < st[r].fn)) {
                    // re-evaluate the winner to get its weights (same inputs, same function -> same bits)
                    const float *f = faces + (face_base + fs_) * 9;
                    const float *iv = ws_inv + (face_base + fs_) * 9;
                    FaceGeo g;
                    g.x0 = f[0]; g.y0 = f[1]; g.z0 = f[2]; g.x1 = f[3]; g.y1 = f[4]; g.z1 = f[5];
                    g.x2 = f[6]; g.y2 = f[7]; g.z2 = f[8];
                    g.i0 = iv[0]; g.i1 = iv[1]; g.i2 = iv[2]; g.i3 = iv[3]; g.i4 = iv[4]; g.i5 = iv[5];
                    g.i6 = iv[6]; g.i7 = iv[7]; g.i8 = iv[8];
                    float zp, w0, w1, w2;
                    eval_pixel(g, s_xp[lxx], s_yp[lyy], (float)px, (float)py, near_d, far_d, zp, w0, w1, w2);
                    st[r].z = zp; st[r].fn = fs_; st[r].w0 = w0; st[r].w1 = w1; st[r].w2 = w2;
                }
            }
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
