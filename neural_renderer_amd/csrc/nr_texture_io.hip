// nr_texture_io.hip -- the data formats either side of the path: the two one-shot kernels of the reference's
// OBJ/MTL texture pipeline.
//
//   k_bake_textures        K10, neural_renderer/load_obj.py:87-144: texture image + per-face uv triangles -> the
//                          [Nf, ts, ts, ts, 3] texture cubes the rasterizer samples (bilinear lookup at the barycentric
//                          point of every texel).
//   k_texture_atlas        K11, neural_renderer/save_obj.py:32-113: texture cubes -> one atlas image of 16x16-pixel tiles
//   k_texture_atlas_seam        (save_obj.py:115-146: the column right of each tile's diagonal repeats its left neighbour).
//
// Both are pure gathers, one thread per output element, run once per mesh; they exist so that textured meshes enter and
// leave the rasterizer in exactly the reference's encoding.  Arithmetic: float32 in the reference's operation order, with
// its double literals (`ts - 1.`, `max(tif, 0.)`, the pasted eps) evaluated in double.
//
// Undefined behaviour of the reference that is given a definition here: image reads outside the image (uv exactly 1, or
// negative uv) are clamped to the nearest valid flat pixel index -- such reads carry weight 0 whenever the reference's
// own result is defined; atlas tiles beyond the last face stay 0.
#include "nr_device.h"

using namespace nr;

namespace {

__device__ __forceinline__ int f2i(float x) { return (int)x; }  // v_cvt_i32_f32: truncates, saturates, NaN -> 0 (as CUDA)

__global__ __launch_bounds__(256) void k_bake_textures(const float *__restrict__ image, const float *__restrict__ faces_uv,
                                                       const int32_t *__restrict__ is_update, float *__restrict__ textures,
                                                       size_t n, int ts, int H, int W)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int fn = (int)(i / ((size_t)ts * ts * ts));
    if (is_update[fn] == 0) return;  // load_obj.py:110
    float dim0 = (float)((double)((i / ((size_t)ts * ts)) % ts) / (ts - 1.));  // :98-100
    float dim1 = (float)((double)((i / ts) % ts) / (ts - 1.));
    float dim2 = (float)((double)(i % ts) / (ts - 1.));
    const float sum = dim0 + dim1 + dim2;  // :103 (0 at texel (0,0,0): the divisions give NaN, as in the reference)
    dim0 /= sum;
    dim1 /= sum;
    dim2 /= sum;
    const float *face = faces_uv + (size_t)fn * 6;
    const float pos_x = (face[0] * dim0 + face[2] * dim1 + face[4] * dim2) * (float)(W - 1);  // :112-113
    const float pos_y = (face[1] * dim0 + face[3] * dim1 + face[5] * dim2) * (float)(H - 1);  // :114-115
    const int xi = f2i(pos_x), yi = f2i(pos_y), yi1 = f2i(pos_y + 1.0f);
    const float wx1 = pos_x - (float)xi, wx0 = 1.0f - wx1;  // :118-121
    const float wy1 = pos_y - (float)yi, wy0 = 1.0f - wy1;
    const long long last = (long long)H * W - 1;
    auto px = [&](int row, int col) -> const float * {
        long long p = (long long)row * W + col;
        p = p < 0 ? 0 : (p > last ? last : p);
        return image + p * 3;
    };
    const float *p00 = px(yi, xi), *p10 = px(yi1, xi), *p01 = px(yi, xi + 1), *p11 = px(yi1, xi + 1);
    float *texture = textures + i * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) {  // :123-128
        float c = 0.0f;
        c += p00[k] * (wx0 * wy0);
        c += p10[k] * (wx0 * wy1);
        c += p01[k] * (wx1 * wy0);
        c += p11[k] * (wx1 * wy1);
        texture[k] = c;
    }
}

// one thread per atlas pixel (x, y); tile (x / tso, y / tso) belongs to face fn = x / tso + (y / tso) * tile_width
__global__ __launch_bounds__(256) void k_texture_atlas(float *__restrict__ image, const float *__restrict__ vertices_all,
                                                       const float *__restrict__ textures, size_t n, int num_faces, int tsi,
                                                       int tso, int tile_width)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int width = tile_width * tso;
    const int x = (int)(i % width);
    const int y = (int)(i / width);
    const int fn = x / tso + (y / tso) * tile_width;  // save_obj.py:39-41
    float *out = image + i * 3;
    if (fn >= num_faces) {
        out[0] = out[1] = out[2] = 0.0f;
        return;
    }
    const float *texture = textures + (size_t)fn * tsi * tsi * tsi * 3;
    const float *p0 = vertices_all + (size_t)fn * 6, *p1 = p0 + 2, *p2 = p0 + 4;

    float face_inv[9] = {p1[1] - p2[1], p2[0] - p1[0], p1[0] * p2[1] - p2[0] * p1[1],   // :54-57
                         p2[1] - p0[1], p0[0] - p2[0], p2[0] * p0[1] - p0[0] * p2[1],
                         p0[1] - p1[1], p1[0] - p0[0], p0[0] * p1[1] - p1[0] * p0[1]};
    const float den = p2[0] * (p0[1] - p1[1]) + p0[0] * (p1[1] - p2[1]) + p1[0] * (p2[1] - p0[1]);  // :58-61
#pragma unroll
    for (int k = 0; k < 9; k++) face_inv[k] /= den;

    float weight[3];
    float weight_sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; k++) {  // :65-69
        weight[k] = face_inv[3 * k + 0] * (float)x + face_inv[3 * k + 1] * (float)y + face_inv[3 * k + 2];
        weight_sum += weight[k];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) weight[k] = (float)((double)weight[k] / ((double)weight_sum + 1e-5));  // :70

    float tif[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {  // :73-79
        float t = weight[k] * (float)(tsi - 1);
        t = (float)fmax((double)t, 0.);
        t = (float)fmin((double)t, (double)(tsi - 1) - 1e-5);
        tif[k] = t;
    }

    float new_pixel[3] = {0.0f, 0.0f, 0.0f};
    for (int pn = 0; pn < 8; pn++) {  // :82-97
        float w = 1.0f;
        int idx[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int ti = f2i(tif[k]);
            if ((pn >> k) % 2 == 0) {
                w *= 1.0f - (tif[k] - (float)ti);
                idx[k] = ti;
            } else {
                w *= tif[k] - (float)ti;
                idx[k] = ti + 1;
            }
        }
        const int isc = idx[0] * tsi * tsi + idx[1] * tsi + idx[2];
#pragma unroll
        for (int k = 0; k < 3; k++) new_pixel[k] += w * texture[isc * 3 + k];
    }
    out[0] = new_pixel[0];
    out[1] = new_pixel[1];
    out[2] = new_pixel[2];
}

// save_obj.py:115-146: pixels one step right of a tile's diagonal copy their left neighbour
__global__ __launch_bounds__(256) void k_texture_atlas_seam(float *__restrict__ image, size_t n, int tso, int tile_width)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int width = tile_width * tso;
    const int x = (int)(i % width);
    const int y = (int)(i / width);
    if ((y % tso + 1) == (x % tso)) {
        const size_t src = (size_t)y * width + (x - 1);  // never itself a seam pixel: no race
#pragma unroll
        for (int k = 0; k < 3; k++) image[i * 3 + k] = image[src * 3 + k];
    }
}

}  // namespace

NR_API int nr_load_textures(const float *image, const float *faces_uv, const int32_t *is_update, float *textures,
                            int32_t num_faces, int32_t texture_size, int32_t image_height, int32_t image_width, void *stream)
{
    if (!image || !faces_uv || !is_update || !textures) return NR_E_NULL;
    if (num_faces < 1 || texture_size < 2 || image_height < 1 || image_width < 1) return NR_E_SIZE;
    const size_t n = (size_t)num_faces * texture_size * texture_size * texture_size;
    if (n > 0x7fffffffull * 64) return NR_E_SIZE;
    hipLaunchKernelGGL(k_bake_textures, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, image, faces_uv,
                       is_update, textures, n, texture_size, image_height, image_width);
    return launch_status();
}

NR_API int nr_create_texture_image(const float *textures, const float *tile_vertices, float *image, int32_t num_faces,
                                   int32_t texture_size_in, int32_t texture_size_out, int32_t tile_width, int32_t tile_height,
                                   void *stream)
{
    if (!textures || !tile_vertices || !image) return NR_E_NULL;
    if (num_faces < 1 || texture_size_in < 2 || texture_size_out < 2 || tile_width < 1 || tile_height < 1) return NR_E_SIZE;
    if ((long long)tile_width * tile_height < num_faces) return NR_E_SIZE;
    const size_t n = (size_t)tile_width * texture_size_out * (size_t)tile_height * texture_size_out;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_texture_atlas, grid, block, 0, st, image, tile_vertices, textures, n, num_faces, texture_size_in,
                       texture_size_out, tile_width);
    int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(k_texture_atlas_seam, grid, block, 0, st, image, n, texture_size_out, tile_width);
    return launch_status();
}
