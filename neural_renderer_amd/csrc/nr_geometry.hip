// nr_geometry.hip -- the caller immediately in front of the rasterizer (north_star "HIP atomics for
// the face->vertex gradient scatter"): vertices_to_faces (reference neural_renderer/vertices_to_faces.py:4-21) and
// its backward, which in the reference is Chainer's get_item backward (a scatter-add over the gathered rows).
//
// Measured motivation (scripts/glue_profile.py): through torch advanced indexing the backward of this gather is a
// sort-based index_put (rocprim merge sort + indexing_backward_kernel): 450 us per Renderer.render_silhouettes step at
// the headline size, 900 us per Renderer.render step (the gather is used twice there) -- more than the whole rasterizer
// forward + backward.  The scatter has low contention (a vertex is shared by ~6 faces, 12 with fill_back), which is
// exactly the case hardware float atomics are good at.
#include "nr_device.h"

using namespace nr;

namespace {

// one thread per (batch, face, corner): copies the corner's xyz
__global__ __launch_bounds__(256) void k_vertices_to_faces(const float *__restrict__ vertices,
                                                           const int32_t *__restrict__ faces_idx,
                                                           float *__restrict__ out, int Nv, size_t n_corners,
                                                           size_t corners_per_batch, int idx_per_batch)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_corners) return;
    const size_t b = i / corners_per_batch;
    // indices are validated on the host (IndexError, like the reference's get_item); the clamp only keeps a bad index that
    // slipped through a raw C-ABI caller inside the buffer
    const int32_t v = min(max(faces_idx[idx_per_batch ? i : i - b * corners_per_batch], 0), Nv - 1);
    const float *src = vertices + ((size_t)b * Nv + v) * 3;
    float *dst = out + i * 3;
    dst[0] = src[0];
    dst[1] = src[1];
    dst[2] = src[2];
}

// backward: grad_vertices[b, faces[b, f, k], :] += grad_faces[b, f, k, :]   (-munsafe-fp-atomics: global_atomic_add_f32)
__global__ __launch_bounds__(256) void k_vertices_to_faces_backward(const float *__restrict__ grad_faces,
                                                                    const int32_t *__restrict__ faces_idx,
                                                                    float *__restrict__ grad_vertices, int Nv,
                                                                    size_t n_corners, size_t corners_per_batch,
                                                                    int idx_per_batch)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_corners) return;
    const size_t b = i / corners_per_batch;
    const int32_t v = min(max(faces_idx[idx_per_batch ? i : i - b * corners_per_batch], 0), Nv - 1);
    const float *g = grad_faces + i * 3;
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    float *dst = grad_vertices + ((size_t)b * Nv + v) * 3;
    if (g0 != 0.0f) atomicAdd(dst + 0, g0);
    if (g1 != 0.0f) atomicAdd(dst + 1, g1);
    if (g2 != 0.0f) atomicAdd(dst + 2, g2);
}

}  // namespace

NR_API int nr_vertices_to_faces(const float *vertices, const int32_t *faces_idx, float *faces_out, int32_t B, int32_t Nv,
                                int32_t Nf, int32_t idx_per_batch, void *stream)
{
    if (!vertices || !faces_idx || !faces_out) return NR_E_NULL;
    if (B < 1 || Nv < 1 || Nf < 1) return NR_E_SIZE;
    const size_t cpb = (size_t)Nf * 3, n = (size_t)B * cpb;
    hipLaunchKernelGGL(k_vertices_to_faces, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vertices,
                       faces_idx, faces_out, Nv, n, cpb, idx_per_batch);
    return launch_status();
}

NR_API int nr_vertices_to_faces_backward(const float *grad_faces, const int32_t *faces_idx, float *grad_vertices,
                                         int32_t B, int32_t Nv, int32_t Nf, int32_t idx_per_batch, void *stream)
{
    if (!grad_faces || !faces_idx || !grad_vertices) return NR_E_NULL;
    if (B < 1 || Nv < 1 || Nf < 1) return NR_E_SIZE;
    hipStream_t st = (hipStream_t)stream;
    const int e = fill_bytes(grad_vertices, 0, (size_t)B * Nv * 3 * sizeof(float), st);
    if (e != 0) return e;
    const size_t cpb = (size_t)Nf * 3, n = (size_t)B * cpb;
    hipLaunchKernelGGL(k_vertices_to_faces_backward, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, grad_faces,
                       faces_idx, grad_vertices, Nv, n, cpb, idx_per_batch);
    return launch_status();
}
