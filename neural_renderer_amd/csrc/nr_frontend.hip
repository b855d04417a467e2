// nr_frontend.hip -- the caller immediately in front of the rasterizer, fused.
//
// One kernel replaces the chain `Renderer.render*` runs before `rasterize` (reference neural_renderer/renderer.py:35-107):
//   fill_back          faces ++ reversed faces, textures ++ transposed textures               renderer.py:37-38, 77-79
//   lighting           per-face ambient + Lambert factor multiplied into the textures          lighting.py:8-51
//   look_at / look     camera rotation built from eye / at (or direction) / up                 look_at.py:7-46, look.py:7-45
//   perspective        x/z/tan, y/z/tan, z                                                      perspective.py:5-19
//   vertices_to_faces  the gather into [B, F, 3, 3]                                             vertices_to_faces.py:4-21
// and one kernel (plus a per-image camera kernel) replaces its whole backward, including the gradient that flows from
// the lit textures back into the vertices through the face normals, the face->vertex scatter (hardware float atomics)
// and the gradient of a learnable camera position (example4 optimises `eye`).
//
// Why: in stock torch this chain is ~60 small launches forward and ~100 backward on [B,3]- to [B,Nv,3]-sized tensors,
// i.e. ~1.1 ms of host launch latency around a 0.9 ms rasterizer step at the headline size (scripts/renderer_e2e.py),
// plus three full passes over the [B, 2F, ts^3, 3] texture tensor (concat, multiply, and their backward).
//
// Work decomposition: LANES = 8 consecutive lanes per (image, face).  Every lane recomputes the (cheap) camera basis and
// the face's light colour; the lanes stride over the ts^3 texels, lanes 0/1 write the front/back copies of the face.
//
// Arithmetic follows the reference's float32 operation order where it is defined by the Python source
// (normalize = x / (|x| + 1e-5), light = ia*ca + id*(cd*cos), x / z / width); the 3x3 rotation is applied as
// ((t0*r0 + t1*r1) + t2*r2), which is one valid evaluation order of the reference's BLAS matmul.
#include "nr_device.h"

using namespace nr;

namespace {

constexpr int FE_LANES = 8;
constexpr int FE_THREADS = 256;
constexpr float NORM_EPS = 1e-5f;  // chainer.functions.normalize default eps

struct FrontendParams {
    int camera_mode;  // NR_CAMERA_LOOK_AT / NR_CAMERA_LOOK
    int perspective;
    int eye_per_batch;
    int idx_per_batch;
    int fill_back;
    int has_directional;
    float target[3];  // `at` (look_at) or `direction` (look)
    float up[3];
    float width;  // tan(viewing angle)
    float ia, id;
    float ca[3], cd[3], ldir[3];
};

__device__ __forceinline__ float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

__device__ __forceinline__ void cross3(const float *a, const float *b, float *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// chainer normalize: v / (|v| + eps)
__device__ __forceinline__ void normalize3(const float *v, float *o)
{
    const float s = sqrtf(dot3(v, v)) + NORM_EPS;
    o[0] = v[0] / s;
    o[1] = v[1] / s;
    o[2] = v[2] / s;
}

// backward of normalize3: g_v = g / (r + eps) - v * (g.v) / ((r + eps)^2 * r)
__device__ __forceinline__ void normalize3_bwd(const float *v, const float *g, float *o)
{
    const float r = sqrtf(dot3(v, v));
    const float s = r + NORM_EPS;
    const float k = r > 0.0f ? dot3(g, v) / (s * s * r) : 0.0f;
    o[0] = g[0] / s - v[0] * k;
    o[1] = g[1] / s - v[1] * k;
    o[2] = g[2] / s - v[2] * k;
}

struct CameraBasis {
    float r[9];  // rows: x axis, y axis, z axis
    float eye[3];
    float d[3], cx[3], cy[3];  // pre-normalisation vectors (needed by the backward)
};

__device__ __forceinline__ void camera_basis(const FrontendParams &P, const float *__restrict__ eye, int b, CameraBasis &C)
{
    const float *e = eye + (P.eye_per_batch ? 3 * b : 0);
    C.eye[0] = e[0];
    C.eye[1] = e[1];
    C.eye[2] = e[2];
    if (P.camera_mode == NR_CAMERA_LOOK_AT) {  // look_at.py:30
        C.d[0] = P.target[0] - C.eye[0];
        C.d[1] = P.target[1] - C.eye[1];
        C.d[2] = P.target[2] - C.eye[2];
    } else {  // look.py:29
        C.d[0] = P.target[0];
        C.d[1] = P.target[1];
        C.d[2] = P.target[2];
    }
    normalize3(C.d, C.r + 6);
    cross3(P.up, C.r + 6, C.cx);  // look_at.py:31
    normalize3(C.cx, C.r + 0);
    cross3(C.r + 6, C.r + 0, C.cy);  // look_at.py:32
    normalize3(C.cy, C.r + 3);
}

// world vertex -> rasterizer input (x, y in NDC, z = camera depth); `cam` receives the camera-space point
__device__ __forceinline__ void project(const FrontendParams &P, const CameraBasis &C, const float *w, float *cam, float *out)
{
    const float t0 = w[0] - C.eye[0], t1 = w[1] - C.eye[1], t2 = w[2] - C.eye[2];  // look_at.py:42-43
#pragma unroll
    for (int i = 0; i < 3; i++) cam[i] = (t0 * C.r[3 * i] + t1 * C.r[3 * i + 1]) + t2 * C.r[3 * i + 2];  // :44
    if (P.perspective) {  // perspective.py:15-17
        out[0] = cam[0] / cam[2] / P.width;
        out[1] = cam[1] / cam[2] / P.width;
    } else {
        out[0] = cam[0];
        out[1] = cam[1];
    }
    out[2] = cam[2];
}

// light colours of a face and of its reversed copy (lighting.py:31-47); n = unnormalised normal, dotn = n_hat . direction
__device__ __forceinline__ void face_light(const FrontendParams &P, const float *w0, const float *w1, const float *w2, float *n,
                                           float &dotn, float *light_f, float *light_b)
{
    float cos_f = 0.0f, cos_b = 0.0f;
    dotn = 0.0f;
    if (P.has_directional) {
        const float v10[3] = {w0[0] - w1[0], w0[1] - w1[1], w0[2] - w1[2]};  // lighting.py:37-38
        const float v12[3] = {w2[0] - w1[0], w2[1] - w1[1], w2[2] - w1[2]};
        cross3(v10, v12, n);
        float nh[3];
        normalize3(n, nh);  // :40
        dotn = dot3(nh, P.ldir);
        cos_f = fmaxf(dotn, 0.0f);   // relu, :45
        cos_b = fmaxf(-dotn, 0.0f);  // the reversed face has exactly the negated normal
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float amb = P.ia != 0.0f ? P.ia * P.ca[c] : 0.0f;                       // :28-29
        light_f[c] = P.has_directional ? amb + P.id * (P.cd[c] * cos_f) : amb;        // :46
        light_b[c] = P.has_directional ? amb + P.id * (P.cd[c] * cos_b) : amb;
    }
}

__global__ __launch_bounds__(FE_THREADS) void k_frontend_forward(const float *__restrict__ vertices,
                                                                 const int32_t *__restrict__ faces_idx,
                                                                 const float *__restrict__ textures,
                                                                 const float *__restrict__ eye, float *__restrict__ faces_out,
                                                                 float *__restrict__ textures_out, int Nv, int Nf, int ts,
                                                                 FrontendParams P, float *__restrict__ light_out)
{
    const int b = blockIdx.y;
    const int f = (blockIdx.x * FE_THREADS + threadIdx.x) / FE_LANES;
    const int lane = threadIdx.x % FE_LANES;
    if (f >= Nf) return;
    const int Fout = P.fill_back ? 2 * Nf : Nf;
    const int32_t *idx = faces_idx + ((size_t)(P.idx_per_batch ? b : 0) * Nf + f) * 3;
    const float *vb = vertices + (size_t)b * Nv * 3;
    float w[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float *src = vb + (size_t)min(max(idx[k], 0), Nv - 1) * 3;  // validated on the host; clamped for memory safety
        w[k][0] = src[0];
        w[k][1] = src[1];
        w[k][2] = src[2];
    }

    if (lane < 2 && (lane == 0 || P.fill_back)) {
        CameraBasis C;
        camera_basis(P, eye, b, C);
        float *dst = faces_out + ((size_t)b * Fout + (lane == 0 ? f : Nf + f)) * 9;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float cam[3], o[3];
            project(P, C, w[k], cam, o);
            float *d = dst + 3 * (lane == 0 ? k : 2 - k);  // reversed vertex order for the back copy
            d[0] = o[0];
            d[1] = o[1];
            d[2] = o[2];
        }
    }

    if (light_out && lane == 0) {  // the colours only: the rasterizer multiplies its samples by them (nr_hip.h: nr_face_light)
        float n[3], dotn, lf[3], lb[3];
        face_light(P, w[0], w[1], w[2], n, dotn, lf, lb);
        float *of = light_out + ((size_t)b * Fout + f) * 3;
        float *ob = light_out + ((size_t)b * Fout + Nf + f) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            of[c] = lf[c];
            if (P.fill_back) ob[c] = lb[c];
        }
    }
    if (textures) {
        float n[3], dotn, lf[3], lb[3];
        face_light(P, w[0], w[1], w[2], n, dotn, lf, lb);
        const int T = ts * ts * ts;
        const float *tex = textures + ((size_t)b * Nf + f) * T * 3;
        float *of = textures_out + ((size_t)b * Fout + f) * T * 3;
        float *ob = textures_out + ((size_t)b * Fout + Nf + f) * T * 3;
        for (int t = lane; t < T; t += FE_LANES) {
            const float t0 = tex[3 * t], t1 = tex[3 * t + 1], t2 = tex[3 * t + 2];
            of[3 * t] = t0 * lf[0];  // lighting.py:50-51
            of[3 * t + 1] = t1 * lf[1];
            of[3 * t + 2] = t2 * lf[2];
            if (P.fill_back) {
                const int u = transpose_texel(t, ts);
                ob[3 * u] = t0 * lb[0];
                ob[3 * u + 1] = t1 * lb[1];
                ob[3 * u + 2] = t2 * lb[2];
            }
        }
    }
}

// --------------------------------------------------------------------------------------------------------------------
// backward
__global__ __launch_bounds__(FE_THREADS) void k_frontend_backward(
    const float *__restrict__ vertices, const int32_t *__restrict__ faces_idx, const float *__restrict__ textures,
    const float *__restrict__ eye, const float *__restrict__ g_faces, const float *__restrict__ g_tex_out,
    float *__restrict__ grad_vertices, float *__restrict__ grad_textures, double *__restrict__ cam_acc, int Nv, int Nf, int ts,
    FrontendParams P, const float *__restrict__ g_light)
{
    __shared__ double s_acc[FE_THREADS / 64][12];
    const int b = blockIdx.y;
    const int f = (blockIdx.x * FE_THREADS + threadIdx.x) / FE_LANES;
    const int lane = threadIdx.x % FE_LANES;
    const bool live = f < Nf;
    const int Fout = P.fill_back ? 2 * Nf : Nf;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; k++) acc[k] = 0.0f;

    if (live) {
        const int32_t *idx = faces_idx + ((size_t)(P.idx_per_batch ? b : 0) * Nf + f) * 3;
        const float *vb = vertices + (size_t)b * Nv * 3;
        const int vi[3] = {min(max(idx[0], 0), Nv - 1), min(max(idx[1], 0), Nv - 1), min(max(idx[2], 0), Nv - 1)};
        float w[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float *src = vb + (size_t)vi[k] * 3;
            w[k][0] = src[0];
            w[k][1] = src[1];
            w[k][2] = src[2];
        }
        float gw[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // gradient w.r.t. the face's three world-space vertices

        // ---- textures / lighting ----
        if ((textures && g_tex_out) || g_light) {
            float n[3], dotn, lf[3], lb[3];
            face_light(P, w[0], w[1], w[2], n, dotn, lf, lb);
            const int T = g_light ? 0 : ts * ts * ts;
            const float *tex = textures + ((size_t)b * Nf + f) * T * 3;
            const float *gf = g_tex_out + ((size_t)b * Fout + f) * T * 3;
            const float *gb = g_tex_out + ((size_t)b * Fout + Nf + f) * T * 3;
            float *gt = grad_textures ? grad_textures + ((size_t)b * Nf + f) * T * 3 : nullptr;
            float glf[3] = {0, 0, 0}, glb[3] = {0, 0, 0};
            if (g_light && lane == 0) {  // the colours' gradient arrives summed (the other lanes add zeros below)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    glf[c] = g_light[((size_t)b * Fout + f) * 3 + c];
                    if (P.fill_back) glb[c] = g_light[((size_t)b * Fout + Nf + f) * 3 + c];
                }
            }
            for (int t = lane; t < T; t += FE_LANES) {
                const int u = P.fill_back ? transpose_texel(t, ts) : 0;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float x = tex[3 * t + c];
                    const float a = gf[3 * t + c];
                    const float bb = P.fill_back ? gb[3 * u + c] : 0.0f;
                    if (gt) gt[3 * t + c] = P.fill_back ? a * lf[c] + bb * lb[c] : a * lf[c];
                    glf[c] += a * x;
                    glb[c] += bb * x;
                }
            }
            if (P.has_directional && grad_vertices) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
#pragma unroll
                    for (int m = 1; m < FE_LANES; m <<= 1) {
                        glf[c] += __shfl_xor(glf[c], m);
                        glb[c] += __shfl_xor(glb[c], m);
                    }
                }
                if (lane == 0) {
                    // light = amb + id * (cd * cos): d loss / d cos
                    const float gcf = P.id * (P.cd[0] * glf[0] + P.cd[1] * glf[1] + P.cd[2] * glf[2]);
                    const float gcb = P.id * (P.cd[0] * glb[0] + P.cd[1] * glb[1] + P.cd[2] * glb[2]);
                    float gdot = 0.0f;  // relu: the front copy sees dotn, the back copy -dotn
                    if (dotn > 0.0f) gdot += gcf;
                    if (-dotn > 0.0f) gdot -= gcb;
                    if (gdot != 0.0f) {
                        const float gnh[3] = {gdot * P.ldir[0], gdot * P.ldir[1], gdot * P.ldir[2]};
                        float gn[3];
                        normalize3_bwd(n, gnh, gn);
                        // n = v10 x v12:  g_v10 = v12 x g_n,  g_v12 = g_n x v10
                        const float v10[3] = {w[0][0] - w[1][0], w[0][1] - w[1][1], w[0][2] - w[1][2]};
                        const float v12[3] = {w[2][0] - w[1][0], w[2][1] - w[1][1], w[2][2] - w[1][2]};
                        float ga[3], gb2[3];
                        cross3(v12, gn, ga);
                        cross3(gn, v10, gb2);
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            gw[0][c] += ga[c];
                            gw[2][c] += gb2[c];
                            gw[1][c] -= ga[c] + gb2[c];
                        }
                    }
                }
            }
        }

        // ---- geometry: perspective, rotation, gather ----
        if (lane == 0 && grad_vertices) {
            CameraBasis C;
            camera_basis(P, eye, b, C);
            const float *g0 = g_faces + ((size_t)b * Fout + f) * 9;
            const float *g1 = g_faces + ((size_t)b * Fout + Nf + f) * 9;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float g[3] = {g0[3 * k], g0[3 * k + 1], g0[3 * k + 2]};
                if (P.fill_back) {
                    g[0] += g1[3 * (2 - k)];
                    g[1] += g1[3 * (2 - k) + 1];
                    g[2] += g1[3 * (2 - k) + 2];
                }
                float cam[3], o[3];
                project(P, C, w[k], cam, o);
                float gc[3];  // gradient w.r.t. the camera-space point
                if (P.perspective) {
                    const float zw = cam[2] * P.width;
                    gc[0] = g[0] / zw;
                    gc[1] = g[1] / zw;
                    gc[2] = g[2] - (g[0] * cam[0] + g[1] * cam[1]) / (cam[2] * zw);
                } else {
                    gc[0] = g[0];
                    gc[1] = g[1];
                    gc[2] = g[2];
                }
                // cam = R (w - eye):  g_w = R^T g_cam,  g_eye -= g_w,  g_R[i][j] += g_cam[i] * (w - eye)[j]
                float gwk[3];
#pragma unroll
                for (int j = 0; j < 3; j++) gwk[j] = (gc[0] * C.r[j] + gc[1] * C.r[3 + j]) + gc[2] * C.r[6 + j];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    gw[k][j] += gwk[j];
                    acc[j] += gwk[j];
                }
                if (cam_acc) {
                    const float t[3] = {w[k][0] - C.eye[0], w[k][1] - C.eye[1], w[k][2] - C.eye[2]};
#pragma unroll
                    for (int i = 0; i < 3; i++)
#pragma unroll
                        for (int j = 0; j < 3; j++) acc[3 + 3 * i + j] += gc[i] * t[j];
                }
            }
        }

        if (lane == 0 && grad_vertices) {  // face -> vertex scatter (get_item backward)
            float *gv = grad_vertices + (size_t)b * Nv * 3;
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int c = 0; c < 3; c++)
                    if (gw[k][c] != 0.0f) atomicAdd(gv + (size_t)vi[k] * 3 + c, gw[k][c]);
        }
    }

    if (cam_acc) {  // per-image sums for the camera backward: [0..2] = sum of g_w, [3..11] = g_R
        const int wave = threadIdx.x / 64, wl = threadIdx.x % 64;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            double v = (double)acc[k];
#pragma unroll
            for (int m = FE_LANES; m < 64; m <<= 1) v += __shfl_xor(v, m);
            if (wl == 0) s_acc[wave][k] = v;
        }
        __syncthreads();
        if (threadIdx.x < 12) {
            double v = 0.0;
#pragma unroll
            for (int q = 0; q < FE_THREADS / 64; q++) v += s_acc[q][threadIdx.x];
            if (v != 0.0) atomicAdd(cam_acc + (size_t)b * 12 + threadIdx.x, v);
        }
    }
}

// one thread per image: the gradient of the camera position through `cam = R(eye) (w - eye)`
__global__ void k_camera_backward(const float *__restrict__ eye, const double *__restrict__ cam_acc,
                                  float *__restrict__ grad_eye, int B, FrontendParams P)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    CameraBasis C;
    camera_basis(P, eye, b, C);
    const double *a = cam_acc + (size_t)b * 12;
    float ge[3] = {-(float)a[0], -(float)a[1], -(float)a[2]};  // direct term: cam = R (w - eye)
    if (P.camera_mode == NR_CAMERA_LOOK_AT) {                  // R depends on eye only in look_at mode
        const float gx[3] = {(float)a[3], (float)a[4], (float)a[5]};
        const float gy[3] = {(float)a[6], (float)a[7], (float)a[8]};
        float gz[3] = {(float)a[9], (float)a[10], (float)a[11]};
        // y = N(cy), cy = z x x
        float gcy[3], t[3], gxt[3];
        normalize3_bwd(C.cy, gy, gcy);
        cross3(C.r + 0, gcy, t);  // g_z += x x g_cy
        gz[0] += t[0];
        gz[1] += t[1];
        gz[2] += t[2];
        cross3(gcy, C.r + 6, t);  // g_x += g_cy x z
        gxt[0] = gx[0] + t[0];
        gxt[1] = gx[1] + t[1];
        gxt[2] = gx[2] + t[2];
        // x = N(cx), cx = up x z
        float gcx[3];
        normalize3_bwd(C.cx, gxt, gcx);
        cross3(gcx, P.up, t);  // g_z += g_cx x up
        gz[0] += t[0];
        gz[1] += t[1];
        gz[2] += t[2];
        // z = N(d), d = at - eye
        float gd[3];
        normalize3_bwd(C.d, gz, gd);
        ge[0] -= gd[0];
        ge[1] -= gd[1];
        ge[2] -= gd[2];
    }
    if (P.eye_per_batch) {
        grad_eye[3 * b] = ge[0];
        grad_eye[3 * b + 1] = ge[1];
        grad_eye[3 * b + 2] = ge[2];
    } else {  // one camera shared by the batch: grad_eye [3] zero-filled by the host call
        atomicAdd(grad_eye + 0, ge[0]);
        atomicAdd(grad_eye + 1, ge[1]);
        atomicAdd(grad_eye + 2, ge[2]);
    }
}

int fill_params(FrontendParams &P, const nr_camera *cam, const nr_light *light, int idx_per_batch, int eye_per_batch,
                int fill_back, bool with_textures)
{
    if (!cam) return NR_E_NULL;
    if (cam->mode != NR_CAMERA_LOOK_AT && cam->mode != NR_CAMERA_LOOK) return NR_E_MODE;
    if (with_textures && !light) return NR_E_NULL;
    P.camera_mode = cam->mode;
    P.perspective = cam->perspective != 0;
    P.eye_per_batch = eye_per_batch != 0;
    P.idx_per_batch = idx_per_batch != 0;
    P.fill_back = fill_back != 0;
    for (int k = 0; k < 3; k++) {
        P.target[k] = cam->target[k];
        P.up[k] = cam->up[k];
    }
    P.width = cam->width;
    P.ia = P.id = 0.0f;
    P.has_directional = 0;
    for (int k = 0; k < 3; k++) P.ca[k] = P.cd[k] = P.ldir[k] = 0.0f;
    if (light) {
        P.ia = light->intensity_ambient;
        P.id = light->intensity_directional;
        P.has_directional = light->intensity_directional != 0.0f;
        for (int k = 0; k < 3; k++) {
            P.ca[k] = light->color_ambient[k];
            P.cd[k] = light->color_directional[k];
            P.ldir[k] = light->direction[k];
        }
    }
    return 0;
}

inline int frontend_sizes(int B, int Nv, int Nf, int ts, bool with_textures)
{
    if (B < 1 || Nv < 1 || Nf < 1 || B > 65535) return NR_E_SIZE;
    if (with_textures && ts < 1) return NR_E_SIZE;
    if ((size_t)B * (size_t)Nf > 0x7fffffffull / 18) return NR_E_SIZE;
    return 0;
}

}  // namespace

NR_API size_t nr_frontend_workspace_bytes(int32_t B) { return B < 1 ? 0 : (size_t)B * 12 * sizeof(double); }

NR_API int nr_frontend_forward(const float *vertices, const int32_t *faces_idx, const float *textures, const float *eye,
                               float *faces_out, float *textures_out, int32_t B, int32_t Nv, int32_t Nf, int32_t ts,
                               int32_t idx_per_batch, int32_t eye_per_batch, int32_t fill_back, const nr_camera *camera,
                               const nr_light *light, void *stream)
{
    if (!vertices || !faces_idx || !eye || !faces_out) return NR_E_NULL;
    if ((textures == nullptr) != (textures_out == nullptr)) return NR_E_MODE;
    int rc = frontend_sizes(B, Nv, Nf, ts, textures != nullptr);
    if (rc) return rc;
    FrontendParams P;
    rc = fill_params(P, camera, light, idx_per_batch, eye_per_batch, fill_back, textures != nullptr);
    if (rc) return rc;
    const dim3 grid((unsigned)(((size_t)Nf * FE_LANES + FE_THREADS - 1) / FE_THREADS), (unsigned)B);
    hipLaunchKernelGGL(k_frontend_forward, grid, dim3(FE_THREADS), 0, (hipStream_t)stream, vertices, faces_idx, textures, eye,
                       faces_out, textures_out, Nv, Nf, ts, P, (float *)nullptr);
    return launch_status();
}

NR_API int nr_frontend_forward_light(const float *vertices, const int32_t *faces_idx, const float *eye, float *faces_out,
                                     float *light_out, int32_t B, int32_t Nv, int32_t Nf, int32_t idx_per_batch,
                                     int32_t eye_per_batch, int32_t fill_back, const nr_camera *camera,
                                     const nr_light *light, void *stream)
{
    if (!vertices || !faces_idx || !eye || !faces_out || !light_out) return NR_E_NULL;
    int rc = frontend_sizes(B, Nv, Nf, 0, false);
    if (rc) return rc;
    FrontendParams P;
    rc = fill_params(P, camera, light, idx_per_batch, eye_per_batch, fill_back, true);
    if (rc) return rc;
    const dim3 grid((unsigned)(((size_t)Nf * FE_LANES + FE_THREADS - 1) / FE_THREADS), (unsigned)B);
    hipLaunchKernelGGL(k_frontend_forward, grid, dim3(FE_THREADS), 0, (hipStream_t)stream, vertices, faces_idx,
                       (const float *)nullptr, eye, faces_out, (float *)nullptr, Nv, Nf, 0, P, light_out);
    return launch_status();
}

namespace {
int frontend_backward(const float *vertices, const int32_t *faces_idx, const float *textures, const float *eye,
                      const float *grad_faces, const float *grad_textures_out, const float *grad_light,
                      float *grad_vertices, float *grad_textures, float *grad_eye, int32_t B, int32_t Nv, int32_t Nf,
                      int32_t ts, int32_t idx_per_batch, int32_t eye_per_batch, int32_t fill_back,
                      const nr_camera *camera, const nr_light *light, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!vertices || !faces_idx || !eye || !grad_faces) return NR_E_NULL;
    if (!grad_vertices && !grad_textures && !grad_eye) return NR_E_MODE;
    if (grad_eye && !grad_vertices) return NR_E_MODE;  // the camera sums are produced by the vertex pass
    if (grad_textures && !(textures && grad_textures_out)) return NR_E_MODE;
    if (grad_textures_out && !textures) return NR_E_MODE;
    const bool lit = textures != nullptr || grad_light != nullptr;
    int rc = frontend_sizes(B, Nv, Nf, ts, textures != nullptr);
    if (rc) return rc;
    FrontendParams P;
    rc = fill_params(P, camera, light, idx_per_batch, eye_per_batch, fill_back, lit);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    double *cam_acc = nullptr;
    if (grad_eye) {
        if (!workspace || workspace_bytes < nr_frontend_workspace_bytes(B)) return NR_E_WORKSPACE;
        cam_acc = (double *)workspace;
        int e = fill_bytes(cam_acc, 0, nr_frontend_workspace_bytes(B), st);
        if (e != 0) return e;
        if (!eye_per_batch) {
            e = fill_bytes(grad_eye, 0, 3 * sizeof(float), st);
            if (e != 0) return e;
        }
    }
    if (grad_vertices) {
        const int e = fill_bytes(grad_vertices, 0, (size_t)B * Nv * 3 * sizeof(float), st);
        if (e != 0) return e;
    }
    const dim3 grid((unsigned)(((size_t)Nf * FE_LANES + FE_THREADS - 1) / FE_THREADS), (unsigned)B);
    hipLaunchKernelGGL(k_frontend_backward, grid, dim3(FE_THREADS), 0, st, vertices, faces_idx, textures, eye, grad_faces,
                       grad_textures_out, grad_vertices, grad_textures, cam_acc, Nv, Nf, ts, P, grad_light);
    rc = launch_status();
    if (rc) return rc;
    if (grad_eye) {
        hipLaunchKernelGGL(k_camera_backward, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, eye, cam_acc, grad_eye, B, P);
        rc = launch_status();
    }
    return rc;
}
}  // namespace

NR_API int nr_frontend_backward(const float *vertices, const int32_t *faces_idx, const float *textures, const float *eye,
                                const float *grad_faces, const float *grad_textures_out, float *grad_vertices,
                                float *grad_textures, float *grad_eye, int32_t B, int32_t Nv, int32_t Nf, int32_t ts,
                                int32_t idx_per_batch, int32_t eye_per_batch, int32_t fill_back, const nr_camera *camera,
                                const nr_light *light, void *workspace, size_t workspace_bytes, void *stream)
{
    return frontend_backward(vertices, faces_idx, textures, eye, grad_faces, grad_textures_out, nullptr, grad_vertices,
                             grad_textures, grad_eye, B, Nv, Nf, ts, idx_per_batch, eye_per_batch, fill_back, camera, light,
                             workspace, workspace_bytes, stream);
}

NR_API int nr_frontend_backward_light(const float *vertices, const int32_t *faces_idx, const float *eye,
                                      const float *grad_faces, const float *grad_light, float *grad_vertices,
                                      float *grad_eye, int32_t B, int32_t Nv, int32_t Nf, int32_t idx_per_batch,
                                      int32_t eye_per_batch, int32_t fill_back, const nr_camera *camera,
                                      const nr_light *light, void *workspace, size_t workspace_bytes, void *stream)
{
    if (grad_light && !light) return NR_E_NULL;
    return frontend_backward(vertices, faces_idx, nullptr, eye, grad_faces, nullptr, grad_light, grad_vertices, nullptr,
                             grad_eye, B, Nv, Nf, 0, idx_per_batch, eye_per_batch, fill_back, camera, light, workspace,
                             workspace_bytes, stream);
}
