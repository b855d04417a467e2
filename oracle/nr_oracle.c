/*
 * nr_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A literal, single-threaded, float32 restatement of the hot path of
 * hiroharu-kato/neural_renderer 1.1.3, file neural_renderer/rasterize.py (safe path).
 * Every function cites the reference lines it follows.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product (neural_renderer_amd)
 * never does.
 *
 * Numeric conventions:
 *   - the reference kernels are CUDA C++ strings; `float op int` is a float operation, a double
 *     literal (`0.5`, `2.`, `1.`, the pasted near/eps) promotes the expression to double, and the
 *     result is rounded once when stored into a float.  This file reproduces those promotions.
 *   - NO floating-point contraction (build with -ffp-contract=off): each `a*b+c` of the reference
 *     source is a rounded multiply followed by a rounded add.  nvcc would be allowed to fuse some of
 *     them; the reference cannot be executed here, so the un-fused reading of the source is the
 *     convention shared by this oracle and the HIP kernels.
 *   - CUDA min/max ignore NaN (fmin/fmax semantics); float->int conversion truncates, saturates
 *     and maps NaN to 0 (cvt.rzi.s32): see f2i().
 *   - atomicAdd accumulations (K7, K8) are performed in pixel-index order; any order is a valid
 *     serialisation of the reference.
 *
 * Parity pin: oracle/../tests/test_oracle_golden.py checks this file against every fixture the
 * reference's test-suite ships for the path (teapot_blender.png, test_depth.png,
 * test_rasterize{1,2}.png, the grad_ref constants).  backward_textures (K7) and
 * backward_depth_map (K8) have no effective reference test: "parity unpinned" for
 * those two, they are pinned only by this literal restatement plus finite differences
 * (oracle side) and by the analytic / finite-difference tests of tests/test_gradient_pins_gpu.py (GPU side).
 *
 * Threads: built with -fopenmp the loops run in parallel over units that do not interact -- pixels for K2 / K4 / K5,
 * faces for K1 / K6, batch elements for K7 / K8 (whose per-face accumulation order therefore stays the pixel order) --
 * so the results do not depend on the thread count (tests/test_oracle_threads.py).  oracle_set_threads(1) gives the
 * single-thread port that bench.py times as `cpu_baseline` (cores = 1).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define API __attribute__((visibility("default")))

static int g_threads = 0; /* 0 = OpenMP default (all cores) */
API void oracle_set_threads(int n) { g_threads = n > 0 ? n : 0; }
API int oracle_get_threads(void)
{
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}
#ifdef _OPENMP
#define NTHREADS (g_threads > 0 ? g_threads : omp_get_max_threads())
#else
#define NTHREADS 1
#endif

/* CUDA `(int)x` for float/double x: round toward zero, saturate, NaN -> 0. */
static inline int f2i(double x)
{
    if (x != x) return 0;
    if (x >= 2147483647.0) return 2147483647;
    if (x <= -2147483648.0) return (-2147483647 - 1);
    return (int)x;
}

/* Contraction study (tests/test_contraction.py).  The convention of this oracle and of the HIP kernels is the
 * UN-fused reading of the reference source.  nvcc (-fmad=true, its default) is allowed to contract `a*b+c` into one fused
 * multiply-add; the reference cannot be run here, so oracle_set_contraction(1) switches K1 and K2 to the contraction an
 * LLVM-style compiler performs on the same expressions -- fadd(fmul(a,b), x) -> fma(a,b,x) with the FIRST product of a sum
 * fused and the other operand evaluated (rounded) beforehand; products that are only compared (:252, :306, :310-312) have
 * no add to fuse with -- to MEASURE how far the outputs can move: which pixels change owner, max |d depth|, max |d weight|.
 * It is a measuring device for the error bar on "bit-exact", never the parity target. */
static int g_fma = 0;
API void oracle_set_contraction(int on) { g_fma = on ? 1 : 0; }
API int oracle_get_contraction(void) { return g_fma; }
/* a*b + c, a*b - c*d and (a*x + b*y) + c as the source has them / as a contracting compiler evaluates them */
static inline float muladd(float a, float b, float c) { return g_fma ? fmaf(a, b, c) : a * b + c; }
static inline float mulsub2(float a, float b, float c, float d) { return g_fma ? fmaf(a, b, -(c * d)) : a * b - c * d; }
static inline float dot2c(float a, float x, float b, float y, float c) { return g_fma ? fmaf(a, x, b * y) + c : a * x + b * y + c; }

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* back-face test shared by K1/K2/K6: rasterize.py:252, :306, :540 */
static inline int is_backside(const float *face)
{
    return (face[7] - face[1]) * (face[3] - face[0]) < (face[4] - face[1]) * (face[6] - face[0]);
}

/* ------------------------------------------------------------------------------------------------
 * K1: rasterize.py:240-277 -- per-face inverse of the barycentric matrix (zeros for back faces).
 * faces [B*F*9], faces_inv [B*F*9] (output, fully written).
 */
API void oracle_forward_face_inv(const float *faces, float *faces_inv, int batch_size, int num_faces,
                                 int image_size)
{
    const int is = image_size;
    const long n = (long)batch_size * num_faces;
    memset(faces_inv, 0, sizeof(float) * 9 * (size_t)n); /* :240 xp.zeros_like */
#pragma omp parallel for schedule(static) num_threads(NTHREADS)
    for (long i = 0; i < n; i++) {
        const float *face = faces + i * 9;
        float *face_inv_g = faces_inv + i * 9;
        if (is_backside(face)) continue; /* :252 */

        float p[3][2];
        for (int num = 0; num < 3; num++)
            for (int dim = 0; dim < 2; dim++)
                p[num][dim] = (float)(0.5 * (double)(muladd(face[3 * num + dim], (float)is, (float)is) - 1.0f)); /* :258 */

        float face_inv[9] = {/* :261-264 */
                             p[1][1] - p[2][1], p[2][0] - p[1][0], mulsub2(p[1][0], p[2][1], p[2][0], p[1][1]),
                             p[2][1] - p[0][1], p[0][0] - p[2][0], mulsub2(p[2][0], p[0][1], p[0][0], p[2][1]),
                             p[0][1] - p[1][1], p[1][0] - p[0][0], mulsub2(p[0][0], p[1][1], p[1][0], p[0][1])};
        float face_inv_denominator = (/* :265-268: (m1 + m2) + m3 */
                                      muladd(p[1][0], p[2][1] - p[0][1],
                                             muladd(p[2][0], p[0][1] - p[1][1], p[0][0] * (p[1][1] - p[2][1]))));
        if (!g_fma)
            face_inv_denominator = p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) +
                                   p[1][0] * (p[2][1] - p[0][1]);
        for (int k = 0; k < 9; k++) face_inv[k] /= face_inv_denominator; /* :269 */
        for (int k = 0; k < 9; k++) face_inv_g[k] = face_inv[k];         /* :272 */
    }
}

/* ------------------------------------------------------------------------------------------------
 * K2: rasterize.py:279-359 -- per-pixel brute-force visibility.
 * Outputs must be pre-initialised by the caller exactly like forward_gpu does (rasterize.py:478-496):
 * face_index_map = -1, weight_map = 0, depth_map = far, face_inv_map = 0.
 * face_inv_map may be NULL when return_depth == 0.
 */
API void oracle_forward_face_index_map(const float *faces, const float *faces_inv, int32_t *face_index_map,
                                       float *weight_map, float *depth_map, float *face_inv_map,
                                       int batch_size, int num_faces, int image_size, double near,
                                       double far, int return_depth)
{
    const int is = image_size;
    const int nf = num_faces;
    const long n = (long)batch_size * is * is;
#pragma omp parallel for schedule(dynamic, 64) num_threads(NTHREADS)
    for (long i = 0; i < n; i++) {
        const int bn = (int)(i / ((long)is * is));
        const int pn = (int)(i % ((long)is * is));
        const int yi = pn / is;
        const int xi = pn % is;
        const float yp = (float)((2. * yi + 1 - is) / is); /* :291 */
        const float xp = (float)((2. * xi + 1 - is) / is); /* :292 */

        const float *face = faces + (long)bn * nf * 9 - 9;
        const float *face_inv = faces_inv + (long)bn * nf * 9 - 9;
        float depth_min = (float)far; /* :296 */
        int face_index_min = -1;
        float weight_min[3] = {0, 0, 0};
        float face_inv_min[9] = {0};
        for (int fn = 0; fn < nf; fn++) {
            face += 9;
            face_inv += 9;
            if (is_backside(face)) continue; /* :306 */

            /* :310-312 */
            if (((yp - face[1]) * (face[3] - face[0]) < (xp - face[0]) * (face[4] - face[1])) ||
                ((yp - face[4]) * (face[6] - face[3]) < (xp - face[3]) * (face[7] - face[4])) ||
                ((yp - face[7]) * (face[0] - face[6]) < (xp - face[6]) * (face[1] - face[7])))
                continue;

            /* :317-319 */
            float w[3];
            w[0] = dot2c(face_inv[3 * 0 + 0], (float)xi, face_inv[3 * 0 + 1], (float)yi, face_inv[3 * 0 + 2]);
            w[1] = dot2c(face_inv[3 * 1 + 0], (float)xi, face_inv[3 * 1 + 1], (float)yi, face_inv[3 * 1 + 2]);
            w[2] = dot2c(face_inv[3 * 2 + 0], (float)xi, face_inv[3 * 2 + 1], (float)yi, face_inv[3 * 2 + 2]);

            /* :322-327 */
            float w_sum = 0;
            for (int k = 0; k < 3; k++) {
                w[k] = (float)fmin(fmax((double)w[k], 0.), 1.);
                w_sum += w[k];
            }
            for (int k = 0; k < 3; k++) w[k] /= w_sum;

            /* :330-331 */
            const float zp = (float)(1. / (double)(w[0] / face[2] + w[1] / face[5] + w[2] / face[8]));
            if ((double)zp <= near || far <= (double)zp) continue;

            /* :334-339 */
            if (zp < depth_min) {
                depth_min = zp;
                face_index_min = fn;
                for (int k = 0; k < 3; k++) weight_min[k] = w[k];
                if (return_depth)
                    for (int k = 0; k < 9; k++) face_inv_min[k] = face_inv[k];
            }
        }

        /* :343-348 */
        if (0 <= face_index_min) {
            depth_map[i] = depth_min;
            face_index_map[i] = face_index_min;
            for (int k = 0; k < 3; k++) weight_map[3 * i + k] = weight_min[k];
            if (return_depth)
                for (int k = 0; k < 9; k++) face_inv_map[9 * i + k] = face_inv_min[k];
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * K3: rasterize.py:102-236 -- the "unsafe" visibility kernel (USE_UNSAFE_IMPLEMENTATION, use_unsafe_rasterizer()): one
 * thread per face scan-converts its triangle column by column and updates the maps under a per-pixel spin lock.
 * This is the race-free SEQUENTIAL emulation: faces of an image in ascending order, each pixel update atomic.  Any
 * serialisation of the lock is a valid execution of the reference; this one resolves depth ties to the lowest face
 * index (the strict `<` of :206), like the safe path (on the GPU the lock order decides).
 * What differs from K1+K2 by construction: vertices are permuted by x (pi[], :123-131) before face_inv is formed
 * (:147-155) and weights / face_inv are written back through the permutation (:210, :213-214); coverage comes from the
 * column / row spans of edge interpolation (:158-181) instead of the three edge functions (:310-312); a face whose
 * leftmost and rightmost vertex share x is dropped (:144).  No test of the reference covers it ("parity unpinned" in the
 * reference; pinned here against the safe path by tests/test_oracle_golden.py with SURVEY Appendix B's bounds).
 * Outputs pre-initialised by the caller like forward_gpu does (:478-496); face_inv_map may be NULL when !return_depth.
 */
API void oracle_forward_face_index_map_unsafe(const float *faces, int32_t *face_index_map, float *weight_map,
                                              float *depth_map, float *face_inv_map, int batch_size, int num_faces,
                                              int image_size, double near, double far, int return_depth)
{
    const int is = image_size;
#pragma omp parallel for schedule(dynamic, 1) num_threads(NTHREADS)
    for (int bn = 0; bn < batch_size; bn++) {
        for (int fn = 0; fn < num_faces; fn++) {
            const float *face = faces + ((long)bn * num_faces + fn) * 9; /* :117 */
            if (is_backside(face)) continue;                              /* :120 */

            /* :123-131 pi[0], pi[1], pi[2] = leftmost, middle, rightmost points */
            int pi[3];
            if (face[0] < face[3]) {
                if (face[6] < face[0]) pi[0] = 2; else pi[0] = 0;
                if (face[3] < face[6]) pi[2] = 2; else pi[2] = 1;
            } else {
                if (face[6] < face[3]) pi[0] = 2; else pi[0] = 1;
                if (face[0] < face[6]) pi[2] = 2; else pi[2] = 0;
            }
            pi[1] = 0; /* (:131 leaves it unset when pi[0] == pi[2], which the branches above never produce) */
            for (int k = 0; k < 3; k++)
                if (pi[0] != k && pi[2] != k) pi[1] = k;

            /* :134-143 */
            float p[3][3];
            for (int num = 0; num < 3; num++)
                for (int dim = 0; dim < 3; dim++) {
                    if (dim != 2)
                        p[num][dim] = (float)(0.5 * (double)(face[3 * pi[num] + dim] * (float)is + (float)is - 1.0f));
                    else
                        p[num][dim] = face[3 * pi[num] + dim];
                }
            if (p[0][0] == p[2][0]) continue; /* :144 line, not triangle */

            /* :147-155 */
            float face_inv[9] = {p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
                                 p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
                                 p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};
            const float face_inv_denominator =
                (p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) + p[1][0] * (p[2][1] - p[0][1]));
            for (int k = 0; k < 9; k++) face_inv[k] /= face_inv_denominator;

            /* :158-160 from left to right */
            const int xi_min = f2i(fmax((double)ceilf(p[0][0]), 0.));
            const int xi_max = f2i(fmin((double)p[2][0], is - 1.));
            for (int xi = xi_min; xi <= xi_max; xi++) {
                /* :162-176 */
                float yi1, yi2;
                if ((float)xi <= p[1][0]) {
                    if (p[1][0] - p[0][0] != 0)
                        yi1 = (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]) * ((float)xi - p[0][0]) + p[0][1];
                    else
                        yi1 = p[1][1];
                } else {
                    if (p[2][0] - p[1][0] != 0)
                        yi1 = (p[2][1] - p[1][1]) / (p[2][0] - p[1][0]) * ((float)xi - p[1][0]) + p[1][1];
                    else
                        yi1 = p[1][1];
                }
                yi2 = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]) * ((float)xi - p[0][0]) + p[0][1];

                /* :179-181 from up to bottom */
                const int yi_min = f2i(fmax(0., (double)ceilf(fminf(yi1, yi2))));
                const int yi_max = f2i(fmin((double)fmaxf(yi1, yi2), is - 1.));
                for (int yi = yi_min; yi <= yi_max; yi++) {
                    const long index = (long)bn * is * is + (long)yi * is + xi; /* :183 */

                    /* :186-188 */
                    float w[3];
                    for (int k = 0; k < 3; k++)
                        w[k] = face_inv[3 * k + 0] * (float)xi + face_inv[3 * k + 1] * (float)yi + face_inv[3 * k + 2];

                    /* :191-196 */
                    float w_sum = 0;
                    for (int k = 0; k < 3; k++) {
                        w[k] = (float)fmin(fmax((double)w[k], 0.), 1.);
                        w_sum += w[k];
                    }
                    for (int k = 0; k < 3; k++) w[k] /= w_sum;

                    /* :199-200 */
                    const float zp = (float)(1. / (double)(w[0] / p[0][2] + w[1] / p[1][2] + w[2] / p[2][2]));
                    if ((double)zp <= near || far <= (double)zp) continue;

                    /* :203-223, the lock taken at once */
                    if (zp < depth_map[index]) {
                        depth_map[index] = zp;
                        face_index_map[index] = fn;
                        for (int k = 0; k < 3; k++) weight_map[3 * index + pi[k]] = w[k]; /* :210 */
                        if (return_depth)
                            for (int k = 0; k < 3; k++)
                                for (int l = 0; l < 3; l++)
                                    face_inv_map[9 * index + 3 * pi[l] + k] = face_inv[3 * l + k]; /* :213-214 */
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * K2, cache-blocked evaluation order (same arguments, same results bit for bit as oracle_forward_face_index_map;
 * tests/test_oracle_threads.py asserts that).  The literal loop streams all faces once per pixel, which for a
 * 655 360-face mesh is 23.6 MB per pixel; here a thread takes PB consecutive pixels of a row and scans the faces once
 * for the block, keeping the running minimum of each pixel (:296-339) in small arrays.  Every pixel still meets the
 * faces in ascending order with the strict `<` of :334, and every (pixel, face) evaluation is the expression of
 * :306-331, so nothing about the result changes -- only the order in which independent pixels are visited.
 * Used for the full-size BASELINE configurations (tests/test_full_size_gpu.py) and the all-cores cpu_baseline.
 */
#define K2_PB 64
API void oracle_forward_face_index_map_blocked(const float *faces, const float *faces_inv, int32_t *face_index_map,
                                               float *weight_map, float *depth_map, float *face_inv_map,
                                               int batch_size, int num_faces, int image_size, double near,
                                               double far, int return_depth)
{
    const int is = image_size;
    const int nf = num_faces;
    const int blocks_per_row = (is + K2_PB - 1) / K2_PB;
    const long n_blocks = (long)batch_size * is * blocks_per_row;
#pragma omp parallel for schedule(dynamic, 4) num_threads(NTHREADS)
    for (long blk = 0; blk < n_blocks; blk++) {
        const int bn = (int)(blk / ((long)is * blocks_per_row));
        const int rem = (int)(blk % ((long)is * blocks_per_row));
        const int yi = rem / blocks_per_row;
        const int x0 = (rem % blocks_per_row) * K2_PB;
        const int nx = imin(K2_PB, is - x0);
        const float yp = (float)((2. * yi + 1 - is) / is); /* :291 */
        float xp[K2_PB], depth_min[K2_PB], weight_min[K2_PB][3];
        int face_index_min[K2_PB];
        for (int j = 0; j < nx; j++) {
            xp[j] = (float)((2. * (x0 + j) + 1 - is) / is); /* :292 */
            depth_min[j] = (float)far;                      /* :296 */
            face_index_min[j] = -1;
            weight_min[j][0] = weight_min[j][1] = weight_min[j][2] = 0;
        }
        const float *face = faces + (long)bn * nf * 9 - 9;
        const float *face_inv = faces_inv + (long)bn * nf * 9 - 9;
        for (int fn = 0; fn < nf; fn++) {
            face += 9;
            face_inv += 9;
            if (is_backside(face)) continue; /* :306 */
            /* the pixel-independent factors of :310-312 */
            const float e0x = face[3] - face[0], e0y = face[4] - face[1];
            const float e1x = face[6] - face[3], e1y = face[7] - face[4];
            const float e2x = face[0] - face[6], e2y = face[1] - face[7];
            const float a0 = (yp - face[1]) * e0x, a1 = (yp - face[4]) * e1x, a2 = (yp - face[7]) * e2x;
            unsigned char hit[K2_PB];
            int any = 0;
            for (int j = 0; j < nx; j++) { /* :310-312 */
                const int out = (a0 < (xp[j] - face[0]) * e0y) | (a1 < (xp[j] - face[3]) * e1y) |
                                (a2 < (xp[j] - face[6]) * e2y);
                hit[j] = (unsigned char)!out;
                any |= !out;
            }
            if (!any) continue;
            for (int j = 0; j < nx; j++) {
                if (!hit[j]) continue;
                const int xi = x0 + j;
                float w[3]; /* :317-319 */
                w[0] = dot2c(face_inv[3 * 0 + 0], (float)xi, face_inv[3 * 0 + 1], (float)yi, face_inv[3 * 0 + 2]);
                w[1] = dot2c(face_inv[3 * 1 + 0], (float)xi, face_inv[3 * 1 + 1], (float)yi, face_inv[3 * 1 + 2]);
                w[2] = dot2c(face_inv[3 * 2 + 0], (float)xi, face_inv[3 * 2 + 1], (float)yi, face_inv[3 * 2 + 2]);
                float w_sum = 0; /* :322-327 */
                for (int k = 0; k < 3; k++) {
                    w[k] = (float)fmin(fmax((double)w[k], 0.), 1.);
                    w_sum += w[k];
                }
                for (int k = 0; k < 3; k++) w[k] /= w_sum;
                const float zp = (float)(1. / (double)(w[0] / face[2] + w[1] / face[5] + w[2] / face[8])); /* :330 */
                if ((double)zp <= near || far <= (double)zp) continue;                                        /* :331 */
                if (zp < depth_min[j]) { /* :334-339 */
                    depth_min[j] = zp;
                    face_index_min[j] = fn;
                    for (int k = 0; k < 3; k++) weight_min[j][k] = w[k];
                }
            }
        }
        for (int j = 0; j < nx; j++) { /* :343-348 */
            if (0 <= face_index_min[j]) {
                const long i = ((long)bn * is + yi) * is + x0 + j;
                depth_map[i] = depth_min[j];
                face_index_map[i] = face_index_min[j];
                for (int k = 0; k < 3; k++) weight_map[3 * i + k] = weight_min[j][k];
                if (return_depth) {
                    const float *fi = faces_inv + ((long)bn * nf + face_index_min[j]) * 9;
                    for (int k = 0; k < 9; k++) face_inv_map[9 * i + k] = fi[k];
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * K4: rasterize.py:361-438 -- trilinear texture sampling.
 * rgb_map / sampling maps pre-initialised to 0 by the caller (:482-484).
 * fix_batch_z == 0 reproduces the reference literally: the face's z is read from
 * faces[face_index * 9] (batch 0's geometry, :389 -- SURVEY quirk Q1); != 0 uses the pixel's own batch.
 */
API void oracle_forward_texture_sampling(const float *faces, const float *textures, const int32_t *face_index_map,
                                         const float *weight_map, const float *depth_map, float *rgb_map,
                                         int32_t *sampling_index_map, float *sampling_weight_map, int batch_size,
                                         int num_faces, int image_size, int texture_size, double eps,
                                         int fix_batch_z)
{
    const int is = image_size;
    const int nf = num_faces;
    const int ts = texture_size;
    const long n = (long)batch_size * is * is;
#pragma omp parallel for schedule(static) num_threads(NTHREADS)
    for (long i = 0; i < n; i++) {
        const int face_index = face_index_map[i];
        if (0 <= face_index) {
            const int bn = (int)(i / ((long)is * is));
            const float *face =
                fix_batch_z ? faces + ((long)bn * nf + face_index) * 9 : faces + (long)face_index * 9; /* :389 */
            const float *texture = textures + ((long)bn * nf + face_index) * ts * ts * ts * 3;          /* :390 */
            float *pixel = rgb_map + i * 3;
            const float *weight = weight_map + i * 3;
            const float depth = depth_map[i];

            /* :398-404 */
            float texture_index_float[3];
            for (int k = 0; k < 3; k++) {
                float tif = weight[k] * (float)(ts - 1) * (depth / (face[3 * k + 2]));
                tif = (float)fmax((double)tif, 0.);
                tif = (float)fmin((double)tif, (double)(ts - 1) - eps);
                texture_index_float[k] = tif;
            }

            /* :407-426 */
            float new_pixel[3] = {0, 0, 0};
            for (int pn = 0; pn < 8; pn++) {
                float w = 1;
                int texture_index_int[3];
                for (int k = 0; k < 3; k++) {
                    const int ti = f2i(texture_index_float[k]);
                    if ((pn >> k) % 2 == 0) {
                        w *= 1.0f - (texture_index_float[k] - (float)ti);
                        texture_index_int[k] = ti;
                    } else {
                        w *= texture_index_float[k] - (float)ti;
                        texture_index_int[k] = ti + 1;
                    }
                }
                int isc = texture_index_int[0] * ts * ts + texture_index_int[1] * ts + texture_index_int[2];
                /* isc >= ts^3: an index float hit ts - 1 exactly (eps lost in the float rounding of :402); the reference then
                 * reads past the cube with weight 0 -- not dereferenced here */
                if (isc < ts * ts * ts)
                    for (int k = 0; k < 3; k++) new_pixel[k] += w * texture[isc * 3 + k];
                if (sampling_index_map) sampling_index_map[i * 8 + pn] = isc;
                if (sampling_weight_map) sampling_weight_map[i * 8 + pn] = w;
            }
            for (int k = 0; k < 3; k++) pixel[k] = new_pixel[k];
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * K5: rasterize.py:440-465 -- alpha map and background blending.
 * background: [3] when bg_per_batch == 0, [B,3] otherwise (:462-465). Either map may be NULL.
 */
API void oracle_forward_background_alpha(const int32_t *face_index_map, float *rgb_map, float *alpha_map,
                                         const float *background, int bg_per_batch, int batch_size,
                                         int image_size)
{
    const int is = image_size;
    const long n = (long)batch_size * is * is;
#pragma omp parallel for schedule(static) num_threads(NTHREADS)
    for (long i = 0; i < n; i++) {
        const int bn = (int)(i / ((long)is * is));
        const float mask = (0 <= face_index_map[i]) ? 1.0f : 0.0f; /* :461 */
        if (rgb_map) {
            const float *bg = background + (bg_per_batch ? 3 * bn : 0);
            for (int k = 0; k < 3; k++) rgb_map[3 * i + k] = rgb_map[3 * i + k] * mask + (1.0f - mask) * bg[k]; /* :463 */
        }
        if (alpha_map && mask != 0.0f) alpha_map[i] = 1.0f; /* :449 */
    }
}

/* ------------------------------------------------------------------------------------------------
 * K6: rasterize.py:517-748 -- approximate gradient of rgb / alpha w.r.t. vertex x, y.
 * grad_faces [B*F*9] is STORED (every face written, zeros for back faces since the caller zero-fills
 * it first, :851, and back faces `return` before the store, :540).
 * Optional visit counter (may be NULL): number of pixel visits in the two sweeps (work statistic).
 * accumulate_double != 0 is NOT the reference: every per-pixel term is still computed with the reference's
 * float arithmetic, but the running sums are kept in double and rounded once at the end.  It isolates the
 * per-term arithmetic (what a parallel implementation must reproduce) from the order-dependent float
 * summation noise of the reference's serial loop; tests use both forms.
 */
API void oracle_backward_pixel_map(const float *faces, const int32_t *face_index_map, const float *rgb_map,
                                   const float *alpha_map, const float *grad_rgb_map,
                                   const float *grad_alpha_map, float *grad_faces, int batch_size, int num_faces,
                                   int image_size, double eps, int return_rgb, int return_alpha,
                                   long long *visit_counter, int accumulate_double)
{
    const int is = image_size;
    const long n = (long)batch_size * num_faces;
    long long visits = 0;
    if ((!return_rgb) && (!return_alpha)) return; /* :523 */
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : visits) num_threads(NTHREADS)
    for (long i = 0; i < n; i++) {
        const int bn = (int)(i / num_faces);
        const int fn = (int)(i % num_faces);
        const float *face = faces + i * 9;
        float grad_face[9] = {0};
        double grad_face_d[9] = {0};

        if (is_backside(face)) continue; /* :540 */

        for (int edge_num = 0; edge_num < 3; edge_num++) {
            int pi[3];
            float pp[3][2];
            for (int num = 0; num < 3; num++) pi[num] = (edge_num + num) % 3;
            for (int num = 0; num < 3; num++)
                for (int dim = 0; dim < 2; dim++)
                    pp[num][dim] = (float)(0.5 * (double)(face[3 * pi[num] + dim] * (float)is + (float)is - 1.0f)); /* :549 */

            for (int axis = 0; axis < 2; axis++) {
                float p[3][2];
                for (int num = 0; num < 3; num++)
                    for (int dim = 0; dim < 2; dim++) p[num][dim] = pp[num][(dim + axis) % 2];

                int direction; /* :559-564 */
                if (axis == 0) {
                    if (p[0][0] < p[1][0]) direction = -1; else direction = 1;
                } else {
                    if (p[0][0] < p[1][0]) direction = 1; else direction = -1;
                }

                /* :567-569 */
                const int d0_from = f2i(fmax((double)ceilf(fminf(p[0][0], p[1][0])), 0.));
                const int d0_to = f2i(fmin((double)fmaxf(p[0][0], p[1][0]), is - 1.));
                for (int d0 = d0_from; d0 <= d0_to; d0++) {
                    int d1_in, d1_out;
                    const float d1_cross =
                        (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]) * ((float)d0 - p[0][0]) + p[0][1]; /* :573 */
                    if (0 < direction) d1_in = f2i(floorf(d1_cross)); else d1_in = f2i(ceilf(d1_cross));
                    d1_out = d1_in + direction;

                    if (d1_in < 0 || is <= d1_in) continue;   /* :578 */
                    if (d1_out < 0 || is <= d1_out) continue; /* :579 */

                    float alpha_in = 0, alpha_out = 0;
                    const float *rgb_in = 0, *rgb_out = 0;
                    long map_index_in, map_index_out;
                    if (axis == 0) {
                        map_index_in = (long)bn * is * is + (long)d1_in * is + d0;
                        map_index_out = (long)bn * is * is + (long)d1_out * is + d0;
                    } else {
                        map_index_in = (long)bn * is * is + (long)d0 * is + d1_in;
                        map_index_out = (long)bn * is * is + (long)d0 * is + d1_out;
                    }
                    if (return_alpha) {
                        alpha_in = alpha_map[map_index_in];
                        alpha_out = alpha_map[map_index_out];
                    }
                    if (return_rgb) {
                        rgb_in = rgb_map + map_index_in * 3;
                        rgb_out = rgb_map + map_index_out * 3;
                    }

                    /* out: :604-659 */
                    const int is_in_fn = (face_index_map[map_index_in] == fn);
                    if (is_in_fn) {
                        int d1_limit;
                        if (0 < direction) d1_limit = is - 1; else d1_limit = 0;
                        const int d1_from = imax(imin(d1_out, d1_limit), 0);
                        const int d1_to = imin(imax(d1_out, d1_limit), is - 1);
                        long map_offset, map_index_from;
                        if (axis == 0) {
                            map_offset = is;
                            map_index_from = (long)bn * is * is + (long)d1_from * is + d0;
                        } else {
                            map_offset = 1;
                            map_index_from = (long)bn * is * is + (long)d0 * is + d1_from;
                        }
                        long idx = map_index_from;
                        for (int d1 = d1_from; d1 <= d1_to; d1++, idx += map_offset) {
                            float diff_grad = 0;
                            visits++;
                            if (return_alpha) diff_grad += (alpha_map[idx] - alpha_in) * grad_alpha_map[idx];
                            if (return_rgb)
                                for (int k = 0; k < 3; k++)
                                    diff_grad += (rgb_map[idx * 3 + k] - rgb_in[k]) * grad_rgb_map[idx * 3 + k];
                            if (diff_grad <= 0) continue; /* :647 */
                            if (p[1][0] != (float)d0) {   /* :648-652 */
                                float dist = (float)((double)((p[1][0] - p[0][0]) / (p[1][0] - (float)d0) *
                                                              ((float)d1 - d1_cross)) * 2. / is);
                                dist = (0 < dist) ? (float)((double)dist + eps) : (float)((double)dist - eps);
                                { const float term = diff_grad / dist; grad_face[pi[0] * 3 + (1 - axis)] -= term; grad_face_d[pi[0] * 3 + (1 - axis)] -= (double)term; }
                            }
                            if (p[0][0] != (float)d0) { /* :653-657 */
                                float dist = (float)((double)((p[1][0] - p[0][0]) / ((float)d0 - p[0][0]) *
                                                              ((float)d1 - d1_cross)) * 2. / is);
                                dist = (0 < dist) ? (float)((double)dist + eps) : (float)((double)dist - eps);
                                { const float term = diff_grad / dist; grad_face[pi[1] * 3 + (1 - axis)] -= term; grad_face_d[pi[1] * 3 + (1 - axis)] -= (double)term; }
                            }
                        }
                    }

                    /* in: :662-730 */
                    {
                        int d1_limit;
                        float d0_cross2;
                        if (((float)d0 - p[0][0]) * ((float)d0 - p[2][0]) < 0) {
                            d0_cross2 = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]) * ((float)d0 - p[0][0]) + p[0][1];
                        } else {
                            d0_cross2 = (p[1][1] - p[2][1]) / (p[1][0] - p[2][0]) * ((float)d0 - p[2][0]) + p[2][1];
                        }
                        if (0 < direction) d1_limit = f2i(ceilf(d0_cross2)); else d1_limit = f2i(floorf(d0_cross2));
                        const int d1_from = imax(imin(d1_in, d1_limit), 0);
                        const int d1_to = imin(imax(d1_in, d1_limit), is - 1);

                        long map_offset, map_index_from;
                        if (axis == 0) map_offset = is; else map_offset = 1;
                        if (axis == 0) {
                            map_index_from = (long)bn * is * is + (long)d1_from * is + d0;
                        } else {
                            map_index_from = (long)bn * is * is + (long)d0 * is + d1_from;
                        }
                        long idx = map_index_from;
                        for (int d1 = d1_from; d1 <= d1_to; d1++, idx += map_offset) {
                            visits++;
                            if (face_index_map[idx] != fn) continue; /* :707 */
                            float diff_grad = 0;
                            if (return_alpha) diff_grad += (alpha_map[idx] - alpha_out) * grad_alpha_map[idx];
                            if (return_rgb)
                                for (int k = 0; k < 3; k++)
                                    diff_grad += (rgb_map[idx * 3 + k] - rgb_out[k]) * grad_rgb_map[idx * 3 + k];
                            if (diff_grad <= 0) continue; /* :717 */
                            if (p[1][0] != (float)d0) {   /* :719-723 */
                                float dist = (float)((double)((p[1][0] - p[0][0]) / (p[1][0] - (float)d0) *
                                                              ((float)d1 - d1_cross)) * 2. / is);
                                dist = (0 < dist) ? (float)((double)dist + eps) : (float)((double)dist - eps);
                                { const float term = diff_grad / dist; grad_face[pi[0] * 3 + (1 - axis)] -= term; grad_face_d[pi[0] * 3 + (1 - axis)] -= (double)term; }
                            }
                            if (p[0][0] != (float)d0) { /* :724-728 */
                                float dist = (float)((double)((p[1][0] - p[0][0]) / ((float)d0 - p[0][0]) *
                                                              ((float)d1 - d1_cross)) * 2. / is);
                                dist = (0 < dist) ? (float)((double)dist + eps) : (float)((double)dist - eps);
                                { const float term = diff_grad / dist; grad_face[pi[1] * 3 + (1 - axis)] -= term; grad_face_d[pi[1] * 3 + (1 - axis)] -= (double)term; }
                            }
                        }
                    }
                }
            }
        }
        for (int k = 0; k < 9; k++) grad_faces[i * 9 + k] = accumulate_double ? (float)grad_face_d[k] : grad_face[k]; /* :736 */
    }
    if (visit_counter) *visit_counter = visits;
}

/* ------------------------------------------------------------------------------------------------
 * K7: rasterize.py:750-792 -- scatter of grad_rgb into the 8 sampled texels (accumulates).
 */
API void oracle_backward_textures(const int32_t *face_index_map, const float *sampling_weight_map,
                                  const int32_t *sampling_index_map, const float *grad_rgb_map,
                                  float *grad_textures, int batch_size, int num_faces, int image_size,
                                  int texture_size, double *acc_d)
{
    /* acc_d != NULL (NOT the reference): accumulate into this double buffer [same shape as grad_textures]
     * instead of grad_textures; see the note on accumulate_double at K6. */
    const int is = image_size;
    const int nf = num_faces;
    const int ts = texture_size;
    /* one batch element per thread: a face's texels only receive terms from its own image, in pixel order */
#pragma omp parallel for schedule(dynamic, 1) num_threads(NTHREADS)
    for (int bn = 0; bn < batch_size; bn++)
    for (long i = (long)bn * is * is; i < (long)(bn + 1) * is * is; i++) {
        const int face_index = face_index_map[i];
        if (0 <= face_index) {
            const long toff = ((long)bn * nf + face_index) * ts * ts * ts * 3;
            float *grad_texture = grad_textures + toff;
            for (int pn = 0; pn < 8; pn++) {
                const float w = sampling_weight_map[i * 8 + pn];
                const int isc = sampling_index_map[i * 8 + pn];
                if (isc >= ts * ts * ts) continue; /* a zero-weight tap outside the cube (see K4): the reference adds 0 there */
                for (int k = 0; k < 3; k++) {
                    const float term = w * grad_rgb_map[i * 3 + k]; /* :780 */
                    if (acc_d) acc_d[toff + isc * 3 + k] += (double)term; else grad_texture[isc * 3 + k] += term;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * K8: rasterize.py:794-847 -- analytic gradient of the depth map w.r.t. the winning face (accumulates
 * on top of what K6 stored, :881-883).
 */
API void oracle_backward_depth_map(const float *faces, const float *depth_map, const int32_t *face_index_map,
                                   const float *face_inv_map, const float *weight_map,
                                   const float *grad_depth_map, float *grad_faces, int batch_size, int num_faces,
                                   int image_size, double *acc_d)
{
    /* acc_d != NULL (NOT the reference): add the terms to this double buffer [B*F*9] instead of grad_faces. */
    const int is = image_size;
    const int nf = num_faces;
#pragma omp parallel for schedule(dynamic, 1) num_threads(NTHREADS)
    for (int bn = 0; bn < batch_size; bn++)
    for (long i = (long)bn * is * is; i < (long)(bn + 1) * is * is; i++) {
        const int fn = face_index_map[i];
        if (0 <= fn) {
            const float *face = faces + ((long)bn * nf + fn) * 9;
            const float depth = depth_map[i];
            const float depth2 = depth * depth;
            const float *face_inv = face_inv_map + i * 9;
            const float *weight = weight_map + i * 3;
            const float grad_depth = grad_depth_map[i];
            float *grad_face = grad_faces + ((long)bn * nf + fn) * 9;
            double *grad_face_d = acc_d ? acc_d + ((long)bn * nf + fn) * 9 : 0;

            /* :824-827 */
            for (int k = 0; k < 3; k++) {
                const float z_k = face[3 * k + 2];
                const float term = grad_depth * weight[k] * depth2 / (z_k * z_k);
                if (grad_face_d) grad_face_d[3 * k + 2] += (double)term; else grad_face[3 * k + 2] += term;
            }

            /* :830-837 */
            float tmp[3] = {0, 0, 0};
            for (int k = 0; k < 3; k++)
                for (int l = 0; l < 3; l++) tmp[k] += -face_inv[3 * l + k] / face[3 * l + 2];
            for (int k = 0; k < 3; k++)
                for (int l = 0; l < 2; l++)
                {
                    const float term = -grad_depth * tmp[l] * weight[k] * depth2 * (float)is / 2.0f;
                    if (grad_face_d) grad_face_d[3 * k + l] += (double)term; else grad_face[3 * k + l] += term;
                }
        }
    }
}

API int oracle_version(void) { return 4; }
