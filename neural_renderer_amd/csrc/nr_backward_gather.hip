// nr_backward_gather.hip -- K7 backward_textures (rasterize.py:750-792) and K8 backward_depth_map
// (rasterize.py:794-847) + their C-ABI entry points.
//
// The reference scatters from pixels: 24 (K7) / 9 (K8) float atomicAdds per covered pixel, all pixels of a
// face hitting the same few addresses.  On MI355X that formulation is bound by same-address atomic
// serialisation in L2 (measured: 0.9 ms / 0.33 ms at the headline size, profiles/r01a).  Both gradients are
// sums over "the pixels a face owns", so here they are GATHERED per face instead:
//   * a group of lanes (16, 64 or 256 depending on the texture size) owns one face, walks the pixels of the
//     face's screen box (the same box the forward used), keeps the pixels whose face_index equals the face,
//     evaluates the reference's per-pixel terms and accumulates them privately;
//   * K7: texture_size 2 (the Renderer default) has a static tap pattern -> 24 register accumulators per
//     lane, group reduction, one 96-byte store per face; larger cubes accumulate in LDS (double);
//     every element of grad_textures is STORED (zeros for faces that own nothing): no zero fill, no atomics;
//   * K8: 9 register accumulators, group reduction, one read-modify-write of the face's 9 floats by a single
//     lane (K6 stored them before, rasterize.py:881-883): no atomics.
// Per-pixel terms use the reference's arithmetic; the order of the additions differs (as it does between
// any two runs of the reference, whose atomics are unordered).
// Fused backward (nr_backward_rasterize): one gather serves K7 and K8, visits only the faces of K6's visible lists and either
// finishes K6 in its epilogue (large calls) or shares ONE launch with K6's line setup and the zeros of grad_textures, in front of
// K6's band kernel (calls of up to 96 k faces: k_setup_gather below).
#include "nr_device.h"
#include "nr_band_lines.h"
#include <type_traits>

using namespace nr;

namespace {

constexpr int BIG_PX = 2048;  // candidate sets above this size are walked by k_backward_big (A/B on config 4: 256 cost 0.25 ms)

__device__ __forceinline__ float group_sum(float v, int width)
{
    for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// Sum over a 16-lane group (= one DPP row), delivered in the group's LAST lane (sub == 15): four v_add_f32 with a row_shr
// DPP operand (lanes shifted in from outside the row read 0) instead of four LDS-crossbar swizzles + adds per value.  The 33
// sums of a face (24 texel + 9 vertex accumulators) make this the longest instruction run of the gather kernels.
__device__ __forceinline__ float row16_sum_last(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));  // row_shr:1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));  // row_shr:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));  // row_shr:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true));  // row_shr:8
    return v;
}

// K8's per-face constants (rasterize.py:830-833) when the inverse matrix is recomputed from the vertices: tmp_l = sum_m
// -face_inv[m][l] / z_m, evaluated ONCE per face with the reference's own operations (the three terms cancel: reciprocal
// shortcuts here showed up as 5e-4 in grad_faces), instead of once per pixel.  zz[k] = z_k * z_k (:826).
struct DepthConst {
    float tmp[3], zz[3];
};
__device__ __forceinline__ DepthConst depth_constants(const float f[9], int S)
{
    const float fs = (float)S;
    const float px[3] = {to_pixel(f[0], fs), to_pixel(f[3], fs), to_pixel(f[6], fs)};
    const float py[3] = {to_pixel(f[1], fs), to_pixel(f[4], fs), to_pixel(f[7], fs)};
    float inv[9];
    compute_face_inv(px, py, inv);
    DepthConst d;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        d.tmp[k] = 0.0f;
#pragma unroll
        for (int l = 0; l < 3; l++) d.tmp[k] += -inv[3 * l + k] / f[3 * l + 2];
        d.zz[k] = f[3 * k + 2] * f[3 * k + 2];
    }
    return d;
}

// Walk of a face's candidate pixels by its group of L lanes, the lanes of a wave in step, in two passes.  Ownership pass: one
// lane per candidate, only face_index_map is read; the owned pixels (a quarter of a typical box) are ballot-compacted into
// the group's LDS queue.  Evaluation pass: whenever a group has QL pixels waiting -- and once at the end, for all groups of the
// wave together -- each lane takes one owned pixel and calls eval(pixel offset in the image).  (With the ownership test in
// front of the evaluation in one loop, the evaluation ran in every candidate step in which *any* lane of the wave owned its
// pixel: three to four times per wave instead of once or twice.)  The order in which a lane meets its pixels, and therefore
// the rounding of its float sums, is fixed by the candidate order: results are reproducible and identical between the
// kernels that use this walk.
//   n_mine: candidates of this lane's face (0: none, the lane only keeps step); queue: 2 * QL words of LDS per queue group
//   (QL = min(L, 64) lanes: the face's group, or one wave of it when L == 256).
template <int STEPS, class Eval>
__device__ __forceinline__ void walk_owned_pixels(const Cand &cd, int n_mine, int fn, const int32_t *__restrict__ fi_img,
                                                  int S, int sub, int L, int *__restrict__ queue_base, Eval eval)
{
    const int tid = threadIdx.x;
    const int QL = L < 64 ? L : 64;
    int *queue = queue_base + (tid / QL) * (2 * QL);
    const int qsub = tid & (QL - 1);
    const int qshift = (tid & 63) & ~(QL - 1);
    const unsigned long long qmask = QL == 64 ? ~0ull : ((1ull << QL) - 1ull);
    int waiting = 0;
    // STEPS candidate steps per round (1 or 2): with 2, both ownership words are requested before either is used -- a face's
    // walk is a chain of dependent round trips (list, vertices, ownership, pixel data) and a workgroup's time is that chain's,
    // not its instructions' (K8's gather 62 -> 58 us) -- where the second pair of registers does not cost a wave of occupancy
    // (K7 + K8 with static taps: 95 -> 102 VGPRs, four waves per SIMD instead of five, 252 -> 261 us for the fused backward).
    // The steps themselves run one after the other, so the order in which a lane meets its pixels is the candidate order.
    for (int i = sub;; i += STEPS * L) {
        bool more2[STEPS], owned2[STEPS];
        int off2[STEPS];
#pragma unroll
        for (int h = 0; h < STEPS; ++h) {
            const int ii = i + h * L;
            more2[h] = ii < n_mine;
            int x = 0, y = 0, f = 0;
            off2[h] = 0;
            bool in = false;
            if (more2[h] && cand_pixel(cd, ii, S, x, y)) {
                off2[h] = y * S + x;
                f = fi_img[off2[h]];
                in = true;
            }
            owned2[h] = in && f == fn;
        }
        bool done = false;
#pragma unroll
        for (int h = 0; h < STEPS; ++h) {
            if (done) break;
            const bool more = more2[h], owned = owned2[h];
            const int off = off2[h];
            const bool any_more = __ballot(more) != 0ull;
            const unsigned long long m = (__ballot(owned) >> qshift) & qmask;
            if (owned) queue[waiting + __popcll(m & ((1ull << qsub) - 1ull))] = off;
            waiting += __popcll(m);
            const bool ready = waiting >= QL || (!any_more && waiting > 0);
            if (__ballot(ready) != 0ull) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int take = ready ? (waiting < QL ? waiting : QL) : 0;
                const int e = qsub < take ? queue[qsub] : 0;
                const int rest = (ready && qsub + QL < waiting) ? queue[qsub + QL] : 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                waiting -= take;
                if (ready && qsub < waiting) queue[qsub] = rest;
                if (qsub < take) eval(e);
            }
            if (!any_more) done = true;  // (the evaluation above took everything that was waiting)
        }
        if (done) break;
    }
}

// --------------------------------------------------------------------------------------------------
// B2: one face per group of L lanes (L = 16 | 64 | 256, a power of two; 256 / L faces per workgroup).
// TS2 = true: texture_size == 2 and eps > 0, so every tap index is static: corner pn -> texel
// (pn & 1) * 4 + ((pn >> 1) & 1) * 2 + ((pn >> 2) & 1)   (floor(tif) == 0 because tif <= 1 - eps, :402).
// DEPTH = true additionally evaluates K8 (backward_depth_map) for the same owned pixels and adds the face's 9
// sums onto grad_faces, so that one walk of the screen box serves both gradients (fused backward only).
// LIT = true: per-face light colours (FaceLight in nr_device.h) -- a template parameter, so that the kernels of the plain
// path are exactly what they were without it (as a run-time branch it cost them registers: K7 alone 72 -> 76 VGPRs, one
// wave of occupancy, 62 -> 71 us).
struct FaceGatherArgs {
    const int32_t *face_index_map;
    const float *sampling_weight_map;
    const int32_t *sampling_index_map;
    const float *faces, *zbase, *weight_map, *depth_map, *g_rgb;
    float *grad_textures;
    int n_faces_total, F, S, ts;
    double eps;
    int fix_batch_z, L;
    const int *vis_list, *vis_count;
    const float *g_depth;
    float *grad_faces;
    const double *k6_scratch;
    const int *slot_of;
    FaceLight lit;
};

// bx, by: the workgroup's place in the gather's grid (blockIdx of k_backward_textures_face)
template <bool TS2, bool DEPTH, bool LIT>
__device__ __forceinline__ void face_gather_body(const FaceGatherArgs &a, const int bx, const int by)
{
    const int32_t *__restrict__ face_index_map = a.face_index_map;
    const float *__restrict__ sampling_weight_map = a.sampling_weight_map;
    const int32_t *__restrict__ sampling_index_map = a.sampling_index_map;
    const float *__restrict__ faces = a.faces, *__restrict__ zbase = a.zbase, *__restrict__ weight_map = a.weight_map,
                *__restrict__ depth_map = a.depth_map, *__restrict__ g_rgb = a.g_rgb;
    float *__restrict__ grad_textures = a.grad_textures;
    const int n_faces_total = a.n_faces_total, F = a.F, S = a.S, ts = a.ts;
    const double eps = a.eps;
    const int fix_batch_z = a.fix_batch_z, L = a.L;
    const int *__restrict__ vis_list = a.vis_list, *__restrict__ vis_count = a.vis_count;
    const float *__restrict__ g_depth = a.g_depth;
    float *__restrict__ grad_faces = a.grad_faces;
    const double *__restrict__ k6_scratch = a.k6_scratch;
    const FaceLight &lit = a.lit;
    extern __shared__ __attribute__((aligned(16))) double s_acc[];  // [256 / L][ts^3 * 3] (general path)
    __shared__ int s_queue[512];  // owned pixels waiting for their evaluation (walk_owned_pixels)
    __shared__ int s_own;         // lit, L == 256: does the workgroup's face own a pixel?
    __shared__ float s_gl[3];     // lit, L == 256: the face's light-colour gradient

    const int tid = threadIdx.x;
    const int grp = tid / L, sub = tid - grp * L;
    const int n_tex = ts * ts * ts * 3;
    int gi = bx * (256 / L) + grp;  // global face index b * F + fn
    bool face_ok = gi < n_faces_total;
    int slot = 0;
    if (vis_list) {  // by = image, slot -> face through the image's visible list
        // The grid covers F list slots per image, the list holds the ~1/6 of them that own a pixel: the other workgroups
        // leave here (they used to run the 24-sum reduction below on zeros -- 40 % of the kernel's instructions at the
        // headline size).
        // (Fused backward with K6's scratch handed over: the epilogue also finishes K6 for the listed faces.  The unlisted
        // faces' zeros come from K6's compaction kernel (grad_faces) and from the fill in front of this launch (grad_textures):
        // storing them from here -- every workgroup the faces with its numbers -- was measured: neutral at the headline size,
        // +19 us on config 4 and 2.5x this kernel's time on 1024 views of 32 x 32, where 5 M faces mean 300 k workgroups that
        // each wait for a slot_of load before they can leave.)
        slot = gi;
        const int n_vis = vis_count[by];
        if (bx * (256 / L) >= n_vis) return;
        face_ok = slot < n_vis;
        // (requesting the list entry beside the list's length instead of behind it -- one round trip less in front of the walk
        // -- was measured: 48.7 vs 46.6 us at 64 views, 20.4 vs 20.0 at 8; five of six workgroups only want the length)
        gi = face_ok ? by * F + vis_list[(size_t)by * F + slot] : 0;
    }
    double *acc_l = s_acc + (size_t)grp * n_tex;

    float acc[24];
#pragma unroll
    for (int k = 0; k < 24; k++) acc[k] = 0.0f;
    float dacc[9];
#pragma unroll
    for (int k = 0; k < 9; k++) dacc[k] = 0.0f;
    bool any_box = false;
    bool own = false;  // this lane evaluated a pixel of the face (lit: only such faces store, see FaceLight)
    if (!TS2) {
        for (int k = sub; k < n_tex; k += L) acc_l[k] = 0.0;
        if (LIT) {
            if (tid < 3) s_gl[tid] = 0.0f;
            if (tid == 0) s_own = 0;
        }
        __syncthreads();
    }

    Cand cd;
    cd.n = 0;
    int fn = 0;
    size_t img = 0;
    DepthConst dc;
#pragma unroll
    for (int k = 0; k < 3; k++) dc.tmp[k] = dc.zz[k] = 0.0f;
    float face_z[3] = {1.0f, 1.0f, 1.0f};
    // lit: the face's original cube.  Its reversed copy shares it and samples it with axes 0 and 2 exchanged: the walk below
    // flattens that copy's taps in the ORIGINAL layout (compute_taps' flip), so the sums need no transposition afterwards.
    bool flip = false;
    size_t cube = 0;  // b * Nf + original face
    if (LIT) {
        const int b = vis_list ? by : gi / F, f = gi - b * F;
        flip = f >= lit.tex_faces;
        cube = (size_t)b * lit.tex_faces + (flip ? f - lit.tex_faces : f);
    }
    if (face_ok) {
        const int b = gi / F;
        fn = gi - b * F;
        const float *f = faces + (size_t)gi * 9;
        cd = face_candidates(f[0], f[1], f[3], f[4], f[6], f[7], S);
        if (cd.n > 0 && (L == 256 || (cd.n <= BIG_PX))) {  // the rest is k_backward_big's
            any_box = true;
            if (DEPTH) {
                float fv[9];
#pragma unroll
                for (int k = 0; k < 9; k++) fv[k] = f[k];
                dc = depth_constants(fv, S);
            }
            // z of the three vertices as the forward sampled them: batch 0's geometry (zbase) unless fixed (:389, Q1)
            const float *fz = (fix_batch_z ? faces + (size_t)b * F * 9 : zbase) + (size_t)fn * 9;
            face_z[0] = fz[2]; face_z[1] = fz[5]; face_z[2] = fz[8];
            img = (size_t)b * S * S;
        }
    }
    walk_owned_pixels<(TS2 && DEPTH) ? 1 : 2>(cd, any_box ? cd.n : 0, fn, face_index_map + img, S, sub, L, s_queue, [&](int off) {
        const size_t p = img + (size_t)off;
        float wk[3] = {0.0f, 0.0f, 0.0f}, depth = 0.0f, gd = 0.0f;
        if (weight_map) { wk[0] = weight_map[3 * p]; wk[1] = weight_map[3 * p + 1]; wk[2] = weight_map[3 * p + 2]; }
        if (depth_map) depth = depth_map[p];
        if (DEPTH) gd = g_depth[p];
        const float g[3] = {g_rgb[3 * p], g_rgb[3 * p + 1], g_rgb[3 * p + 2]};
        if (LIT) own = true;
        Taps t;
        if (sampling_weight_map) {
#pragma unroll
            for (int pn = 0; pn < 8; pn++) {
                t.w[pn] = sampling_weight_map[8 * p + pn];
                t.isc[pn] = sampling_index_map[8 * p + pn];
            }
        } else {
            compute_taps(face_z, wk, depth, ts, eps, t, LIT && flip);
        }
        if (DEPTH) {  // K8 terms of this pixel (rasterize.py:824-837), as in k_backward_depth_face
            const float depth2 = depth * depth;
#pragma unroll
            for (int k = 0; k < 3; k++) dacc[3 * k + 2] += gd * wk[k] * depth2 / dc.zz[k];
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int l = 0; l < 2; l++) dacc[3 * k + l] += -gd * dc.tmp[l] * wk[k] * depth2 * (float)S / 2.0f;
        }
#pragma unroll
        for (int pn = 0; pn < 8; pn++) {
            if (TS2) {
                acc[3 * pn + 0] += t.w[pn] * g[0];  // :780
                acc[3 * pn + 1] += t.w[pn] * g[1];
                acc[3 * pn + 2] += t.w[pn] * g[2];
            } else {
                if (t.isc[pn] * 3 >= n_tex) continue;  // outside the cube: weight 0 (compute_taps)
                double *q = acc_l + t.isc[pn] * 3;
                atomicAdd(q + 0, (double)(t.w[pn] * g[0]));
                atomicAdd(q + 1, (double)(t.w[pn] * g[1]));
                atomicAdd(q + 2, (double)(t.w[pn] * g[2]));
            }
        }
    });

    // lit: does the face own a pixel at all?  (only then it stores: FaceLight)
    bool owned = false;
    if (LIT) {
        if (L <= 64) {
            const unsigned long long bm = __ballot(own);
            owned = L == 64 ? bm != 0ull : ((bm >> (tid & 48)) & 0xffffull) != 0ull;
        } else {
            if (own) s_own = 1;
            __syncthreads();
            owned = s_own != 0;
        }
    }
    if (TS2) {
        // L == 16 here: reduce inside the 16-lane row, its last lane stores the face's 24 floats (96 B)
#pragma unroll
        for (int k = 0; k < 24; k++) acc[k] = row16_sum_last(acc[k]);
        if (LIT) {
            if (face_ok && sub == 15 && owned) {
                // corner pn holds texel bitrev3(pn) of the sampled cube (the static taps above); the reversed copy samples the
                // transposed cube, whose texel bitrev3(pn) is texel pn of the original one
                const float *lc = lit.light + (size_t)gi * 3;
                const float l3[3] = {lc[0], lc[1], lc[2]};
                float tx[24];
                if (lit.textures) {  // 96 B per cube, 16 B aligned (nr_hip.h)
                    const float4 *src = reinterpret_cast<const float4 *>(lit.textures + cube * 24);
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        const float4 v = src[k];
                        tx[4 * k] = v.x; tx[4 * k + 1] = v.y; tx[4 * k + 2] = v.z; tx[4 * k + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 24; k++) tx[k] = 0.0f;
                }
                float o[24], gl[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int r = (u & 1) * 4 + (u & 2) + (u >> 2);
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float a = flip ? acc[3 * u + c] : acc[3 * r + c];
                        o[3 * u + c] = a * l3[c];
                        gl[c] += a * tx[3 * u + c];
                    }
                }
                float4 *dst = reinterpret_cast<float4 *>(grad_textures + cube * 24);
#pragma unroll
                for (int k = 0; k < 6; k++) dst[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
                if (lit.grad_light) {
                    float *gd = lit.grad_light + (size_t)gi * 3;
                    gd[0] = gl[0]; gd[1] = gl[1]; gd[2] = gl[2];
                }
            }
        } else if (face_ok && sub == 15) {
            float o[24];
#pragma unroll
            for (int pn = 0; pn < 8; pn++) {
                const int isc = (pn & 1) * 4 + ((pn >> 1) & 1) * 2 + ((pn >> 2) & 1);
                o[3 * isc + 0] = acc[3 * pn + 0];
                o[3 * isc + 1] = acc[3 * pn + 1];
                o[3 * isc + 2] = acc[3 * pn + 2];
            }
            float4 *dst = reinterpret_cast<float4 *>(grad_textures + (size_t)gi * 24);
#pragma unroll
            for (int k = 0; k < 6; k++) dst[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
        }
    } else if (LIT) {
        __syncthreads();
        float gl0 = 0.0f, gl1 = 0.0f, gl2 = 0.0f;
        if (face_ok && owned) {
            const float *lc = lit.light + (size_t)gi * 3;
            const float l3[3] = {lc[0], lc[1], lc[2]};
            const float *tex = lit.textures ? lit.textures + cube * n_tex : nullptr;
            float *dst = grad_textures + cube * n_tex;
            // (the loads of the cube in a loop of their own: interleaved with the stores, which the compiler must assume to
            // alias them, every one of them would be a separate round trip)
            if (tex) {
                for (int k = sub; k < n_tex; k += L) {
                    const int c = k % 3;
                    const float v = (float)acc_l[k] * tex[k];
                    gl0 += c == 0 ? v : 0.0f;
                    gl1 += c == 1 ? v : 0.0f;
                    gl2 += c == 2 ? v : 0.0f;
                }
            }
            for (int k = sub; k < n_tex; k += L) {
                const int c = k % 3;
                dst[k] = (float)acc_l[k] * (c == 0 ? l3[0] : (c == 1 ? l3[1] : l3[2]));
            }
        }
        if (lit.grad_light) {
            if (L <= 64) {
                gl0 = group_sum(gl0, L); gl1 = group_sum(gl1, L); gl2 = group_sum(gl2, L);
            } else {
                if (gl0 != 0.0f) atomicAdd(&s_gl[0], gl0);
                if (gl1 != 0.0f) atomicAdd(&s_gl[1], gl1);
                if (gl2 != 0.0f) atomicAdd(&s_gl[2], gl2);
                __syncthreads();
                gl0 = s_gl[0]; gl1 = s_gl[1]; gl2 = s_gl[2];
            }
            if (face_ok && owned && sub == 0) {
                float *gd = lit.grad_light + (size_t)gi * 3;
                gd[0] = gl0; gd[1] = gl1; gd[2] = gl2;
            }
        }
    } else {
        __syncthreads();
        if (face_ok) {
            float *dst = grad_textures + (size_t)gi * n_tex;
            for (int k = sub; k < n_tex; k += L) dst[k] = (float)acc_l[k];
        }
    }
    if (DEPTH || k6_scratch) {  // L <= 64 when DEPTH (the host only fuses K8 when a face group fits in one wave)
        if (DEPTH && __ballot(any_box) != 0ull) {
#pragma unroll
            for (int k = 0; k < 9; k++) dacc[k] = (L == 16) ? row16_sum_last(dacc[k]) : group_sum(dacc[k], L);
        }
        if (face_ok && sub == ((L == 16) ? 15 : 0)) {
            float *gf = grad_faces + (size_t)gi * 9;
            if (k6_scratch) {
                // K6's result for this face (rasterize.py:736 stores, K8 then accumulates, :881-883): the double sums of
                // its list position rounded to float, z = 0; the K8 sums (zero without a box of this kernel's) on top
                const double *sc = k6_scratch + ((size_t)by * F + slot) * 6;
#pragma unroll
                for (int v = 0; v < 3; v++) {
                    gf[3 * v + 0] = (float)sc[2 * v + 0] + dacc[3 * v + 0];
                    gf[3 * v + 1] = (float)sc[2 * v + 1] + dacc[3 * v + 1];
                    gf[3 * v + 2] = 0.0f + dacc[3 * v + 2];
                }
            } else if (any_box) {
#pragma unroll
                for (int k = 0; k < 9; k++) gf[k] += dacc[k];
            }
        }
    }
}

template <bool TS2, bool DEPTH, bool LIT>
__global__ __launch_bounds__(256) void k_backward_textures_face(FaceGatherArgs a)
{
    face_gather_body<TS2, DEPTH, LIT>(a, (int)blockIdx.x, (int)blockIdx.y);
}

// --------------------------------------------------------------------------------------------------
// The fused backward's launch between K6's compaction and its band kernel: the line setup (nr_band_lines.h), the K7 / K8
// gather and the zeros of grad_textures in ONE grid.  All three need the visible-face lists and nothing else of each other;
// the first two are latency-bound launches that each leave most of the chip idle (at the headline size 25.8 us on ~1000
// working workgroups and 44.6 us on ~1900), the third is what would otherwise be a fill in front of the gather.  Side by
// side they take what the longest takes.  (On two streams with event fork / join the same overlap cost 8 + 6 us of barrier
// packets on the critical stream and a finishing launch: 3 us gained of 250; LAB-NOTEBOOK, round 4.)
//   grid.x = [line setup | gather | zeros], grid.y = image.  The line-setup workgroups come first: the band kernel behind
//   this launch waits for their records, the chip's dispatcher hands out workgroups in grid order.
//   Zeros: every unlisted face's cube (slot_of < 0: the gather stores the listed ones completely), 2048 elements of 16 or 4
//   bytes per workgroup, so that no fill has to finish before the gather may store.
struct ZeroArgs {
    float *grad_textures;
    const int *slot_of;
    int F;
    int epf;        // elements per face cube: ts^3 * 3 floats, or a quarter of that in 16-byte elements
    int vec;        // 16-byte elements (the cube is a multiple of four floats and the array 16-byte aligned)
    unsigned wgs;   // zero workgroups per image
};

__device__ __forceinline__ void zero_unlisted_body(const ZeroArgs &z, const int bx, const int by)
{
    const size_t per_image = (size_t)z.F * z.epf;
    const size_t e0 = (size_t)bx * 2048, e1 = e0 + 2048 < per_image ? e0 + 2048 : per_image;
    const int *__restrict__ slot = z.slot_of + (size_t)by * z.F;
    for (size_t e = e0 + threadIdx.x; e < e1; e += 256) {
        const int face = (int)(e / (unsigned)z.epf);
        if (slot[face] >= 0) continue;
        const size_t at = (size_t)by * per_image + e;
        if (z.vec) reinterpret_cast<float4 *>(z.grad_textures)[at] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        else z.grad_textures[at] = 0.0f;
    }
}

template <bool TS2, bool DEPTH, bool LIT>
__global__ __launch_bounds__(256) void k_setup_gather(LineSetupArgs ls, FaceGatherArgs g, ZeroArgs z, unsigned gather_x)
{
    const unsigned bx = blockIdx.x;
    if (bx < ls.grid_x) line_setup_body(ls, (int)bx, (int)blockIdx.y);
    else if (bx < ls.grid_x + gather_x) face_gather_body<TS2, DEPTH, LIT>(g, (int)(bx - ls.grid_x), (int)blockIdx.y);
    else zero_unlisted_body(z, (int)(bx - ls.grid_x - gather_x), (int)blockIdx.y);
}

// --------------------------------------------------------------------------------------------------
// Faces with many candidate pixels (a ground plane, a backdrop, the strip of a needle) would keep one 16-lane group of
// the kernels above busy for thousands of iterations while the rest of the chip idles.  Those kernels therefore leave
// every face whose candidate set exceeds BIG_PX pixels untouched (zeros stored / nothing added), and this
// kernel, launched right after them, gives each such face a whole workgroup: one thread per face finds the big ones of a
// 256-face range, then all 256 lanes walk each of them in turn (coalesced rows), texel sums in LDS doubles, depth sums
// through a wave + LDS reduction.  With no big face in the range the workgroup exits after ~150 instructions.
// TEX: 0 no textures, 1 grad_textures through per-face LDS double accumulators (any texture_size <= 8), 2 texture_size 2 with
// static taps (24 register sums per lane, as in the TS2 gather); DEPTH: the K8 terms.
template <int TEX, bool DEPTH, bool LIT>
__global__ __launch_bounds__(256) void k_backward_big(
    const int32_t *__restrict__ face_index_map, const float *__restrict__ sampling_weight_map,
    const int32_t *__restrict__ sampling_index_map, const float *__restrict__ face_inv_map, const float *__restrict__ faces,
    const float *__restrict__ zbase, const float *__restrict__ weight_map, const float *__restrict__ depth_map, const float *__restrict__ g_rgb,
    float *__restrict__ grad_textures, int n_faces_total, int F, int S, int ts, double eps, int fix_batch_z,
    const int *__restrict__ vis_list, const int *__restrict__ vis_count, const float *__restrict__ g_depth,
    float *__restrict__ grad_faces, const unsigned char *__restrict__ visible, FaceLight lit,
    const double *__restrict__ k6_scratch)
{
    extern __shared__ __attribute__((aligned(16))) double s_tex[];  // [ts^3 * 3] texel sums of the face being walked
    __shared__ int s_list[256];
    __shared__ int s_wave_n[4];
    __shared__ float s_red[36];  // 9 depth sums + 24 texel sums (TEX == 2) + 3 light-colour sums (lit)
    __shared__ int s_own;        // lit: the face being walked owns a pixel
    const int tid = threadIdx.x;
    if (vis_list && (int)blockIdx.x * 256 >= vis_count[blockIdx.y]) return;  // slots behind the image's list
    int n_big;
    {   // one face per thread: is its candidate set this kernel's business?  The list is built in thread order, so that the
        // gridDim.z workgroups that scan the same range agree on it and can share it out (entry q -> workgroup q % gridDim.z:
        // with one workgroup per range a 2048 x 2048 view, whose faces are all "big", kept 20 workgroups busy for 5 ms).
        int gi = blockIdx.x * 256 + tid;
        bool ok = gi < n_faces_total;
        if (vis_list) {
            ok = gi < vis_count[blockIdx.y];
            gi = ok ? (int)blockIdx.y * F + vis_list[(size_t)blockIdx.y * F + gi] : 0;
        }
        if (k6_scratch && ok && blockIdx.z == 0) {
            // K6's last step for the listed faces rides in this launch (the fused backward whose gather ran in front of the band
            // kernel: grad_faces holds the gather's K8 sums, or zeros): K6's double sums of the face's list position, rounded, go
            // on top -- as float atomics, requested here beside the face's vertices: a face of this kernel's own also receives
            // its K8 sums from one of the launch's workgroups (below), and the two additions onto the gather's zero commute.
            const double *sc = k6_scratch + ((size_t)blockIdx.y * F + blockIdx.x * 256 + tid) * 6;
            float *gf = grad_faces + (size_t)gi * 9;
#pragma unroll
            for (int v = 0; v < 3; v++) {
                atomicAdd(gf + 3 * v + 0, (float)sc[2 * v + 0]);
                atomicAdd(gf + 3 * v + 1, (float)sc[2 * v + 1]);
            }
        }
        bool big = false;
        if (ok && !vis_list && visible && !visible[gi]) ok = false;
        if (ok) {
            const float *f = faces + (size_t)gi * 9;
            const Cand cd = face_candidates(f[0], f[1], f[3], f[4], f[6], f[7], S);
            big = cd.n > BIG_PX;
        }
        const unsigned long long m = __ballot(big);
        const int lane = tid & 63, wave = tid >> 6;
        if (lane == 0) s_wave_n[wave] = __popcll(m);
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; w++) before += s_wave_n[w];
        if (big) s_list[before + __popcll(m & ((1ull << lane) - 1ull))] = gi;
        n_big = s_wave_n[0] + s_wave_n[1] + s_wave_n[2] + s_wave_n[3];
        __syncthreads();
    }
    const int n_tex = TEX ? ts * ts * ts * 3 : 0;
    const int n_lds = TEX == 1 ? n_tex : 0;
    for (int q = blockIdx.z; q < n_big; q += gridDim.z) {
        const int gi = s_list[q];
        const int b = gi / F, fn = gi - b * F;
        const float *fp = faces + (size_t)gi * 9;
        float f[9];
#pragma unroll
        for (int k = 0; k < 9; k++) f[k] = fp[k];
        const Cand cd = face_candidates(f[0], f[1], f[3], f[4], f[6], f[7], S);
        DepthConst dc;
        if (DEPTH) dc = depth_constants(f, S);
        const float *fz = (fix_batch_z ? faces + (size_t)b * F * 9 : zbase) + (size_t)fn * 9;  // :389, Q1
        const float face_z[3] = {fz[2], fz[5], fz[8]};
        const bool flip = TEX && LIT && fn >= lit.tex_faces;  // the reversed copy: taps in the original cube's layout
        for (int k = tid; k < n_lds; k += 256) s_tex[k] = 0.0;
        if (tid < 36) s_red[tid] = 0.0f;
        if (LIT && tid == 0) s_own = 0;
        __syncthreads();
        bool own = false;
        float dacc[9], tacc[24];
#pragma unroll
        for (int k = 0; k < 9; k++) dacc[k] = 0.0f;
#pragma unroll
        for (int k = 0; k < 24; k++) tacc[k] = 0.0f;
        const size_t img = (size_t)b * S * S;
        for (int i = tid; i < cd.n; i += 256) {
            int x, y;
            if (!cand_pixel(cd, i, S, x, y)) continue;
            const size_t p = img + (size_t)y * S + x;
            if (face_index_map[p] != fn) continue;
            if (LIT) own = true;
            float wk[3] = {0.0f, 0.0f, 0.0f}, depth = 0.0f;
            if (weight_map) { wk[0] = weight_map[3 * p]; wk[1] = weight_map[3 * p + 1]; wk[2] = weight_map[3 * p + 2]; }
            if (depth_map) depth = depth_map[p];
            if (TEX) {
                Taps t;
                if (sampling_weight_map) {
#pragma unroll
                    for (int pn = 0; pn < 8; pn++) {
                        t.w[pn] = sampling_weight_map[8 * p + pn];
                        t.isc[pn] = sampling_index_map[8 * p + pn];
                    }
                } else {
                    compute_taps(face_z, wk, depth, ts, eps, t, flip);
                }
                const float g[3] = {g_rgb[3 * p], g_rgb[3 * p + 1], g_rgb[3 * p + 2]};
#pragma unroll
                for (int pn = 0; pn < 8; pn++) {
                    if (TEX == 2) {
                        tacc[3 * pn + 0] += t.w[pn] * g[0];  // :780
                        tacc[3 * pn + 1] += t.w[pn] * g[1];
                        tacc[3 * pn + 2] += t.w[pn] * g[2];
                    } else {
                        if (t.isc[pn] * 3 >= n_tex) continue;  // outside the cube: weight 0 (compute_taps)
                        double *a = s_tex + t.isc[pn] * 3;
                        atomicAdd(a + 0, (double)(t.w[pn] * g[0]));
                        atomicAdd(a + 1, (double)(t.w[pn] * g[1]));
                        atomicAdd(a + 2, (double)(t.w[pn] * g[2]));
                    }
                }
            }
            if (DEPTH) {  // rasterize.py:824-837, as in k_backward_depth_face
                const float gd = g_depth[p];
                const float depth2 = depth * depth;
                float tmp[3] = {dc.tmp[0], dc.tmp[1], dc.tmp[2]};
                if (face_inv_map) {  // the reference's per-pixel residual: its values, its divisions
                    tmp[0] = tmp[1] = tmp[2] = 0.0f;
#pragma unroll
                    for (int k = 0; k < 3; k++)
#pragma unroll
                        for (int l = 0; l < 3; l++) tmp[k] += -face_inv_map[9 * p + 3 * l + k] / f[3 * l + 2];
                }
#pragma unroll
                for (int k = 0; k < 3; k++) dacc[3 * k + 2] += gd * wk[k] * depth2 / dc.zz[k];
#pragma unroll
                for (int k = 0; k < 3; k++)
#pragma unroll
                    for (int l = 0; l < 2; l++) dacc[3 * k + l] += -gd * tmp[l] * wk[k] * depth2 * (float)S / 2.0f;
            }
        }
        if (DEPTH) {
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const float v = group_sum(dacc[k], 64);
                if ((tid & 63) == 0 && v != 0.0f) atomicAdd(&s_red[k], v);
            }
        }
        if (TEX == 2) {
#pragma unroll
            for (int k = 0; k < 24; k++) {
                const float v = group_sum(tacc[k], 64);
                if ((tid & 63) == 0 && v != 0.0f) atomicAdd(&s_red[9 + k], v);
            }
        }
        if (TEX && LIT && own) s_own = 1;
        __syncthreads();
        if (TEX && LIT) {  // the original cube, times the face's light colour; only a face that owns a pixel stores
            if (s_own) {
                const size_t cube = (size_t)b * lit.tex_faces + (flip ? fn - lit.tex_faces : fn);
                const float *lc = lit.light + (size_t)gi * 3;
                const float *tex = lit.textures ? lit.textures + cube * n_tex : nullptr;
                float *dst = grad_textures + cube * n_tex;
                for (int k = tid; k < n_tex; k += 256) {
                    const int q = k / 3, c = k - 3 * q;
                    int u = k;
                    float a;
                    if (TEX == 2) {  // k = 3 * corner + c, the corner's texel as in the TS2 gather
                        u = 3 * (flip ? q : (q & 1) * 4 + (q & 2) + (q >> 2)) + c;
                        a = s_red[9 + k];
                    } else {
                        a = (float)s_tex[k];
                    }
                    dst[u] = a * lc[c];
                    if (tex && a != 0.0f) atomicAdd(&s_red[33 + c], a * tex[u]);
                }
                if (lit.grad_light) {
                    __syncthreads();
                    if (tid < 3) lit.grad_light[(size_t)gi * 3 + tid] = s_red[33 + tid];
                }
            }
        } else {
        if (TEX == 1) {
            float *dst = grad_textures + (size_t)gi * n_tex;
            for (int k = tid; k < n_tex; k += 256) dst[k] = (float)s_tex[k];
        }
        if (TEX == 2 && tid < 24) {  // corner pn = tid / 3 -> texel (pn & 1) * 4 + ((pn >> 1) & 1) * 2 + ((pn >> 2) & 1)
            const int pn = tid / 3, c = tid - 3 * pn;
            const int isc = (pn & 1) * 4 + ((pn >> 1) & 1) * 2 + ((pn >> 2) & 1);
            grad_textures[(size_t)gi * 24 + 3 * isc + c] = s_red[9 + tid];
        }
        }
        if (DEPTH && tid < 9) {
            if (k6_scratch) atomicAdd(grad_faces + (size_t)gi * 9 + tid, s_red[tid]);  // (see the top of the kernel)
            else grad_faces[(size_t)gi * 9 + tid] += s_red[tid];
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------------------
// B2 (fallback, texture_size > 13): per-pixel scatter with hardware f32 atomics (-munsafe-fp-atomics =>
// global_atomic_add_f32), the reference's own formulation (rasterize.py:750-792).  The caller's
// zero fill is replaced by a hipMemsetAsync in the entry point.
__global__ __launch_bounds__(256) void k_backward_textures_atomic(
    const int32_t *__restrict__ face_index_map, const float *__restrict__ sampling_weight_map,
    const int32_t *__restrict__ sampling_index_map, const float *__restrict__ faces,
    const float *__restrict__ zbase, const float *__restrict__ weight_map, const float *__restrict__ depth_map,
    const float *__restrict__ g_rgb, float *__restrict__ grad_textures, int F, int S, int ts, double eps,
    int fix_batch_z, size_t n_pixels)
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
        const float *face = (fix_batch_z ? faces + (size_t)b * F * 9 : zbase) + (size_t)fi * 9;
        const float w[3] = {weight_map[3 * i], weight_map[3 * i + 1], weight_map[3 * i + 2]};
        const float fz[3] = {face[2], face[5], face[8]};
        compute_taps(fz, w, depth_map[i], ts, eps, t);
    }
    const float g[3] = {g_rgb[3 * i], g_rgb[3 * i + 1], g_rgb[3 * i + 2]};
    float *gt = grad_textures + ((size_t)b * F + fi) * ts * ts * ts * 3;
#pragma unroll
    for (int pn = 0; pn < 8; pn++) {
        if (t.isc[pn] >= ts * ts * ts) continue;  // outside the cube: weight 0 (compute_taps)
        float *p = gt + t.isc[pn] * 3;
        atomicAdd(p + 0, t.w[pn] * g[0]);  // :780
        atomicAdd(p + 1, t.w[pn] * g[1]);
        atomicAdd(p + 2, t.w[pn] * g[2]);
    }
}


// the same with per-face light colours (FaceLight): the taps are flattened in the original cube's layout, the contribution
// carries the face's colour, and the colour's own gradient -- g_c times the pixel's unlit sample -- goes to grad_light
__global__ __launch_bounds__(256) void k_backward_textures_atomic_lit(
    const int32_t *__restrict__ face_index_map, const float *__restrict__ faces, const float *__restrict__ zbase,
    const float *__restrict__ weight_map, const float *__restrict__ depth_map, const float *__restrict__ g_rgb,
    float *__restrict__ grad_textures, int F, int S, int ts, double eps, int fix_batch_z, size_t n_pixels, FaceLight lit)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    const int fi = face_index_map[i];
    if (fi < 0) return;
    const int b = (int)(i / ((size_t)S * S));
    const bool flip = fi >= lit.tex_faces;
    const size_t cube = ((size_t)b * lit.tex_faces + (flip ? fi - lit.tex_faces : fi)) * ts * ts * ts * 3;
    Taps t;
    const float *face = (fix_batch_z ? faces + (size_t)b * F * 9 : zbase) + (size_t)fi * 9;
    const float w[3] = {weight_map[3 * i], weight_map[3 * i + 1], weight_map[3 * i + 2]};
    const float fz[3] = {face[2], face[5], face[8]};
    compute_taps(fz, w, depth_map[i], ts, eps, t, flip);
    const float g[3] = {g_rgb[3 * i], g_rgb[3 * i + 1], g_rgb[3 * i + 2]};
    const float *lc = lit.light + ((size_t)b * F + fi) * 3;
    const float l3[3] = {lc[0], lc[1], lc[2]};
    float *gt = grad_textures + cube;
    const float *tex = lit.textures ? lit.textures + cube : nullptr;
    float sample[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int pn = 0; pn < 8; pn++) {
        if (t.isc[pn] >= ts * ts * ts) continue;  // outside the cube: weight 0 (compute_taps)
        float *p = gt + t.isc[pn] * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            atomicAdd(p + c, (t.w[pn] * g[c]) * l3[c]);
            if (tex) sample[c] += t.w[pn] * tex[t.isc[pn] * 3 + c];
        }
    }
    if (lit.grad_light) {
        float *gl = lit.grad_light + ((size_t)b * F + fi) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) atomicAdd(gl + c, g[c] * sample[c]);
    }
}

// --------------------------------------------------------------------------------------------------
// B3: one face per group of 16 lanes; 9 register accumulators; a single lane adds the totals onto what K6
// stored.  The per-face inverse matrix is recomputed from the vertices with the forward's arithmetic
// (compute_face_inv) unless the caller supplies the reference's per-pixel face_inv_map residual.
__global__ __launch_bounds__(256) void k_backward_depth_face(
    const float *__restrict__ faces, const float *__restrict__ depth_map, const int32_t *__restrict__ face_index_map,
    const float *__restrict__ face_inv_map, const float *__restrict__ weight_map, const float *__restrict__ g_depth,
    float *__restrict__ grad_faces, int n_faces_total, int F, int S, const int *__restrict__ vis_list,
    const int *__restrict__ vis_count, const unsigned char *__restrict__ visible)
{
    constexpr int L = 16;
    const int tid = threadIdx.x;
    const int grp = tid / L, sub = tid - grp * L;
    if (vis_list && (int)blockIdx.x * (256 / L) >= vis_count[blockIdx.y]) return;  // slots behind the image's list
    int gi = blockIdx.x * (256 / L) + grp;
    bool face_ok = gi < n_faces_total;
    if (vis_list) {
        const int slot = gi;
        face_ok = slot < vis_count[blockIdx.y];
        gi = face_ok ? (int)blockIdx.y * F + vis_list[(size_t)blockIdx.y * F + slot] : 0;
    } else if (visible && face_ok && !visible[gi]) {
        face_ok = false;  // the forward's flags (depth-only rendering has no K6 lists): 5/6 of a mesh's faces own no pixel
    }
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] = 0.0f;
    bool any_box = false;
    __shared__ int s_queue[512];
    Cand cd;
    cd.n = 0;
    int fn = 0;
    size_t img = 0;
    float f[9];
#pragma unroll
    for (int k = 0; k < 9; k++) f[k] = 1.0f;
    DepthConst dc;
#pragma unroll
    for (int k = 0; k < 3; k++) dc.tmp[k] = dc.zz[k] = 0.0f;
    if (face_ok) {
        const int b = gi / F;
        fn = gi - b * F;
        const float *fp = faces + (size_t)gi * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) f[k] = fp[k];
        cd = face_candidates(f[0], f[1], f[3], f[4], f[6], f[7], S);
        if (cd.n > 0 && cd.n <= BIG_PX) {  // the rest is k_backward_big's
            any_box = true;
            dc = depth_constants(f, S);
            img = (size_t)b * S * S;
        }
    }
    walk_owned_pixels<2>(cd, any_box ? cd.n : 0, fn, face_index_map + img, S, sub, L, s_queue, [&](int off) {
        const size_t p = img + (size_t)off;
        const float depth = depth_map[p];
        const float w[3] = {weight_map[3 * p], weight_map[3 * p + 1], weight_map[3 * p + 2]};
        const float gd = g_depth[p];
        const float depth2 = depth * depth;
        // :824-827
#pragma unroll
        for (int k = 0; k < 3; k++) acc[3 * k + 2] += gd * w[k] * depth2 / dc.zz[k];
        // :830-837
        float tmp[3] = {dc.tmp[0], dc.tmp[1], dc.tmp[2]};
        if (face_inv_map) {  // the reference's per-pixel residual: its values, its divisions
            tmp[0] = tmp[1] = tmp[2] = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int l = 0; l < 3; l++) tmp[k] += -face_inv_map[9 * p + 3 * l + k] / f[3 * l + 2];
        }
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int l = 0; l < 2; l++) acc[3 * k + l] += -gd * tmp[l] * w[k] * depth2 * (float)S / 2.0f;
    });
    if (__ballot(any_box) == 0ull) return;  // whole wave has nothing to add
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] = row16_sum_last(acc[k]);
    if (face_ok && any_box && sub == 15) {
        float *gf = grad_faces + (size_t)gi * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) gf[k] += acc[k];
    }
}

// k_backward_big's grid: one workgroup per range of 256 faces (or list slots), times as many workgroups per range (z) as it
// takes to put ~4096 workgroups on the chip -- they share out the range's big faces.  (Small launches: a workgroup per 64
// faces of the call, at least 1024 -- with nothing to do, as on a fine mesh, the kernel costs what dispatching it costs.)
inline dim3 big_grid(bool listed, int B, int F)
{
    const size_t n = (size_t)B * F;
    const dim3 g = listed ? dim3((unsigned)((F + 255) / 256), (unsigned)B) : dim3((unsigned)((n + 255) / 256));
    const size_t ranges = (size_t)g.x * g.y;
    const size_t target = n / 64 < 1024 ? 1024 : (n / 64 > 4096 ? 4096 : n / 64);
    const size_t z = target / ranges;
    return dim3(g.x, g.y, (unsigned)(z < 1 ? 1 : (z > 64 ? 64 : z)));
}

}  // namespace

// ====================================================================================================
int nr::run_backward_textures(const int32_t *face_index_map, const float *sampling_weight_map,
                              const int32_t *sampling_index_map, const float *faces, const float *faces_z_ref,
                              const float *weight_map,
                              const float *depth_map, const float *grad_rgb_map, float *grad_textures, int B, int F,
                              int S, int ts, double eps, int flags, const int *vis_list, const int *vis_count,
                              hipStream_t st, const float *g_depth, float *grad_faces, int *depth_done,
                              const double *k6_scratch, const int *slot_of, int *k6_finalized, const FaceLight &lit,
                              bool prefilled, int phase, const LineSetupArgs *ls, const int *zero_slot_of)
{
    if (depth_done) *depth_done = 0;
    if (k6_finalized) *k6_finalized = 0;
    if (!face_index_map || !grad_rgb_map || !grad_textures || !faces) return NR_E_NULL;
    if ((sampling_index_map == nullptr) != (sampling_weight_map == nullptr)) return NR_E_MODE;
    if (!sampling_weight_map && (!weight_map || !depth_map)) return NR_E_NULL;
    if (int e = check_sizes(B, F, S)) return e;
    if (ts < 2 || ts > 1024) return NR_E_SIZE;
    if (lit.light && sampling_weight_map) return NR_E_MODE;  // (the taps are recomputed in the original cube's layout)
    const int fix = (flags & NR_FLAG_FIX_TEXTURE_BATCH_Z) ? 1 : 0;
    const float *zbase = faces_z_ref ? faces_z_ref : faces;  // :389 reads batch 0 of the GLOBAL batch (see nr_hip.h)
    const int n = B * F;
    const size_t n_tex = (size_t)ts * ts * ts * 3;
    if (ts > 13) vis_list = nullptr;  // the atomic fallback walks pixels, not faces
    if (ts > 8 || sampling_weight_map || !grad_faces) g_depth = nullptr;  // K8 is fused only into the one-wave-per-group gathers
    // static taps (TS2 path): valid when the clamp of rasterize.py:402 keeps every index float below 1, i.e. when
    // (ts - 1) - eps still rounds below ts - 1 in float32 (eps > 2^-25); otherwise a coordinate can be exactly 1.0
    const bool ts2_static = ts == 2 && (float)(1.0 - eps) < 1.0f;
    if (ts == 2 && !ts2_static) g_depth = nullptr;
    if (g_depth && depth_done) *depth_done = 1;
    // K6's finish folded into the gather (see the kernel): needs the lists, a face-per-group kernel and somewhere to store
    // (phase 2: the finish rides in k_backward_big's launch instead, on top of what the gather of phase 1 left in grad_faces)
    const double *finish_k6 = (phase == 2 && vis_list && k6_scratch && grad_faces && ts <= 8) ? k6_scratch : nullptr;
    const bool fold = phase != 2 && vis_list && k6_scratch && slot_of && grad_faces && ts <= 13;
    if (!fold) k6_scratch = nullptr, slot_of = nullptr;
    if ((fold || finish_k6) && k6_finalized) *k6_finalized = 1;
    // The line setup rides in the gather's launch (k_setup_gather) when there is a face-walking gather on the lists and the two
    // fit one launch's dynamic LDS; the zeros of grad_textures ride along too (plain path; zero_slot_of: K6's face -> position
    // table), else they are filled in front.
    const bool face_kernel = ts <= 13;  // (above: the per-pixel scatter)
    const size_t gather_lds = (ts2_static && !sampling_weight_map) ? 0 : (size_t)(256 / (ts <= 5 ? 16 : (ts <= 8 ? 64 : 256))) * n_tex * sizeof(double);
    // (the shared launch's dynamic LDS is the larger of the two bodies' and its static arrays -- both bodies', ~6 KB -- come on top;
    // without a hipFuncSetAttribute call a launch may use 64 KB in all, so the dynamic part is kept to 40 KB: texture_size 12
    // (41.5 KB of gather accumulators) and rasters whose line setup needs more than 32 KB take the two launches)
    const bool fuse = ls && vis_list && face_kernel && phase != 2 && ls->lds_bytes <= 32768 && gather_lds <= 40960;
    const bool zero_in_launch = fuse && !lit.light && zero_slot_of && !prefilled && ((size_t)grad_textures & 15) == 0;
    if (phase != 2) {
    if (lit.light) {
        // original cubes: a face and its reversed copy share one, so only the one that owns a pixel stores; the rest is zero
        int e = prefilled ? 0 : fill_bytes(grad_textures, 0, (size_t)B * lit.tex_faces * n_tex * sizeof(float), st);
        if (e == 0 && lit.grad_light) e = fill_bytes(lit.grad_light, 0, (size_t)n * 3 * sizeof(float), st);
        if (e != 0) return e;
    } else if (vis_list && !prefilled && !zero_in_launch) {
        // only visible faces are visited: everything else is zero.  (Round 4 tried to spare the listed faces' cubes, which the
        // gathers store completely -- config 5: a 4 GB fill, 565 us at 7.1 TB/s -- with a fill predicated on K6's face ->
        // position table: 622-787 us in four forms, the division / table load / predicate cost more than the ~10 % of the
        // bytes they save; as a launch of its own the plain fill stays.)
        const int e = fill_bytes(grad_textures, 0, (size_t)n * n_tex * sizeof(float), st);
        if (e != 0) return e;
    }
    }
    bool setup_launched = false;
    // (the kernels take the per-face light mode as a template parameter: see k_backward_textures_face)
    auto launch = [&](auto lit_mode) {
    constexpr bool LITV = decltype(lit_mode)::value;
    if (phase != 2 && ts <= 13) {
        const bool st2 = ts2_static && !sampling_weight_map;
        const int L = st2 ? 16 : (ts <= 5 ? 16 : (ts <= 8 ? 64 : 256));
        const int per = 256 / L;
        const size_t lds = st2 ? 0 : (size_t)per * n_tex * sizeof(double);
        const dim3 grid = vis_list ? dim3((unsigned)((F + per - 1) / per), (unsigned)B) : dim3((unsigned)((n + per - 1) / per));
        const bool depth = g_depth && L <= 64;  // K8 rides along in the one-wave-per-group gathers
        const FaceGatherArgs ga = {face_index_map, sampling_weight_map, sampling_index_map, faces, zbase, weight_map, depth_map,
                                   grad_rgb_map, grad_textures, n, F, S, ts, eps, fix, L, vis_list, vis_count,
                                   depth ? g_depth : (const float *)nullptr,
                                   depth ? grad_faces : (fold ? grad_faces : (float *)nullptr), k6_scratch, slot_of, lit};
        auto go = [&](auto ts2, auto dep) {
            constexpr bool T = decltype(ts2)::value, D = decltype(dep)::value;
            if (fuse) {
                ZeroArgs z = {grad_textures, zero_slot_of, F, (int)n_tex, 0, 0u};
                if (zero_in_launch) {
                    z.vec = n_tex % 4 == 0;
                    z.epf = z.vec ? (int)(n_tex / 4) : (int)n_tex;
                    z.wgs = (unsigned)(((size_t)F * z.epf + 2047) / 2048);
                }
                const size_t both = lds > ls->lds_bytes ? lds : ls->lds_bytes;
                hipLaunchKernelGGL((k_setup_gather<T, D, LITV>), dim3(ls->grid_x + grid.x + z.wgs, (unsigned)B), dim3(256), both, st,
                                   *ls, ga, z, grid.x);
                setup_launched = true;
            } else {
                hipLaunchKernelGGL((k_backward_textures_face<T, D, LITV>), grid, dim3(256), lds, st, ga);
            }
        };
        using T = std::true_type;
        using N = std::false_type;
        if (st2) { if (depth) go(T(), T()); else go(T(), N()); }
        else { if (depth) go(N(), T()); else go(N(), N()); }
    }
    if (ts <= 8 && phase != 1) {
        // faces the gathers above left out (more than BIG_PX candidates): a workgroup each
        const dim3 grid = big_grid(vis_list != nullptr, B, F);
        const bool st2 = ts2_static && !sampling_weight_map;
        const size_t lds = st2 ? 0 : n_tex * sizeof(double);
#define NR_BIG(T, D)                                                                                                    \
    hipLaunchKernelGGL((k_backward_big<T, D, LITV>), grid, dim3(256), lds, st, face_index_map, sampling_weight_map,   \
                       sampling_index_map, (const float *)nullptr, faces, zbase, weight_map, depth_map, grad_rgb_map, \
                       grad_textures, n, F, S, ts, eps, fix, vis_list, vis_count, D ? g_depth : (const float *)nullptr, \
                       (D || finish_k6) ? grad_faces : (float *)nullptr, (const unsigned char *)nullptr, lit, finish_k6)
        if (st2) { if (g_depth) NR_BIG(2, true); else NR_BIG(2, false); }
        else { if (g_depth) NR_BIG(1, true); else NR_BIG(1, false); }
#undef NR_BIG
    }
    };
    if (lit.light) launch(std::true_type()); else launch(std::false_type());
    if (phase == 2) return launch_status();
    if (ls && !setup_launched)  // the line setup as a launch of its own (no face-walking gather to share one with)
        if (int rc = run_line_setup(*ls, st)) return rc;
    if (ts > 13 && lit.light) {
        // (grad_textures and grad_light were zero-filled above)
        const size_t np = (size_t)B * S * S;
        hipLaunchKernelGGL(k_backward_textures_atomic_lit, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, face_index_map,
                           faces, zbase, weight_map, depth_map, grad_rgb_map, grad_textures, F, S, ts, eps, fix, np, lit);
    } else if (ts > 13) {
        // huge cubes: the reference's per-pixel scatter with hardware atomics
        const int e = fill_bytes(grad_textures, 0, (size_t)n * n_tex * sizeof(float), st);
        if (e != 0) return e;
        const size_t np = (size_t)B * S * S;
        hipLaunchKernelGGL(k_backward_textures_atomic, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st,
                           face_index_map, sampling_weight_map, sampling_index_map, faces, zbase, weight_map, depth_map,
                           grad_rgb_map, grad_textures, F, S, ts, eps, fix, np);
    }
    return launch_status();
}

int nr::run_backward_depth_map(const float *faces, const float *depth_map, const int32_t *face_index_map,
                               const float *face_inv_map, const float *weight_map, const float *grad_depth_map,
                               float *grad_faces, int B, int F, int S, const int *vis_list, const int *vis_count,
                               hipStream_t st, const unsigned char *visible)
{
    if (!faces || !depth_map || !face_index_map || !weight_map || !grad_depth_map || !grad_faces) return NR_E_NULL;
    if (int e = check_sizes(B, F, S)) return e;
    const int n = B * F;
    const dim3 grid = vis_list ? dim3((unsigned)((F + 15) / 16), (unsigned)B) : dim3((unsigned)((n + 15) / 16));
    hipLaunchKernelGGL(k_backward_depth_face, grid, dim3(256), 0, st, faces, depth_map, face_index_map, face_inv_map,
                       weight_map, grad_depth_map, grad_faces, n, F, S, vis_list, vis_count, visible);
    const dim3 grid_big = big_grid(vis_list != nullptr, B, F);
    hipLaunchKernelGGL((k_backward_big<0, true, false>), grid_big, dim3(256), 0, st, face_index_map, (const float *)nullptr,
                       (const int32_t *)nullptr, face_inv_map, faces, faces, weight_map, depth_map, (const float *)nullptr,
                       (float *)nullptr, n, F, S, 2, 0.0, 0, vis_list, vis_count, grad_depth_map, grad_faces, visible, FaceLight(),
                       (const double *)nullptr);
    return launch_status();
}

NR_API int nr_backward_textures(const int32_t *face_index_map, const float *sampling_weight_map,
                                const int32_t *sampling_index_map, const float *faces, const float *faces_z_ref,
                                const float *weight_map,
                                const float *depth_map, const float *grad_rgb_map, float *grad_textures, int32_t B,
                                int32_t F, int32_t S, int32_t ts, double eps, int32_t flags, void *stream)
{
    return run_backward_textures(face_index_map, sampling_weight_map, sampling_index_map, faces, faces_z_ref, weight_map,
                                 depth_map, grad_rgb_map, grad_textures, B, F, S, ts, eps, flags, nullptr, nullptr,
                                 (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}

NR_API int nr_backward_depth_map(const float *faces, const float *depth_map, const int32_t *face_index_map,
                                 const float *face_inv_map, const float *weight_map, const float *grad_depth_map,
                                 float *grad_faces, int32_t B, int32_t F, int32_t S, void *stream)
{
    return run_backward_depth_map(faces, depth_map, face_index_map, face_inv_map, weight_map, grad_depth_map,
                                  grad_faces, B, F, S, nullptr, nullptr, (hipStream_t)stream, nullptr);
}
