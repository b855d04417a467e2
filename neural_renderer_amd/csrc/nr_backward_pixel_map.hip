// nr_backward_pixel_map.hip -- K6, the approximate gradient of rgb / alpha w.r.t. vertex x, y
// (reference Rasterize.backward_pixel_map_gpu, rasterize.py:517-748) + its C-ABI entry point.
#include "nr_device.h"

using namespace nr;

namespace {

// --------------------------------------------------------------------------------------------------
// B1: backward_pixel_map (rasterize.py:517-748).
//
// Work decomposition.  The reference runs ONE thread per face through 3 edges x 2 axes x every integer
// column/row d0 crossed by the edge x two pixel sweeps along d1 (an "in" sweep from the edge to the
// opposite edge, an "out" sweep from the edge to the image border).  All sweeps of one (edge, axis) item
// feed the same two outputs (vertex pi[0] and pi[1], coordinate 1 - axis), so here
//   * one wave owns one face; its lanes form six groups of GRP = 10 lanes, one group per (edge, axis) item
//     (lanes 60..63 idle);
//   * lines are handled in batches of GRP: phase A sets up one line per lane (crossing point, in/out pixels,
//     reference colours, the two distance coefficients, the sweep ranges) and parks it in LDS; phase B lets
//     lane `sub` of a group visit pixels from + sub, + GRP, ... of EVERY sweep of its group.  Phase B is a
//     flattened per-lane state machine (fetch next sweep | visit one pixel), so a group with long sweeps
//     does not stall the other groups line by line: wave time = max over groups of their total work;
//   * per-lane partial sums are kept in double and reduced ONCE per item (not per sweep); the six results
//     of a face are exchanged between group leaders and STORED: no atomics, no zero fill, deterministic.
// Every per-pixel term uses the reference's arithmetic (same operations, same precision); only the order
// of the additions differs, and the sums are carried in double so that the result is the correctly
// rounded sum of the reference's terms (the reference's own serial float sum carries more rounding noise).
constexpr int GRP = 10;
constexpr int NGRP = 6;

struct __attribute__((aligned(16))) LineRec {
    int in_rng;   // from | to << 16 (from > to: empty)
    int out_rng;  // from | to << 16
    int base;     // pixel index of (d0, d1 = 0)
    int flags;    // 2: has out sweep, 4: has0 (p1x != d0), 8: has1 (p0x != d0)
    float cross, c0, c1, pad;
    float in_c[4];   // alpha, r, g, b of the in pixel  (reference colour of the OUT sweep)
    float out_c[4];  // alpha, r, g, b of the out pixel (reference colour of the IN sweep)
};

template <bool RGB, bool ALPHA>
__global__ __launch_bounds__(WAVE) void k_backward_pixel_map(
    const float *__restrict__ faces, const int32_t *__restrict__ fi_map, const float *__restrict__ rgb_map,
    const float *__restrict__ alpha_map, const float *__restrict__ g_rgb, const float *__restrict__ g_alpha,
    float *__restrict__ grad_faces, int F, int S, double eps, int axis_mask)
{
    __shared__ LineRec recs[NGRP][GRP];

    const int lane = threadIdx.x;
    const int gi = blockIdx.x;  // global face index b * F + fn
    const int b = gi / F, fn = gi - b * F;
    const float *f = faces + (size_t)gi * 9;
    const float fx[3] = {f[0], f[3], f[6]}, fy[3] = {f[1], f[4], f[7]};
    float *out = grad_faces + (size_t)gi * 9;
    if (is_backside(fx[0], fy[0], fx[1], fy[1], fx[2], fy[2])) {  // :540 (grad_faces was zero-filled, :851)
        if (lane < 9) out[lane] = 0.0f;
        return;
    }
    const int g = lane / GRP, sub = lane - g * GRP;
    const bool lane_on = g < NGRP;
    const int edge = (g >> 1) % 3, axis = g & 1;
    const double s_d = (double)S, two_over_s = 2.0 / (double)S;
    const bool s_pow2 = (S & (S - 1)) == 0;

    // ---- item setup: rasterize.py:543-569
    const float fs = (float)S;
    const int i0 = edge, i1 = (edge + 1) % 3, i2 = (edge + 2) % 3;
    const float ppx[3] = {to_pixel(fx[i0], fs), to_pixel(fx[i1], fs), to_pixel(fx[i2], fs)};
    const float ppy[3] = {to_pixel(fy[i0], fs), to_pixel(fy[i1], fs), to_pixel(fy[i2], fs)};
    // p[num][dim] = pp[num][(dim + axis) % 2]: axis 1 swaps the roles of x and y (:556)
    const float p0x = axis ? ppy[0] : ppx[0], p0y = axis ? ppx[0] : ppy[0];
    const float p1x = axis ? ppy[1] : ppx[1], p1y = axis ? ppx[1] : ppy[1];
    const float p2x = axis ? ppy[2] : ppx[2], p2y = axis ? ppx[2] : ppy[2];
    int direction;
    if (axis == 0) direction = (p0x < p1x) ? -1 : 1; else direction = (p0x < p1x) ? 1 : -1;  // :559-564
    const int d0_from = (int)fmax((double)ceilf(fminf(p0x, p1x)), 0.0);      // :568
    const int d0_to = (int)fmin((double)fmaxf(p0x, p1x), S - 1.0);           // :569
    int n_lines = 0;
    // p0x == p1x: the only possible d0 equals both, so both contributions are skipped (:648, :653)
    if (lane_on && p0x != p1x && d0_to >= d0_from && ((axis_mask >> axis) & 1)) n_lines = d0_to - d0_from + 1;
    const float slope = (p1y - p0y) / (p1x - p0x);  // :573, invariant along the edge
    // strides of d0 / d1 in the row-major maps (:587-593)
    const int sd0 = axis ? S : 1, sd1 = axis ? 1 : S;
    const int img_base = b * S * S;

    int max_lines = n_lines;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) max_lines = max(max_lines, __shfl_xor(max_lines, o, WAVE));

    double G0 = 0.0, G1 = 0.0;  // running sums for vertex pi[0] / pi[1], coordinate (1 - axis)

    for (int batch0 = 0; batch0 < max_lines; batch0 += GRP) {
        // ---------------- phase A: one line per lane -> LDS
        if (lane_on) {
            LineRec r;
            r.in_rng = 1; r.out_rng = 1; r.base = 0; r.flags = 0;  // from 1 > to 0: empty
            r.cross = r.c0 = r.c1 = r.pad = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; k++) { r.in_c[k] = 0.0f; r.out_c[k] = 0.0f; }
            const int it = batch0 + sub;
            if (it < n_lines) {
                const int d0 = d0_from + it;
                const float d0f = (float)d0;
                const float d1_cross = slope * (d0f - p0x) + p0y;                                     // :573
                const int d1_in = (0 < direction) ? (int)floorf(d1_cross) : (int)ceilf(d1_cross);     // :574
                const int d1_out = d1_in + direction;                                                 // :575
                if (!(d1_in < 0 || S <= d1_in) && !(d1_out < 0 || S <= d1_out)) {                     // :578-579
                    const int line_base = img_base + d0 * sd0;
                    const int idx_in = line_base + d1_in * sd1, idx_out = line_base + d1_out * sd1;
                    if (ALPHA) { r.in_c[0] = alpha_map[idx_in]; r.out_c[0] = alpha_map[idx_out]; }    // :594-597
                    if (RGB) {                                                                        // :598-601
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            r.in_c[1 + k] = rgb_map[3 * (size_t)idx_in + k];
                            r.out_c[1 + k] = rgb_map[3 * (size_t)idx_out + k];
                        }
                    }
                    const bool is_in_fn = (fi_map[idx_in] == fn);                                     // :604
                    int flags = 0;
                    if (p1x != d0f) flags |= 4;
                    if (p0x != d0f) flags |= 8;
                    r.c0 = (p1x - p0x) / (p1x - d0f);  // :649 leading factor, invariant along the sweep
                    r.c1 = (p1x - p0x) / (d0f - p0x);  // :654
                    if (is_in_fn) {                     // :606-609
                        const int lim = (0 < direction) ? S - 1 : 0;
                        const int o_from = max(min(d1_out, lim), 0), o_to = min(max(d1_out, lim), S - 1);
                        r.out_rng = o_from | (o_to << 16);
                        flags |= 2;
                    }
                    float d0_cross2;                    // :665-672
                    if ((d0f - p0x) * (d0f - p2x) < 0)
                        d0_cross2 = (p2y - p0y) / (p2x - p0x) * (d0f - p0x) + p0y;
                    else
                        d0_cross2 = (p1y - p2y) / (p1x - p2x) * (d0f - p2x) + p2y;
                    const int lim2 = (0 < direction) ? (int)ceilf(d0_cross2) : (int)floorf(d0_cross2);
                    const int i_from = max(min(d1_in, lim2), 0), i_to = min(max(d1_in, lim2), S - 1);
                    r.in_rng = i_from | (i_to << 16);
                    r.base = line_base;
                    r.flags = flags;
                    r.cross = d1_cross;
                }
            }
            recs[g][sub] = r;
        }
        __syncthreads();

        // ---------------- phase B: flattened walk over (line, sweep, pixel) of the own group
        const int nl = lane_on ? min(max(n_lines - batch0, 0), GRP) : 0;
        int l = -1, ph = 0, d1 = 0, d1_end = -1, base = 0, cur_flags = 0;
        float cross = 0, c0 = 0, c1 = 0, ref_a = 0, ref_r = 0, ref_g = 0, ref_b = 0;
        while (__ballot(l < nl) != 0ull) {
            if (d1 > d1_end && l < nl) {  // fetch the next sweep: in(l) -> out(l) -> in(l + 1) ...
                if (ph == 0 && (cur_flags & 2)) ph = 1; else { ph = 0; ++l; }
                if (l < nl) {
                    const LineRec *r = &recs[g][l];
                    const int4 h = *reinterpret_cast<const int4 *>(r);
                    const float4 q = *reinterpret_cast<const float4 *>(&r->cross);
                    const float4 col = *reinterpret_cast<const float4 *>(ph == 0 ? r->out_c : r->in_c);
                    const int rng = ph == 0 ? h.x : h.y;
                    d1 = (rng & 0xffff) + sub;
                    d1_end = rng >> 16;
                    base = h.z;
                    cur_flags = h.w;
                    cross = q.x; c0 = q.y; c1 = q.z;
                    ref_a = col.x; ref_r = col.y; ref_g = col.z; ref_b = col.w;
                }
            }
            if (l < nl && d1 <= d1_end) {  // one pixel visit: rasterize.py:630-657 (out) / :697-728 (in)
                const int idx = base + d1 * sd1;
                bool skip = (ph == 0) && (fi_map[idx] != fn);  // :707
                float diff = 0.0f;
                if (!skip) {
                    if (ALPHA) diff += (alpha_map[idx] - ref_a) * g_alpha[idx];
                    if (RGB) {
                        const float *pp = rgb_map + 3 * (size_t)idx;
                        const float *gg = g_rgb + 3 * (size_t)idx;
                        diff += (pp[0] - ref_r) * gg[0];
                        diff += (pp[1] - ref_g) * gg[1];
                        diff += (pp[2] - ref_b) * gg[2];
                    }
                }
                if (!skip && !(diff <= 0.0f)) {  // :647 / :717
                    const float t = (float)d1 - cross;
                    if (cur_flags & 4) {  // :648-652
                        const float ct = c0 * t;
                        float dist = (float)(s_pow2 ? (double)ct * two_over_s : (double)ct * 2.0 / s_d);
                        dist = (0.0f < dist) ? (float)((double)dist + eps) : (float)((double)dist - eps);
                        G0 -= (double)(diff / dist);
                    }
                    if (cur_flags & 8) {  // :653-657
                        const float ct = c1 * t;
                        float dist = (float)(s_pow2 ? (double)ct * two_over_s : (double)ct * 2.0 / s_d);
                        dist = (0.0f < dist) ? (float)((double)dist + eps) : (float)((double)dist - eps);
                        G1 -= (double)(diff / dist);
                    }
                }
                d1 += GRP;
            }
        }
        __syncthreads();
    }

    // ---- reduce the GRP partial sums of each group (leader = sub 0), exchange between leaders, store.
    //      component (vertex v, coord 1 - axis) = G0 of item (edge v, axis) + G1 of item (edge v + 2 mod 3, axis)
    //      (pi[] of :547, :651, :656)
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        const double t0 = __shfl_down(G0, o, WAVE), t1 = __shfl_down(G1, o, WAVE);
        if (sub + o < GRP) { G0 += t0; G1 += t1; }
    }
    const int partner = (2 * ((edge + 2) % 3) + axis) * GRP;
    const double G1p = __shfl(G1, partner, WAVE);
    if (lane_on && sub == 0) {
        out[3 * edge + (1 - axis)] = (float)(G0 + G1p);
        if (axis == 0) out[3 * edge + 2] = 0.0f;  // K6 never touches z
    }
}

}  // namespace

// ====================================================================================================
NR_API size_t nr_backward_workspace_bytes(int32_t B, int32_t F, int32_t S, int32_t return_rgb, int32_t return_alpha)
{
    (void)return_rgb; (void)return_alpha;
    if (check_sizes(B, F, S)) return 0;
    return 0;  // the first-generation kernel sweeps the row-major maps directly
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
    if ((size_t)B * S * S > 0x7fffffffull / 3) return NR_E_SIZE;  // int32 pixel indexing inside the kernel
    const int n = B * F;
    const char *am = getenv("NR_K6_AXIS_MASK");  // experiment knob (default: both axes)
    const int axis_mask = am ? atoi(am) : 3;
    const dim3 grid((unsigned)n), block(WAVE);
    hipStream_t st = (hipStream_t)stream;
    if (return_rgb && return_alpha)
        hipLaunchKernelGGL((k_backward_pixel_map<true, true>), grid, block, 0, st, faces, face_index_map, rgb_map,
                           alpha_map, grad_rgb_map, grad_alpha_map, grad_faces, F, S, eps, axis_mask);
    else if (return_rgb)
        hipLaunchKernelGGL((k_backward_pixel_map<true, false>), grid, block, 0, st, faces, face_index_map, rgb_map,
                           alpha_map, grad_rgb_map, grad_alpha_map, grad_faces, F, S, eps, axis_mask);
    else
        hipLaunchKernelGGL((k_backward_pixel_map<false, true>), grid, block, 0, st, faces, face_index_map, rgb_map,
                           alpha_map, grad_rgb_map, grad_alpha_map, grad_faces, F, S, eps, axis_mask);
    return launch_status();
}

